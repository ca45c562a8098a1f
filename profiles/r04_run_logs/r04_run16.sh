#!/bin/bash
# reduction 10 (32 workgroups): the Gram accumulator sharded 4 / 2 ways against one accumulator (product build), same box
cd /root/repo
export TMPDIR=/tmp
cp modl_amd/libmodl_hip.so /tmp/new.so
for rep in 1 2; do
  for v in s4 s2; do cp build_ab/$v/libmodl_hip.so modl_amd/libmodl_hip.so; echo $v; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1; done
  cp /tmp/new.so modl_amd/libmodl_hip.so; echo product; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1
done
