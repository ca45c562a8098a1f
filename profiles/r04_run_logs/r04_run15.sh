#!/bin/bash
# r = 1: 96-feature workgroups (105 of them) against 64-feature ones (157): build_ab/rt3 against the product build, same box
cd /root/repo
export TMPDIR=/tmp
cp modl_amd/libmodl_hip.so /tmp/new.so
for rep in 1 2; do
  cp build_ab/rt3/libmodl_hip.so modl_amd/libmodl_hip.so; echo rt3; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
  cp /tmp/new.so modl_amd/libmodl_hip.so; echo product; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
done
