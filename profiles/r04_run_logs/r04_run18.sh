#!/bin/bash
# gathers of bcd_setup_kernel / prep_kernel with eight rows per workgroup: new build against HEAD's (build_ab/head), same box
cd /root/repo
export TMPDIR=/tmp
cp modl_amd/libmodl_hip.so /tmp/new.so
for rep in 1 2; do
  cp build_ab/head/libmodl_hip.so modl_amd/libmodl_hip.so; echo head; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
  cp /tmp/new.so modl_amd/libmodl_hip.so; echo new; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
