#!/bin/bash
# the tail of the block launch (Gram record + apply): new build against the build of commit 6450b39 (build_ab/base), same box
cd /root/repo
export TMPDIR=/tmp
timeout 300 python scripts/diag_stamps.py > gpurun_out/bcd_block_stamps_new.txt 2>&1
grep -v amdgpu.ids gpurun_out/bcd_block_stamps_new.txt
cp modl_amd/libmodl_hip.so /tmp/new.so
for rep in 1 2; do
  cp build_ab/base/libmodl_hip.so modl_amd/libmodl_hip.so; echo base; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
  cp /tmp/new.so modl_amd/libmodl_hip.so; echo new; timeout 300 python scripts/ab_minibatch.py 10 2>&1 | tail -1; timeout 300 python scripts/ab_minibatch.py 1 2>&1 | tail -1
done
timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -5
