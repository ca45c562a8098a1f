#!/bin/bash
# look-ahead mode of the blocked dictionary update: A/B + the step tests under it
cd /root/repo
export TMPDIR=/tmp
timeout 600 python scripts/ab_bcd_ahead.py 10 > gpurun_out/ab_ahead_r10.txt 2>&1
tail -12 gpurun_out/ab_ahead_r10.txt
timeout 600 python scripts/ab_bcd_ahead.py 1 > gpurun_out/ab_ahead_r1.txt 2>&1
tail -12 gpurun_out/ab_ahead_r1.txt
MODL_TEST_BCD_ACC=2 timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -m gpu > gpurun_out/ahead_tests.txt 2>&1
tail -15 gpurun_out/ahead_tests.txt
