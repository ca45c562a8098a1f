#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 300 python scripts/diag_stamps_ahead.py > gpurun_out/stamps_ahead.txt 2>&1
cat gpurun_out/stamps_ahead.txt
timeout 600 python scripts/ab_bcd_ahead.py 10 > gpurun_out/ab_ahead_r10.txt 2>&1
tail -10 gpurun_out/ab_ahead_r10.txt
timeout 600 python scripts/ab_bcd_ahead.py 1 > gpurun_out/ab_ahead_r1.txt 2>&1
tail -10 gpurun_out/ab_ahead_r1.txt
