#!/bin/bash
# round 4, GPU run 10: two tile rows per LDS instruction in the chain wave
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_10
mkdir -p $OUT /tmp/w
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_kernels.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
timeout 1800 python3 -m pytest tests/test_gpu_step.py -m gpu -q -x > $OUT/pytest_step.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_step.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 1000 --steady-burn-in 400 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
