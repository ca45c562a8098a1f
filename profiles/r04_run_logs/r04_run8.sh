#!/bin/bash
# round 4, GPU run 7: early tile requests, symmetric Gram product
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_8
mkdir -p $OUT /tmp/w
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_kernels.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q -s --deselect tests/test_gpu_kernels.py > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 1000 --steady-burn-in 400 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
for r in 10 1; do
  rm -rf /tmp/w/kt$r; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt$r -o t -- python3 $R/bench.py --reduction $r --steps 300 --warmup 200 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/kt$r.log 2>&1
  python3 $R/scripts/prof_summary.py $(find /tmp/w/kt$r -name "*.db" | head -1) 0.5 > $OUT/kernel_trace_r$r.txt 2>&1
done
ls -la $OUT
