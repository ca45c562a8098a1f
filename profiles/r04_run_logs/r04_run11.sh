#!/bin/bash
# round 4, GPU run 11: block-kernel stamps out of the product build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_11
mkdir -p $OUT /tmp/w
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python3 scripts/diag_stamps.py > $OUT/diag_stamps.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 1000 --steady-burn-in 400 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/w/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt -o t -- python3 $R/bench.py --steps 300 --warmup 200 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/kt.log 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/w/kt -name "*.db" | head -1) 0.5 > $OUT/kernel_trace_r10.txt 2>&1
ls -la $OUT
