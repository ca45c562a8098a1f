#!/bin/bash
O=gpurun_out/r05_25; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_kernels.py -q -x -k "headline or chunk_call or trajector or config1 or cd_" > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 200 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/kt.log 2>&1; python3 $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/kt -name "*.db" | head -1) 0.5 > $GRAFT_REPO_ROOT/$O/trace.txt 2>&1; head -9 $GRAFT_REPO_ROOT/$O/trace.txt
