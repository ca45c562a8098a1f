#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_30; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 0 --no-breakdown > $O/line.json 2>/tmp/kt.err
python3 $GRAFT_REPO_ROOT/scripts/dump_driver_timeline.py $(find /tmp/kt -name "*.db" | head -1) > $O/timeline.txt 2>&1; tail -24 $O/timeline.txt
