#!/bin/bash
O=gpurun_out/r05_35; mkdir -p $O /tmp/w; R=$PWD
timeout 600 python scripts/ab_stats_resident.py > $O/ab.txt 2>&1; tail -12 $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt -o t -- python3 $R/bench.py --reduction 1 --steps 300 --warmup 200 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/kt.log 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/w/kt -name "*.db" | head -1) 0.5 > $R/$O/kt_r1.txt 2>&1
head -6 $R/$O/kt_r1.txt
rm -rf /tmp/w/c5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c5 -o t -- python3 $R/scripts/diag_hcp_shape.py > /tmp/w/c5.log 2>&1
{ tail -8 /tmp/w/c5.log; python3 $R/scripts/prof_summary.py $(find /tmp/w/c5 -name "*.db" | head -1) 0.3; } > $R/$O/kt_c5.txt 2>&1
grep -A3 "^kernel" $R/$O/kt_c5.txt | head -5
