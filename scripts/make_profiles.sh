#!/bin/bash
# Regenerates the round's profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/make_profiles.sh r01_m
# writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r01_x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/w
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench.err
rm -rf /tmp/w/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt -o t -- python3 $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline > /tmp/w/kt.log 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/w/kt -name "*.db" | head -1) 0.5 > $OUT/${TAG}_kernel_trace_bench_r10.txt 2>&1
DBS=""
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/w/pmc_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/w/pmc_$c -o t -- python3 $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline > /tmp/w/pmc_$c.log 2>&1
  DBS="$DBS $(find /tmp/w/pmc_$c -name '*.db' | head -1)"
done
python3 $R/scripts/pmc_summary.py $DBS --json $OUT/${TAG}_pmc_hbm_traffic.json > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
SQC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
rm -rf /tmp/w/sq; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/sq -o t -- python3 $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline > /tmp/w/sq.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $SQC"; echo "(averages over the counter-instance rows of all dispatches; SQ_* cycle counters are in quad-cycles; use the RATIOS)"; echo; python3 $R/scripts/pmc_summary.py $(find /tmp/w/sq -name '*.db' | head -1); } > $OUT/${TAG}_pmc_sq_counters.txt 2>&1
ls -la $OUT | grep $TAG
