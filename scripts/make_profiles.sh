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
DBS=""
for c in "SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  n=$(echo $c | tr ' ' '_'); rm -rf /tmp/w/sq_$n
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/w/sq_$n -o t -- python3 $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline > /tmp/w/sq_$n.log 2>&1
  DBS="$DBS $(find /tmp/w/sq_$n -name '*.db' | head -1)"
done
python3 $R/scripts/pmc_summary.py $DBS > $OUT/${TAG}_pmc_sq_counters.txt 2>&1
ls -la $OUT | grep $TAG
