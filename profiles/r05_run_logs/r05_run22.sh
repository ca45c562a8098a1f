#!/bin/bash
# SQ counters of the statistics product BEFORE (MODL_DEBUG_STATS_RESIDENT = 0: 32 x 32 tiles at reduction 1, k-wide tiles at config 5)
# and AFTER, LDS bank conflicts included
R=$PWD; O=$R/gpurun_out; mkdir -p /tmp/w
cd /tmp && export TMPDIR=/tmp
SQC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for v in 0 1; do
  rm -rf /tmp/w/a$v; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/a$v -o t -- python3 $R/bench.py --reduction 1 --steps 100 --warmup 40 --no-cpu-baseline --steady-steps 0 --no-breakdown --debug-set 10=$v > /tmp/w/a$v.log 2>&1
  { echo "rocprofv3 --kernel-trace --pmc $SQC -- python3 bench.py --reduction 1 --steps 100 --warmup 40 --debug-set 10=$v"; python3 $R/scripts/pmc_summary.py $(find /tmp/w/a$v -name '*.db' | head -1) | grep -E "^kernel|gemm_stats|gemm_dense_pair"; } > $O/r05_pmc_sq_stats_r1_resident$v.txt 2>&1
  rm -rf /tmp/w/b$v; timeout 600 rocprofv3 --kernel-trace --pmc $SQC -d /tmp/w/b$v -o t -- python3 $R/bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 --no-cpu-baseline --no-breakdown --debug-set 10=$v > /tmp/w/b$v.log 2>&1
  { echo "rocprofv3 --kernel-trace --pmc $SQC -- python3 bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 --debug-set 10=$v"; python3 $R/scripts/pmc_summary.py $(find /tmp/w/b$v -name '*.db' | head -1) | grep -E "^kernel|gemm_stats|gemm_dense_pair"; } > $O/r05_pmc_sq_stats_c5_resident$v.txt 2>&1
done
tail -20 $O/r05_pmc_sq_stats_c5_resident0.txt | cut -c1-160
