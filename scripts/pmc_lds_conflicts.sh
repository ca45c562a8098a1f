#!/bin/bash
# LDS bank-conflict share of the dense tile kernels: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel, at config 5 and at
# reduction 1 (the same counters profiles/r05_pmc_sq_stats_* hold for the tiles before the XOR swizzle of gemm_dense.hpp).
# usage (on the GPU box): bash scripts/pmc_lds_conflicts.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:-r06}; OUT=$R/gpurun_out; mkdir -p $OUT /tmp/w; cd /tmp; export TMPDIR=/tmp
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf /tmp/w/lc5; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/w/lc5 -o t -- python3 $R/bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 --no-cpu-baseline --no-breakdown --steady-steps 0 > /tmp/w/lc5.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $C -- python3 bench.py --features 200000 --reduction 12 --steps 40 --warmup 20 --no-cpu-baseline --no-breakdown --steady-steps 0"; python3 $R/scripts/pmc_summary.py $(find /tmp/w/lc5 -name '*.db' | head -1) | grep -E "^kernel|modl::"; } > $OUT/${TAG}_pmc_lds_conflicts_c5.txt 2>&1
rm -rf /tmp/w/lc1; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/w/lc1 -o t -- python3 $R/bench.py --reduction 1 --steps 100 --warmup 40 --no-cpu-baseline --no-breakdown --steady-steps 0 > /tmp/w/lc1.log 2>&1
{ echo "rocprofv3 --kernel-trace --pmc $C -- python3 bench.py --reduction 1 --steps 100 --warmup 40 --no-cpu-baseline --no-breakdown --steady-steps 0"; python3 $R/scripts/pmc_summary.py $(find /tmp/w/lc1 -name '*.db' | head -1) | grep -E "^kernel|modl::"; } > $OUT/${TAG}_pmc_lds_conflicts_r1.txt 2>&1
