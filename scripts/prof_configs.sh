#!/bin/bash
# Kernel traces + host profiles of BASELINE configs 2-4 (scripts/bench_configs.py) on the GPU box:
#   bash scripts/prof_configs.sh r01_n     -> gpurun_out/<tag>_cfg_*.txt
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/w
cd /tmp && export TMPDIR=/tmp
for c in c2 c3 c4; do
  case $c in
    c2) A="--c2-patches 50000";;
    c3) A="--c3-records 10";;
    c4) A="--c4-batches 500 --c4-nnz 3000000";;
  esac
  rm -rf /tmp/w/kt_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt_$c -o t -- python3 $R/scripts/bench_configs.py --only $c $A > /tmp/w/kt_$c.log 2>&1
  { tail -1 /tmp/w/kt_$c.log; python3 $R/scripts/prof_summary.py $(find /tmp/w/kt_$c -name "*.db" | head -1) 0.1; } > $OUT/${TAG}_cfg_${c}_kernel_trace.txt 2>&1
  timeout 600 python3 -c "
import cProfile, pstats, sys
sys.argv = ['bench_configs.py', '--only', '$c'] + '$A'.split()
sys.path.insert(0, '$R/scripts')
import bench_configs
cProfile.run('bench_configs.main()', '/tmp/w/cp_$c.prof')
pstats.Stats('/tmp/w/cp_$c.prof').sort_stats('cumulative').print_stats(45)
" > $OUT/${TAG}_cfg_${c}_host_cprofile.txt 2>&1
done
ls -la $OUT | grep ${TAG}_cfg
