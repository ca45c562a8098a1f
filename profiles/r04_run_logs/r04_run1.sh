#!/bin/bash
# round 4, GPU run 1: the whole GPU suite, the bench lines, the two-phase (multi-GPU) step's single-GPU cost
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_1
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
# the fused step against the two-phase step (world = 1, forced all-reduce) on both routes: 600 steps after 400, no events
for v in fused native torch; do
  case $v in
    fused) F="" ;;
    native) F="--force-reduce --native-rccl" ;;
    torch) F="--force-reduce --torch-collective" ;;
  esac
  MASTER_PORT=2951$RANDOM python3 $R/bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown $F > $OUT/two_phase_$v.json 2> $OUT/two_phase_$v.err
done
ls -la $OUT
