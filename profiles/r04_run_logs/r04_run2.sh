#!/bin/bash
# round 4, GPU run 2: the new four-wavefront solver (suite, stamps, timing), the two-phase step's single-GPU cost
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_2
mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in fused native torch; do
  case $v in
    fused) F="" ;;
    native) F="--force-reduce --native-rccl" ;;
    torch) F="--force-reduce --torch-collective" ;;
  esac
  MASTER_PORT=$((20000 + RANDOM % 20000)) python3 $R/bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown $F > $OUT/two_phase_$v.json 2> $OUT/two_phase_$v.err
done
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 600 --steady-burn-in 400 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
ls -la $OUT
