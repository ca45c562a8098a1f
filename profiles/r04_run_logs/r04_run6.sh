#!/bin/bash
# round 4, GPU run 6: multi-row update waves; where the launch gap in front of the solver comes from
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_6
mkdir -p $OUT /tmp/w
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_kernels.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
python3 scripts/diag_cd_split_stamps.py 128 256 1000 > $OUT/cd_stamps_k128.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q -s --deselect tests/test_gpu_kernels.py > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python3 scripts/diag_f32_noise.py 10 6 1 > $OUT/f32_noise_split1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 600 --steady-burn-in 400 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
for v in split cdk; do
  case $v in
    split) F="" ;;
    cdk) F="--debug-set 2=0" ;;
  esac
  rm -rf /tmp/w/kt_$v; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt_$v -o t -- python3 $R/bench.py --steps 300 --warmup 200 --no-cpu-baseline --steady-steps 0 --no-breakdown $F > /tmp/w/kt_$v.log 2>&1
  DB=$(find /tmp/w/kt_$v -name "*.db" | head -1)
  python3 $R/scripts/step_timeline.py $DB 1 > $OUT/step_timeline_$v.txt 2>&1
done
ls -la $OUT
