#!/bin/bash
# round 4, GPU run 3: the four-wavefront solver with look-ahead + DPP near path
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_4
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_kernels.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q -s --deselect tests/test_gpu_kernels.py > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python3 scripts/diag_f32_noise.py 10 8 1 > $OUT/f32_noise_split1.txt 2>&1
python3 scripts/diag_f32_noise.py 10 8 0 > $OUT/f32_noise_split0.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 600 --steady-burn-in 400 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
ls -la $OUT
