#!/bin/bash
# round 4, GPU run 5: solver with pipelined update-wave fetch; the launch gap in front of it; C6; first r04 profiles
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_5
mkdir -p $OUT /tmp/w
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_kernels.log
python3 scripts/diag_cd_split_stamps.py > $OUT/cd_stamps.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q -s --deselect tests/test_gpu_kernels.py > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 --steady-steps 600 --steady-burn-in 400 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
rm -rf /tmp/w/gap; timeout 300 rocprofv3 --kernel-trace -d /tmp/w/gap -o t -- python3 $R/scripts/diag_cd_launch_gap.py > /tmp/w/gap.log 2>&1
python3 $R/scripts/dump_trace.py $(find /tmp/w/gap -name "*.db" | head -1) 24 > $OUT/cd_launch_gap.txt 2>&1
rm -rf /tmp/w/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/kt -o t -- python3 $R/bench.py --steps 600 --warmup 400 --no-cpu-baseline --steady-steps 0 --no-breakdown > /tmp/w/kt.log 2>&1
DB=$(find /tmp/w/kt -name "*.db" | head -1)
python3 $R/scripts/prof_summary.py $DB 0.5 > $OUT/kernel_trace_r10.txt 2>&1
python3 $R/scripts/step_timeline.py $DB 1 > $OUT/step_timeline_r10.txt 2>&1
timeout 900 python3 $R/scripts/bench_configs.py --only c2,c6 > $OUT/bench_configs_c2_c6.jsonl 2> $OUT/bench_configs_c2_c6.err
ls -la $OUT
