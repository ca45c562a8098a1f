#!/bin/bash
# final: the whole GPU suite, then the round's profiles with the final kernels
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests_final.txt 2>&1
tail -5 gpurun_out/r04_gpu_tests_final.txt
bash scripts/make_profiles.sh r04 > gpurun_out/r04_make_profiles.log 2>&1
bash scripts/prof_configs.sh r04 > gpurun_out/r04_prof_configs.log 2>&1
MODL_AMD_DIAG=1 timeout 300 python scripts/diag_cd_split_stamps.py > gpurun_out/r04_cd_split_stamps.txt 2>&1
timeout 300 python scripts/diag_stamps.py > gpurun_out/r04_bcd_block_stamps.txt 2>&1
timeout 600 python scripts/diag_f32_noise.py 10 8 1 > gpurun_out/r04_f32_noise_one_step.txt 2>&1
ls gpurun_out | grep r04_ | wc -l
