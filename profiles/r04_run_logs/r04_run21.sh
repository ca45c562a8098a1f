#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "generic_dictionary_update_f32_groups" 2>&1 | grep -v amdgpu.ids | tail -30
