#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 900 python scripts/bench_configs.py --only c6 2>&1 | tail -1 | cut -c1-420
timeout 2000 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "generic or golden or trajector or fmri or nmf" 2>&1 | tail -12
