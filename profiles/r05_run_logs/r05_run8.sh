#!/bin/bash
O=gpurun_out/r05_22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_step.py -q -x -k "dictionary_update or headline or config1 or gram_accumulator" > $O/pytest_a.log 2>&1; tail -2 $O/pytest_a.log
timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python scripts/diag_persist_stamps.py 10 > $O/stamps.txt 2>&1; tail -22 $O/stamps.txt
