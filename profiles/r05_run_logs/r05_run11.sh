#!/bin/bash
O=gpurun_out/r05_24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_step.py -q -x -k "gives_up or headline or dictionary_update" > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
