#!/bin/bash
O=gpurun_out/r05_19; mkdir -p $O
timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline --debug-set 9=2 > $O/bench_force.json 2> $O/bench_force.err
timeout 300 python bench.py --steps 20 --warmup 5 --steady-steps 0 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
timeout 300 python bench.py --features 200000 --reduction 12 --steps 60 --warmup 20 --no-cpu-baseline --debug-set 9=2 > $O/bench_c5_force.json 2> $O/bench_c5_force.err
