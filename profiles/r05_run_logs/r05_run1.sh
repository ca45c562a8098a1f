#!/bin/bash
# round 5, call 1: full GPU suite on the round's first commit, the default bench line, C5 as a bench workload (fused / forced reduce)
O=gpurun_out/r05_1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --steady-steps 0 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
timeout 600 python bench.py --features 200000 --reduction 12 --steps 60 --warmup 20 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py --features 200000 --reduction 12 --steps 60 --warmup 20 --force-reduce --no-cpu-baseline > $O/bench_c5_reduce.json 2> $O/bench_c5_reduce.err
timeout 600 python bench.py --features 200000 --reduction 12 --steps 60 --warmup 20 --force-reduce --torch-collective --no-cpu-baseline > $O/bench_c5_reduce_torch.json 2> $O/bench_c5_reduce_torch.err
tail -3 $O/pytest.log
