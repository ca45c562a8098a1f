#!/bin/bash
O=gpurun_out/r05_20; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for v in 1 0; do
  timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline --debug-set 9=$v > $O/bench_persist$v.json 2> $O/bench_persist$v.err
done
timeout 300 python scripts/diag_persist_stamps.py 10 > $O/stamps.txt 2>&1
