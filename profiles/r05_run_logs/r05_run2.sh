#!/bin/bash
# round 5, call 2: the persistent dictionary update - parity tests under a timeout, A/B against one launch per block, stamps
O=gpurun_out/r05_2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -k "dictionary_update or headline or trajector or config1 or two_phase_equals or gram_accumulator" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
for v in 1 0; do
  timeout 300 python bench.py --steps 200 --warmup 50 --steady-steps 1000 --steady-burn-in 300 --no-cpu-baseline --debug-set 9=$v > $O/bench_persist$v.json 2> $O/bench_persist$v.err
done
timeout 300 python scripts/diag_persist_stamps.py > $O/stamps.txt 2>&1
tail -30 $O/stamps.txt
