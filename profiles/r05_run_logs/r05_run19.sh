#!/bin/bash
O=gpurun_out/r05_37; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q  > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python - <<'XEOF'
import json
d=json.loads(open('gpurun_out/r05_37/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
for r in d.get('steady_state', []): print(r.get('reduction'), r.get('samples_per_s'), r.get('ms_per_step'), r.get('parity', {}).get('within_1e5'), r.get('sections_ms'))
XEOF
timeout 600 python bench.py --features 200000 --reduction 12 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5.json
