#!/bin/bash
O=gpurun_out/r05_27; mkdir -p $O
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 0 > $O/with_$i.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 0 --no-breakdown > $O/without_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --steady-steps 0 --no-breakdown > $O/without_200.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_27/*.json')):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(r['value']), round(r['ms_per_step'],4), round(r['host_enqueue_ms_per_step'],4), r['cd_sweeps_mean'], r['cd_sweeps_max'])
PY
