#!/bin/bash
O=gpurun_out/r05_38; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 0 > $O/drv_$i.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/drv_$i.json').read().strip().splitlines()[-1]);print('driver args', round(d['value']), d['ms_per_step'])"; done
python bench.py --no-cpu-baseline > $O/def.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/def.json').read().strip().splitlines()[-1]);print('default', round(d['value']), d['ms_per_step'], [(r['reduction'], round(r['value'])) for r in d['steady_state']])"
timeout 900 python -m pytest tests -m gpu -q -x -k "resident or status or gives_up or chunk_call or two_phase" 2>&1 | tail -3
