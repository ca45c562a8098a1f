#!/bin/bash
# tuning sweep: rows per workgroup of the block kernel x rider distribution
for rt in 256 2000; do for nr in 1 0; do for rl in 0 4 2; do
  if [ $nr = 1 ] && [ $rl != 0 ]; then continue; fi
  export MODL_RT1_MAX=$rt; export MODL_RIDER_LAUNCHES=$rl
  if [ $nr = 1 ]; then export MODL_NO_RIDER=1; else unset MODL_NO_RIDER; fi
  echo -n "RT1_MAX=$rt NO_RIDER=$nr RIDER_LAUNCHES=$rl: "
  timeout 300 python bench.py --steps 2000 --warmup 400 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['sections'].items() if k in ('stats_gemm','dict_update')})"
done; done; done
