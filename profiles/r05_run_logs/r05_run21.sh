#!/bin/bash
O=gpurun_out/r05_45; mkdir -p $O /tmp/w; R=$PWD
python bench.py --features 200000 --reduction 12 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]);print('c5', round(d['value']), d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['sections'].items()})"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w/c5; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c5 -o t -- python3 $R/scripts/diag_hcp_shape.py > /tmp/w/c5.log 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/w/c5 -name "*.db" | head -1) 0.3 | grep -E "prep_kernel|gemm_dense_pair|^kernel" | cut -c1-160
cd $R; timeout 600 python -m pytest tests -m gpu -q -x -k "c5 or wide or chunk or headline or trajectory_small" 2>&1 | tail -2
