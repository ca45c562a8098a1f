#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/w; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w/c6; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/c6 -o t -- python3 $R/scripts/bench_configs.py --only c6 --c6-batches 5 > /tmp/w/c6.log 2>&1
python3 $R/scripts/prof_summary.py $(find /tmp/w/c6 -name "*.db" | head -1) 0.3 2>&1 | head -8
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/w/c6/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.grid_size_x, (d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%atom_step_group%' order by d.start").fetchall()
import collections
agg = collections.defaultdict(list)
for n, g, dt in rows[len(rows)//3:]:
    agg[g].append(dt)
for g, v in sorted(agg.items()):
    print('grid', g, 'launches', len(v), 'avg us', sum(v)/len(v)/1e3)
PY
