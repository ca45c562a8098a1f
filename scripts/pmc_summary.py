"""Per-kernel averages of rocprofv3 --pmc counters from rocpd databases.
usage: python scripts/pmc_summary.py <results.db> [<results.db> ...] [--json out.json]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch; on gfx950 FETCH_SIZE counts a wide
coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section), so `fetch_bytes_corrected` doubles it."""
import json
import re
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
out_json = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
if out_json:
    args = [a for a in args if a != out_json]
agg = {}
for path in args:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, counter_name, counter_value, start from pmc_events order by start").fetchall()
    if not rows:
        continue
    t0, t1 = rows[0][3], rows[-1][3]
    for name, cname, val, st in rows:
        if st < t0 + 0.3 * (t1 - t0):
            continue                                   # skip allocation / warm-up
        short = re.sub(r'\(.*$', '', re.sub(r'^void ', '', name))[:90]
        a = agg.setdefault(short, {}).setdefault(cname, [0, 0.0])
        a[0] += 1
        a[1] += val
res = {}
print('%-92s %-12s %14s %8s' % ('kernel', 'counter', 'avg/dispatch', 'n'))
for kname, cs in sorted(agg.items()):
    ent = {}
    for cname, (n, tot) in sorted(cs.items()):
        print('%-92s %-12s %14.1f %8d' % (kname, cname, tot / n, n))
        ent[cname] = tot / n
        ent['dispatches'] = n
    if 'FETCH_SIZE' in ent:
        ent['fetch_bytes_corrected'] = 2.0 * 1024.0 * ent['FETCH_SIZE']
    if 'WRITE_SIZE' in ent:
        ent['write_bytes'] = 1024.0 * ent['WRITE_SIZE']
    res[kname] = ent
if out_json:
    json.dump(res, open(out_json, 'w'), indent=1, sort_keys=True)
