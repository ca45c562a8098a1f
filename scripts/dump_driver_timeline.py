"""the LAST 20 minibatches of a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5 --steady-steps 0 --no-breakdown` (the
driver's arguments): per minibatch the span, the time inside kernels and the persistent / solver kernel durations"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
preps = [i for i, r in enumerate(rows) if 'prep_kernel' in r[0]]
preps = preps[-20:]
ends = preps[1:] + [max(i for i, r in enumerate(rows) if 'bcd_persist_kernel' in r[0] or 'bcd_block_kernel' in r[0]) + 1]
tot_span = tot_busy = 0.0
for a, b in zip(preps, ends):
    seg = rows[a:b]
    span = (seg[-1][2] - seg[0][1]) / 1e3
    nxt = rows[b][1] if b < len(rows) and 'prep_kernel' in rows[b][0] else seg[-1][2]
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    per = {re.sub(r'<.*$', '', r[0]).replace('modl::', '')[:22]: round((r[2] - r[1]) / 1e3, 1) for r in seg}
    print('span to next step %7.1f us  in kernels %7.1f  %s' % ((nxt - seg[0][1]) / 1e3, busy, per))
    tot_span += (nxt - seg[0][1]) / 1e3; tot_busy += busy
print('20 steps: span %.1f us, in kernels %.1f us' % (tot_span, tot_busy))
