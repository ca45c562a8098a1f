"""Timeline of a few masked minibatches from a rocprofv3 --kernel-trace database: dispatches with duration and the gap to the
previous dispatch's end.   usage: python scripts/c4_timeline.py <results.db>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'^void ', '', n))[:60]
i0 = next(i for i in range(len(rows) // 2, len(rows)) if 'recsys_fused' in rows[i][0])
prev = None
for name, st, en, gx, wx in rows[i0:i0 + 12]:
    gap = (st - prev) / 1e3 if prev else 0.0
    print('%-62s dur %7.2f us  gap %6.2f us  %d x %d' % (short(name), (en - st) / 1e3, gap, gx // max(wx, 1), wx))
    prev = en
sel = [r for r in rows[len(rows) // 2:] if any(t in r[0] for t in ('recsys_fused', 'update_B', 'bcd_prepare', 'bcd_few', 'gemm_kernel', 'bcd_gram', 'bcd_resolve', 'bcd_apply', 'recsys_stage'))]
busy = sum(r[2] - r[1] for r in sel); span = sel[-1][2] - sel[0][1]
n = sum(1 for r in sel if 'recsys_fused' in r[0])
print('over %d minibatches: %.1f us per minibatch, %.1f us in kernels, %.1f us in gaps' % (n, span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3))
