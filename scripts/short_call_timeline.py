"""The timed region of `bench.py --steps K --warmup W` in a rocprofv3 --kernel-trace database: per minibatch the span and the
time in kernels, the idle gaps above 3 us, and the span of the whole region (first dispatch of minibatch W to the last
dispatch's end of minibatch W + K - 1).  usage: python scripts/short_call_timeline.py <results.db> [K] [W]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'^void ', '', n))[:60]
starts = [i for i, r in enumerate(rows) if 'prep_kernel' in r[0]]
if len(starts) < W + K + 1:
    raise SystemExit('only %d minibatches in the trace' % len(starts))
first, last = starts[W], starts[W + K]
reg = rows[first:last]
print('timed region: %d dispatches, span %.1f us, in kernels %.1f us' % (len(reg), (reg[-1][2] - reg[0][1]) / 1e3, sum(r[2] - r[1] for r in reg) / 1e3))
print('idle before its first dispatch (since the warm-up ended): %.1f us' % ((reg[0][1] - rows[first - 1][2]) / 1e3))
for s in range(K):
    a, b = starts[W + s], starts[W + s + 1]
    seg = rows[a:b]
    span = (rows[b][1] - seg[0][1]) / 1e3 if s + 1 < K else (seg[-1][2] - seg[0][1]) / 1e3
    gaps = [(seg[i][1] - seg[i - 1][2]) / 1e3 for i in range(1, len(seg))]
    big = ['%s +%.1f' % (short(seg[i][0])[:28], (seg[i][1] - seg[i - 1][2]) / 1e3) for i in range(1, len(seg)) if (seg[i][1] - seg[i - 1][2]) > 3000]
    print('minibatch %2d: span %7.1f us, kernels %7.1f us, %d dispatches; %s' % (s, span, sum(r[2] - r[1] for r in seg) / 1e3, len(seg), '; '.join(big)))
    if s == 0 or s == K - 1:
        for r in seg:
            print('      %-60s %8.1f %7.1f' % (short(r[0]), (r[1] - seg[0][1]) / 1e3, (r[2] - r[1]) / 1e3))
