"""Every dispatch of the last `n` of a rocprofv3 --kernel-trace rocpd database with the gap to the previous one's end.
usage: python scripts/dump_trace.py <results.db> [n]"""
import re
import sqlite3
import sys

rows = sqlite3.connect(sys.argv[1]).cursor().execute(
    "select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
short = lambda s: re.sub(r'\(.*$', '', re.sub(r'^void ', '', s))[:70]
print('%-72s %9s %8s %8s  %s' % ('kernel', 'start_us', 'dur_us', 'gap_us', 'grid(wg) x block, lds, vgpr'))
t0, prev = rows[0][1], None
for name, st, en, gx, wx, lds, vg in rows:
    gap = (st - prev) / 1e3 if prev is not None else 0.0
    print('%-72s %9.2f %8.2f %8.2f  %d x %d lds%s v%s' % (short(name), (st - t0) / 1e3, (en - st) / 1e3, gap, gx // max(wx, 1), wx, lds, vg))
    prev = en
