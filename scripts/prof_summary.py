"""Summarise a rocprofv3 (--kernel-trace) rocpd sqlite database: per-kernel calls, total/avg duration.
usage: python scripts/prof_summary.py <results.db> [min_start_fraction]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size, start from kernels order by start").fetchall()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1 = rows[0][8], rows[-1][8]
rows = [r for r in rows if r[8] >= t0 + frac * (t1 - t0)]
agg = {}
for name, dur, gx, gy, gz, wx, vg, lds, st in rows:
    short = re.sub(r'^void ', '', name)
    short = re.sub(r'\(.*$', '', short)
    a = agg.setdefault(short, [0, 0, None, vg, lds, []])
    a[0] += 1
    a[1] += dur
    a[2] = (gx // max(wx, 1), gy, gz, wx)
    a[5].append(dur)
tot = sum(a[1] for a in agg.values())
print('%-96s %7s %10s %9s %9s %9s %6s  %s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'median_us', 'max_us', '%', 'grid(wg) x block, vgpr, lds'))
for k_, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    d = sorted(a[5])
    print('%-96s %7d %10.3f %9.2f %9.2f %9.1f %6.1f  %s v%s lds%s' % (k_[:96], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, d[len(d) // 2] / 1e3, d[-1] / 1e3,
                                                                 100.0 * a[1] / tot, a[2], a[3], a[4]))
print('total kernel time %.3f ms over %d dispatches' % (tot / 1e6, len(rows)))
