"""Timeline of single minibatch steps from a rocprofv3 --kernel-trace rocpd database: every dispatch of a few
steps in the steady state with its start offset, duration and the gap to the previous dispatch's end.
usage: python scripts/step_timeline.py <results.db> [n_steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
short = lambda n: re.sub(r'\(.*$', '', re.sub(r'^void ', '', n))[:72]
# a step starts at its parameter staging launch - or, inside a chunk of minibatches (round 3: the copy rides on the previous
# step's last launch), at prep_kernel
def step_start(i):
    n = rows[i][0]
    if 'stage_params_kernel' in n:
        return True
    return 'prep_kernel' in n and not (i > 0 and 'stage_params_kernel' in rows[i - 1][0])
starts = [i for i in range(len(rows)) if step_start(i)]
if len(starts) < nsteps + 4:
    raise SystemExit('not enough steps in the trace')
starts_set = set(starts)
first = starts[int(len(starts) * 0.8)]
last = starts[int(len(starts) * 0.8) + nsteps]
t0 = rows[first][1]
prev_end = None
print('%-74s %9s %8s %8s  %s' % ('kernel', 'start_us', 'dur_us', 'gap_us', 'grid(wg) x block'))
for name, st, en, gx, wx in rows[first:last]:
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    if starts and rows.index((name, st, en, gx, wx)) in starts_set:
        print('---- step')
    print('%-74s %9.2f %8.2f %8.2f  %d x %d' % (short(name), (st - t0) / 1e3, (en - st) / 1e3, gap, gx // max(wx, 1), wx))
    prev_end = en
# aggregate over all steps of the last 40 %: time in kernels vs gaps
sel = rows[starts[int(len(starts) * 0.6)]:starts[-1]]
busy = sum(r[2] - r[1] for r in sel)
span = sel[-1][2] - sel[0][1]
n = len([i for i in starts if starts[int(len(starts) * 0.6)] <= i < starts[-1]])
print('over %d steps: %.1f us per step, %.1f us in kernels, %.1f us in gaps, %.1f dispatches per step'
      % (n, span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3, len(sel) / n))
