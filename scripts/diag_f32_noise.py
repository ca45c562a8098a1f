"""Single-step f32 noise of the GPU step against the oracle's own f32 noise, at the metric's shape.

The f64 oracle run is the truth.  Before every minibatch the f32 GPU estimator AND the f32 oracle are RE-SYNCHRONISED
to the f64 state (dictionary, statistics, code rows, norm budgets), then each fits the minibatch: what is printed is the
rounding noise ONE step adds (rel. Frobenius against the f64 step from the same state), per quantity - where the GPU's
f32 arithmetic is noisier than the reference algorithm's (numpy/BLAS f32), it shows here without 48 steps of
accumulation on top.   python scripts/diag_f32_noise.py [reduction] [steps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modl_amd import DictFact  # noqa: E402
from oracle import somf_oracle as orc  # noqa: E402
from tests.conftest import m1_rows, HEADLINE_KW, rel_fro  # noqa: E402

r = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
split = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # 0: the one-wavefront solver (the reference's operation order)
from modl_amd._lib import lib, check, DEBUG_CD_SPLIT  # noqa: E402
check(lib.modl_debug_set(DEBUG_CD_SPLIT, split))
p, b = 10000, 256
n = steps * b
X32 = m1_rows(n, p, seed=77)
X64 = X32.astype(np.float64)
kw = dict(HEADLINE_KW, reduction=r)
pr = orc.SomfParams(**kw)
st64 = orc.prepare(pr, n_samples=n, X=X64)
st32 = orc.prepare(pr, n_samples=n, X=X32)
est = DictFact(**kw)
est.prepare(n_samples=n, X=X32)
st64.sweeps, st32.sweeps = [], []
print('cd_split %d' % split)
print('reduction %g: per minibatch, noise of ONE f32 step from the f64 state: GPU | oracle f32   (rel. Frobenius vs the f64 step)' % r)
for t in range(steps):
    rows = slice(t * b, (t + 1) * b)
    idx = np.arange(rows.start, rows.stop)
    # re-synchronise both f32 runs to the truth
    for name in ('D', 'C', 'B', 'comp_norm'):
        getattr(st32, name)[...] = getattr(st64, name).astype(np.float32)
    st32.code[rows] = st64.code[rows].astype(np.float32)
    est.components_ = st64.D.astype(np.float32)
    est.C_ = st64.C.astype(np.float32)
    est.B_ = st64.B.astype(np.float32)
    est.comp_norm_ = st64.comp_norm.astype(np.float32)
    est._python_loop = True
    orc.partial_fit(st64, pr, X64[rows], idx)
    orc.partial_fit(st32, pr, X32[rows], idx)
    est.partial_fit(X32[rows], idx)
    sw = est._backend.last_sweeps()
    out = []
    for name, g, o, ref in (('code', est.code_[rows], st32.code[rows], st64.code[rows]), ('C', est.C_, st32.C, st64.C),
                            ('B', est.B_, st32.B, st64.B), ('D', est.components_, st32.D, st64.D),
                            ('norm', est.comp_norm_, st32.comp_norm, st64.comp_norm)):
        out.append('%s %.2e | %.2e' % (name, rel_fro(g, ref), rel_fro(o, ref)))
    print('t=%2d  %s   sweeps agree: gpu/f64 %.3f  oracle32/f64 %.3f' % (
        t, '   '.join(out), float(np.mean(sw == st64.sweeps[-1])), float(np.mean(st32.sweeps[-1] == st64.sweeps[-1]))))
