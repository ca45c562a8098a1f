"""f32 parity diagnostics: GPU f32 vs oracle f32 vs oracle f64 on one minibatch from identical state."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from modl_amd import DictFact
from oracle import somf_oracle as orc

def rf(a, b):
    return np.linalg.norm(np.float64(a) - np.float64(b)) / np.linalg.norm(np.float64(b))

for r in (1, 10):
    rs = np.random.RandomState(0)
    n, p, k, b, k0 = 600, 2000, 256, 256, 32
    X64 = ((rs.randn(n, k0) * (rs.rand(n, k0) < 0.3)).dot(rs.randn(k0, p)) / np.sqrt(0.3 * k0) + 0.1 * rs.randn(n, p))
    X32 = X64.astype(np.float32)
    kw = dict(n_components=k, batch_size=b, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est = DictFact(**kw); est.prepare(n_samples=n, X=X32)
    D0 = est.components_
    pr32 = orc.SomfParams(**kw); s32 = orc.prepare(pr32, n_samples=n, X=X32); s32.sweeps = []
    pr64 = orc.SomfParams(**kw); s64 = orc.prepare(pr64, n_samples=n, X=X32.astype(np.float64)); s64.sweeps = []
    print('r=%d  D0 gpu vs orc32 %.2e' % (r, rf(D0, s32.D)))
    for step in range(2):
        rows = slice(step * b, (step + 1) * b)
        est.partial_fit(X32[rows], np.arange(rows.start, rows.stop))
        sg = est._backend.last_sweeps()
        orc.partial_fit(s32, pr32, X32[rows], np.arange(rows.start, rows.stop))
        orc.partial_fit(s64, pr64, X32[rows].astype(np.float64), np.arange(rows.start, rows.stop))
        cg, c32, c64 = est.code_[rows], s32.code[rows], s64.code[rows]
        per = np.linalg.norm(cg - c32, axis=1) / np.linalg.norm(c32, axis=1)
        print(' step %d sweeps mismatch gpu/orc32 %d, orc32/orc64 %d, gpu/orc64 %d ; mean sweeps %.1f' % (
            step, (sg != s32.sweeps[-1]).sum(), (s32.sweeps[-1] != s64.sweeps[-1]).sum(), (sg != s64.sweeps[-1]).sum(), sg.mean()))
        print('   codes: gpu-orc32 %.2e  gpu-orc64 %.2e  orc32-orc64 %.2e ; per-sample median %.1e max %.1e n>1e-4 %d' % (
            rf(cg, c32), rf(cg, c64), rf(c32, c64), np.median(per), per.max(), (per > 1e-4).sum()))
        print('   D:     gpu-orc32 %.2e  gpu-orc64 %.2e  orc32-orc64 %.2e' % (rf(est.components_, s32.D), rf(est.components_, s64.D), rf(s32.D, s64.D)))
        print('   B:     gpu-orc32 %.2e  gpu-orc64 %.2e  orc32-orc64 %.2e' % (rf(est.B_, s32.B), rf(est.B_, s64.B), rf(s32.B, s64.B)))
        print('   C:     gpu-orc32 %.2e  gpu-orc64 %.2e  orc32-orc64 %.2e' % (rf(est.C_, s32.C), rf(est.C_, s64.C), rf(s32.C, s64.C)))
