import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from modl_amd import DictFact
from oracle import somf_oracle as orc
rs = np.random.RandomState(0)
n, p, k = 96, 200, 16
X64 = (rs.randn(n, 8) @ rs.randn(8, p))
kw = dict(n_components=k, batch_size=32, reduction=2, code_alpha=0.1, random_state=0, learning_rate=0.9)
res = {}
for dt in (np.float32, np.float64):
    X = X64.astype(dt)
    est = DictFact(**kw); est.prepare(n_samples=n, X=X); est.partial_fit(X)
    pr = orc.SomfParams(**kw); st = orc.prepare(pr, n_samples=n, X=X); orc.partial_fit(st, pr, X)
    res[dt] = (est.components_.astype(np.float64), est.code_.astype(np.float64), st.D.astype(np.float64), st.code.astype(np.float64))
    print(dt.__name__, 'gpu vs oracle: D %.2e code %.2e' % (np.linalg.norm(res[dt][0]-res[dt][2])/np.linalg.norm(res[dt][2]), np.linalg.norm(res[dt][1]-res[dt][3])/np.linalg.norm(res[dt][3])))
t = res[np.float64]
for name, (D, c, oD, oc) in (('f32', res[np.float32]),):
    print('vs f64 truth: gpu-f32 D %.2e code %.2e | oracle-f32 D %.2e code %.2e' % (np.linalg.norm(D-t[2])/np.linalg.norm(t[2]), np.linalg.norm(c-t[3])/np.linalg.norm(t[3]), np.linalg.norm(oD-t[2])/np.linalg.norm(t[2]), np.linalg.norm(oc-t[3])/np.linalg.norm(t[3])))
print('--- well-conditioned variant (full-rank data)')
for seed in (0, 1, 2):
    rs = np.random.RandomState(seed)
    X = (rs.randn(n, 32) @ rs.randn(32, p) + 0.5 * rs.randn(n, p)).astype(np.float32)
    est = DictFact(**kw); est.prepare(n_samples=n, X=X); est.partial_fit(X)
    pr = orc.SomfParams(**kw); st = orc.prepare(pr, n_samples=n, X=X); orc.partial_fit(st, pr, X)
    X6 = X.astype(np.float64)
    st6 = orc.prepare(pr, n_samples=n, X=X6); orc.partial_fit(st6, pr, X6)
    f = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
    print('seed %d: gpu-f32 vs oracle-f32 D %.2e code %.2e | vs f64 truth: gpu D %.2e code %.2e, oracle-f32 D %.2e code %.2e' % (
        seed, f(est.components_, st.D.astype(np.float64)), f(est.code_, st.code.astype(np.float64)),
        f(est.components_, st6.D), f(est.code_, st6.code), f(st.D, st6.D), f(st.code, st6.code)))
