import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from modl_amd import DictFact
from oracle import somf_oracle as orc
def rel(a, b): return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)
for (p, k, b, red, n) in [(8, 4, 4, 50.0, 40), (12, 8, 5, 6.0, 33), (64, 8, 16, 64.0, 64), (5, 3, 7, 1.0, 20)]:
    rs = np.random.RandomState(1)
    X = rs.randn(n, p)
    kw = dict(n_components=k, batch_size=b, reduction=red, code_alpha=0.1, learning_rate=0.9, random_state=0)
    est = DictFact(**kw); est.prepare(n_samples=n, X=X); est.partial_fit(X)
    pr = orc.SomfParams(**kw); st = orc.prepare(pr, n_samples=n, X=X)
    try:
        orc.partial_fit(st, pr, X)
    except ValueError as e:      # the reference itself fails on an empty feature subset (scipy ger on a 0-column block)
        print((p, k, b, red, n), 'oracle/reference cannot run this case:', str(e)[:60], '| GPU finite:', bool(np.isfinite(est.components_).all()))
        continue
    print((p, k, b, red, n), 'D %.1e code %.1e C %.1e B %.1e' % (rel(est.components_, st.D), rel(est.code_, st.code), rel(est.C_, st.C), rel(est.B_, st.B)))
