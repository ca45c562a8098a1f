"""diagnostics: the one-launch masked minibatch against the separate launches, minibatch by minibatch, on MovieLens-shaped rows
(how many sweep workgroups each minibatch takes; where the two paths part)"""
import sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from modl_amd.recsys import RecsysDictFact
from modl_amd._lib import lib, check, DEBUG_RECSYS_FUSED
rs = np.random.RandomState(0)
n_users, n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 120, 10677
pi = 1.0 / np.arange(1, n_items + 1) ** 0.9
pi /= pi.sum()
rows, cols = [], []
for u in range(n_users):
    deg = int(np.clip(rs.pareto(1.5) * 60 + 20, 20, 1500))
    cols.append(rs.choice(n_items, size=deg, replace=False, p=pi)); rows.append(np.full(deg, u))
rows, cols = np.concatenate(rows), np.concatenate(cols)
X = sp.csr_matrix((rs.randint(1, 11, size=len(rows)) / 2.0, (rows, cols)), shape=(n_users, n_items))
kw = dict(n_components=50, alpha=1, beta=.1, batch_size=10, learning_rate=.95, n_epochs=1, random_state=0)
res = {}
for fused in (0, 1):
    check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, fused))
    trace = []
    def cb(est):
        pass
    est = RecsysDictFact(callback=cb, **kw)
    # the Python loop (callback set): record the state after every minibatch
    orig = est._single_batch_fit
    def traced(Xm, batch, _o=orig, _t=trace, _e=est):
        _o(Xm, batch)
        d = _e._dev
        u = len(np.unique(np.concatenate([X.indices[X.indptr[i]:X.indptr[i + 1]] for i in batch])))
        _t.append((u, d.Dt.double().norm().item(), d.Bt.double().norm().item(), d.C.double().norm().item(), d.code.double().norm().item(), d.comp_norm.double().abs().sum().item(), d.comp_norm.cpu().numpy().copy(), d.Dt.cpu().numpy().copy()))
    est._single_batch_fit = traced
    t0 = time.time()
    est.fit(X)
    res[fused] = trace
    print('fused', fused, 'counts', est._dev.launch_counts(), '%.2f s' % (time.time() - t0))
check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, 1))
for t, (a, b) in enumerate(zip(res[0], res[1])):
    rel = [abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a[1:6], b[1:6])]
    print('mb %3d  u=%5d W=%d  rel diff of norms D %.1e B %.1e C %.1e code %.1e | cn split %.3e fused %.3e nan %d | max|dD| %.2e' % (t, a[0], -(-a[0] // 512), *rel[:4], a[5], b[5], int(np.isnan(b[6]).sum()), np.abs(a[7] - b[7]).max()))
    if t == 0:
        print('cn split', a[6][:8]); print('cn fused', b[6][:8])
