"""The driver's bench arguments (--steps 20 --warmup 5) against the steady state: wall time per minibatch of a
20-minibatch partial_fit call issued 5 / 50 / 200 minibatches after initialisation (synchronised before and after), with
the mean sweep count of its last minibatch, and the fixed cost of a call (1, 5, 20 minibatches per call)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
b = 256
X = bench.M1Stream(10000, 1234, dev).rows(0, 65536)
for warm in (5, 5, 50, 200):
    est = DictFact(n_components=256, batch_size=b, reduction=10, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    est.partial_fit(X[:warm * b], np.arange(warm * b), _sync=False)
    torch.cuda.synchronize()
    row = warm * b
    t0 = time.perf_counter()
    est.partial_fit(X[row:row + 20 * b], np.arange(row, row + 20 * b), _sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('20 minibatches, %3d after init: %.4f ms per minibatch, host returns after %.3f ms, %.2f sweeps' %
          (warm, (t2 - t0) * 1e3 / 20, (t1 - t0) * 1e3, float(est._backend.last_sweeps().mean())))
row = 0
for n in (1, 5, 20):
    ts = []
    for rep in range(5):
        idx = np.arange(row, row + n * b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est.partial_fit(X[row:row + n * b], idx, _sync=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3))
        row += n * b
    ts.sort()
    print('%4d minibatches per call: %.3f ms wall (%.3f ms per minibatch), host returns after %.3f ms' % (n, ts[2][0], ts[2][0] / n, ts[2][1]))
# section times (HIP events around every section: each pair costs a bubble, the sums are still comparable) early and late
for warm in (5, 200):
    est = DictFact(n_components=256, batch_size=b, reduction=10, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    est.partial_fit(X[:warm * b], np.arange(warm * b), _sync=False)
    torch.cuda.synchronize()
    be = est._backend
    be.prof_enable(True); be.prof_reset()
    row = warm * b
    est.partial_fit(X[row:row + 20 * b], np.arange(row, row + 20 * b), _sync=False)
    torch.cuda.synchronize()
    got = be.prof_get()
    be.prof_enable(False)
    print('%3d after init:' % warm, ', '.join('%s %.1f us' % (n, e['ms'] / e['calls'] * 1e3) for n, e in got.items() if e['calls']),
          '| sweeps max %d' % int(be.last_sweeps().max()))
