"""Long run at the headline shape: 2 epochs over a 64k-row resident chunk (512 minibatches), finiteness,
atoms inside the ball, objective on held-out rows before / after."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
n = 65536
X = bench.M1Stream(bench.P_FEAT, 99, dev).rows(0, n + 1024)
Xtest = X[n:]
est = DictFact(n_components=256, batch_size=256, reduction=10, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=n, X=X[:256])
s0 = est.score(Xtest)
t0 = time.perf_counter()
for ep in range(2):
    est.partial_fit(X[:n], np.arange(n))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
D = est.components_
s1 = est.score(Xtest)
print('512 minibatches in %.2f s (%.0f samples/s incl. second-epoch revisits)' % (dt, 2 * n / dt))
print('finite %s, max row norm %.5f, objective on held-out rows %.4f -> %.4f' % (bool(np.isfinite(D).all()), float(np.sqrt((D.astype(np.float64) ** 2).sum(1)).max()), s0, s1))
