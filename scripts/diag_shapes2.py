"""Minibatch time over shapes around the metric's (number of atoms, batch size, aggregation mode): a cliff finder."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 120)
def run(k, b, r, **kw):
    est = DictFact(n_components=k, batch_size=b, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0, **kw)
    n = (X.shape[0] // b) * b
    est.prepare(n_samples=n, X=X[:max(k, 256)])
    nb = n // b
    est.partial_fit(X[:b * (nb // 3)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[b * (nb // 3):b * nb])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (nb - nb // 3)
    print('k=%4d b=%4d r=%3g %-40s %8.4f ms per minibatch  %9.0f samples/s' % (k, b, r, kw, dt * 1e3, b / dt))
for k in (30, 50, 70, 100, 150, 250, 254, 256):
    run(k, 256, 10)
for b in (10, 64, 100, 200, 512, 1024):
    run(256, b, 10)
run(256, 256, 10, G_agg='full', Dx_agg='full')
run(256, 256, 10, G_agg='average', Dx_agg='average')
run(200, 256, 10, G_agg='average', Dx_agg='average')
run(256, 256, 10, code_l1_ratio=0.0)
run(200, 256, 10, code_l1_ratio=0.0)
run(256, 256, 10, comp_l1_ratio=1.0)
run(256, 256, 10, optimizer='sgd')
