"""Minibatch time at shapes other than the metric's (k not a power of two, wide p): python scripts/diag_shapes.py [lib.so|new] [shape index]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import modl_amd._lib as L
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):     # another build of the library to compare with
    L.LIB_PATH = sys.argv[1]
import numpy as np, torch
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
shapes = ((512, 10000, 10), (384, 10000, 10), (256, 40000, 10), (256, 100000, 12), (128, 10000, 10), (320, 10000, 10), (200, 10000, 10), (250, 10000, 10), (70, 10000, 10))
if len(sys.argv) > 2:
    shapes = (shapes[int(sys.argv[2])],)
for (k, p, r) in shapes:
    X = bench.M1Stream(p, 1234, dev).rows(0, 256 * 300)
    est = DictFact(n_components=k, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=X.shape[0], X=X[:1024])
    est.partial_fit(X[:256 * 100])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[256 * 100:256 * 300])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print('%s k=%d p=%d r=%g: %.4f ms per minibatch' % (sys.argv[1] if len(sys.argv) > 1 else 'new', k, p, r, dt * 1e3))
    del est, X
