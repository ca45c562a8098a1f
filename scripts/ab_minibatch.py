"""Minibatch time of the library currently at modl_amd/libmodl_hip.so, four blocks of 250 fresh minibatches (A/B two builds on ONE box: copy one
library over the other between two runs of this script inside the same gpurun call - boxes differ by ~1 %)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import modl_amd._lib as L
import numpy as np, torch
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
RED = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0          # python scripts/ab_minibatch.py [reduction]
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 1400)
est = DictFact(n_components=256, batch_size=256, reduction=RED, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=65536, X=X[:256])
est.partial_fit(X[:256 * 400], np.arange(256 * 400) % 65536)
ts = []
for rep in range(4):
    a = 256 * (400 + 250 * rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[a:a + 256 * 250], np.arange(a, a + 256 * 250) % 65536)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 250 * 1e3)
print(os.path.basename(L.LIB_PATH), ' '.join('%.4f' % t for t in ts), 'ms per minibatch')
