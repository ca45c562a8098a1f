"""C6 (the reference's HCP configuration, scripts/bench_configs.py: c6) beyond its first minibatches: ms per minibatch in groups of
four, up to N minibatches - how much of the young dictionary's projection work (ten Michelot passes per atom) is a transient.
usage (GPU box): python scripts/diag_c6_steady.py [N=48]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench_configs as bc  # noqa: E402
from modl_amd import DictFact  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device('cuda')
kw = dict(bc.HCP_KW)
b, k, p = kw['batch_size'], kw['n_components'], 200000
X = bc.hcp_rows(N * b, p, 0, dev)
est = DictFact(**kw)
est.prepare(n_samples=X.shape[0], X=X[:k])
G = 4
for g0 in range(0, N, G):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    est.partial_fit(X[g0 * b:(g0 + G) * b], np.arange(g0 * b, (g0 + G) * b))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    D = est._backend.Dt
    nz = float((D != 0).float().mean().item())
    print('minibatches %3d-%3d: %.2f ms per minibatch; share of non-zero dictionary entries %.3f' % (g0, g0 + G - 1, dt / G * 1e3, nz), flush=True)
