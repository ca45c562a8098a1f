"""A/B of the register-resident statistics product (MODL_DEBUG_STATS_RESIDENT 1 against 0): same minibatches, the
statistics and the dictionary afterwards compared, then the step time of both at reduction 1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from modl_amd import DictFact
from modl_amd._lib import lib
SW = 10

def run(p, k, b, r, nb, sw, seed=0):
    lib.modl_debug_set(SW, sw)
    rng = np.random.RandomState(seed)
    X = rng.randn(nb * b, p).astype(np.float32)
    est = DictFact(n_components=k, batch_size=b, reduction=r, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
                   learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    est.prepare(n_samples=X.shape[0], X=X[:k])
    est.partial_fit(X)
    out = (np.array(est.B_), np.array(est.C_), np.array(est.components_))
    lib.modl_debug_set(SW, 1)
    return out

for (p, k, b, r, nb) in ((10000, 256, 256, 1, 3), (4104, 100, 200, 1, 3), (8192, 256, 64, 1, 5), (4100, 36, 256, 1, 2), (10000, 256, 200, 1, 2),
                         (6000, 256, 256, 1, 2)):
    a = run(p, k, b, r, nb, 0)
    c = run(p, k, b, r, nb, 1)
    print('p=%d k=%d b=%d r=%g: rel diff B %.2e  C %.2e  D %.2e' % ((p, k, b, r) + tuple(
        float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)) for x, y in zip(a, c))), flush=True)

import argparse, bench
args = argparse.Namespace(torch_collective=False, backend='nccl', force_reduce=False)
dev = torch.device('cuda', 0)
for sw in (0, 1, 0, 1):
    lib.modl_debug_set(SW, sw)
    run_ = bench.Run(args, 1.0, 0, 1, dev, 700)
    run_.fit(100); run_.sync()
    t0 = time.perf_counter(); run_.fit(500); run_.sync(); t1 = time.perf_counter()
    print('switch %d: reduction 1 step %.1f us' % (sw, (t1 - t0) / 500 * 1e6), flush=True)
lib.modl_debug_set(SW, 1)
