"""Time of the code-solver kernel as a function of the number of sweeps (tol = 0 -> exactly max_iter
sweeps): per-sweep cost of a dense sweep.  Torch tensors in, timed with CUDA events on the current stream."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modl_amd.dict_fact_fast import _enet_regression_single_gram
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
b, k, p = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 1000
D = torch.randn(k, p, device=dev, generator=g)
X = torch.randn(b, p, device=dev, generator=g)
G = D @ D.T
Dx = X @ D.T
idx = np.arange(b)
for alpha in (1.0, 200.0):
    for mi in (1, 2, 4, 8):
        ts = []
        for rep in range(5):
            code = torch.ones(b, k, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _enet_regression_single_gram(G, Dx, X, code, idx, 1.0, alpha, False, 0.0, mi)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        nnz = (code != 0).sum(1).float().mean().item()
        print('alpha %g max_iter %d: %.1f us (min of 5)  nnz %.1f' % (alpha, mi, min(ts), nnz))
