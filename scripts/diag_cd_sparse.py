"""Cost of one solver sweep against the number of active coordinates, dense vs sparse sweeps
(MODL_CD_SPARSE_PCT=0: always dense; =100: always sparse).  tol = 0 -> exactly max_iter sweeps.
usage: python scripts/diag_cd_sparse.py [f32|f64] [k]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modl_amd.dict_fact_fast import _enet_regression_single_gram
dev = torch.device('cuda')
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == 'f64') else torch.float32
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device=dev).manual_seed(0)
b, p = 100, 1000
D = torch.randn(k, p, device=dev, generator=g, dtype=dt)
X = torch.randn(b, p, device=dev, generator=g, dtype=dt)
G = D @ D.T
Dx = X @ D.T
idx = np.arange(b)
out = []
for alpha in (1.0, 20.0, 40.0, 60.0, 80.0, 100.0, 120.0):
    ts = {}
    for mi in (10, 30):
        best = 1e9
        for rep in range(3):
            code = torch.ones(b, k, device=dev, dtype=dt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _enet_regression_single_gram(G, Dx, X, code, idx, 1.0, alpha, False, 0.0, mi)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        ts[mi] = best
    nnz = (code != 0).sum(1).float()
    out.append('alpha %5g: %.2f us per late sweep  nnz mean %.1f max %d' % (alpha, (ts[30] - ts[10]) / 20, nnz.mean().item(), int(nnz.max().item())))
print(os.environ.get('MODL_CD_SPARSE_PCT', 'default'), str(dt), 'k', k)
print('\n'.join(out))
