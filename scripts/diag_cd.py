"""Diagnostics of the code solver on the bench stream: sparsity of the codes, sweeps."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
X = bench.M1Stream(bench.P_FEAT, 1234, dev).rows(0, 8192)
for red in (10.0, 1.0):
    est = DictFact(n_components=256, batch_size=256, reduction=red, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
                   learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    est.prepare(n_samples=8192, X=X[:256])
    for s in range(24):
        est.partial_fit(X[s * 256:(s + 1) * 256], np.arange(s * 256, (s + 1) * 256))
        if s in (0, 1, 5, 23):
            code = est.code_[s * 256:(s + 1) * 256]
            nnz = (code != 0).sum(1)
            sw = est._backend.last_sweeps()
            print('r=%g step %d: nnz mean %.1f max %d min %d | sweeps mean %.2f max %d' % (red, s, nnz.mean(), nnz.max(), nnz.min(), sw.mean(), sw.max()))
