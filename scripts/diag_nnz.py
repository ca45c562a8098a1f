import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 700)
for r in (10, 1):
    est = DictFact(n_components=256, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=X.shape[0], X=X[:256])
    est.partial_fit(X[:256 * 600])
    code = est._backend.get('code') if hasattr(est._backend, 'get') else None
    c = est.code_[256 * 599: 256 * 600] if code is None else None
    code = torch.as_tensor(est.code_)[256 * 599: 256 * 600]
    nnz = (code != 0).sum(1).float()
    print('r=%g: nnz per sample mean %.1f min %d max %d of 256; sweeps mean %.2f' % (r, nnz.mean().item(), nnz.min().item(), nnz.max().item(), est._backend.last_sweeps().mean()))
