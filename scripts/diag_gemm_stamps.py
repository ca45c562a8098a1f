"""Diagnostic: where one tile of the head statistics product (K = b = 256) spends its time.
Run with MODL_GEMM_STAMPS=1."""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')     # the stamps only exist in the diagnostics build (libmodl_hip_diag.so)
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from modl_amd import DictFact
from modl_amd._lib import lib, check

rs = np.random.RandomState(0)
n, p, k, b = 4096, 10000, 256, 256
X = torch.from_numpy((rs.randn(n, 64) @ rs.randn(64, p) + rs.randn(n, p)).astype(np.float32)).cuda()
est = DictFact(n_components=k, batch_size=b, reduction=10, code_alpha=1.0, random_state=0, learning_rate=0.92)
est.prepare(n_samples=n, X=X[:k])
est.partial_fit(X, np.arange(n))
out = (C.c_ulonglong * 8)()
check(lib.modl_somf_debug_gemm_stamps(est._backend.plan, out))
s = list(out)
print('cycles: loads issued %d, first K-tile landed+in LDS %d, K loop %d, epilogue %d, total %d' % (
    s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0]))
wall = (s[7] - s[6]) / 100e6
print('wall %.2f us -> shader clock %.2f GHz' % (wall * 1e6, (s[4] - s[0]) / wall / 1e9))
