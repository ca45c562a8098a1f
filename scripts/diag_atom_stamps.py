import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')     # the stamps only exist in the diagnostics build (libmodl_hip_diag.so)
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
dev = torch.device('cuda')
p, n, b, k = 60000, 400, 50, 70
X = bench.M1Stream(p, 3, dev, k0=64).rows(0, n)
est = DictFact(n_components=k, batch_size=b, reduction=12, code_alpha=1e-3, code_l1_ratio=0, comp_l1_ratio=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=n, X=X[:k]); est.partial_fit(X[:200], np.arange(200))
out = (C.c_ulonglong * 48)()
check(lib.modl_somf_debug_stamps(est._backend.plan, out))
o = [int(v) for v in out[:8]]
print('   projection: start +%d, Michelot loop until +%d (%d passes, %d active), end +%d' % (o[7]-o[2], o[6]-o[2], o[4], o[5], o[3]-o[2]))
print('last workgroup of the last atom launch: grad+partials %d  arrive %d  projection+writeback %d cycles' % (o[1]-o[0], o[2]-o[1], o[3]-o[2]))
