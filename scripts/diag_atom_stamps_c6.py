"""Stamps of the last-arriving workgroup of atom_step_kernel at config 6's shape (k = 1024, p = 200 000, reduction 20: 10 000
sampled features, positive l1 atoms - one launch per atom).  Diagnostics build."""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
for item in sys.argv[1:]:                                  # what=value: modl_debug_set before anything runs
    what, value = item.split('=')
    check(lib.modl_debug_set(int(what), int(value)))
dev = torch.device('cuda')
p, n, b, k = 200000, 1200, 200, 1024
X = bench.M1Stream(p, 3, dev, k0=64).rows(0, n)
est = DictFact(n_components=k, batch_size=b, reduction=20, code_alpha=1e-3, code_l1_ratio=0, comp_l1_ratio=1.0, comp_pos=True,
               learning_rate=0.92, random_state=0)
est.prepare(n_samples=n, X=X[:k])
est.partial_fit(X[:400], np.arange(400))
out = (C.c_ulonglong * 48)()
check(lib.modl_somf_debug_stamps(est._backend.plan, out))
o = [int(v) for v in out[:8]]
print('last workgroup of the last atom launch: gradient rows + partial norms %d  arrival %d  projection + write-back %d cycles' % (o[1]-o[0], o[2]-o[1], o[3]-o[2]))
print('   projection: radius known +%d, %d Michelot passes until +%d, projected vector in LDS +%d, written back +%d' % (o[7]-o[2], o[4] & 0xffff, o[6]-o[2], o[5]-o[2], o[3]-o[2]))
o = [int(v) for v in out[8:13]]
print('spread projection (workgroup 0 of the last atom launch): candidates + radius %d cycles, projection %d cycles (%d exchanges of the search + the last one), tail %d' % (o[1]-o[0], o[2]-o[1], o[4], o[3]-o[2]))
