"""In-kernel stamps of the blocked dictionary update at the C5 shape (p = 200 000, k = 256, reduction 12: 96 features per
workgroup, bcd_block_kernel<3, 8>), workgroup 0 of the last full block launch."""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')     # the stamps only exist in the diagnostics build (libmodl_hip_diag.so)
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
dev = torch.device('cuda')
p, n, b = 200000, 512, 64
X = bench.M1Stream(p, 1234, dev).rows(0, n)
est = DictFact(n_components=256, batch_size=b, reduction=12, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=n, X=X[:256])
est.partial_fit(X[:256])
out = (C.c_ulonglong * 48)()
check(lib.modl_somf_debug_stamps(est._backend.plan, out))
o = [float(v) for v in out]
print('cycles (workgroup 0, last full block launch): total %d' % (o[12] - o[0]))
rows = [('B: Gram of the previous block -> barrier 1', o[1] - o[0]),
        ('C resolver wave', o[2] - o[1]),
        ('C workers: all loads requested', o[13] - o[1]),
        ('C workers: main MFMA product + a-tile staging', o[3] - o[13]),
        ('barrier 2 (after the longer of the two)', o[4] - o[1]),
        ('D apply -> barrier 3', o[5] - o[4]),
        ('E correction + cross-wave -> barrier 4', o[6] - o[5]),
        ('F epilogue -> barrier 5', o[7] - o[6]),
        ('G Gram contribution', o[12] - o[7])]
for nme, v in rows:
    print('   %-56s %8d' % (nme, v))
print('   product wave 1 loads requested %d, done %d, wave 2 done %d, wave 3 done %d (after barrier 1)' % (o[20] - o[1], o[15] - o[1], o[18] - o[1], o[19] - o[1]))
