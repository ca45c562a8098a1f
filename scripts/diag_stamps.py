import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')     # the stamps only exist in the diagnostics build (libmodl_hip_diag.so)
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 4096)
for r in (10, 1):
    est = DictFact(n_components=256, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=4096, X=X[:256])
    est.partial_fit(X[:2048])
    out = (C.c_ulonglong * 48)()
    check(lib.modl_somf_debug_stamps(est._backend.plan, out))
    o = [float(v) for v in out]
    print('r=%g cycles (workgroup 0, last full block launch): total %d' % (r, o[12] - o[0]))
    rows = [('B: Gram records of the previous block -> barrier 1', o[1] - o[0]),
            ('C resolver wave', o[2] - o[1]),
            ('C workers: all loads requested', o[13] - o[1]),
            ('C workers: main MFMA product + a-tile staging', o[3] - o[13]),
            ('barrier 2 (after the longer of the two)', o[4] - o[1]),
            ('D apply -> barrier 3', o[5] - o[4]),
            ('E correction + cross-wave -> barrier 4', o[6] - o[5]),
            ('F epilogue -> barrier 5', o[7] - o[6]),
            ('G Gram record', o[12] - o[7])]
    for n, v in rows:
        print('   %-56s %8d' % (n, v))
    print('   G detail (cycles after barrier 5): wave 0 tile formed %d, its entries added %d; wave 1 done %d, wave 2 %d, wave 3 %d' % (
        o[30] - o[7], o[12] - o[7], o[33] - o[7], o[35] - o[7], o[34] - o[7]))
    print('   helper wave done %d, product wave 1 loads requested %d, done %d, wave 2 done %d, wave 3 done %d (after barrier 1)' % (o[14] - o[1], o[20] - o[1], o[15] - o[1], o[18] - o[1], o[19] - o[1]))
    print('   first steps (cycles after barrier 1):', [int(o[24 + u] - o[1]) for u in range(8)])
    print('   product wave 2: operands requested %d, product issued %d (after barrier 1)' % (o[21] - o[1], o[22] - o[1]))
    if r == 10:
        print('   first riding tile of the last carrier launch: requests issued %d, first operands in LDS %d, contraction %d, epilogue %d (cycles)' % (out[41] - out[40], out[42] - out[40], out[43] - out[42], out[44] - out[43]))
    print('   B detail: entry -> record loads start %d, records summed + in LDS %d, barrier 1 %d' % (o[16] - o[0], o[17] - o[16], o[1] - o[17]))
    print('   resolve: setup %d  steps 0-7 %d  8-15 %d  16-23 %d  24-31+stores %d' % (o[8] - o[1], o[9] - o[8], o[10] - o[9], o[11] - o[10], o[2] - o[11]))
