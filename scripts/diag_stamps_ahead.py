"""In-kernel phase stamps (workgroup 0, last full block launch) of the blocked dictionary update in look-ahead mode
(modl_debug_set(MODL_DEBUG_BCD_ACC, 2)), next to the accumulator mode's; diagnostics build."""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_BCD_ACC
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 4096)
for mode in (1, 2):
    check(lib.modl_debug_set(DEBUG_BCD_ACC, mode))
    for r in (10, 1):
        est = DictFact(n_components=256, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
        est.prepare(n_samples=4096, X=X[:256])
        est.partial_fit(X[:2048])
        out = (C.c_ulonglong * 48)()
        check(lib.modl_somf_debug_stamps(est._backend.plan, out))
        o = [float(v) for v in out]
        if mode == 2:
            print('mode 2 r=%g: total %d | shadow workers: operands requested %d | head: entry->loads %d, loads+sums->LDS %d, transform + barrier 1 %d | '
                  'after barrier 1: chain done %d, helper %d, product done %d, N\' in LDS %d, pieces issued %d, barrier 2 %d | apply %d  correction %d  epilogue %d' % (
                      r, o[7] - o[0], o[21] - o[0], o[16] - o[0], o[17] - o[16], o[1] - o[17], o[2] - o[1], o[14] - o[1], o[22] - o[1], o[23] - o[1],
                      o[12] - o[1], o[4] - o[1], o[5] - o[4], o[6] - o[5], o[7] - o[6]))
        else:
            print('mode 1 r=%g: total %d | head: entry->loads %d, loads+sums->LDS %d, barrier 1 %d | after barrier 1: chain done %d, helper %d, '
                  'loads requested %d, product done %d, barrier 2 %d | D %d  E %d  F %d  G %d' % (
                      r, o[12] - o[0], o[16] - o[0], o[17] - o[16], o[1] - o[17], o[2] - o[1], o[14] - o[1], o[13] - o[1], o[22] - o[1],
                      o[4] - o[1], o[5] - o[4], o[6] - o[5], o[7] - o[6], o[12] - o[7]))
        print('      chain steps: setup %d  0-7 %d  8-15 %d  16-23 %d  24-31 %d' % (o[8] - o[1], o[9] - o[8], o[10] - o[9], o[11] - o[10], o[2] - o[11]))
check(lib.modl_debug_set(DEBUG_BCD_ACC, 1))
