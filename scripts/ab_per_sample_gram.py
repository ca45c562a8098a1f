"""A/B of the code solve with one Gram matrix per sample (G_agg = Dx_agg = 'average', l1 codes) at sizes that are not
one of the four-wavefront solver's strides: zero-padded slots + that solver (default) against the general one-wavefront
kernel (modl_debug_set(MODL_DEBUG_CD_SPLIT, 0)).  Timing through the host wrapper of the C-ABI regression entry point (the same
host-to-device copies on both sides)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from modl_amd import dict_fact_fast as fast
from modl_amd._lib import lib, check, DEBUG_CD_SPLIT
rs = np.random.RandomState(0)
for k, b in ((70, 64), (200, 64), (320, 32), (600, 16)):
    p = 2 * k
    Gm = np.empty((b, k, k), dtype=np.float32); Dx = np.empty((b, k), dtype=np.float32); X = np.empty((b, p), dtype=np.float32)
    for i in range(b):
        D = rs.randn(k, p).astype(np.float32); D /= np.sqrt((D ** 2).sum(1))[:, None]
        X[i] = ((rs.randn(k) * (rs.rand(k) < 0.1)).dot(D) + 0.1 * rs.randn(p)).astype(np.float32)
        G = D.dot(D.T); Gm[i] = (G + G.T) / 2; Dx[i] = X[i].dot(D.T)
    idx = np.arange(b, dtype=np.int64)
    res = {}
    for split in (0, 1):
        check(lib.modl_debug_set(DEBUG_CD_SPLIT, split))
        ts = []
        for rep in range(5):
            code = np.ones((b, k), dtype=np.float32)
            t0 = time.perf_counter()
            fast._enet_regression_multi_gram(Gm, Dx.copy(), X, code, idx, 0.9, 0.3, False, 1e-2, 100)
            ts.append((time.perf_counter() - t0) * 1e3)
        res[split] = (sorted(ts)[2], code)
    check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
    print('k = %4d, b = %3d: one-wavefront kernel %.3f ms, padded slots + four-wavefront solver %.3f ms; identical: %s' %
          (k, b, res[0][0], res[1][0], np.array_equal(res[0][1], res[1][1])))
