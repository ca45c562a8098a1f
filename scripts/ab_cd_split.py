"""A/B of the two coordinate-descent kernels on the GPU (cd_solver.hip vs cd_split.hip): the results must be
bit-identical (codes and sweep counts); prints the time of the solve alone at several shapes.
    python scripts/ab_cd_split.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from modl_amd import dict_fact_fast as fast  # noqa: E402
from modl_amd._lib import lib, check, DEBUG_CD_SPLIT  # noqa: E402


def case(dt, k, b, p, alpha, pos=False, l1=1.0, tol=1e-2, mi=100, seed=0, dens=0.1):
    rs = np.random.RandomState(seed + k + b)
    D = rs.randn(k, p).astype(dt)
    D /= np.sqrt((D ** 2).sum(1))[:, None]
    X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < dens)).dot(D) + 0.1 * rs.randn(b, p)).astype(dt))
    G = np.ascontiguousarray(D.dot(D.T).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray(X.dot(D.T).astype(dt))
    idx = np.arange(b, dtype=np.int64)
    out = {}
    for split in (0, 1):
        check(lib.modl_debug_set(DEBUG_CD_SPLIT, split))
        best = 1e9
        for rep in range(3):
            code = np.ones((b, k), dtype=dt)
            sw = np.zeros(b, dtype=np.int32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fast._enet_regression_single_gram(G, Dx.copy(), X, code, idx, l1, alpha, pos, tol, mi, sweeps=sw)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out[split] = (code, sw, best)
    check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
    same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    print('%s k=%4d b=%4d p=%5d alpha=%.2f pos=%d l1=%.1f: identical=%s sweeps mean %.1f max %d  (host-timed call: old %.2f ms, '
          'split %.2f ms)' % (np.dtype(dt).name, k, b, p, alpha, pos, l1, same, out[0][1].mean(), out[0][1].max(),
                              out[0][2] * 1e3, out[1][2] * 1e3), flush=True)
    if not same:
        d = np.abs(out[0][0] - out[1][0])
        print('   max |diff| %.3e, rows differing %d, sweeps differing %d' % (d.max(), (d.max(1) > 0).sum(),
                                                                             (out[0][1] != out[1][1]).sum()))
    return same


if __name__ == '__main__':
    ok = True
    for dt in (np.float32, np.float64):
        for (k, b, p, alpha) in ((256, 256, 1000, 0.3), (256, 64, 300, 0.3), (200, 33, 400, 0.2), (128, 40, 300, 0.3),
                                 (100, 17, 200, 0.3), (512, 24, 700, 0.3), (330, 9, 600, 0.3), (256, 40, 12, 0.05),
                                 (250, 100, 64, 0.1), (1024, 12, 1200, 0.3), (600, 7, 900, 0.3)):
            ok &= case(dt, k, b, p, alpha)
            ok &= case(dt, k, b, p, alpha, pos=True, l1=0.7)
    print('ALL IDENTICAL' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)
