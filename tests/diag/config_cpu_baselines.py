"""CPU-oracle throughput on bounded prefixes of the inputs of scripts/bench_configs.py (BASELINE configs 2-4), on the
host cores of the box it runs on.  Lives under tests/ because it runs the oracle (test infrastructure).

    python tests/diag/config_cpu_baselines.py [--only c2,c3,c4] [--budget 15]
Prints one JSON line per configuration."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))

import bench_configs as bc                                   # the input generators (no oracle in there)
from oracle import somf_oracle as orc
from oracle import wrappers_oracle as wo


def _sliding_patches(img, m):
    from numpy.lib.stride_tricks import sliding_window_view
    rs = np.random.RandomState(0)
    win = sliding_window_view(img, (8, 8, img.shape[2]))
    idx = np.c_[np.where(np.ones(win.shape[:3]))][rs.permutation(win.shape[0] * win.shape[1])[:m]]
    P = win[tuple(idx.T)].astype(np.float64)                 # (m, 8, 8, c)
    P -= P.mean(axis=(1, 2))[:, None, None, :]
    sd = np.sqrt((P ** 2).sum(axis=(1, 2)))
    sd[sd == 0] = 1
    P /= sd[:, None, None, :] * np.sqrt(P.shape[3])
    return P.reshape(len(P), -1)


def c2(budget):
    P = _sliding_patches(bc.synth_image(512, 512, 1), 20000)
    m = len(P)
    pr = orc.SomfParams(n_components=256, batch_size=100, reduction=10, code_alpha=0.1, code_l1_ratio=1, comp_l1_ratio=0,
                        learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0, tol=1e-2)
    st = orc.prepare(pr, n_samples=m, X=P[:256])
    t0, done = time.perf_counter(), 0
    for r0 in range(0, m, 1000):
        orc.partial_fit(st, pr, P[r0:r0 + 1000], np.arange(r0, min(m, r0 + 1000)))
        done = min(m, r0 + 1000)
        if time.perf_counter() - t0 > budget:
            break
    return dict(config='C2', samples_per_s=done / (time.perf_counter() - t0), sample='%d patches' % done)


def c3(budget):
    recs, init = bc.fmri_records(n_records=4)
    t0 = time.perf_counter()
    wo.fmri_fit(recs, method='masked', n_components=70, reduction=12, batch_size=20, alpha=1e-3, learning_rate=0.92,
                dict_init=init, random_state=0, n_epochs=1)
    return dict(config='C3', samples_per_s=len(recs) * 176 / (time.perf_counter() - t0), sample='%d records' % len(recs))


def c4(budget):
    X = bc.ml10m_like(nnz=3_000_000)[:1500]
    t0 = time.perf_counter()
    wo.recsys_fit(X, alpha=1, beta=.1, n_components=50, learning_rate=.95, batch_size=10, n_epochs=1, random_state=0,
                  detrend=True)
    return dict(config='C4', samples_per_s=X.shape[0] / (time.perf_counter() - t0), sample='%d rows' % X.shape[0])


def c6(budget):
    """one minibatch (two if the budget allows) of the HCP configuration on the oracle"""
    kw = dict(bc.HCP_KW)
    b, k, p = kw['batch_size'], kw['n_components'], 200000
    X = bc.hcp_rows(max(2 * b, k), p, 0).numpy()
    pr = orc.SomfParams(**kw)
    st = orc.prepare(pr, n_samples=X.shape[0], X=X[:k])
    t0, done = time.perf_counter(), 0
    for r0 in range(0, 2 * b, b):
        orc.partial_fit(st, pr, X[r0:r0 + b], np.arange(r0, r0 + b))
        done += b
        if time.perf_counter() - t0 > budget:
            break
    return dict(config='C6', samples_per_s=done / (time.perf_counter() - t0), sample='%d records' % done)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='c2,c3,c4')
    ap.add_argument('--budget', type=float, default=15.0)
    a = ap.parse_args()
    for name in a.only.split(','):
        out = dict(c2=c2, c3=c3, c4=c4, c6=c6)[name](a.budget)
        out['cores'] = os.cpu_count()
        print(json.dumps(out), flush=True)
