"""Throughput of BASELINE.json's configs 2-4 on one MI355X, at the synthetic shapes of SURVEY.md 8(d)
(C2 ImageDictFact 8x8 patches k=256; C3 fMRI-shaped records k=70 r=12; C4 MovieLens-10M-shaped CSR k=50).
These are parity-test configurations, not bench lines (bench.py measures the headline metric); this script
records what the wrappers deliver end to end (host loops included).  The CPU oracle on a bounded prefix of the same
inputs is timed by tests/diag/config_cpu_baselines.py (the oracle is test infrastructure: only tests/ runs it).

    python scripts/bench_configs.py [--only c2,c3,c4] [--c4-batches 2000]
Prints one JSON line per configuration."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth_image(h, w, c, seed=0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, c))
    for ch in range(c):
        for _ in range(6):
            fy, fx, ph = rs.uniform(0.05, 0.6), rs.uniform(0.05, 0.6), rs.uniform(0, 6.28)
            img[:, :, ch] += np.sin(fy * yy + fx * xx + ph)
    img += 0.05 * rs.randn(h, w, c)
    return (img - img.min()) / (img.max() - img.min())


def fmri_records(n_records=40, n_rows=176, p=60000, k=70, seed=0, dtype=np.float32):
    rs = np.random.RandomState(seed)
    maps = np.zeros((k, p), dtype=dtype)
    width = p // k
    for j in range(k):
        maps[j, j * width:(j + 1) * width] = 1.0
    recs = []
    for _ in range(n_records):
        L = rs.randn(n_rows, k).astype(dtype)
        R = L @ maps + 0.01 * rs.randn(n_rows, p).astype(dtype)
        R -= R.mean(axis=1, keepdims=True)
        R /= R.std(axis=1, keepdims=True)
        recs.append(R.astype(dtype))
    init = maps + rs.randn(k, p).astype(dtype)
    return recs, init


def ml10m_like(n_users=69878, n_items=10677, nnz=10_000_000, seed=0, dtype=np.float64):
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    pu = 1.0 / np.arange(1, n_users + 1) ** 0.6
    pi = 1.0 / np.arange(1, n_items + 1) ** 0.9
    u = rs.choice(n_users, size=nnz, p=pu / pu.sum())
    i = rs.choice(n_items, size=nnz, p=pi / pi.sum())
    key = np.unique(u.astype(np.int64) * n_items + i)
    u, i = key // n_items, key % n_items
    v = rs.randint(1, 11, size=len(key)).astype(dtype) / 2
    X = sp.csr_matrix((v, (u, i)), shape=(n_users, n_items))
    X = X[rs.permutation(n_users)]
    return sp.csr_matrix(X)


def c2(args):
    import torch
    from modl_amd.image import ImageDictFact
    img = synth_image(512, 512, 1)
    est = ImageDictFact(patch_size=(8, 8), n_components=256, method='masked', setting='dictionary learning',
                        random_state=0, n_epochs=1, max_patches=args.c2_patches)
    t0 = time.perf_counter()
    est.fit(img)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = est.n_iter_
    out = dict(config='C2 ImageDictFact 512x512x1, 8x8 patches (p=64), k=256, b=100, r=10, l1 codes', patches=int(n),
               seconds=dt, samples_per_s=n / dt, finite=bool(np.all(np.isfinite(est.components_))))
    return out


def c3(args):
    import torch
    from modl_amd.fmri import fMRIDictFact
    recs, init = fmri_records(n_records=args.c3_records)
    est = fMRIDictFact(method='masked', n_components=70, reduction=12, batch_size=20, alpha=1e-3, learning_rate=0.92,
                       dict_init=init, random_state=0, n_epochs=1)
    t0 = time.perf_counter()
    est.fit(recs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = est.dict_fact_.n_iter_
    out = dict(config='C3 fMRI-shaped: %d records x 176 x p=60000 f32, k=70, r=12, b=20, ridge codes, l1 atoms' % len(recs),
               samples=int(n), seconds=dt, samples_per_s=n / dt, io_s=est.io_time_, fit_s=est.cpu_time_,
               finite=bool(np.all(np.isfinite(est.components_))))
    return out


def c4(args):
    import torch
    from modl_amd.recsys import RecsysDictFact
    t0 = time.perf_counter()
    X = ml10m_like(nnz=args.c4_nnz)
    gen_s = time.perf_counter() - t0
    n_rows = min(X.shape[0], args.c4_batches * 10)
    Xs = X[:n_rows]
    est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1,
                         random_state=0)
    t0 = time.perf_counter()
    est.fit(Xs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    score = est.score(Xs)
    out = dict(config='C4 MovieLens-10M-shaped CSR %dx%d (%d ratings; first %d rows fitted), k=50, b=10, detrend'
                      % (X.shape[0], X.shape[1], X.nnz, n_rows),
               samples=int(n_rows), ratings=int(Xs.nnz), seconds=dt, samples_per_s=n_rows / dt, ratings_per_s=Xs.nnz / dt,
               train_rmse=float(score), score_s=time.perf_counter() - t0, gen_s=gen_s)
    return out


HCP_KW = dict(n_components=1024, batch_size=200, reduction=20, learning_rate=0.92, code_alpha=1e-4, code_l1_ratio=0,
              comp_l1_ratio=1, comp_pos=True, G_agg='masked', Dx_agg='masked', random_state=0)


def hcp_rows(n, p, seed, device=None, k0=64):
    """rest-fMRI-like rows (n, p) float32: a few dozen smooth positive maps mixed by random loadings + noise, rows
    standardised as fmri.py does; produced on the device when one is given (the same rows on the host otherwise)"""
    import torch
    dev = device if device is not None else torch.device('cpu')
    g = torch.Generator(device=dev).manual_seed(seed)
    maps = torch.relu(torch.randn(k0, p, device=dev, generator=g) - 1.0)
    L = torch.randn(n, k0, device=dev, generator=g)
    X = L @ maps + 0.5 * torch.randn(n, p, device=dev, generator=g)
    X -= X.mean(dim=1, keepdim=True)
    X /= X.std(dim=1, keepdim=True)
    return X.float().contiguous()


def c6(args):
    """The reference's own largest published run (exps/hcp/decompose_hcp.py:50-60): 1024 maps, minibatches of 200
    records, reduction 20, ridge codes, positive l1 atoms - at p = 200 000 voxels, DictFact on device-resident rows."""
    import torch
    from modl_amd import DictFact
    dev = torch.device('cuda')
    p, nb = 200000, args.c6_batches
    kw = dict(HCP_KW)
    b, k = kw['batch_size'], kw['n_components']
    X = hcp_rows(max(nb * b, k), p, 0, dev)
    est = DictFact(**kw)
    est.prepare(n_samples=X.shape[0], X=X[:k])
    est.partial_fit(X[:2 * b], np.arange(2 * b))             # warm-up (allocations, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    est.partial_fit(X[2 * b:nb * b], np.arange(2 * b, nb * b))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = (nb - 2) * b
    be = est._backend
    be.prof_enable(True)
    be.prof_reset()
    est.partial_fit(X[:2 * b], np.arange(2 * b))
    torch.cuda.synchronize()
    prof = {name: v['ms'] / v['calls'] for name, v in be.prof_get().items() if v['calls']}
    be.prof_enable(False)
    D = est._backend.Dt
    return dict(config='C6 the reference\'s HCP run (exps/hcp/decompose_hcp.py:50-60): k=1024, b=200, r=20, ridge codes, '
                       'positive l1 atoms, p=%d f32, %d minibatches timed' % (p, nb - 2),
                samples=int(n), seconds=dt, samples_per_s=n / dt, ms_per_minibatch=dt / (nb - 2) * 1e3,
                sections_ms=prof, finite=bool(torch.isfinite(D).all().item()),
                nonneg=bool((D >= 0).all().item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='c2,c3,c4')
    ap.add_argument('--c6-batches', type=int, default=8)
    ap.add_argument('--c2-patches', type=int, default=None)
    ap.add_argument('--c3-records', type=int, default=40)
    ap.add_argument('--c4-batches', type=int, default=7000)
    ap.add_argument('--c4-nnz', type=int, default=10_000_000)
    ap.add_argument('--debug-set', action='append', default=[], metavar='WHAT=VALUE', help='modl_debug_set before anything runs (A/B runs)')
    args = ap.parse_args()
    for item in args.debug_set:
        from modl_amd._lib import lib, check
        what, value = item.split('=')
        check(lib.modl_debug_set(int(what), int(value)), 'modl_debug_set')
    for name in args.only.split(','):
        out = dict(c2=c2, c3=c3, c4=c4, c6=c6)[name](args)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
