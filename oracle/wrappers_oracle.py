"""ORACLE (test infrastructure, NOT product code): CPU restatements of the
wrapper loops that feed the hot path, built on oracle/somf_oracle.py.

  fmri_fit        : modl/decomposition/fmri.py:423-546 (_compute_components) + :549-556 (_flip),
                    on raw 2-D records.  PINNED by tests/golden/fmri.npz, recorded from the reference's
                    own two functions (taken out of fmri.py at run time by make_golden.gen_fmri and run
                    against stand-ins for the nilearn masking layer).  fmri.py:460 rebinds `method` to the
                    dict of aggregation modes, so the tests on the method NAME at :508 ('gram' switch at
                    epoch 5), :511 ('reducing ratio') and :535 (sample_indices for 'average' / 'gram')
                    never fire in the reference: `intended_schedules=False` (default) restates that
                    EFFECTIVE behaviour, True what the code was meant to do.
  recsys_*        : modl/decomposition/recsys.py (pinned by tests/golden/recsys.npz).
"""
from math import sqrt

import numpy as np
import scipy
import scipy.sparse as sp
from numpy import linalg

from . import somf_oracle as orc

FMRI_METHODS = {'masked': ('masked', 'masked'), 'dictionary only': ('full', 'full'), 'gram': ('masked', 'masked'),
                'average': ('average', 'average'), 'reducing ratio': ('masked', 'masked')}


def fmri_fit(records, method='masked', step_size=1, n_components=20, n_epochs=1, alpha=0.1, dict_init=None,
             random_state=None, batch_size=20, reduction=1, learning_rate=1, positive=False, intended_schedules=False):
    if dict_init is not None:
        dict_init = np.asarray(dict_init)[:n_components]
        n_components = dict_init.shape[0]
    rng = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) else random_state
    if method == 'sgd':
        optimizer, G_agg, Dx_agg, reduction = 'sgd', 'full', 'full', 1
    else:
        G_agg, Dx_agg = FMRI_METHODS[method]
        optimizer = 'variational'
    lengths = [r.shape[0] for r in records]
    dtype = records[-1].dtype
    idx = np.zeros(len(records) + 1, dtype=int)
    idx[1:] = np.cumsum(lengths)
    pr = orc.SomfParams(n_components=n_components, code_alpha=alpha, code_l1_ratio=0, comp_l1_ratio=1,
                        comp_pos=positive, reduction=reduction, Dx_agg=Dx_agg, optimizer=optimizer,
                        step_size=step_size, G_agg=G_agg, learning_rate=learning_rate, batch_size=batch_size,
                        random_state=rng)
    st = orc.prepare(pr, n_samples=int(idx[-1]) + 1, n_features=records[0].shape[1],
                     X=None if dict_init is None else dict_init.astype(dtype), dtype=dtype)
    named = method if intended_schedules else None          # fmri.py:460: the name is gone after the rebinding
    for i in range(n_epochs):
        if named == 'gram' and i == 5:
            pr.G_agg, pr.Dx_agg = 'full', 'average'
            st.G = st.D.dot(st.D.T)
            if st.Dx_average is None:
                st.Dx_average = np.zeros((st.code.shape[0], n_components), dtype=dtype)
        if named == 'reducing ratio':
            reduction = 1 + (reduction - 1) / sqrt(i + 1)
            pr.reduction = reduction
        for record in rng.permutation(len(records)):
            data = records[record].astype(dtype)
            perm = rng.permutation(data.shape[0])
            si = np.arange(idx[record], idx[record + 1])[perm] if named in ('average', 'gram') else None
            orc.partial_fit(st, pr, data[perm], sample_indices=si)
    D = st.D.copy()
    for comp in D:
        if np.sum(comp < 0) > np.sum(comp > 0):
            comp *= -1
    return D, st


# --------------------------------------------------------------------------
# RecsysDictFact (modl/decomposition/recsys.py), pinned by tests/golden/recsys.npz

def recsys_biases(X, beta=0):
    """recsys.py:268-306"""
    X = sp.csr_matrix(X.copy())
    acc_u, acc_m = np.zeros(X.shape[0]), np.zeros(X.shape[1])
    n_u, n_m = X.getnnz(axis=1), X.getnnz(axis=0)
    n_u[n_u == 0] = 1
    n_m[n_m == 0] = 1
    avg = np.mean(X.data)
    for _ in range(2):
        w_u = (np.asarray(X.sum(axis=1))[:, 0] + avg * beta) / (n_u + beta)
        for i, (l, r) in enumerate(zip(X.indptr[:-1], X.indptr[1:])):
            X.data[l:r] -= w_u[i]
        w_m = np.asarray(X.sum(axis=0))[0] / (n_m + beta)
        X.data -= w_m.take(X.indices, mode='clip')
        acc_u += w_u
        acc_m += w_m
    return acc_u, acc_m


def recsys_fit(X, alpha=1.0, beta=0.0, n_components=30, learning_rate=1.0, batch_size=1, n_epochs=1,
               random_state=None, detrend=False):
    """recsys.py:81-213.  Returns a dict of the fitted attributes."""
    X = sp.csr_matrix(X, dtype=X.dtype if X.dtype in (np.float32, np.float64) else np.float64, copy=True)
    dtype = X.dtype
    n, p = X.shape
    k = n_components
    rng = np.random.RandomState(random_state)
    out = {}
    if detrend:
        rm, cm = recsys_biases(X, beta)
        for i in range(n):
            X.data[X.indptr[i]:X.indptr[i + 1]] -= rm[i]
        X.data -= cm.take(X.indices, mode='clip')
        out['row_mean'], out['col_mean'] = rm, cm
    D = rng.randn(k, p).astype(dtype)
    D /= np.sqrt(np.sum(D ** 2, axis=1))[:, None]
    code = np.zeros((n, k), dtype=dtype)

    def solve_row(i):
        s, e = X.indptr[i], X.indptr[i + 1]
        sub, xs = X.indices[s:e], X.data[s:e]
        Ds = D[:, sub]
        G = Ds.dot(Ds.T)
        G.flat[::k + 1] += alpha / (p / len(sub))
        return linalg.solve(G, Ds.dot(xs)), sub, xs

    def refit():
        for i in range(n):
            if X.indptr[i + 1] > X.indptr[i]:
                code[i] = solve_row(i)[0]
    refit()
    fni = np.zeros(p, dtype=int)
    comp_norm = np.zeros(k, dtype=dtype)
    Cm = np.zeros((k, k), dtype=dtype)
    B = np.zeros((k, p), dtype=dtype)
    ger, = scipy.linalg.get_blas_funcs(('ger',), (Cm, D))
    n_iter = 0
    for _ in range(n_epochs):
        perm = rng.permutation(n)
        for b0 in range(0, n, batch_size):
            batch = perm[b0:b0 + batch_size]
            bs = len(batch)
            n_iter += bs
            w = orc.batch_weight(n_iter, bs, learning_rate, 0)
            for i in batch:                                              # recsys.py:168-185
                if X.indptr[i + 1] - X.indptr[i] != 0:
                    c, sub, xs = solve_row(i)
                    fni[sub] += 1
                    code[i] = c
                    w_B = np.minimum(1, w * n_iter / fni[sub])
                    B[:, sub] *= 1 - w_B
                    B[:, sub] += np.outer(code[i], xs * w_B)
            Cm *= 1 - w
            Cm += w / bs * code[batch].T.dot(code[batch])
            subset = np.unique(np.concatenate([X.indices[X.indptr[i]:X.indptr[i + 1]] for i in batch]))
            Ds = D[:, subset]                                             # recsys.py:187-213
            gs = B[:, subset]
            gs -= Cm.dot(Ds)
            order = rng.permutation(k)
            comp_norm += np.sum(Ds ** 2, axis=1)
            for j in order:
                gs = ger(1.0, Cm[j], Ds[j], a=gs, overwrite_a=True)
                if Cm[j, j] > 1e-20:
                    Ds[j] = gs[j] / Cm[j, j]
                nrm, lim = sqrt(np.sum(Ds[j] ** 2)), sqrt(comp_norm[j])
                if nrm > lim:
                    Ds[j] /= nrm / lim
                gs = ger(-1.0, Cm[j], Ds[j], a=gs, overwrite_a=True)
            comp_norm -= np.sum(Ds ** 2, axis=1)
            D[:, subset] = Ds
    refit()
    out.update(D=D, code=code, C=Cm, B=B, comp_norm=comp_norm, X=X)
    return out


def recsys_predict(fit, Xp, detrend=False):
    Xp = sp.csr_matrix(Xp)
    data = np.zeros(Xp.nnz)
    orc.predict_csr(data, Xp.indices, Xp.indptr, fit['code'], fit['D'])
    if detrend:
        for i in range(Xp.shape[0]):
            data[Xp.indptr[i]:Xp.indptr[i + 1]] += fit['row_mean'][i]
        data += fit['col_mean'].take(Xp.indices, mode='clip')
    return data
