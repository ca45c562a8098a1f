"""ORACLE (test infrastructure, NOT product code).

CPU restatement of MODL's per-minibatch SOMF loop: numpy (same BLAS the
reference reaches through ``ndarray.dot``) for the contractions, scipy's
``ger``/``posv`` where the reference calls them, and ``oracle/liboracle.so``
(plain C, see rk_oracle.c / somf_oracle.c) for what the reference keeps in
Cython.  Every function cites the reference lines it follows
(paths relative to /root/reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It is pinned against the real reference by the
golden vectors in ``tests/golden/`` (made by ``tests/golden/make_golden.py``,
which compiles and imports the reference in a scratch directory) — see
``tests/test_oracle_golden.py``.  Float summation order of the third-party
BLAS is not part of the reference's contract: numeric parity is a tolerance
(1e-5 rel. Frobenius for f32, 1e-10 for f64), integer draws are bit-exact.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import scipy.linalg
from scipy.linalg import lapack

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_INT = np.iinfo(np.int64).max


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference tree exists)."""
    so = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('rk_oracle.c', 'somf_oracle.c', 'somf_oracle_impl.inc')]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, 'liboracle.so'], stdout=subprocess.DEVNULL)
    if os.path.isdir('/root/reference/modl/utils/randomkit'):
        subprocess.call(['make', '-C', _HERE, 'ref'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.ork_state_new.restype = C.c_void_p
        L.ork_state_new.argtypes = [C.c_uint64]
        L.ork_state_free.argtypes = [C.c_void_p]
        L.ork_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.ork_random.restype = C.c_uint32
        L.ork_random.argtypes = [C.c_void_p]
        L.ork_interval.restype = C.c_uint64
        L.ork_interval.argtypes = [C.c_void_p, C.c_uint64]
        L.ork_double.restype = C.c_double
        L.ork_double.argtypes = [C.c_void_p]
        L.ork_binomial.restype = C.c_long
        L.ork_binomial.argtypes = [C.c_void_p, C.c_long, C.c_double]
        L.ork_shuffle_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.ork_permutation.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.ork_shuffle_trace.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.ork_apply_swaps_rows.argtypes = [C.c_void_p, C.c_long, C.c_size_t, C.c_void_p]
        L.ork_apply_swaps_i64.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.ork_sampler_new.restype = C.c_void_p
        L.ork_sampler_new.argtypes = [C.c_long, C.c_int, C.c_int, C.c_uint64]
        L.ork_sampler_free.argtypes = [C.c_void_p]
        L.ork_sampler_yield.restype = C.c_long
        L.ork_sampler_yield.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.osf_batch_weight.restype = C.c_double
        L.osf_batch_weight.argtypes = [C.c_long, C.c_long, C.c_double, C.c_double]
        L.osf_predict_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                                      C.c_void_p, C.c_long]
        for sfx, ct in (('f32', C.c_float), ('f64', C.c_double)):
            f = getattr(L, 'osf_enet_regression_gram_' + sfx)
            f.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p,
                          C.c_long, C.c_long, ct, ct, C.c_int, ct, C.c_int, C.c_void_p]
            f = getattr(L, 'osf_update_G_average_' + sfx)
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long]
            f = getattr(L, 'osf_enet_norm_' + sfx)
            f.restype = ct
            f.argtypes = [C.c_void_p, C.c_long, ct]
            f = getattr(L, 'osf_enet_scale_' + sfx)
            f.argtypes = [C.c_void_p, C.c_long, ct, ct]
            f = getattr(L, 'osf_enet_projection_' + sfx)
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_long, ct, ct]
    return _LIB


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return 'f32'
    if dtype == np.float64:
        return 'f64'
    raise TypeError('float32 or float64 expected, got %s' % dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------
# RNG / sampler   (modl/utils/randomkit/random_fast.pyx:49-150, sampler.pyx:9-70)

class OracleRandomState:
    def __init__(self, seed):
        self._h = lib().ork_state_new(int(seed) & 0xFFFFFFFFFFFFFFFF)

    def __del__(self):
        if getattr(self, '_h', None) and _LIB is not None:
            _LIB.ork_state_free(self._h)
            self._h = None

    def randint(self, high):                      # random_fast.pyx:76
        return int(lib().ork_interval(self._h, int(high)))

    def random_u32(self):
        return int(lib().ork_random(self._h))

    def double(self):
        return float(lib().ork_double(self._h))

    def binomial(self, n, p):                     # random_fast.pyx:146
        return int(lib().ork_binomial(self._h, int(n), float(p)))

    def permutation(self, n):                     # random_fast.pyx:79
        out = np.empty(int(n), dtype=np.int64)
        lib().ork_permutation(self._h, _p(out), int(n))
        return out

    def shuffle(self, x):                         # random_fast.pyx:87 (1-D int64 in place)
        assert x.dtype == np.int64 and x.ndim == 1 and x.flags.c_contiguous
        lib().ork_shuffle_i64(self._h, _p(x), x.shape[0])

    def shuffle_with_trace(self, arrays):         # random_fast.pyx:127
        n = arrays[0].shape[0]
        trace = np.empty(n, dtype=np.int64)
        swaps = np.zeros(n, dtype=np.int64)
        lib().ork_shuffle_trace(self._h, n, _p(trace), _p(swaps))
        for a in arrays:
            assert a.flags.c_contiguous and a.shape[0] == n
            if a.ndim == 1 and a.dtype == np.int64:
                lib().ork_apply_swaps_i64(_p(a), n, _p(swaps))
            else:
                lib().ork_apply_swaps_rows(_p(a), n, a.strides[0], _p(swaps))
        return trace


class OracleSampler:
    def __init__(self, range_, rand_size, replacement, random_seed):
        self.range = int(range_)
        self._h = lib().ork_sampler_new(self.range, int(bool(rand_size)), int(bool(replacement)),
                                        int(random_seed) & 0xFFFFFFFFFFFFFFFF)
        self._buf = np.empty(max(self.range, 1), dtype=np.int64)

    def __del__(self):
        if getattr(self, '_h', None) and _LIB is not None:
            _LIB.ork_sampler_free(self._h)
            self._h = None

    def yield_subset(self, reduction):
        n = lib().ork_sampler_yield(self._h, float(reduction), _p(self._buf))
        return self._buf[:n].copy()


def batch_weight(count, batch_size, learning_rate, offset=0.0):
    """dict_fact_fast.pyx:115-122"""
    return float(lib().osf_batch_weight(int(count), int(batch_size), float(learning_rate), float(offset)))


# --------------------------------------------------------------------------
# atom geometry (modl/utils/math/enet.pyx)

def enet_norm(v, l1_ratio):
    v = np.ascontiguousarray(v)
    return float(getattr(lib(), 'osf_enet_norm_' + _sfx(v.dtype))(_p(v), v.shape[0], l1_ratio))


def enet_scale(v, l1_ratio, radius=1.0):
    assert v.flags.c_contiguous
    getattr(lib(), 'osf_enet_scale_' + _sfx(v.dtype))(_p(v), v.shape[0], l1_ratio, radius)


def enet_projection(v, out, radius, l1_ratio):
    v = np.ascontiguousarray(v)
    assert out.flags.c_contiguous and out.dtype == v.dtype
    getattr(lib(), 'osf_enet_projection_' + _sfx(v.dtype))(_p(v), _p(out), v.shape[0], radius, l1_ratio)


# --------------------------------------------------------------------------
# code solvers (modl/decomposition/dict_fact_fast.pyx:33-215)

def enet_regression_single_gram(G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter,
                                sweeps=None):
    """dict_fact_fast.pyx:125-215.  In place on rows `indices` of `code`; Dx is
    overwritten by the solution in the ridge branch, as in the reference."""
    sfx = _sfx(code.dtype)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    b, k = Dx.shape
    if l1_ratio == 0:                                   # :174-197, posv on (G + alpha I)
        A = np.array(G, dtype=code.dtype, copy=True)
        A.flat[::k + 1] += code.dtype.type(alpha)
        posv = lapack.sposv if sfx == 'f32' else lapack.dposv
        _, sol, _info = posv(A, np.ascontiguousarray(Dx.T), lower=0)   # info ignored, :188-192
        Dx[:, :] = sol.T
        code[indices] = Dx
    else:
        G = np.ascontiguousarray(G)
        Dx = np.ascontiguousarray(Dx)
        X = np.ascontiguousarray(X)
        getattr(lib(), 'osf_enet_regression_gram_' + sfx)(
            _p(G), 0, _p(Dx), _p(X), X.shape[1], _p(code), _p(indices), b, k,
            l1_ratio, alpha, int(bool(positive)), tol, int(max_iter),
            _p(sweeps) if sweeps is not None else None)
    return code


def enet_regression_multi_gram(G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter,
                               sweeps=None):
    """dict_fact_fast.pyx:33-113 (one Gram per sample, G: (b, k, k))."""
    sfx = _sfx(code.dtype)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    b, k = Dx.shape
    if l1_ratio == 0:                                   # :82-94
        posv = lapack.sposv if sfx == 'f32' else lapack.dposv
        for ii in range(b):
            A = np.array(G[ii], copy=True)
            A.flat[::k + 1] += code.dtype.type(alpha)
            _, sol, _info = posv(A, np.ascontiguousarray(Dx[ii][:, None]), lower=0)
            code[indices[ii]] = sol[:, 0]
    else:
        G = np.ascontiguousarray(G)
        Dx = np.ascontiguousarray(Dx)
        X = np.ascontiguousarray(X)
        getattr(lib(), 'osf_enet_regression_gram_' + sfx)(
            _p(G), k * k, _p(Dx), _p(X), X.shape[1], _p(code), _p(indices), b, k,
            l1_ratio, alpha, int(bool(positive)), tol, int(max_iter),
            _p(sweeps) if sweeps is not None else None)
    return code


def update_G_average(G_average, G, w_sample):
    """dict_fact_fast.pyx:217-228"""
    sfx = _sfx(G_average.dtype)
    w_sample = np.ascontiguousarray(w_sample, dtype=G_average.dtype)
    G = np.ascontiguousarray(G, dtype=G_average.dtype)
    getattr(lib(), 'osf_update_G_average_' + sfx)(_p(G_average), _p(G), _p(w_sample),
                                                 G_average.shape[0], G_average.shape[1])
    return G_average


def predict_csr(data, indices, indptr, P, Q):
    """recsys_fast.pyx:10-38"""
    P = np.ascontiguousarray(P, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    lib().osf_predict_csr(_p(data), _p(np.ascontiguousarray(indices, dtype=np.int32)),
                          _p(np.ascontiguousarray(indptr, dtype=np.int32)), _p(P), P.shape[0], P.shape[1],
                          _p(Q), Q.shape[1])


# --------------------------------------------------------------------------
# estimator state and the minibatch loop (modl/decomposition/dict_fact.py)

@dataclass
class SomfParams:
    """Constructor arguments of the reference estimator (dict_fact.py:128-153)."""
    reduction: float = 1
    learning_rate: float = 1
    sample_learning_rate: float = 0.76
    Dx_agg: str = 'masked'
    G_agg: str = 'masked'
    optimizer: str = 'variational'
    dict_init: Optional[np.ndarray] = None
    code_alpha: float = 1
    code_l1_ratio: float = 1
    comp_l1_ratio: float = 0
    step_size: float = 1
    tol: float = 1e-2
    max_iter: int = 100
    code_pos: bool = False
    comp_pos: bool = False
    random_state: object = None
    n_epochs: int = 1
    n_components: int = 10
    batch_size: int = 10
    rand_size: bool = True
    replacement: bool = True
    n_threads: int = 1              # dict_fact.py:150: the code solve of a minibatch split over a thread pool (:584-621)


@dataclass
class SomfState:
    D: np.ndarray = None            # components_  (k, p)
    C: np.ndarray = None            # C_           (k, k)
    B: np.ndarray = None            # B_           (k, p)
    code: np.ndarray = None         # code_        (n, k)
    comp_norm: np.ndarray = None    # comp_norm_   (k,)
    G: Optional[np.ndarray] = None  # G_           (k, k), G_agg == 'full'
    Dx_average: Optional[np.ndarray] = None
    G_average: Optional[np.ndarray] = None
    labels: np.ndarray = None
    n_iter: int = 0
    sample_n_iter: np.ndarray = None
    rng: np.random.RandomState = None
    sampler: OracleSampler = None
    trace: Optional[list] = None    # when a list: one dict per minibatch (tests)
    sweeps: Optional[list] = None


def _check_rng(seed):
    if seed is None or isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    return seed


def prepare(pr: SomfParams, n_samples=None, n_features=None, dtype=None, X=None) -> SomfState:
    """dict_fact.py:381-489.  Raises where the reference returns
    the exception object (:422,:424)."""
    if X is not None:
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        if dtype is None:
            dtype = X.dtype
        if n_samples is None:
            n_samples = X.shape[0]
        if n_features is None:
            n_features = X.shape[1]
        elif n_features != X.shape[1]:
            raise ValueError('n_features and X does not match')
    else:
        if n_features is None or n_samples is None:
            raise ValueError('Either provide shape or data to function prepare.')
        if dtype is None:
            dtype = np.float64
    dtype = np.dtype(dtype)
    if dtype not in (np.float32, np.float64):
        raise ValueError('dtype should be float32 or float64')
    if pr.optimizer not in ('variational', 'sgd'):
        raise ValueError("optimizer should be 'variational' or 'sgd'")
    if pr.optimizer == 'sgd':                               # :425-428
        pr.reduction = 1
        pr.G_agg = 'full'
        pr.Dx_agg = 'full'
    k = pr.n_components
    st = SomfState()
    if pr.G_agg == 'average':
        st.G_average = np.zeros((n_samples, k, k), dtype=dtype)   # memmap of zeros in the reference
    if pr.Dx_agg == 'average':
        st.Dx_average = np.zeros((n_samples, k), dtype=dtype)
    st.C = np.zeros((k, k), dtype=dtype)
    st.B = np.zeros((k, n_features), dtype=dtype)
    st.rng = _check_rng(pr.random_state)
    if X is None:                                            # :450-455
        st.D = np.empty((k, n_features), dtype=dtype)
        st.D[:, :] = st.rng.randn(k, n_features)
    else:                                                    # :459-461 first k rows
        st.D = np.array(X[:k], dtype=dtype, copy=True, order='C')
    if pr.comp_pos:                                          # :462-464
        neg = st.D <= 0
        st.D[neg] = -st.D[neg]
    for i in range(st.D.shape[0]):                           # :465-468
        enet_scale(st.D[i], pr.comp_l1_ratio, 1.0)
    st.code = np.ones((n_samples, k), dtype=dtype)           # :470
    st.labels = np.arange(n_samples)
    st.comp_norm = np.zeros(k, dtype=dtype)
    if pr.G_agg == 'full':
        st.G = st.D.dot(st.D.T)
    st.n_iter = 0
    st.sample_n_iter = np.zeros(n_samples, dtype=np.int64)
    seed = st.rng.randint(MAX_INT)                           # :482
    st.sampler = OracleSampler(n_features, pr.rand_size, pr.replacement, seed)
    return st


def _gen_batches(n, b):
    start = 0
    while start < n:
        yield slice(start, min(start + b, n))
        start += b


def sub_indices(indices, batch):
    """modl/utils/__init__.py:4-27"""
    if indices is None:
        return np.arange(batch.start, batch.stop)
    if isinstance(indices, slice):
        return np.arange(indices.start + batch.start, indices.start + batch.stop)
    return indices[batch]


_POOLS = {}


def _pool(n):
    from concurrent.futures import ThreadPoolExecutor
    if n not in _POOLS:
        _POOLS[n] = ThreadPoolExecutor(n)                     # dict_fact.py:44-45
    return _POOLS[n]


def compute_code(st: SomfState, pr: SomfParams, X, idx, w_sample, subset):
    """dict_fact.py:577-648"""
    r = pr.reduction
    dt = st.D.dtype
    if pr.Dx_agg != 'full' or pr.G_agg != 'full':
        D_sub = st.D[:, subset]
    if pr.Dx_agg == 'full':
        Dx = X.dot(st.D.T)
    else:
        Dx = X[:, subset].dot(D_sub.T) * r
        if Dx.dtype != dt:
            Dx = Dx.astype(dt)
        if pr.Dx_agg == 'average':
            st.Dx_average[idx] *= 1 - w_sample[:, None]
            st.Dx_average[idx] += Dx * w_sample[:, None]
            Dx = st.Dx_average[idx]
    G_avg = None
    if pr.G_agg != 'full':
        G = D_sub.dot(D_sub.T) * r
        if G.dtype != dt:
            G = G.astype(dt)
        if pr.G_agg == 'average':
            G_avg = np.array(st.G_average[idx], copy=True)
            update_G_average(G_avg, G, w_sample)
            st.G_average[idx] = G_avg
    else:
        G = st.G
    sweeps = np.zeros(len(idx), dtype=np.int32) if st.sweeps is not None else None
    args = (pr.code_l1_ratio, pr.code_alpha, pr.code_pos, pr.tol, pr.max_iter)
    Dx = np.ascontiguousarray(Dx)
    nt = int(getattr(pr, 'n_threads', 1) or 1)
    if nt > 1 and pr.code_l1_ratio != 0:
        # dict_fact.py:584-586, 608-621: contiguous slices of the minibatch on a ThreadPoolExecutor (the compiled solver
        # releases the GIL, as the reference's nogil Cython does); a sample's solve does not depend on the others, so the
        # result is the serial run's bit for bit
        size_job = -(-len(idx) // nt)
        jobs = list(_gen_batches(len(idx), size_job))
        idx = np.ascontiguousarray(idx, dtype=np.int64)

        def job(batch):
            sw = sweeps[batch] if sweeps is not None else None
            if pr.G_agg == 'average':
                enet_regression_multi_gram(G_avg[batch], Dx[batch], X[batch], st.code, idx[batch], *args, sweeps=sw)
            else:
                enet_regression_single_gram(G, Dx[batch], X[batch], st.code, idx[batch], *args, sweeps=sw)
        list(_pool(nt).map(job, jobs))
    elif pr.G_agg == 'average':
        enet_regression_multi_gram(G_avg, Dx, X, st.code, idx, *args, sweeps=sweeps)
    else:
        enet_regression_single_gram(G, Dx, X, st.code, idx, *args, sweeps=sweeps)
    if sweeps is not None:
        st.sweeps.append(sweeps)


def update_stats(st: SomfState, pr: SomfParams, X, code, w):
    """dict_fact.py:559-575"""
    b = X.shape[0]
    if pr.optimizer == 'variational':
        st.C *= 1 - w
        st.C += w * code.T.dot(code) / b
        st.B *= 1 - w
        st.B += w * code.T.dot(X) / b
    else:
        st.C = code.T.dot(code) / b
        st.B = code.T.dot(X) / b


def update_dict(st: SomfState, pr: SomfParams, subset, w, order=None):
    """dict_fact.py:650-715"""
    dt = st.D.dtype
    k, p = st.D.shape
    s = subset.shape[0]
    rho = pr.comp_l1_ratio
    ger, = scipy.linalg.get_blas_funcs(('ger',), (st.C, st.D))
    Ds = st.D[:, subset]
    gs = np.asfortranarray(st.B[:, subset])        # gradient_ is F-ordered (:446-447, :532, :665)
    tmp = np.zeros(s, dtype=dt)
    if pr.G_agg == 'full' and s < p / 2.:
        st.G -= Ds.dot(Ds.T)
    gs -= st.C.dot(Ds)
    if order is None:
        order = st.rng.permutation(k)                # :672
    if pr.optimizer == 'variational':
        for j in order:
            st.comp_norm[j] += enet_norm(Ds[j], rho)
            gs = ger(1.0, st.C[j], Ds[j], a=gs, overwrite_a=True)
            if st.C[j, j] > 1e-20:
                Ds[j] = gs[j] / st.C[j, j]
            if pr.comp_pos:
                Ds[Ds < 0] = 0
            enet_projection(Ds[j], tmp, st.comp_norm[j], rho)
            Ds[j] = tmp
            st.comp_norm[j] -= enet_norm(Ds[j], rho)
            gs = ger(-1.0, st.C[j], Ds[j], a=gs, overwrite_a=True)
    else:                                            # :695-708
        for j in order:
            st.comp_norm[j] += enet_norm(Ds[j], rho)
        Ds += w * pr.step_size * gs
        for j in range(k):
            enet_projection(Ds[j], tmp, st.comp_norm[j], rho)
            Ds[j] = tmp
            st.comp_norm[j] -= enet_norm(Ds[j], rho)
    st.D[:, subset] = Ds
    if pr.G_agg == 'full':
        if s < p / 2.:
            st.G += Ds.dot(Ds.T)
        else:
            st.G[:] = st.D.dot(st.D.T)
    return order


def minibatch_step(st: SomfState, pr: SomfParams, X, idx, subset=None, order=None):
    """dict_fact.py:495-533 (one SOMF iteration)."""
    dt = st.D.dtype
    if subset is None:
        subset = st.sampler.yield_subset(pr.reduction)                   # :507
    b = X.shape[0]
    st.n_iter += b
    st.sample_n_iter[idx] += 1
    w_sample = np.power(st.sample_n_iter[idx].astype(np.float64), -pr.sample_learning_rate).astype(dt)
    w = batch_weight(st.n_iter, b, pr.learning_rate, 0)
    compute_code(st, pr, X, idx, w_sample, subset)
    code = st.code[idx]
    update_stats(st, pr, X, code, w)
    order = update_dict(st, pr, subset, w, order)
    if st.trace is not None:
        st.trace.append(dict(subset=subset.copy(), w=w, order=np.asarray(order).copy(),
                             code=code.copy()))
    return subset, order, w


def partial_fit(st: SomfState, pr: SomfParams, X, sample_indices=None):
    """dict_fact.py:313-337"""
    X = np.ascontiguousarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    for batch in _gen_batches(X.shape[0], pr.batch_size):
        minibatch_step(st, pr, X[batch], sub_indices(sample_indices, batch))
    return st


def shuffle(st: SomfState, pr: SomfParams):
    """dict_fact.py:359-379"""
    seed = st.rng.randint(MAX_INT)
    rs = OracleRandomState(seed)
    arrays = [st.code]
    if pr.G_agg == 'average':
        arrays.append(st.G_average.reshape(st.G_average.shape[0], -1))
    if pr.Dx_agg == 'average':
        arrays.append(st.Dx_average)
    perm = rs.shuffle_with_trace(arrays)
    st.labels = st.labels[perm]
    return perm


def fit(pr: SomfParams, X, trace=False, sweeps=False) -> SomfState:
    """dict_fact.py:286-311"""
    X = np.ascontiguousarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    init = X if pr.dict_init is None else np.asarray(pr.dict_init, dtype=X.dtype)
    st = prepare(pr, n_samples=X.shape[0], X=init)
    if trace:
        st.trace = []
    if sweeps:
        st.sweeps = []
    for _ in range(pr.n_epochs):
        partial_fit(st, pr, X)
        perm = shuffle(st, pr)
        X = X[perm]
    return st


def transform(pr: SomfParams, D, X, G=None):
    """dict_fact.py:47-92"""
    X = np.ascontiguousarray(X, dtype=D.dtype)
    if G is None:
        G = D.dot(D.T)
    Dx = X.dot(D.T)
    code = np.ones((X.shape[0], D.shape[0]), dtype=D.dtype)
    enet_regression_single_gram(G, Dx, X, code, np.arange(X.shape[0]),
                                pr.code_l1_ratio, pr.code_alpha, pr.code_pos, pr.tol, pr.max_iter)
    return code


def score(pr: SomfParams, D, X, G=None):
    """dict_fact.py:94-114"""
    code = transform(pr, D, X, G)
    loss = np.sum((X - code.dot(D)) ** 2) / 2
    regul = pr.code_alpha * (np.sum(np.abs(code)) * pr.code_l1_ratio
                             + (1 - pr.code_l1_ratio) * np.sum(code ** 2) / 2)
    return (loss + regul) / X.shape[0]
