"""GPU parity of the fine-grained C-ABI entry points (the ones that are 1:1 with
the reference's Cython functions) against the golden vectors recorded from the
reference and against the CPU oracle."""
import numpy as np
import pytest

from .conftest import load_golden, rel_fro, assert_within_f32_noise

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-10}


@pytest.fixture(scope='module')
def fast():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    from modl_amd import dict_fact_fast
    return dict_fact_fast


def test_cd_and_ridge_golden(fast):
    g = load_golden('cd')
    bad = []
    for i, row in enumerate(g['meta']):
        dti, k, p, b, n, l1, alpha, pos, tol, mi = row[:10]
        dt = np.dtype(np.float32 if dti == 0 else np.float64)
        tag = '%s_%d' % ('f32' if dti == 0 else 'f64', int(k))
        G, Gm, Dx0, X = g['G_' + tag], g['Gm_' + tag], g['Dx_' + tag], g['X_' + tag]
        idx = g['idx_%d' % i]
        code = g['code_in_%d' % i].copy()
        fast._enet_regression_single_gram(G, Dx0.copy(), X, code, idx, l1, alpha, bool(pos), tol, int(mi))
        e1 = rel_fro(code, g['single_code_%d' % i])
        code = g['code_in_%d' % i].copy()
        fast._enet_regression_multi_gram(Gm.copy(), Dx0.copy(), X, code, idx, l1, alpha, bool(pos), tol, int(mi))
        e2 = rel_fro(code, g['multi_code_%d' % i])
        if not (e1 < TOL[dt] and e2 < TOL[dt]):
            bad.append((i, tuple(row[:10]), e1, e2))
    assert not bad, bad


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k,b,p', [(256, 64, 300), (70, 20, 150), (300, 9, 400), (5, 3, 7), (1024, 6, 1100)])
def test_cd_vs_oracle_sweeps_and_codes(fast, oracle, dt, k, b, p):
    """Same number of sweeps per sample as the CPU restatement, codes within tolerance."""
    rs = np.random.RandomState(k + b)
    D = rs.randn(k, p).astype(dt)
    D /= np.sqrt((D ** 2).sum(1))[:, None]
    X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < 0.1)).dot(D) + 0.1 * rs.randn(b, p)).astype(dt))
    G = np.ascontiguousarray(D.dot(D.T).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray(X.dot(D.T).astype(dt))
    n = b + 3
    idx = rs.permutation(n)[:b].astype(np.int64)
    for l1, alpha, pos in ((1.0, 0.3, False), (0.7, 0.1, True)):
        c_gpu = np.ones((n, k), dtype=dt)
        c_cpu = np.ones((n, k), dtype=dt)
        sw_gpu = np.zeros(b, dtype=np.int32)
        sw_cpu = np.zeros(b, dtype=np.int32)
        fast._enet_regression_single_gram(G, Dx.copy(), X, c_gpu, idx, l1, alpha, pos, 1e-2, 100, sweeps=sw_gpu)
        oracle.enet_regression_single_gram(G, Dx.copy(), X, c_cpu, idx, l1, alpha, pos, 1e-2, 100, sweeps=sw_cpu)
        assert rel_fro(c_gpu, c_cpu) < TOL[np.dtype(dt)], (l1, alpha, pos, sw_gpu, sw_cpu)
        if dt == np.float64:
            np.testing.assert_array_equal(sw_gpu, sw_cpu)
        untouched = np.setdiff1d(np.arange(n), idx)
        assert np.all(c_gpu[untouched] == 1)


_SPARSE_CASES = [(256, 40, 300, 1.5), (256, 40, 12, 0.05), (128, 17, 200, 2.0), (512, 9, 600, 2.5), (1024, 5, 1100, 3.0),
                 (100, 11, 130, 1.5)]


def _sparse_regime_check(fast, oracle, dt, k, b, p, alpha):
    """Sparse codes (few active coordinates per sweep: the active-set sweeps of csrc/cd_solver.hip), a rank-deficient
    Gram matrix among them (p << k: the active set keeps changing and the solver runs into max_iter)."""
    rs = np.random.RandomState(k + b + p)
    D = rs.randn(k, p).astype(dt)
    D /= np.sqrt((D ** 2).sum(1))[:, None]
    X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < 0.05)).dot(D) + 0.05 * rs.randn(b, p)).astype(dt))
    G = np.ascontiguousarray(D.dot(D.T).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray(X.dot(D.T).astype(dt))
    idx = np.arange(b, dtype=np.int64)
    for pos in (False, True):
        c_gpu, c_cpu = np.ones((b, k), dtype=dt), np.ones((b, k), dtype=dt)
        sw_gpu, sw_cpu = np.zeros(b, dtype=np.int32), np.zeros(b, dtype=np.int32)
        fast._enet_regression_single_gram(G, Dx.copy(), X, c_gpu, idx, 1.0, alpha, pos, 1e-3, 60, sweeps=sw_gpu)
        oracle.enet_regression_single_gram(G, Dx.copy(), X, c_cpu, idx, 1.0, alpha, pos, 1e-3, 60, sweeps=sw_cpu)
        assert (c_cpu != 0).mean() < (0.2 if p >= k else 0.5), 'not a sparse case'
        if dt == np.float64:
            np.testing.assert_array_equal(sw_gpu, sw_cpu)
            np.testing.assert_array_equal(c_gpu != 0, c_cpu != 0)
            assert rel_fro(c_gpu, c_cpu) < 1e-10, (k, p, alpha, pos)
        else:
            same = sw_gpu == sw_cpu
            # f32 on a singular Gram matrix: the sweeps never settle and rounding differences (H = Q w is formed in
            # another order) are amplified over the 60 sweeps; the f64 run above is the exactness check
            tol = 2e-5 if p >= k else 2e-3
            assert same.mean() >= 0.9 and rel_fro(c_gpu[same], c_cpu[same]) < tol, (k, p, alpha, pos, sw_gpu, sw_cpu)


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k,b,p,alpha', _SPARSE_CASES)
def test_cd_sparse_regime_vs_oracle(fast, oracle, dt, k, b, p, alpha):
    _sparse_regime_check(fast, oracle, dt, k, b, p, alpha)


@pytest.mark.parametrize('pct', [100, 0])
def test_cd_forced_sweep_kind(fast, oracle, pct):
    """The same checks with every sweep forced through the active-set path (100) or the dense path (0) -
    modl_debug_set(MODL_DEBUG_CD_SPARSE_PCT): dense and sparse sweeps produce the same iterates."""
    from modl_amd._lib import lib, check, DEBUG_CD_SPARSE_PCT
    check(lib.modl_debug_set(DEBUG_CD_SPARSE_PCT, pct))
    try:
        for dt in (np.float32, np.float64):
            for case in _SPARSE_CASES:
                _sparse_regime_check(fast, oracle, dt, *case)
        test_cd_and_ridge_golden(fast)
    finally:
        check(lib.modl_debug_set(DEBUG_CD_SPARSE_PCT, -1))


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k', [50, 70, 161, 200, 256, 330, 511])
def test_ridge_vs_oracle(fast, oracle, dt, k):
    rs = np.random.RandomState(k)
    b, p = 20, max(400, k + 100)        # full-rank Gram: the f32 bound below is cond(G + aI) * eps
    D = rs.randn(k, p).astype(dt)
    X = np.ascontiguousarray(rs.randn(b, p).astype(dt))
    G = np.ascontiguousarray((D.dot(D.T) / p).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray((X.dot(D.T) / p).astype(dt))
    Gm = np.ascontiguousarray(np.stack([G * (1 + 0.05 * j) for j in range(b)]).astype(dt))
    idx = np.arange(b, dtype=np.int64)[::-1].copy()
    d8 = np.float64
    for alpha in (0.1, 1e-3):
        for multi in (False, True):
            gpu_f = fast._enet_regression_multi_gram if multi else fast._enet_regression_single_gram
            cpu_f = oracle.enet_regression_multi_gram if multi else oracle.enet_regression_single_gram
            Gx = Gm if multi else G
            c1, c2 = np.ones((b, k), dtype=dt), np.ones((b, k), dtype=dt)
            gpu_f(Gx.copy(), Dx.copy(), X, c1, idx, 0.0, alpha, False, 1e-2, 100)
            cpu_f(Gx.copy(), Dx.copy(), X, c2, idx, 0.0, alpha, False, 1e-2, 100)
            if dt == np.float64:
                assert rel_fro(c1, c2) < 1e-9, (k, alpha, multi)
            else:
                # f32: the error of a solve is cond(G + aI) * eps whoever performs it - the yardstick is LAPACK's own
                # f32 solve (the oracle) against its f64 solve of the same float32 system
                c3 = np.ones((b, k), dtype=d8)
                cpu_f(Gx.astype(d8), Dx.astype(d8), X.astype(d8), c3, idx, 0.0, alpha, False, 1e-2, 100)
                assert_within_f32_noise(c1, c2, c3, (k, alpha, multi))


def test_update_G_average_golden(fast):
    g = load_golden('cd')
    for tag in ('f32_16', 'f32_64', 'f64_16', 'f64_64'):
        Ga = g['Gm_' + tag].copy()
        fast._update_G_average(Ga, g['G_' + tag], g['Gavg_w_' + tag])
        np.testing.assert_allclose(Ga, g['Gavg_out_' + tag], rtol=2e-6 if 'f32' in tag else 1e-14)


def test_enet_golden():
    from modl_amd.enet import enet_norm, enet_projection, enet_scale
    g = load_golden('enet')
    bad = []
    for i, (dti, n, l1, radius, nrm, sc_radius) in enumerate(g['meta']):
        v = g['v_%d' % i]
        f32 = dti == 0
        got = enet_norm(v, l1)
        if abs(got - nrm) > (1e-5 if f32 else 1e-12) * max(1.0, abs(nrm)):
            bad.append(('norm', i, got, nrm))
        out = np.zeros_like(v)
        enet_projection(v, out, radius, l1)
        ref = g['proj_%d' % i]
        if not np.allclose(out, ref, rtol=2e-5 if f32 else 1e-10, atol=2e-6 if f32 else 1e-12):
            bad.append(('proj', i, n, l1, radius, float(np.abs(out - ref).max())))
        vs = v.copy()
        enet_scale(vs, l1, sc_radius)
        if not np.allclose(vs, g['scaled_%d' % i], rtol=1e-5 if f32 else 1e-11):
            bad.append(('scale', i))
    assert not bad, bad[:10]


def test_enet_reference_properties():
    """modl/utils/math/tests/test_enet.py:99-156 (norm of a projection = radius, l2/l1 balls, scale)."""
    from modl_amd.enet import enet_norm, enet_projection, enet_scale
    rs = np.random.RandomState(0)
    norms = np.zeros(10)
    for i in range(10):
        a = rs.randn(20000)
        a /= np.sqrt(np.sum(a ** 2))
        c = np.zeros(20000)
        enet_projection(a, c, 1, 0.15)
        norms[i] = enet_norm(c, l1_ratio=0.15)
    np.testing.assert_array_almost_equal(norms, np.ones(10))
    for i in range(10):
        a = rs.randn(100)
        c = np.zeros(100)
        enet_projection(a, c, 2, 0.0)
        assert abs(np.sqrt(np.sum(c ** 2)) - np.sqrt(2)) < 1e-6
        b = np.zeros(100)
        enet_projection(a, b, 1, 1.0)
        assert abs(np.sum(np.abs(b)) - 1) < 1e-6
    a = rs.randn(100)
    for r in (1., 2.):
        for l1_ratio in (0., 0.5, 1.):
            enet_scale(a, l1_ratio, r)
            assert abs(enet_norm(a, l1_ratio) - r) < 1e-6
    # batched rows
    A = rs.randn(7, 333)
    out = np.zeros_like(A)
    enet_projection(A, out, 0.7, 0.3)
    np.testing.assert_allclose(enet_norm(out, 0.3), 0.7, rtol=1e-9)


def test_predict_csr_matches_dense(oracle):
    import ctypes as C
    import scipy.sparse as sp
    import torch
    from modl_amd._lib import lib, check
    rs = np.random.RandomState(0)
    P, Q = rs.randn(30, 6), rs.randn(6, 17)
    mask = rs.rand(30, 17) < 0.3
    X = sp.csr_matrix(mask.astype(np.float64))
    dev = torch.device('cuda')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data, ind, ptr_, dP, dQ = t(np.zeros(X.nnz)), t(X.indices.astype(np.int32)), t(X.indptr.astype(np.int32)), t(P), t(Q)
    check(lib.modl_predict_csr(C.c_void_p(data.data_ptr()), C.c_void_p(ind.data_ptr()), C.c_void_p(ptr_.data_ptr()),
                               C.c_void_p(dP.data_ptr()), 30, 6, C.c_void_p(dQ.data_ptr()), 17, None))
    exp = np.zeros(X.nnz)
    oracle.predict_csr(exp, X.indices, X.indptr, P, Q)
    np.testing.assert_allclose(data.cpu().numpy(), exp, rtol=1e-13)
    np.testing.assert_allclose(exp, P.dot(Q)[mask], rtol=1e-13)


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k', [520, 1024])
def test_ridge_wide_systems_vs_oracle(fast, oracle, dt, k):
    """k > 512 (the reference's HCP experiment runs fMRIDictFact - ridge codes - at n_components = 1024,
    exps/hcp/decompose_hcp.py:50-60): blocked Cholesky + blocked substitutions on the matrix cores, shared Gram and
    one Gram per sample, against LAPACK posv through the oracle."""
    rs = np.random.RandomState(k)
    b, p = 7, k + 200
    D = rs.randn(k, p).astype(dt)
    X = np.ascontiguousarray(rs.randn(b, p).astype(dt))
    G = np.ascontiguousarray((D.dot(D.T) / p).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray((X.dot(D.T) / p).astype(dt))
    idx = np.arange(b, dtype=np.int64)[::-1].copy()
    alpha = 0.1
    c1, c2 = np.ones((b + 2, k), dtype=dt), np.ones((b + 2, k), dtype=dt)
    fast._enet_regression_single_gram(G, Dx.copy(), X, c1, idx, 0.0, alpha, False, 1e-2, 100)
    oracle.enet_regression_single_gram(G, Dx.copy(), X, c2, idx, 0.0, alpha, False, 1e-2, 100)
    if dt == np.float64:
        assert rel_fro(c1, c2) < 1e-9, (k, rel_fro(c1, c2))
    else:                                                   # f32: LAPACK's own f32 noise on this system is the yardstick
        c3 = np.ones((b + 2, k), dtype=np.float64)
        oracle.enet_regression_single_gram(G.astype(np.float64), Dx.astype(np.float64), X.astype(np.float64), c3, idx, 0.0,
                                           alpha, False, 1e-2, 100)
        assert_within_f32_noise(c1, c2, c3, k)
    assert np.all(c1[b:] == 1)
    Gm = np.ascontiguousarray(np.stack([G * (1 + 0.05 * j) for j in range(3)]).astype(dt))
    c1, c2 = np.ones((3, k), dtype=dt), np.ones((3, k), dtype=dt)
    i3 = np.arange(3, dtype=np.int64)
    fast._enet_regression_multi_gram(Gm.copy(), Dx[:3].copy(), X[:3], c1, i3, 0.0, alpha, False, 1e-2, 100)
    oracle.enet_regression_multi_gram(Gm.copy(), Dx[:3].copy(), X[:3], c2, i3, 0.0, alpha, False, 1e-2, 100)
    if dt == np.float64:
        assert rel_fro(c1, c2) < 1e-9, (k, rel_fro(c1, c2))
    else:
        c3 = np.ones((3, k), dtype=np.float64)
        oracle.enet_regression_multi_gram(Gm.astype(np.float64), Dx[:3].astype(np.float64), X[:3].astype(np.float64), c3, i3,
                                          0.0, alpha, False, 1e-2, 100)
        assert_within_f32_noise(c1, c2, c3, k)


def _assert_solvers_agree(dt, got, ref, what):
    """Two roundings of the same sweep (the four-wavefront solver applies w_new - w_old with ONE fused multiply-add where
    the one-wavefront kernel - the reference's operation order - uses two): f64 to 1e-10 with identical sweep counts;
    f32 identical sweep counts on >= 95 % of the samples and 2e-5 on those."""
    (code_a, sw_a), (code_b, sw_b) = got, ref
    if dt == np.float64:
        np.testing.assert_array_equal(sw_a, sw_b)
        err = np.linalg.norm(code_a - code_b) / np.linalg.norm(code_b)
        assert err < 1e-10, (what, err)
    else:
        same = sw_a == sw_b
        assert same.mean() >= 0.95, (what, same.mean())
        return same


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k,b,p,alpha', [(256, 48, 600, 0.3), (200, 33, 400, 0.2), (128, 40, 300, 0.3), (100, 17, 200, 0.3),
                                         (512, 12, 700, 0.3), (330, 9, 600, 0.3), (256, 24, 12, 0.05), (250, 30, 64, 0.1),
                                         (1024, 6, 1100, 0.3), (600, 5, 800, 0.3)])
def test_cd_two_solvers_agree(fast, dt, k, b, p, alpha):
    """The four-wavefront solver (csrc/cd_split_impl.hpp: chain / update waves / tile loader, shared Gram, 32 <= k <= 1024)
    against the one-wavefront solver (csrc/cd_solver.hip, the reference's operation order; modl_debug_set(
    MODL_DEBUG_CD_SPLIT, 0) forces it): same sweep order, skip rule, step formula and stopping tests, the k-wide update
    rounded once instead of twice - full and padded k, both stopping rules, positivity, a singular Gram matrix running
    into max_iter.  (Every other test of this file runs whichever the library picks, i.e. the four-wavefront one where
    it applies, against the oracle.)"""
    from modl_amd import _lib as L
    from modl_amd._lib import check, DEBUG_CD_SPLIT, load_diag
    # the four-wavefront solver under test is the one that SHIPS (the product library, compiled without
    # MODL_DIAG: other register allocation, no stamp code); the one-wavefront kernel comes from the product library where
    # it exists there (k <= 256) and from the diagnostics build beyond
    libs = {1: L.lib, 0: L.lib if k <= 256 else load_diag()}
    rs = np.random.RandomState(k + b)
    D = rs.randn(k, p).astype(dt)
    D /= np.sqrt((D ** 2).sum(1))[:, None]
    X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < 0.1)).dot(D) + 0.1 * rs.randn(b, p)).astype(dt))
    G = np.ascontiguousarray(D.dot(D.T).astype(dt))
    G = (G + G.T) / 2
    Dx = np.ascontiguousarray(X.dot(D.T).astype(dt))
    idx = rs.permutation(b + 5)[:b].astype(np.int64)
    for l1, pos, mi in ((1.0, False, 100), (0.7, True, 30)):
        out = {}
        try:
            for split in (0, 1):
                lib = libs[split]
                check(lib.modl_debug_set(DEBUG_CD_SPLIT, split))
                code = np.ones((b + 5, k), dtype=dt)
                sw = np.zeros(b, dtype=np.int32)
                fast._enet_regression_single_gram(G, Dx.copy(), X, code, idx, l1, alpha, pos, 1e-2, mi, sweeps=sw, _lib=lib)
                check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
                out[split] = (code[idx], sw)
        finally:
            for lib in libs.values():
                check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
        same = _assert_solvers_agree(dt, out[1], out[0], (k, l1))
        if dt == np.float32:
            a_, b_ = out[1][0][same].astype(np.float64), out[0][0][same].astype(np.float64)
            # (a rank-deficient Gram matrix - p < k - has no unique minimiser: codes then agree through the fit D^T w)
            if p < k:
                a_, b_ = a_.dot(D.astype(np.float64)), b_.dot(D.astype(np.float64))
            err = np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1e-30)
            assert err < (1e-4 if p < k else 2e-5), (k, l1, err)
        assert out[0][1].max() > 1


@pytest.mark.parametrize('k', [512, 1024])
def test_cd_wide_misaligned_gram_and_split_switch(fast, k):
    """beyond 256 coefficients the product library only has the four-wavefront solver, which wants
    16-byte aligned rows.  A caller's k = 512 / 1024 Gram matrix that is NOT aligned (a view one element into a buffer)
    used to come back as MODL_EINVAL; it is copied into the aligned scratch now - same bits as the aligned call - and
    modl_debug_set(MODL_DEBUG_CD_SPLIT, 0) still selects a correct path for k > 256 (shared and per-sample Gram)."""
    import torch
    from modl_amd import _lib as L
    from modl_amd._lib import check, DEBUG_CD_SPLIT
    rs = np.random.RandomState(k)
    b, p = 5, k + 100
    D = rs.randn(k, p).astype(np.float32)
    D /= np.sqrt((D ** 2).sum(1))[:, None]
    X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < 0.1)).dot(D) + 0.1 * rs.randn(b, p)).astype(np.float32))
    G = D.dot(D.T).astype(np.float32)
    G = np.ascontiguousarray((G + G.T) / 2)
    Dx = np.ascontiguousarray(X.dot(D.T).astype(np.float32))
    idx = np.arange(b, dtype=np.int64)
    dev = torch.device('cuda', 0)

    def solve(Gdev):
        code = torch.ones((b, k), dtype=torch.float32, device=dev)
        fast._enet_regression_single_gram(Gdev, torch.from_numpy(Dx).to(dev), torch.from_numpy(X).to(dev), code, idx,
                                          1.0, 0.3, False, 1e-2, 100)
        return code.cpu().numpy()
    aligned = torch.from_numpy(G).to(dev)
    buf = torch.empty(k * k + 1, dtype=torch.float32, device=dev)
    shifted = buf[1:].view(k, k)
    shifted.copy_(aligned)
    assert aligned.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    want = solve(aligned)
    np.testing.assert_array_equal(solve(shifted), want)
    try:                                                       # the diagnostics switch selects between correct paths
        check(L.lib.modl_debug_set(DEBUG_CD_SPLIT, 0))
        np.testing.assert_array_equal(solve(aligned), want)
        Gm = np.ascontiguousarray(np.broadcast_to(G, (b, k, k)))
        code = np.ones((b, k), dtype=np.float32)
        fast._enet_regression_multi_gram(Gm, Dx.copy(), X, code, idx, 1.0, 0.3, False, 1e-2, 100)
        assert rel_fro(code, want) < 2e-5            # (H0 = Qw formed in another order than the shared-matrix route)
    finally:
        check(L.lib.modl_debug_set(DEBUG_CD_SPLIT, 1))


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('k,b', [(128, 9), (256, 5), (70, 9), (200, 7), (600, 3)])
def test_cd_two_solvers_agree_per_sample_gram(fast, dt, k, b):
    """A Gram matrix per sample (G_agg = 'average', dict_fact_fast.pyx:33-113) on the four-wavefront solver - k one of
    its strides: solved from where the matrices are stored; any other k: through zero-padded copies of a slice of the
    minibatch (launch_cd_per_sample) - against the one-wavefront kernel."""
    from modl_amd import _lib as L
    from modl_amd._lib import check, DEBUG_CD_SPLIT, load_diag
    libs = {1: L.lib, 0: L.lib if k <= 256 else load_diag()}       # (as in test_cd_two_solvers_agree)
    rs = np.random.RandomState(k + b)
    p = 2 * k
    Gm = np.empty((b, k, k), dtype=dt)
    Dx = np.empty((b, k), dtype=dt)
    X = np.empty((b, p), dtype=dt)
    for i in range(b):
        D = rs.randn(k, p).astype(dt)
        D /= np.sqrt((D ** 2).sum(1))[:, None]
        X[i] = ((rs.randn(k) * (rs.rand(k) < 0.1)).dot(D) + 0.1 * rs.randn(p)).astype(dt)
        G = D.dot(D.T).astype(dt)
        Gm[i] = (G + G.T) / 2
        Dx[i] = X[i].dot(D.T)
    idx = np.arange(b, dtype=np.int64)
    out = {}
    try:
        for split in (0, 1):
            lib = libs[split]
            check(lib.modl_debug_set(DEBUG_CD_SPLIT, split))
            code = np.ones((b, k), dtype=dt)
            sw = np.zeros(b, dtype=np.int32)
            fast._enet_regression_multi_gram(Gm.copy(), Dx.copy(), X, code, idx, 0.9, 0.3, False, 1e-2, 100, sweeps=sw, _lib=lib)
            check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
            out[split] = (code, sw)
    finally:
        for lib in libs.values():
            check(lib.modl_debug_set(DEBUG_CD_SPLIT, 1))
    same = _assert_solvers_agree(dt, out[1], out[0], k)
    if dt == np.float32:
        err = np.linalg.norm(out[1][0][same] - out[0][0][same]) / np.linalg.norm(out[0][0][same])
        assert err < 2e-5, (k, err)
    assert out[0][1].max() > 1
