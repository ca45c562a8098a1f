"""RecsysDictFact masked path: the oracle restatement against
the golden recorded from the reference (CPU), and the GPU estimator against the
golden and the reference's own functional tests (modl/decomposition/tests/test_recsys.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from .conftest import load_golden, rel_fro


def _toy():
    rs = np.random.RandomState(0)
    n, p, k = 40, 25, 4
    full = rs.randn(n, k).dot(rs.randn(k, p))
    mask = rs.rand(n, p) < 0.4
    return sp.csr_matrix(np.where(mask, full, 0.0)), k


KW = dict(alpha=0.1, beta=0.5, batch_size=5, n_epochs=2, learning_rate=0.9, random_state=0)


@pytest.mark.parametrize('detrend', [False, True])
def test_oracle_recsys_golden(detrend):
    from oracle import wrappers_oracle as wo
    g = load_golden('recsys')
    X, k = _toy()
    np.testing.assert_array_equal(X.indices, g['X_indices'])
    fit = wo.recsys_fit(X, n_components=k, detrend=detrend, **KW)
    tag = 'det%d' % int(detrend)
    assert rel_fro(fit['D'], g['D_' + tag]) < 1e-9
    assert rel_fro(fit['code'], g['code_' + tag]) < 1e-9
    assert rel_fro(fit['C'], g['C_' + tag]) < 1e-9
    assert rel_fro(fit['B'], g['B_' + tag]) < 1e-9
    assert rel_fro(wo.recsys_predict(fit, X, detrend), g['pred_' + tag]) < 1e-9
    if detrend:
        np.testing.assert_allclose(fit['row_mean'], g['row_mean'], rtol=1e-12)
        np.testing.assert_allclose(fit['col_mean'], g['col_mean'], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('detrend', [False, True])
def test_gpu_recsys_golden(detrend):
    from modl_amd.recsys import RecsysDictFact
    g = load_golden('recsys')
    X, k = _toy()
    est = RecsysDictFact(n_components=k, detrend=detrend, **KW)
    est.fit(X)
    tag = 'det%d' % int(detrend)
    assert rel_fro(est.components_, g['D_' + tag]) < 1e-9
    assert rel_fro(est.code_, g['code_' + tag]) < 1e-9
    assert rel_fro(est.C_, g['C_' + tag]) < 1e-9
    assert rel_fro(est.B_, g['B_' + tag]) < 1e-9
    assert rel_fro(est.predict(X).data, g['pred_' + tag]) < 1e-9
    assert abs(est.score(X) - g['score_' + tag]) < 1e-9


@pytest.mark.gpu
def test_gpu_recsys_f32_vs_oracle():
    from modl_amd.recsys import RecsysDictFact
    from oracle import wrappers_oracle as wo
    X, k = _toy()
    X = X.astype(np.float32)
    est = RecsysDictFact(n_components=k, **KW).fit(X)
    fit = wo.recsys_fit(X, n_components=k, **KW)
    assert est.components_.dtype == np.float32
    # f32: held to the reference algorithm's own f32 noise (the oracle in f32 against the oracle in f64 on the same
    # float32 ratings), not to a flat tolerance
    fit64 = wo.recsys_fit(X.astype(np.float64), n_components=k, **KW)
    from .conftest import assert_within_f32_noise
    assert_within_f32_noise(est.components_, fit['D'], fit64['D'], 'dictionary')
    assert_within_f32_noise(est.code_, fit['code'], fit64['code'], 'codes')


@pytest.mark.gpu
@pytest.mark.parametrize('detrend', [False, True])
def test_ref_dict_completion(detrend):                       # test_recsys.py:12-60
    from modl_amd.recsys import RecsysDictFact
    rng = np.random.RandomState(0)
    X = np.dot(rng.rand(50, 3), rng.rand(3, 20))
    mf = RecsysDictFact(n_components=3, n_epochs=1, alpha=1e-3, random_state=0, detrend=detrend, verbose=0)
    mf.fit(X)
    Y = np.dot(mf.code_, mf.components_)
    if detrend:
        Y += mf.col_mean_[np.newaxis, :]
        Y += mf.row_mean_[:, np.newaxis]
    np.testing.assert_array_almost_equal(Y, mf.predict(X).toarray())
    np.testing.assert_almost_equal(np.sqrt(np.mean((X - Y) ** 2)), mf.score(X))


@pytest.mark.gpu
def test_ref_dict_completion_missing():                       # test_recsys.py:63-86
    from modl_amd.recsys import RecsysDictFact
    rng = np.random.RandomState(0)
    X = np.dot(rng.rand(100, 4), rng.rand(4, 20))
    mask = rng.rand(100, 20) < 0.7
    Xtr = sp.csr_matrix(np.where(mask, X, 0))
    Xte = sp.csr_matrix(np.where(~mask, X, 0))
    mf = RecsysDictFact(n_components=4, n_epochs=10, alpha=1, random_state=0, batch_size=1, detrend=True,
                        learning_rate=0.9)
    mf.fit(Xtr)
    pred = mf.predict(Xte)
    rmse = np.sqrt(np.mean((Xte.data - pred.data) ** 2))
    base = np.sqrt(np.mean((Xte.data - (np.repeat(mf.row_mean_, np.diff(Xte.indptr))
                                        + mf.col_mean_.take(Xte.indices))) ** 2))
    assert rmse < base


def _ragged_ratings(n, p, k, heavy, empty, seed, dtype):
    """CSR ratings with a few HEAVY rows (several chunks of 128 ratings in the one-launch kernel), rows WITHOUT ratings
    (recsys.py:170: they keep their code) and a low-rank signal"""
    rs = np.random.RandomState(seed)
    full = rs.randn(n, k).dot(rs.randn(k, p)) + 0.1 * rs.randn(n, p)
    dens = np.full(n, 0.04)
    dens[rs.choice(n, heavy, replace=False)] = 0.9
    dens[rs.choice(n, empty, replace=False)] = 0.0
    mask = rs.rand(n, p) < dens[:, None]
    return sp.csr_matrix(np.where(mask, full, 0.0).astype(dtype))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,k,p,b', [(np.float64, 50, 700, 10), (np.float64, 20, 300, 7), (np.float32, 64, 500, 16),
                                         (np.float32, 30, 1500, 64), (np.float64, 50, 2500, 10)])
def test_gpu_recsys_one_launch_vs_separate_launches(dtype, k, p, b):
    """Round 6: a masked minibatch with at most 64 (f64: 56) atoms and 64 rows runs its codes and C_ as ONE launch
    (csrc/recsys.hip: recsys_fused_kernel: rating chunks of 128 with a ticketed fixed-order sum per row, Cholesky codes, the last
    workgroup C_), B_ and the blocked dictionary update behind it (the default), or - MODL_DEBUG_RECSYS_FUSED = 2 / 4 - B_ and the
    dictionary sweep inside that launch with one item per thread, on one workgroup or on up to four that exchange every atom's
    sums through memory.  All of them against the separate launches of rounds 2-5 (= 0; pinned to the reference golden above) on
    ratings with heavy rows (several chunks), rows without ratings and - p = 2500 - minibatches that touch more items than the
    sweep's workgroups hold: f64 to 1e-9, f32 within the reference algorithm's own f32 noise; and against the oracle."""
    from modl_amd.recsys import RecsysDictFact
    from modl_amd._lib import lib, check, DEBUG_RECSYS_FUSED
    from oracle import wrappers_oracle as wo
    from .conftest import assert_within_f32_noise
    X = _ragged_ratings(160, p, 6, heavy=5, empty=6, seed=p + k, dtype=dtype)
    kw = dict(alpha=0.2, beta=0.0, batch_size=b, n_epochs=1, learning_rate=0.9, random_state=1)
    out = {}
    for fused in (1, 2, 4, 0):                               # the default, the in-kernel sweep (one workgroup / up to four), the old launches
        check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, fused))
        try:
            est = RecsysDictFact(n_components=k, **kw).fit(X)
            out[fused] = dict(D=est.components_, code=est.code_, C=est.C_, B=est.B_, cn=est.comp_norm_, fn=est.feature_n_iter_)
            counts = est._dev.launch_counts()
        finally:
            check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, 1))
        if fused:
            assert counts[0] > 0, counts                              # (a minibatch of more than 64 chunks takes the separate launches)
        else:
            assert counts[0] == 0
    fit64 = wo.recsys_fit(X.astype(np.float64), n_components=k, **kw)
    fit32 = wo.recsys_fit(X, n_components=k, **kw) if dtype == np.float32 else None
    for fused in (1, 2, 4):
        assert np.array_equal(out[fused]['fn'], out[0]['fn'])
        if dtype == np.float64:
            for key in ('D', 'code', 'C', 'B'):
                assert rel_fro(out[fused][key], out[0][key]) < 1e-9, (fused, key)
            assert np.allclose(out[fused]['cn'], out[0]['cn'], rtol=0, atol=1e-10)      # (budgets left: ~0, differences of O(1) numbers)
            for key in ('D', 'code', 'C', 'B'):
                assert rel_fro(out[fused][key], fit64[key]) < 1e-8, (fused, key)
        else:
            for key in ('D', 'code', 'C', 'B'):
                assert_within_f32_noise(out[fused][key], fit32[key], fit64[key], (fused, key))
        assert np.all(np.isfinite(out[fused]['D']))


@pytest.mark.gpu
def test_gpu_recsys_fit_batches_equals_python_loop():
    """The epoch's host loop behind the ABI (modl_recsys_fit_batches_*: minibatch weights, numpy's legacy permutation(k) drawn by a
    generator loaded with numpy's state) == the Python loop of _single_batch_fit (taken when a callback is set), bit for bit,
    with both generators left in step."""
    from modl_amd.recsys import RecsysDictFact
    X = _ragged_ratings(123, 400, 5, heavy=3, empty=4, seed=9, dtype=np.float64)
    kw = dict(n_components=12, alpha=0.3, batch_size=10, n_epochs=2, learning_rate=0.9, random_state=3)
    a = RecsysDictFact(**kw).fit(X)
    b = RecsysDictFact(callback=lambda est: None, **kw).fit(X)
    for name in ('components_', 'code_', 'C_', 'B_', 'comp_norm_', 'feature_n_iter_'):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.n_iter_ == b.n_iter_ == 2 * 123
    assert a.random_state.randint(1 << 30) == b.random_state.randint(1 << 30)


def test_shuffle_split_partitions_the_entries():
    """modl/utils/recsys/cross_validation.py:8-42: a partition of the stored entries, reproducible for a seed"""
    import scipy.sparse as sp
    from modl_amd.utils.recsys import ShuffleSplit, train_test_split
    rs = np.random.RandomState(0)
    X = sp.random(30, 20, density=0.3, random_state=rs, format='csr')
    tr, te = train_test_split(X, train_size=0.75, random_state=3)
    assert tr.shape == te.shape == X.shape and tr.nnz == int(0.75 * X.nnz) and tr.nnz + te.nnz == X.nnz
    assert abs((tr + te) - X).sum() == 0 and tr.multiply(te).nnz == 0
    ind = np.random.RandomState(3).permutation(X.nnz)                      # the reference's draw
    coo = X.tocoo()
    assert np.array_equal(tr.row, coo.row[ind[:tr.nnz]]) and np.array_equal(tr.data, coo.data[ind[:tr.nnz]])
    splits = list(ShuffleSplit(n_iter=3, train_size=0.5, random_state=1).split(X))
    assert len(splits) == 3 and (splits[0][0] != splits[1][0]).nnz > 0
