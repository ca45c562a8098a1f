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
