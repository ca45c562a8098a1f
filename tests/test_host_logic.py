"""CPU tests of the estimator's HOST logic (no GPU): modl_amd.DictFact driven
with the oracle standing in for the device kernels (tests/oracle_backend.py).
Covers: the minibatch loop and its RNG draws against the reference's golden
trajectories, and the 2-rank data-parallel protocol over gloo — an R-rank run
with local batch b must equal a 1-rank run with batch R*b on the concatenated
rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from numpy.testing import assert_array_equal

from .conftest import load_golden, rel_fro
from .test_oracle_golden import small_case_params, _cases


def _host_estimator():
    from modl_amd.dict_fact import DictFact
    from .oracle_backend import OracleBackend

    class HostDictFact(DictFact):
        def _make_backend(self):
            return OracleBackend()
    return HostDictFact


@pytest.mark.parametrize('name', [c for c in _cases() if c.endswith('f64')])
def test_host_loop_matches_reference_trajectories(name):
    """Same estimator code as on the GPU, oracle kernels: subsets bit-exact, state to 1e-9."""
    g = load_golden('traj_small')
    kw, X, dt = small_case_params(name)
    est = _host_estimator()(**kw)
    est.fit(X)
    tol = 1e-9
    assert rel_fro(est.components_, g[name + '/D_final']) < tol
    assert rel_fro(est.C_, g[name + '/C_final']) < tol
    assert rel_fro(est.B_, g[name + '/B_final']) < tol
    assert rel_fro(est.code_, g[name + '/code_final']) < tol
    assert est.n_iter_ == int(g[name + '/n_iter'])


def test_set_params_and_transform_host():
    Est = _host_estimator()
    rs = np.random.RandomState(0)
    X = rs.randn(60, 12)
    est = Est(n_components=3, batch_size=10, reduction=2, random_state=0, code_alpha=0.1)
    est.fit(X)
    assert est.set_params(G_agg='full', Dx_agg='average') is est
    assert est.G_agg == 'full' and est.Dx_agg == 'average'
    np.testing.assert_allclose(est.G_, est.components_.dot(est.components_.T), rtol=1e-12)
    est.partial_fit(X[:20], np.arange(20))
    code = est.transform(X)
    assert code.shape == (60, 3) and np.isfinite(est.score(X))
    with pytest.raises(ValueError):
        Est(optimizer='adam').prepare(n_samples=5, n_features=4)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, kw, X_parts, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        est = _host_estimator()(**kw)
        X = X_parts[rank]
        est.prepare(n_samples=X.shape[0], X=X_parts[0])        # every rank initialises from the same rows
        est.partial_fit(X)
        out[rank] = dict(D=est.components_, C=est.C_, B=est.B_, code=est.code_, n_iter=est.n_iter_)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('variant', ['l1_masked', 'ridge_l1atoms', 'full'])
def test_two_rank_gloo_equals_double_batch(variant, oracle):
    extra = {'l1_masked': dict(code_alpha=0.1), 'ridge_l1atoms': dict(code_l1_ratio=0, comp_l1_ratio=1, code_alpha=0.01),
             'full': dict(G_agg='full', Dx_agg='full', code_alpha=0.1)}[variant]
    rs = np.random.RandomState(3)
    b, steps, p, k = 8, 6, 30, 5
    X0 = rs.randn(b * steps, 6).dot(rs.randn(6, p))
    X1 = rs.randn(b * steps, 6).dot(rs.randn(6, p))
    kw = dict(n_components=k, batch_size=b, reduction=2, random_state=0, learning_rate=0.9, **extra)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main, args=(2, _free_port(), kw, [X0, X1], out), nprocs=2, join=True)
    # single-process oracle with batch 2b on [rank0 batch t ; rank1 batch t]
    Xc = np.concatenate([np.concatenate([X0[t * b:(t + 1) * b], X1[t * b:(t + 1) * b]]) for t in range(steps)])
    kw1 = dict(kw, batch_size=2 * b)
    pr = oracle.SomfParams(**kw1)
    st = oracle.prepare(pr, n_samples=Xc.shape[0], X=X0)
    oracle.partial_fit(st, pr, Xc)
    for r in (0, 1):
        assert rel_fro(out[r]['D'], st.D) < 1e-10, (variant, r)
        assert rel_fro(out[r]['C'], st.C) < 1e-10
        assert rel_fro(out[r]['B'], st.B) < 1e-10
        assert out[r]['n_iter'] == st.n_iter
    assert_array_equal(out[0]['D'], out[1]['D'])               # replicas stay bit-identical
    codes = np.concatenate([np.concatenate([out[0]['code'][t * b:(t + 1) * b], out[1]['code'][t * b:(t + 1) * b]])
                            for t in range(steps)])
    assert rel_fro(codes, st.code) < 1e-10


def _rank_main_unseeded(rank, world, port, kw, X_parts, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        est = _host_estimator()(**kw)
        X = X_parts[rank]
        est.prepare(n_samples=X.shape[0], X=X)                  # every rank offers ITS OWN rows: rank 0's atoms win
        try:
            est.partial_fit(X)
            out[rank] = dict(D=est.components_, C=est.C_, B=est.B_, n_iter=est.n_iter_, err=None)
        except ValueError as e:
            out[rank] = dict(err=str(e))
    finally:
        dist.destroy_process_group()


def test_two_rank_unseeded_ragged_replicas_identical():
    """random_state=None (each process would seed itself from the OS), different initial rows per rank and a ragged
    last minibatch (4 + 1 rows): prepare() makes every rank continue rank 0's generator and start from rank 0's
    atoms, partial_fit weighs every minibatch with the true global batch size."""
    rs = np.random.RandomState(4)
    b, p, k = 8, 24, 4
    X0 = rs.randn(44, 6).dot(rs.randn(6, p))
    X1 = rs.randn(41, 6).dot(rs.randn(6, p))
    kw = dict(n_components=k, batch_size=b, reduction=2, random_state=None, learning_rate=0.9, code_alpha=0.1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_unseeded, args=(2, _free_port(), kw, [X0, X1], out), nprocs=2, join=True)
    assert out[0]['err'] is None and out[1]['err'] is None
    assert_array_equal(out[0]['D'], out[1]['D'])
    assert_array_equal(out[0]['C'], out[1]['C'])               # the attribute is the sum over the ranks on every rank
    assert_array_equal(out[0]['B'], out[1]['B'])
    assert out[0]['n_iter'] == out[1]['n_iter'] == 85
    assert np.all(np.isfinite(out[0]['D']))


def test_two_rank_unequal_batch_counts_raise():
    rs = np.random.RandomState(4)
    X0, X1 = rs.randn(40, 12), rs.randn(17, 12)                 # 5 and 3 minibatches of 8 rows
    kw = dict(n_components=3, batch_size=8, random_state=0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_unseeded, args=(2, _free_port(), kw, [X0, X1], out), nprocs=2, join=True)
    for r in (0, 1):
        assert out[r]['err'] and 'same number of minibatches' in out[r]['err']


def test_bench_self_launch_relays_failure_without_hanging():
    """`python bench.py --gpus 2` with no launcher starts its two ranks itself.  Without a GPU every rank refuses to
    run (no CPU fallback); the launcher must come back with that status instead of raising its own error or hanging,
    and must not have imported torch or the library itself (it forks the ranks before anything touches the GPU)."""
    import subprocess
    import sys
    from .conftest import ROOT
    if torch.cuda.is_available():
        pytest.skip('a GPU is present: covered by test_bench_launches_its_own_ranks')
    env = {k_: v for k_, v in os.environ.items() if k_ not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--steady-steps', '0'], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs a GPU') == 2, r.stderr[-2000:]      # both ranks were started
    assert 'must be launched with' not in r.stderr
