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


from modl_amd.dict_fact import DictFact as _DictFact


class HostDictFact(_DictFact):
    """the product's estimator with the oracle standing in for the device kernels (module level: picklable)"""

    def _make_backend(self):
        from .oracle_backend import OracleBackend
        return OracleBackend()


def _host_estimator():
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
    # (the first rank to fail makes the launcher end the other one, which may not have spoken yet)
    assert r.stderr.count('bench.py needs a GPU') >= 1, r.stderr[-2000:]
    assert 'must be launched with' not in r.stderr


def _rank_main_pickle(rank, world, port, kw, X_parts, out):
    import pickle
    import warnings
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        est = _host_estimator()(**kw)
        X = X_parts[rank]
        est.prepare(n_samples=4 * X.shape[0], X=X_parts[0])
        est.partial_fit(X, np.arange(X.shape[0]))
        res = dict(B=est.B_, C=est.C_, local_B=est.local_B_)     # B_ / C_: collective reads, both ranks
        if rank == 0:                                            # a rank-0-only checkpoint must not hang ...
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter('always')
                res['partial_pickle'] = pickle.dumps(est)
            res['warned'] = any('partial' in str(w_.message) for w_ in wlist)
        est.consolidate_statistics()                             # collective
        res['local_B_after'] = est.local_B_                      # no communication
        if rank == 0:
            with warnings.catch_warnings():
                warnings.simplefilter('error')                   # ... and after consolidation it is silent and whole
                res['pickle'] = pickle.dumps(est)
        est.partial_fit(X[:kw['batch_size']], np.arange(kw['batch_size']))     # the run goes on from the consolidated state
        res['D_next'], res['B_next'] = est.components_, est.B_
        if rank == 0:
            # ... and its statistics are per-rank partial sums again, whatever route the minibatches took (the Python
            # loop here, one library call per chunk on the GPU): a pickle written now must say so
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter('always')
                res['pickle_after_fit'] = pickle.dumps(est)
            res['warned_after_fit'] = any('partial' in str(w_.message) for w_ in wlist)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_two_rank_pickle_consolidated_loads_on_one_rank(oracle):
    """Several ranks keep C_ / B_ as per-rank partial sums.  A pickle written on rank 0 after consolidate_statistics()
    holds the SUMMED statistics (the reference's contract, dict_fact.py:116-124): loaded in a one-rank process it
    continues like the one-rank run with the double batch.  One written without consolidation says so and refuses to
    load anywhere else.  Setting B_ / C_ with several ranks sets the sum, not world x the value."""
    import pickle
    rs = np.random.RandomState(5)
    b, steps, p, k = 8, 4, 30, 5
    X0 = rs.randn(b * steps, 6).dot(rs.randn(6, p))
    X1 = rs.randn(b * steps, 6).dot(rs.randn(6, p))
    Xnext = rs.randn(2 * b * 3, 6).dot(rs.randn(6, p))
    kw = dict(n_components=k, batch_size=b, reduction=2, random_state=0, learning_rate=0.9, code_alpha=0.1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_pickle, args=(2, _free_port(), kw, [X0, X1], out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert_array_equal(r0['B'], r1['B'])
    assert_array_equal(r0['local_B_after'], r0['B'])                     # rank 0 holds the sum, bit for bit ...
    assert not r1['local_B_after'].any()                                 # ... the other rank zeros
    assert np.abs(r0['local_B'] - r0['B']).max() > 0                     # (before, rank 0 held a share only)
    assert r0['warned']
    with pytest.raises(ValueError, match='rank 0 of 2'):
        pickle.loads(r0['partial_pickle'])
    assert r0['warned_after_fit']                                        # consolidate -> partial_fit -> pickle: partial again
    with pytest.raises(ValueError, match='rank 0 of 2'):
        pickle.loads(r0['pickle_after_fit'])
    # the one-rank reference run: batch 2b on [rank0 batch t ; rank1 batch t], then the next rows
    Xc = np.concatenate([np.concatenate([X0[t * b:(t + 1) * b], X1[t * b:(t + 1) * b]]) for t in range(steps)])
    ref = _host_estimator()(**dict(kw, batch_size=2 * b))
    ref.prepare(n_samples=4 * b * steps, X=X0)
    ref.partial_fit(Xc, np.arange(Xc.shape[0]))
    est = pickle.loads(r0['pickle'])                                     # a one-rank process (no process group here)
    assert_array_equal(est.B_, r0['B'])
    assert_array_equal(est.C_, r0['C'])
    assert rel_fro(est.B_, ref.B_) < 1e-10 and rel_fro(est.C_, ref.C_) < 1e-10
    assert rel_fro(est.components_, ref.components_) < 1e-10 and est.n_iter_ == ref.n_iter_
    # continuing from the loaded state == continuing from the 2-rank state set into a fresh one-rank estimator, bit
    # for bit (the pickle lost nothing) and == the one-rank run up to summation order
    est.batch_size = 2 * b
    idx = np.arange(2 * b * steps, 2 * b * steps + Xnext.shape[0])
    est.partial_fit(Xnext, idx)
    ref.partial_fit(Xnext, idx)
    assert rel_fro(est.components_, ref.components_) < 1e-9
    assert rel_fro(est.B_, ref.B_) < 1e-9 and rel_fro(est.C_, ref.C_) < 1e-9
    # the 2-rank run itself went on from the consolidated state exactly as from the spread one (sum unchanged)
    assert_array_equal(out[0]['D_next'], out[1]['D_next'])
    assert np.all(np.isfinite(out[0]['B_next']))


def _rank_main_setters(rank, world, port, kw, X0, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        est = _host_estimator()(**kw)
        est.prepare(n_samples=X0.shape[0], X=X0)
        v = np.arange(kw['n_components'] * X0.shape[1], dtype=np.float64).reshape(kw['n_components'], -1)
        est.B_ = v
        est.C_ = np.eye(kw['n_components'])
        res = dict(B=est.B_, C=est.C_, v=v)
        est.local_B_ = (rank + 1) * v                                    # the local twin sets THIS rank's share as given
        res['local_B'] = est.local_B_
        res['B_from_local'] = est.B_                                     # (collective) = 1 v + 2 v
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_two_rank_setters_set_the_sum():
    rs = np.random.RandomState(6)
    X0 = rs.randn(16, 12)
    kw = dict(n_components=3, batch_size=8, random_state=0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_setters, args=(2, _free_port(), kw, X0, out), nprocs=2, join=True)
    for r in (0, 1):
        assert_array_equal(out[r]['B'], out[r]['v'])
        assert_array_equal(out[r]['C'], np.eye(3))
        assert_array_equal(out[r]['local_B'], (r + 1) * out[r]['v'])
        assert_array_equal(out[r]['B_from_local'], 3 * out[r]['v'])


# ---- world sizes 4 and 8 (round 6: the driver's 8-GPU run must not be the first time eight ranks meet) ---------------------
def _interleave(parts, b, steps):
    """the one-rank run's rows: minibatch t = [rank 0's minibatch t ; rank 1's ; ...] (ragged tails included)"""
    rows = []
    for t in range(steps):
        for X in parts:
            rows.append(X[t * b:(t + 1) * b])
    return np.concatenate(rows)


def _rank_main_wide(rank, world, port, kw, X_parts, out):
    import pickle
    import warnings
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        est = _host_estimator()(**kw)
        X = X_parts[rank]
        est.prepare(n_samples=4 * max(x.shape[0] for x in X_parts), X=X_parts[0])
        est.partial_fit(X, np.arange(X.shape[0]))
        res = dict(D=est.components_, C=est.C_, B=est.B_, code=est.code_[:X.shape[0]], n_iter=est.n_iter_)
        est.consolidate_statistics()                             # collective: rank 0 ends with the sums, the others with zeros
        res['local_B_after'] = est.local_B_
        if rank == 0:
            with warnings.catch_warnings():
                warnings.simplefilter('error')
                res['pickle'] = pickle.dumps(est)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_wide_world_gloo_equals_wide_batch(world, oracle):
    """W ranks with local batch b == one rank with batch W b on the interleaved rows - at W = 4 and 8, with a RAGGED last
    minibatch of a different length on every rank (weighed with the true global batch size), replicas bit-identical, the
    statistics read through the estimator the same sums on every rank, and rank 0's pickle after consolidate_statistics()
    holding the whole statistics (loads in a one-rank process)."""
    import pickle
    rs = np.random.RandomState(30 + world)
    b, steps, p, k = 4, 4, 24, 4
    tails = [1 + (r * 3) % b for r in range(world)]                       # rows of every rank's last minibatch: 1 .. b
    parts = [rs.randn((steps - 1) * b + tails[r], 6).dot(rs.randn(6, p)) for r in range(world)]
    kw = dict(n_components=k, batch_size=b, reduction=2, random_state=0, learning_rate=0.9, code_alpha=0.1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_wide, args=(world, _free_port(), kw, parts, out), nprocs=world, join=True)
    Xc = _interleave(parts, b, steps)
    assert Xc.shape[0] == sum(x.shape[0] for x in parts)
    pr = oracle.SomfParams(**dict(kw, batch_size=world * b))
    st = oracle.prepare(pr, n_samples=Xc.shape[0], X=parts[0])
    oracle.partial_fit(st, pr, Xc)
    for r in range(world):
        assert rel_fro(out[r]['D'], st.D) < 1e-10, r
        assert rel_fro(out[r]['C'], st.C) < 1e-10 and rel_fro(out[r]['B'], st.B) < 1e-10
        assert out[r]['n_iter'] == st.n_iter == Xc.shape[0]
        assert_array_equal(out[r]['D'], out[0]['D'])                      # replicas bit-identical
        assert_array_equal(out[r]['B'], out[0]['B'])                      # (collective read: the same sum everywhere)
        if r:
            assert not out[r]['local_B_after'].any()
    codes = _interleave([out[r]['code'] for r in range(world)], b, steps)
    assert rel_fro(codes, st.code) < 1e-10
    assert_array_equal(out[0]['local_B_after'], out[0]['B'])
    est = pickle.loads(out[0]['pickle'])                                  # no process group here: a one-rank process
    assert_array_equal(est.B_, out[0]['B'])
    assert_array_equal(est.C_, out[0]['C'])
    assert_array_equal(est.components_, out[0]['D'])


@pytest.mark.parametrize('world', [4, 8])
def test_wide_world_unequal_batch_counts_raise(world):
    rs = np.random.RandomState(4)
    parts = [rs.randn(40 if r != world - 1 else 17, 12) for r in range(world)]     # 5 minibatches everywhere but on the last rank (3)
    kw = dict(n_components=3, batch_size=8, random_state=0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_unseeded, args=(world, _free_port(), kw, parts, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r]['err'] and 'same number of minibatches' in out[r]['err']


@pytest.mark.parametrize('world', [4, 8])
def test_wide_world_unseeded_replicas_identical(world):
    rs = np.random.RandomState(40 + world)
    b, p, k = 8, 24, 4
    parts = [rs.randn(41 + (r % 7), 6).dot(rs.randn(6, p)) for r in range(world)]   # 6 minibatches each, ragged tails 1 .. 7
    kw = dict(n_components=k, batch_size=b, reduction=2, random_state=None, learning_rate=0.9, code_alpha=0.1)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main_unseeded, args=(world, _free_port(), kw, parts, out), nprocs=world, join=True)
    assert all(out[r]['err'] is None for r in range(world))
    for r in range(1, world):
        assert_array_equal(out[0]['D'], out[r]['D'])
        assert_array_equal(out[0]['C'], out[r]['C'])
        assert_array_equal(out[0]['B'], out[r]['B'])
    assert out[0]['n_iter'] == sum(x.shape[0] for x in parts)
    assert np.all(np.isfinite(out[0]['D']))


def test_failed_minibatch_rewinds_the_subset_lookahead():
    """The feature subsets of the coming minibatches are drawn ahead on a worker thread.  When a minibatch fails, the
    draws nobody consumed must not be lost: after the error the sampler continues the reference's MT19937 subset
    stream (sampler.pyx:41-70) right behind the last subset that reached a minibatch."""
    rs = np.random.RandomState(7)
    X = rs.randn(96, 20)
    kw = dict(n_components=4, batch_size=8, reduction=2, random_state=0, code_alpha=0.1)
    clean = HostDictFact(**kw)
    clean.prepare(n_samples=96, X=X)
    seen = []
    orig = clean.feature_sampler_.yield_subset
    for _ in range(12):
        seen.append(orig(2))                                   # the stream itself: 12 subsets
    est = HostDictFact(**kw)
    est.prepare(n_samples=96, X=X)
    calls = []
    real_fit = est._single_batch_fit.__func__

    def failing(self, Xh, batch, idx, b_global=None):
        if len(calls) == 3:
            calls.append('boom')
            raise RuntimeError('boom')
        calls.append(batch.start)
        return real_fit(self, Xh, batch, idx, b_global=b_global)
    est._single_batch_fit = failing.__get__(est)
    with pytest.raises(RuntimeError):
        est.partial_fit(X, np.arange(96))                      # 12 minibatches, the 4th fails before it draws
    # three subsets were consumed: the next draw is the fourth of the stream, whatever the worker had drawn ahead
    assert_array_equal(est.feature_sampler_.yield_subset(2), seen[3])


def test_chunk_call_error_books_the_enqueued_minibatches():
    """The one-call-per-chunk route (modl_somf_partial_fit_chunk) reports how many minibatches it enqueued when one
    fails and leaves n_iter / the generators behind the last of them (GPU side: test_chunk_call_error_is_consistent).
    The Python side must book exactly those minibatches - n_iter_, sample_n_iter_ - before it raises, so that a
    retry weighs the next minibatch as the reference would (dict_fact.py:510-515)."""
    from modl_amd._lib import ModlError
    rs = np.random.RandomState(8)
    X = rs.randn(96, 20)
    b = 8
    est = HostDictFact(n_components=4, batch_size=b, reduction=2, random_state=0, code_alpha=0.1)
    est.prepare(n_samples=200, X=X)
    be = est._backend
    calls = []

    def fit_chunk(Xh, batch_size, idx, sampler, np_rs, n_iter, lr, red, b_global=None, comm=None):
        calls.append((Xh.shape[0], n_iter))
        if len(calls) == 2:                                   # the second chunk fails in its 4th minibatch
            e = ModlError('modl_somf_partial_fit_chunk failed: bad argument (code -1)')
            e.n_iter, e.n_done = n_iter + 3 * batch_size, 3
            raise e
        return n_iter + Xh.shape[0], -(-Xh.shape[0] // batch_size)
    be.fit_chunk = fit_chunk
    est._chunk_call_applies = lambda be_, idx_: True
    est._native_comm = lambda be_: None
    est._host_chunk_rows = 5 * b                              # chunks of 5 minibatches: 40 + 40 + 16 rows
    idx = np.arange(100, 196)

    class Chunks:                                             # (stands in for the pinned host stream of the GPU backend)
        def __init__(self, be_, X_, rows):
            self.X, self.rows = X_, rows

        def __iter__(self):
            for r0 in range(0, self.X.shape[0], self.rows):
                yield r0, self.X[r0:r0 + self.rows]
    import modl_amd.dict_fact as mod
    orig, be.plan = mod._HostChunks, object()
    mod._HostChunks = Chunks
    try:
        with pytest.raises(ModlError):
            est.partial_fit(X, idx)
    finally:
        mod._HostChunks = orig
    assert calls == [(40, 0), (40, 40)]
    assert est.n_iter_ == 40 + 3 * b                          # the first chunk and three minibatches of the second
    want = np.zeros(200, dtype=int)
    want[100:100 + 40 + 3 * b] = 1
    assert_array_equal(est.sample_n_iter_, want)


def test_failed_native_communicator_takes_the_torch_route():
    """when the library's RCCL communicator cannot be created (native_comm() returns None on every
    rank) the one-call-per-chunk route must NOT run with comm = None - that is the single-GPU step without any
    reduction, and the replicas would diverge silently.  The per-minibatch phase1 / all-reduce / phase2 loop with
    torch's collective runs instead, from the very first partial_fit on."""
    rs = np.random.RandomState(11)
    X = rs.randn(48, 20)
    est = HostDictFact(n_components=4, batch_size=8, reduction=2, random_state=0, code_alpha=0.1)
    est.prepare(n_samples=48, X=X)
    be = est._backend
    est._force_reduce = True                                  # (what world > 1 switches on; one process here)
    est._native_rccl = True                                   # the library's communicator is wanted ...
    trace = []

    def native_comm(dist_):                                   # ... and cannot be created
        trace.append('native_comm')
        be._comm_failed = True
        return None
    be.native_comm = native_comm

    def fit_chunk(*a, **kw):
        raise AssertionError('the chunk call ran without a communicator: no reduction at all')
    be.fit_chunk = fit_chunk
    be.step_dist = lambda *a, **kw: (_ for _ in ()).throw(AssertionError('step_dist without a communicator'))
    p1, p2 = be.phase1, be.phase2
    be.phase1 = lambda *a, **kw: (trace.append('phase1'), p1(*a, **kw))[1]
    be.phase2 = lambda *a, **kw: (trace.append('phase2'), p2(*a, **kw))[1]
    est._all_reduce = lambda head: trace.append('all_reduce')
    assert est._chunk_call_applies(be, None)                  # the route the first call would have taken
    est.partial_fit(X)
    assert trace[0] == 'native_comm'
    assert trace[1:] == ['phase1', 'all_reduce', 'phase2'] * 6
    # the same rows through the plain one-rank loop: the identity "reduction" changes nothing
    ref = HostDictFact(n_components=4, batch_size=8, reduction=2, random_state=0, code_alpha=0.1)
    ref.prepare(n_samples=48, X=X)
    ref.partial_fit(X)
    assert rel_fro(est.components_, ref.components_) < 1e-12
    # a second call goes straight to torch's route (the failure is remembered)
    est.partial_fit(X)
    assert trace.count('native_comm') == 1
