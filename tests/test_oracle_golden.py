"""The oracle (CPU restatement) against golden vectors recorded from the real
reference (tests/golden/make_golden.py) and against the reference's own known
answers (modl/utils/randomkit/tests/test_random.py, test_sampler.py)."""
import ctypes as C
import os

import numpy as np
import pytest
from numpy.testing import assert_array_equal

from .conftest import load_golden, rel_fro, ROOT

TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-10}


def test_rng_known_answers(oracle):
    # modl/utils/randomkit/tests/test_random.py:10-38
    rs = oracle.OracleRandomState(0)
    vals = [rs.randint(10) for _ in range(10000)]
    assert abs(np.mean(vals) - 5.018) < 1e-12
    vals = [rs.binomial(1000, 0.8) for _ in range(10000)]
    assert abs(np.mean(vals) - 799.8564) < 1e-9
    rs = oracle.OracleRandomState(0)
    x = np.arange(10, dtype=np.int64)
    rs.shuffle(x)
    assert_array_equal(x, [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    rs = oracle.OracleRandomState(0)
    assert_array_equal(rs.permutation(10), [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    rs = oracle.OracleRandomState(0)
    a = np.arange(10, dtype=np.int64)
    b = np.arange(9, -1, -1, dtype=np.int64)
    perm = rs.shuffle_with_trace([a, b])
    assert_array_equal(a, [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    assert_array_equal(b, [7, 1, 5, 0, 8, 3, 2, 6, 9, 4])
    assert_array_equal(a, perm)


def test_rng_golden(oracle):
    g = load_golden('rng')
    for a, s in enumerate(g['seeds']):
        rs = oracle.OracleRandomState(int(s))
        for b, h in enumerate(g['highs']):
            got = [rs.randint(int(h)) for _ in range(8)]
            assert_array_equal(got, g['randint'][a, b])
    rs = oracle.OracleRandomState(0)
    assert_array_equal([rs.randint(10) for _ in range(10000)], g['ka_randint10'])
    assert_array_equal([rs.binomial(1000, 0.8) for _ in range(10000)], g['ka_binomial'])
    rs = oracle.OracleRandomState(7)
    for rep in range(3):
        for i, (n, p) in enumerate(zip(g['binom_n'], g['binom_p'])):
            got = [rs.binomial(int(n), float(p)) for _ in range(40)]
            assert_array_equal(got, g['binom_draws'][rep, i], err_msg='binomial(%d, %g)' % (n, p))
    for n in (1, 2, 10, 257, 1000):
        rs = oracle.OracleRandomState(0)
        x = np.arange(n, dtype=np.int64)
        rs.shuffle(x)
        assert_array_equal(x, g['shuffle_%d' % n])
        assert_array_equal(rs.permutation(n), g['perm_%d' % n])
    rs = oracle.OracleRandomState(3)
    a = np.arange(12, dtype=np.int64)
    b2 = np.arange(24, dtype=np.float64).reshape(12, 2)
    tr = rs.shuffle_with_trace([a, b2])
    assert_array_equal(tr, g['trace_perm'])
    assert_array_equal(a, g['trace_a'])
    assert_array_equal(b2, g['trace_b'])


def test_rng_against_compiled_reference(oracle):
    """oracle/_ref/librk_ref.so = the reference's own randomkit.c + distributions.c
    compiled in place (oracle/Makefile `ref`); only present where it was built."""
    so = os.path.join(ROOT, 'oracle', '_ref', 'librk_ref.so')
    if not os.path.exists(so):
        pytest.skip('oracle/_ref not built (no /root/reference on this box)')
    ref = C.CDLL(so)
    buf = C.create_string_buffer(8192)      # rk_state is ~5.2 KB
    ref.rk_seed.argtypes = [C.c_ulong, C.c_void_p]
    ref.rk_random.restype = C.c_ulong
    ref.rk_random.argtypes = [C.c_void_p]
    ref.rk_interval.restype = C.c_ulong
    ref.rk_interval.argtypes = [C.c_ulong, C.c_void_p]
    ref.rk_double.restype = C.c_double
    ref.rk_double.argtypes = [C.c_void_p]
    ref.rk_binomial.restype = C.c_long
    ref.rk_binomial.argtypes = [C.c_void_p, C.c_long, C.c_double]
    for seed in (0, 5, 123456789, 2 ** 40 + 17):
        ref.rk_seed(seed, buf)
        rs = oracle.OracleRandomState(seed)
        assert [ref.rk_random(buf) for _ in range(2000)] == [rs.random_u32() for _ in range(2000)]
        for mx in (1, 7, 1000, 2 ** 33):
            assert [ref.rk_interval(mx, buf) for _ in range(50)] == [rs.randint(mx) for _ in range(50)]
        assert [ref.rk_double(buf) for _ in range(50)] == [rs.double() for _ in range(50)]
        for (n, p) in ((10000, 0.1), (50, 0.3), (10000, 1 / 12.), (200000, 0.91), (10000, 0.1)):
            assert [ref.rk_binomial(buf, n, p) for _ in range(200)] == [rs.binomial(n, p) for _ in range(200)]


def test_sampler_known_answers(oracle):
    # modl/utils/randomkit/tests/test_sampler.py:6-45
    s = oracle.OracleSampler(100, True, True, 0)
    assert_array_equal(s.yield_subset(10), [14, 58, 11, 49, 36, 62, 87, 45, 72, 47, 48, 13, 98, 97, 25, 93])
    assert np.mean([s.yield_subset(10).shape[0] for _ in range(100)]) == 10.19
    s = oracle.OracleSampler(100, False, False, 0)
    A = np.concatenate([s.yield_subset(10) for _ in range(10)])
    assert_array_equal(np.sort(A), np.arange(100))
    s = oracle.OracleSampler(100, False, True, 0)
    assert_array_equal(s.yield_subset(10), [6, 55, 1, 25, 87, 49, 69, 63, 13, 8])
    s = oracle.OracleSampler(100, True, False, 0)
    A = np.concatenate([s.yield_subset(10) for _ in range(20)])
    assert_array_equal(np.sort(A[:100]), np.arange(100))


def test_sampler_golden(oracle):
    g = load_golden('sampler')
    for i, (rng_, rand_size, repl, red, seed) in enumerate(g['cfg']):
        s = oracle.OracleSampler(int(rng_), bool(rand_size), bool(repl), int(seed))
        lens = g['lens_%d' % i]
        draws = [s.yield_subset(red) for _ in range(len(lens))]
        assert_array_equal([len(d) for d in draws], lens)
        assert_array_equal(np.concatenate(draws), g['draws_%d' % i])
    s = oracle.OracleSampler(300, True, False, 99)
    dr = [s.yield_subset(r) for r in g['var_red']]
    assert_array_equal([len(d) for d in dr], g['var_lens'])
    assert_array_equal(np.concatenate(dr), g['var_draws'])


def test_batch_weight_golden(oracle):
    g = load_golden('batch_weight')
    for count, b, lr, off, w in g['cases']:
        got = oracle.batch_weight(int(count), int(b), lr, off)
        assert got == w or (np.isnan(got) and np.isnan(w))    # count < batch_size gives nan in the reference too


def test_enet_golden(oracle):
    g = load_golden('enet')
    for i, (dti, n, l1, radius, nrm, sc_radius) in enumerate(g['meta']):
        dt = np.float32 if dti == 0 else np.float64
        v = g['v_%d' % i]
        assert v.dtype == dt
        assert abs(oracle.enet_norm(v, l1) - nrm) <= 1e-6 * max(1, abs(nrm)) * (1 if dti == 0 else 1e-6)
        out = np.zeros_like(v)
        oracle.enet_projection(v, out, radius, l1)
        np.testing.assert_allclose(out, g['proj_%d' % i], rtol=2e-6 if dti == 0 else 1e-12,
                                   atol=1e-7 if dti == 0 else 1e-14)
        vs = v.copy()
        oracle.enet_scale(vs, l1, sc_radius)
        np.testing.assert_allclose(vs, g['scaled_%d' % i], rtol=2e-6 if dti == 0 else 1e-12)


def test_cd_golden(oracle):
    g = load_golden('cd')
    for i, row in enumerate(g['meta']):
        dti, k, p, b, n, l1, alpha, pos, tol, mi = row[:10]
        dt = np.float32 if dti == 0 else np.float64
        tag = '%s_%d' % ('f32' if dti == 0 else 'f64', int(k))
        G, Gm, Dx0, X = g['G_' + tag], g['Gm_' + tag], g['Dx_' + tag], g['X_' + tag]
        idx = g['idx_%d' % i]
        code = g['code_in_%d' % i].copy()
        oracle.enet_regression_single_gram(G, Dx0.copy(), X, code, idx, l1, alpha, bool(pos), tol, int(mi))
        assert rel_fro(code, g['single_code_%d' % i]) < TOL[np.dtype(dt)], (i, row)
        code = g['code_in_%d' % i].copy()
        oracle.enet_regression_multi_gram(Gm.copy(), Dx0.copy(), X, code, idx, l1, alpha, bool(pos), tol, int(mi))
        assert rel_fro(code, g['multi_code_%d' % i]) < TOL[np.dtype(dt)], (i, row)
    for tag in ('f32_16', 'f32_64', 'f64_16', 'f64_64'):
        Ga = g['Gm_' + tag].copy()
        oracle.update_G_average(Ga, g['G_' + tag], g['Gavg_w_' + tag])
        np.testing.assert_allclose(Ga, g['Gavg_out_' + tag], rtol=1e-6 if 'f32' in tag else 1e-14)


def test_transform_golden(oracle):
    g = load_golden('transform')
    for dn in ('f64', 'f32'):
        X, D = g['X_' + dn], g['D_' + dn]
        for l1, alpha, pos in ((1.0, 0.1, False), (0.0, 0.1, False), (0.5, 0.05, True)):
            pr = oracle.SomfParams(code_l1_ratio=l1, code_alpha=alpha, code_pos=pos)
            key = '%s_%g_%g_%d' % (dn, l1, alpha, pos)
            assert rel_fro(oracle.transform(pr, D, X), g['code_' + key]) < TOL[D.dtype]
            assert abs(oracle.score(pr, D, X) - g['score_' + key]) < 1e-5 * abs(g['score_' + key])


def synth(n, p, k0, seed, dtype):
    rs = np.random.RandomState(seed)
    Q = rs.randn(k0, p)
    X = rs.randn(n, k0).dot(Q)
    return np.ascontiguousarray(X.astype(dtype))


SMALL_BASE = dict(n_components=6, batch_size=10, reduction=2, n_epochs=2, random_state=0, learning_rate=0.9)
VARIANTS = {
    'ridge_l1atoms': dict(code_l1_ratio=0, comp_l1_ratio=1, code_alpha=0.01),
    'ridge_l1atoms_pos': dict(code_l1_ratio=0, comp_l1_ratio=1, comp_pos=True, code_alpha=0.01),
    'enet_atoms': dict(code_l1_ratio=0.5, comp_l1_ratio=0.5, code_alpha=0.05),
    'nmf': dict(comp_pos=True, code_pos=True, code_alpha=0.05),
    'sgd': dict(optimizer='sgd', step_size=0.1, code_alpha=0.05),
    'fixed_norepl': dict(rand_size=False, replacement=False, code_alpha=0.05),
    'r1': dict(reduction=1, code_alpha=0.05),
    'avg_ridge': dict(code_l1_ratio=0, comp_l1_ratio=1, code_alpha=0.01, G_agg='average', Dx_agg='average'),
}


def small_case_params(name):
    """kwargs + input for a case name of traj_small.npz (mirrors make_golden.gen_traj)."""
    parts = name.split('_')
    dn = parts[-1]
    dt = np.float32 if dn == 'f32' else np.float64
    X = synth(120, 40, 6, 0, dt)
    kw = dict(SMALL_BASE)
    if parts[0] == 'agg':
        kw.update(code_alpha=0.1, G_agg=parts[1], Dx_agg=parts[2])
    else:
        vn = '_'.join(parts[1:-1])
        kw.update(VARIANTS[vn])
        if vn == 'nmf':
            X = np.abs(X)
    return kw, X, dt


def _cases():
    return [str(c) for c in load_golden('traj_small')['cases']]


@pytest.mark.parametrize('name', _cases())
def test_trajectory_small(oracle, name):
    g = load_golden('traj_small')
    kw, X, dt = small_case_params(name)
    st = oracle.fit(oracle.SomfParams(**kw), X, trace=True)
    lens = g[name + '/subset_len']
    assert_array_equal([len(t['subset']) for t in st.trace], lens)              # bit-exact draws
    assert_array_equal(np.concatenate([t['subset'] for t in st.trace]), g[name + '/subset'])
    tol = 2e-5 if dt == np.float32 else 1e-9
    # f32 trajectories: tolerance-stopped CD + unpinned BLAS order -> compare early snapshots tightly,
    # the end state loosely
    codes = np.concatenate([t['code'] for t in st.trace])
    nb_first = int(g[name + '/code_len'][0])
    assert rel_fro(codes[:nb_first], g[name + '/code'][:nb_first]) < tol
    if dt == np.float64:
        assert rel_fro(codes, g[name + '/code']) < tol
        assert rel_fro(st.D, g[name + '/D_final']) < tol
        assert rel_fro(st.C, g[name + '/C_final']) < tol
        assert rel_fro(st.B, g[name + '/B_final']) < tol
        assert rel_fro(st.code, g[name + '/code_final']) < tol
        if (name + '/G_final') in g:
            assert rel_fro(st.G, g[name + '/G_final']) < tol
    else:
        assert rel_fro(st.D, g[name + '/D_final']) < 1e-3


def test_trajectory_config1(oracle):
    """BASELINE config 1: 2000 x 500, k = 16, r = 1, f64 (CPU reference path)."""
    g = load_golden('traj_c1')
    X = synth(2000, 500, 16, 0, np.float64)
    pr = oracle.SomfParams(n_components=16, reduction=1, random_state=0, n_epochs=1, code_alpha=1e-4)
    st = oracle.fit(pr, X, trace=True)
    assert_array_equal([len(t['subset']) for t in st.trace], g['c1/subset_len'])
    assert_array_equal(np.concatenate([t['subset'] for t in st.trace])[:1500], g['c1/subset_head'])
    assert rel_fro(st.D, g['c1/D_final']) < 1e-9
    assert rel_fro(st.C, g['c1/C_final']) < 1e-9
    assert rel_fro(st.code[:64], g['c1/code_final_head']) < 1e-9
    assert st.n_iter == int(g['c1/n_iter'])


@pytest.mark.parametrize('r', [1, 10])
@pytest.mark.parametrize('dn', ['f64', 'f32'])
def test_trajectory_headline_shape(oracle, dn, r):
    """The metric's shape recorded from the REAL reference (k = 256, p = 10 000, b = 256, 4 minibatches,
    tests/golden/traj_headline.npz): draws bit-exact; f64 everything <= 1e-9; f32 the first minibatch <= 1e-5 and the
    later ones within the reference's own f32-vs-f64 distance (float summation order is unpinned)."""
    from .conftest import m1_rows, HEADLINE_KW, headline_observables, subset_checksum
    g = load_golden('traj_headline')
    n, p, k, b = (int(v) for v in g['shape'])
    dt = np.float32 if dn == 'f32' else np.float64
    X = np.ascontiguousarray(m1_rows(n, p).astype(dt))
    name = 'm1_r%d_%s/' % (r, dn)
    pr = oracle.SomfParams(reduction=r, **HEADLINE_KW)
    st = oracle.prepare(pr, n_samples=n, X=X)
    st.trace = []
    D_heads = []
    for t in range(n // b):
        oracle.partial_fit(st, pr, X[t * b:(t + 1) * b], np.arange(t * b, (t + 1) * b))
        D_heads.append(st.D[:, :64].copy())
    tr = st.trace
    assert_array_equal([len(t['subset']) for t in tr], g[name + 'subset_len'])
    assert_array_equal(np.stack([t['subset'][:64] for t in tr]), g[name + 'subset_head'])
    assert_array_equal([subset_checksum(t['subset']) for t in tr], g[name + 'subset_sum'])
    code_heads = np.stack([t['code'][:32] for t in tr])
    obs = headline_observables(st.D, st.C, st.B, st.comp_norm)
    if dn == 'f64':
        assert rel_fro(code_heads, g[name + 'code_head']) < 1e-9
        assert rel_fro(np.stack(D_heads), g[name + 'D_head']) < 1e-9
        for key, val in obs.items():
            if key == 'comp_norm':                           # leftover budgets: rounding residue around 0
                assert np.allclose(val, g[name + key], atol=1e-9)
            else:
                assert rel_fro(val, g[name + key]) < 1e-9, key
    else:
        ref64 = 'm1_r%d_f64/' % r
        assert rel_fro(code_heads[0], g[name + 'code_head'][0]) < 1e-5
        assert rel_fro(D_heads[0], g[name + 'D_head'][0]) < 1e-5
        for t in range(1, n // b):                           # within the reference's own f32 noise (+ margin)
            noise_c = rel_fro(g[name + 'code_head'][t], g[ref64 + 'code_head'][t])
            noise_D = rel_fro(g[name + 'D_head'][t], g[ref64 + 'D_head'][t])
            assert rel_fro(code_heads[t], g[ref64 + 'code_head'][t]) <= 2 * noise_c + 1e-5, t
            assert rel_fro(D_heads[t], g[ref64 + 'D_head'][t]) <= 2 * noise_D + 1e-5, t
    assert st.n_iter == int(g[name + 'n_iter'])
