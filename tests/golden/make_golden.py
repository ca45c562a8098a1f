"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, gcc, Cython):

    python tests/golden/make_golden.py

It copies the hot-path sources of the reference into a scratch directory
OUTSIDE the repository ($MODL_REF_SCRATCH or /tmp/modl_refbuild), compiles its
five Cython extensions there with plain setuptools, imports the resulting
package and records inputs/outputs as small .npz fixtures.  Nothing of the
reference (source, generated C, .so, bytecode) is written into the repository;
the fixtures are data only.  The numeric fixtures depend on the BLAS numpy and
scipy link in this container (float summation order is not part of the
reference's contract); integer draws do not.
"""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('MODL_REF', '/root/reference/modl')
SCRATCH = os.environ.get('MODL_REF_SCRATCH', '/tmp/modl_refbuild')


def build_reference():
    pkg = os.path.join(SCRATCH, 'modl')
    marker = os.path.join(SCRATCH, '.built')
    if os.path.exists(marker):
        return
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    for d in ('decomposition', 'utils/math', 'utils/randomkit', 'utils/recsys', 'feature_extraction', 'input_data'):
        os.makedirs(os.path.join(pkg, d))
    cp = lambda rel: shutil.copy(os.path.join(REF, rel), os.path.join(pkg, rel))
    for f in ('dict_fact.py', 'dict_fact_fast.pyx', 'recsys.py', 'recsys_fast.pyx'):
        cp('decomposition/' + f)
    cp('utils/__init__.py')
    for f in ('enet.pyx', 'enet.pxd'):
        cp('utils/math/' + f)
    for f in ('__init__.py', 'sampler.pyx', 'sampler.pxd', 'random_fast.pyx', 'random_fast.pxd',
              'randomkit.c', 'randomkit.h', 'distributions.c', 'distributions.h'):
        cp('utils/randomkit/' + f)
    cp('utils/recsys/cross_validation.py')
    for f in ('decomposition/image.py', 'feature_extraction/image.py', 'input_data/image.py', 'input_data/image_fast.pyx'):
        cp(f)
    # sklearn made extract_patches private; the scratch copy (never the repository) is pointed at it
    fe = os.path.join(pkg, 'feature_extraction', 'image.py')
    src = open(fe).read().replace('from sklearn.feature_extraction.image import extract_patches',
                                  'from sklearn.feature_extraction.image import _extract_patches as extract_patches')
    open(fe, 'w').write(src)
    for d in ('', 'decomposition', 'utils/math', 'utils/recsys', 'feature_extraction', 'input_data'):
        open(os.path.join(pkg, d, '__init__.py'), 'a').close()
    with open(os.path.join(SCRATCH, 'setup_ref.py'), 'w') as f:
        f.write(textwrap.dedent('''
            import numpy as np
            from setuptools import setup, Extension
            from Cython.Build import cythonize
            inc = [np.get_include(), 'modl/utils/randomkit']
            E = lambda name, srcs, **kw: Extension(name, srcs, include_dirs=inc, **kw)
            rk = 'modl/utils/randomkit/'
            exts = [
                E('modl.decomposition.dict_fact_fast', ['modl/decomposition/dict_fact_fast.pyx']),
                E('modl.decomposition.recsys_fast', ['modl/decomposition/recsys_fast.pyx']),
                E('modl.utils.math.enet', ['modl/utils/math/enet.pyx']),
                E('modl.input_data.image_fast', ['modl/input_data/image_fast.pyx']),
                E('modl.utils.randomkit.random_fast',
                  [rk + 'random_fast.pyx', rk + 'randomkit.c', rk + 'distributions.c'], language='c++'),
                E('modl.utils.randomkit.sampler', [rk + 'sampler.pyx'], language='c++'),
            ]
            setup(name='modl_ref', ext_modules=cythonize(
                exts, compiler_directives={'language_level': 3, 'legacy_implicit_noexcept': True}))
        '''))
    subprocess.check_call([sys.executable, 'setup_ref.py', 'build_ext', '--inplace'], cwd=SCRATCH,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    open(marker, 'w').close()


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------

def gen_rng():
    from modl.utils.randomkit import RandomState
    out = {}
    seeds = np.array([0, 1, 42, 2 ** 32 + 5, 2 ** 63 - 2], dtype=np.uint64)
    out['seeds'] = seeds
    highs = np.array([0, 1, 9, 10, 255, 256, 1000, 2 ** 31 - 1, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 3],
                     dtype=np.uint64)
    out['highs'] = highs
    rint = np.zeros((len(seeds), len(highs), 8), dtype=np.int64)
    for a, s in enumerate(seeds):
        rs = RandomState(seed=int(s))
        for b_, h in enumerate(highs):
            for c in range(8):
                rint[a, b_, c] = rs.randint(int(h))
    out['randint'] = rint
    # reference known answers (modl/utils/randomkit/tests/test_random.py:10-38)
    rs = RandomState(seed=0)
    out['ka_randint10'] = np.array([rs.randint(10) for _ in range(10000)], dtype=np.int64)
    out['ka_binomial'] = np.array([rs.binomial(1000, 0.8) for _ in range(10000)], dtype=np.int64)
    # binomial over both algorithms, both sides of p = .5, and parameter switches
    cases = [(100, 0.1), (100, 0.5), (10000, 0.1), (10000, 0.9), (500, 1 / 12.), (200000, 1 / 12.),
             (10, 0.05), (10000, 1.0), (64, 0.999), (20, 0.5), (3, 0.2), (10000, 0.001)]
    out['binom_n'] = np.array([c[0] for c in cases], dtype=np.int64)
    out['binom_p'] = np.array([c[1] for c in cases], dtype=np.float64)
    rs = RandomState(seed=7)
    draws = np.zeros((3, len(cases), 40), dtype=np.int64)
    for rep in range(3):                       # cases revisited -> exercises the cached set-up
        for i, (n, p) in enumerate(cases):
            for j in range(40):
                draws[rep, i, j] = rs.binomial(n, p)
    out['binom_draws'] = draws
    for n in (1, 2, 10, 257, 1000):
        rs = RandomState(seed=0)
        x = np.arange(n)
        rs.shuffle(x)
        out['shuffle_%d' % n] = x
        out['perm_%d' % n] = np.asarray(rs.permutation(n))
    rs = RandomState(seed=3)
    a = np.arange(12)
    b2 = np.arange(24, dtype=np.float64).reshape(12, 2)
    tr = rs.shuffle_with_trace([a, b2])
    out['trace_perm'] = tr
    out['trace_a'] = a
    out['trace_b'] = b2
    save('rng', **out)


def gen_sampler():
    from modl.utils.randomkit import Sampler
    out = {}
    cfg = []
    idx = 0
    for rng_ in (16, 100, 500):
        for rand_size in (True, False):
            for repl in (True, False):
                for red in (1, 2.5, 10, 12):
                    if red > rng_ / 2:
                        continue
                    smp = Sampler(rng_, rand_size, repl, 1234 + idx)
                    draws = [np.asarray(smp.yield_subset(red)).copy() for _ in range(14)]
                    out['draws_%d' % idx] = np.concatenate(draws).astype(np.int64)
                    out['lens_%d' % idx] = np.array([len(d) for d in draws], dtype=np.int64)
                    cfg.append((rng_, int(rand_size), int(repl), red, 1234 + idx))
                    idx += 1
    out['cfg'] = np.array(cfg, dtype=np.float64)
    # the reference's own known answers (modl/utils/randomkit/tests/test_sampler.py:6-29)
    s = Sampler(100, True, True, 0)
    out['ka_first'] = np.asarray(s.yield_subset(10)).astype(np.int64)
    out['ka_mean_len'] = np.array(np.mean([s.yield_subset(10).shape[0] for _ in range(100)]))
    s = Sampler(100, False, True, 0)
    out['ka_fixed_first'] = np.asarray(s.yield_subset(10)).astype(np.int64)
    # changing reduction between calls (the 'reducing ratio' schedules do that)
    s = Sampler(300, True, False, 99)
    seq = [3, 3, 2, 5, 1, 4, 4, 1.5]
    dr = [np.asarray(s.yield_subset(r)).copy() for r in seq]
    out['var_red'] = np.array(seq)
    out['var_draws'] = np.concatenate(dr).astype(np.int64)
    out['var_lens'] = np.array([len(d) for d in dr], dtype=np.int64)
    save('sampler', **out)


def gen_batch_weight():
    from modl.decomposition.dict_fact_fast import _batch_weight
    cases = []
    for b in (1, 10, 256):
        for lr in (1.0, 0.92, 0.76):
            for step in (1, 2, 3, 10, 1000):
                cases.append((step * b, b, lr, 0.0, _batch_weight(step * b, b, lr, 0.0)))
    cases.append((5, 10, 0.9, 0.0, _batch_weight(5, 10, 0.9, 0.0)))
    cases.append((100, 10, 0.9, 3.0, _batch_weight(100, 10, 0.9, 3.0)))
    save('batch_weight', cases=np.array(cases, dtype=np.float64))


def gen_enet():
    from modl.utils.math.enet import enet_norm, enet_projection, enet_scale
    out = {}
    rs = np.random.RandomState(0)
    i = 0
    meta = []
    for dt in (np.float32, np.float64):
        for n in (1, 7, 100, 400):
            for l1 in (0.0, 0.15, 0.5, 1.0):
                v = (rs.randn(n) * (1 + 3 * rs.rand())).astype(dt)
                if n == 7:
                    v[2] = 0
                    v[5] = v[1]        # ties and exact zeros
                nrm = enet_norm(v, l1)
                for radius in (0.0, 0.3, 1.0, 1e6):
                    o = np.zeros(n, dtype=dt)
                    enet_projection(v, o, radius, l1)
                    out['proj_%d' % i] = o
                    out['v_%d' % i] = v
                    vs = v.copy()
                    enet_scale(vs, l1, max(radius, 0.5))
                    out['scaled_%d' % i] = vs
                    meta.append((0 if dt is np.float32 else 1, n, l1, radius, nrm, max(radius, 0.5)))
                    i += 1
    out['meta'] = np.array(meta, dtype=np.float64)
    save('enet', **out)


def gen_cd():
    from modl.decomposition.dict_fact_fast import (_enet_regression_single_gram,
                                                   _enet_regression_multi_gram, _update_G_average)
    out = {}
    meta = []
    i = 0
    rs = np.random.RandomState(1)
    for dt in (np.float32, np.float64):
        for (k, p, b, n) in ((16, 40, 6, 9), (64, 100, 5, 5)):
            D = rs.randn(k, p).astype(dt)
            D /= np.sqrt((D ** 2).sum(1))[:, None]
            X = (rs.randn(b, k) * (rs.rand(b, k) < 0.3)).astype(dt).dot(D) + 0.05 * rs.randn(b, p).astype(dt)
            X = np.ascontiguousarray(X.astype(dt))
            G = np.ascontiguousarray(D.dot(D.T).astype(dt))
            Dx0 = np.ascontiguousarray(X.dot(D.T).astype(dt))
            Gm = np.ascontiguousarray(np.stack([G * (1 + 0.1 * j) for j in range(b)]).astype(dt))
            idx = rs.permutation(n)[:b].astype(np.int64)
            for l1 in (1.0, 0.5, 0.0):
                for alpha in (0.05, 0.5):
                    for pos in (False, True):
                        for tol, mi in ((1e-2, 100), (1e-8, 3)):
                            if l1 == 0.0 and (pos or mi == 3):
                                continue
                            code = np.ones((n, k), dtype=dt)
                            code[idx[0]] = 0.0                      # a zero warm start too
                            c_in = code.copy()
                            Dx = Dx0.copy()
                            _enet_regression_single_gram(G, Dx, X, code, idx, l1, alpha, pos, tol, mi)
                            out['single_code_%d' % i] = code
                            code2 = c_in.copy()
                            Dx = Dx0.copy()
                            _enet_regression_multi_gram(Gm.copy(), Dx, X, code2, idx, l1, alpha, pos, tol, mi)
                            out['multi_code_%d' % i] = code2
                            out['code_in_%d' % i] = c_in
                            meta.append((0 if dt is np.float32 else 1, k, p, b, n, l1, alpha, int(pos), tol, mi,
                                         len(out) and i))
                            out['idx_%d' % i] = idx
                            i += 1
            tag = '%s_%d' % ('f32' if dt is np.float32 else 'f64', k)
            out['G_' + tag] = G
            out['Gm_' + tag] = Gm
            out['Dx_' + tag] = Dx0
            out['X_' + tag] = X
            ws = rs.rand(b).astype(dt)
            Ga = Gm.copy()
            _update_G_average(Ga, G, ws)
            out['Gavg_w_' + tag] = ws
            out['Gavg_out_' + tag] = Ga
    out['meta'] = np.array(meta, dtype=np.float64)
    save('cd', **out)


def synth(n, p, k0, seed, dtype):
    """Recipe of modl/decomposition/tests/test_dict_fact.py:40-52."""
    rs = np.random.RandomState(seed)
    Q = rs.randn(k0, p)
    X = rs.randn(n, k0).dot(Q)
    return np.ascontiguousarray(X.astype(dtype))


def gen_traj():
    """Full-trajectory snapshots of the reference estimator on small problems
    (all aggregation modes, both optimizers, atom constraints), plus BASELINE
    config 1 (2000 x 500, k = 16, r = 1)."""
    from modl.decomposition.dict_fact import DictFact

    class Recorder(DictFact):
        def _single_batch_fit(self, X, sample_indices):
            smp = self.feature_sampler_
            rec = self._rec
            box = {}

            class Proxy:
                def yield_subset(_, r):
                    s = np.asarray(smp.yield_subset(r)).copy()
                    box['subset'] = s
                    return s
            self.feature_sampler_ = Proxy()
            try:
                DictFact._single_batch_fit(self, X, sample_indices)
            finally:
                self.feature_sampler_ = smp
            rec['subset'].append(box['subset'])
            rec['code'].append(self.code_[sample_indices].copy())
            if len(rec['subset']) in rec['snap_at']:
                rec['D'].append(self.components_.copy())
                rec['C'].append(self.C_.copy())
                rec['B'].append(self.B_.copy())

    def run(name, X, snap_at, **kw):
        est = Recorder(**kw)
        est._rec = dict(subset=[], code=[], D=[], C=[], B=[], snap_at=set(snap_at))
        est.fit(X)
        rec = est._rec
        out = dict(subset=np.concatenate(rec['subset']).astype(np.int64),
                   subset_len=np.array([len(s) for s in rec['subset']], dtype=np.int64),
                   code=np.concatenate(rec['code']),
                   code_len=np.array([len(c) for c in rec['code']], dtype=np.int64),
                   snap_at=np.array(sorted(snap_at)),
                   D_snap=np.stack(rec['D']), C_snap=np.stack(rec['C']), B_snap=np.stack(rec['B']),
                   D_final=est.components_, C_final=est.C_, B_final=est.B_, code_final=est.code_,
                   comp_norm=est.comp_norm_, n_iter=np.array(est.n_iter_))
        if hasattr(est, 'G_') and est.G_agg == 'full':
            out['G_final'] = est.G_
        return {name + '/' + k_: v for k_, v in out.items()}

    out = {}
    cases = []
    for dt, dn in ((np.float64, 'f64'), (np.float32, 'f32')):
        X = synth(120, 40, 6, 0, dt)
        for G_agg in ('masked', 'full', 'average'):
            for Dx_agg in ('masked', 'full', 'average'):
                name = 'agg_%s_%s_%s' % (G_agg, Dx_agg, dn)
                kw = dict(n_components=6, batch_size=10, reduction=2, n_epochs=2, random_state=0,
                          code_alpha=0.1, G_agg=G_agg, Dx_agg=Dx_agg, learning_rate=0.9)
                out.update(run(name, X, (1, 2, 10), **kw))
                cases.append(name)
        variants = {
            'ridge_l1atoms': dict(code_l1_ratio=0, comp_l1_ratio=1, code_alpha=0.01),
            'ridge_l1atoms_pos': dict(code_l1_ratio=0, comp_l1_ratio=1, comp_pos=True, code_alpha=0.01),
            'enet_atoms': dict(code_l1_ratio=0.5, comp_l1_ratio=0.5, code_alpha=0.05),
            'nmf': dict(comp_pos=True, code_pos=True, code_alpha=0.05),
            'sgd': dict(optimizer='sgd', step_size=0.1, code_alpha=0.05),
            'fixed_norepl': dict(rand_size=False, replacement=False, code_alpha=0.05),
            'r1': dict(reduction=1, code_alpha=0.05),
            'avg_ridge': dict(code_l1_ratio=0, comp_l1_ratio=1, code_alpha=0.01, G_agg='average',
                              Dx_agg='average'),
        }
        for vn, extra in variants.items():
            name = 'var_%s_%s' % (vn, dn)
            kw = dict(n_components=6, batch_size=10, reduction=2, n_epochs=2, random_state=0,
                      learning_rate=0.9)
            kw.update(extra)
            Xv = np.abs(X) if vn == 'nmf' else X
            out.update(run(name, Xv, (1, 2, 10), **kw))
            cases.append(name)
    out['cases'] = np.array(cases)
    save('traj_small', **out)

    # BASELINE config 1: only the dictionary / stats are kept
    X = synth(2000, 500, 16, 0, np.float64)
    r = run('c1', X, (1, 2, 10), n_components=16, reduction=1, random_state=0, n_epochs=1, code_alpha=1e-4)
    keep = {k_: v for k_, v in r.items() if k_.split('/')[1] in
            ('subset_len', 'snap_at', 'D_snap', 'D_final', 'C_final', 'comp_norm', 'n_iter')}
    keep['c1/subset_head'] = r['c1/subset'][:1500]
    keep['c1/code_final_head'] = r['c1/code_final'][:64]
    keep['c1/B_final_head'] = r['c1/B_final'][:, :32]
    save('traj_c1', **keep)


def gen_transform():
    from modl.decomposition.dict_fact import DictFact, Coder
    out = {}
    for dt, dn in ((np.float64, 'f64'), (np.float32, 'f32')):
        X = synth(60, 30, 5, 3, dt)
        rs = np.random.RandomState(5)
        D = rs.randn(5, 30).astype(dt)
        for l1, alpha, pos in ((1.0, 0.1, False), (0.0, 0.1, False), (0.5, 0.05, True)):
            cd = Coder(D, code_alpha=alpha, code_l1_ratio=l1, code_pos=pos)
            key = '%s_%g_%g_%d' % (dn, l1, alpha, pos)
            out['code_' + key] = cd.transform(X)
            out['score_' + key] = np.array(cd.score(X))
        out['X_' + dn] = X
        out['D_' + dn] = D
    save('transform', **out)


def gen_recsys():
    import scipy.sparse as sp
    from modl.decomposition.recsys import RecsysDictFact
    out = {}
    rs = np.random.RandomState(0)
    n, p, k = 40, 25, 4
    P = rs.randn(n, k)
    Qm = rs.randn(k, p)
    full = P.dot(Qm)
    mask = rs.rand(n, p) < 0.4
    # NB the reference's _refit divides by the row's nnz (recsys.py:260): no empty rows
    X = sp.csr_matrix(np.where(mask, full, 0.0))
    for detrend in (False, True):
        est = RecsysDictFact(n_components=k, alpha=0.1, beta=0.5, batch_size=5, n_epochs=2,
                             learning_rate=0.9, detrend=detrend, random_state=0)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            est.fit(X)
        tag = 'det%d' % int(detrend)
        out['D_' + tag] = est.components_
        out['code_' + tag] = est.code_
        out['C_' + tag] = est.C_
        out['B_' + tag] = est.B_
        out['comp_norm_' + tag] = est.comp_norm_
        out['pred_' + tag] = est.predict(X).data
        out['score_' + tag] = np.array(est.score(X))
        if detrend:
            out['row_mean'] = est.row_mean_
            out['col_mean'] = est.col_mean_
    out['X_data'] = X.data
    out['X_indices'] = X.indices
    out['X_indptr'] = X.indptr
    out['shape'] = np.array(X.shape)
    save('recsys', **out)


def synth_image(h, w, c, seed=0, holes=False):
    """Sum of 2-D sinusoids + noise in [0, 1]; optional missing (-1) pixels."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, c))
    for ch in range(c):
        for _ in range(4):
            fy, fx, ph = rs.uniform(0.05, 0.6), rs.uniform(0.05, 0.6), rs.uniform(0, 6.28)
            img[:, :, ch] += np.sin(fy * yy + fx * xx + ph)
    img += 0.05 * rs.randn(h, w, c)
    img = (img - img.min()) / (img.max() - img.min())
    if holes:
        img[rs.rand(h, w) < 0.01] = -1
    return img


def gen_image():
    import contextlib, io
    from modl.decomposition.image import ImageDictFact
    from modl.feature_extraction.image import LazyCleanPatchExtractor
    from modl.input_data.image import scale_patches
    out = {}
    cases = []
    for name, kw, (h, w, c), holes in (
            ('masked', dict(method='masked', n_epochs=2, reduction=2), (28, 30, 1), False),
            ('masked_rgb_holes', dict(method='masked', n_epochs=1, reduction=2), (24, 26, 3), True),
            ('average', dict(method='average', n_epochs=2, reduction=2), (26, 26, 1), False),
            ('dict_only', dict(method='dictionary only', n_epochs=1), (26, 26, 1), False),
            ('reducing', dict(method='reducing ratio', n_epochs=3, reduction=3), (26, 26, 1), False),
            ('sgd', dict(method='sgd', n_epochs=1, step_size=0.05), (26, 26, 1), False),
            ('nmf', dict(method='masked', setting='NMF', n_epochs=1, reduction=2), (26, 26, 1), False)):
        img = synth_image(h, w, c, seed=len(cases), holes=holes)
        est = ImageDictFact(patch_size=(6, 6), n_components=7, batch_size=25, alpha=0.05, random_state=0,
                            max_patches=300, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            est.fit(img)
        out[name + '/D'] = est.components_
        out[name + '/code'] = est.dict_fact_.code_
        out[name + '/n_iter'] = np.array(est.n_iter_)
        test = LazyCleanPatchExtractor(patch_size=(6, 6), max_patches=40, random_state=1).fit(img).transform()
        out[name + '/test_patches'] = test
        out[name + '/test_code'] = est.transform(test.copy())
        out[name + '/test_score'] = np.array(est.score(test.copy()))
        cases.append(name)
    img = synth_image(12, 13, 2, seed=9, holes=True)
    ex = LazyCleanPatchExtractor(patch_size=(3, 4), random_state=0).fit(img)
    out['extract/indices'] = ex.indices_3d
    out['extract/patches'] = ex.transform()
    out['extract/scaled'] = scale_patches(ex.transform(), with_mean=True, with_std=True, copy=True)
    out['cases'] = np.array(cases)
    save('image', **out)


def m1_rows(n, p, seed=1234, k0=256, density=0.1, noise=0.1):
    """Rows of the M1 recipe from numpy's legacy generator, float32: the headline fixture's input is
    regenerated by the tests from this same function (tests/conftest.py holds the twin), never stored."""
    rs = np.random.RandomState(seed)
    Q = rs.randn(k0, p)
    Z = rs.randn(n, k0) * (rs.rand(n, k0) < density)
    X = Z.dot(Q) / np.sqrt(density * k0) + noise * rs.randn(n, p)
    return np.ascontiguousarray(X.astype(np.float32))


def gen_headline():
    """The metric's shape from the REAL reference: k = 256, p = 10 000, b = 256, l1 codes, l2 atoms,
    reduction in {1, 10}, f32 and f64 (the f64 runs read the SAME float32 rows, cast), 4 minibatches through
    prepare + partial_fit (dict_fact.py:313-337).  Heads and aggregates only: D[:, :64] after every minibatch,
    the first 32 code rows of every minibatch, C_, B_[:, :32], the squared row norms and the column sums of the
    final dictionary (sensitive to every entry), the subset lengths and the first 64 indices of every subset."""
    from modl.decomposition.dict_fact import DictFact
    n, p, k, b = 1024, 10000, 256, 256
    X32 = m1_rows(n, p)
    out = {}

    class Recorder(DictFact):
        def _single_batch_fit(self, X, sample_indices):
            smp = self.feature_sampler_
            box = {}

            class Proxy:
                def yield_subset(_, r):
                    s_ = np.asarray(smp.yield_subset(r)).copy()
                    box['subset'] = s_
                    return s_
            self.feature_sampler_ = Proxy()
            try:
                DictFact._single_batch_fit(self, X, sample_indices)
            finally:
                self.feature_sampler_ = smp
            rec = self._rec
            rec['subset_len'].append(len(box['subset']))
            rec['subset_head'].append(box['subset'][:64].copy())
            rec['subset_sum'].append(int(np.sum(box['subset'].astype(np.int64) * np.arange(1, len(box['subset']) + 1))))
            rec['code_head'].append(self.code_[sample_indices[:32]].copy())
            rec['D_head'].append(self.components_[:, :64].copy())

    for dt, dn in ((np.float32, 'f32'), (np.float64, 'f64')):
        X = np.ascontiguousarray(X32.astype(dt))
        for r in (1, 10):
            est = Recorder(n_components=k, batch_size=b, reduction=r, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
                           learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
            est._rec = dict(subset_len=[], subset_head=[], subset_sum=[], code_head=[], D_head=[])
            est.prepare(n_samples=n, X=X)
            est.partial_fit(X)
            name = 'm1_r%d_%s/' % (r, dn)
            rec = est._rec
            out[name + 'subset_len'] = np.array(rec['subset_len'], dtype=np.int64)
            out[name + 'subset_head'] = np.stack(rec['subset_head']).astype(np.int64)
            out[name + 'subset_sum'] = np.array(rec['subset_sum'], dtype=np.int64)
            out[name + 'code_head'] = np.stack(rec['code_head'])
            out[name + 'D_head'] = np.stack(rec['D_head'])
            D = est.components_
            out[name + 'D_rownorm2'] = np.sum(D.astype(np.float64) ** 2, axis=1)
            out[name + 'D_colsum'] = np.sum(D.astype(np.float64), axis=0)
            out[name + 'C_final'] = est.C_.copy()
            out[name + 'B_head'] = est.B_[:, :32].copy()
            out[name + 'comp_norm'] = est.comp_norm_.copy()
            out[name + 'n_iter'] = np.array(est.n_iter_)
    out['shape'] = np.array([n, p, k, b])
    save('traj_headline', **out)


def gen_fmri():
    """The record loop of the reference's fMRI estimator (decomposition/fmri.py:423-556) from the REAL reference.
    fmri.py cannot be imported here (nilearn / nibabel are absent), so the two functions of the loop -
    `_compute_components` and `_flip` - are taken out of the reference's source AT RUN TIME (ast, from
    /root/reference; nothing of it is written anywhere) and executed against stand-ins for the masking / IO layer,
    which is out of scope: a masker whose `transform` returns the 2-D record it is given (the `MultiRawMasker`
    contract, input_data/fmri/unmask.py:37-55), `_lazy_scan` answering with the records' lengths and dtype, and a
    `check_niimg` stand-in that only serves `n_voxels`.  Everything numerical is the reference's own code, with its
    own DictFact.  NB the loop rebinds `method` to the dict of aggregation modes (:460), so its three later tests on
    the method NAME (:508, :511, :535) never fire: no switch at epoch 5, no shrinking reduction, and
    sample_indices is always None - the fixture records that EFFECTIVE behaviour."""
    import ast
    import contextlib
    import io
    import itertools
    import time
    from math import log, sqrt
    from sklearn.utils import check_random_state
    from modl.decomposition.dict_fact import DictFact
    src = open(os.path.join(REF, 'decomposition', 'fmri.py')).read()
    tree = ast.parse(src)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('_compute_components', '_flip')]
    assert len(wanted) == 2
    mod = ast.Module(body=wanted, type_ignores=[])

    class _Mask:
        def __init__(self, p):
            self.p = p

        def get_data(self):
            return np.ones(self.p)

    class _Masker:
        def __init__(self, p):
            self.mask_img_ = _Mask(p)

        def _check_fitted(self):
            pass

        def transform(self, img, confounds=None):
            return img

    ns = dict(np=np, itertools=itertools, time=time, sqrt=sqrt, log=log, check_random_state=check_random_state,
              DictFact=DictFact, check_niimg=lambda m: m,
              _check_dict_init=lambda dict_init, mask_img=None, n_components=None: dict_init,
              _lazy_scan=lambda imgs: ([im.shape[0] for im in imgs], imgs[0].dtype))
    exec(compile(mod, '<reference fmri.py, extracted>', 'exec'), ns)
    compute = ns['_compute_components']

    out = {}
    cases = []
    rs = np.random.RandomState(0)
    p, k, n_rec, T = 120, 4, 5, 30
    maps = np.zeros((k, p))
    for j in range(k):
        maps[j, j * 30:(j + 1) * 30] = 1.0
    for dt, dn in ((np.float64, 'f64'), (np.float32, 'f32')):
        records = []
        for r_ in range(n_rec):
            L = rs.randn(T, k)
            rec = L.dot(maps) + 0.01 * rs.randn(T, p)
            rec -= rec.mean(1, keepdims=True)
            rec /= rec.std(1, keepdims=True)
            records.append(np.ascontiguousarray(rec.astype(dt)))
        dict_init = (maps + rs.randn(k, p)).astype(dt)
        out['records_' + dn] = np.stack(records)
        out['dict_init_' + dn] = dict_init
        for method, extra in (('masked', dict(n_epochs=2, reduction=3)), ('average', dict(n_epochs=2, reduction=3)),
                              ('gram', dict(n_epochs=7, reduction=3)), ('reducing ratio', dict(n_epochs=3, reduction=4)),
                              ('dictionary only', dict(n_epochs=1)), ('sgd', dict(n_epochs=1, step_size=0.05)),
                              ('masked_pos', dict(n_epochs=2, reduction=3, positive=True))):
            kw = dict(alpha=1e-3, reduction=1, learning_rate=0.92, n_components=k, batch_size=10, random_state=0)
            kw.update(extra)
            m = method.split('_')[0]
            with contextlib.redirect_stdout(io.StringIO()):
                comp = compute(_Masker(p), records, dict_init=dict_init.copy(), method=m, **kw)
            name = '%s_%s' % (method.replace(' ', '-'), dn)
            out[name + '/components'] = comp
            cases.append(name)
    out['cases'] = np.array(cases)
    save('fmri', **out)


if __name__ == '__main__':
    build_reference()
    sys.path.insert(0, SCRATCH)
    gen_rng()
    gen_sampler()
    gen_batch_weight()
    gen_enet()
    gen_cd()
    gen_transform()
    gen_traj()
    gen_recsys()
    gen_image()
    gen_headline()
    gen_fmri()
