import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # A/B runs of the whole suite under another mode of the blocked dictionary update (include/modl_hip.h, MODL_DEBUG_BCD_ACC)
    mode = os.environ.get('MODL_TEST_BCD_ACC')
    if mode is not None:
        from modl_amd._lib import lib, check, DEBUG_BCD_ACC
        check(lib.modl_debug_set(DEBUG_BCD_ACC, int(mode)))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


@pytest.fixture(scope='session')
def oracle():
    from oracle import somf_oracle
    somf_oracle.lib()
    return somf_oracle


def m1_rows(n, p, seed=1234, k0=256, density=0.1, noise=0.1):
    """Twin of tests/golden/make_golden.py::m1_rows: the float32 input of the headline fixture (M1 recipe,
    SURVEY 8d, numpy's legacy generator), regenerated instead of stored."""
    rs = np.random.RandomState(seed)
    Q = rs.randn(k0, p)
    Z = rs.randn(n, k0) * (rs.rand(n, k0) < density)
    X = Z.dot(Q) / np.sqrt(density * k0) + noise * rs.randn(n, p)
    return np.ascontiguousarray(X.astype(np.float32))


HEADLINE_KW = dict(n_components=256, batch_size=256, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
                   learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)


def headline_observables(D, C, B, comp_norm):
    """the aggregates tests/golden/traj_headline.npz keeps of a final state"""
    D64 = np.asarray(D, dtype=np.float64)
    return dict(D_rownorm2=np.sum(D64 ** 2, axis=1), D_colsum=np.sum(D64, axis=0), C_final=np.asarray(C),
                B_head=np.asarray(B)[:, :32], comp_norm=np.asarray(comp_norm))


def subset_checksum(subset):
    s = np.asarray(subset).astype(np.int64)
    return int(np.sum(s * np.arange(1, len(s) + 1)))


def assert_within_f32_noise(got32, ref32, ref64, what=''):
    """The f32 parity rule of this suite: float summation order is unpinned, so an f32 result is held to
    the REFERENCE ALGORITHM'S OWN f32 noise on the same input - `ref32` (the reference / the pinned oracle in f32)
    against `ref64` (the same in f64) - not to a flat tolerance: err(got32, ref64) <= 2 noise + 1e-5."""
    ref64 = np.asarray(ref64, dtype=np.float64)
    noise = rel_fro(np.asarray(ref32, dtype=np.float64), ref64)
    err = rel_fro(np.asarray(got32, dtype=np.float64), ref64)
    assert err <= 2 * noise + 1e-5, (what, err, noise)
    return err, noise
