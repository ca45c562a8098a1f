"""CPU-only checks of the C-ABI boundary: the library loads without a GPU,
exports every symbol include/modl_hip.h declares, and its host-side entry
points (RNG, sampler, batch weight) reproduce the reference's draws bit for bit."""
import ctypes as C
import os
import pickle
import re

import numpy as np
import pytest
from numpy.testing import assert_array_equal

from .conftest import load_golden, ROOT


@pytest.fixture(scope='module')
def lib():
    from modl_amd import _lib
    return _lib.lib


def test_library_exports_header(lib):
    hdr = open(os.path.join(ROOT, 'include', 'modl_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(modl_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.modl_abi_version() == 5
    assert lib.modl_device_count() >= 0
    assert lib.modl_error_string(-1) == b'invalid argument'


def test_no_gpu_is_reported_not_faked(lib):
    from modl_amd import _lib
    if lib.modl_device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(_lib.ModlError):
        _lib.require_gpu()
    d = _lib.SomfDesc()
    d.dtype, d.k, d.p, d.n_samples, d.max_batch, d.max_iter = 0, 4, 8, 10, 2, 10
    d.code_l1_ratio, d.code_alpha = 1.0, 1.0
    h = C.c_void_p()
    assert lib.modl_somf_plan_create(C.byref(d), C.byref(h)) == -4      # MODL_ENOGPU


def test_argument_validation(lib):
    assert lib.modl_rk_create(0, None) == -1
    assert lib.modl_sampler_create(-1, 1, 1, 0, C.byref(C.c_void_p())) == -1
    out = C.c_double()
    assert lib.modl_batch_weight(10, 10, 0.9, 0.0, C.byref(out)) == 0 and out.value == 1.0


def test_rng_known_answers_through_abi():
    # modl/utils/randomkit/tests/test_random.py:10-47
    from modl_amd.randomkit import RandomState
    rs = RandomState(seed=0)
    assert abs(np.mean([rs.randint(10) for _ in range(10000)]) - 5.018) < 1e-12
    assert abs(np.mean([rs.binomial(1000, 0.8) for _ in range(10000)]) - 799.8564) < 1e-9
    ind = np.arange(10)
    rs = RandomState(seed=0)
    rs.shuffle(ind)
    assert_array_equal(ind, [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    ind, ind2 = np.arange(10), np.arange(9, -1, -1)
    rs = RandomState(seed=0)
    perm = rs.shuffle_with_trace([ind, ind2])
    assert_array_equal(ind, [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    assert_array_equal(ind2, [7, 1, 5, 0, 8, 3, 2, 6, 9, 4])
    assert_array_equal(ind, perm)
    rs = RandomState(seed=0)
    assert_array_equal(rs.permutation(10), [2, 8, 4, 9, 1, 6, 7, 3, 0, 5])
    rs = RandomState(seed=0)
    a = rs.randint(5)
    assert pickle.loads(pickle.dumps(rs)).randint(5) == a


def test_rng_golden_through_abi():
    from modl_amd.randomkit import RandomState
    g = load_golden('rng')
    for a, s in enumerate(g['seeds']):
        rs = RandomState(int(s))
        for b, h in enumerate(g['highs']):
            assert_array_equal([rs.randint(int(h)) for _ in range(8)], g['randint'][a, b])
    rs = RandomState(7)
    for rep in range(3):
        for i, (n, p) in enumerate(zip(g['binom_n'], g['binom_p'])):
            assert_array_equal([rs.binomial(int(n), float(p)) for _ in range(40)], g['binom_draws'][rep, i])
    for n in (1, 2, 10, 257, 1000):
        rs = RandomState(0)
        x = np.arange(n)
        rs.shuffle(x)
        assert_array_equal(x, g['shuffle_%d' % n])
        assert_array_equal(rs.permutation(n), g['perm_%d' % n])
    rs = RandomState(3)
    a, b2 = np.arange(12), np.arange(24, dtype=np.float64).reshape(12, 2)
    tr = rs.shuffle_with_trace([a, b2])
    assert_array_equal(tr, g['trace_perm'])
    assert_array_equal(a, g['trace_a'])
    assert_array_equal(b2, g['trace_b'])


def test_sampler_through_abi():
    from modl_amd.randomkit import Sampler
    # modl/utils/randomkit/tests/test_sampler.py:6-45
    s = Sampler(100, rand_size=True, replacement=True, random_seed=0)
    assert_array_equal(s.yield_subset(10), [14, 58, 11, 49, 36, 62, 87, 45, 72, 47, 48, 13, 98, 97, 25, 93])
    assert np.mean([s.yield_subset(10).shape[0] for _ in range(100)]) == 10.19
    s = Sampler(100, rand_size=False, replacement=False, random_seed=0)
    assert_array_equal(np.sort(np.concatenate([s.yield_subset(10) for _ in range(10)])), np.arange(100))
    s = Sampler(100, rand_size=False, replacement=True, random_seed=0)
    assert_array_equal(s.yield_subset(10), [6, 55, 1, 25, 87, 49, 69, 63, 13, 8])
    s = Sampler(100, rand_size=True, replacement=False, random_seed=0)
    A = np.concatenate([s.yield_subset(10) for _ in range(20)])
    assert_array_equal(np.sort(A[:100]), np.arange(100))
    g = load_golden('sampler')
    for i, (rng_, rand_size, repl, red, seed) in enumerate(g['cfg']):
        s = Sampler(int(rng_), bool(rand_size), bool(repl), int(seed))
        lens = g['lens_%d' % i]
        draws = [s.yield_subset(red) for _ in range(len(lens))]
        assert_array_equal([len(d) for d in draws], lens)
        assert_array_equal(np.concatenate(draws), g['draws_%d' % i])
    s = Sampler(300, True, False, 99)
    dr = [s.yield_subset(r) for r in g['var_red']]
    assert_array_equal(np.concatenate(dr), g['var_draws'])


def test_sampler_state_roundtrip():
    from modl_amd.randomkit import Sampler
    s = Sampler(500, True, False, 5)
    for _ in range(3):
        s.yield_subset(7)
    s2 = pickle.loads(pickle.dumps(s))
    for _ in range(10):
        assert_array_equal(s.yield_subset(7), s2.yield_subset(7))
    assert s.lim_sup == s2.lim_sup and np.array_equal(s.box, s2.box)


def test_batch_weight_golden():
    from modl_amd.randomkit import batch_weight
    g = load_golden('batch_weight')
    for count, b, lr, off, w in g['cases']:
        got = batch_weight(int(count), int(b), lr, off)
        assert got == w or (np.isnan(got) and np.isnan(w))


def test_get_sub_slice():
    # modl/utils/tests/test_utils.py:8-14
    from modl_amd.utils import get_sub_slice
    a, b = slice(10, 100), slice(20, 30)
    assert_array_equal(get_sub_slice(a, b), np.arange(30, 40))
    assert_array_equal(get_sub_slice(None, b), np.arange(20, 30))
    assert_array_equal(get_sub_slice(np.arange(10, 100), b), np.arange(30, 40))


def test_new_entry_points_validate_arguments_without_gpu(lib):
    """host-only image helpers work on the CPU; device plans report the missing GPU instead of faking one"""
    out = np.empty((6, 3), dtype=np.int64)
    assert lib.modl_image_fill(1, 2, 3, out.ctypes.data_as(C.c_void_p)) == 0
    assert_array_equal(out, np.c_[np.where(np.ones((1, 2, 3)))])
    assert lib.modl_image_fill(-1, 2, 3, out.ctypes.data_as(C.c_void_p)) == -1
    img = np.zeros((4, 4, 1))
    n = C.c_int64()
    assert lib.modl_image_clean_mask_f64(img.ctypes.data_as(C.c_void_p), 4, 4, 1, 5, 2, 1, None, C.byref(n)) == -1   # window > image
    assert lib.modl_image_clean_mask_f64(img.ctypes.data_as(C.c_void_p), 4, 4, 1, 2, 2, 1, None, C.byref(n)) == 0 and n.value == 9
    assert lib.modl_objective_workspace(0, 10, 100) >= 10 * 100 * 4
    h = C.c_void_p()
    assert lib.modl_recsys_plan_create(7, 10, 4, 2, 8, C.byref(h)) == -1                    # bad dtype
    if lib.modl_device_count() == 0:
        assert lib.modl_recsys_plan_create(1, 10, 4, 2, 8, C.byref(h)) == -4                # MODL_ENOGPU


def test_rk_continues_numpy_legacy_stream():
    """modl_rk_set_mt_state / get_mt_state: a modl_rk loaded with numpy's legacy MT19937 state draws numpy's
    RandomState.permutation(k) bit for bit (legacy shuffle == randomkit's masked-rejection shuffle) and hands back a
    state from which numpy continues in step - how the atom order of dict_fact.py:672 is drawn behind the ABI."""
    import ctypes as C
    from modl_amd._lib import lib, check
    rs, ref = np.random.RandomState(11), np.random.RandomState(11)
    rs.randn(5), ref.randn(5)
    h = C.c_void_p()
    check(lib.modl_rk_create(0, C.byref(h)))
    try:
        for k in (256, 70, 1, 1000):
            kind, key, pos, hg, cg = rs.get_state()
            key = np.ascontiguousarray(key, dtype=np.uint32)
            check(lib.modl_rk_set_mt_state(h, key.ctypes.data_as(C.c_void_p), int(pos)))
            for _ in range(40):
                o = np.empty(k, dtype=np.int64)
                check(lib.modl_rk_permutation(h, k, o.ctypes.data_as(C.c_void_p)))
                assert np.array_equal(o, ref.permutation(k))
            k2, p2 = np.empty(624, dtype=np.uint32), C.c_int32()
            check(lib.modl_rk_get_mt_state(h, k2.ctypes.data_as(C.c_void_p), C.byref(p2)))
            rs.set_state((kind, k2, int(p2.value), hg, cg))
            assert rs.randint(1 << 30) == ref.randint(1 << 30)
        assert lib.modl_rk_set_mt_state(h, k2.ctypes.data_as(C.c_void_p), 625) != 0      # pos out of range
    finally:
        lib.modl_rk_destroy(h)


def test_debug_switch_numbers_match_the_header(lib):
    """The diagnostics switches of modl_debug_set: the numbers the Python side uses are the header's, they are distinct,
    and the library accepts each of them (restoring the default) and rejects an unknown one - no GPU involved."""
    import re
    from modl_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'modl_hip.h')).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define\s+MODL_(DEBUG_\w+)\s+(\d+)', hdr)}
    assert len(set(defs.values())) == len(defs)
    for name, value in defs.items():
        if hasattr(_lib, name):
            assert getattr(_lib, name) == value, name
    defaults = {'DEBUG_CD_SPARSE_PCT': -1, 'DEBUG_CD_SPLIT': 1, 'DEBUG_BCD_ACC': 1, 'DEBUG_BCD_TINY': 1, 'DEBUG_STAGE_AHEAD': 1, 'DEBUG_BCD_PERSIST': 1, 'DEBUG_STATS_RESIDENT': 1, 'DEBUG_RECSYS_FUSED': 1, 'DEBUG_ATOM_MWG': 1, 'DEBUG_BCD_FEW': 1, 'DEBUG_ATOM_PIPE': 1,
                'DEBUG_CD_STAMPS': 0, 'DEBUG_ATOM_STAMPS': 0}
    assert set(defaults) == set(defs)
    # the product library only has switches between CORRECT code paths: the stamp switches (they write to a
    # caller-supplied device buffer) and round 3's timing-experiment switch (4: wrong results) are rejected
    diag_only = {'DEBUG_CD_STAMPS', 'DEBUG_ATOM_STAMPS'}
    assert lib.modl_is_diag_build() == 0
    for name, dflt in defaults.items():
        assert (lib.modl_debug_set(defs[name], dflt) == 0) == (name not in diag_only), name
    assert lib.modl_debug_set(4, 1) != 0
    dlib = _lib.load_diag()
    assert dlib.modl_is_diag_build() == 1
    for name, dflt in defaults.items():
        assert dlib.modl_debug_set(defs[name], dflt) == 0, name
    assert dlib.modl_debug_set(4, 1) != 0
    assert lib.modl_debug_set(max(defs.values()) + 1, 0) != 0


def test_product_library_has_no_register_spills():
    """Every kernel of the product library fits its registers: no scratch instruction in any gfx950 code object (the
    variants that spill - kept only to be compared with - live in libmodl_hip_diag.so)."""
    import importlib.util
    import shutil
    root = os.path.join(os.path.dirname(__file__), '..')
    spec = importlib.util.spec_from_file_location('count_scratch', os.path.join(root, 'scripts', 'count_scratch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP) or shutil.which('objcopy') is None:
        pytest.skip('llvm-objdump / objcopy not available')
    nco, scratch = mod.count(os.path.join(root, 'modl_amd', 'libmodl_hip.so'))
    assert nco >= 8 and scratch == 0, (nco, scratch)
