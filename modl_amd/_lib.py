"""ctypes binding of libmodl_hip.so (the C-ABI of include/modl_hip.h).

This module takes the place of the reference's compiled extension imports
(`from .dict_fact_fast import ...`, modl/decomposition/dict_fact.py:14-18).
There is no CPU fallback: if the shared library is missing the import fails
loudly with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmodl_hip.so')

MODL_F32, MODL_F64 = 0, 1
FLAG_NO_RIDER, FLAG_GEMM_STAMPS = 1, 2          # modl_somf_desc.flags (diagnostics)
DEBUG_CD_SPARSE_PCT = 1                         # modl_debug_set
DEBUG_CD_SPLIT = 2
DEBUG_BCD_ACC = 5
DEBUG_ATOM_STAMPS = 6
DEBUG_BCD_TINY = 7
DEBUG_STAGE_AHEAD = 8
DEBUG_BCD_PERSIST = 9
DEBUG_STATS_RESIDENT = 10
DEBUG_RECSYS_FUSED = 11
DEBUG_ATOM_MWG = 12
DEBUG_BCD_FEW = 13
DEBUG_ATOM_PIPE = 14
AGG = {'masked': 0, 'full': 1, 'average': 2}
OPT = {'variational': 0, 'sgd': 1}


class ModlError(RuntimeError):
    pass


class SomfDesc(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('k', C.c_int32), ('p', C.c_int64), ('n_samples', C.c_int64),
                ('G_agg', C.c_int32), ('Dx_agg', C.c_int32), ('optimizer', C.c_int32), ('code_pos', C.c_int32),
                ('comp_pos', C.c_int32), ('max_iter', C.c_int32), ('code_alpha', C.c_double),
                ('code_l1_ratio', C.c_double), ('comp_l1_ratio', C.c_double), ('tol', C.c_double),
                ('step_size', C.c_double), ('max_batch', C.c_int32), ('flags', C.c_int32)]


class SomfState(C.Structure):
    _fields_ = [('d_Dt', C.c_void_p), ('d_Bt', C.c_void_p), ('d_C', C.c_void_p), ('d_code', C.c_void_p),
                ('d_comp_norm', C.c_void_p), ('d_G', C.c_void_p), ('d_Dx_average', C.c_void_p),
                ('d_G_average', C.c_void_p)]


class SomfBatch(C.Structure):
    _fields_ = [('d_X', C.c_void_p), ('ldx', C.c_int64), ('b', C.c_int32), ('s', C.c_int32),
                ('h_sample_idx', C.c_void_p), ('h_subset', C.c_void_p), ('h_order', C.c_void_p),
                ('h_w_sample', C.c_void_p), ('w', C.c_double), ('reduction', C.c_double),
                ('b_global', C.c_int64)]


class ProfEntry(C.Structure):
    _fields_ = [('name', C.c_char_p), ('ms_total', C.c_double), ('launches', C.c_int64), ('calls', C.c_int64)]


def _load():
    # MODL_AMD_DIAG=1 (scripts/diag_*.py that read in-kernel stamps of a whole estimator run): the diagnostics build of the
    # same sources takes the product library's place for this process
    if os.environ.get('MODL_AMD_DIAG') == '1':
        path = os.path.join(_HERE, 'libmodl_hip_diag.so')
        if not os.path.exists(path):
            raise ImportError('modl_amd: MODL_AMD_DIAG=1 but %s is missing (make -C modl_amd/csrc)' % path)
        return C.CDLL(path)
    # MODL_HIP_LIBRARY=<path>: an A/B build of the same sources (scripts/build_variant.sh -> build_ab/<name>/libmodl_hip.so)
    # takes the product library's place for THIS process; the product library on disk is never overwritten
    override = os.environ.get('MODL_HIP_LIBRARY')
    if override:
        if not os.path.exists(override):
            raise ImportError('modl_amd: MODL_HIP_LIBRARY=%s does not exist' % override)
        return C.CDLL(os.path.abspath(override))
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'modl_amd: %s is missing. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C modl_amd/csrc` (needs hipcc, --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    try:
        return C.CDLL(LIB_PATH)
    except OSError as e:                                   # pragma: no cover
        raise ImportError('modl_amd: cannot load %s: %s' % (LIB_PATH, e))


_vp, _i32, _i64, _u64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_size_t
_P = C.POINTER


def bind(lib):
    """declares the argument / result types of every entry point on a loaded library (the product library, or the
    diagnostics build of the same sources, load_diag())"""
    def _sig(name, restype, *argtypes):
        f = getattr(lib, name)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f

    _sig('modl_abi_version', C.c_int)
    _sig('modl_device_count', C.c_int)
    _sig('modl_error_string', C.c_char_p, C.c_int)
    _sig('modl_debug_set', C.c_int, C.c_int, _i64)
    _sig('modl_is_diag_build', C.c_int)
    _sig('modl_rk_create', C.c_int, _u64, _P(_vp))
    _sig('modl_rk_destroy', None, _vp)
    _sig('modl_rk_seed', C.c_int, _vp, _u64)
    _sig('modl_rk_random', C.c_int, _vp, _P(C.c_uint32))
    _sig('modl_rk_randint', C.c_int, _vp, _u64, _P(_i64))
    _sig('modl_rk_double', C.c_int, _vp, _P(_f64))
    _sig('modl_rk_binomial', C.c_int, _vp, _i64, _f64, _P(_i64))
    _sig('modl_rk_permutation', C.c_int, _vp, _i64, _vp)
    _sig('modl_rk_get_mt_state', C.c_int, _vp, _vp, _P(C.c_int32))
    _sig('modl_rk_set_mt_state', C.c_int, _vp, _vp, C.c_int32)
    _sig('modl_rk_shuffle_i64', C.c_int, _vp, _vp, _i64)
    _sig('modl_rk_shuffle_trace', C.c_int, _vp, _i64, _vp, _vp)
    _sig('modl_apply_swaps_rows', C.c_int, _vp, _i64, _sz, _vp)
    _sig('modl_apply_swaps_rows_device', C.c_int, _vp, _i64, _sz, _vp, _vp)
    _sig('modl_sampler_create', C.c_int, _i64, C.c_int, C.c_int, _u64, _P(_vp))
    _sig('modl_sampler_destroy', None, _vp)
    _sig('modl_sampler_yield_subset', C.c_int, _vp, _f64, _vp, _P(_i64))
    _sig('modl_sampler_get', C.c_int, _vp, _P(_i64), _P(_i64), _P(_i64), _vp)
    _sig('modl_sampler_state_bytes', _sz, _vp)
    _sig('modl_sampler_get_state', C.c_int, _vp, _vp, _sz)
    _sig('modl_sampler_set_state', C.c_int, _vp, _vp, _sz)
    _sig('modl_batch_weight', C.c_int, _i64, _i64, _f64, _f64, _P(_f64))
    _sig('modl_enet_regression_workspace', _sz, C.c_int, _i64, _i64, C.c_int)
    for _sfx, _ct in (('f32', C.c_float), ('f64', C.c_double)):
        for _kind in ('single', 'multi'):
            _sig('modl_enet_regression_%s_gram_%s' % (_kind, _sfx), C.c_int, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64,
                 _i64, _ct, _ct, C.c_int, _ct, C.c_int, _vp, _vp, _sz, _vp)
        _sig('modl_update_G_average_' + _sfx, C.c_int, _vp, _vp, _vp, _i64, _i64, _vp)
        _sig('modl_enet_norm_' + _sfx, C.c_int, _vp, _i64, _i64, _i64, _i64, _ct, _vp, _vp)
        _sig('modl_enet_projection_' + _sfx, C.c_int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _ct, _vp)
        _sig('modl_enet_scale_' + _sfx, C.c_int, _vp, _i64, _i64, _i64, _i64, _ct, _ct, _vp)
        _sig('modl_transpose_' + _sfx, C.c_int, _vp, _vp, _i64, _i64, _vp)
        _sig('modl_gather_rows_' + _sfx, C.c_int, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp)
    for _sfx, _ct in (('f32', C.c_float), ('f64', C.c_double)):
        _sig('modl_recsys_codes_' + _sfx, C.c_int, _vp, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp)
        _sig('modl_recsys_update_B_' + _sfx, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _i64, _vp)
        _sig('modl_recsys_minibatch_' + _sfx, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _f64, _f64, _f64,
             _vp, _vp, _vp, _vp, _vp, _vp, _vp)
        _sig('modl_recsys_fit_batches_' + _sfx, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _f64, _f64,
             _P(_i64), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _P(_i64))
        _sig('modl_recsys_predict_' + _sfx, C.c_int, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp, _vp)
        _sig('modl_gram_axpby_' + _sfx, C.c_int, _vp, _i64, C.c_int, _vp, _ct, _ct, _vp)
        _sig('modl_dict_update_' + _sfx, C.c_int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int, C.c_int, C.c_int, _f64,
             _f64, _f64, _vp, _sz, _vp)
        _sig('modl_image_clean_mask_' + _sfx, C.c_int, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _P(_i64))
        _sig('modl_image_patches_' + _sfx, C.c_int, _vp, _i64, _i64, _i64, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int,
             C.c_int, _vp, _i64, _vp)
        _sig('modl_objective_' + _sfx, C.c_int, _vp, _i64, _i64, _i64, _vp, C.c_int, _vp, _vp, _sz, _vp, _vp)
    _sig('modl_image_fill', C.c_int, _i64, _i64, _i64, _vp)
    _sig('modl_objective_workspace', _sz, C.c_int, _i64, _i64)
    _sig('modl_dict_update_workspace', _sz, C.c_int, _i64, C.c_int)
    _sig('modl_recsys_plan_create', C.c_int, C.c_int, _i64, C.c_int, _i64, _i64, _P(_vp))
    _sig('modl_recsys_plan_destroy', None, _vp)
    _sig('modl_recsys_plan_counts', C.c_int, _vp, _P(_i64), _P(_i64))
    _sig('modl_recsys_plan_wait_ms', C.c_int, _vp, _P(_f64))
    _sig('modl_recsys_plan_status', C.c_int, _vp, _vp)
    _sig('modl_recsys_plan_stamps', C.c_int, _vp, C.c_int, _vp)
    _sig('modl_predict_csr', C.c_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp)
    _sig('modl_somf_plan_create', C.c_int, _P(SomfDesc), _P(_vp))
    _sig('modl_somf_plan_destroy', None, _vp)
    _sig('modl_somf_plan_update', C.c_int, _vp, _P(SomfDesc))
    _sig('modl_somf_delta_elems', _i64, _P(SomfDesc))
    _sig('modl_somf_code_and_partials', C.c_int, _vp, _P(SomfState), _P(SomfBatch), _vp, _vp)
    _sig('modl_somf_apply_and_update_dict', C.c_int, _vp, _P(SomfState), _P(SomfBatch), _vp, _vp)
    _sig('modl_somf_step', C.c_int, _vp, _P(SomfState), _P(SomfBatch), _vp)
    _sig('modl_somf_partial_fit_chunk', C.c_int, _vp, _P(SomfState), _vp, _i64, _i64, C.c_int32, _vp, _vp, _vp, _P(_i64), _f64,
         _f64, _vp, _vp, _P(_i64), _vp)
    _sig('modl_somf_head_elems', C.c_int, _vp, _P(_i64))
    _sig('modl_comm_unique_id', C.c_int, _vp)
    _sig('modl_comm_create', C.c_int, _vp, C.c_int, C.c_int, _P(_vp))
    _sig('modl_comm_destroy', None, _vp)
    _sig('modl_comm_all_reduce_sum', C.c_int, _vp, _vp, _i64, C.c_int, _vp)
    _sig('modl_comm_wait', C.c_int, _vp, _vp, C.c_double)
    _sig('modl_comm_abort', C.c_int, _vp)
    _sig('modl_somf_step_dist', C.c_int, _vp, _P(SomfState), _P(SomfBatch), _vp, _vp)
    _sig('modl_somf_full_gram', C.c_int, _vp, _vp, _vp, _vp)
    _sig('modl_somf_transform', C.c_int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp)
    _sig('modl_somf_debug_stamps', C.c_int, _vp, _vp)
    _sig('modl_somf_status', C.c_int, _vp, _vp)
    _sig('modl_somf_persist_recoveries', C.c_int, _vp, _P(_i64))
    _sig('modl_somf_debug_persist_stamps', C.c_int, _vp, _vp)
    _sig('modl_somf_debug_gemm_stamps', C.c_int, _vp, _vp)
    _sig('modl_somf_last_sweeps', C.c_int, _vp, _vp, C.c_int, _P(C.c_int), _vp)
    _sig('modl_somf_sweeps_history', C.c_int, _vp, _vp, _i64)
    _sig('modl_somf_prof_enable', C.c_int, _vp, C.c_int)
    _sig('modl_somf_prof_stride', C.c_int, _vp, C.c_int)
    _sig('modl_somf_host_wait_ms', C.c_int, _vp, _P(_f64), C.c_int)
    _sig('modl_somf_prof_get', C.c_int, _vp, _P(ProfEntry), C.c_int, _P(C.c_int))
    _sig('modl_somf_prof_reset', C.c_int, _vp)

    return lib


lib = bind(_load())


def load_diag():
    """libmodl_hip_diag.so: the product's sources built with -DMODL_DIAG - adds the kernel variants that only exist to be
    compared with (the one-wavefront coordinate-descent kernel for k > 256, which spills) and the shader-clock stamp
    switches.  Tests and scripts/ load it next to the product library; nothing in modl_amd uses it."""
    path = os.path.join(_HERE, 'libmodl_hip_diag.so')
    if not os.path.exists(path):
        raise ImportError('modl_amd: %s is missing (make -C modl_amd/csrc)' % path)
    return bind(C.CDLL(path))

# every symbol declared in include/modl_hip.h (checked by tests/test_abi.py)
DECLARED = [n for n in dir(lib) if n.startswith('modl_')]


def error_string(rc):
    return lib.modl_error_string(rc).decode()


def check(rc, what=''):
    if rc != 0:
        msg = lib.modl_error_string(rc).decode()
        raise ModlError('%s failed: %s (code %d)' % (what or 'libmodl_hip call', msg, rc))


def require_gpu():
    if lib.modl_device_count() <= 0:
        raise ModlError('modl_amd needs an AMD GPU (gfx950); no HIP device is visible and there is no CPU fallback')
