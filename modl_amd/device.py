"""Device plumbing: torch is used for device memory, streams and (optionally)
torch.distributed only; all arithmetic runs in libmodl_hip.so."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, require_gpu, MODL_F32, MODL_F64

_TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def dtype_id(np_dtype):
    np_dtype = np.dtype(np_dtype)
    if np_dtype == np.float32:
        return MODL_F32
    if np_dtype == np.float64:
        return MODL_F64
    raise TypeError('float32 or float64 expected, got %s' % np_dtype)


def sfx(np_dtype):
    return 'f32' if dtype_id(np_dtype) == MODL_F32 else 'f64'


def torch_dtype(np_dtype):
    return _TORCH_DT[np.dtype(np_dtype)]


def np_dtype_of(t):
    return np.dtype(np.float32) if t.dtype == torch.float32 else np.dtype(np.float64)


def default_device():
    require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def to_device(a, device, dtype=None):
    """numpy array / torch tensor -> contiguous device tensor (no copy if already there)."""
    if isinstance(a, torch.Tensor):
        t = a
        if dtype is not None and t.dtype != torch_dtype(dtype):
            t = t.to(torch_dtype(dtype))
        if t.device != device:
            t = t.to(device)
        return t.contiguous()
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).to(device)


def transpose_to(src, rows, cols):
    """out[c][r] = src[r][c] on the device (components_ <-> feature-major Dt)."""
    out = torch.empty((cols, rows), dtype=src.dtype, device=src.device)
    f = getattr(lib, 'modl_transpose_' + ('f32' if src.dtype == torch.float32 else 'f64'))
    with torch.cuda.device(src.device):
        check(f(ptr(src), ptr(out), rows, cols, stream_ptr(src.device)), 'modl_transpose')
    return out


def gather_rows(src, perm):
    """src[perm] for a 2-D device tensor with contiguous rows, by the library's gather kernel."""
    idx = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int64)).to(src.device)
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=src.dtype, device=src.device)
    f = getattr(lib, 'modl_gather_rows_' + ('f32' if src.dtype == torch.float32 else 'f64'))
    with torch.cuda.device(src.device):
        check(f(ptr(src), src.stride(0), ptr(idx), idx.shape[0], src.shape[1], ptr(out), out.stride(0),
                stream_ptr(src.device)), 'modl_gather_rows')
    return out
