"""Drop-in for the reference's compiled module modl/decomposition/dict_fact_fast.pyx:
the same four callables with the same arguments and in-place behaviour, running
on the GPU through libmodl_hip.so.  NumPy arguments are staged to the device and
the mutated arrays copied back; torch CUDA tensors are used in place."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check
from .device import default_device, dtype_id, sfx, ptr, stream_ptr, to_device
from .randomkit import batch_weight as _batch_weight  # noqa: F401  (dict_fact_fast.pyx:115)


def _regression(kind, G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter, sweeps=None, _lib=None):
    lib = _lib if _lib is not None else globals()['lib']        # (tests: the diagnostics build, _lib.load_diag())
    dev = default_device()
    np_in = isinstance(code, np.ndarray)
    dt = code.dtype if np_in else (np.float32 if code.dtype == torch.float32 else np.float64)
    for name, a in (('G', G), ('Dx', Dx), ('X', X)):
        adt = a.dtype if isinstance(a, np.ndarray) else (np.float32 if a.dtype == torch.float32 else np.float64)
        if np.dtype(adt) != np.dtype(dt):
            raise TypeError('%s has dtype %s, code has %s (the reference dispatches on one fused type)' % (name, adt, dt))
    dG, dDx, dX, dcode = (to_device(a, dev) for a in (G, Dx, X, code))
    idx = torch.from_numpy(np.ascontiguousarray(np.asarray(indices), dtype=np.int64)).to(dev)
    b, k = dDx.shape
    p = dX.shape[1]
    multi = 1 if kind == 'multi' else 0
    nbytes = lib.modl_enet_regression_workspace(dtype_id(dt), b, k, multi)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    dsw = torch.zeros(b, dtype=torch.int32, device=dev) if sweeps is not None else None
    f = getattr(lib, 'modl_enet_regression_%s_gram_%s' % (kind, sfx(dt)))
    check(f(ptr(dG), ptr(dDx), ptr(dX), dX.stride(0), p, ptr(dcode), ptr(idx), b, k, l1_ratio, alpha,
            int(bool(positive)), tol, int(max_iter), ptr(dsw), ptr(ws), nbytes, stream_ptr(dev)),
          'modl_enet_regression_%s_gram' % kind)
    if sweeps is not None:
        sweeps[:] = dsw.cpu().numpy()
    if np_in:
        code[:] = dcode.cpu().numpy()
        if isinstance(Dx, np.ndarray) and l1_ratio == 0:
            Dx[:] = dDx.cpu().numpy()               # the ridge branch leaves the solution in Dx
        return code
    return dcode


def _enet_regression_single_gram(G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter, sweeps=None, _lib=None):
    """dict_fact_fast.pyx:125-215"""
    return _regression('single', G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter, sweeps, _lib)


def _enet_regression_multi_gram(G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter, sweeps=None, _lib=None):
    """dict_fact_fast.pyx:33-113"""
    return _regression('multi', G, Dx, X, code, indices, l1_ratio, alpha, positive, tol, max_iter, sweeps, _lib)


def _update_G_average(G_average, G, w_sample):
    """dict_fact_fast.pyx:217-228 (in place)"""
    dev = default_device()
    np_in = isinstance(G_average, np.ndarray)
    dt = G_average.dtype if np_in else (np.float32 if G_average.dtype == torch.float32 else np.float64)
    dGa = to_device(G_average, dev)
    dG = to_device(G, dev, dtype=dt)
    dw = to_device(np.asarray(w_sample) if not isinstance(w_sample, torch.Tensor) else w_sample, dev, dtype=dt)
    b, k = dGa.shape[0], dGa.shape[1]
    f = getattr(lib, 'modl_update_G_average_' + sfx(dt))
    check(f(ptr(dGa), ptr(dG), ptr(dw), b, k, stream_ptr(dev)), 'modl_update_G_average')
    if np_in:
        G_average[:] = dGa.cpu().numpy()
        return G_average
    return dGa
