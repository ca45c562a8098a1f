"""Drop-in for modl/utils/math/enet.pyx (enet_norm :125, enet_projection :38,
enet_scale :150) on the GPU; 1-D vectors or, batched, the rows of a 2-D array."""
import numpy as np
import torch

from ._lib import lib, check
from .device import default_device, sfx, ptr, stream_ptr, to_device


def _dt(a):
    return a.dtype if isinstance(a, np.ndarray) else (np.float32 if a.dtype == torch.float32 else np.float64)


def enet_norm(v, l1_ratio):
    dev = default_device()
    dv = to_device(v, dev)
    rows = 1 if dv.dim() == 1 else dv.shape[0]
    n = dv.shape[-1]
    out = torch.empty(rows, dtype=dv.dtype, device=dev)
    check(getattr(lib, 'modl_enet_norm_' + sfx(_dt(v)))(ptr(dv), rows, n, n, 1, l1_ratio, ptr(out), stream_ptr(dev)))
    res = out.cpu().numpy()
    return float(res[0]) if dv.dim() == 1 else res


def enet_projection(v, out, radius, l1_ratio):
    dev = default_device()
    dv = to_device(v, dev)
    rows = 1 if dv.dim() == 1 else dv.shape[0]
    n = dv.shape[-1]
    dout = torch.empty_like(dv)
    rad = to_device(np.broadcast_to(np.asarray(radius, dtype=_dt(v)), (rows,)).copy(), dev)
    check(getattr(lib, 'modl_enet_projection_' + sfx(_dt(v)))(ptr(dv), ptr(dout), rows, n, n, 1, ptr(rad), l1_ratio,
                                                              stream_ptr(dev)))
    if isinstance(out, np.ndarray):
        out[...] = dout.cpu().numpy()
    else:
        out.copy_(dout)


def enet_scale(X, l1_ratio, radius=1):
    dev = default_device()
    dv = to_device(X, dev)
    if isinstance(X, torch.Tensor) and dv.data_ptr() != X.data_ptr():
        raise ValueError('enet_scale works in place: pass a contiguous device tensor or a numpy array')
    rows = 1 if dv.dim() == 1 else dv.shape[0]
    n = dv.shape[-1]
    check(getattr(lib, 'modl_enet_scale_' + sfx(_dt(X)))(ptr(dv), rows, n, n, 1, l1_ratio, radius, stream_ptr(dev)))
    if isinstance(X, np.ndarray):
        X[...] = dv.cpu().numpy()
