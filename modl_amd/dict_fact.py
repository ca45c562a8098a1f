"""DictFact / Coder: the scikit-learn style surface of the reference
(modl/decomposition/dict_fact.py:23-745) over the MI355X kernels of
libmodl_hip.so.

The Python estimator stays host code: it draws the feature subset
(``Sampler``), the atom order (numpy legacy ``RandomState.permutation``,
dict_fact.py:672) and the minibatch weight exactly as the reference does, and
hands them to the device step (``modl_somf_code_and_partials`` +
``modl_somf_apply_and_update_dict``).  All state lives in HBM between
minibatches; attributes such as ``components_`` are copied back on access.

With ``torch.distributed`` initialised, every rank processes its own rows of a
global minibatch and the statistics increment ``[code^T code | X^T code]`` is
all-reduced (RCCL over xGMI when the backend is ``nccl``) between the two
phases; all ranks then perform the identical dictionary update.
"""
import ctypes as C
import os
import time
from math import ceil

import numpy as np
import torch
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array, check_random_state
from sklearn.utils.validation import check_is_fitted

from . import _lib


def gen_batches(n, batch_size):
    """sklearn.utils.gen_batches (slices of batch_size rows, the last one shorter; dict_fact.py:510) without its parameter
    validation - an inspect.signature() binding per call, 25 us in front of the first launch of every partial_fit."""
    start = 0
    for _ in range(int(n // batch_size)):
        end = start + batch_size
        yield slice(start, end)
        start = end
    if start < n:
        yield slice(start, n)

from ._lib import lib, check, SomfDesc, SomfState, SomfBatch, ProfEntry, AGG, OPT
from .device import (default_device, dtype_id, sfx, torch_dtype, ptr, stream_ptr, to_device, transpose_to, gather_rows)
from .randomkit import RandomState, Sampler, batch_weight
from .utils import get_sub_slice

MAX_INT = np.iinfo(np.int64).max


def _as_float_array(X):
    """check_array(order='C', dtype=[f32, f64]) for numpy input (dict_fact.py:299,328);
    device tensors pass through."""
    if isinstance(X, torch.Tensor):
        if X.dtype not in (torch.float32, torch.float64):
            X = X.to(torch.float64)
        return X
    return check_array(X, order='C', dtype=[np.float32, np.float64])


class HipBackend:
    """Device state + calls into libmodl_hip.so for one estimator."""

    name = 'hip'

    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else default_device()
        self.plan = None
        self.head = None
        self.flags = 0            # modl_somf_desc.flags (diagnostics)
        self._pid = os.getpid()   # a forked child must never release the parent's device objects

    # -- allocation ---------------------------------------------------------
    def allocate(self, desc_kwargs, n_samples, p, k, dtype):
        if k > 1024:
            raise ValueError('modl_amd supports n_components <= 1024 (the code solvers keep a sample\'s %d coefficients in '
                             'the registers of one wavefront); got %d' % (1024, k))
        self.dtype = np.dtype(dtype)
        self.k, self.p, self.n = k, p, n_samples
        td, dev = torch_dtype(dtype), self.device
        z = lambda *s: torch.zeros(s, dtype=td, device=dev)
        self.Dt, self.Bt, self.C = z(p, k), z(p, k), z(k, k)
        self.code = torch.ones((n_samples, k), dtype=td, device=dev)        # dict_fact.py:470
        self.comp_norm = z(k)
        self.G = z(k, k) if desc_kwargs['G_agg'] == 'full' else None
        self.Dx_average = z(n_samples, k) if desc_kwargs['Dx_agg'] == 'average' else None
        self.G_average = None
        if desc_kwargs['G_agg'] == 'average':
            # n k^2 elements (a disk memmap in the reference, dict_fact.py:431-439): in HBM while it fits next to
            # everything else, otherwise in pinned host memory that the kernels read and write over the host link
            # (the b Gram matrices of a minibatch: 2 b k^2 elements per minibatch, 2.4 ms at b = k = 256)
            nbytes = n_samples * k * k * self.dtype.itemsize
            free = torch.cuda.mem_get_info(dev)[0] if dev.type == 'cuda' else 0
            on_host = getattr(self, 'g_average_on_host', None)
            if on_host is None:
                on_host = nbytes > 0.6 * free
            if on_host:
                self.G_average = torch.zeros((n_samples, k, k), dtype=td, pin_memory=True)
            else:
                self.G_average = z(n_samples, k, k)
        self._make_plan(desc_kwargs)
        self.head = None          # [C | rows of Bt, compact]: allocated by the first two-phase (multi-GPU) step

    def _desc(self, kw):
        d = SomfDesc()
        d.dtype = dtype_id(self.dtype)
        d.k, d.p, d.n_samples = self.k, self.p, self.n
        d.G_agg, d.Dx_agg, d.optimizer = AGG[kw['G_agg']], AGG[kw['Dx_agg']], OPT[kw['optimizer']]
        d.code_pos, d.comp_pos, d.max_iter = int(bool(kw['code_pos'])), int(bool(kw['comp_pos'])), int(kw['max_iter'])
        d.code_alpha, d.code_l1_ratio = float(kw['code_alpha']), float(kw['code_l1_ratio'])
        d.comp_l1_ratio, d.tol, d.step_size = float(kw['comp_l1_ratio']), float(kw['tol']), float(kw['step_size'])
        d.max_batch = int(kw['max_batch'])
        d.flags = int(getattr(self, 'flags', 0))
        return d

    def _make_plan(self, kw):
        self.release_plan()
        self._desc_kw = dict(kw)
        d = self._desc(kw)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.modl_somf_plan_create(C.byref(d), C.byref(h)), 'modl_somf_plan_create')
        self.plan = h
        self.persist_recoveries = 0       # (of this plan: persistent launches completed by one workgroup, synchronize())

    def update_plan(self, kw):
        """set_params between minibatches (dict_fact.py:339-357)."""
        if kw['max_batch'] != self._desc_kw['max_batch'] or \
                (kw['G_agg'] == 'average' and self._desc_kw['G_agg'] != 'average'):
            self._make_plan(kw)
            return
        td, dev = torch_dtype(self.dtype), self.device
        if kw['Dx_agg'] == 'average' and self.Dx_average is None:
            self.Dx_average = torch.zeros((self.n, self.k), dtype=td, device=dev)
        if kw['G_agg'] == 'full' and self.G is None:
            self.G = torch.zeros((self.k, self.k), dtype=td, device=dev)
        self._desc_kw = dict(kw)
        d = self._desc(kw)
        check(lib.modl_somf_plan_update(self.plan, C.byref(d)), 'modl_somf_plan_update')

    def release_plan(self):
        if getattr(self, '_pid', None) != os.getpid():
            self.plan = None
            return
        if getattr(self, 'plan', None):
            lib.modl_somf_plan_destroy(self.plan)
            self.plan = None

    def release_all(self):
        self.release_plan()
        if getattr(self, '_pid', None) != os.getpid():
            self.tplan = self.comm = self._order_rk = None
            return
        if getattr(self, 'comm', None):
            lib.modl_comm_destroy(self.comm)
            self.comm = None
        if getattr(self, 'tplan', None):
            lib.modl_somf_plan_destroy(self.tplan)
            self.tplan = None
        if getattr(self, '_order_rk', None):
            lib.modl_rk_destroy(self._order_rk)
            self._order_rk = None
        self._host_chunk_buffers = None               # 2 x 256 MB pinned + 2 x 256 MB of HBM (_HostChunks)

    def __del__(self):
        try:
            self.release_all()
        except Exception:
            pass

    # -- state access ---------------------------------------------------------
    def _state(self):
        s = SomfState()
        s.d_Dt, s.d_Bt, s.d_C, s.d_code = ptr(self.Dt), ptr(self.Bt), ptr(self.C), ptr(self.code)
        s.d_comp_norm, s.d_G = ptr(self.comp_norm), ptr(self.G)
        s.d_Dx_average, s.d_G_average = ptr(self.Dx_average), ptr(self.G_average)
        return s

    def set_dictionary(self, D):                      # D: (k, p) numpy or tensor
        Dd = to_device(D, self.device, dtype=self.dtype)
        self.Dt = transpose_to(Dd, self.k, self.p)

    def get_dictionary(self):
        return transpose_to(self.Dt, self.p, self.k).cpu().numpy()

    def broadcast_dictionary(self, dist):
        dist.broadcast(self.Dt, src=0)

    def set_B(self, B):
        self.Bt = transpose_to(to_device(B, self.device, dtype=self.dtype), self.k, self.p)

    def get_B(self):
        return transpose_to(self.Bt, self.p, self.k).cpu().numpy()

    def get(self, name):
        t = getattr(self, name)
        if t is None:
            return None
        if t.device.type == 'cpu':                 # host-resident state the kernels write (G_average_): let them finish
            self.synchronize()
            return t.numpy().copy()
        return t.cpu().numpy()

    def set(self, name, value):
        cur = getattr(self, name)
        new = to_device(value, self.device, dtype=self.dtype)
        if cur is not None and tuple(cur.shape) == tuple(new.shape):
            cur.copy_(new)
        else:
            setattr(self, name, new.clone())

    def scale_atoms(self, l1_ratio, radius=1.0):      # enet_scale on every atom (dict_fact.py:465-468)
        f = getattr(lib, 'modl_enet_scale_' + sfx(self.dtype))
        with torch.cuda.device(self.device):
            check(f(ptr(self.Dt), self.k, self.p, 1, self.k, l1_ratio, radius, stream_ptr(self.device)), 'modl_enet_scale')

    def full_gram(self):
        if self.G is None:
            self.G = torch.zeros((self.k, self.k), dtype=torch_dtype(self.dtype), device=self.device)
        check(lib.modl_somf_full_gram(self.plan, ptr(self.Dt), ptr(self.G), stream_ptr(self.device)), 'modl_somf_full_gram')

    def n_rows(self, name):
        return getattr(self, name).shape[0]

    def shuffle_rows(self, name, swaps):
        t = getattr(self, name)
        row_bytes = t[0].numel() * t.element_size()
        with torch.cuda.device(self.device):
            check(lib.modl_apply_swaps_rows_device(ptr(t), t.shape[0], row_bytes, swaps.ctypes.data_as(C.c_void_p),
                                                   stream_ptr(self.device)), 'modl_apply_swaps_rows_device')

    # -- data ---------------------------------------------------------------
    def stage_X(self, X):
        return to_device(X, self.device, dtype=self.dtype)

    def stage_image(self, image):
        """(H, W, C) float32 / float64 image -> HBM, once per fit (modl_amd/image.py)."""
        return torch.from_numpy(np.ascontiguousarray(image)).to(self.device)

    def image_patches(self, d_image, indices_3d, patch_shape, with_mean, with_std):
        """Flattened, channel-wise centred / normalised patches at the origins indices_3d (n, 3) -> (n, x*y*z) tensor."""
        H, W, Cc = d_image.shape
        x, y, z = (int(v) for v in patch_shape)
        idx = torch.from_numpy(np.ascontiguousarray(indices_3d, dtype=np.int64)).to(self.device)
        n = idx.shape[0]
        out = torch.empty((n, x * y * z), dtype=d_image.dtype, device=self.device)
        f = getattr(lib, 'modl_image_patches_' + ('f32' if d_image.dtype == torch.float32 else 'f64'))
        with torch.cuda.device(self.device):
            check(f(ptr(d_image), H, W, Cc, ptr(idx), n, x, y, z, int(bool(with_mean)), int(bool(with_std)), ptr(out),
                    x * y * z, stream_ptr(self.device)), 'modl_image_patches')
        return out

    def take_rows(self, Xh, perm):
        return gather_rows(Xh, perm)

    # seconds a rank waits for its stream to drain while it holds the library's RCCL communicator before it aborts the
    # communicator and raises (a peer that died leaves the others inside ncclAllReduce: modl_comm_wait); <= 0: no limit
    COMM_TIMEOUT_S = float(os.environ.get('MODL_COMM_TIMEOUT_S', '1800'))

    def synchronize(self):
        comm = getattr(self, 'comm', None)
        if comm is not None and getattr(self, '_comm_world', 1) > 1:
            # several ranks on the library's communicator: a bounded wait that watches RCCL's asynchronous errors - an
            # RCCL failure surfaces as MODL_ERCCL here instead of a hang in hipStreamSynchronize
            check(lib.modl_comm_wait(comm, stream_ptr(self.device), self.COMM_TIMEOUT_S), 'modl_comm_wait')
        if self.plan is not None:
            # (also a synchronisation of the stream) a persistent dictionary-update launch whose workgroups were not all
            # resident never hangs: before its first block it is completed on the device by one workgroup (counted here,
            # with a warning; the plan keeps one launch per block from then on), later it leaves the update incomplete and
            # that must not pass for a fit (MODL_ETIMEOUT)
            check(lib.modl_somf_status(self.plan, stream_ptr(self.device)), 'modl_somf_status')
            n = C.c_int64(0)
            check(lib.modl_somf_persist_recoveries(self.plan, C.byref(n)), 'modl_somf_persist_recoveries')
            if n.value > self.persist_recoveries:
                import warnings
                warnings.warn('modl_amd: %d persistent dictionary-update launch(es) could not run (their workgroups were not '
                              'resident together: is another process using this GPU?); each was completed by one workgroup '
                              'and this estimator now runs one launch per block of atoms' % (n.value - self.persist_recoveries),
                              RuntimeWarning, stacklevel=3)
                self.persist_recoveries = n.value
        torch.cuda.synchronize(self.device)

    # -- the step -------------------------------------------------------------
    def _batch(self, Xh, batch, idx, subset, order, w_sample, w, reduction, b_global):
        bt = SomfBatch()
        rows = Xh[batch]
        bt.d_X, bt.ldx, bt.b = ptr(rows), Xh.stride(0), rows.shape[0]
        keep = [rows]
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        order = np.ascontiguousarray(order, dtype=np.int64)
        keep += [idx, order]
        bt.h_sample_idx, bt.h_order = idx.ctypes.data, order.ctypes.data
        if subset is None or len(subset) == self.p:
            bt.s, bt.h_subset = self.p, None                  # every feature: no gather (any order is the same set)
        else:
            subset = np.ascontiguousarray(subset, dtype=np.int64)
            keep.append(subset)
            bt.s, bt.h_subset = len(subset), subset.ctypes.data
        if w_sample is not None:
            w_sample = np.ascontiguousarray(w_sample, dtype=self.dtype)
            keep.append(w_sample)
            bt.h_w_sample = w_sample.ctypes.data
        bt.w, bt.reduction, bt.b_global = float(w), float(reduction), int(b_global)
        return bt, keep

    def phase1(self, Xh, batch, idx, subset, order, w_sample, w, reduction, b_global):
        """Several GPUs, per rank: codes, update of the rank's partial statistics; returns the head
        [C_r | rows of B_r of the sampled features] that the caller sums over the ranks."""
        bt, keep = self._batch(Xh, batch, idx, subset, order, w_sample, w, reduction, b_global)
        st = self._state()
        if self.head is None:
            self.head = torch.zeros(self.k * self.k + self.p * self.k, dtype=torch_dtype(self.dtype), device=self.device)
        check(lib.modl_somf_code_and_partials(self.plan, C.byref(st), C.byref(bt), ptr(self.head),
                                              stream_ptr(self.device)), 'modl_somf_code_and_partials')
        self._pending = (bt, keep)
        n = C.c_int64()
        check(lib.modl_somf_head_elems(self.plan, C.byref(n)), 'modl_somf_head_elems')
        return self.head[:n.value]

    def step(self, Xh, batch, idx, subset, order, w_sample, w, reduction, b_global):
        """The whole minibatch in one call (single GPU): the statistics update rides in the GEMM epilogues."""
        bt, keep = self._batch(Xh, batch, idx, subset, order, w_sample, w, reduction, b_global)
        st = self._state()
        check(lib.modl_somf_step(self.plan, C.byref(st), C.byref(bt), stream_ptr(self.device)), 'modl_somf_step')

    def fit_chunk(self, Xh, batch_size, sample_indices, sampler, np_random_state, n_iter, learning_rate, reduction,
                  b_global=None, comm=None):
        """A whole chunk of minibatches in ONE call (modl_somf_partial_fit_chunk): subset draws, minibatch weights and
        atom orders are drawn inside the library - the atom order by a generator that is loaded with numpy's legacy
        MT19937 state and handed back afterwards, so `np_random_state` continues exactly as if it had drawn them
        (dict_fact.py:672).  Returns (new n_iter_, minibatches enqueued); an error raised here carries both as
        `e.n_iter` / `e.n_done` - n_iter_, the sampler and `np_random_state` are then where the last enqueued minibatch
        left them."""
        if getattr(self, '_order_rk', None) is None:
            h = C.c_void_p()
            check(lib.modl_rk_create(0, C.byref(h)), 'modl_rk_create')
            self._order_rk = h
        kind, key, pos, has_gauss, cached = np_random_state.get_state()
        key = np.ascontiguousarray(key, dtype=np.uint32)
        check(lib.modl_rk_set_mt_state(self._order_rk, key.ctypes.data_as(C.c_void_p), int(pos)), 'modl_rk_set_mt_state')
        idx = None if sample_indices is None else np.ascontiguousarray(sample_indices, dtype=np.int64)
        bg = None if b_global is None else np.ascontiguousarray(b_global, dtype=np.int64)
        n = C.c_int64(int(n_iter))
        done = C.c_int64(0)
        st = self._state()
        try:
            rc = lib.modl_somf_partial_fit_chunk(
                self.plan, C.byref(st), ptr(Xh), Xh.stride(0), Xh.shape[0], int(batch_size),
                None if idx is None else idx.ctypes.data_as(C.c_void_p), sampler._h, self._order_rk, C.byref(n),
                float(learning_rate), float(reduction), None if bg is None else bg.ctypes.data_as(C.c_void_p), comm,
                C.byref(done), stream_ptr(self.device))
        finally:
            out = np.empty(624, dtype=np.uint32)
            p2 = C.c_int32()
            check(lib.modl_rk_get_mt_state(self._order_rk, out.ctypes.data_as(C.c_void_p), C.byref(p2)))
            np_random_state.set_state((kind, out, int(p2.value), has_gauss, cached))
        if rc != 0:
            # the library has rewound both generators and n_iter to the last minibatch it enqueued completely: the caller
            # books those minibatches before it raises (ChunkError carries the counts)
            try:
                check(rc, 'modl_somf_partial_fit_chunk')
            except Exception as e:
                e.n_iter, e.n_done = int(n.value), int(done.value)
                raise
        return int(n.value), int(done.value)

    def native_comm(self, dist):
        """An RCCL communicator owned by the library (modl_comm_*): rank 0 draws the unique id, torch.distributed
        only ships its 128 bytes (dist=None: one rank, nothing is shipped).  Returns None - on EVERY rank - when the
        communicator cannot be created on some rank (no librccl, RCCL error): the caller then uses torch's collective."""
        if getattr(self, 'comm', None) is None and not getattr(self, '_comm_failed', False):
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
            ident = (C.c_char * 128)()
            rc = lib.modl_comm_unique_id(ident) if rank == 0 else 0
            box = [bytes(ident.raw), rc]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            h = C.c_void_p()
            if box[1] == 0:
                ident = C.create_string_buffer(box[0], 128)
                with torch.cuda.device(self.device):
                    rc = lib.modl_comm_create(ident, rank, world, C.byref(h))
            else:
                rc = box[1]
            if world > 1:                                   # the same decision on every rank
                flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32,
                                    device=self.device if dist.get_backend() == 'nccl' else 'cpu')
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item())
            else:
                ok = rc == 0
            if ok:
                self.comm = h
                self._comm_world = world
            else:
                if rc == 0:
                    lib.modl_comm_destroy(h)
                self._comm_failed = True
                import warnings
                warnings.warn('modl_amd: the library\'s own RCCL communicator could not be created (%s); the all-reduce '
                              'of the statistics head goes through torch.distributed instead'
                              % _lib.error_string(rc if rc != 0 else -6))
        return getattr(self, 'comm', None)

    def step_dist(self, comm, Xh, batch, idx, subset, order, w_sample, w, reduction, b_global):
        """Several GPUs, ONE call: phase 1, ncclAllReduce of the head, phase 2 - all on the current stream."""
        bt, keep = self._batch(Xh, batch, idx, subset, order, w_sample, w, reduction, b_global)
        st = self._state()
        check(lib.modl_somf_step_dist(self.plan, C.byref(st), C.byref(bt), comm, stream_ptr(self.device)),
              'modl_somf_step_dist')

    def phase2(self, head):
        """The dictionary update from the summed head (identical on every rank)."""
        bt, keep = self._pending
        st = self._state()
        check(lib.modl_somf_apply_and_update_dict(self.plan, C.byref(st), C.byref(bt), ptr(self.head),
                                                  stream_ptr(self.device)), 'modl_somf_apply_and_update_dict')
        self._pending = None

    def transform(self, Xh, kw, G=None, to_host=True):
        """Codes from a warm start of ones, in chunks of a dedicated plan (large max_batch)."""
        kw = dict(kw, max_batch=4096, G_agg='masked', Dx_agg='masked', optimizer='variational')
        if getattr(self, 'tplan', None) is None or kw != self._tplan_kw:
            if getattr(self, 'tplan', None):
                lib.modl_somf_plan_destroy(self.tplan)
                self.tplan = None
            d = self._desc(kw)
            h = C.c_void_p()
            with torch.cuda.device(self.device):
                check(lib.modl_somf_plan_create(C.byref(d), C.byref(h)), 'modl_somf_plan_create')
            self.tplan, self._tplan_kw = h, kw
        n = Xh.shape[0]
        out = torch.empty((n, self.k), dtype=torch_dtype(self.dtype), device=self.device)
        check(lib.modl_somf_transform(self.tplan, ptr(self.Dt), ptr(G), ptr(Xh), Xh.stride(0), n, ptr(out),
                                      stream_ptr(self.device)), 'modl_somf_transform')
        return out.cpu().numpy() if to_host else out

    def objective(self, Xh, code):
        """[sum (X - code D)^2, sum |code|, sum code^2] of device-resident X and codes (dict_fact.py:108-112)."""
        n, p = Xh.shape
        nbytes = lib.modl_objective_workspace(dtype_id(self.dtype), n, p)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        out = torch.empty(3, dtype=torch.float64, device=self.device)
        f = getattr(lib, 'modl_objective_' + sfx(self.dtype))
        with torch.cuda.device(self.device):
            check(f(ptr(Xh), Xh.stride(0), n, p, ptr(self.Dt), self.k, ptr(code), ptr(ws), nbytes, ptr(out),
                    stream_ptr(self.device)), 'modl_objective')
        return out.cpu().numpy()

    def last_sweeps(self):
        out = np.zeros(self._desc_kw['max_batch'], dtype=np.int32)
        n = C.c_int()
        check(lib.modl_somf_last_sweeps(self.plan, out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(n),
                                        stream_ptr(self.device)))
        return out[:n.value]

    def sweeps_history(self, n_minibatches):
        """diagnostics: keep the sweep counts of the next `n_minibatches` minibatches (a ring); returns a function that
        reads them back as an (n, max_batch) array, oldest first (None: off)"""
        if not n_minibatches:
            check(lib.modl_somf_sweeps_history(self.plan, None, 0))
            self._sweep_hist = None
            return None
        mb = self._desc_kw['max_batch']
        self._sweep_hist = torch.zeros((int(n_minibatches), mb), dtype=torch.int32, device=self.device)
        check(lib.modl_somf_sweeps_history(self.plan, ptr(self._sweep_hist), int(n_minibatches)))
        return lambda: self._sweep_hist.cpu().numpy()

    # -- profiling ------------------------------------------------------------
    PROF_SECTIONS = ('code_gemm', 'code_solve', 'stats_gemm', 'stats_apply', 'dict_update')

    def prof_enable(self, on=True, sections=None, every=1):
        """Time the step's sections with HIP events; `sections` restricts the events to the named ones, `every`
        to one minibatch in `every`."""
        check(lib.modl_somf_prof_stride(self.plan, int(every)))
        flag = int(bool(on))
        if on and sections is not None:
            flag = 0
            for name in sections:
                flag |= 1 << (self.PROF_SECTIONS.index(name) + 1)
        check(lib.modl_somf_prof_enable(self.plan, flag))

    def host_wait_ms(self, reset=True):
        """host time spent waiting for the device (a free staging slot) since the last reset"""
        out = C.c_double()
        check(lib.modl_somf_host_wait_ms(self.plan, C.byref(out), int(reset)))
        return out.value

    def prof_reset(self):
        check(lib.modl_somf_prof_reset(self.plan))

    def prof_get(self):
        arr = (ProfEntry * 16)()
        n = C.c_int()
        check(lib.modl_somf_prof_get(self.plan, arr, 16, C.byref(n)))
        return {arr[i].name.decode(): dict(ms=arr[i].ms_total, launches=arr[i].launches, calls=arr[i].calls)
                for i in range(n.value)}


HOST_CHUNK_BYTES = 1 << 28      # host input larger than this is streamed to HBM in chunks of about this size


class _HostChunks:
    """Streams a host array (numpy / np.memmap, possibly larger than HBM) to the device in chunks of whole
    minibatches: while chunk c is being fitted, a worker thread copies chunk c + 1 into pinned memory and starts its
    upload on a side stream (two pinned + two device buffers).  The repeated-`partial_fit` streaming contract of the
    reference (dict_fact.py:313-337) without ever holding X on the device."""

    def __init__(self, be, X, rows_per_chunk):
        from concurrent.futures import ThreadPoolExecutor
        self.X, self.rows, self.device = X, int(rows_per_chunk), be.device
        td = torch_dtype(be.dtype)
        self.n = X.shape[0]
        self.n_chunks = (self.n + self.rows - 1) // self.rows
        rows = min(self.rows, self.n)
        # the two pinned and the two device buffers are kept on the backend across partial_fit calls (a pinned
        # allocation of 256 MB costs tens of milliseconds)
        cache = getattr(be, '_host_chunk_buffers', None)
        if cache is None or cache[0] != (rows, X.shape[1], td):
            cache = ((rows, X.shape[1], td),
                     [torch.empty((rows, X.shape[1]), dtype=td, pin_memory=True) for _ in range(2)],
                     [torch.empty((rows, X.shape[1]), dtype=td, device=self.device) for _ in range(2)],
                     [None, None], [None, None])
            be._host_chunk_buffers = cache
        self.cache = cache
        self.pinned, self.dev = cache[1], cache[2]
        # events carried over from the last call on these buffers (it returns without synchronising):
        self.consumed = list(cache[3])    # the fit has finished reading dev[i]
        self.copied = list(cache[4])      # the upload out of pinned[i] / into dev[i] has finished
        self.stream = torch.cuda.Stream(self.device)
        self.pool = ThreadPoolExecutor(1)

    def _stage(self, c):
        i = c % 2
        r0, r1 = c * self.rows, min(self.n, (c + 1) * self.rows)
        if self.copied[i] is not None:
            self.copied[i].synchronize()                  # the pinned buffer is free again
        host = self.pinned[i][:r1 - r0]
        np.copyto(host.numpy(), self.X[r0:r1], casting='same_kind')
        with torch.cuda.stream(self.stream):
            if self.consumed[i] is not None:
                self.stream.wait_event(self.consumed[i])  # the device buffer is free again
            self.dev[i][:r1 - r0].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.copied[i] = ev
        return r0, r1

    def __iter__(self):
        finished = False
        try:
            pending = self.pool.submit(self._stage, 0)
            for c in range(self.n_chunks):
                r0, r1 = pending.result()
                if c + 1 < self.n_chunks:
                    pending = self.pool.submit(self._stage, c + 1)
                main = torch.cuda.current_stream(self.device)
                main.wait_event(self.copied[c % 2])
                yield r0, self.dev[c % 2][:r1 - r0]
                ev = torch.cuda.Event()
                ev.record(main)
                self.consumed[c % 2] = ev
            finished = True
        finally:
            self.pool.shutdown(wait=True)
            if finished:
                # every upload has been waited for by the fit (main.wait_event above); the device buffers are still
                # being read by the enqueued minibatches: the NEXT call waits for these events before it reuses them
                self.cache[3][:] = self.consumed
                self.cache[4][:] = self.copied
            else:
                # an error in a minibatch (or a generator that is dropped half way): no upload out of the pinned
                # buffers and no fit reading the device buffers may outlive the call - the buffers are reused
                self.stream.synchronize()
                torch.cuda.current_stream(self.device).synchronize()
                self.cache[3][:] = [None, None]
                self.cache[4][:] = [None, None]


class _SubsetsAhead:
    """Draws the feature subsets of the coming minibatches on a worker thread while the current one is being
    enqueued: the bit-exact MT19937 shuffle of p indices (sampler.pyx:41-70) is ~8 ns per feature of pure host time
    per minibatch and depends on nothing but the sampler's own state, so the draws - same generator, same order -
    can run ahead (ctypes releases the GIL around the call)."""

    _pool_pid = None
    _pool = None          # one worker thread for the whole process, started at first use (a thread per partial_fit
                          # call costs ~0.1 ms, which shows in short calls)

    def __init__(self, sampler, reduction, n, depth=8):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        if _SubsetsAhead._pool is None or _SubsetsAhead._pool_pid != os.getpid():     # (threads do not survive a fork)
            _SubsetsAhead._pool = ThreadPoolExecutor(1, thread_name_prefix='modl-subsets')
            _SubsetsAhead._pool_pid = os.getpid()
        self.q = queue.Queue(maxsize=depth)
        self.stop = False
        self.sampler, self.reduction = sampler, reduction
        # to rewind the draws nobody consumed (see close()); a stand-in sampler without `restore` is left as it is
        self.state0 = sampler.__getstate__() if hasattr(sampler, 'restore') else None
        self.drawn = self.consumed = 0

        def put(item):                                       # a bounded put, so that close() can always end the task
            while not self.stop:
                try:
                    self.q.put(item, timeout=0.05)
                    return
                except queue.Full:
                    pass

        def work():
            try:
                for _ in range(n):
                    if self.stop:
                        return
                    item = sampler.yield_subset(reduction)
                    self.drawn += 1
                    put(item)
            except BaseException as e:                       # handed to the consumer
                put(e)
        self.task = _SubsetsAhead._pool.submit(work)

    def next(self):
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        self.consumed += 1
        return item

    def close(self):
        """Ends the worker (it is free for the next call).  If the consumer stopped early - an error in a minibatch -
        the sampler has drawn subsets nobody used: it is rewound and re-plays exactly the consumed draws, so that a
        retry continues the reference's MT19937 subset stream (sampler.pyx:41-70) where the last FITTED minibatch
        left it."""
        self.stop = True
        self.task.result()
        if self.drawn != self.consumed and self.state0 is not None:
            self.sampler.restore(self.state0)
            for _ in range(self.consumed):
                self.sampler.yield_subset(self.reduction)


class _DeviceRows:
    """Adapter so RandomState.shuffle_with_trace can permute device-resident rows."""

    def __init__(self, backend, name):
        self.backend, self.name = backend, name

    def __len__(self):
        return self.backend.n_rows(self.name)

    def _modl_device_rows(self, swaps):
        self.backend.shuffle_rows(self.name, swaps)


class CodingMixin(TransformerMixin):
    def _set_coding_params(self, n_components, code_alpha=1, code_l1_ratio=1, tol=1e-2, max_iter=100,
                           code_pos=False, random_state=None, n_threads=1):
        self.n_components = n_components
        self.code_l1_ratio = code_l1_ratio
        self.code_alpha = code_alpha
        self.code_pos = code_pos
        self.random_state = random_state
        self.tol = tol
        self.max_iter = max_iter
        self.n_threads = n_threads            # kept for API compatibility; the GPU step has no thread pool

    def _make_backend(self):
        return HipBackend(getattr(self, 'device', None))

    def _plan_kwargs(self, max_batch):
        g = lambda n, d: getattr(self, n, d)
        return dict(G_agg=g('G_agg', 'masked'), Dx_agg=g('Dx_agg', 'masked'), optimizer=g('optimizer', 'variational'),
                    code_pos=self.code_pos, comp_pos=g('comp_pos', False), max_iter=self.max_iter,
                    code_alpha=self.code_alpha, code_l1_ratio=self.code_l1_ratio,
                    comp_l1_ratio=g('comp_l1_ratio', 0), tol=self.tol, step_size=g('step_size', 1),
                    max_batch=max_batch)

    def _transform(self, X, to_host):
        check_is_fitted(self, 'components_')
        be = self._backend
        if not isinstance(X, torch.Tensor):
            X = check_array(X, order='C', dtype=be.dtype.type)
        Xh = be.stage_X(X)
        if Xh.shape[1] != be.p:
            raise ValueError('X has %d features, the dictionary has %d' % (Xh.shape[1], be.p))
        use_G = getattr(self, 'G_agg', None) == 'full' and be.G is not None
        return Xh, be.transform(Xh, self._plan_kwargs(4096), be.G if use_G else None, to_host=to_host)

    def transform(self, X):
        """Codes of the rows of X on the dictionary (dict_fact.py:47-92)."""
        return self._transform(X, True)[1]

    def score(self, X):
        """Objective value on test data X (dict_fact.py:94-114).  Test data, codes and dictionary stay on the device:
        only the three sums of the objective come back (X may be a device tensor, e.g. a scorer's resident test set)."""
        Xh, code = self._transform(X, False)
        sq_res, norm1_code, norm2_code = self._backend.objective(Xh, code)
        regul = self.code_alpha * (norm1_code * self.code_l1_ratio + (1 - self.code_l1_ratio) * norm2_code / 2)
        return float((sq_res / 2 + regul) / Xh.shape[0])


def _dist():
    """torch.distributed when a process group with more than one rank is up, else None"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _sum_over_ranks(arr, device):
    """Several GPUs: C_ and B_ are kept as per-rank partial sums; the attribute is their sum (a collective: every
    rank has to read it)."""
    dist = _dist()
    if dist is None or arr is None:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dist.get_backend() == 'nccl':
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _rank0_share(value):
    """Several ranks: a statistic that is kept as per-rank partial sums is SET by giving rank 0 the value and every
    other rank zeros, so that the sum over the ranks is the value (and not world x value)."""
    dist = _dist()
    if dist is None or dist.get_rank() == 0:
        return value
    return np.zeros_like(np.asarray(value))


def _state_property(name, getter=None, setter=None, summed=False, local=False):
    """summed: the attribute of a statistic kept as per-rank partial sums (C_, B_) - reading it is a COLLECTIVE with
    several ranks (every rank has to read it); `local=True` gives the non-collective twin (this rank's partial sum)."""
    def fget(self):
        be = self.__dict__.get('_backend')
        if be is None or getattr(be, 'Dt', None) is None:
            raise AttributeError(name)
        val = getter(be) if getter else be.get(name)
        if val is None:
            raise AttributeError(name)
        return _sum_over_ranks(val, be.device) if summed and not local else val

    def fset(self, value):
        be = self.__dict__.get('_backend')
        if be is None:
            raise AttributeError('%s can only be set after prepare()' % name)
        if summed:
            if not local:                                 # (the local twin sets THIS rank's share as given)
                value = _rank0_share(value)
            self.__dict__['_stats_on_rank0'] = False      # (true again after the next consolidate_statistics())
        (setter(be, value) if setter else be.set(name, value))
    return property(fget, fset)


class DictFact(CodingMixin, BaseEstimator):
    """Stochastic-subsampled online matrix factorization (SOMF) on one or several
    MI355X.  Constructor, attributes and methods follow the reference estimator
    (modl/decomposition/dict_fact.py:127-284); ``device`` is the only addition."""

    def __init__(self, reduction=1, learning_rate=1, sample_learning_rate=0.76, Dx_agg='masked', G_agg='masked',
                 optimizer='variational', dict_init=None, code_alpha=1, code_l1_ratio=1, comp_l1_ratio=0,
                 step_size=1, tol=1e-2, max_iter=100, code_pos=False, comp_pos=False, random_state=None,
                 n_epochs=1, n_components=10, batch_size=10, verbose=0, callback=None, n_threads=1,
                 rand_size=True, replacement=True, device=None):
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.sample_learning_rate = sample_learning_rate
        self.Dx_agg = Dx_agg
        self.G_agg = G_agg
        self.reduction = reduction
        self.dict_init = dict_init
        self._set_coding_params(n_components, code_l1_ratio=code_l1_ratio, code_alpha=code_alpha,
                                code_pos=code_pos, random_state=random_state, tol=tol, max_iter=max_iter,
                                n_threads=n_threads)
        self.comp_l1_ratio = comp_l1_ratio
        self.comp_pos = comp_pos
        self.optimizer = optimizer
        self.step_size = step_size
        self.n_epochs = n_epochs
        self.verbose = verbose
        self.callback = callback
        self.rand_size = rand_size
        self.replacement = replacement
        self.device = device

    # device-resident attributes of the reference (dict_fact.py:225-249)
    components_ = _state_property('components_', lambda be: be.get_dictionary(), lambda be, v: be.set_dictionary(v))
    # Several ranks: C_ and B_ live as per-rank partial sums.  `B_` / `C_` are the sums over the ranks -
    # a COLLECTIVE read, every rank must perform it; `local_B_` / `local_C_` are this rank's share, no communication
    # (after consolidate_statistics() rank 0's share is the whole statistic).  Setting `B_` / `C_` puts the value on
    # rank 0 (zeros elsewhere: the sum is the value); setting a `local_` twin sets this rank's share as given.
    B_ = _state_property('B_', lambda be: be.get_B(), lambda be, v: be.set_B(v), summed=True)
    C_ = _state_property('C', summed=True)
    local_B_ = _state_property('B_', lambda be: be.get_B(), lambda be, v: be.set_B(v), summed=True, local=True)
    local_C_ = _state_property('C', summed=True, local=True)
    code_ = _state_property('code')
    comp_norm_ = _state_property('comp_norm')
    G_ = _state_property('G')
    Dx_average_ = _state_property('Dx_average')
    G_average_ = _state_property('G_average')

    # ------------------------------------------------------------------ fit
    def fit(self, X):
        """dict_fact.py:286-311"""
        X = _as_float_array(X)
        if self.dict_init is None:
            dict_init = X
        else:
            dict_init = check_array(self.dict_init, dtype=(np.float32 if X.dtype in (np.float32, torch.float32)
                                                           else np.float64))
        self.prepare(n_samples=X.shape[0], X=dict_init)
        Xh = self._backend.stage_X(X)
        for _ in range(self.n_epochs):
            self.partial_fit(Xh)
            permutation = self.shuffle()
            Xh = self._backend.take_rows(Xh, permutation)
        return self

    def partial_fit(self, X, sample_indices=None, _sync=True):
        """dict_fact.py:313-337.  X: numpy array or device tensor (n, n_features).
        `_sync=False` (not part of the reference's surface) returns as soon as the minibatches are enqueued on
        the stream, so that a streaming caller can overlap the production of its next chunk; `time_` then only
        counts the host time."""
        be = self._backend
        if not (isinstance(X, np.memmap) and X.dtype == be.dtype and X.flags.c_contiguous):
            X = _as_float_array(X)                           # (a memmap of the right type is left on disk)
        if X.shape[1] != be.p:
            raise ValueError('X has %d features, expected %d' % (X.shape[1], be.p))
        if self.batch_size > be._desc_kw['max_batch']:
            be.update_plan(self._plan_kwargs(self.batch_size))
        t0 = time.perf_counter()
        self._cb_time = 0.0
        n = X.shape[0]
        batches = list(gen_batches(n, self.batch_size))
        b_global = self._global_batch_sizes(n, len(batches))
        if b_global is not None and len(batches):
            # several ranks: whatever route the minibatches take (Python loop, one call per chunk), C_ / B_ are per-rank
            # partial sums again afterwards - a pickle written now must say so (consolidate_statistics() resets this)
            self._stats_on_rank0 = False
        # host input that does not fit a chunk is streamed: pinned, double-buffered chunks of whole minibatches
        chunk_rows = getattr(self, '_host_chunk_rows', None) or \
            max(1, HOST_CHUNK_BYTES // max(1, X.shape[1] * be.dtype.itemsize))
        chunk_rows = max(self.batch_size, chunk_rows // self.batch_size * self.batch_size)
        streamed = not isinstance(X, torch.Tensor) and hasattr(be, 'plan') and n > chunk_rows
        chunks = _HostChunks(be, X, chunk_rows) if streamed else [(0, be.stage_X(X))]
        # (a callback may change `reduction` between two minibatches: then every subset is drawn when it is needed)
        chunk_call = self._chunk_call_applies(be, sample_indices)
        comm = None
        if chunk_call and (self._world() > 1 or getattr(self, '_force_reduce', False)):
            # the communicator is resolved BEFORE the route is chosen: when it cannot be created (no librccl, an RCCL
            # error - decided on every rank together, HipBackend.native_comm) the chunk call must not run with comm = None,
            # which is the single-GPU step without any reduction; the per-minibatch loop with torch's collective runs instead
            comm = self._native_comm(be)
            if comm is None:
                chunk_call = False
        ahead = not chunk_call and len(batches) >= 4 and self.callback is None and not self.verbose
        self._subsets = _SubsetsAhead(self.feature_sampler_, self.reduction, len(batches)) if ahead else None
        t = 0
        try:
            if chunk_call:
                # the whole per-minibatch loop in ONE library call per chunk (same draws, same order, same bits)
                for r0, Xh in chunks:
                    nb = -(-Xh.shape[0] // self.batch_size)
                    idx = get_sub_slice(sample_indices, slice(r0, r0 + Xh.shape[0]))
                    err = None
                    try:
                        self.n_iter_, done = be.fit_chunk(Xh, self.batch_size, idx, self.feature_sampler_,
                                                          self.random_state, self.n_iter_, self.learning_rate,
                                                          self.reduction,
                                                          None if b_global is None else b_global[t:t + nb], comm)
                    except Exception as e:
                        # the library left n_iter, the sampler and the order generator where the last minibatch it
                        # enqueued left them: book exactly those minibatches, then report
                        if not hasattr(e, 'n_done'):
                            raise
                        self.n_iter_, done, err = e.n_iter, e.n_done, e
                    for batch in list(gen_batches(Xh.shape[0], self.batch_size))[:done]:   # dict_fact.py:511
                        self.sample_n_iter_[idx[batch]] += 1
                    if err is not None:
                        raise err
                    t += nb
                chunks = ()
            for r0, Xh in chunks:
                for batch in gen_batches(Xh.shape[0], self.batch_size):
                    whole = slice(r0 + batch.start, r0 + batch.stop)
                    self._single_batch_fit(Xh, batch, get_sub_slice(sample_indices, whole),
                                           b_global=None if b_global is None else b_global[t])
                    t += 1
        finally:
            if self._subsets is not None:
                self._subsets.close()
                self._subsets = None
        if _sync:
            be.synchronize()
        self.time_ += time.perf_counter() - t0 - self._cb_time
        return self

    # round 5: no feature limit any more - the chunk call draws the subsets of wide problems ahead on a worker thread of
    # its own (somf_step.hip: DrawAhead; the bit-exact shuffle of p indices is ~5 ns per feature: 1 ms at p = 200 000,
    # as long as the device step), so that BASELINE config 5 with several ranks is ONE library call per chunk too
    CHUNK_CALL_MAX_FEATURES = None

    def _chunk_call_applies(self, be, sample_indices):
        """One library call per chunk instead of the Python loop over minibatches: the plain configuration only -
        no callback / verbose output between minibatches, no per-sample averages (their weights come from
        sample_n_iter_, which lives here), one rank or the library's own RCCL communicator, the numpy legacy
        generator as random_state, and the library's sampler (tests wrap it to record the draws)."""
        if not hasattr(be, 'fit_chunk') or getattr(self, '_python_loop', False):
            return False
        if self.callback is not None or self.verbose or getattr(self, '_two_phase', False):
            return False
        if self.G_agg == 'average' or self.Dx_agg == 'average' or \
                (self.CHUNK_CALL_MAX_FEATURES is not None and be.p > self.CHUNK_CALL_MAX_FEATURES):
            return False
        if not isinstance(self.random_state, np.random.RandomState) or not isinstance(self.feature_sampler_, Sampler):
            return False
        if (self._world() > 1 or getattr(self, '_force_reduce', False)) and not self._wants_native_rccl(be):
            return False
        return True

    def _wants_native_rccl(self, be):
        """Several ranks: is the head summed by the library's own RCCL communicator (one call per chunk of minibatches,
        the all-reduce on the compute stream) or by torch.distributed between two calls per minibatch?  `_native_rccl`
        = True / False decides; unset, the library's communicator is used whenever the process group runs RCCL
        (backend 'nccl') - gloo groups (tests, ranks sharing a GPU) keep torch's collective.  A communicator that
        cannot be created falls back to torch's, on every rank (HipBackend.native_comm)."""
        if not hasattr(be, 'native_comm') or getattr(be, '_comm_failed', False):
            return False
        flag = getattr(self, '_native_rccl', None)
        if flag is not None:
            return bool(flag)
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'

    def _native_comm(self, be):
        """the library's communicator for this call, or None (one rank without a forced reduction / torch's route)"""
        world = self._world()
        if world == 1 and not getattr(self, '_force_reduce', False):
            return None
        if not self._wants_native_rccl(be):
            return None
        return be.native_comm(_dist())

    def set_params(self, **params):
        """dict_fact.py:339-357: only a switch of G_agg to 'full' is honoured for
        G_agg (G_ is then computed from the current dictionary)."""
        G_agg = params.pop('G_agg', None)
        be = self.__dict__.get('_backend')
        if G_agg == 'full' and self.G_agg != 'full':
            self.G_agg = 'full'
            if be is not None and getattr(be, 'Dt', None) is not None:
                be.update_plan(self._plan_kwargs(be._desc_kw['max_batch']))
                be.full_gram()
        BaseEstimator.set_params(self, **params)
        if be is not None and getattr(be, 'Dt', None) is not None:
            be.update_plan(self._plan_kwargs(be._desc_kw['max_batch']))
        return self

    def shuffle(self):
        """Shuffle code_, G_average_, Dx_average_ rows; returns the permutation (dict_fact.py:359-379)."""
        random_seed = self.random_state.randint(MAX_INT)
        random_state = RandomState(random_seed)
        be = self._backend
        arrays = [_DeviceRows(be, 'code')]
        if self.G_agg == 'average':
            arrays.append(_DeviceRows(be, 'G_average'))
        if self.Dx_agg == 'average' and be.Dx_average is not None:
            arrays.append(_DeviceRows(be, 'Dx_average'))
        perm = random_state.shuffle_with_trace(arrays)
        self.labels_ = self.labels_[perm]
        return perm

    def prepare(self, n_samples=None, n_features=None, dtype=None, X=None):
        """Allocate and initialise the estimator state (dict_fact.py:381-489)."""
        if X is not None:
            if isinstance(X, torch.Tensor):
                X = X[:self.n_components].cpu().numpy()
            X = check_array(X, order='C', dtype=[np.float32, np.float64])
            if dtype is None:
                dtype = X.dtype
            if n_samples is None:
                n_samples = X.shape[0]
            if n_features is None:
                n_features = X.shape[1]
            elif n_features != X.shape[1]:
                raise ValueError('n_features and X does not match')
        else:
            if n_features is None or n_samples is None:
                raise ValueError('Either provide shape or data to function prepare.')
            if dtype is None:
                dtype = np.float64
        dtype = np.dtype(dtype)
        if dtype not in (np.float32, np.float64):
            raise ValueError('dtype should be float32 or float64')       # the reference returns it (:422)
        if self.optimizer not in ['variational', 'sgd']:
            raise ValueError("optimizer should be 'variational' or 'sgd'")
        if self.optimizer == 'sgd':
            self.reduction = 1
            self.G_agg = 'full'
            self.Dx_agg = 'full'
        k = self.n_components
        self._backend = be = self._make_backend()
        be.allocate(self._plan_kwargs(self.batch_size), n_samples, n_features, k, dtype)

        self.random_state = check_random_state(self.random_state)
        dist = _dist()
        if dist is not None:
            # every rank has to draw the same feature subsets, atom orders and seeds: all of them continue rank 0's
            # generator (with random_state=None each process would otherwise seed itself from the OS)
            state = [self.random_state.get_state()]
            dist.broadcast_object_list(state, src=0)
            self.random_state = np.random.RandomState()
            self.random_state.set_state(state[0])
        if X is None:
            D = np.empty((k, n_features), dtype=dtype)
            D[:, :] = self.random_state.randn(k, n_features)
        else:
            D = check_array(X[:k], dtype=dtype.type, copy=True)          # first k rows (:459-461)
            if D.shape[0] < k:
                raise ValueError('X should have at least n_components rows')
        if self.comp_pos:
            D[D <= 0] = -D[D <= 0]
        be.set_dictionary(D)
        be.scale_atoms(self.comp_l1_ratio, 1.0)
        if dist is not None:
            be.broadcast_dictionary(dist)                  # replicas start from rank 0's atoms, bit for bit
        self.labels_ = np.arange(n_samples)
        if self.G_agg == 'full':
            be.full_gram()
        self.n_iter_ = 0
        self.sample_n_iter_ = np.zeros(n_samples, dtype='int')
        self.random_state = check_random_state(self.random_state)
        random_seed = self.random_state.randint(MAX_INT)
        self.feature_sampler_ = Sampler(n_features, self.rand_size, self.replacement, random_seed)
        if self.verbose:
            self.verbose_iter_ = np.linspace(0, n_samples * self.n_epochs, self.verbose).tolist()
        self.time_ = 0
        self._cb_time = 0.0
        return self

    # ------------------------------------------------------------- internals
    def _callback(self):
        if self.callback is not None:
            self.callback(self)

    def _world(self):
        dist = _dist()
        return dist.get_world_size() if dist is not None else 1

    def _all_reduce(self, head):
        import torch.distributed as dist
        return dist.all_reduce(head, op=dist.ReduceOp.SUM)

    def _global_batch_sizes(self, n_local, n_batches):
        """Several ranks: rows of the global minibatch t = sum over the ranks of their t-th local batch (the last one
        may be ragged).  Every rank must run the same number of minibatches - the collectives of a step would
        otherwise wait for ever - which is checked here, once per partial_fit."""
        dist = _dist()
        if dist is None:
            return None
        world = dist.get_world_size()
        # the common case - every rank holds the same number of rows - is settled by ONE small all-reduce (max of
        # [n, -n]); only unequal counts need the full list (an object gather costs a millisecond with RCCL)
        probe = torch.tensor([int(n_local), -int(n_local)], dtype=torch.int64,
                             device=self._backend.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(probe, op=dist.ReduceOp.MAX)
        hi, lo = int(probe[0].item()), -int(probe[1].item())
        if hi == lo:
            counts = [hi] * world
        else:
            counts = [None] * world
            dist.all_gather_object(counts, int(n_local))
        b = self.batch_size
        nb = [int(ceil(c / b)) for c in counts]
        if len(set(nb)) != 1:
            raise ValueError('partial_fit: the ranks hold %s rows, i.e. %s minibatches of %d rows: every rank must '
                             'run the same number of minibatches' % (counts, nb, b))
        return [sum(min(b, c - t * b) for c in counts) for t in range(n_batches)]

    def _single_batch_fit(self, Xh, batch, sample_indices, b_global=None):
        """One SOMF iteration (dict_fact.py:495-526)."""
        if self.verbose and self.verbose_iter_ and self.n_iter_ >= self.verbose_iter_[0]:
            tc = time.perf_counter()
            print('Iteration %i' % self.n_iter_)
            self.verbose_iter_ = self.verbose_iter_[1:]
            self._callback()
            self._cb_time += time.perf_counter() - tc
        be = self._backend
        world = self._world()
        ahead = getattr(self, '_subsets', None)
        subset = ahead.next() if ahead is not None else self.feature_sampler_.yield_subset(self.reduction)
        batch_size = batch.stop - batch.start
        if b_global is None:
            b_global = batch_size * world
        self.n_iter_ += b_global
        self.sample_n_iter_[sample_indices] += 1
        w_sample = None
        if self.G_agg == 'average' or self.Dx_agg == 'average':
            w_sample = np.power(self.sample_n_iter_[sample_indices].astype(np.float64),
                                -self.sample_learning_rate).astype(be.dtype)
        w = batch_weight(self.n_iter_, b_global, self.learning_rate, 0)
        order = self.random_state.permutation(self.n_components)       # dict_fact.py:672
        if world == 1 and hasattr(be, 'step') and not getattr(self, '_two_phase', False):
            be.step(Xh, batch, sample_indices, subset, order, w_sample, w, self.reduction, b_global)
            return
        # Several ranks: every rank keeps its own partial C_ / B_ (both recursions are linear in the increments);
        # only what the dictionary update reads - C_ and the sampled rows of B_ - is summed over the ranks.
        if world > 1:
            self._stats_on_rank0 = False
        comm = self._native_comm(be) if hasattr(be, 'step_dist') else None
        if comm is not None:
            # the exchange inside the library (modl_somf_step_dist): RCCL called directly, on the compute stream
            be.step_dist(comm, Xh, batch, sample_indices, subset, order, w_sample, w, self.reduction, b_global)
            return
        head = be.phase1(Xh, batch, sample_indices, subset, order, w_sample, w, self.reduction, b_global)
        if world > 1 or getattr(self, '_force_reduce', False):       # (the latter: single-rank RCCL test of this path)
            self._all_reduce(head)
        be.phase2(head)

    # ------------------------------------------------- several ranks: state
    def consolidate_statistics(self):
        """COLLECTIVE (every rank calls it; a no-op with one rank).  With several ranks `C_` and `B_` are kept as
        per-rank partial sums; this sums them over the ranks, leaves the sums on rank 0 and zeros on the others - the
        sum over the ranks, which is all the next minibatch uses, is unchanged - so that rank 0's state IS the
        reference's state (dict_fact.py:116-124): a pickle written on rank 0 afterwards can be loaded anywhere."""
        dist = _dist()
        if dist is None:
            return self
        be, rank0 = self._backend, dist.get_rank() == 0
        on_device = dist.get_backend() == 'nccl' and isinstance(getattr(be, 'Bt', None), torch.Tensor)
        if on_device:
            be.synchronize()
            for t in (be.C, be.Bt):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                if not rank0:
                    t.zero_()
        else:
            C_sum, B_sum = _sum_over_ranks(be.get('C'), be.device), _sum_over_ranks(be.get_B(), be.device)
            be.set('C', C_sum if rank0 else np.zeros_like(C_sum))
            be.set_B(B_sum if rank0 else np.zeros_like(B_sum))
        self._stats_on_rank0 = True
        return self

    # ---------------------------------------------------------------- pickle
    def __getstate__(self):
        """Pickling is LOCAL (no collective: `if rank == 0: pickle.dump(est)` must not hang).  With several ranks the
        pickle therefore holds this rank's share of C_ / B_, and says so: unless consolidate_statistics() ran right
        before on every rank (then rank 0's share is the whole statistic), loading it anywhere but on the same rank
        of a world of the same size raises."""
        state = dict(self.__dict__)
        be = state.pop('_backend', None)
        state.pop('_subsets', None)
        if be is not None and getattr(be, 'Dt', None) is not None:
            be.synchronize()
            dist = _dist()
            world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
            whole = world == 1 or (rank == 0 and bool(state.get('_stats_on_rank0', False)))
            if not whole:
                import warnings
                warnings.warn('pickling rank %d of %d: C_ and B_ are per-rank partial sums; call '
                              'consolidate_statistics() on every rank first and pickle on rank 0 to get a checkpoint '
                              'that loads elsewhere' % (rank, world))
            state['_saved'] = dict(
                dtype=str(be.dtype), n=be.n, p=be.p, k=be.k, components_=be.get_dictionary(), B_=be.get_B(),
                C=be.get('C'), code=be.get('code'), comp_norm=be.get('comp_norm'), G=be.get('G'),
                Dx_average=be.get('Dx_average'), G_average=be.get('G_average'),
                partial_of=None if whole else (rank, world))
        state.pop('_stats_on_rank0', None)
        return state

    def __setstate__(self, state):
        saved = state.pop('_saved', None)
        self.__dict__.update(state)
        if saved is not None:
            dist = _dist()
            world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
            part = saved.get('partial_of')
            if part is not None and tuple(part) != (rank, world):
                raise ValueError('this pickle holds the share of rank %d of %d of the statistics C_ / B_ (it was written '
                                 'without consolidate_statistics()); it can only be loaded by that rank of a world of '
                                 'that size, not by rank %d of %d' % (part[0], part[1], rank, world))
            self._backend = be = self._make_backend()
            be.allocate(self._plan_kwargs(self.batch_size), saved['n'], saved['p'], saved['k'], np.dtype(saved['dtype']))
            be.set_dictionary(saved['components_'])
            # a whole statistic loaded by several ranks goes to rank 0 (zeros elsewhere): the sum over the ranks is it
            be.set_B(saved['B_'] if part is not None else _rank0_share(saved['B_']))
            be.set('C', saved['C'] if part is not None else _rank0_share(saved['C']))
            self._stats_on_rank0 = part is None and world > 1
            for name in ('code', 'comp_norm', 'G', 'Dx_average', 'G_average'):
                if saved[name] is not None:
                    be.set(name, saved[name])


class Coder(CodingMixin, BaseEstimator):
    """Transform-only estimator over a fixed dictionary (dict_fact.py:724-745)."""

    def __init__(self, dictionary, code_alpha=1, code_l1_ratio=1, tol=1e-2, max_iter=100, code_pos=False,
                 random_state=None, n_threads=1, device=None):
        self._set_coding_params(dictionary.shape[0], code_l1_ratio=code_l1_ratio, code_alpha=code_alpha,
                                code_pos=code_pos, random_state=random_state, tol=tol, max_iter=max_iter,
                                n_threads=n_threads)
        self.dictionary = dictionary
        self.device = device
        D = np.ascontiguousarray(dictionary)
        if D.dtype not in (np.float32, np.float64):
            D = D.astype(np.float64)
        self._backend = be = self._make_backend()
        be.allocate(self._plan_kwargs(min(4096, 1 << 20)), 1, D.shape[1], D.shape[0], D.dtype)
        be.set_dictionary(D)

    @property
    def components_(self):
        return self._backend.get_dictionary()

    def fit(self, X=None):
        return self
