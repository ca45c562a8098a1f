"""RecsysDictFact: masked (missing-data) online matrix factorization on a CSR
rating matrix (reference: modl/decomposition/recsys.py:17-314).  Same
constructor, attributes and methods; the per-sample Python loop of the
reference (`_single_sample_update`, recsys.py:168-185) runs as batched GPU
kernels (csrc/recsys.hip), the dictionary update reuses the dense path's
block-coordinate kernels."""
import ctypes as C
import os
from math import log, ceil

import numpy as np
import scipy.sparse as sp
import torch
from sklearn.base import BaseEstimator
from sklearn.utils import check_array, check_random_state, gen_batches

from ._lib import lib, check
from .device import default_device, dtype_id, sfx, torch_dtype, ptr, stream_ptr, to_device, transpose_to
from .randomkit import batch_weight


def compute_biases(X, beta=0, inplace=False):
    """Row / column centring of a CSR matrix (recsys.py:268-306), host code."""
    if not inplace:
        X = X.copy()
    X = sp.csr_matrix(X)
    acc_u, acc_m = np.zeros(X.shape[0]), np.zeros(X.shape[1])
    n_u, n_m = X.getnnz(axis=1), X.getnnz(axis=0)
    n_u[n_u == 0] = 1
    n_m[n_m == 0] = 1
    average_rating = np.mean(X.data)
    for _ in range(2):
        w_u = (np.asarray(X.sum(axis=1))[:, 0] + average_rating * beta) / (n_u + beta)
        X.data -= np.repeat(w_u, np.diff(X.indptr))
        w_m = np.asarray(X.sum(axis=0))[0] / (n_m + beta)
        X.data -= w_m.take(X.indices, mode='clip')
        acc_u += w_u
        acc_m += w_m
    return acc_u, acc_m


def rmse(X_true, X_pred):
    """recsys.py:309-314"""
    X_true = check_array(X_true, accept_sparse='csr')
    X_pred = check_array(X_pred, accept_sparse='csr')
    return np.sqrt(np.mean((X_true.data - X_pred.data) ** 2))


class _RecsysDevice:
    """Device state and launches of one RecsysDictFact."""

    def __init__(self, X, k, dtype, device=None):
        self.device = torch.device(device) if device is not None else default_device()
        self.dtype = np.dtype(dtype)
        self.n, self.p = X.shape
        self.k = k
        dev = self.device
        self.plan, self.plan_batch = None, 0
        # the CSR matrix on both sides: the host copy feeds the per-batch grouping, the device copy the kernels
        self.h_indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
        self.h_indices = np.ascontiguousarray(X.indices, dtype=np.int32)
        self.h_data = np.ascontiguousarray(X.data, dtype=self.dtype)
        self.indptr = torch.from_numpy(self.h_indptr).to(dev)
        self.indices = torch.from_numpy(self.h_indices).to(dev)
        self.data = torch.from_numpy(self.h_data).to(dev)
        td = torch_dtype(self.dtype)
        self.Dt = torch.zeros((self.p, k), dtype=td, device=dev)
        self.Bt = torch.zeros((self.p, k), dtype=td, device=dev)
        self.C = torch.zeros((k, k), dtype=td, device=dev)
        self.code = torch.zeros((self.n, k), dtype=td, device=dev)
        self.comp_norm = torch.zeros(k, dtype=td, device=dev)
        self.feature_n_iter = torch.zeros(self.p, dtype=torch.int64, device=dev)
        nbytes = lib.modl_dict_update_workspace(dtype_id(self.dtype), self.p, k)
        self.ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self.ws_bytes = nbytes

    def set_dictionary(self, D):
        self.Dt = transpose_to(to_device(D, self.device, dtype=self.dtype), self.k, self.p)

    def get_dictionary(self):
        return transpose_to(self.Dt, self.p, self.k).cpu().numpy()

    def codes(self, rows, alpha):
        """ridge codes of the CSR rows `rows` (None = all), written to code[rows]"""
        f = getattr(lib, 'modl_recsys_codes_' + sfx(self.dtype))
        if rows is None:
            check(f(ptr(self.Dt), self.p, self.k, ptr(self.indptr), ptr(self.indices), ptr(self.data), None, None,
                    self.n, float(alpha), ptr(self.code), stream_ptr(self.device)), 'modl_recsys_codes')
        else:
            r = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(self.device)
            check(f(ptr(self.Dt), self.p, self.k, ptr(self.indptr), ptr(self.indices), ptr(self.data), ptr(r), None,
                    len(rows), float(alpha), ptr(self.code), stream_ptr(self.device)), 'modl_recsys_codes')

    def batch_update(self, batch, alpha, w, n_iter, order):
        """One minibatch (recsys.py:147-165) for the CSR rows `batch` (in this order): one asynchronous call
        (modl_recsys_minibatch_*: host grouping of the batch's ratings by item, staging, codes, B_, C_, dictionary)."""
        batch = np.ascontiguousarray(batch, dtype=np.int64)
        order = np.ascontiguousarray(order, dtype=np.int64)
        if self.plan is None or len(batch) > self.plan_batch:
            self._make_plan(len(batch))
        f = getattr(lib, 'modl_recsys_minibatch_' + sfx(self.dtype))
        check(f(self.plan, self.h_indptr.ctypes.data_as(C.c_void_p), self.h_indices.ctypes.data_as(C.c_void_p),
                self.h_data.ctypes.data_as(C.c_void_p), self.n, ptr(self.indptr), ptr(self.indices), ptr(self.data),
                batch.ctypes.data_as(C.c_void_p), len(batch), order.ctypes.data_as(C.c_void_p), float(alpha), float(w),
                float(n_iter), ptr(self.Dt), ptr(self.Bt), ptr(self.C), ptr(self.code), ptr(self.comp_norm),
                ptr(self.feature_n_iter), stream_ptr(self.device)), 'modl_recsys_minibatch')

    def fit_batches(self, rows, batch_size, alpha, learning_rate, n_iter, np_random_state):
        """A run of minibatches in ONE call (modl_recsys_fit_batches_*: the host loop of recsys.py:135-139 behind the ABI).
        `rows`: the permuted row ids of the run; the atom orders are drawn inside the library by a generator loaded with
        numpy's legacy MT19937 state and handed back afterwards, so `np_random_state` continues as if it had drawn them
        (recsys.py:196).  Returns the new n_iter_."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if self.plan is None or batch_size > self.plan_batch:
            self._make_plan(batch_size)
        if getattr(self, '_order_rk', None) is None:
            h = C.c_void_p()
            check(lib.modl_rk_create(0, C.byref(h)), 'modl_rk_create')
            self._order_rk = h
        kind, key, pos, has_gauss, cached = np_random_state.get_state()
        key = np.ascontiguousarray(key, dtype=np.uint32)
        check(lib.modl_rk_set_mt_state(self._order_rk, key.ctypes.data_as(C.c_void_p), int(pos)), 'modl_rk_set_mt_state')
        n = C.c_int64(int(n_iter))
        done = C.c_int64(0)
        f = getattr(lib, 'modl_recsys_fit_batches_' + sfx(self.dtype))
        try:
            rc = f(self.plan, self.h_indptr.ctypes.data_as(C.c_void_p), self.h_indices.ctypes.data_as(C.c_void_p),
                   self.h_data.ctypes.data_as(C.c_void_p), self.n, ptr(self.indptr), ptr(self.indices), ptr(self.data),
                   rows.ctypes.data_as(C.c_void_p), len(rows), int(batch_size), self._order_rk, float(alpha),
                   float(learning_rate), C.byref(n), ptr(self.Dt), ptr(self.Bt), ptr(self.C), ptr(self.code),
                   ptr(self.comp_norm), ptr(self.feature_n_iter), stream_ptr(self.device), C.byref(done))
        finally:
            out = np.empty(624, dtype=np.uint32)
            p2 = C.c_int32()
            check(lib.modl_rk_get_mt_state(self._order_rk, out.ctypes.data_as(C.c_void_p), C.byref(p2)))
            np_random_state.set_state((kind, out, int(p2.value), has_gauss, cached))
        check(rc, 'modl_recsys_fit_batches')
        return int(n.value)

    def launch_counts(self):
        """(minibatches that ran as one launch, minibatches that ran as separate launches) of the current plan"""
        a, b = C.c_int64(0), C.c_int64(0)
        if self.plan is not None:
            check(lib.modl_recsys_plan_counts(self.plan, C.byref(a), C.byref(b)), 'modl_recsys_plan_counts')
        return int(a.value), int(b.value)

    def _make_plan(self, batch_size):
        self._free_plan()
        # the largest number of ratings a batch of this size can hold: its batch_size longest rows
        lens = np.sort(np.diff(self.h_indptr))[::-1]
        max_entries = int(lens[:batch_size].sum())
        h = C.c_void_p()
        self._pid = os.getpid()
        with torch.cuda.device(self.device):
            check(lib.modl_recsys_plan_create(dtype_id(self.dtype), self.p, self.k, batch_size, max_entries, C.byref(h)),
                  'modl_recsys_plan_create')
        self.plan, self.plan_batch = h, batch_size

    def _free_plan(self):
        # (a forked child - e.g. multiprocessing's helpers - must never release the parent's device objects)
        if getattr(self, 'plan', None) and getattr(self, '_pid', os.getpid()) == os.getpid():
            lib.modl_recsys_plan_destroy(self.plan)
            if getattr(self, '_order_rk', None):
                lib.modl_rk_destroy(self._order_rk)
                self._order_rk = None
        self.plan, self.plan_batch = None, 0

    def __del__(self):
        try:
            self._free_plan()
        except Exception:
            pass

    def predict(self, Xp):
        dev = self.device
        out = torch.zeros(Xp.nnz, dtype=torch.float64, device=dev)
        ind = torch.from_numpy(Xp.indices.astype(np.int32)).to(dev)
        iptr = torch.from_numpy(Xp.indptr.astype(np.int32)).to(dev)
        f = getattr(lib, 'modl_recsys_predict_' + sfx(self.dtype))
        check(f(ptr(out), ptr(ind), ptr(iptr), ptr(self.code), Xp.shape[0], self.k, ptr(self.Dt), stream_ptr(dev)),
              'modl_recsys_predict')
        return out.cpu().numpy()


class RecsysDictFact(BaseEstimator):
    """Matrix factorization with missing data by masked online dictionary learning
    (recsys.py:17-79 for the parameters)."""

    def __init__(self, alpha=1.0, beta=.0, n_components=30, learning_rate=1., batch_size=1, dict_init=None,
                 l1_ratio=0, n_epochs=1, random_state=None, verbose=0, detrend=False, crop=None, callback=None,
                 device=None):
        self.callback = callback
        self.verbose = verbose
        self.random_state = random_state
        self.n_epochs = n_epochs
        self.l1_ratio = l1_ratio
        self.dict_init = dict_init
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.n_components = n_components
        self.alpha = alpha
        self.beta = beta
        self.detrend = detrend
        self.crop = crop
        self.device = device

    # device-resident attributes
    @property
    def components_(self):
        return self._dev.get_dictionary()

    @property
    def code_(self):
        return self._dev.code.cpu().numpy()

    @property
    def C_(self):
        return self._dev.C.cpu().numpy()

    @property
    def B_(self):
        return transpose_to(self._dev.Bt, self._dev.p, self._dev.k).cpu().numpy()

    @property
    def comp_norm_(self):
        return self._dev.comp_norm.cpu().numpy()

    @property
    def feature_n_iter_(self):
        return self._dev.feature_n_iter.cpu().numpy()

    def fit(self, X, y=None):
        """recsys.py:81-141"""
        if not sp.issparse(X):
            X = sp.csr_matrix(X)
        X = check_array(X, accept_sparse='csr', dtype=[np.float32, np.float64], copy=True)
        dtype = X.dtype
        n_samples, n_features = X.shape
        self.random_state = check_random_state(self.random_state)
        if self.detrend:
            self.row_mean_, self.col_mean_ = compute_biases(X, beta=self.beta, inplace=False)
            X.data -= np.repeat(self.row_mean_, np.diff(X.indptr)).astype(dtype)
            X.data -= self.col_mean_.take(X.indices, mode='clip').astype(dtype)
        D = self.random_state.randn(self.n_components, n_features).astype(dtype)
        D /= np.sqrt(np.sum(D ** 2, axis=1))[:, np.newaxis]
        self._dev = dev = _RecsysDevice(X, self.n_components, dtype, self.device)
        dev.set_dictionary(D)
        self._refit()
        self.feature_freq_ = np.bincount(X.indices, minlength=n_features) / n_samples
        sparsity = X.nnz / n_samples / n_features
        batch_size = int(ceil(1. / sparsity)) if self.batch_size is None else self.batch_size
        self.n_iter_ = 0
        if self.verbose:
            log_lim = log(n_samples * self.n_epochs / batch_size, 10)
            self.verbose_iter_ = ((np.logspace(0, log_lim, self.verbose, base=10) - 1) * batch_size).tolist()
        for _ in range(self.n_epochs):
            permutation = self.random_state.permutation(n_samples)
            if self.verbose or self.callback is not None or not hasattr(self.random_state, 'get_state'):
                for batch in gen_batches(n_samples, batch_size):
                    self._single_batch_fit(X, permutation[batch])
            else:
                # nothing to report between minibatches: the epoch's host loop runs behind the ABI, one call
                self.n_iter_ = dev.fit_batches(permutation, batch_size, self.alpha, self.learning_rate, self.n_iter_,
                                               self.random_state)
        if dev.plan is not None:         # (a dictionary-update launch whose workgroups could not meet must not pass for a fit)
            check(lib.modl_recsys_plan_status(dev.plan, stream_ptr(dev.device)), 'modl_recsys_plan_status')
        self._refit()
        return self

    def _callback(self):
        if self.callback is not None:
            self.callback(self)

    def _single_batch_fit(self, X, batch):
        """recsys.py:147-165"""
        if self.verbose and self.verbose_iter_ and self.n_iter_ >= self.verbose_iter_[0]:
            print('Iteration %i' % self.n_iter_)
            self.verbose_iter_ = self.verbose_iter_[1:]
            self._callback()
        batch_size = batch.shape[0]
        self.n_iter_ += batch_size
        w = batch_weight(self.n_iter_, batch_size, self.learning_rate, 0)
        order = self.random_state.permutation(self.n_components)     # recsys.py:196 (drawn for every batch)
        self._dev.batch_update(batch, self.alpha, w, self.n_iter_, order)

    def _refit(self):
        """recsys.py:254-265: ridge codes of every row with the current dictionary"""
        self._dev.codes(None, self.alpha)

    def predict(self, X):
        """recsys.py:215-245"""
        if not sp.issparse(X):
            X = sp.csr_matrix(X)
        X = check_array(X, accept_sparse='csr')
        out = self._dev.predict(X)
        if self.detrend:
            out += np.repeat(self.row_mean_, np.diff(X.indptr))
            out += self.col_mean_.take(X.indices, mode='clip')
        if self.crop is not None:
            out[out > self.crop[1]] = self.crop[1]
            out[out < self.crop[0]] = self.crop[0]
        return sp.csr_matrix((out, X.indices, X.indptr), shape=X.shape)

    def score(self, X):
        """Root mean squared error of the prediction at the loci of X (recsys.py:247-252)"""
        if not sp.issparse(X):
            X = sp.csr_matrix(X)
        X = check_array(X, accept_sparse='csr')
        return rmse(X, self.predict(X))
