"""RecsysDictFact: masked (missing-data) online matrix factorization on a CSR
rating matrix (reference: modl/decomposition/recsys.py:17-314).  Same
constructor, attributes and methods; the per-sample Python loop of the
reference (`_single_sample_update`, recsys.py:168-185) runs as batched GPU
kernels (csrc/recsys.hip), the dictionary update reuses the dense path's
block-coordinate kernels."""
import ctypes as C
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
        self.indptr = torch.from_numpy(X.indptr.astype(np.int32)).to(dev)
        self.indices = torch.from_numpy(X.indices.astype(np.int32)).to(dev)
        self.data = torch.from_numpy(np.ascontiguousarray(X.data, dtype=self.dtype)).to(dev)
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

    def batch_update(self, X, batch, alpha, w, n_iter, order):
        """One minibatch (recsys.py:147-165) for the CSR rows `batch` (in this order)."""
        dev, k = self.device, self.k
        self.codes(batch, alpha)
        rows_t = torch.from_numpy(np.ascontiguousarray(batch, dtype=np.int64)).to(dev)
        code_b = self.code.index_select(0, rows_t).contiguous()
        # the batch's entries grouped by feature, in batch order inside a feature
        starts, ends = X.indptr[batch], X.indptr[batch + 1]
        lens = ends - starts
        if lens.sum() > 0:
            pos = np.repeat(np.arange(len(batch), dtype=np.int32), lens)
            flat = np.concatenate([np.arange(s, e) for s, e in zip(starts, ends)])
            cols = X.indices[flat]
            o = np.argsort(cols, kind='stable')
            subset, counts = np.unique(cols[o], return_counts=True)
            fptr = np.zeros(len(subset) + 1, dtype=np.int32)
            fptr[1:] = np.cumsum(counts)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
            d_subset, d_fptr = t(subset, np.int32), t(fptr, np.int32)
            d_es, d_ev = t(pos[o], np.int32), t(X.data[flat][o], self.dtype)
            f = getattr(lib, 'modl_recsys_update_B_' + sfx(self.dtype))
            check(f(ptr(self.Bt), k, ptr(self.feature_n_iter), ptr(d_subset), ptr(d_fptr), ptr(d_es), ptr(d_ev),
                    ptr(code_b), float(w) * float(n_iter), len(subset), stream_ptr(dev)), 'modl_recsys_update_B')
        else:
            subset, d_subset = np.zeros(0, dtype=np.int32), None
        ct = C.c_float if self.dtype == np.float32 else C.c_double
        f = getattr(lib, 'modl_gram_axpby_' + sfx(self.dtype))
        check(f(ptr(code_b), len(batch), k, ptr(self.C), ct(1 - w), ct(w / len(batch)), stream_ptr(dev)),
              'modl_gram_axpby')
        if len(subset):
            d_order = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(dev)
            h_order = np.ascontiguousarray(order, dtype=np.int64)
            f = getattr(lib, 'modl_dict_update_' + sfx(self.dtype))
            check(f(ptr(self.Dt), ptr(self.Bt), ptr(self.C), ptr(self.comp_norm), ptr(d_subset), len(subset),
                    ptr(d_order), h_order.ctypes.data_as(C.c_void_p), k, 0, 0, 0.0, float(w), 1.0, ptr(self.ws),
                    self.ws_bytes, stream_ptr(dev)), 'modl_dict_update')
            torch.cuda.synchronize(dev)          # h_order / staging tensors must outlive the launches

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
            for batch in gen_batches(n_samples, batch_size):
                self._single_batch_fit(X, permutation[batch])
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
        self._dev.batch_update(X, batch, self.alpha, w, self.n_iter_, order)

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
