"""fMRIDictFact on raw 2-D records: the streaming loop of the reference's
fMRI estimator (modl/decomposition/fmri.py:423-546 `_compute_components`, :549-556
`_flip`, :364-367 the ridge `Coder`) without the nilearn masking layer, which is
IO and out of scope.  A record is what the
reference's `MultiRawMasker` yields (modl/input_data/fmri/unmask.py:37-55): a
2-D array (time points x voxels) or the path of a .npy file holding one.

The reference's loop rebinds `method` to the dict of aggregation modes (fmri.py:460), so its later tests on the
method NAME - the 'gram' switch to G_agg='full' / Dx_agg='average' at epoch 5 (:508), the shrinking reduction of
'reducing ratio' (:511-513) and the per-record sample_indices of 'average' / 'gram' (:535-539) - never fire.  The
default here reproduces that EFFECTIVE behaviour (same results as the reference on the same inputs, pinned by
tests/golden/fmri.npz); `intended_schedules=True` runs what the code was written to do."""
import time
from concurrent.futures import ThreadPoolExecutor
from math import sqrt
from os.path import join

import numpy as np
import torch
from sklearn.base import BaseEstimator
from sklearn.utils import check_random_state

from .device import gather_rows
from .dict_fact import DictFact, Coder

METHODS = {'masked': {'G_agg': 'masked', 'Dx_agg': 'masked'},           # fmri.py:440-445
           'dictionary only': {'G_agg': 'full', 'Dx_agg': 'full'},
           'gram': {'G_agg': 'masked', 'Dx_agg': 'masked'},             # first epochs; switched at epoch 5
           'average': {'G_agg': 'average', 'Dx_agg': 'average'},
           'reducing ratio': {'G_agg': 'masked', 'Dx_agg': 'masked'}}


def _load(record, mmap=False):
    if isinstance(record, str):
        return np.load(record, mmap_mode='r' if mmap else None)
    return record


def _flip(components):
    """Flip a map when its negative part is larger (fmri.py:549-556)."""
    components = components.copy()
    for component in components:
        if np.sum(component < 0) > np.sum(component > 0):
            component *= -1
    return components


class _RecordStager:
    """Double-buffered record streaming: while a record is being fitted, a worker thread loads the
    next one (np.load of the .npy file, the dtype conversion of fmri.py:533) into pinned host memory and copies it to
    HBM on a side stream; the row permutation of fmri.py:535-541 is then a gather on the device.  Backends without a
    GPU (the oracle-backed test backend) get the plain host path."""

    def __init__(self, dict_fact, dtype):
        be = dict_fact._backend
        self.device = getattr(be, 'device', None)
        self.on_gpu = self.device is not None and self.device.type == 'cuda'
        self.dtype = np.dtype(dtype)
        self.pool = ThreadPoolExecutor(1) if self.on_gpu else None
        self.stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.pending = None

    def _stage(self, record):
        arr = np.asarray(_load(record))
        if not self.on_gpu:
            return arr.astype(self.dtype)
        host = torch.empty(arr.shape, dtype=torch.float32 if self.dtype == np.float32 else torch.float64, pin_memory=True)
        np.copyto(host.numpy(), arr, casting='unsafe')               # load + dtype conversion straight into pinned memory
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        return dev, done, host                                       # (host kept alive until the copy is consumed)

    def prefetch(self, record):
        self.pending = self.pool.submit(self._stage, record) if self.on_gpu else record

    def take(self):
        if not self.on_gpu:
            return self._stage(self.pending)
        dev, done, _host = self.pending.result()
        torch.cuda.current_stream(self.device).wait_event(done)
        return dev

    def rows(self, data, permutation):
        if not self.on_gpu:
            return data[permutation]
        return gather_rows(data, permutation)

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)


class fMRIDictFact(BaseEstimator):
    """Constructor arguments follow fmri.py:273-291 (masking arguments dropped)."""

    _dict_fact_class = DictFact
    _coder_class = Coder

    def __init__(self, method='masked', step_size=1, n_components=20, n_epochs=1, alpha=0.1, dict_init=None,
                 random_state=None, batch_size=20, reduction=1, learning_rate=1, positive=False, verbose=0,
                 callback=None, n_jobs=1, intended_schedules=False):
        self.intended_schedules = intended_schedules
        self.method = method
        self.step_size = step_size
        self.n_components = n_components
        self.n_epochs = n_epochs
        self.alpha = alpha
        self.dict_init = dict_init
        self.random_state = random_state
        self.batch_size = batch_size
        self.reduction = reduction
        self.learning_rate = learning_rate
        self.positive = positive
        self.verbose = verbose
        self.callback = callback
        self.n_jobs = n_jobs

    def fit(self, records, y=None):
        """records: list of 2-D arrays / .npy paths.  fmri.py:423-546."""
        if records is None:
            raise ValueError('records is None, use Coder instead')
        n_components = self.n_components
        dict_init = self.dict_init
        if dict_init is not None:                                    # fmri.py:415-416, 468-469
            dict_init = np.asarray(dict_init)[:n_components]
            n_components = dict_init.shape[0]
        random_state = check_random_state(self.random_state)
        reduction = self.reduction
        if self.method == 'sgd':
            optimizer, G_agg, Dx_agg, reduction = 'sgd', 'full', 'full', 1
        else:
            G_agg, Dx_agg = METHODS[self.method]['G_agg'], METHODS[self.method]['Dx_agg']
            optimizer = 'variational'
        n_records = len(records)
        lengths, dtype, n_voxels = [], None, None
        for rec in records:                                          # _lazy_scan, fmri.py:559-575
            arr = _load(rec, mmap=True)
            lengths.append(arr.shape[0])
            dtype, n_voxels = arr.dtype, arr.shape[1]
        indices_list = np.zeros(n_records + 1, dtype='int')
        indices_list[1:] = np.cumsum(lengths)
        n_samples = int(indices_list[-1]) + 1                        # fmri.py:476 (the + 1 is the reference's)
        if dtype not in (np.float32, np.float64):
            dtype = np.dtype(np.float64)
        dict_fact = self._dict_fact_class(
            n_components=n_components, code_alpha=self.alpha, code_l1_ratio=0, comp_l1_ratio=1,
            comp_pos=self.positive, reduction=reduction, Dx_agg=Dx_agg, optimizer=optimizer, step_size=self.step_size,
            G_agg=G_agg, learning_rate=self.learning_rate, batch_size=self.batch_size, random_state=random_state,
            n_threads=self.n_jobs, verbose=0)
        dict_fact.prepare(n_samples=n_samples, n_features=n_voxels,
                          X=None if dict_init is None else dict_init.astype(dtype), dtype=dtype)
        self.cpu_time_, self.io_time_ = 0.0, 0.0
        stager = _RecordStager(dict_fact, dtype)
        if n_records > 0:
            verbose_iter_ = np.linspace(0, n_records * self.n_epochs, self.verbose).tolist() if self.verbose else []
            current_n_records = 0
            named = self.method if self.intended_schedules else None      # fmri.py:460 (see the module docstring)
            for i in range(self.n_epochs):
                if named == 'gram' and i == 5:
                    dict_fact.set_params(G_agg='full', Dx_agg='average')
                if named == 'reducing ratio':
                    reduction = 1 + (reduction - 1) / sqrt(i + 1)    # compounds across epochs (fmri.py:511-513)
                    dict_fact.set_params(reduction=reduction)
                record_list = random_state.permutation(n_records)
                stager.prefetch(records[record_list[0]])
                for pos, record in enumerate(record_list):
                    if self.verbose and verbose_iter_ and current_n_records >= verbose_iter_[0]:
                        print('Record %i' % current_n_records)
                        if self.callback is not None:
                            self.callback(self, dict_fact, self.cpu_time_, self.io_time_)
                        verbose_iter_ = verbose_iter_[1:]
                    t0 = time.perf_counter()
                    data = stager.take()                             # staged while the previous record was fitted
                    if pos + 1 < n_records:
                        stager.prefetch(records[record_list[pos + 1]])
                    self.io_time_ += time.perf_counter() - t0
                    t0 = time.perf_counter()
                    permutation = random_state.permutation(data.shape[0])
                    if named in ['average', 'gram']:
                        sample_indices = np.arange(indices_list[record], indices_list[record + 1])[permutation]
                    else:
                        sample_indices = None
                    dict_fact.partial_fit(stager.rows(data, permutation), sample_indices=sample_indices)
                    current_n_records += 1
                    self.cpu_time_ += time.perf_counter() - t0
        stager.close()
        self.dict_fact_ = dict_fact
        self.components_ = _flip(dict_fact.components_)
        self.coder_ = self._coder_class(dictionary=self.components_, code_alpha=self.alpha, code_l1_ratio=0).fit()
        return self

    def transform(self, records):
        """Loadings of each record on the learned maps (fmri.py:131-164)."""
        return [self.coder_.transform(np.asarray(_load(r))) for r in records]

    def score(self, records):
        """Length-weighted mean objective (fmri.py:95-129)."""
        arrays = [np.asarray(_load(r)) for r in records]
        scores = np.array([self.coder_.score(a) for a in arrays])
        lens = np.array([a.shape[0] for a in arrays])
        return np.sum(scores * lens) / np.sum(lens)


class fMRICoder(BaseEstimator):
    """Loadings / objective of raw records on a FIXED set of maps: the reference's fMRICoder (fmri.py:371-402 over
    fMRICoderMixin.fit / score / transform, :76-164) on masked 2-D records (arrays or .npy paths, the MultiRawMasker
    contract of input_data/fmri/unmask.py:37-55) - the masking arguments are dropped as for fMRIDictFact, the rest of
    the constructor is the reference's.  `dictionary`: (n_components, n_voxels) array, or a .npy path."""

    _coder_class = Coder

    def __init__(self, dictionary, alpha=0.1, transform_batch_size=None, n_components=None, n_jobs=1, verbose=0):
        self.dictionary = dictionary
        self.alpha = alpha
        self.transform_batch_size = transform_batch_size
        self.n_components = n_components
        self.n_jobs = n_jobs
        self.verbose = verbose

    def fit(self, records=None, y=None):
        """fmri.py:76-93: the maps are `dictionary` (its first n_components rows, _check_dict_init :405-420); nothing is
        learned.  Returns self (the reference's mixin returns None: a slip, sklearn's contract kept here)."""
        D = np.asarray(_load(self.dictionary))
        if D.ndim != 2:
            raise ValueError('dictionary must be (n_components, n_voxels), got shape %s' % (D.shape,))
        if self.n_components is not None:
            D = D[:self.n_components]
        if D.dtype not in (np.float32, np.float64):
            D = D.astype(np.float64)
        self.components_ = np.ascontiguousarray(D)
        self.coder_ = self._coder_class(dictionary=self.components_, code_alpha=self.alpha, code_l1_ratio=0).fit()
        return self

    def _records(self, records):
        if isinstance(records, str) or (isinstance(records, np.ndarray) and records.ndim == 2):
            records = [records]                                      # one record (fmri.py:117, :150)
        return records

    def _rows(self, record):
        X = np.asarray(_load(record))
        if X.shape[1] != self.components_.shape[1]:
            raise ValueError('record has %d voxels, the maps have %d' % (X.shape[1], self.components_.shape[1]))
        return np.ascontiguousarray(X, dtype=self.components_.dtype)

    def transform(self, records):
        """Loadings of each record, one (n_samples, n_components) array per record (fmri.py:131-164).  A record is
        coded in slices of transform_batch_size rows when that is set (the codes of a row do not depend on the others)."""
        if not hasattr(self, 'coder_'):
            raise ValueError('fMRICoder is not fitted: call fit() first')
        out = []
        for rec in self._records(records):
            X = self._rows(rec)
            step = self.transform_batch_size or X.shape[0] or 1
            parts = [self.coder_.transform(X[a:a + step]) for a in range(0, X.shape[0], step)]
            out.append(np.concatenate(parts) if parts else np.zeros((0, self.components_.shape[0]), dtype=X.dtype))
        return out

    def score(self, records):
        """Length-weighted mean of the objective over the records (fmri.py:95-129); lower is a better fit."""
        if not hasattr(self, 'coder_'):
            raise ValueError('fMRICoder is not fitted: call fit() first')
        arrays = [self._rows(r) for r in self._records(records)]
        scores = np.array([self.coder_.score(a) for a in arrays])
        lens = np.array([a.shape[0] for a in arrays])
        return float(np.sum(scores * lens) / np.sum(lens))


class rfMRIDictionaryScorer:
    """Callback computing the test objective along a fit (fmri.py:588-633), on raw 2-D test records.

    The test records are staged in HBM ONCE (first call: np.load / dtype conversion, one upload per record) and
    every later call scores them where they are - codes from the current dictionary (`DictFact.score`: transform +
    the three sums of the objective on the device), so neither the test set nor the dictionary crosses the host
    link again; three doubles per record come back.  Signature of the reference: `scorer(masker, dict_fact,
    cpu_time, io_time)`; the first argument (the masker there, the fMRIDictFact here) is not used for raw records.
    `artifact_dir`: `info.pkl` as in the reference (:621-625) and the flipped maps as `components_<n_iter>.npy`
    (the reference writes a NIfTI image through the masker, which is out of scope)."""

    def __init__(self, test_records, test_confounds=None, info=None, artifact_dir=None):
        self.start_time = time.perf_counter()
        self.test_records = test_records
        self.test_confounds = test_confounds             # kept for signature compatibility; raw records carry none
        self.test_time = 0
        self.score = []
        self.iter = []
        self.time = []
        self.cpu_time = []
        self.io_time = []
        self.info = info
        self.artifact_dir = artifact_dir

    def _stage(self, dict_fact):
        be = dict_fact._backend
        self.data = [be.stage_X(np.ascontiguousarray(np.asarray(_load(r)), dtype=be.dtype)) for r in self.test_records]
        self.lengths = np.array([d.shape[0] for d in self.data])

    def __call__(self, masker, dict_fact, cpu_time, io_time):
        test_time = time.perf_counter()
        if not hasattr(self, 'data'):
            self._stage(dict_fact)
        scores = np.array([dict_fact.score(data) for data in self.data])
        score = np.sum(scores * self.lengths) / np.sum(self.lengths)
        self.test_time += time.perf_counter() - test_time
        this_time = time.perf_counter() - self.start_time - self.test_time
        self.score.append(score)
        self.time.append(this_time)
        self.cpu_time.append(cpu_time)
        self.io_time.append(io_time)
        self.iter.append(dict_fact.n_iter_)
        if self.info is not None:
            self.info['time'] = self.cpu_time
            self.info['score'] = self.score
            self.info['iter'] = self.iter
            if self.artifact_dir is not None:
                from joblib import dump
                dump(self.info, join(self.artifact_dir, 'info.pkl'))
        if self.artifact_dir is not None:
            np.save(join(self.artifact_dir, 'components_%i.npy' % dict_fact.n_iter_), _flip(dict_fact.components_))
