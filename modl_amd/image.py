"""ImageDictFact: dictionary learning on image patches (reference:
modl/decomposition/image.py:13-224), driving modl_amd.DictFact's
prepare / partial_fit / shuffle / set_params exactly as the reference does
(image.py:96-152).  The image is uploaded once and stays resident in HBM; every
buffer of patches is gathered, centred, normalised and flattened by one HIP launch
(modl_image_patches_*, csrc/image.hip: modl/feature_extraction/image.py:54-63 +
modl/input_data/image.py:4-23 fused) and handed to the SOMF step as a device tensor.
The patch-origin lists (fill / clean_mask, image_fast.pyx:12-74) are integer host
work behind the same C-ABI.  The numpy `scale_patches` below serves the public
transform / score on patches the caller holds in host memory."""
import ctypes as C
import time
from math import sqrt

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from sklearn.base import BaseEstimator
from sklearn.utils import check_random_state, gen_batches

from ._lib import lib, check
from .dict_fact import DictFact


def scale_patches(X, with_mean=True, with_std=True, channel_wise=True, copy=True):
    """Centred / l2-normalised patches, X: (n, ph, pw, c).  Same arithmetic, in the same order, as the reference's
    modl/input_data/image.py:4-23 (statistics per patch and channel over the ph x pw positions, the norm times
    sqrt(c); or per patch over everything when not channel_wise) written once over a reduction-axes tuple; the
    training path never calls this - its patches are produced already scaled by modl_image_patches_* on the device."""
    X = np.array(X, copy=True) if copy else X
    axes = (1, 2) if channel_wise else (1, 2, 3)
    if with_mean:
        X -= X.mean(axis=axes, keepdims=True)
    if with_std:
        norm = np.sqrt(np.square(X).sum(axis=axes, keepdims=True))
        norm[norm == 0] = 1
        if channel_wise:
            norm = norm * sqrt(X.shape[3])
        X /= norm
    return X


def fill(p, q, r):
    """All patch coordinates in C order (image_fast.pyx:59-74)."""
    out = np.empty((p * q * r, 3), dtype=np.int64)
    check(lib.modl_image_fill(p, q, r, out.ctypes.data_as(C.c_void_p)), 'modl_image_fill')
    return out


def clean_mask(patches, image):
    """Coordinates of the patches without any missing (-1) pixel (image_fast.pyx:12-57)."""
    x, y, z = patches.shape[3:]
    if image.dtype not in (np.float32, np.float64):
        raise TypeError('clean_mask: float32 or float64 image expected (fused `floating`, image_fast.pyx:12)')
    image = np.ascontiguousarray(image)
    H, W, Cc = image.shape
    f = getattr(lib, 'modl_image_clean_mask_' + ('f32' if image.dtype == np.float32 else 'f64'))
    out = np.empty(((H - x + 1) * (W - y + 1) * (Cc - z + 1), 3), dtype=np.int64)
    n = C.c_int64()
    check(f(image.ctypes.data_as(C.c_void_p), H, W, Cc, x, y, z, out.ctypes.data_as(C.c_void_p), C.byref(n)),
          'modl_image_clean_mask')
    return out[:n.value].copy()


class LazyCleanPatchExtractor(BaseEstimator):
    """Patch origins of an image, drawn once, patches produced on demand (the estimator surface of
    modl/feature_extraction/image.py:8-83: `fit`, `transform`, `partial_transform`, `shuffle`, `indices_3d`,
    `patches_`).  The state is the (n, 3) array of origins - every window without a missing (-1) pixel
    (`clean_mask`) or simply all of them (`fill`), permuted by the random state and cut to `max_patches`; a batch of
    patches is a fancy index into the strided window view on the host (`_take`) or one launch on the HBM-resident
    image (`partial_transform_scaled`)."""

    def __init__(self, patch_size=None, random_state=None, max_patches=None):
        self.patch_size = patch_size
        self.max_patches = max_patches
        self.random_state = random_state

    def fit(self, X, y=None):
        self.random_state = check_random_state(self.random_state)
        height, width, n_channels = X.shape
        ph, pw = self.patch_size if self.patch_size is not None else (height // 10, width // 10)
        self.image_, self._device_image = X, None
        self.patches_ = sliding_window_view(X, (ph, pw, n_channels))          # every window, as a strided view
        has_holes = bool(np.any(X == -1))
        origins = clean_mask(self.patches_, X) if has_holes else fill(*self.patches_.shape[:3])
        keep = self.random_state.permutation(origins.shape[0])[:self.max_patches]
        self.indices_3d = origins[keep]
        return self

    def _take(self, origins):
        return self.patches_[origins[:, 0], origins[:, 1], origins[:, 2]]

    def transform(self, X=None):
        if X is not None:
            self.fit(X)
        return self._take(self.indices_3d)

    def partial_transform(self, X=None, batch=None):
        if X is not None:
            self.fit(X)
        if batch is None:
            return self._take(self.indices_3d)
        return self._take(self.indices_3d[slice(0, batch) if isinstance(batch, int) else batch])

    def partial_transform_scaled(self, backend, batch, with_mean=True, with_std=True):
        """Device path of partial_transform + scale_patches + flattening: the patches `batch` of the HBM-resident
        image as a (n, patch_size) device tensor (image.py:139-144 in one launch)."""
        if isinstance(batch, int):
            batch = slice(0, batch)
        if self._device_image is None or self._device_image[0] is not backend:
            self._device_image = (backend, backend.stage_image(self.image_))
        return backend.image_patches(self._device_image[1], self.indices_3d[batch], self.patch_shape_, with_mean, with_std)

    def shuffle(self, permutation=None):
        if permutation is None:
            permutation = self.random_state.permutation(self.indices_3d.shape[0])
        self.indices_3d = self.indices_3d[permutation]

    @property
    def n_patches_(self):
        return self.indices_3d.shape[0]

    @property
    def patch_shape_(self):
        return self.patches_.shape[-3:]


def _flatten_patches(patches, with_mean=True, with_std=True, copy=False):
    n_patches = patches.shape[0]
    patches = scale_patches(patches, with_mean=with_mean, with_std=with_std, copy=copy)
    return patches.reshape((n_patches, -1))


class ImageDictFact(BaseEstimator):
    methods = {'masked': {'G_agg': 'masked', 'Dx_agg': 'masked'},
               'dictionary only': {'G_agg': 'full', 'Dx_agg': 'full'},
               'gram': {'G_agg': 'masked', 'Dx_agg': 'masked'},      # first epochs; switched at epoch 4
               'average': {'G_agg': 'average', 'Dx_agg': 'average'},
               'reducing ratio': {'G_agg': 'masked', 'Dx_agg': 'masked'}}

    settings = {'dictionary learning': {'comp_l1_ratio': 0, 'code_l1_ratio': 1, 'comp_pos': False,
                                        'code_pos': False, 'with_std': True, 'with_mean': True},
                'NMF': {'comp_l1_ratio': 0, 'code_l1_ratio': 1, 'comp_pos': True, 'code_pos': True,
                        'with_std': True, 'with_mean': False}}

    def __init__(self, method='masked', setting='dictionary learning', patch_size=(8, 8), batch_size=100,
                 buffer_size=None, step_size=1e-3, n_components=50, alpha=0.1, learning_rate=0.92, reduction=10,
                 n_epochs=1, random_state=None, callback=None, max_patches=None, verbose=0, n_threads=1):
        self.n_threads = n_threads
        self.step_size = step_size
        self.verbose = verbose
        self.callback = callback
        self.random_state = random_state
        self.n_epochs = n_epochs
        self.reduction = reduction
        self.learning_rate = learning_rate
        self.alpha = alpha
        self.n_components = n_components
        self.batch_size = batch_size
        self.method = method
        self.setting = setting
        self.patch_size = patch_size
        self.buffer_size = buffer_size
        self.max_patches = max_patches

    _dict_fact_class = DictFact

    def fit(self, image, y=None):
        """image.py:68-153"""
        self.random_state = check_random_state(self.random_state)
        if self.method != 'sgd':
            method = ImageDictFact.methods[self.method]
            G_agg, Dx_agg = method['G_agg'], method['Dx_agg']
            reduction = self.reduction
            optimizer = 'variational'
        else:
            optimizer, reduction, G_agg, Dx_agg = 'sgd', 1, 'full', 'full'
        setting = ImageDictFact.settings[self.setting]
        with_std, with_mean = setting['with_std'], setting['with_mean']
        buffer_size = self.batch_size * 10 if self.buffer_size is None else self.buffer_size

        self.dict_fact_ = self._dict_fact_class(
            n_epochs=self.n_epochs, random_state=self.random_state, n_components=self.n_components,
            comp_l1_ratio=setting['comp_l1_ratio'], learning_rate=self.learning_rate, comp_pos=setting['comp_pos'],
            optimizer=optimizer, step_size=self.step_size, code_pos=setting['code_pos'], batch_size=self.batch_size,
            G_agg=G_agg, Dx_agg=Dx_agg, reduction=reduction, code_alpha=self.alpha,
            code_l1_ratio=setting['code_l1_ratio'], tol=1e-2, callback=self._callback, verbose=self.verbose,
            n_threads=self.n_threads)

        patch_extractor = LazyCleanPatchExtractor(patch_size=self.patch_size, max_patches=self.max_patches,
                                                  random_state=self.random_state)
        patch_extractor.fit(image)
        n_patches = patch_extractor.n_patches_
        self.patch_shape_ = patch_extractor.patch_shape_

        # device pipeline whenever the image is float32 / float64 (the dtype the reference's in-place scale_patches
        # accepts) and the step's backend owns a GPU
        probe = self.dict_fact_._make_backend()
        on_device = hasattr(probe, 'image_patches') and image.dtype in (np.float32, np.float64)

        def scaled(batch):
            if on_device:
                return patch_extractor.partial_transform_scaled(probe, batch, with_mean=with_mean, with_std=with_std)
            return _flatten_patches(patch_extractor.partial_transform(batch=batch), with_mean=with_mean,
                                    with_std=with_std, copy=True)

        self.dict_fact_.prepare(n_samples=n_patches, X=scaled(slice(0, self.n_components)))
        for i in range(self.n_epochs):
            if i >= 1:
                permutation = self.dict_fact_.shuffle()
                patch_extractor.shuffle(permutation)
            buffers = gen_batches(n_patches, buffer_size)
            if self.method == 'gram' and i == 4:
                self.dict_fact_.set_params(G_agg='full', Dx_agg='average')
            if self.method == 'reducing ratio':
                reduction = 1 + (self.reduction - 1) / sqrt(i + 1)
                self.dict_fact_.set_params(reduction=reduction)
            for buffer in buffers:
                self.dict_fact_.partial_fit(scaled(buffer), buffer)
        return self

    def _prep(self, patches):
        s = ImageDictFact.settings[self.setting]
        return _flatten_patches(np.asarray(patches), with_mean=s['with_mean'], with_std=s['with_std'], copy=True)

    def transform(self, patches):
        return self.dict_fact_.transform(self._prep(patches))

    def score(self, patches):
        return self.dict_fact_.score(self._prep(patches))

    def stage_test_patches(self, patches):
        """Scaled, flattened test patches as a tensor on the estimator's device: `score_staged` then evaluates the
        objective without any host copy of the test set or of the dictionary (scoring callbacks, image.py:202-225)."""
        return self.dict_fact_._backend.stage_X(self._prep(patches))

    def score_staged(self, staged):
        return self.dict_fact_.score(staged)

    @property
    def n_iter_(self):
        return self.dict_fact_.n_iter_

    @property
    def time_(self):
        return self.dict_fact_.time_

    @property
    def components_(self):
        return self.dict_fact_.components_.reshape((self.n_components,) + tuple(self.patch_shape_))

    def _callback(self, *args):
        if self.callback is not None:
            self.callback(self)


class DictionaryScorer:
    """image.py:202-225 (time.clock is gone from Python 3.8: perf_counter)."""

    def __init__(self, test_data, info=None):
        self.start_time = time.perf_counter()
        self.test_data = test_data
        self.test_time = 0
        self.time, self.cpu_time, self.score, self.iter = [], [], [], []
        self.info = info

    def __call__(self, dict_fact):
        t0 = time.perf_counter()
        if hasattr(dict_fact, 'stage_test_patches'):
            # the test set is scaled and uploaded once per (estimator, setting); every later call scores it in HBM
            key = (id(dict_fact), getattr(dict_fact, 'setting', None))
            if getattr(self, '_staged_key', None) != key:
                self._staged, self._staged_key = dict_fact.stage_test_patches(self.test_data), key
            score = dict_fact.score_staged(self._staged)
        else:
            score = dict_fact.score(self.test_data)
        self.test_time += time.perf_counter() - t0
        self.time.append(time.perf_counter() - self.start_time - self.test_time)
        self.score.append(score)
        self.iter.append(dict_fact.n_iter_)
        self.cpu_time.append(dict_fact.time_)
        if self.info is not None:
            self.info['time'], self.info['score'], self.info['iter'] = self.cpu_time, self.score, self.iter
