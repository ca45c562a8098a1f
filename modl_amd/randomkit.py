"""RandomState and Sampler with the reference's Python surface
(modl/utils/randomkit/random_fast.pyx:49-150, sampler.pyx:9-70), backed by the
host C++ streams of libmodl_hip.so (csrc/rk_host.cpp).  Draws are bit-identical
to the reference for a given seed."""
import ctypes as C

import numpy as np

from ._lib import lib, check


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RandomState:
    def __init__(self, seed=None):
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        if not isinstance(seed, (int, np.integer)):
            raise ValueError('Wrong seed')
        self.initial_seed = int(seed)
        h = C.c_void_p()
        check(lib.modl_rk_create(self.initial_seed & 0xFFFFFFFFFFFFFFFF, C.byref(h)), 'modl_rk_create')
        self._h = h

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            lib.modl_rk_destroy(h)
            self._h = None

    def __reduce__(self):                     # random_fast.pyx:56-57: restored from the initial seed
        return RandomState, (self.initial_seed,)

    def seed(self, seed=None):
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        elif not isinstance(seed, (int, np.integer)):
            raise ValueError('Wrong seed')
        check(lib.modl_rk_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def randint(self, high):
        out = C.c_int64()
        check(lib.modl_rk_randint(self._h, int(high), C.byref(out)))
        return out.value

    def random_u32(self):
        out = C.c_uint32()
        check(lib.modl_rk_random(self._h, C.byref(out)))
        return out.value

    def double(self):
        out = C.c_double()
        check(lib.modl_rk_double(self._h, C.byref(out)))
        return out.value

    def binomial(self, n, p):
        out = C.c_int64()
        check(lib.modl_rk_binomial(self._h, int(n), float(p), C.byref(out)))
        return out.value

    def permutation(self, size):
        out = np.empty(int(size), dtype=np.int64)
        check(lib.modl_rk_permutation(self._h, int(size), _p(out)))
        return out

    def _draw_swaps(self, n):
        trace = np.empty(n, dtype=np.int64)
        swaps = np.empty(n, dtype=np.int64)
        check(lib.modl_rk_shuffle_trace(self._h, n, _p(trace), _p(swaps)))
        return trace, swaps

    @staticmethod
    def _apply(x, swaps):
        if isinstance(x, np.ndarray) and x.flags.c_contiguous and x.flags.writeable:
            row_bytes = x.strides[0] if x.ndim > 0 and x.shape[0] > 0 else 0
            check(lib.modl_apply_swaps_rows(_p(x), x.shape[0], row_bytes, _p(swaps)))
        else:                                  # generic sequence (lists, strided views)
            for i in range(len(x) - 1, 0, -1):
                j = int(swaps[i])
                if isinstance(x, np.ndarray) and x.ndim > 1:
                    x[[i, j]] = x[[j, i]]
                else:
                    x[i], x[j] = x[j], x[i]

    def shuffle(self, x, swap=None):
        """In-place shuffle of a sequence / of the rows of an array (random_fast.pyx:87-125)."""
        n = len(x)
        if swap is None:
            if isinstance(x, np.ndarray) and x.dtype == np.int64 and x.ndim == 1 and x.flags.c_contiguous:
                check(lib.modl_rk_shuffle_i64(self._h, _p(x), n))
                return
            _, swap = self._draw_swaps(n)
        self._apply(x, np.ascontiguousarray(swap, dtype=np.int64))

    def shuffle_with_trace(self, arrays):
        """random_fast.pyx:127-144: one swap sequence applied to every array; returns the permutation."""
        n = len(arrays[0])
        trace, swaps = self._draw_swaps(n)
        for x in arrays:
            if hasattr(x, '_modl_device_rows'):            # device-resident rows (DictFact state)
                x._modl_device_rows(swaps)
            else:
                self._apply(x, swaps)
        return trace


class Sampler:
    """Feature-subset generator (sampler.pyx:9-70)."""

    def __init__(self, range, rand_size, replacement, random_seed):
        self.range = int(range)
        self.rand_size = bool(rand_size)
        self.replacement = bool(replacement)
        h = C.c_void_p()
        check(lib.modl_sampler_create(self.range, int(self.rand_size), int(self.replacement),
                                      int(random_seed) & 0xFFFFFFFFFFFFFFFF, C.byref(h)), 'modl_sampler_create')
        self._h = h
        self._buf = np.empty(max(self.range, 1), dtype=np.int64)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            lib.modl_sampler_destroy(h)
            self._h = None

    def yield_subset(self, reduction):
        n = C.c_int64()
        check(lib.modl_sampler_yield_subset(self._h, float(reduction), _p(self._buf), C.byref(n)),
              'modl_sampler_yield_subset')
        return self._buf[:n.value].copy()

    def _get(self):
        r, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        box = np.empty(max(self.range, 1), dtype=np.int64)
        check(lib.modl_sampler_get(self._h, C.byref(r), C.byref(lo), C.byref(hi), _p(box)))
        return lo.value, hi.value, box[:self.range]

    @property
    def box(self):
        return self._get()[2]

    @property
    def lim_inf(self):
        return self._get()[0]

    @property
    def lim_sup(self):
        return self._get()[1]

    def __getstate__(self):
        nbytes = lib.modl_sampler_state_bytes(self._h)
        buf = np.empty(nbytes, dtype=np.uint8)
        check(lib.modl_sampler_get_state(self._h, _p(buf), nbytes))
        return dict(range=self.range, rand_size=self.rand_size, replacement=self.replacement, blob=buf)

    def restore(self, st):
        """rewind THIS sampler to a state taken from it with __getstate__ (no new handle)"""
        blob = np.ascontiguousarray(st['blob'])
        check(lib.modl_sampler_set_state(self._h, _p(blob), blob.nbytes))

    def __setstate__(self, st):
        self.__init__(st['range'], st['rand_size'], st['replacement'], 0)
        blob = np.ascontiguousarray(st['blob'])
        check(lib.modl_sampler_set_state(self._h, _p(blob), blob.nbytes))


def batch_weight(count, batch_size, learning_rate, offset=0.0):
    """_batch_weight (modl/decomposition/dict_fact_fast.pyx:115-122)."""
    out = C.c_double()
    check(lib.modl_batch_weight(int(count), int(batch_size), float(learning_rate), float(offset), C.byref(out)))
    return out.value
