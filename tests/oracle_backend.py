"""Test-only backend: lets the HOST logic of modl_amd.DictFact (minibatch loop,
RNG draws, weights, multi-rank protocol with the all-reduce of the statistics
increment) run on a machine without a GPU, with the CPU oracle standing in for
the device kernels.  Lives under tests/ on purpose: the product never imports it."""
import numpy as np
import torch

from oracle import somf_oracle as orc


class OracleBackend:
    name = 'oracle'

    def __init__(self, device=None):
        self.device = torch.device('cpu')
        self.Dt = None

    def allocate(self, kw, n_samples, p, k, dtype):
        self.dtype = np.dtype(dtype)
        self.n, self.p, self.k = n_samples, p, k
        self._desc_kw = dict(kw)
        self.pr = orc.SomfParams(
            G_agg=kw['G_agg'], Dx_agg=kw['Dx_agg'], optimizer=kw['optimizer'], code_alpha=kw['code_alpha'],
            code_l1_ratio=kw['code_l1_ratio'], comp_l1_ratio=kw['comp_l1_ratio'], step_size=kw['step_size'],
            tol=kw['tol'], max_iter=kw['max_iter'], code_pos=kw['code_pos'], comp_pos=kw['comp_pos'], n_components=k)
        st = orc.SomfState()
        st.D = np.zeros((k, p), dtype=dtype)
        st.C = np.zeros((k, k), dtype=dtype)
        st.B = np.zeros((k, p), dtype=dtype)
        st.code = np.ones((n_samples, k), dtype=dtype)
        st.comp_norm = np.zeros(k, dtype=dtype)
        st.G = np.zeros((k, k), dtype=dtype) if kw['G_agg'] == 'full' else None
        st.Dx_average = np.zeros((n_samples, k), dtype=dtype) if kw['Dx_agg'] == 'average' else None
        st.G_average = np.zeros((n_samples, k, k), dtype=dtype) if kw['G_agg'] == 'average' else None
        self.st = st
        self.Dt = True                                       # "allocated" marker used by the estimator

    def update_plan(self, kw):
        self._desc_kw = dict(kw)
        for f in ('G_agg', 'Dx_agg', 'code_alpha', 'code_l1_ratio', 'tol', 'max_iter', 'code_pos'):
            setattr(self.pr, f, kw[f])
        if kw['Dx_agg'] == 'average' and self.st.Dx_average is None:      # allocated lazily, like the device backend
            self.st.Dx_average = np.zeros((self.n, self.k), dtype=self.dtype)
        if kw['G_agg'] == 'full' and self.st.G is None:
            self.st.G = np.zeros((self.k, self.k), dtype=self.dtype)

    # state access
    def set_dictionary(self, D):
        self.st.D = np.array(D, dtype=self.dtype, copy=True)

    def get_dictionary(self):
        return self.st.D.copy()

    def get_B(self):
        return self.st.B.copy()

    def set_B(self, B):
        self.st.B = np.array(B, dtype=self.dtype)

    _NAMES = {'C': 'C', 'code': 'code', 'comp_norm': 'comp_norm', 'G': 'G', 'Dx_average': 'Dx_average',
              'G_average': 'G_average'}

    def get(self, name):
        v = getattr(self.st, self._NAMES[name])
        return None if v is None else v.copy()

    def set(self, name, value):
        setattr(self.st, self._NAMES[name], np.array(value, dtype=self.dtype))

    @property
    def G(self):
        return self.st.G

    @property
    def Dx_average(self):
        return self.st.Dx_average

    def scale_atoms(self, l1_ratio, radius=1.0):
        for i in range(self.k):
            orc.enet_scale(self.st.D[i], l1_ratio, radius)

    def full_gram(self):
        self.st.G = self.st.D.dot(self.st.D.T)

    def n_rows(self, name):
        return getattr(self.st, self._NAMES[name]).shape[0]

    def shuffle_rows(self, name, swaps):
        arr = getattr(self.st, self._NAMES[name])
        flat = arr.reshape(arr.shape[0], -1)
        orc.lib().ork_apply_swaps_rows(flat.ctypes.data, flat.shape[0], flat.strides[0], swaps.ctypes.data)

    # data
    def stage_X(self, X):
        return np.ascontiguousarray(X.cpu().numpy() if isinstance(X, torch.Tensor) else X, dtype=self.dtype)

    def take_rows(self, Xh, perm):
        return Xh[perm]

    def synchronize(self):
        pass

    def broadcast_dictionary(self, dist):
        t = torch.from_numpy(self.st.D)                      # shares memory with st.D
        dist.broadcast(t, src=0)

    # the step, split like the device step: st.C / st.B are this rank's PARTIAL statistics
    def phase1(self, Xh, batch, idx, subset, order, w_sample, w, reduction, b_global):
        X = Xh[batch]
        self.pr.reduction = reduction
        ws = w_sample if w_sample is not None else np.ones(X.shape[0], dtype=self.dtype)
        subset = np.asarray(subset)
        orc.compute_code(self.st, self.pr, X, idx, ws, subset)
        code = self.st.code[idx]
        st, pr = self.st, self.pr
        dC, dB = code.T.dot(code), code.T.dot(X)
        if pr.optimizer == 'variational':                    # dict_fact.py:559-575 with the global batch size
            st.C *= 1 - w
            st.C += w * dC / b_global
            st.B *= 1 - w
            st.B += w * dB / b_global
        else:
            st.C = dC / b_global
            st.B = np.ascontiguousarray(dB / b_global)
        self._pending = (subset, np.asarray(order), w)
        head = np.concatenate([st.C.ravel(), np.ascontiguousarray(st.B[:, subset].T).ravel()])   # [k*k | s*k feature-major]
        self.head = torch.from_numpy(head.astype(self.dtype))
        return self.head

    def phase2(self, head):
        subset, order, w = self._pending
        d = head.numpy()
        k = self.k
        st, pr = self.st, self.pr
        C_own, B_own = st.C, st.B
        st.C = d[:k * k].reshape(k, k).copy()                # the summed statistics, for the dictionary update only
        st.B = B_own.copy()
        st.B[:, subset] = d[k * k:].reshape(len(subset), k).T
        orc.update_dict(st, pr, subset, w, order)
        st.C, st.B = C_own, B_own

    def transform(self, Xh, kw, G=None, to_host=True):
        pr = orc.SomfParams(code_alpha=kw['code_alpha'], code_l1_ratio=kw['code_l1_ratio'], code_pos=kw['code_pos'],
                            tol=kw['tol'], max_iter=kw['max_iter'])
        return orc.transform(pr, self.st.D, Xh, G)

    def objective(self, Xh, code):
        Xh = np.asarray(Xh)
        return np.array([np.sum((Xh - code.dot(self.st.D)) ** 2), np.sum(np.abs(code)), np.sum(code ** 2)])
