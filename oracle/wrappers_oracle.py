"""ORACLE (test infrastructure, NOT product code): CPU restatements of the
wrapper loops that feed the hot path, built on oracle/somf_oracle.py.

  fmri_fit        : modl/decomposition/fmri.py:423-546 (_compute_components) + :549-556 (_flip),
                    on raw 2-D records (the reference's masking layer needs nilearn, absent here:
                    this loop is pinned by its kwargs table :440-463,481-495 and streaming order
                    :514,532-541 only — "parity unpinned" for the wrapper itself, the DictFact
                    calls underneath are pinned by tests/golden).
  recsys_*        : modl/decomposition/recsys.py (pinned by tests/golden/recsys.npz).
"""
from math import sqrt

import numpy as np
import scipy
import scipy.sparse as sp
from numpy import linalg

from . import somf_oracle as orc

FMRI_METHODS = {'masked': ('masked', 'masked'), 'dictionary only': ('full', 'full'), 'gram': ('masked', 'masked'),
                'average': ('average', 'average'), 'reducing ratio': ('masked', 'masked')}


def fmri_fit(records, method='masked', step_size=1, n_components=20, n_epochs=1, alpha=0.1, dict_init=None,
             random_state=None, batch_size=20, reduction=1, learning_rate=1, positive=False):
    if dict_init is not None:
        dict_init = np.asarray(dict_init)[:n_components]
        n_components = dict_init.shape[0]
    rng = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) else random_state
    if method == 'sgd':
        optimizer, G_agg, Dx_agg, reduction = 'sgd', 'full', 'full', 1
    else:
        G_agg, Dx_agg = FMRI_METHODS[method]
        optimizer = 'variational'
    lengths = [r.shape[0] for r in records]
    dtype = records[-1].dtype
    idx = np.zeros(len(records) + 1, dtype=int)
    idx[1:] = np.cumsum(lengths)
    pr = orc.SomfParams(n_components=n_components, code_alpha=alpha, code_l1_ratio=0, comp_l1_ratio=1,
                        comp_pos=positive, reduction=reduction, Dx_agg=Dx_agg, optimizer=optimizer,
                        step_size=step_size, G_agg=G_agg, learning_rate=learning_rate, batch_size=batch_size,
                        random_state=rng)
    st = orc.prepare(pr, n_samples=int(idx[-1]) + 1, n_features=records[0].shape[1],
                     X=None if dict_init is None else dict_init.astype(dtype), dtype=dtype)
    for i in range(n_epochs):
        if method == 'gram' and i == 5:
            pr.G_agg, pr.Dx_agg = 'full', 'average'
            st.G = st.D.dot(st.D.T)
            if st.Dx_average is None:
                st.Dx_average = np.zeros((st.code.shape[0], n_components), dtype=dtype)
        if method == 'reducing ratio':
            reduction = 1 + (reduction - 1) / sqrt(i + 1)
            pr.reduction = reduction
        for record in rng.permutation(len(records)):
            data = records[record].astype(dtype)
            perm = rng.permutation(data.shape[0])
            si = np.arange(idx[record], idx[record + 1])[perm] if method in ('average', 'gram') else None
            orc.partial_fit(st, pr, data[perm], sample_indices=si)
    D = st.D.copy()
    for comp in D:
        if np.sum(comp < 0) > np.sum(comp > 0):
            comp *= -1
    return D, st
