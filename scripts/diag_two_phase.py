"""Diagnostic: two-phase (head-first) step against the fused step, element-wise difference."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from modl_amd import DictFact

rng = np.random.RandomState(3)
X = rng.randn(300, 96).astype(np.float32)
for steps in (1, 5):
    out = []
    for two_phase in (False, True):
        est = DictFact(n_components=32, batch_size=40, reduction=3, code_alpha=0.5, learning_rate=0.9, random_state=0)
        est._two_phase = two_phase
        est.prepare(n_samples=300, X=X)
        est.partial_fit(X[:40 * steps], np.arange(40 * steps))
        out.append((est.components_, est.code_.copy(), est.C_, est.B_))
    for name, a, b in zip(('D', 'code', 'C', 'B'), out[0], out[1]):
        d = np.abs(a - b)
        print(steps, name, 'max abs diff %.3e' % d.max(), 'rel fro %.3e' % (np.linalg.norm(a - b) / np.linalg.norm(a)),
              'n differing', int((d > 0).sum()), 'of', d.size)
        if name == 'B' and d.max() > 0:
            rows = np.where(d.max(axis=1) > 0)[0]
            print('   differing B rows (features):', rows[:40], '... cols:', np.where(d.max(axis=0) > 0)[0][:40])
