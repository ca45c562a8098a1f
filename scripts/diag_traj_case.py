"""one small golden trajectory with the persistent dictionary update on / off: where do the end states part?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.conftest import load_golden, rel_fro
from tests.test_oracle_golden import small_case_params
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_BCD_PERSIST
name = sys.argv[1] if len(sys.argv) > 1 else 'agg_masked_full_f32'
g = load_golden('traj_small')
kw, X, dt = small_case_params(name)
ref64 = name[:-3] + 'f64'
res = {}
for persist in (1, 0):
    check(lib.modl_debug_set(DEBUG_BCD_PERSIST, persist))
    est = DictFact(**kw)
    est.prepare(n_samples=X.shape[0], X=X)
    Xh = X
    sw = []
    b = kw['batch_size']
    for ep in range(kw['n_epochs']):
        for r0 in range(0, X.shape[0], b):
            est.partial_fit(Xh[r0:r0 + b], np.arange(r0, min(r0 + b, X.shape[0])))
            sw.append(est._backend.last_sweeps().copy())
        perm = est.shuffle()
        Xh = Xh[perm]
    res[persist] = (est.components_.copy(), est.code_.copy(), est.C_.copy(), sw)
    for key, val in (('D_final', res[persist][0]), ('code_final', res[persist][1]), ('C_final', res[persist][2])):
        print('persist=%d %-10s err vs ref f64 %.3e   ref f32 noise %.3e' % (persist, key, rel_fro(val, g[ref64 + '/' + key]),
                                                                             rel_fro(g[name + '/' + key], g[ref64 + '/' + key])))
check(lib.modl_debug_set(DEBUG_BCD_PERSIST, 1))
s1, s0 = res[1][3], res[0][3]
for t, (a, c) in enumerate(zip(s1, s0)):
    if not np.array_equal(a, c):
        print('minibatch %d: %d samples with another sweep count (persist on / off): %s' % (t, int((a != c).sum()), list(zip(a[a != c], c[a != c]))[:8]))
print('D on/off', rel_fro(res[1][0], res[0][0]), 'codes on/off', rel_fro(res[1][1], res[0][1]))
