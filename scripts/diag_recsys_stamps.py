"""diagnostics: where a one-launch masked minibatch spends its time (stamps of its last workgroup, 100 MHz wall clock), over
minibatches of a MovieLens-10M-shaped stream"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bench_configs import ml10m_like
from modl_amd.recsys import RecsysDictFact
from modl_amd.randomkit import batch_weight
from modl_amd._lib import lib, check, DEBUG_RECSYS_FUSED
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, v))
X = ml10m_like(nnz=10_000_000)[:3000]
est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1, random_state=0,
                     callback=lambda e: None)
names = ['ids', 'gram', 'records', 'factor', 'solve+ticket', 'wait/entry', 'codes+C', 'B', 'sweep', 'tail']
acc, cnt = {}, {}
orig = est._single_batch_fit
def traced(Xm, batch):
    orig(Xm, batch)
    d = est._dev
    if d.plan is None:
        return
    out = (C.c_ulonglong * 16)()
    check(lib.modl_recsys_plan_stamps(d.plan, 1, out))
    st = list(out)
    if st[10] == 0:
        return
    key = (int(st[12]), min(int(st[11]) // 256, 8))
    dur = np.diff(np.array(st[:11], dtype=np.float64)) / 100.0       # us
    acc[key] = acc.get(key, 0) + dur
    cnt[key] = cnt.get(key, 0) + 1
est._single_batch_fit = traced
est.fit(X)
print('variant', v, ' columns (us):', ' '.join(names), '| total')
for key in sorted(acc):
    d = acc[key] / cnt[key]
    print('W=%d items %4d-%4d  n=%3d : ' % (key[0], key[1] * 256, key[1] * 256 + 255, cnt[key]) + ' '.join('%6.1f' % x for x in d) + ' | %6.1f' % d.sum())
