"""fMRI-like shape (config 3): k = 70 atoms with l1 atoms (generic per-atom dictionary update), p = 60 000 voxels,
reduction 12, batch 50 records — section times per minibatch."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
p, n, b, k = 60000, 1000, 50, 70
X = bench.M1Stream(p, 3, dev, k0=64).rows(0, n)
for l1 in (1.0, 0.0):
    est = DictFact(n_components=k, batch_size=b, reduction=12, code_alpha=1e-3, code_l1_ratio=0, comp_l1_ratio=l1,
                   learning_rate=0.92, random_state=0)
    est.prepare(n_samples=n, X=X[:k])
    est.partial_fit(X[:200], np.arange(200))
    be = est._backend
    be.prof_enable(True); be.prof_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[200:1000], np.arange(200, 1000))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('comp_l1_ratio=%g: %.3f ms / minibatch (%d records/s)' % (l1, dt / 16 * 1e3, 800 / dt))
    for name, e in be.prof_get().items():
        if e['calls']:
            print('   %-12s %.3f ms (%d launches)' % (name, e['ms'] / e['calls'], e['launches'] / e['calls']))
    be.prof_enable(False)
