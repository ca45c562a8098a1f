"""A/B of the parameter staging inside a chunk of minibatches: the copy riding on the previous step's last launch
(default) against a staging launch at the head of every step (modl_debug_set(MODL_DEBUG_STAGE_AHEAD, 0)); same box,
metric's shape, reduction 10 and 1; the dictionaries must be identical."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_STAGE_AHEAD
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 900)
for red in (10.0, 1.0):
    out = {}
    for ahead in (0, 1, 0, 1):
        check(lib.modl_debug_set(DEBUG_STAGE_AHEAD, ahead))
        est = DictFact(n_components=256, batch_size=256, reduction=red, code_alpha=1.0, learning_rate=0.92, random_state=0)
        est.prepare(n_samples=65536, X=X[:256])
        est.partial_fit(X[:256 * 300], np.arange(256 * 300) % 65536)
        ts = []
        for rep in range(3):
            a = 256 * (300 + 200 * rep)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            est.partial_fit(X[a:a + 256 * 200], np.arange(a, a + 256 * 200) % 65536)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
        out.setdefault(ahead, []).append(est.components_.copy())
        print('reduction %g, ahead=%d: %s ms per minibatch' % (red, ahead, ' '.join('%.4f' % t for t in ts)), flush=True)
    print('   identical dictionaries:', np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]))
check(lib.modl_debug_set(DEBUG_STAGE_AHEAD, 1))
