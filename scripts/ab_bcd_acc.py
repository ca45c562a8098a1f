"""A/B of the blocked dictionary update with the atomic fixed-point Gram accumulator (default) against the per-workgroup
Gram records (modl_debug_set(MODL_DEBUG_BCD_ACC, 0)): minibatch time at the metric's shape and the distance between the
two dictionaries after the same minibatches.  python scripts/ab_bcd_acc.py [reduction]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_BCD_ACC
dev = torch.device('cuda')
RED = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 900)
out = {}
for acc in (0, 1, 0, 1):
    check(lib.modl_debug_set(DEBUG_BCD_ACC, acc))
    est = DictFact(n_components=256, batch_size=256, reduction=RED, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    est.partial_fit(X[:256 * 300], np.arange(256 * 300) % 65536)
    ts = []
    for rep in range(3):
        a = 256 * (300 + 200 * rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est.partial_fit(X[a:a + 256 * 200], np.arange(a, a + 256 * 200) % 65536)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
    D = est.components_.astype(np.float64)
    print('acc=%d: %s ms per minibatch, finite %s' % (acc, ' '.join('%.4f' % t for t in ts), bool(np.isfinite(D).all())), flush=True)
    out.setdefault(acc, []).append(D)
check(lib.modl_debug_set(DEBUG_BCD_ACC, 1))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print('records run-to-run %.2e, accumulator run-to-run %.2e (both must be 0), accumulator vs records after 900 minibatches %.2e'
      % (rel(out[0][0], out[0][1]), rel(out[1][0], out[1][1]), rel(out[1][0], out[0][0])))
