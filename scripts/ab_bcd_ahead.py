"""A/B of the blocked dictionary update in look-ahead mode (modl_debug_set(MODL_DEBUG_BCD_ACC, 2): the Gram matrix of a
block from cross products accumulated in the shadow of its predecessor's recursion, gram_ahead in csrc/bcd.hip) against
the accumulator mode (1): minibatch time at the metric's shape and the distance between the two dictionaries after the
same minibatches (the modes differ by f32 roundings of the candidates).   python scripts/ab_bcd_ahead.py [reduction] [k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_BCD_ACC
dev = torch.device('cuda')
RED = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 900)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
# (1) distance after 1, 2, 4, 16 minibatches from the same state
snap = {}
for mode in (1, 2):
    check(lib.modl_debug_set(DEBUG_BCD_ACC, mode))
    est = DictFact(n_components=K, batch_size=256, reduction=RED, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    done = 0
    for n in (1, 2, 4, 16):
        est.partial_fit(X[256 * done:256 * n], np.arange(256 * done, 256 * n))
        done = n
        snap[(mode, n)] = (est.components_.astype(np.float64), est.comp_norm_.astype(np.float64))
for n in (1, 2, 4, 16):
    print('after %2d minibatches: look-ahead vs accumulator D %.2e  norm budgets %.2e  finite %s' % (
        n, rel(snap[(2, n)][0], snap[(1, n)][0]), rel(snap[(2, n)][1], snap[(1, n)][1]), bool(np.isfinite(snap[(2, n)][0]).all())), flush=True)
# (2) time
out = {}
for mode in (1, 2, 1, 2):
    check(lib.modl_debug_set(DEBUG_BCD_ACC, mode))
    est = DictFact(n_components=K, batch_size=256, reduction=RED, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    est.partial_fit(X[:256 * 300], np.arange(256 * 300) % 65536)
    ts = []
    for rep in range(3):
        a = 256 * (300 + 200 * rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est.partial_fit(X[a:a + 256 * 200], np.arange(a, a + 256 * 200) % 65536)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
    D = est.components_.astype(np.float64)
    print('mode=%d: %s ms per minibatch, finite %s' % (mode, ' '.join('%.4f' % t for t in ts), bool(np.isfinite(D).all())), flush=True)
    out.setdefault(mode, []).append(D)
check(lib.modl_debug_set(DEBUG_BCD_ACC, 1))
print('accumulator run-to-run %.2e, look-ahead run-to-run %.2e (both must be 0), look-ahead vs accumulator after 900 minibatches %.2e'
      % (rel(out[1][0], out[1][1]), rel(out[2][0], out[2][1]), rel(out[2][0], out[1][0])))
