"""Same-box A/B of the two kernels of round 5 in their final state, and a soak of the persistent launch.
  persistent dictionary update  : MODL_DEBUG_BCD_PERSIST 1 against 0, reduction 10 and 1 (minibatch time, run-to-run identity)
  register-resident statistics  : MODL_DEBUG_STATS_RESIDENT 1 against 0, reduction 1
  soak                          : 30 000 minibatches at reduction 10 through ONE estimator (chunk calls of 256 minibatches, rows
                                  revisited modulo the chunk), the status word checked after every call
python scripts/ab_final_r05.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_BCD_PERSIST, DEBUG_STATS_RESIDENT
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 256 * 900)
res = {}

def fit_time(red, nrep=3):
    est = DictFact(n_components=256, batch_size=256, reduction=red, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=65536, X=X[:256])
    est.partial_fit(X[:256 * 300], np.arange(256 * 300) % 65536)
    ts = []
    for rep in range(nrep):
        a = 256 * (300 + 200 * rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        est.partial_fit(X[a:a + 256 * 200], np.arange(a, a + 256 * 200) % 65536)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
    return ts, est.components_.astype(np.float64)

rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name, sw, reds in (('persistent_dictionary_update', DEBUG_BCD_PERSIST, (10.0, 1.0)), ('resident_statistics_product', DEBUG_STATS_RESIDENT, (1.0,))):
    for red in reds:
        out = {}
        for v in (0, 1, 0, 1):
            check(lib.modl_debug_set(sw, v))
            ts, D = fit_time(red)
            out.setdefault(v, []).append((ts, D))
            print('%s reduction %g switch %d: %s ms per minibatch' % (name, red, v, ' '.join('%.4f' % t for t in ts)), flush=True)
        check(lib.modl_debug_set(sw, 1))
        res['%s_r%g' % (name, red)] = dict(
            off_ms=[min(o[0]) for o in out[0]], on_ms=[min(o[0]) for o in out[1]],
            run_to_run_off=rel(out[0][0][1], out[0][1][1]), run_to_run_on=rel(out[1][0][1], out[1][1][1]),
            on_vs_off_after_900_minibatches=rel(out[1][0][1], out[0][0][1]))
        print(json.dumps({k: v for k, v in res['%s_r%g' % (name, red)].items()}), flush=True)

# soak
est = DictFact(n_components=256, batch_size=256, reduction=10.0, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=65536, X=X[:256])
t0 = time.perf_counter()
n = 0
for call in range(118):
    a = (call % 3) * 256 * 256
    est.partial_fit(X[a:a + 256 * 256], np.arange(a, a + 256 * 256) % 65536)      # (synchronises and checks the status word)
    n += 256
dt = time.perf_counter() - t0
D = est.components_
res['soak'] = dict(minibatches=n, seconds=dt, ms_per_minibatch=dt / n * 1e3, finite=bool(np.isfinite(D).all()),
                   max_row_norm=float(np.sqrt((D.astype(np.float64) ** 2).sum(1)).max()))
print(json.dumps(res['soak']), flush=True)
print(json.dumps(res))
