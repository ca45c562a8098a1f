cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, json
sys.path.insert(0, 'scripts'); sys.path.insert(0, '.')
import numpy as np, torch
from bench_configs import ml10m_like
from modl_amd.recsys import RecsysDictFact
import modl_amd.recsys as R
from modl_amd._lib import lib, check, DEBUG_RECSYS_FUSED
X = ml10m_like(nnz=3_000_000)
Xs = X[:20000]
orig = R._RecsysDevice.fit_batches
def timed(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *a, **k)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    import ctypes as C
    w = C.c_double(0); lib.modl_recsys_plan_wait_ms(self.plan, C.byref(w))
    print('   fit_batches: host %.1f us / minibatch (of which %.1f waiting for the device), device drained %.1f us later per minibatch' % ((t1 - t0) / 2000 * 1e6, w.value / 2000 * 1e3, (t2 - t1) / 2000 * 1e6))
    return r
R._RecsysDevice.fit_batches = timed
for v in (1, 3, 2, 0, 1):
    check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, v))
    est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1, random_state=0)
    t0 = time.perf_counter(); est.fit(Xs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('fused=%d  %.0f rows/s  %.1f us per minibatch  counts %s' % (v, Xs.shape[0] / dt, dt / (Xs.shape[0] / 10) * 1e6, est._dev.launch_counts()))
PY
