cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, json
sys.path.insert(0, 'scripts'); sys.path.insert(0, '.')
import numpy as np, torch
from bench_configs import ml10m_like
from modl_amd.recsys import RecsysDictFact
from modl_amd._lib import lib, check, DEBUG_RECSYS_FUSED
for nnz in (10_000_000, 3_000_000, 1_000_000):
    X = ml10m_like(nnz=nnz)
    Xs = X[:20000]
    for v in (1, 2, 3, 4, 0, 1, 2, 3, 4):
        check(lib.modl_debug_set(DEBUG_RECSYS_FUSED, v))
        est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1, random_state=0)
        t0 = time.perf_counter(); est.fit(Xs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('nnz %8d (%.0f ratings per row)  fused=%d  %.0f rows/s  %.1f us per minibatch  counts %s' % (nnz, Xs.nnz / Xs.shape[0], v, Xs.shape[0] / dt, dt / (Xs.shape[0] / 10) * 1e6, est._dev.launch_counts()))
PY
