"""Config 5 shape on one GPU: p = 200 000 features, k = 256, b = 256, reduction 12 — a few minibatches,
finiteness + throughput (one rank's share of the HCP-shaped stream)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
p, n = 200000, 10752                                     # 2 + 40 minibatches of 256 rows (8.6 GB of f32 rows)
X = bench.M1Stream(p, 7, dev).rows(0, n)
est = DictFact(n_components=256, batch_size=256, reduction=12, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=n, X=X[:256])
if len(sys.argv) > 1:                                   # diagnostics: modl_somf_desc.flags (1 = no rider)
    est._backend.flags = int(sys.argv[1])
    est._backend.update_plan(est._plan_kwargs(256))
est.partial_fit(X[:512], np.arange(512))
torch.cuda.synchronize()
t0 = time.perf_counter()
est._backend.host_wait_ms()
est.partial_fit(X[512:n], np.arange(512, n), _sync=False)
enq = time.perf_counter() - t0
waited = est._backend.host_wait_ms() * 1e-3
torch.cuda.synchronize()
dt = time.perf_counter() - t0
nb = (n - 512) // 256
D = est.components_
print('p=%d: %.2f ms / minibatch over %d minibatches, %.0f samples/s, finite %s, max |D row norm| %.4f' % (p, dt / nb * 1e3, nb, (n - 512) / dt, bool(np.isfinite(D).all()), float(np.sqrt((D.astype(np.float64) ** 2).sum(1)).max())))
print('   host: %.2f ms / minibatch to enqueue, of which %.2f ms waiting for a free staging slot (i.e. for the device)' % (enq / nb * 1e3, waited / nb * 1e3))
be = est._backend
be.prof_enable(True); be.prof_reset()
est.partial_fit(X[:1536], np.arange(1536))
torch.cuda.synchronize()
for name, e in be.prof_get().items():
    if e['calls']:
        print('   %-12s %.3f ms / minibatch (%d launches)' % (name, e['ms'] / e['calls'], e['launches'] / e['calls']))
be.prof_enable(False)
