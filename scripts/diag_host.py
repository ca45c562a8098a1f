"""Host-side cost of one minibatch: (a) the Python loop + sampler + RNG with the device call stubbed out,
(b) the device call (ctypes + staging + ~20 launches) enqueued on an idle stream."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
X = bench.M1Stream(bench.P_FEAT, 1234, dev).rows(0, 32768)
est = DictFact(n_components=256, batch_size=256, reduction=10, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
               learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
est.prepare(n_samples=32768, X=X[:256])
est.partial_fit(X[:2048], np.arange(2048))
torch.cuda.synchronize()
be = est._backend
real_step = be.step
calls = []
be.step = lambda *a, **k: calls.append(1)
t0 = time.perf_counter()
est.partial_fit(X[2048:2048 + 256 * 64], np.arange(2048, 2048 + 256 * 64))
t1 = time.perf_counter()
print('python loop + sampler + rng, device call stubbed: %.1f us / minibatch (%d minibatches)' % ((t1 - t0) / len(calls) * 1e6, len(calls)))
# the device call alone: 6 enqueues on an idle stream (ring has 8 slots), no waiting
durs = []
def timed_step(*a, **k):
    t = time.perf_counter()
    real_step(*a, **k)
    durs.append(time.perf_counter() - t)
be.step = timed_step
for rep in range(6):
    torch.cuda.synchronize()
    r0 = 20000 + rep * 1536
    est.partial_fit(X[r0:r0 + 1536], np.arange(r0, r0 + 1536))
d = np.array(durs).reshape(6, 6)
print('device call (ctypes + staging + launches), us, by position in a burst of 6:', np.round(d.mean(0) * 1e6, 1))
import cProfile, pstats
be.step = lambda *a, **k: None
pr = cProfile.Profile()
pr.enable()
est.partial_fit(X[2048:2048 + 256 * 64], np.arange(2048, 2048 + 256 * 64))
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
