import sys, time
sys.path.insert(0, 'scripts'); sys.path.insert(0, '.')
import numpy as np, torch, cProfile, pstats
from bench_configs import ml10m_like
from modl_amd.recsys import RecsysDictFact
X = ml10m_like(nnz=10_000_000)
est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1, random_state=0)
est.fit(X[:2000])            # warm
est = RecsysDictFact(n_components=50, alpha=1, beta=.1, batch_size=10, detrend=True, learning_rate=.95, n_epochs=1, random_state=0)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); est.fit(X); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print('fit %.3f s' % dt)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
