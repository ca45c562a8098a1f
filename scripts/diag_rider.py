"""Dictionary-update launches with and without the riding statistics product (MODL_FLAG_NO_RIDER): run under
rocprofv3 --kernel-trace and summarise with scripts/step_timeline.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 32768)
flag = int(sys.argv[1]) if len(sys.argv) > 1 else 0
est = DictFact(n_components=256, batch_size=256, reduction=10, code_alpha=1.0, learning_rate=0.92, random_state=0)
est.prepare(n_samples=32768, X=X[:256])
est._backend.flags = flag
est._backend.update_plan(est._plan_kwargs(256))
est.partial_fit(X[:256 * 120])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
est.partial_fit(X[:256 * 120])
torch.cuda.synchronize()
print('flags %d: %.1f us per step' % (flag, (time.perf_counter() - t0) / 120 * 1e6))
