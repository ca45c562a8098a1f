"""Why does the four-wavefront solver start ~5.7 us after the kernel in front of it has ended (every step, r = 10 and
r = 1: profiles/r03_step_timeline_*.txt) when no other launch of the step shows a gap?  Run under
    rocprofv3 --kernel-trace -d /tmp/w/gap -o t -- python3 scripts/diag_cd_launch_gap.py
and dump the dispatches with scripts/dump_trace.py: the regression entry point enqueues row_norm2 -> H0 product ->
cd_split back to back, 40 times, on device-resident inputs (no copies in between)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from modl_amd import dict_fact_fast as fast  # noqa: E402

k, b, p = 256, 256, 1000
rs = np.random.RandomState(0)
D = rs.randn(k, p).astype(np.float32)
D /= np.sqrt((D ** 2).sum(1))[:, None]
X = ((rs.randn(b, k) * (rs.rand(b, k) < 0.1)).dot(D) + 0.1 * rs.randn(b, p)).astype(np.float32)
G = D.dot(D.T).astype(np.float32)
G = (G + G.T) / 2
dev = torch.device('cuda')
dG, dX = torch.from_numpy(G).to(dev), torch.from_numpy(X).to(dev)
dDx = torch.from_numpy(X.dot(D.T).astype(np.float32)).to(dev)
idx = np.arange(b, dtype=np.int64)
for rep in range(40):
    code = torch.ones((b, k), dtype=torch.float32, device=dev)
    fast._enet_regression_single_gram(dG, dDx, dX, code, idx, 1.0, 0.3, False, 1e-2, 100)
torch.cuda.synchronize()
