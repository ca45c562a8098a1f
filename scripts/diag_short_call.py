"""What a SHORT timed region costs beyond its minibatches (the driver times `bench.py --steps 20 --warmup 5`): calls of
1, 2, 5, 10, 20, 40 and 80 minibatches after a warm-up of 5, each bracketed exactly as bench.timed does; a straight line through
them separates the per-minibatch slope from the fixed cost of a timed region (the host's preamble before the first launch, the
pipeline fill, the wake-up after the last launch).  usage (GPU box): python scripts/diag_short_call.py [reduction]"""
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

red = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
args = types.SimpleNamespace(torch_collective=False, backend='nccl', force_reduce=False)
sizes = [1, 2, 5, 10, 20, 40, 80, 20, 20, 20]
run = bench.Run(args, red, 0, 1, dev, 5 + sum(sizes) + 8)
run.fit(5)
run.sync()
pts = []
for n in sizes:
    run.sync()
    run.be.host_wait_ms()
    t0 = time.perf_counter()
    run.fit(n)
    t1 = time.perf_counter()
    run.be.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    pts.append((n, (t3 - t0) * 1e3))
    print('%3d minibatches: %.3f ms (%.1f us each) | enqueue returned after %.3f ms, estimator wait %.3f ms, device sync %.3f ms'
          % (n, (t3 - t0) * 1e3, (t3 - t0) * 1e6 / n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
x = np.array([p[0] for p in pts], float)
y = np.array([p[1] for p in pts], float)
a, b = np.polyfit(x, y, 1)
print('fit: %.1f us per minibatch + %.1f us per timed region' % (a * 1e3, b * 1e3))
