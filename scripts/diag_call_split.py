"""Where the wall time of the driver's timed region (bench.py --steps 20 --warmup 5) goes, outside the kernels: host timestamps
around the same calls bench.py's timed() makes, for three fresh estimators."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(torch_collective=False, backend='nccl', force_reduce=False)
dev = torch.device('cuda', 0)
for rep in range(3):
    run = bench.Run(args, 10.0, 0, 1, dev, 80)
    be = run.be
    marks = {}
    orig = be.fit_chunk
    def fit_chunk(*a, **k):
        marks['chunk_in'] = time.perf_counter()
        r = orig(*a, **k)
        marks['chunk_out'] = time.perf_counter()
        return r
    be.fit_chunk = fit_chunk
    run.fit(5)
    run.sync()
    t0 = time.perf_counter()
    run.fit(20)
    t1 = time.perf_counter()
    be.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    print('20 minibatches: python before the library call %.0f us, library call (enqueue) %.0f us, python after %.0f us, '
          'estimator wait (polling) %.0f us, torch.cuda.synchronize after it %.0f us; wall %.0f us = %.1f us per minibatch'
          % ((marks['chunk_in'] - t0) * 1e6, (marks['chunk_out'] - marks['chunk_in']) * 1e6, (t1 - marks['chunk_out']) * 1e6,
             (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, (t3 - t0) / 20 * 1e6), flush=True)
