"""Host-side split of one partial_fit call of the driver's regime (20 minibatches): Python preamble before the library's
chunk call, the chunk call itself (enqueue), synchronize() (stream wait + status words)."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(torch_collective=False, backend='nccl', force_reduce=False)
dev = torch.device('cuda', 0)
run = bench.Run(args, 10.0, 0, 1, dev, 600)
be = run.est._backend
marks = {}
orig_chunk = be.fit_chunk
def fit_chunk(*a, **k):
    marks['chunk_in'] = time.perf_counter()
    r = orig_chunk(*a, **k)
    marks['chunk_out'] = time.perf_counter()
    return r
be.fit_chunk = fit_chunk
run.fit(5); run.sync()
for n in (20, 20, 20, 1, 1, 1):
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    t0 = time.perf_counter()
    e0.record()
    run.fit(n)
    t1 = time.perf_counter()
    run.sync()
    t2 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    print('fit(%d): preamble %.1f us, chunk call %.1f us, return %.1f us, sync after the call %.1f us; wall %.1f us; stream e0->e1 %.1f us'
          % (n, (marks['chunk_in'] - t0) * 1e6, (marks['chunk_out'] - marks['chunk_in']) * 1e6,
             (t1 - marks['chunk_out']) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
