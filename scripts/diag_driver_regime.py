"""Where does the 20-step timed region of the driver's arguments (--steps 20 --warmup 5) spend its time?  Host timestamps and
stream events around the same calls bench.py's timed() makes."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(torch_collective=False, backend='nccl', force_reduce=False)
dev = torch.device('cuda', 0)
for steps in (20, 20, 100):
    run = bench.Run(args, 10.0, 0, 1, dev, steps + 5 + 50)
    run.fit(5)
    run.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run.fit(steps)
    t1 = time.perf_counter()
    e1.record()
    run.sync()
    t2 = time.perf_counter()
    print('steps %d: host enqueue %.3f ms, wall %.3f ms (%.1f us/step), stream events %.3f ms (%.1f us/step), sync tail after enqueue %.3f ms'
          % (steps, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / steps * 1e6, e0.elapsed_time(e1), e0.elapsed_time(e1) / steps * 1e3, (t2 - t1) * 1e3))
# single calls: wall time of fit(n) + sync for small n (fixed cost of a partial_fit call + n steps)
run = bench.Run(args, 10.0, 0, 1, dev, 400)
run.fit(5); run.sync()
for n in (1, 1, 2, 4, 8, 1, 1):
    t0 = time.perf_counter(); run.fit(n); t1 = time.perf_counter(); run.sync(); t2 = time.perf_counter()
    print('fit(%d): host %.3f ms, wall %.3f ms' % (n, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run.fit(1); pr.disable(); run.sync()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
