import sys, os, time, argparse, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
import bench
args = argparse.Namespace(torch_collective=False, backend='nccl', force_reduce=False)
dev = torch.device('cuda', 0)
run = bench.Run(args, 10.0, 0, 1, dev, 600)
run.fit(5); run.sync()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    run.fit(1)
pr.disable()
run.sync()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
