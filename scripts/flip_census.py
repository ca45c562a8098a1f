"""The sweep-flip census of DESIGN section 5 at a chosen size, with another seed than the committed one (bench.py: flip_rate_block:
free-running fits of the same fresh rows by the f64 oracle, the f32 oracle and the GPU estimator; a flip = a sample whose sweep
count differs from the f64 run's).  usage (GPU box): python scripts/flip_census.py <minibatches r=10> <minibatches r=1> [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

n10, n1 = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 97531
out = []
for r, n in ((10.0, n10), (1.0, n1)):
    if n > 0:
        out.append(bench.flip_rate_block(r, n, torch.device('cuda', 0), seed=seed, log=lambda m: sys.stderr.write(m + '\n')))
print(json.dumps(out))
