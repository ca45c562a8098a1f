"""Parity at the FULL shape of the reference's HCP configuration (C6: k = 1024 positive l1 atoms, ridge codes, b = 200, r = 20,
p = 200 000 - scripts/bench_configs.py: HCP_KW, hcp_rows): the GPU estimator against the oracle in f64 and in f32 on the same
float32 records, over N minibatches from identical state (the oracle needs about a minute per minibatch and precision).
usage (GPU box): python scripts/parity_c6_full_shape.py [N=2]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench_configs as bc  # noqa: E402
from modl_amd import DictFact  # noqa: E402
from oracle import somf_oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = dict(bc.HCP_KW)
b, k, p = kw['batch_size'], kw['n_components'], 200000
n = max(k, N * b)
Xd = bc.hcp_rows(n, p, 0, torch.device('cuda'))
X = Xd.cpu().numpy()
rel = lambda a, ref: float(np.linalg.norm(np.asarray(a, np.float64) - ref) / np.linalg.norm(ref))
est = DictFact(**kw)
est.prepare(n_samples=n, X=X[:k])
t0 = time.perf_counter()
est.partial_fit(X[:N * b], np.arange(N * b))
gpu_s = time.perf_counter() - t0
out = dict(shape=dict(k=k, p=p, b=b, reduction=kw['reduction'], minibatches=N), gpu_s=gpu_s)
pr = orc.SomfParams(n_threads=min(32, os.cpu_count() or 1), **kw)
states = {}
for name, dt in (('f64', np.float64), ('f32', np.float32)):
    t0 = time.perf_counter()
    st = orc.prepare(pr, n_samples=n, X=X[:k].astype(dt))
    orc.partial_fit(st, pr, X[:N * b].astype(dt), np.arange(N * b))
    states[name] = st
    out['oracle_%s_s' % name] = time.perf_counter() - t0
D64, C64 = states['f64'].D, states['f64'].code[:N * b]
D = est.components_
out['gpu_vs_f64'] = dict(D=rel(D, D64), code=rel(est.code_[:N * b], C64), comp_norm=rel(est.comp_norm_, states['f64'].comp_norm))
out['oracle_f32_vs_f64'] = dict(D=rel(states['f32'].D, D64), code=rel(states['f32'].code[:N * b], C64),
                               comp_norm=rel(states['f32'].comp_norm, states['f64'].comp_norm))
out['gpu_vs_oracle_f32'] = dict(D=rel(D, states['f32'].D.astype(np.float64)))
out['nonneg'] = bool((D >= 0).all())
out['share_nonzero'] = dict(gpu=float((D != 0).mean()), oracle_f64=float((D64 != 0).mean()))
out['supports_equal_share'] = float(((D != 0) == (D64 != 0)).mean())
print(json.dumps(out))
