import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
dev = torch.device('cuda')
X = bench.make_stream(4096, 10000, 1234, dev)
for r in (10, 1):
    est = DictFact(n_components=256, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=4096, X=X[:256])
    est.partial_fit(X[:2048])
    out = (C.c_ulonglong * 8)()
    check(lib.modl_somf_debug_stamps(est._backend.plan, out))
    t = np.array(list(out)[:7], dtype=np.float64)
    d = np.diff(t)
    names = ['prologue(loads,CP,apply)', 'A-loads+mfma+red', 'epilogue+gram+stores', 'ticket', 'reduce_partials', 'resolve_wave']
    print('r=%g' % r, {n: round(v / 100.0, 2) for n, v in zip(names, d)}, 'us (100 MHz s_memtime)')
