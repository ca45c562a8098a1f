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
    out = (C.c_ulonglong * 16)()
    check(lib.modl_somf_debug_stamps(est._backend.plan, out))
    t = np.array(list(out)[:7], dtype=np.float64)
    d = np.diff(t)
    names = ['prologue(loads,CP,apply)', 'A-loads+mfma+red', 'epilogue+gram+stores', 'ticket', 'reduce_partials', 'resolve_wave']
    print('r=%g cycles:' % r, {n: int(v) for n, v in zip(names, d)})
    print('   prologue (WG 0): issue', out[12], ' wait vmcnt(0)', out[13], ' barrier', out[14])
    rs = np.array(list(out)[8:12], dtype=np.float64)
    print('   resolve: start->loads done', int(rs[0] - t[5]), ' steps 0-7', int(rs[1]-rs[0]), ' 8-15', int(rs[2]-rs[1]), ' 16-23', int(rs[3]-rs[2]), ' 24-31', int(t[6]-rs[3]))
