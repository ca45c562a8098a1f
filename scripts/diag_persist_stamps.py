"""In-kernel phase stamps of the persistent dictionary-update launch (csrc/bcd_persist.hip; diagnostics build): the
resolver workgroup and row workgroup 0, per block of 32 atoms.  usage: python scripts/diag_persist_stamps.py [r ...]"""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check
if os.environ.get('MODL_DIAG_NO_RIDER') == '1':          # the statistics product of the rows that were not sampled as a launch of its own
    class DictFact(DictFact):
        def _make_backend(self):
            be = super()._make_backend()
            be.flags = 1                                # FLAG_NO_RIDER
            return be
dev = torch.device('cuda')
X = bench.M1Stream(10000, 1234, dev).rows(0, 4096)
for r in [float(a) for a in sys.argv[1:]] or (10.0, 1.0):
    est = DictFact(n_components=256, batch_size=256, reduction=r, code_alpha=1.0, learning_rate=0.92, random_state=0)
    est.prepare(n_samples=4096, X=X[:256])
    est.partial_fit(X[:2048])
    out = (C.c_ulonglong * 192)()
    check(lib.modl_somf_debug_persist_stamps(est._backend.plan, out))
    o = [int(v) for v in out]
    t0 = o[0]
    nblk = 8
    print('r=%g  resolver, cycles since its start' % r)
    print('   block 0: arrivals complete %d, Gram in LDS %d' % (o[1] - t0, o[2] - t0))
    for b in range(nblk):
        rec_done, gram_next = o[3 + 4 * b] - t0, o[4 + 4 * b] - t0
        arr, loaded = o[5 + 4 * b] - t0, o[6 + 4 * b] - t0
        start = (o[4 + 4 * (b - 1)] if b else o[2]) - t0
        print('   block %d: recursion %7d -> S flagged %7d (%5d) | pieces of block %d: arrived %7d, in LDS %7d | barrier %7d | transform done %7d (%5d)' % (
            b, start, o[72 + b] - t0, o[72 + b] - t0 - start, b + 1, arr if b + 1 < nblk else -1, loaded if b + 1 < nblk else -1, rec_done, gram_next, gram_next - rec_done))
    print('   total %d cycles' % (o[4 + 4 * (nblk - 1)] - t0))
    print('   transform of block 3 (cycles after the barrier): P stored + met %d, R / Z stored + met %d, combined %d' % (o[88] - o[3 + 8], o[89] - o[3 + 8], o[90] - o[3 + 8]))
    q = o[96:]
    r0 = q[0]
    print('   row workgroup 0, cycles since ITS start: operands requested %d, row loads requested %d, rows in LDS %d, block 0 candidates %d, pieces issued %d, handed over %d' % (q[5] - r0, q[6] - r0, q[1] - r0, q[2] - r0, q[3] - r0, q[4] - r0))
    for b in range(1, nblk):
        v = [q[7 + 5 * (b - 1) + i] - r0 for i in range(5)]
        prev = (q[11 + 5 * (b - 2)] if b > 1 else q[3]) - r0
        print('   block %d: stable product done %7d (%5d after the last pieces) | S fetched %7d (%5d) | applied + corrected %7d (%5d) | candidates %7d (%5d) | pieces issued %7d (%5d)' % (
            b, v[0], v[0] - prev, v[1] if b > 1 else -1, v[1] - v[0] if b > 1 else 0, v[2], v[2] - (v[1] if b > 1 else v[0]), v[3], v[3] - v[2], v[4], v[4] - v[3]))
    print('   end %d' % (q[89] - r0))
    print('   phase 3 detail (cycles after S fetched): applied %d, barrier %d, corrected %d, + rank 32 %d; signalled %d after pieces issued' % (q[90] - q[8 + 5 * 2], q[91] - q[8 + 5 * 2], q[92] - q[8 + 5 * 2], q[9 + 5 * 2] - q[8 + 5 * 2], q[93] - q[11 + 5 * 2]))
