"""Cycle breakdown of the grouped atom update (bcd.hip: atom_grad_group_kernel + atom_project_group_kernel) on the fMRI
shape (config 3: k = 70 l1 atoms, p = 60 000, reduction 12): time per minibatch, then the sums over the launches of the
projecting workgroup's stamps (MODL_DEBUG_ATOM_STAMPS; the stamps themselves cost a memory round trip per atom)."""
import os as _os
_os.environ.setdefault('MODL_AMD_DIAG', '1')     # the stamps only exist in the diagnostics build (libmodl_hip_diag.so)
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from modl_amd import DictFact
from modl_amd._lib import lib, check, DEBUG_ATOM_STAMPS
dev = torch.device('cuda')
p, n, b, k = 60000, 1400, 20, 70
X = bench.M1Stream(p, 3, dev, k0=64).rows(0, n)


def fit(stamps):
    est = DictFact(n_components=k, batch_size=b, reduction=12, code_alpha=1e-3, code_l1_ratio=0, comp_l1_ratio=1.0,
                   learning_rate=0.92, random_state=0)
    est.prepare(n_samples=n, X=X[:k])
    est.partial_fit(X[:400], np.arange(400))
    st = torch.zeros(64, dtype=torch.int64, device=dev)
    if stamps:
        check(lib.modl_debug_set(DEBUG_ATOM_STAMPS, st.data_ptr()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[400:1400], np.arange(400, 1400))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    check(lib.modl_debug_set(DEBUG_ATOM_STAMPS, 0))
    return dt / (1000 // b) * 1e3, est.components_.copy(), st.cpu().numpy().astype(np.float64)


ms0, D0, _ = fit(False)
ms1, D1, _ = fit(False)
print('%.3f / %.3f ms per minibatch; run-to-run identical: %s' % (ms0, ms1, np.array_equal(D0, D1)))
_, _, o = fit(True)
nl = o[0]
print('projection launches %d: prologue %.0f, whole kernel %.0f cycles' % (nl, o[3] / nl, o[20] / nl))
for a in range(4):
    print('   atom %d: %.0f cycles (%.2f passes, %.2f warm)' % (a, o[4 + 4 * a] / nl, o[5 + 4 * a] / nl, o[7 + 4 * a] / nl))
na = max(o[28], 1)
print('   per atom: corrections + candidate %.0f, level search %.0f, output row + norm + change %.0f' %
      (o[24] / na, o[25] / na, o[26] / na))
if o[40]:
    print('ridge_small_kernel (thread 0): load %.0f, factor %.0f, substitutions + stores %.0f cycles' % (o[41] / o[40], o[42] / o[40], o[43] / o[40]))
