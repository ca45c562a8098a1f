"""Shader-clock stamps of sample 0 of the four-wavefront solver (cd_split_impl.hpp): chain wave block starts / ends and
update wave group starts / H publications.  python scripts/diag_cd_split_stamps.py [k b p]
(the stamp switch only exists in the diagnostics build of the library: libmodl_hip_diag.so)"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from modl_amd import dict_fact_fast as fast  # noqa: E402
from modl_amd._lib import check, load_diag  # noqa: E402

lib = load_diag()

k, b, p = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 256, 1000)
dt = np.float32
rs = np.random.RandomState(0)
D = rs.randn(k, p).astype(dt)
D /= np.sqrt((D ** 2).sum(1))[:, None]
X = np.ascontiguousarray(((rs.randn(b, k) * (rs.rand(b, k) < 0.1)).dot(D) + 0.1 * rs.randn(b, p)).astype(dt))
G = np.ascontiguousarray(D.dot(D.T).astype(dt))
G = (G + G.T) / 2
Dx = np.ascontiguousarray(X.dot(D.T).astype(dt))
idx = np.arange(b, dtype=np.int64)
st = torch.zeros(1024, dtype=torch.int64, device='cuda')
for rep in range(2):
    st.zero_()
    check(lib.modl_debug_set(3, st.data_ptr()))
    code = np.ones((b, k), dtype=dt)
    sw = np.zeros(b, dtype=np.int32)
    fast._enet_regression_single_gram(G, Dx.copy(), X, code, idx, 1.0, 0.3, False, 1e-2, 100, sweeps=sw, _lib=lib)
    torch.cuda.synchronize()
check(lib.modl_debug_set(3, 0))
s = st.cpu().numpy()
c, u, ld = s[:512], s[512:768], s[768:]
c, u, ld = c[c > 0], u[u > 0], ld[ld > 0]
t0 = min(c[0], u[0])
print('sample 0: %d sweeps; all samples mean %.2f max %d' % (sw[0], sw.mean(), sw.max()))
print('chain wave (cycles since first stamp): block start -> block end (chain cycles), wait for H of next block')
for i in range(0, len(c) - 1, 2):
    nxt = c[i + 2] - c[i + 1] if i + 2 < len(c) else 0
    print('  blk %3d: start %8d  chain %6d  then wait %6d' % (i // 2, c[i] - t0, c[i + 1] - c[i], nxt))
print('update wave: stamps (cycles since first), deltas')
print('  ', [int(v - t0) for v in u[:40]])
print('  deltas', [int(v) for v in np.diff(u[:60])])
print('row loader: stamps (issue start, issue end, landed published ... in program order), deltas')
print('  ', [int(v - t0) for v in ld[:60]])
print('  deltas', [int(v) for v in np.diff(ld[:120])])
