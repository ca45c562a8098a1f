"""C2 shape (ImageDictFact: p = 64, k = 256, b = 100, r = 10): solver sweep counts and section times, f32 and f64."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import synth_image
from modl_amd import DictFact
from modl_amd.image import LazyCleanPatchExtractor, _flatten_patches
img = synth_image(512, 512, 1)
ext = LazyCleanPatchExtractor(patch_size=(8, 8), random_state=0).fit(img)
n = 30000
P = _flatten_patches(ext.partial_transform(batch=slice(0, n)), copy=True)
for dt in (np.float64, np.float32):
    X = torch.from_numpy(P.astype(dt)).cuda()
    est = DictFact(n_components=256, batch_size=100, reduction=10, code_alpha=0.1, code_l1_ratio=1, comp_l1_ratio=0,
                   learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0, tol=1e-2)
    est.prepare(n_samples=n, X=P[:256].astype(dt))
    be = est._backend
    sw = []
    for r0 in range(0, 10000, 100):
        est.partial_fit(X[r0:r0 + 100], np.arange(r0, r0 + 100))
        s = be.last_sweeps()
        sw.append((s.mean(), s.max()))
    sw = np.array(sw)
    print(dt.__name__, 'sweeps mean/max per step: first 5', sw[:5].tolist(), ' last 5', sw[-5:].tolist(), 'overall mean %.1f max %d' % (sw[:, 0].mean(), sw[:, 1].max()))
    code = be.get('code')[:10000]
    print('  nnz per code row: mean %.1f max %d' % ((code != 0).sum(1).mean(), (code != 0).sum(1).max()))
    be.prof_enable(True); be.prof_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.partial_fit(X[10000:30000], np.arange(10000, 30000))
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    pr = be.prof_get()
    print('  %.3f ms per step;' % (t / 200 * 1e3), {k_: round(v['ms'] / max(v['calls'], 1), 4) for k_, v in pr.items()})
    be.prof_enable(False)
