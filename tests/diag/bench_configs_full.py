"""BASELINE configs 2-4 on one MI355X WITH the CPU oracle timed next to each on a bounded prefix of the same input
(same process, host cores stated) and the roofline fraction of the section that dominates the GPU run.
Lives under tests/ because it runs the oracle (test infrastructure); the GPU side is scripts/bench_configs.py.

    python tests/diag/bench_configs_full.py [--only c2,c3,c4] [--budget 15] > profiles/rNN_bench_configs.jsonl
One JSON line per configuration."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'diag'))

import bench_configs as gpu                                   # noqa: E402
import config_cpu_baselines as cpu                            # noqa: E402

PEAK = {'f32_mfma': 157.3, 'f64_mfma': 78.6, 'hbm': 8000.0}   # TFLOP/s, TFLOP/s, GB/s (MI355X_MICROARCH.md)


def section_roofline(be, k, p, b, s, e, dtype):
    """dominant section of the last profiled pass + its fraction of the roof that bounds it (the work model of
    bench.py / SURVEY 8d at this configuration's shape, measured sweeps)."""
    import bench
    prof = be.prof_get()
    prof = {n: v for n, v in prof.items() if v['calls']}
    if not prof:
        return None
    dom = max(prof, key=lambda n: prof[n]['ms'])
    ms = prof[dom]['ms'] / prof[dom]['calls']
    sweeps = float(be.last_sweeps().mean()) if dom == 'code_solve' else 4.0
    ride = s < p
    fl = bench.step_flops(k, p, b, s, sweeps, ride=ride)
    by = bench.step_bytes(k, p, b, s, e=e, ride=ride)
    tf, gbs = fl[dom] / ms / 1e9, by[dom] / ms / 1e6
    peak_c = PEAK['f32_mfma' if dtype == 'f32' else 'f64_mfma']
    compute_bound = fl[dom] / by[dom] * PEAK['hbm'] / 1e3 > peak_c
    share = prof[dom]['ms'] / sum(v['ms'] for v in prof.values())
    out = dict(section=dom, share_of_gpu_time=share, ms_per_minibatch=ms, sweeps=sweeps if dom == 'code_solve' else None)
    if compute_bound:
        out.update(bound='mfma' if dom != 'code_solve' else 'valu (f32/f64 vector rate = matrix rate; informational)',
                   achieved=tf, peak=peak_c, unit='TFLOP/s', frac=tf / peak_c)
    else:
        out.update(bound='hbm', achieved=gbs, peak=PEAK['hbm'], unit='GB/s', frac=gbs / PEAK['hbm'])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='c2,c3,c4')
    ap.add_argument('--budget', type=float, default=15.0)
    a = ap.parse_args()
    import torch
    from types import SimpleNamespace
    for name in a.only.split(','):
        args = SimpleNamespace(c2_patches=None, c3_records=40, c4_batches=7000, c4_nnz=10_000_000, c6_batches=8)
        rec = dict(gpu=getattr(gpu, name)(args))
        # the section split of a second, shorter run (HIP events on the stream slow it down: not the timed one)
        if name == 'c2':
            from modl_amd.image import ImageDictFact
            img_est = ImageDictFact(patch_size=(8, 8), n_components=256, method='masked', setting='dictionary learning',
                                    random_state=0, n_epochs=1, max_patches=20000)
            img_est.fit(gpu.synth_image(512, 512, 1))
            be2 = img_est.dict_fact_._backend
            rs = np.random.RandomState(0)
            Xp = rs.randn(2000, 64)
            Xp -= Xp.mean(1, keepdims=True)
            Xp /= np.linalg.norm(Xp, axis=1, keepdims=True)
            be2.prof_enable(True); be2.prof_reset()
            img_est.dict_fact_.partial_fit(Xp, np.arange(2000))
            torch.cuda.synchronize()
            rec['roofline'] = section_roofline(be2, 256, 64, 100, 64 / 10.0, 8, 'f64')
            be2.prof_enable(False)
        elif name == 'c3':
            from modl_amd.fmri import fMRIDictFact
            recs, init = gpu.fmri_records(n_records=3)
            est = fMRIDictFact(method='masked', n_components=70, reduction=12, batch_size=20, alpha=1e-3, learning_rate=0.92,
                               dict_init=init, random_state=0, n_epochs=1)
            est.fit(recs[:2])
            be = est.dict_fact_._backend
            be.prof_enable(True); be.prof_reset()
            est.dict_fact_.partial_fit(recs[2], None)
            torch.cuda.synchronize()
            rec['roofline'] = section_roofline(be, 70, 60000, 20, 60000 / 12.0, 4, 'f32')
            be.prof_enable(False)
        elif name == 'c6':
            g = rec['gpu']
            sec = g['sections_ms']
            dom = max(sec, key=sec.get)
            kw = gpu.HCP_KW
            k, b, p, e = kw['n_components'], kw['batch_size'], 200000, 4
            s_ = p / kw['reduction']
            # the l1 projection of 1024 atoms is a chain of per-atom passes over the s sampled features: algorithmic bytes
            # of the dictionary update = read + write the sampled columns of D and read those of B_ once, plus C_
            by = e * (k * k + 3 * s_ * k)
            rec['roofline'] = dict(section=dom, share_of_gpu_time=sec[dom] / sum(sec.values()), ms_per_minibatch=sec[dom],
                                   bound='hbm', achieved=by / sec[dom] / 1e6, peak=PEAK['hbm'], unit='GB/s',
                                   frac=by / sec[dom] / 1e6 / PEAK['hbm'],
                                   note='dictionary update with l1 atoms and more than 6144 sampled features: one launch per '
                                        'atom (atom_corr_project_kernel: candidates on 39 workgroups, the last one projects from '
                                        'registers; the next group\'s gradient rows ride along), 1024 dependent launches per minibatch')
        else:
            # the masked path has its own plan (no section events): algorithmic HBM bytes per rating - its code row,
            # the item's dictionary column and the read-modify-write of its B_ column: (3 k + k) e bytes, f64
            g = rec['gpu']
            by = g['ratings_per_s'] * (4 * 50) * 8 / 1e9
            rec['roofline'] = dict(section='whole minibatch (masked path)', bound='hbm', achieved=by, peak=PEAK['hbm'],
                                   unit='GB/s', frac=by / PEAK['hbm'],
                                   note='latency-bound chain of small launches per minibatch of 10 rows; see the kernel trace')
        base = getattr(cpu, name)(a.budget)
        base['cores'] = os.cpu_count()
        base['kind'] = 'port (oracle/)'
        rec['cpu_baseline'] = base
        rec['gpu_over_cpu'] = rec['gpu']['samples_per_s'] / base['samples_per_s']
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
