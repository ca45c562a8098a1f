"""Calibrates the speed of the CPU port (oracle/somf_oracle.py, what bench.py's `cpu_baseline` leg times on the GPU
box) against the REAL reference, in the build container.

    python scripts/calibrate_cpu_baseline.py          # needs /root/reference, gcc, Cython (like make_golden.py)

The reference is compiled into the scratch directory of tests/golden/make_golden.py (outside the repository) and
imported from there.  Both implementations run the same inputs on the same cores, one after the other:
  * C1: 2000 x 500, k = 16, reduction = 1, b = 10 (default), f64, one epoch, through `fit`;
  * a 4096-row prefix of the M1 recipe (tests/conftest.py::m1_rows): k = 256, p = 10 000, b = 256, f32,
    reduction in {1, 10}, through prepare + partial_fit.
Output: profiles/r02_cpu_calibration.json  (ratio = port samples/s / reference samples/s; > 1: the port is faster).
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_reference():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.build_reference()
    sys.path.insert(0, mg.SCRATCH)
    from modl.decomposition.dict_fact import DictFact
    return DictFact, mg


def best_of(fn, reps):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def main():
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle import somf_oracle as orc
    RefDictFact, mg = load_reference()
    cores = os.cpu_count()
    out = dict(cores=cores, blas=[dict(api=i.get('internal_api'), threads=i.get('num_threads'), version=i.get('version'))
                                  for i in threadpool_info() if i.get('user_api') == 'blas'], cases=[])
    with threadpool_limits(limits=cores, user_api='blas'):
        # ---- C1
        X = mg.synth(2000, 500, 16, 0, np.float64)
        kw = dict(n_components=16, reduction=1, random_state=0, n_epochs=1, code_alpha=1e-4)
        t_ref = best_of(lambda: RefDictFact(**kw).fit(X), 3)
        t_port = best_of(lambda: orc.fit(orc.SomfParams(**kw), X), 3)
        out['cases'].append(dict(name='C1 (2000x500, k=16, r=1, b=10, f64, fit)', rows=2000, ref_samples_s=2000 / t_ref,
                                 port_samples_s=2000 / t_port, ratio=t_ref / t_port))
        # ---- M1 prefix
        n, p, k, b = 4096, 10000, 256, 256
        X32 = mg.m1_rows(n, p)
        for r in (10, 1):
            kw = dict(n_components=k, batch_size=b, reduction=r, code_alpha=1.0, code_l1_ratio=1, comp_l1_ratio=0,
                      learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)

            def run_ref():
                est = RefDictFact(**kw)
                est.prepare(n_samples=n, X=X32)
                est.partial_fit(X32)

            def run_port():
                pr = orc.SomfParams(**kw)
                st = orc.prepare(pr, n_samples=n, X=X32)
                orc.partial_fit(st, pr, X32)

            t_ref, t_port = best_of(run_ref, 2), best_of(run_port, 2)
            out['cases'].append(dict(name='M1 prefix (4096x10000, k=256, b=256, f32, reduction=%d)' % r, rows=n,
                                     ref_samples_s=n / t_ref, port_samples_s=n / t_port, ratio=t_ref / t_port))
    m1 = out['cases'][1]
    out['summary'] = dict(ref_samples_s=m1['ref_samples_s'], port_samples_s=m1['port_samples_s'], cores=cores,
                          ratio_port_over_ref=m1['ratio'], case=m1['name'],
                          where='build container (%d cores), real reference compiled from /root/reference' % cores)
    path = os.path.join(ROOT, 'profiles', 'r02_cpu_calibration.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
