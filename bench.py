"""Headline benchmark: samples/sec through DictFact.partial_fit at k = 256, p = 10k
(BASELINE.json), on N MI355X of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one SOMF minibatch (code solve, statistics, dictionary update) of
256 rows per GPU.  The synthetic stream M1 of SURVEY.md §8(d) is generated on
the device and is resident in HBM when the timed region starts.  With N > 1 every
rank works on its own rows of a global minibatch of N*256 rows and the statistics
increment is all-reduced over RCCL before each dictionary update (weak scaling).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).  The
`cpu_baseline` leg times the CPU oracle (oracle/somf_oracle.py: numpy + OpenBLAS
for the contractions, C for the solver/projection — the reference's algorithm and
operation order) on a bounded prefix of the same stream on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_COMP, P_FEAT, BATCH, CHUNK = 256, 10000, 256, 65536
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0
PMC_FILE = 'r01_o_pmc_hbm_traffic.json'
DOM_KERNEL = {'dict_update': 'modl::bcd_block_kernel', 'code_solve': 'modl::cd_kernel', 'stats_gemm': 'modl::gemm_stats_pair_kernel',
              'code_gemm': 'modl::gemm_dense_pair_kernel<float, false', 'stats_apply': 'modl::stats_apply2_kernel'}


def make_stream(n_rows, p, seed, device, k0=256, density=0.1, noise=0.1, row_seed=None):
    """M1 stream: X = (Z o M) Q / sqrt(density k0) + noise E  (unit-variance entries).  The mixing matrix Q is drawn
    from `seed` (the same on every rank), the rows from `row_seed` (one stream of rows per rank)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    Q = torch.randn(k0, p, device=device, generator=g)
    if row_seed is not None:
        g = torch.Generator(device=device).manual_seed(row_seed)
    X = torch.empty(n_rows, p, device=device, dtype=torch.float32)
    step = 8192
    for r0 in range(0, n_rows, step):
        r1 = min(n_rows, r0 + step)
        Z = torch.randn(r1 - r0, k0, device=device, generator=g)
        M = (torch.rand(r1 - r0, k0, device=device, generator=g) < density).float()
        X[r0:r1] = (Z * M) @ Q / (density * k0) ** 0.5 + noise * torch.randn(r1 - r0, p, device=device, generator=g)
    return X


def step_flops(k, p, b, s, sweeps, ride=False):
    """Algorithmic flops of one minibatch (SURVEY.md §8d work model, blocked dictionary update).
    ride: the single-GPU step with a proper feature subset — the statistics section only carries the head
    (C increment + sampled rows of the B increment); the p x k product rides along the dictionary update's launches."""
    dx = 2.0 * b * s * k
    gram = 2.0 * k * k * s
    h0 = 2.0 * b * k * k
    cd = 4.0 * k * k * sweeps * b                  # two axpy(k) per coordinate, upper bound
    c_inc = 2.0 * k * k * b
    b_inc = 2.0 * b * k * p
    bcd = 2.0 * k * k * s
    if ride:
        return dict(code_gemm=dx + gram, code_solve=h0 + cd, stats_gemm=c_inc + 2.0 * b * k * s,
                    stats_apply=3.0 * (k * k + p * k), dict_update=bcd + b_inc)
    return dict(code_gemm=dx + gram, code_solve=h0 + cd, stats_gemm=c_inc + b_inc, stats_apply=3.0 * (k * k + p * k),
                dict_update=bcd)


def step_bytes(k, p, b, s, e=4, ride=False):
    """Algorithmic HBM bytes per minibatch per section (read each operand once, write each result once)."""
    if ride:                                       # B_ is read and written once by the riding product
        return dict(code_gemm=e * (b * s + s * k + b * k + k * k),
                    code_solve=e * (k * k + 3 * b * k),
                    stats_gemm=e * (b * s + b * k + 2 * k * k + 2 * s * k),
                    stats_apply=e * 3 * (k * k + p * k),
                    dict_update=e * (k * k + 3 * s * k) + e * (b * p + b * k + 2 * (p - s) * k))
    return dict(code_gemm=e * (b * s + s * k + b * k + k * k),
                code_solve=e * (k * k + 3 * b * k),
                stats_gemm=e * (b * p + b * k + k * k + p * k),
                stats_apply=e * 3 * (k * k + p * k),
                dict_update=e * (k * k + 3 * s * k))


def run_gpu(args, reduction, steps, warmup, rank, world, device, breakdown=True):
    """Warm-up (all sections timed -> picks the dominant one), the timed region (HIP events around the
    dominant section only: every timed section costs two event records, i.e. a stream bubble of a few
    microseconds each), then an untimed pass with all sections timed for the breakdown."""
    import torch
    import torch.distributed as dist
    from modl_amd import DictFact
    extra = min(steps, 50) if breakdown else 0
    n_rows = min(CHUNK, max(4096, (steps + warmup + extra) * BATCH))
    X = make_stream(n_rows, P_FEAT, 1234, device, row_seed=None if world == 1 else 5000 + rank)
    # the dictionary is initialised from the same rows on every rank (replicas stay identical: same init, same draws,
    # all-reduced statistics)
    X0 = X[:K_COMP] if world == 1 else make_stream(K_COMP, P_FEAT, 1234, device, row_seed=4999)
    est = DictFact(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
                   comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    est.prepare(n_samples=n_rows, X=X0)
    if getattr(args, 'force_reduce', False):                 # testing only: the N > 1 step with one rank
        est._two_phase = True
        est._force_reduce = True

    def run(nsteps, start_step):
        done = 0
        while done < nsteps:
            r0 = ((start_step + done) * BATCH) % n_rows
            todo = min(nsteps - done, (n_rows - r0) // BATCH)
            est.partial_fit(X[r0:r0 + todo * BATCH], np.arange(r0, r0 + todo * BATCH))
            done += todo

    be = est._backend
    # the dominant section is picked on the LAST quarter of the warm-up (the first minibatches of a fresh
    # dictionary need several times more solver sweeps than the steady state)
    w_head = warmup - max(warmup // 4, 1) if warmup >= 4 else warmup
    run(w_head, 0)
    be.prof_enable(True)
    be.prof_reset()
    run(warmup - w_head, w_head)
    torch.cuda.synchronize()
    pre = be.prof_get()
    dom = max(pre, key=lambda n: pre[n]['ms']) if warmup - w_head > 0 and pre else 'dict_update'
    be.prof_enable(False)
    be.prof_enable(True, sections=[dom])
    be.prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, warmup)
    run_gpu.host_ms_per_step = (time.perf_counter() - t0) / steps * 1e3   # host time to ENQUEUE a step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof_dom = be.prof_get()[dom]
    be.prof_enable(False)
    lsw = be.last_sweeps()
    sweeps = float(lsw.mean())
    run_gpu.sweeps_max = int(lsw.max())
    prof = {}
    if extra:
        be.prof_enable(True)
        be.prof_reset()
        run(extra, warmup + steps)
        torch.cuda.synchronize()
        prof = be.prof_get()
        be.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    D = est.components_
    ok = bool(np.all(np.isfinite(D)))
    run_gpu.replicas_identical = None
    if world > 1:                                  # the replicas of the dictionary must agree bit for bit
        chk = torch.stack([be.Dt.double().sum(), (be.Dt.double() ** 2).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        run_gpu.replicas_identical = bool(torch.equal(lo, hi))
    return dt, prof, sweeps, ok, dom, prof_dom


def cpu_baseline(reduction, budget_s=20.0):
    """CPU oracle on a bounded prefix of the same stream, all host cores for BLAS."""
    import torch
    from oracle import somf_oracle as orc
    cores = os.cpu_count() or 1
    n_rows = 40 * BATCH                 # bounded sample: the loop below stops after budget_s seconds
    X = make_stream(n_rows, P_FEAT, 1234, torch.device('cpu')).numpy()
    pr = orc.SomfParams(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
                        comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    st = orc.prepare(pr, n_samples=n_rows, X=X[:K_COMP])
    done, t0 = 0, time.perf_counter()
    for r0 in range(0, n_rows, BATCH):
        orc.partial_fit(st, pr, X[r0:r0 + BATCH], np.arange(r0, r0 + BATCH))
        done += BATCH
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit='samples/s', cores=cores, kind='port',
                sample='first %d rows of stream M1 (p=%d, k=%d, b=%d, reduction=%g), %.1f s; numpy/OpenBLAS threads=%d '
                       'for the contractions, single-thread C for the CD solver and projections'
                       % (done, P_FEAT, K_COMP, BATCH, reduction, dt, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # the first few hundred minibatches of a fresh dictionary need several times more solver sweeps than the steady
    # state: the default warm-up covers them (the whole default run takes ~3 s of GPU time + the CPU baseline)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=500)
    ap.add_argument('--reduction', type=float, default=10.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--also-r1', action='store_true', help='also time reduction=1 (OMF) and report it under "also"')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to test the N > 1 path)')
    ap.add_argument('--force-reduce', action='store_true',
                    help='testing only: run the multi-GPU step (two phases + RCCL all-reduces) even with one rank')
    ap.add_argument('--share-gpu', action='store_true', help='testing only: every rank uses cuda:0 (needs --backend gloo)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 or args.force_reduce:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    dt, prof, sweeps, ok, dom, prof_dom = run_gpu(args, args.reduction, args.steps, args.warmup, rank, world, device)
    out = None
    if rank == 0:
        samples = args.steps * BATCH * world
        s_mean = P_FEAT / args.reduction
        ride = world == 1 and args.reduction > 1 and not args.force_reduce and not os.environ.get('MODL_NO_RIDER')
        fl = step_flops(K_COMP, P_FEAT, BATCH, s_mean, sweeps, ride=ride)
        by = step_bytes(K_COMP, P_FEAT, BATCH, s_mean, ride=ride)
        sections = {}
        for name, e in prof.items():
            if e['calls'] == 0:
                continue
            ms = e['ms'] / e['calls']
            sections[name] = dict(ms_per_step=ms, launches_per_step=e['launches'] / e['calls'],
                                  gflops=fl[name] / ms / 1e6, gbs=by[name] / ms / 1e6)
        # roofline of the dominant section, from the events recorded INSIDE the timed region
        roof = None
        if prof_dom['calls'] > 0:
            ms = prof_dom['ms'] / prof_dom['calls']
            nl = prof_dom['launches'] / prof_dom['calls']
            gflops, gbs = fl[dom] / ms / 1e6, by[dom] / ms / 1e6
            ai = fl[dom] / by[dom]
            if ai * PEAK_HBM_GBS / 1e3 > PEAK_MFMA_F32_TFLOPS:       # ridge of the f32 roofline
                roof = dict(bound='mfma', kernel=dom, achieved=gflops / 1e3, peak=PEAK_MFMA_F32_TFLOPS,
                            unit='TFLOP/s', frac=gflops / 1e3 / PEAK_MFMA_F32_TFLOPS, traffic=None)
            else:
                roof = dict(bound='hbm', kernel=dom, achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s',
                            frac=gbs / PEAK_HBM_GBS, traffic=None)
            roof['ms_per_step'] = ms
            roof['launches_per_step'] = nl
            roof['avg_launch_ms'] = ms / max(nl, 1)
            roof['algorithmic_per_launch'] = dict(flops=fl[dom] / max(nl, 1), bytes=by[dom] / max(nl, 1))
            # HBM traffic per launch of the section's main kernel: from the committed rocprofv3 --pmc passes
            # (FETCH_SIZE and WRITE_SIZE need separate passes and cannot be collected from inside this process)
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', PMC_FILE)))
                kern = [k_ for k_ in pmc if k_.startswith(DOM_KERNEL.get(dom, '?'))]
                if kern and abs(args.reduction - 10.0) < 1e-9:
                    e = pmc[kern[0]]
                    roof['traffic'] = e.get('fetch_bytes_corrected', 0.0) + e.get('write_bytes', 0.0)
                    roof['traffic_source'] = 'profiles/%s: %s, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per launch' % (
                        PMC_FILE, kern[0])
            except (OSError, ValueError):
                pass
        total_fl = sum(fl.values())
        out = dict(metric='samples/sec through DictFact.partial_fit at k=256, p=10k', value=samples / dt,
                   unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload='M1 stream: %d-row resident chunk x p=%d f32, n_components=%d, batch_size=%d/GPU, '
                                        'reduction=%g, code_alpha=1 (l1 codes), l2 atoms, learning_rate=0.92, '
                                        'masked/masked' % (CHUNK, P_FEAT, K_COMP, BATCH, args.reduction),
                               reduction=args.reduction, global_batch=BATCH * world,
                               parallelism='dp%d (row-sharded minibatch; all-reduce of the C increment and the sampled rows of the B increment, '
                                           'the rest of the B increment all-reduced under the dictionary update)' % world),
                   roofline=roof, sections=sections, cd_sweeps_mean=sweeps, cd_sweeps_max=getattr(run_gpu, 'sweeps_max', None),
                   step_tflops=total_fl / (dt / args.steps) / 1e12, finite=ok,
                   replicas_identical=getattr(run_gpu, 'replicas_identical', None),
                   host_enqueue_ms_per_step=getattr(run_gpu, 'host_ms_per_step', None))
    if args.also_r1:
        dt1, prof1, sw1, ok1, _, _ = run_gpu(args, 1.0, max(args.steps // 2, 10), max(args.warmup // 2, 2), rank, world, device,
                                             breakdown=False)
        if rank == 0:
            n1 = max(args.steps // 2, 10)
            out['also'] = dict(reduction_1=dict(value=n1 * BATCH * world / dt1, ms_per_step=dt1 / n1 * 1e3,
                                                cd_sweeps_mean=sw1, finite=ok1))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.reduction)
        else:
            out['cpu_baseline'] = None
    if world > 1 or args.force_reduce:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to C stdio, which is flushed at exit when stdout is a pipe: push it out
        # first, so that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.write(json.dumps(out) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
