"""Headline benchmark: samples/sec through DictFact.partial_fit at k = 256, p = 10k
(BASELINE.json), on N MI355X of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      (no launcher: starts its N ranks itself, see self_launch)

One "step" = one SOMF minibatch (code solve, statistics, dictionary update) of
256 rows per GPU.  The synthetic stream M1 of SURVEY.md §8(d) is produced on the
device, chunk by chunk (65 536 rows), by a counter-based generator: block j of
8192 rows is a pure function of (seed, rank, j).  NO ROW IS EVER FITTED TWICE,
and every row a run will fit is produced BEFORE its timed region and is resident
in HBM when it starts (2.6 GB per chunk; beyond 96 GB the stream falls back to a
two-chunk ring fed from a side stream).  The estimator follows the streaming
protocol of the reference (`partial_fit(chunk)` with sample_indices = None: row i
of a chunk warm-starts from `code_[i]`, the code of row i of the PREVIOUS chunk -
dict_fact.py:313-337).  With N > 1 every rank fits its own rows of a global
minibatch of N*256 rows and the head of the statistics is all-reduced over RCCL
before each dictionary update (weak scaling).

Rank 0 prints ONE JSON line.  `value` is the driver-timed run (W warm-up steps
from a fresh dictionary, then exactly K steps).  `steady_state` holds, from the
same process, >= 2000 fresh-row steps after a 500-step burn-in for reduction = 10
AND reduction = 1 (OMF), each with its `cd_sweeps_mean`.  The `cpu_baseline` leg
times the CPU oracle (oracle/somf_oracle.py: the reference's algorithm and
operation order) on a bounded prefix of the same rows on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_COMP, P_FEAT, BATCH, CHUNK, BLOCK = 256, 10000, 256, 65536, 8192
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0
PMC_FILES = ('r06_pmc_hbm_traffic.json', 'r05_pmc_hbm_traffic.json', 'r04_pmc_hbm_traffic.json', 'r03_pmc_hbm_traffic.json')     # the newest committed PMC passes
DOM_KERNEL = {'dict_update': ('modl::bcd_persist_kernel', 'modl::bcd_block_kernel'), 'code_solve': 'modl::cd_split_kernel', 'stats_gemm': 'modl::gemm_stats_pair_kernel',
              'code_gemm': 'modl::gemm_dense_pair_kernel<float, false', 'stats_apply': 'modl::stats_apply2_kernel'}
# the calibration of the CPU port against the real reference, measured in the build container
# (scripts/calibrate_cpu_baseline.py -> profiles/r02_cpu_calibration.json)
CALIBRATION_FILE = 'r02_cpu_calibration.json'


class M1Stream:
    """Stream M1: X = (Z o M) Q / sqrt(density k0) + noise E, unit-variance entries.

    Counter-based: the mixing matrix Q is a function of `seed` (the same on every rank), block j (BLOCK rows) of rank
    `rank` a function of (seed, rank, j) - torch's Philox generator re-seeded per block - so any block can be produced
    independently and in any order.  `resident(total_rows)`: every chunk a run will read is produced up front and stays
    in HBM (the timed region then starts with its inputs resident, as the bench contract asks; 2.6 GB per chunk, up to
    RESIDENT_BYTES).  Beyond that budget `chunk(c)` works as a ring: chunk c in one of two HBM buffers, chunk c + 1
    produced into the other one on a side stream while c is being fitted."""

    RESIDENT_BYTES = 96 << 30

    def __init__(self, p, seed, device, rank=0, k0=256, density=0.1, noise=0.1, chunk_rows=None):
        import torch
        self.torch = torch
        self.p, self.seed, self.device, self.rank = p, seed, device, rank
        self.k0, self.density, self.noise, self.chunk_rows = k0, density, noise, chunk_rows or CHUNK
        g = torch.Generator(device=device).manual_seed(seed)
        self.Q = torch.randn(k0, p, device=device, generator=g) / (density * k0) ** 0.5
        self.gen = torch.Generator(device=device)
        self.cuda = device.type == 'cuda'
        self.buf = [None, None]
        self.have = [-1, -1]                       # chunk held (or being produced) by each buffer
        if self.cuda:
            self.side = torch.cuda.Stream(device=device)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.free = [None, None]
        self.generated_rows = 0
        self.held = None                           # resident mode: the chunks, all of them

    def resident(self, total_rows):
        """produce every chunk the run will read now (if they fit the budget); returns True in resident mode"""
        torch = self.torch
        n_chunks = max(1, -(-int(total_rows) // self.chunk_rows))
        if not self.cuda or n_chunks * self.chunk_rows * self.p * 4 > self.RESIDENT_BYTES:
            return False
        self.held = []
        for c in range(n_chunks):
            buf = torch.empty(self.chunk_rows, self.p, device=self.device, dtype=torch.float32)
            for a in range(0, self.chunk_rows, BLOCK):
                self.block_into(buf[a:a + BLOCK], (c * self.chunk_rows + a) // BLOCK)
            self.held.append(buf)
        torch.cuda.synchronize(self.device)
        return True

    def block_into(self, out, j):
        """rows [j * BLOCK, j * BLOCK + len(out)) of this rank's stream"""
        torch = self.torch
        n = out.shape[0]
        self.gen.manual_seed((self.seed * 1000003 + self.rank) * 1000003 + j)
        Z = torch.randn(n, self.k0, device=self.device, generator=self.gen)
        M = torch.rand(n, self.k0, device=self.device, generator=self.gen) < self.density
        out.normal_(0.0, self.noise, generator=self.gen)           # noise E, in place (one pass over the block)
        out.addmm_(Z * M, self.Q)
        self.generated_rows += n

    def rows(self, r0, r1):
        """a fresh tensor with rows [r0, r1) (r0 on a block boundary)"""
        assert r0 % BLOCK == 0
        out = self.torch.empty(r1 - r0, self.p, device=self.device, dtype=self.torch.float32)
        for a in range(r0, r1, BLOCK):
            self.block_into(out[a - r0:min(a + BLOCK, r1) - r0], a // BLOCK)
        return out

    def _produce(self, c, slot):
        torch = self.torch
        if self.buf[slot] is None:
            self.buf[slot] = torch.empty(self.chunk_rows, self.p, device=self.device, dtype=torch.float32)
        self.have[slot] = c
        r0 = c * self.chunk_rows
        for a in range(0, self.chunk_rows, BLOCK):
            self.block_into(self.buf[slot][a:a + BLOCK], (r0 + a) // BLOCK)

    def chunk(self, c, prefetch_next=True):
        torch = self.torch
        if self.held is not None:
            return self.held[c]
        slot = c % 2
        if not self.cuda:
            if self.have[slot] != c:
                self._produce(c, slot)
            return self.buf[slot]
        main = torch.cuda.current_stream(self.device)
        if self.have[slot] != c:                                   # not prefetched: produce it now
            self._launch(c, slot, main)
        main.wait_event(self.ready[slot])
        if prefetch_next and self.have[1 - slot] != c + 1:
            self._launch(c + 1, 1 - slot, main)
        return self.buf[slot]

    def _launch(self, c, slot, main):
        torch = self.torch
        if self.buf[slot] is None:                                 # allocate on the main stream's pool
            self.buf[slot] = torch.empty(self.chunk_rows, self.p, device=self.device, dtype=torch.float32)
        if self.free[slot] is not None:
            self.side.wait_event(self.free[slot])                  # the fit has finished reading the old chunk
        else:
            self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self._produce(c, slot)
            self.ready[slot].record(self.side)

    def release(self, c):
        """the main stream has enqueued its last read of chunk c"""
        if self.cuda and self.held is None:
            ev = self.torch.cuda.Event()
            ev.record(self.torch.cuda.current_stream(self.device))
            self.free[c % 2] = ev


def step_flops(k, p, b, s, sweeps, ride=False):
    """Algorithmic flops of one minibatch (SURVEY.md §8d work model, blocked dictionary update).
    ride: the single-GPU step with a proper feature subset - the statistics section only carries the head
    (C increment + sampled rows of the B increment); the p x k product rides along the dictionary update's launches
    (`dict_update` = `bcd` + `b_inc`, reported separately as `dict_update_own` / `dict_update_rider`)."""
    dx = 2.0 * b * s * k
    gram = 2.0 * k * k * s
    h0 = 2.0 * b * k * k
    cd = 4.0 * k * k * sweeps * b                  # two axpy(k) per coordinate, upper bound
    c_inc = 2.0 * k * k * b
    b_inc = 2.0 * b * k * p
    bcd = 2.0 * k * k * s
    if ride:
        # the sampled rows of the B increment (2 b k s) are in `stats_gemm`; the rider carries the other p - s rows
        rider = 2.0 * b * k * (p - s)
        return dict(code_gemm=dx + gram, code_solve=h0 + cd, stats_gemm=c_inc + 2.0 * b * k * s,
                    stats_apply=3.0 * (k * k + p * k), dict_update=bcd + rider, dict_update_own=bcd,
                    dict_update_rider=rider)
    return dict(code_gemm=dx + gram, code_solve=h0 + cd, stats_gemm=c_inc + b_inc, stats_apply=3.0 * (k * k + p * k),
                dict_update=bcd, dict_update_own=bcd, dict_update_rider=0.0)


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


class Run:
    """One estimator fitted on the fresh stream: `fit(n)` consumes the next n minibatches."""

    def __init__(self, args, reduction, rank, world, device, total_steps):
        import torch
        from modl_amd import DictFact
        self.torch, self.world, self.device = torch, world, device
        self.stream = M1Stream(P_FEAT, 1234, device, rank=rank)
        self.total_rows = total_steps * BATCH
        self.resident = self.stream.resident(self.total_rows)
        # the dictionary is initialised from the same rows on every rank (replicas stay identical: same init, same
        # draws, all-reduced statistics): the first 256 rows of rank 0's stream
        X0 = M1Stream(P_FEAT, 1234, device, rank=0).rows(0, K_COMP) if rank != 0 else None
        self.est = DictFact(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
                            comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
        first = self.stream.chunk(0, prefetch_next=self.total_rows > CHUNK)
        self.est.prepare(n_samples=CHUNK, X=first[:K_COMP] if X0 is None else X0)
        # N > 1: the head is summed by the library's own RCCL communicator, one library call per chunk of minibatches
        # (default whenever the process group runs RCCL); --torch-collective: dist.all_reduce between two calls per
        # minibatch (also what a gloo group - ranks sharing a GPU in the tests - uses)
        torch_route = bool(getattr(args, 'torch_collective', False)) or getattr(args, 'backend', 'nccl') != 'nccl'
        self.est._native_rccl = not torch_route
        if getattr(args, 'force_reduce', False):                 # testing only: the N > 1 step with one rank
            self.est._force_reduce = True
            self.est._two_phase = torch_route
        self.be = self.est._backend
        self.row = 0                                              # next unseen row of the stream
        self.enqueue_s = 0.0

    def fit(self, nsteps):
        """fits the next nsteps minibatches; returns without synchronising"""
        done = 0
        while done < nsteps:
            c, r0 = divmod(self.row, CHUNK)
            todo = min(nsteps - done, (CHUNK - r0) // BATCH)
            more = self.total_rows > (c + 1) * CHUNK
            Xc = self.stream.chunk(c, prefetch_next=more)
            t0 = time.perf_counter()
            # sample_indices = chunk-local positions: what partial_fit(chunk) with sample_indices=None uses
            self.est.partial_fit(Xc[r0:r0 + todo * BATCH], np.arange(r0, r0 + todo * BATCH), _sync=False)
            self.enqueue_s += time.perf_counter() - t0
            self.row += todo * BATCH
            done += todo
            if r0 + todo * BATCH == CHUNK:
                self.stream.release(c)

    def sync(self):
        # the estimator's own wait first (DictFact.partial_fit's: it polls the stream and checks the status word of the
        # persistent launches), then the device-wide synchronisation the timing contract asks for
        self.be.synchronize()
        self.torch.cuda.synchronize(self.device)


def timed(run, steps, world):
    """barrier + synchronize, exactly `steps` minibatches, synchronize + barrier; max over ranks"""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    run.sync()
    run.enqueue_s = 0.0
    run.be.host_wait_ms()
    t0 = time.perf_counter()
    run.fit(steps)
    enq = run.enqueue_s - run.be.host_wait_ms() * 1e-3     # net of the time the host waited for the device (8-deep ring)
    run.sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=run.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, enq


ROOF_SECTIONS = ('dict_update', 'code_solve')      # the two sections that can dominate the step: both are timed


def run_gpu(args, reduction, steps, warmup, rank, world, device, breakdown=True):
    """Warm-up, the timed region (HIP events around `dict_update` and `code_solve` only, on one minibatch in four:
    every timed section costs two event records, i.e. a stream bubble of a few microseconds each), then an untimed
    pass with all sections timed for the breakdown.  The roofline line is the section with the larger time OVER THE
    TIMED REGION (not over a warm-up minibatch: the line means the same thing whatever --warmup is)."""
    import torch
    import torch.distributed as dist
    extra = min(steps, 50) if breakdown else 0
    run = Run(args, reduction, rank, world, device, steps + warmup + extra)
    be = run.be
    run.fit(warmup)
    run.sync()
    dom = 'dict_update'
    if breakdown:
        # events around two sections, on one minibatch in four: an event pair is a stream bubble of ~9 us
        # (scripts/step_timeline.py on a rocprofv3 trace), i.e. ~2 % of a step this way
        # (another shape than the metric's: all four arithmetic sections are candidates - at C5 the wide statistics
        #  product is the largest; an event pair is nothing against its ~1 ms steps)
        roof_sections = ROOF_SECTIONS if P_FEAT == 10000 else ('dict_update', 'code_solve', 'stats_gemm', 'code_gemm')
        be.prof_enable(True, sections=list(roof_sections), every=4 if steps >= 16 else 1)
        be.prof_reset()
    dt, enq = timed(run, steps, world)
    res = dict(dt=dt, enqueue_ms_per_step=enq / steps * 1e3, dom=dom, prof_dom=None, prof_timed={}, prof={})
    if breakdown:
        got = be.prof_get()
        res['prof_timed'] = {n: got[n] for n in roof_sections if n in got and got[n]['calls'] > 0}
        if res['prof_timed']:
            dom = max(res['prof_timed'], key=lambda n: res['prof_timed'][n]['ms'] / res['prof_timed'][n]['calls'])
        res['dom'] = dom
        res['prof_dom'] = res['prof_timed'].get(dom)
        be.prof_enable(False)
    lsw = be.last_sweeps()
    res['sweeps'], res['sweeps_max'] = float(lsw.mean()), int(lsw.max())
    if extra:
        be.prof_enable(True)
        be.prof_reset()
        run.fit(extra)
        run.sync()
        res['prof'] = be.prof_get()
        be.prof_enable(False)
    res['finite'] = bool(torch.isfinite(be.Dt).all().item())
    res['replicas_identical'] = None
    if world > 1:                                  # the replicas of the dictionary must agree bit for bit
        chk = torch.stack([be.Dt.double().sum(), (be.Dt.double() ** 2).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res['replicas_identical'] = bool(torch.equal(lo, hi))
    # persistent dictionary-update launches that could not run and were completed on the device by one workgroup (DESIGN 3.3:
    # another process holding compute units; 0 on a GPU of one's own), summed over the ranks
    rec = int(getattr(be, 'persist_recoveries', 0))
    if world > 1:
        t = torch.tensor([rec], dtype=torch.int64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t)
        rec = int(t.item())
    res['persist_recoveries'] = rec
    res['rows_generated'] = run.stream.generated_rows
    res['rows_fitted'] = run.row
    res['collective'] = collective_route(run, world, args)
    return res


def collective_route(run, world, args):
    """which of the two exchange routes of DESIGN §7 actually ran (None with one rank and no forced reduction)"""
    if world == 1 and not getattr(args, 'force_reduce', False):
        return None
    if getattr(run.be, 'comm', None):
        return 'native RCCL: modl_somf_partial_fit_chunk + the library\'s own communicator (one call per chunk of minibatches)'
    why = ''
    if getattr(run.be, '_comm_failed', False):
        why = ' (FALLBACK: the library\'s communicator could not be created)'
    return 'torch.distributed all_reduce (%s) between the two phases, two library calls per minibatch%s' % (args.backend, why)


def steady_state(args, reduction, rank, world, device, steps, burn_in):
    """>= 2000 fresh-row minibatches after a burn-in, no events on the stream."""
    import torch
    run = Run(args, reduction, rank, world, device, steps + burn_in + 48)
    run.fit(burn_in)
    dt, enq = timed(run, steps, world)
    lsw = run.be.last_sweeps()
    be = run.be                                               # section breakdown: 48 more minibatches, outside the timed region
    be.prof_enable(True)
    be.prof_reset()
    run.fit(48)
    run.sync()
    prof = be.prof_get()
    be.prof_enable(False)
    sections = {n: dict(ms_per_step=e['ms'] / e['calls'], launches_per_step=e['launches'] / e['calls'])
                for n, e in prof.items() if e['calls']}
    return dict(reduction=reduction, sections=sections, steps=steps, burn_in=burn_in, value=steps * BATCH * world / dt, unit='samples/s',
                ms_per_step=dt / steps * 1e3, host_enqueue_ms_per_step=enq / steps * 1e3,
                cd_sweeps_mean=float(lsw.mean()), cd_sweeps_max=int(lsw.max()),
                finite=bool(torch.isfinite(run.be.Dt).all().item()), rows_fitted=run.row,
                rows='fresh (no row fitted twice; chunk-local warm starts from the previous chunk)')


def cpu_baseline(X, reduction, budget_s=20.0, threads=None):
    """CPU oracle on a bounded prefix of the same rows (X: host array, the first rows of rank 0's stream)."""
    from oracle import somf_oracle as orc
    cores = os.cpu_count() or 1
    threads = threads or min(cores, 32)             # 256-wide GEMMs do not scale past a few dozen threads
    from contextlib import nullcontext
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads, user_api='blas')
    except Exception:                               # pragma: no cover
        limiter, threads = nullcontext(), cores
    n_rows = X.shape[0]
    pr = orc.SomfParams(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
                        comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0,
                        n_threads=threads)          # (the reference's own thread pool over the samples of a minibatch, dict_fact.py:584-621)
    with limiter:
        st = orc.prepare(pr, n_samples=n_rows, X=X[:K_COMP])
        st.sweeps = []                                  # (one int32 per sample and minibatch: for the parity block)
        done, t0 = 0, time.perf_counter()
        for r0 in range(0, n_rows, BATCH):
            orc.partial_fit(st, pr, X[r0:r0 + BATCH], np.arange(r0, r0 + BATCH))
            done += BATCH
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    out = dict(value=done / dt, unit='samples/s', cores=threads, kind='port',
               sample='first %d rows of stream M1 (p=%d, k=%d, b=%d, reduction=%g), %.1f s; numpy/OpenBLAS with %d threads '
                      '(of %d host cores) for the contractions, the compiled CD solver on the reference\'s own thread pool over the '
                      'samples of a minibatch (n_threads=%d, dict_fact.py:584-621; round 6 - it ran on one thread before), '
                      'single-thread C for the projections'
                      % (done, P_FEAT, K_COMP, BATCH, reduction, dt, threads, cores, threads))
    try:                                            # speed of this port relative to the real reference (build container)
        cal = json.load(open(os.path.join(ROOT, 'profiles', CALIBRATION_FILE)))
        out['calibration'] = cal.get('summary', cal)
        out['calibration_note'] = 'measured in the build container with n_threads = 1 on both sides (round 2)'
    except (OSError, ValueError):
        out['calibration'] = None
    return out, st, done


def parity_block(X, done, st32, reduction, device):
    """The timed code path against the CPU oracle, outside the timed region: ONE `partial_fit` call on the rows the
    cpu_baseline leg has just fitted with the oracle (same estimator parameters and seeds), i.e. one
    modl_somf_partial_fit_chunk call of `done / 256` minibatches - staging ring wrap-around, both device parameter
    blocks, the staging copy and the statistics product riding on the dictionary update's launches.
    The yardstick is the reference algorithm's OWN f32 noise: the oracle is run once more in f64 on the same float32
    rows (not timed); `rel_fro_*` = GPU f32 against that f64 run, `oracle_f32_noise_*` = the oracle's f32 run (the
    cpu_baseline leg) against it.  `sweep_flips` = samples, over all minibatches, whose coordinate descent did another
    number of sweeps than in the f64 run (a tolerance-stopped solver flips on samples whose duality gap sits on the
    threshold; `oracle_f32_sweep_flips` is the same count for the oracle's f32 run)."""
    import torch
    from modl_amd import DictFact
    from oracle import somf_oracle as orc
    kw = dict(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
              comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    nb = done // BATCH
    est = DictFact(**kw)
    est.prepare(n_samples=X.shape[0], X=X[:K_COMP])
    hist = est._backend.sweeps_history(nb)
    Xd = torch.from_numpy(X[:done]).to(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    est.partial_fit(Xd, np.arange(done))
    dt = time.perf_counter() - t0
    sw_gpu = hist()
    X64 = X[:done].astype(np.float64)
    pr = orc.SomfParams(n_threads=min(os.cpu_count() or 1, 32), **kw)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=min(os.cpu_count() or 1, 32), user_api='blas')
    except Exception:                               # pragma: no cover
        from contextlib import nullcontext
        limiter = nullcontext()
    with limiter:
        st64 = orc.prepare(pr, n_samples=X.shape[0], X=X[:K_COMP].astype(np.float64))
        st64.sweeps = []
        t_flip, before = None, None                 # first minibatch in which a sample does another number of sweeps than on the GPU
        for t, r0 in enumerate(range(0, done, BATCH)):
            if t_flip is None:
                prev = (st64.D.copy(), st64.C.copy())            # the f64 state after t minibatches
            orc.partial_fit(st64, pr, X64[r0:r0 + BATCH], np.arange(r0, r0 + BATCH))
            if t_flip is None and (np.asarray(st64.sweeps[-1]) != sw_gpu[t, :BATCH]).any():
                t_flip, before = t, prev
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                             max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))
    sw64, sw32 = np.stack(st64.sweeps), np.stack(st32.sweeps)
    got = dict(D=est.components_, C=est.C_, code=est.code_[:done])
    ref32 = dict(D=st32.D, C=st32.C, code=st32.code[:done])
    ref64 = dict(D=st64.D, C=st64.C, code=st64.code[:done])
    out = dict(steps=nb, rows=done, path='DictFact.partial_fit -> modl_somf_partial_fit_chunk (one call)',
               reference='oracle/somf_oracle.py in f64 on the same float32 rows; the f32 noise is the cpu_baseline leg\'s run against it')
    ok = flat = behind = True
    for key in ('D', 'C', 'code'):
        e, noise = rel(got[key], ref64[key]), rel(ref32[key], ref64[key])
        out['rel_fro_' + key] = e
        out['oracle_f32_noise_' + key] = noise
        out['gpu_vs_oracle_f32_' + key] = rel(got[key], ref32[key])
        ok = ok and e <= 2 * noise + 1e-5
        flat = flat and e <= 1e-5
        behind = behind and e <= 2 * noise + 1e-5 + 6e-5
    # `within_1e5`: the north star's bound, flat, on dictionary / C / codes (what the GPU tests assert while no sample has
    # flipped its sweep count); the noise-relative flag is the rule behind a flip
    flips = int((sw_gpu[:, :BATCH] != sw64).sum())
    # `within_tolerance`: the gate of tests/test_gpu_step.py::test_timed_path_long_horizon_vs_oracle - 1e-5 flat while no sample has
    # flipped; behind a flip (a tolerance-stopped solver doing one sweep more or less on a sample whose duality gap sits on the
    # threshold - the oracle's own f32 run does it too, `oracle_f32_sweep_flips`; at most 0.1 % of the samples) 2 x the
    # oracle's f32 noise + 1e-5 + 6e-5
    if t_flip is not None and t_flip > 0:
        # the flat bound where it applies: the state after the last minibatch BEFORE the first flip (a fresh estimator fitted on
        # those minibatches - the same draws, the same bits as the first t_flip steps of the call above)
        e2 = DictFact(**kw)
        e2.prepare(n_samples=X.shape[0], X=X[:K_COMP])
        e2.partial_fit(Xd[:t_flip * BATCH], np.arange(t_flip * BATCH))
        bf = dict(steps=t_flip, rel_fro_D=rel(e2.components_, before[0]), rel_fro_C=rel(e2.C_, before[1]),
                  rel_fro_code=rel(e2.code_[:t_flip * BATCH], st64.code[:t_flip * BATCH]))
        bf['within_1e5'] = bool(max(bf['rel_fro_D'], bf['rel_fro_C'], bf['rel_fro_code']) <= 1e-5)
        out['before_first_flip'] = bf
    out.update(within_1e5=bool(flat) if flips == 0 else None,      # (None: a sample flipped its sweep count - the rule behind a flip applies)
               within_2x_reference_f32_noise_plus_1e5=bool(ok),
               within_tolerance=bool(flat) if flips == 0 else bool(behind and flips <= 1e-3 * sw64.size), sweep_flips=flips,
               oracle_f32_sweep_flips=int((sw32 != sw64).sum()), samples=int(sw64.size),
               sweeps_agree=float(np.mean(sw_gpu[:, :BATCH] == sw64)), n_iter_equal=bool(est.n_iter_ == st64.n_iter),
               gpu_ms_per_step=dt / max(nb, 1) * 1e3)
    return out


def flip_rate_block(reduction, n_minibatches, device, threads=None, seed=4321, log=None):
    """How often does a sample do another number of coordinate-descent sweeps than in the reference algorithm's f64 run - on the
    GPU (f32) and in the reference algorithm's OWN f32 run?  (the parity gates allow for such flips -
    a tolerance-stopped solver on a sample whose duality gap sits on the threshold - and this measures whether the GPU path
    produces more of them than the CPU path does.)  Three free-running fits of the same fresh rows of stream M1 (a stream of its
    own seed), same parameters and draws: the oracle in f64 (the yardstick), the oracle in f32, the GPU estimator in f32 (ONE
    partial_fit call: the code path the bench times); every sample of every minibatch is compared by its sweep count
    (modl_somf_sweeps_history on the GPU).  Counted twice: over the whole run, and up to each run's own first flip (behind a flip
    a trajectory is 2e-5 away from the f64 one and flips more easily - in both runs alike).  `rel_fro_*_flip_free` = each f32
    run against the f64 run after its last flip-free minibatch."""
    import torch
    from modl_amd import DictFact
    from oracle import somf_oracle as orc
    threads = threads or min(os.cpu_count() or 1, 32)
    kw = dict(n_components=K_COMP, batch_size=BATCH, reduction=reduction, code_alpha=1.0, code_l1_ratio=1,
              comp_l1_ratio=0, learning_rate=0.92, G_agg='masked', Dx_agg='masked', random_state=0)
    n = n_minibatches * BATCH
    stream = M1Stream(P_FEAT, seed, device, rank=0)
    Xd = stream.rows(0, n)
    est = DictFact(**kw)
    est.prepare(n_samples=n, X=Xd[:K_COMP])
    hist = est._backend.sweeps_history(n_minibatches)
    est.partial_fit(Xd, np.arange(n))
    sw_gpu = hist()[:, :BATCH]
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                             max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads, user_api='blas')
    except Exception:                               # pragma: no cover
        from contextlib import nullcontext
        limiter = nullcontext()
    X0 = Xd[:K_COMP].cpu().numpy()
    pr = orc.SomfParams(n_threads=threads, **kw)
    t0 = time.perf_counter()
    with limiter:
        st64 = orc.prepare(pr, n_samples=n, X=X0.astype(np.float64))
        st32 = orc.prepare(pr, n_samples=n, X=X0)
        st64.sweeps, st32.sweeps = [], []
        first = dict(gpu=None, o32=None)            # first minibatch with a flip, per run
        free = dict(gpu=None, o32=None)             # the f64 state (D, C) after the last flip-free minibatch of each run, + that run's own
        CH = 8192
        for c0 in range(0, n, CH):
            Xc = Xd[c0:c0 + CH].cpu().numpy()
            Xc64 = Xc.astype(np.float64)
            for r0 in range(0, Xc.shape[0], BATCH):
                t = (c0 + r0) // BATCH
                idx = np.arange(c0 + r0, c0 + r0 + BATCH)
                prev64 = (st64.D.copy(), st64.C.copy()) if (first['gpu'] is None or first['o32'] is None) else None
                prev32 = (st32.D.copy(), st32.C.copy()) if first['o32'] is None else None
                orc.partial_fit(st64, pr, Xc64[r0:r0 + BATCH], idx)
                orc.partial_fit(st32, pr, Xc[r0:r0 + BATCH], idx)
                if first['gpu'] is None and (st64.sweeps[-1] != sw_gpu[t]).any():
                    first['gpu'], free['gpu'] = t, prev64
                if first['o32'] is None and (st32.sweeps[-1] != st64.sweeps[-1]).any():
                    first['o32'], free['o32'] = t, (prev64, prev32)
            if log:
                log('flip_rate r=%g: %d / %d minibatches, %.0f s' % (reduction, min(n, c0 + CH) // BATCH, n_minibatches, time.perf_counter() - t0))
    sw64, sw32 = np.stack(st64.sweeps), np.stack(st32.sweeps)
    f_gpu, f_o32 = sw_gpu != sw64, sw32 != sw64
    out = dict(reduction=reduction, samples=int(sw64.size), minibatches=n_minibatches,
               gpu_f32_flips=int(f_gpu.sum()), oracle_f32_flips=int(f_o32.sum()),
               gpu_f32_first_flip_minibatch=first['gpu'], oracle_f32_first_flip_minibatch=first['o32'],
               gpu_f32_minibatches_with_a_flip=int(f_gpu.any(axis=1).sum()), oracle_f32_minibatches_with_a_flip=int(f_o32.any(axis=1).sum()),
               sweeps_mean_f64=float(sw64.mean()), oracle_threads=threads, oracle_s=time.perf_counter() - t0,
               protocol='free-running fits of the same fresh rows: oracle f64 (yardstick), oracle f32, GPU f32 (one partial_fit call); a flip = a '
                        'sample whose sweep count differs from the f64 run\'s')
    out['gpu_within_2x_oracle_plus_2'] = bool(out['gpu_f32_flips'] <= 2 * out['oracle_f32_flips'] + 2)
    # each f32 run against the f64 run after its last flip-free minibatch (the whole run when it never flips)
    if first['o32'] is None:
        out['oracle_f32_rel_fro_flip_free'] = dict(minibatches=n_minibatches, D=rel(st32.D, st64.D), C=rel(st32.C, st64.C))
    elif first['o32'] > 0:
        (D64, C64), (D32, C32) = free['o32']
        out['oracle_f32_rel_fro_flip_free'] = dict(minibatches=first['o32'], D=rel(D32, D64), C=rel(C32, C64))
    if first['gpu'] is None:
        out['gpu_f32_rel_fro_flip_free'] = dict(minibatches=n_minibatches, D=rel(est.components_, st64.D), C=rel(est.C_, st64.C))
    elif first['gpu'] > 0:
        e2 = DictFact(**kw)
        e2.prepare(n_samples=n, X=Xd[:K_COMP])
        e2.partial_fit(Xd[:first['gpu'] * BATCH], np.arange(first['gpu'] * BATCH))
        out['gpu_f32_rel_fro_flip_free'] = dict(minibatches=first['gpu'], D=rel(e2.components_, free['gpu'][0]), C=rel(e2.C_, free['gpu'][1]))
    out['rel_fro_end'] = dict(gpu_f32_D=rel(est.components_, st64.D), oracle_f32_D=rel(st32.D, st64.D),
                              gpu_f32_C=rel(est.C_, st64.C), oracle_f32_C=rel(st32.C, st64.C))
    return out


def survey_flops_per_sample(k, p, b, s, sweeps):
    """2ks (Dx) + 2kp (B_) + 2k^2 (C_) + 4k^2 n_sw (CD) + (2k^2 s [G] + 2k^2 s [C_ D_sub] + 4k^2 s [2k gers]) / b"""
    return 2.0 * k * s + 2.0 * k * p + 2.0 * k * k + 4.0 * k * k * sweeps + 8.0 * k * k * s / b


def roofline_of(dom, prof_dom, fl, by, reduction):
    if not prof_dom or prof_dom['calls'] <= 0:
        return None
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
    if dom == 'dict_update':
        # the launches of the dictionary update also carry the p x k statistics product of the rows that were not
        # sampled (it rides on the compute units the update leaves idle): the two shares of `achieved`, separately
        own, rider = fl['dict_update_own'], fl['dict_update_rider']
        roof['split'] = dict(block_coordinate_update=dict(flops_per_step=own, achieved=own / ms / 1e9,
                                                          frac=own / ms / 1e9 / PEAK_MFMA_F32_TFLOPS),
                             riding_statistics_product=dict(flops_per_step=rider, achieved=rider / ms / 1e9,
                                                            frac=rider / ms / 1e9 / PEAK_MFMA_F32_TFLOPS))
    if dom == 'code_solve':
        roof['note'] = ('coordinate descent is a VALU kernel (no MFMA): `peak` is the f32 VECTOR rate, which equals the f32 '
                        'matrix rate on this part (157.3 TFLOP/s); the kernel is bound by the dependency chain of k x sweeps '
                        'sequential coordinate steps per sample, not by either (informational)')
    # HBM traffic per launch of the section's main kernel: from the committed rocprofv3 --pmc passes of this round
    # (FETCH_SIZE and WRITE_SIZE need separate profiler passes and cannot be collected from inside this process)
    try:
        PMC_FILE = [f for f in PMC_FILES if os.path.exists(os.path.join(ROOT, 'profiles', f))][0]
        pmc = json.load(open(os.path.join(ROOT, 'profiles', PMC_FILE)))
        names = DOM_KERNEL.get(dom, '?')
        names = names if isinstance(names, tuple) else (names,)
        kern = [k_ for n_ in names for k_ in pmc if k_.startswith(n_)]          # (the first name that the passes hold)
        if kern and abs(reduction - 10.0) < 1e-9:
            e = pmc[kern[0]]
            roof['traffic'] = e.get('fetch_bytes_corrected', 0.0) + e.get('write_bytes', 0.0)
            roof['traffic_source'] = ('OFFLINE: profiles/%s (rocprofv3 --pmc passes of bench.py at reduction=10): %s, '
                                      'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per launch' % (PMC_FILE, kern[0]))
    except (OSError, ValueError, IndexError):
        pass
    return roof


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as child processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set),
    relay rank 0's stdout - whose last line is the JSON line - and return the worst exit status.  If a rank dies the
    others are ended (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as s:                     # a free rendezvous port
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    lines = []

    def relay():
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors='replace'))
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    status, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                status = status or rc
                for o in alive:                    # exactly the processes started above, by pid
                    procs[o].terminate()
        time.sleep(0.05)
    t.join(timeout=10)
    sys.stdout.write(''.join(lines))
    sys.stdout.flush()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed minibatches (default 2000 at the metric\'s shape; 300 at another --features: with the 100 warm-up and the 50 '
                         'breakdown minibatches the stream of p = 200 000 stays resident in HBM, 92 GB - beyond 96 GB it becomes a '
                         'two-chunk ring whose generator runs next to the timed steps)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up minibatches (default 500; 100 at another --features)')
    ap.add_argument('--reduction', type=float, default=10.0)
    ap.add_argument('--features', type=int, default=10000,
                    help='p of the synthetic stream (10000: the metric\'s workload M1; 200000 with --reduction 12: the per-GPU '
                         'shape of BASELINE config 5, the HCP-shaped stream - same line format, config.workload names it)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true', help='no HIP events at all on the stream (for kernel-trace timelines)')
    ap.add_argument('--steady-steps', type=int, default=None,
                    help='fresh-row steps of each steady_state record (0: skip; default 2000 at the metric\'s shape, 0 otherwise)')
    ap.add_argument('--steady-burn-in', type=int, default=500)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to test the N > 1 path)')
    ap.add_argument('--force-reduce', action='store_true',
                    help='testing only: run the multi-GPU step (two phases + RCCL all-reduces) even with one rank')
    ap.add_argument('--native-rccl', action='store_true',
                    help='(the default with --backend nccl; kept for older command lines) N > 1: the all-reduce issued by '
                         'the library itself (RCCL on the compute stream, one library call per chunk of minibatches)')
    ap.add_argument('--torch-collective', action='store_true',
                    help='N > 1: dist.all_reduce of the head between two library calls per minibatch instead of the '
                         'library\'s own RCCL communicator')
    ap.add_argument('--flip-minibatches', default='200,50', metavar='N10,N1',
                    help='minibatches of the sweep-flip census at reduction 10 and 1 (parity.flip_rate; one GPU, with the CPU baseline; '
                         '0,0: off).  Default: 51 200 and 12 800 samples, ~1.5 min of CPU oracle on the box; '
                         'tests/test_gpu_step.py::test_sweep_flip_rate runs 400,200 and profiles/r06_flip_census.json 800,800')
    ap.add_argument('--share-gpu', action='store_true', help='testing only: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--debug-set', action='append', default=[], metavar='WHAT=VALUE',
                    help='diagnostics: modl_debug_set(WHAT, VALUE) before anything runs (A/B runs of scripts/; see include/modl_hip.h)')
    args = ap.parse_args()
    global P_FEAT, CHUNK, BLOCK
    if args.features != P_FEAT:
        # another feature count (C5: p = 200 000): chunks of at most ~8 GB (a power of two of rows, whole minibatches),
        # produced in eight blocks each
        P_FEAT = int(args.features)
        CHUNK = max(8 * BATCH, min(65536, 1 << int(np.log2(8e9 / (4.0 * P_FEAT)))))
        BLOCK = CHUNK // 8
    if args.steady_steps is None:
        args.steady_steps = 2000 if P_FEAT == 10000 else 0
    if args.steps is None:
        args.steps = 2000 if P_FEAT == 10000 else 300
    if args.warmup is None:
        args.warmup = 500 if P_FEAT == 10000 else 100

    if args.gpus > 1 and ('WORLD_SIZE' not in os.environ or (
            os.environ['WORLD_SIZE'] == '1' and 'TORCHELASTIC_RUN_ID' not in os.environ and 'RANK' not in os.environ)):
        # launched plainly (`python bench.py --gpus N`): this process becomes the launcher.  It has not touched the
        # GPU (no torch import, no library load) and never does: the ranks are CHILD processes, never an exec.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:                         # a launcher decided otherwise: its world size is what runs
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)\n' % (args.gpus, world, world))
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

    for item in args.debug_set:
        from modl_amd._lib import lib as _l, check as _c
        what, value = item.split('=')
        _c(_l.modl_debug_set(int(what), int(value)), 'modl_debug_set')
    res = run_gpu(args, args.reduction, args.steps, args.warmup, rank, world, device, breakdown=not args.no_breakdown)
    steady = []
    if args.steady_steps > 0:
        for r in ((10.0, 1.0) if P_FEAT == 10000 else (args.reduction,)):
            steady.append(steady_state(args, r, rank, world, device, args.steady_steps, args.steady_burn_in))
    out = None
    if rank == 0:
        dt, sweeps, dom, prof = res['dt'], res['sweeps'], res['dom'], res['prof']
        samples = args.steps * BATCH * world
        s_mean = P_FEAT / args.reduction
        # (the riding statistics product is active in the two-phase step too; not beyond 65 536 features, where the
        #  p x k product runs as its own launch: somf_step.hip)
        ride = args.reduction > 1 and -(-P_FEAT // 64) < 1024
        fl = step_flops(K_COMP, P_FEAT, BATCH, s_mean, sweeps, ride=ride)
        by = step_bytes(K_COMP, P_FEAT, BATCH, s_mean, ride=ride)
        sections = {}
        for name, e in prof.items():
            if e['calls'] == 0:
                continue
            ms = e['ms'] / e['calls']
            sections[name] = dict(ms_per_step=ms, launches_per_step=e['launches'] / e['calls'],
                                  gflops=fl[name] / ms / 1e6, gbs=by[name] / ms / 1e6)
        roof = roofline_of(dom, res['prof_dom'], fl, by, args.reduction)
        total_fl = sum(fl[n] for n in ('code_gemm', 'code_solve', 'stats_gemm', 'dict_update'))
        if roof is not None:
            # the other candidate, measured over the same timed region, and the step as a whole by SURVEY §8d's model
            other = sorted((n for n in res['prof_timed'] if n != dom),
                           key=lambda n: -res['prof_timed'][n]['ms'] / res['prof_timed'][n]['calls'])
            roof['other_section'] = roofline_of(other[0], res['prof_timed'][other[0]], fl, by, args.reduction) if other else None
            fps = survey_flops_per_sample(K_COMP, P_FEAT, BATCH, s_mean, sweeps)
            tf = samples / dt * fps / 1e12
            roof['whole_step'] = dict(flops_per_sample=fps, achieved=tf, unit='TFLOP/s', frac=tf / PEAK_MFMA_F32_TFLOPS / world,
                                      note='SURVEY 8(d) work model of the REFERENCE algorithm at the measured sweep count '
                                           '(it counts the 4k^2 s of ger updates the blocked dictionary update does not perform)')
        m1 = P_FEAT == 10000
        shape_name = ('M1 stream (SURVEY 8d)' if m1 else
                      'C5 per-GPU shape (BASELINE config 5: HCP-shaped synthetic stream, p = 200 000 features, n_components = 256, '
                      'minibatches sharded over the ranks)' if P_FEAT == 200000 else 'M1-like stream with p = %d' % P_FEAT)
        out = dict(metric='samples/sec through DictFact.partial_fit at k=256, p=%s' % ('10k' if m1 else '%dk' % (P_FEAT // 1000)),
                   value=samples / dt,
                   unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload=shape_name + ': fresh rows only, produced on the device in %d-row chunks by a '
                                        'counter-based generator BEFORE the timed region (resident in HBM; beyond 96 GB a '
                                        'two-chunk ring fed from a side stream), no row fitted twice, chunk-local warm '
                                        'starts, p=%d f32, '
                                        'n_components=%d, batch_size=%d/GPU, reduction=%g, code_alpha=1 (l1 codes), l2 atoms, '
                                        'learning_rate=0.92, masked/masked; timed right after the warm-up steps'
                                        % (CHUNK, P_FEAT, K_COMP, BATCH, args.reduction),
                               reduction=args.reduction, global_batch=BATCH * world,
                               parallelism='dp%d (row-sharded minibatch; all-reduce of the C increment and of the sampled rows '
                                           'of the B increment before each dictionary update; every rank keeps its own partial '
                                           'B_ for the rows that were not sampled)' % world,
                               collective=res['collective']),
                   roofline=roof, sections=sections, cd_sweeps_mean=sweeps, cd_sweeps_max=res['sweeps_max'],
                   step_tflops=total_fl / (dt / args.steps) / 1e12, finite=res['finite'],
                   replicas_identical=res['replicas_identical'], persist_recoveries=res['persist_recoveries'],
                   host_enqueue_ms_per_step=res['enqueue_ms_per_step'],
                   rows=dict(fitted=res['rows_fitted'], generated=res['rows_generated']),
                   steady_state=steady)
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = max(2, min(40, int(2e9 // (4 * P_FEAT * BATCH)))) * BATCH     # (host copy of the prefix <= 2 GB)
            Xh = M1Stream(P_FEAT, 1234, device, rank=0).rows(0, n_cpu).cpu().numpy()
            out['cpu_baseline'], st_ref, done = cpu_baseline(Xh, args.reduction)
            out['parity'] = parity_block(Xh, done, st_ref, args.reduction, device)
            if m1:
                nf = [int(v) for v in args.flip_minibatches.split(',')]
                out['parity']['flip_rate'] = [flip_rate_block(r, nmb, device) for r, nmb in zip((10.0, 1.0), nf) if nmb > 0]
            for rec in steady:
                # the parity block of the other steady-state leg as well (r = 1: one extra call of 16 minibatches and
                # two oracle runs on the same rows, outside every timed region)
                if abs(rec['reduction'] - args.reduction) > 1e-9:
                    _, st_r, done_r = cpu_baseline(Xh[:16 * BATCH], rec['reduction'], budget_s=30.0)
                    rec['parity'] = parity_block(Xh, done_r, st_r, rec['reduction'], device)
        else:
            out['cpu_baseline'] = None
            out['parity'] = None
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
