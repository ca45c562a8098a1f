// The fused, device-resident SOMF minibatch step and the fine-grained solver
// entry points of the C-ABI (include/modl_hip.h).
//
// Replaces DictFact._single_batch_fit and what it calls
//   (reference: modl/decomposition/dict_fact.py:495-533 _single_batch_fit,
//    :577-648 _compute_code, :559-575 _update_C/_update_B, :650-715 _update_dict,
//    :47-92 transform).
//
// Data layout in HBM (T = f32 or f64):
//   Dt [p][k]  dictionary, feature-major (components_.T): a sampled feature is one
//              contiguous k-vector, so the subset is read by row index, never copied;
//   Bt [p][k]  surrogate statistic B_.T, same layout;
//   C  [k][k]  surrogate statistic C_ (bitwise symmetric by construction);
//   code [n][k], G_ [k][k], Dx_average_ [n][k], G_average_ [n][k][k] as in the reference.
// With several GPUs a step is two phases: every rank keeps its own PARTIAL statistics
// (C_ = sum_r C_r, B_ = sum_r B_r: both recursions are linear in the increments), phase 1
// updates them from the rank's rows and writes the head [C_r | rows of B_r of the sampled
// features]; the caller sums the head over the ranks (RCCL) and phase 2 runs the identical
// dictionary update on every rank from the summed head.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <sched.h>
#include <new>
#include <vector>

#include "gemm_dense.hpp"
#include "gemm_wide.hpp"
#include "gemm_resident.hpp"
#include "kernels.hpp"

namespace modl {
std::atomic<int> g_stats_resident{1};      // modl_debug_set(MODL_DEBUG_STATS_RESIDENT, ...)

enum Section { SEC_CODE_GEMM = 0, SEC_CODE_SOLVE, SEC_STATS_GEMM, SEC_STATS_APPLY, SEC_DICT, SEC_COUNT };
static const char *kSectionNames[SEC_COUNT] = {"code_gemm", "code_solve", "stats_gemm", "stats_apply", "dict_update"};

constexpr int kStageSlots = 8;
constexpr int kProfPool = 2048;      // event pairs kept before a flush
#ifndef MODL_STAT_BM
#define MODL_STAT_BM 64
#define MODL_STAT_BN 64
#endif
#ifndef MODL_STAT_BK
#define MODL_STAT_BK 0
#endif
constexpr int kStatBM = MODL_STAT_BM, kStatBN = MODL_STAT_BN, kStatBK = MODL_STAT_BK;   // block tile of the p x k increment product

template <typename T> struct EpiDxAverage {   // dict_fact.py:596-601
    T *Dx; T *avg; const int64_t *idx; const T *w_sample; int64_t k; T alpha;
    __device__ __forceinline__ void operator()(int64_t i, int64_t j, T v) const {
        T *a = avg + (idx ? idx[i] : i) * k + j;
        const T ws = w_sample[i];
        T cur = *a * ((T)1 - ws);
        cur = cur + (alpha * v) * ws;
        *a = cur;
        Dx[i * k + j] = cur;
    }
};

// C <- (1 - w) C + (w / b) v ; the same for Bt (dict_fact.py:559-575) as a GEMM epilogue: the increments never
// travel through HBM.  `mirror` (two-phase step): the updated value is also written to the head buffer.
// (w v) / b: b is the global minibatch size, a power of two more often than not - then the product with its reciprocal is
// the same number (both are the correctly rounded value of the same real; 2^-e is representable), three instructions
// instead of the twelve of a division.  Uniform test, one reciprocal per thread.
template <typename T> __device__ __forceinline__ bool stats_pow2(T b) {
    if constexpr (sizeof(T) == 4) return (__float_as_uint(b) & 0x007fffffu) == 0 && b > (T)0 && b < (T)1e30;
    else return (__double_as_longlong(b) & 0x000fffffffffffffll) == 0 && b > (T)0 && b < (T)1e300;
}
template <typename T> __device__ __forceinline__ T stats_value(T v, T old, T beta, T wt, T bdiv, int replace, bool p2, T rinv) {
    const T x = replace ? v : wt * v;
    const T q = p2 ? x * rinv : x / bdiv;
    return replace ? q : old * beta + q;
}
template <typename T> struct EpiStats {
    static constexpr bool rmw = true;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    T *out; int64_t ld; T beta, wt, bdiv; int replace; T *mirror;
    __device__ __forceinline__ T load(int64_t m, int64_t n) const { return out[m * ld + n]; }   // unconditional
    __device__ __forceinline__ void store(int64_t m, int64_t n, T v, T old) const {
        const bool p2 = stats_pow2(bdiv);
        const T nv = stats_value(v, old, beta, wt, bdiv, replace, p2, (T)1 / bdiv);
        out[m * ld + n] = nv;
        if (mirror) mirror[m * ld + n] = nv;
    }
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { store(m, n, v, load(m, n)); }
    // four consecutive n at once (gemm_resident.hpp: a lane of its transposed tile holds four consecutive atoms of a feature)
    bool vec4_ok() const {
        return ld % 4 == 0 && reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0 &&
               reinterpret_cast<uintptr_t>(mirror) % (4 * sizeof(T)) == 0;
    }
    __device__ __forceinline__ vec4 load4(int64_t m, int64_t n) const { return *reinterpret_cast<const vec4 *>(out + m * ld + n); }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, vec4 v, vec4 old) const {
        const bool p2 = stats_pow2(bdiv);
        const T rinv = (T)1 / bdiv;
        vec4 nv;
#pragma unroll
        for (int c = 0; c < 4; ++c) nv[c] = stats_value(v[c], old[c], beta, wt, bdiv, replace, p2, rinv);
        *reinterpret_cast<vec4 *>(out + m * ld + n) = nv;
        if (mirror) *reinterpret_cast<vec4 *>(mirror + m * ld + n) = nv;
    }
};

// ... for the sampled rows only: product row m is feature rows[m]; the mirror is compact (row m)
template <typename T> struct EpiStatsRows {
    static constexpr bool rmw = true;
    T *out; int64_t ld; const int32_t *rows; T beta, wt, bdiv; int replace; T *mirror;
    __device__ __forceinline__ T load(int64_t m, int64_t n) const { return out[(int64_t)rows[m] * ld + n]; }
    __device__ __forceinline__ void store(int64_t m, int64_t n, T v, T old) const {
        const T nv = stats_value(v, old, beta, wt, bdiv, replace, stats_pow2(bdiv), (T)1 / bdiv);
        out[(int64_t)rows[m] * ld + n] = nv;
        if (mirror) mirror[m * ld + n] = nv;
    }
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { store(m, n, v, load(m, n)); }
};

// Everything the code step gathers, in ONE launch: squared row norms of the minibatch, the sampled
// dictionary rows Ds = Dt[subset], the sampled minibatch columns Xs = X[:, subset], the minibatch's code rows.
template <typename T> struct PrepArgs {
    const T *X; int64_t ldx, p; int b; T *xnorm; int n_norm;                 // n_norm = b or 0
    const T *Dt; const int32_t *subset; int64_t s; int k; T *Ds; int n_rows;  // n_rows = cdiv(s, kPrepRows) workgroups or 0
    int64_t s_pad; T *Xs; int gx; int n_cols;                                 // n_cols = gx * b or 0
    const T *code; const int64_t *idx; T *codeb; int n_code;                  // n_code = cdiv(b, kPrepRows) workgroups or 0
    int32_t *stamp, *pos; int32_t step;                                       // stamp[subset[i]] = step, pos[subset[i]] = i
    int fuse_cols;        // the row-norm workgroups also gather the sampled columns of their row (n_cols == 0 then)
};
constexpr int kPrepRows = 8;      // rows per workgroup of prep_kernel's row gathers
template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(PrepArgs<T> a) {
    __shared__ double red[4];
    extern __shared__ __attribute__((aligned(16))) char prep_row_raw[];     // fuse_cols: the staged row (p elements)
    int id = (int)blockIdx.x;
    if (id < a.n_norm) {                                 // dict_fact_fast.pyx:334 uses dot(y, y)
        const T *x = a.X + (int64_t)id * a.ldx;
        T *rowl = reinterpret_cast<T *>(prep_row_raw);
        const bool stage = a.fuse_cols != 0;              // (workgroup-uniform; implies 16-byte aligned rows)
        constexpr int V = 16 / sizeof(T);
        typedef T vec_t __attribute__((ext_vector_type(V)));
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const bool vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0);
        const int64_t nv = vec ? a.p / V : 0;
        const vec_t *xv = reinterpret_cast<const vec_t *>(x);
        // Twelve 16-byte requests per thread in flight (one pass covers a row of 12 288 floats - the metric's 10 000): with
        // four per pass the row took three memory round trips one after the other, 5 of the kernel's 6.8 us.  Clamped
        // addresses, the tail masked afterwards: no branch around a load.
        constexpr int NQ = 12;
        for (int64_t f0 = 0; f0 < nv; f0 += (int64_t)NQ * 256) {
            vec_t av[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + 256 * q;
                av[q] = xv[f < nv ? f : nv - 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + 256 * q;
                if (f < nv) {
                    if (stage) reinterpret_cast<vec_t *>(rowl)[f] = av[q];
#pragma unroll
                    for (int c = 0; c < V; ++c) {
                        const double v = (double)av[q][c];
                        if ((q & 3) == 0) s0 += v * v;
                        else if ((q & 3) == 1) s1 += v * v;
                        else if ((q & 3) == 2) s2 += v * v;
                        else s3 += v * v;
                    }
                }
            }
        }
        for (int64_t e = nv * V + threadIdx.x; e < a.p; e += 256) {
            if (stage) rowl[e] = x[e];
            s1 += (double)x[e] * (double)x[e];
        }
        double sum = (s0 + s1) + (s2 + s3);
        sum = block_sum(sum, red);                        // (its barriers also publish the staged row)
        if (threadIdx.x == 0) a.xnorm[id] = (T)sum;
        if (stage) {
            // Xs[id][:] = X[id][subset] from the row in LDS: the minibatch is read from HBM once (a gather of 4-byte
            // elements straight from memory drags a whole cache line in for every sampled column: 32 MB per
            // minibatch for 1 MB of values at the metric's shape)
            T *dst = a.Xs + (int64_t)id * a.s_pad;
            constexpr int NC = 8;                             // column indices requested together (one round trip per 2048)
            for (int64_t c0 = 0; c0 < a.s_pad; c0 += (int64_t)NC * 256) {
                int32_t col[NC];
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int64_t c = c0 + threadIdx.x + 256 * q;
                    col[q] = a.subset[c < a.s ? c : a.s - 1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int64_t c = c0 + threadIdx.x + 256 * q;
                    if (c < a.s_pad) dst[c] = (c < a.s) ? rowl[col[q]] : (T)0;
                }
            }
        }
        return;
    }
    id -= a.n_norm;
    if (id < a.n_rows) {                                 // kPrepRows sampled rows per workgroup, their loads requested together
        // (a row each: s workgroups of one element per thread - the launch lasted as long as its 1500 workgroups took to start)
        const int64_t r0 = (int64_t)id * kPrepRows;
        int64_t sub[kPrepRows];
#pragma unroll
        for (int r = 0; r < kPrepRows; ++r) sub[r] = a.subset[r0 + r < a.s ? r0 + r : a.s - 1];
        if (a.stamp && threadIdx.x < kPrepRows && r0 + threadIdx.x < a.s) {
            const int32_t f = a.subset[r0 + threadIdx.x];
            a.stamp[f] = a.step; a.pos[f] = (int32_t)(r0 + threadIdx.x);
        }
        for (int c = threadIdx.x; c < a.k; c += 256) {
            T v[kPrepRows];
#pragma unroll
            for (int r = 0; r < kPrepRows; ++r) v[r] = a.Dt[sub[r] * a.k + c];
#pragma unroll
            for (int r = 0; r < kPrepRows; ++r)
                if (r0 + r < a.s) a.Ds[(r0 + r) * a.k + c] = v[r];
        }
        return;
    }
    id -= a.n_rows;
    if (id < a.n_cols) {
        // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: the gx workgroups of a row - a random gather over
        // the whole row, which only pays one memory fetch per line if the row sits in ONE L2 while they work on it - are numbered
        // so that they all land on the XCD (row % 8), the one the row's norm workgroup ran on.  (With row = id / gx every XCD
        // fetched every row: at p = 200 000 the launch took 111 us for 205 MB of X, 81 us this way.  Measured and not kept: the
        // row norms as partial sums of the same workgroups, a contiguous 1 / gx of the row each, summed by the last to arrive -
        // one pass over X instead of two, and 102 us: 16 384 workgroups with a block reduction, a ticket and 12 KB of
        // contiguous reads each; with agent-scope fences around the ticket 706 us - a release writes back the XCD's whole L2.)
        int i = id / a.gx, bx = id % a.gx;
        if (a.b % 8 == 0) {
            const int xcd = (int)(blockIdx.x % 8), slot = id / 8;
            i = xcd + 8 * (slot / a.gx);
            bx = slot % a.gx;
        }
        const T *row = a.X + (int64_t)i * a.ldx;
        for (int64_t f = (int64_t)bx * 256 + threadIdx.x; f < a.s_pad; f += (int64_t)a.gx * 256)
        {
            const T xv = row[a.subset[f < a.s ? f : a.s - 1]];          // (unconditional, clamped)
            a.Xs[(int64_t)i * a.s_pad + f] = (f < a.s) ? xv : (T)0;
        }
        return;
    }
    id -= a.n_cols;
    if (id < a.n_code) {                                 // kPrepRows code rows per workgroup
        const int r0 = id * kPrepRows;
        int64_t src[kPrepRows];
#pragma unroll
        for (int r = 0; r < kPrepRows; ++r) src[r] = a.idx[r0 + r < a.b ? r0 + r : a.b - 1] * a.k;
        for (int c = threadIdx.x; c < a.k; c += 256) {
            T v[kPrepRows];
#pragma unroll
            for (int r = 0; r < kPrepRows; ++r) v[r] = a.code[src[r] + c];
#pragma unroll
            for (int r = 0; r < kPrepRows; ++r)
                if (r0 + r < a.b) a.codeb[(int64_t)(r0 + r) * a.k + c] = v[r];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T *dst, int64_t n, T v) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) dst[e] = v;
}

}  // namespace modl

namespace modl {
// diagnostics (modl_debug_set(MODL_DEBUG_STAGE_AHEAD, 0)): every minibatch of a chunk stages its own parameters
std::atomic<int> g_stage_ahead{1};
}
using namespace modl;

struct modl_somf_plan {
    modl_somf_desc d;
    int device = 0;                    // the HIP device the plan was created on: every entry point makes it current
    size_t tsz;
    // device arena
    char *dws = nullptr;
    size_t dws_bytes = 0;
    size_t off_Gpad = 0; int ld_gpad = 0;   // zero-padded copy of a shared Gram whose size the vectorised solver does not take as is
    size_t off_params2[2];             // two device parameter blocks: the next minibatch's may be staged while this one's is read
    int cur_par = 0;                   // the block of the minibatch in flight
    size_t off_params, off_xnorm, off_Dx, off_H0, off_G, off_F, off_Linv, off_split, off_du, off_sweeps, off_Ds, off_Xs, off_codeb, off_level;
    int last_b = 0;
    int64_t last_s = 0;
    // the persistent dictionary-update launch's words (pinned, device-mapped; BcdPersistArgs::flags): [0] an update gave up
    // half-way, [1] launches that could not run and were completed by one workgroup.  Read by the host before every enqueue.
    unsigned int *pflags = nullptr, *pflags_dev = nullptr;
    unsigned int recoveries_seen = 0;
    bool persist_ok = true;            // false once a persistent launch of this plan could not run: one launch per block from then on
    size_t split_bytes, du_bytes, params_bytes;
    // per-batch parameter block (device copy of the host arrays), layout within params:
    size_t po_idx, po_subset, po_order, po_wsample;
    // pinned staging ring
    char *hstage[kStageSlots] = {nullptr};
    char *hstage_dev[kStageSlots] = {nullptr};   // device-side addresses of the pinned slots
    // slot protection without a stream event (a hipEventRecord between two kernels costs ~5 us of bubble, measured):
    // the staging kernel acknowledges a slot by writing its use count into the slot's last 8 bytes (host memory)
    unsigned long long slot_uses[kStageSlots] = {0};
    hipStream_t slot_stream[kStageSlots] = {nullptr};      // the stream of each slot's last use (its acknowledgement comes from there)
    double wait_ms = 0;                // host time spent waiting for a staging slot (the host is kStageSlots ahead)
    int slot = 0;
    // currently staged batch
    bool staged = false;
    bool has_idx = false, has_subset = false;
    std::vector<int64_t> h_order_copy;
    // the NEXT minibatch, staged ahead by the previous step (modl_somf_partial_fit_chunk): its block is off_params2[cur_par ^ 1]
    bool ahead = false;
    bool ahead_has_idx = false, ahead_has_subset = false;
    std::vector<int64_t> ahead_order;
    // profiling
    bool prof = false;
    bool head_pending = false;         // the last phase 1 was the two-phase one: phase 2 reads C and the sampled rows
                                       // of B from the (summed) head instead of the rank's partial statistics
    int64_t head_elems = 0;            // k*k + (rows of B in the head) * k
    void *Bsum = nullptr;              // [p][k] (lazily allocated): the summed rows of B_, scattered for the dictionary update
    void *Gslots = nullptr;            // zero-padded copies of per-sample Gram matrices (lazily allocated, launch_cd_per_sample)
    size_t Gslots_bytes = 0;
    void *own_head = nullptr;          // [k*k + p*k] (lazily allocated): the head buffer of modl_somf_step_dist
    bool ride_pending = false;         // single-GPU step: the B_ update of the rows that were not sampled rides along
    StatsRider rider{};                // the dictionary update (see StatsRider)
    int32_t step_id = 0;
    // diagnostics (modl_somf_sweeps_history): the sweep counts of EVERY minibatch, a ring of hist_cap slots of max_batch
    int32_t *hist = nullptr;
    int64_t hist_cap = 0, hist_n = 0;
    const int32_t *last_sweeps_ptr = nullptr;
    size_t off_stamp = 0, off_pos = 0, off_gstamps = 0;
    unsigned prof_mask = ~0u;          // sections that record events
    int prof_stride = 1;               // ... on every prof_stride-th minibatch only (an event pair costs ~9 us of bubble)
    long prof_step = 0;
    std::vector<hipEvent_t> pev;       // 2 * kProfPool events
    std::vector<int> psec, plaunch;
    int pcount = 0;
    double ms[SEC_COUNT] = {0};
    int64_t launches[SEC_COUNT] = {0}, calls[SEC_COUNT] = {0};
};

namespace {

size_t params_layout(modl_somf_plan *pl) {
    const modl_somf_desc &d = pl->d;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 64); return r; };
    pl->po_idx = take(sizeof(int64_t) * (size_t)d.max_batch);
    pl->po_subset = take(sizeof(int32_t) * (size_t)d.p);
    pl->po_order = take(sizeof(int32_t) * (size_t)d.k);
    pl->po_wsample = take(pl->tsz * (size_t)d.max_batch);
    return o;
}

int validate_desc(const modl_somf_desc *d) {
    if (!d) return MODL_EINVAL;
    if (d->dtype != MODL_F32 && d->dtype != MODL_F64) return MODL_EINVAL;
    if (d->k <= 0 || d->k > 1024 || d->p <= 0 || d->p > 0x7fffffff || d->n_samples < 0 || d->max_batch <= 0)
        return MODL_EINVAL;
    if (d->G_agg < 0 || d->G_agg > 2 || d->Dx_agg < 0 || d->Dx_agg > 2) return MODL_EINVAL;
    if (d->optimizer != MODL_OPT_VARIATIONAL && d->optimizer != MODL_OPT_SGD) return MODL_EINVAL;
    if (!(d->code_l1_ratio >= 0.0 && d->code_l1_ratio <= 1.0)) return MODL_EINVAL;
    if (!(d->comp_l1_ratio >= 0.0 && d->comp_l1_ratio <= 1.0)) return MODL_EINVAL;
    return MODL_OK;
}

struct ProfScope {
    modl_somf_plan *pl;
    hipStream_t st;
    int sec, idx = -1;
    int launches = 0;
    ProfScope(modl_somf_plan *p, hipStream_t s, int section) : pl(p), st(s), sec(section) {
        if (!pl->prof || !((pl->prof_mask >> section) & 1u)) return;
        if (pl->prof_stride > 1 && (pl->prof_step % pl->prof_stride) != 0) return;
        if (pl->pcount >= kProfPool) return;           // pool full until the next prof_get/reset
        idx = pl->pcount++;
        (void)hipEventRecord(pl->pev[2 * idx], st);
    }
    ~ProfScope() {
        if (idx < 0) return;
        (void)hipEventRecord(pl->pev[2 * idx + 1], st);
        pl->psec[idx] = sec;
        pl->plaunch[idx] = launches;
    }
};

int prof_flush(modl_somf_plan *pl) {
    for (int i = 0; i < pl->pcount; ++i) {
        MODL_HIP(hipEventSynchronize(pl->pev[2 * i + 1]));
        float ms = 0;
        MODL_HIP(hipEventElapsedTime(&ms, pl->pev[2 * i], pl->pev[2 * i + 1]));
        pl->ms[pl->psec[i]] += ms;
        pl->launches[pl->psec[i]] += pl->plaunch[i];
        pl->calls[pl->psec[i]] += 1;
    }
    pl->pcount = 0;
    return MODL_OK;
}

// Parameter block: pinned host slot -> HBM by a kernel that reads the (device-mapped) pinned memory over
// the host link.  A hipMemcpyAsync here would hop to the copy engine and back between two kernels of the
// same stream, which costs several times the kernel floor.
// ONE workgroup copies the USED ranges of the slot (sample indices, subset, order, sample weights: a few KB of the
// p-sized block); when all its loads have landed the slot may be overwritten by the host, which thread 0 tells it
// by writing the slot's use count into the acknowledgement word of the slot itself (host memory, system scope).
__global__ __launch_bounds__(1024) void stage_params_kernel(StageRide r) {
    stage_copy(r.src, r.dst, r.off16, r.n16, r.ack, r.use, (int)threadIdx.x, 1024);
}

// The host half of staging a minibatch: validate, fill a pinned slot, describe the copy (into device block `par`).
// has_idx / has_subset / order: what the step needs to know about the staged arrays.
struct StageJob { StageRide ride; bool has_idx = false, has_subset = false; std::vector<int64_t> order; };
template <typename T>
int stage_fill(modl_somf_plan *pl, const modl_somf_batch *bt, hipStream_t st, int par, StageJob &job) {
    const modl_somf_desc &d = pl->d;
    if (bt->b <= 0 || bt->b > d.max_batch || !bt->d_X || bt->ldx < d.p) return MODL_EINVAL;
    if (bt->s < 0 || bt->s > d.p || !bt->h_order) return MODL_EINVAL;
    if (!bt->h_subset && bt->s != d.p) return MODL_EINVAL;
    if ((d.G_agg == MODL_AGG_AVERAGE || d.Dx_agg == MODL_AGG_AVERAGE) && !bt->h_w_sample) return MODL_EINVAL;
    // (everything is validated BEFORE a slot is taken: a rejected minibatch leaves the ring as it was)
    if (bt->h_sample_idx) {
        for (int i = 0; i < bt->b; ++i)
            if (bt->h_sample_idx[i] < 0 || bt->h_sample_idx[i] >= d.n_samples) return MODL_EINVAL;
    } else if (bt->b > d.n_samples) {
        return MODL_EINVAL;
    }
    if (bt->h_subset)
        for (int i = 0; i < bt->s; ++i)
            if (bt->h_subset[i] < 0 || bt->h_subset[i] >= d.p) return MODL_EINVAL;
    for (int i = 0; i < d.k; ++i)
        if (bt->h_order[i] < 0 || bt->h_order[i] >= d.k) return MODL_EINVAL;
    const int slot = pl->slot;
    pl->slot = (slot + 1) % kStageSlots;
    char *h = pl->hstage[slot];
    {   // the previous use of this slot (kStageSlots minibatches ago) must have been read by its staging copy
        volatile unsigned long long *ack = reinterpret_cast<volatile unsigned long long *>(h + align_up(pl->params_bytes, 16));
        if (*ack < pl->slot_uses[slot]) {                      // the host is kStageSlots minibatches ahead of the device
            const auto t0 = std::chrono::steady_clock::now();
            for (long spins = 0; *ack < pl->slot_uses[slot]; ++spins) {
                if (spins > 64) {
                    // the acknowledgement comes from the stream the slot was LAST USED on - not necessarily the current
                    // one (a caller that switches streams between minibatches): only if THAT stream has drained and
                    // the word is still missing is the state broken
                    hipStream_t last = pl->slot_stream[slot];
                    if (hipStreamQuery(last) == hipSuccess && *ack < pl->slot_uses[slot]) return MODL_ESTATE;
                    sched_yield();
                }
            }
            pl->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    job.has_idx = bt->h_sample_idx != nullptr;
    if (job.has_idx) std::memcpy(h + pl->po_idx, bt->h_sample_idx, sizeof(int64_t) * (size_t)bt->b);
    job.has_subset = bt->h_subset != nullptr;
    if (job.has_subset) {
        int32_t *dst = reinterpret_cast<int32_t *>(h + pl->po_subset);
        for (int i = 0; i < bt->s; ++i) dst[i] = (int32_t)bt->h_subset[i];
    }
    {
        int32_t *dst = reinterpret_cast<int32_t *>(h + pl->po_order);
        job.order.assign(bt->h_order, bt->h_order + d.k);
        for (int i = 0; i < d.k; ++i) dst[i] = (int32_t)bt->h_order[i];
    }
    if (bt->h_w_sample) std::memcpy(h + pl->po_wsample, bt->h_w_sample, pl->tsz * (size_t)bt->b);
    // the sections of the block start on 64-byte boundaries (params_layout): whole 16-byte words of each
    StageRide &rg = job.ride;
    auto range = [&](int i, size_t off, size_t bytes) {
        rg.off16[i] = (unsigned int)(off / 16);
        rg.n16[i] = (unsigned int)((bytes + 15) / 16);
    };
    range(0, pl->po_idx, job.has_idx ? sizeof(int64_t) * (size_t)bt->b : 0);
    range(1, pl->po_subset, job.has_subset ? sizeof(int32_t) * (size_t)bt->s : 0);
    range(2, pl->po_order, sizeof(int32_t) * (size_t)d.k);
    range(3, pl->po_wsample, bt->h_w_sample ? pl->tsz * (size_t)bt->b : 0);
    rg.src = reinterpret_cast<const uint4 *>(pl->hstage_dev[slot]);
    rg.dst = reinterpret_cast<uint4 *>(pl->dws + pl->off_params2[par]);
    rg.ack = reinterpret_cast<unsigned long long *>(pl->hstage_dev[slot] + align_up(pl->params_bytes, 16));
    rg.use = ++pl->slot_uses[slot];
    rg.consumed = 0;
    pl->slot_stream[slot] = st;
    return MODL_OK;
}
// the copy as a launch of its own (the first minibatch of a call, the Python loop, paths without riders)
static int stage_launch(const StageRide &r, hipStream_t st) {
    hipLaunchKernelGGL(stage_params_kernel, dim3(1), dim3(1024), 0, st, r);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

// copy the per-batch host arrays into the device parameter block through a pinned slot - unless the previous step of
// the chunk loop staged this minibatch ahead (its copy rode on that step's last launch)
template <typename T>
int stage_batch(modl_somf_plan *pl, const modl_somf_batch *bt, hipStream_t st) {
    if (pl->ahead) {
        pl->ahead = false;
        pl->cur_par ^= 1;
        pl->has_idx = pl->ahead_has_idx;
        pl->has_subset = pl->ahead_has_subset;
        pl->h_order_copy.swap(pl->ahead_order);
    } else {
        StageJob job;
        MODL_TRY(stage_fill<T>(pl, bt, st, pl->cur_par ^ 1, job));
        MODL_TRY(stage_launch(job.ride, st));
        pl->cur_par ^= 1;
        pl->has_idx = job.has_idx;
        pl->has_subset = job.has_subset;
        pl->h_order_copy.swap(job.order);
    }
    pl->off_params = pl->off_params2[pl->cur_par];
    pl->staged = true;
    return MODL_OK;
}

template <typename T>
Operand dict_rows(const T *Dt, int k, const int32_t *subset) {   // element (i = atom, kk = feature) of Dt
    Operand o;
    o.ptr = Dt; o.si = 1; o.sk = k; o.gk = gather32(subset);
    return o;
}

// G = scale * Dt[subset]^T Dt[subset]   (k x k, bitwise symmetric)
template <typename T, class Epi>
int gram_of_rows(modl_somf_plan *pl, hipStream_t st, const T *Dt, const int32_t *subset, int64_t s, const Epi &epi,
                 int *nl) {
    const Operand A = dict_rows<T>(Dt, pl->d.k, subset);
    SplitWs ws{pl->dws + pl->off_split, pl->split_bytes};
    return launch_gemm<T, Epi>(st, A, A, pl->d.k, pl->d.k, s, epi, ws, nl);
}

template <typename T>
int solve_codes(modl_somf_plan *pl, hipStream_t st, const T *G, int64_t g_stride, const int64_t *g_idx, T *Dx,
                const T *xnorm2, T *code, const int64_t *d_idx, int b, int32_t *d_sweeps, int *nl, T *H0buf, T *Fbuf,
                T *scatter_dst = nullptr, const int64_t *scatter_idx = nullptr, const int64_t *h_gidx = nullptr) {
    const modl_somf_desc &d = pl->d;
    const int k = d.k;
    if (d.code_l1_ratio == 0.0 && chol_blocked(k, sizeof(T), g_stride == 0)) {   // blocked, on the matrix cores
        if (g_stride && g_idx && !h_gidx) return MODL_EINVAL;         // per-sample Grams are walked from the host
        T *Linv = reinterpret_cast<T *>(pl->dws + pl->off_Linv);
        MODL_TRY(ridge_solve_wide<T>(st, G, g_stride, h_gidx, Fbuf, Linv, Dx, b, k, (T)d.code_alpha, code, d_idx));
        *nl += 2;
        if (scatter_dst) {
            hipLaunchKernelGGL((scatter_rows_T_kernel<T, int64_t>), dim3((unsigned)b), dim3(256), 0, st, scatter_dst,
                               (int64_t)k, scatter_idx, (int64_t)b, (int64_t)k, Dx, (int64_t)k);
            MODL_LAUNCH_CHECK();
            ++*nl;
        }
        return MODL_OK;
    }
    if (d.code_l1_ratio == 0.0) {                                     // ridge: dict_fact_fast.pyx:82-94, 174-197
        const int nmat = g_stride ? b : 1;
        if (ridge_small_applies<T>(k)) {                              // factor + substitutions in one launch, in LDS
            MODL_TRY(launch_ridge_small<T>(st, G, g_stride, g_idx, Dx, b, k, (T)d.code_alpha, code, d_idx));
            *nl += 1;
        } else {
            MODL_TRY(launch_cholesky<T>(st, G, g_stride, g_idx, Fbuf, k, (T)d.code_alpha, nmat));
            MODL_TRY(launch_chol_solve<T>(st, Fbuf, g_stride ? (int64_t)k * k : 0, Dx, b, k, code, d_idx));
            *nl += 2;
        }
        if (scatter_dst) {
            hipLaunchKernelGGL((scatter_rows_T_kernel<T, int64_t>), dim3((unsigned)b), dim3(256), 0, st, scatter_dst,
                               (int64_t)k, scatter_idx, (int64_t)b, (int64_t)k, code, (int64_t)k);
            MODL_LAUNCH_CHECK();
            ++*nl;
        }
        return MODL_OK;
    }
    // H0 = Q w is formed inside the solver (rows stream through its prefetch ring: ~6 us at k = 256, less
    // than the launch of a separate 256 x 256 x 256 product that only occupies 16 workgroups)
    const T *H0 = nullptr;
    (void)H0buf;
    CdArgs<T> a;
    a.G = G; a.g_stride = g_stride; a.g_idx = g_idx; a.Dx = Dx; a.xnorm2 = xnorm2; a.H0 = H0; a.code = code;
    a.idx = d_idx;
    a.code2 = scatter_dst; a.idx2 = scatter_idx;
    a.g_pad_rows = (G == reinterpret_cast<const T *>(pl->dws + pl->off_G)) ? 16 : 0;
    if (!g_stride && pl->ld_gpad) {              // any k on the vectorised kernel (cd_solver.hip: cd_padded_ld)
        T *Gp = reinterpret_cast<T *>(pl->dws + pl->off_Gpad);
        MODL_TRY(launch_cd_pad_gram<T>(st, G, k, Gp, pl->ld_gpad));
        ++*nl;
        a.G = Gp; a.ldg = pl->ld_gpad; a.g_pad_rows = 16;
    }
    a.sweeps = d_sweeps; a.b = b; a.k = k;
    a.alpha = (T)((T)d.code_alpha * (T)d.code_l1_ratio);
    a.beta = (T)((double)(T)d.code_alpha * (1.0 - (double)(T)d.code_l1_ratio));
    a.tol = (T)d.tol; a.max_iter = d.max_iter; a.positive = d.code_pos;
    if (g_stride && k >= 32 && (cd_split_enabled() || !cd_one_wave_covers(k)) && !cd_split_applies<T>(a)) {
        // a Gram matrix per sample of a size the four-wavefront solver does not take as stored: zero-padded copies,
        // a slice of the minibatch at a time (cd_solver.hip: launch_cd_per_sample)
        if (!pl->Gslots) {
            pl->Gslots_bytes = cd_per_sample_scratch_bytes(sizeof(T), d.max_batch, k);
            MODL_HIP(hipMalloc(&pl->Gslots, pl->Gslots_bytes));
            MODL_HIP(hipMemsetAsync(pl->Gslots, 0, pl->Gslots_bytes, st));
        }
        MODL_TRY(launch_cd_per_sample<T>(st, a, static_cast<T *>(pl->Gslots), pl->Gslots_bytes));
        *nl += 2;
        return MODL_OK;
    }
    MODL_TRY(launch_cd<T>(st, a));
    ++*nl;
    return MODL_OK;
}

constexpr int64_t kResidentMinRows = 4096;

// Two statistics products (contraction over the b samples of the minibatch, both operands contiguous along their
// rows) in ONE launch; M0 == 0: only the second.  f32 with b <= 256 goes through the 32 x 32 / 16x16x4 tiling
// (gemm_stats_pair_kernel), anything else through the generic pair kernel or, unaligned, the gather kernel.  Every
// statistics product of the step comes through here, so that the fused, the two-phase and the riding variants of the
// same product sum in the same order (bit-identical results).
template <typename T, class Epi0, class Epi1>
int stats_pair(hipStream_t st, const DenseOperand &A0, const DenseOperand &B0, int64_t M0, int64_t N0, const Epi0 &e0,
               const DenseOperand &A1, const DenseOperand &B1, int64_t M1, int64_t N1, const Epi1 &e1, int64_t K,
               const SplitWs &sws, int *launches, unsigned long long *dbg = nullptr) {
    if constexpr (std::is_same<T, float>::value) {
        auto P0 = plan_stats<Epi0>(A0, B0, M0, N0, K, e0);
        // a VERY tall second problem (the whole p x k product at p >= 65 536): k-wide tiles, X read once (gemm_wide.hpp),
        // 32 features x 256 atoms.  The tile is latency-bound (its code tiles come from L2 once per 32 features, its old
        // values and its X tile from HBM), so what counts is how many workgroups share a compute unit: an 8-sample code
        // tile and swizzled, unpadded LDS rows make it 48 KB - THREE per compute unit: 0.351 ms at p = 200 000 (75 TFLOP/s,
        // 48 % of the f32 matrix peak); 16 samples, padded rows, two per unit: 0.385-0.397; 32 samples, one per unit:
        // 0.489; 4 samples, four per unit: 0.387 (a barrier every 256 matrix-core cycles); 64 features, one per unit:
        // 0.46.  At p = 10 000 the 32 x 32 tiles are faster (27 us against 37 us: ten times the workgroups to hide latency)
        // a TALL second problem (the whole p x k product: reduction 1, or config 5's 200 000 features) with at most 256 atoms:
        // persistent workgroups with the code matrix resident in registers (gemm_resident.hpp), tiles of 16 or 32 features -
        // whichever leaves the slowest workgroup less to contract
        if (P0.ok && M1 >= kResidentMinRows && N1 <= 256 && K % 4 == 0 && !dbg && g_stats_resident.load(std::memory_order_relaxed)) {
            auto W = plan_wide<16, Epi1>(A1, B1, M1, N1, K, e1);
            if (W.ok) {
                static const int ncu = [] {
                    int dev = 0;
                    hipDeviceProp_t prop;
                    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                        return prop.multiProcessorCount;
                    return 256;
                }();
                return launch_gemm_stats_resident_pair<16, Epi0, Epi1>(st, P0, W, ncu, launches);
            }
        }
        if (P0.ok && cdiv(M1, 64) >= 1024 && !dbg) {
            auto W = plan_wide<32, Epi1>(A1, B1, M1, N1, K, e1);
            if (W.ok) return launch_gemm_stats_wide_pair<32, Epi0, Epi1, 256, 8, 0, true>(st, P0, W, launches);
        }   // (32 x 128 wide tiles as their own launch at p = 10 000: 40 us against 28 us for the 32 x 32 tiles, measured)
        auto P1 = plan_stats<Epi1>(A1, B1, M1, N1, K, e1);
        if (P0.ok && P1.ok) {
            P1.dbg = dbg;
            return launch_gemm_stats_pair<Epi0, Epi1>(st, P0, P1, launches);
        }
    }
    DenseProblem<T, Epi0> P0;
    P0.epi = e0;
    bool ok0 = true;
    if (M0 > 0) {
        P0 = plan_dense<T, Epi0>(A0, B0, M0, N0, K, e0, nullptr, 0);
        ok0 = P0.ok;
    }
    auto P1 = plan_dense<T, Epi1>(A1, B1, M1, N1, K, e1, nullptr, 0, 512, 1, kStatBM, kStatBN);
    if (ok0 && P1.ok)
        return launch_gemm_dense_pair<T, true, true, Epi0, true, true, Epi1, kStatBM, kStatBN, kStatBK>(st, P0, P1, launches);
    if (M0 > 0) MODL_TRY((launch_gemm_dense<T, Epi0>(st, A0, B0, M0, N0, K, e0, sws, launches)));
    return launch_gemm_dense<T, Epi1>(st, A1, B1, M1, N1, K, e1, sws, launches);
}

// Codes of the minibatch, then the statistics update C <- (1 - w) C + (w / b_global) code^T code (the same for B_)
// in the epilogues of the increment products.  two_phase: the updated C and the updated rows of B_ that the
// dictionary update will read are also written to the head buffer `delta` = [C (k*k) | rows of Bt (compact)].
template <typename T>
int phase1(modl_somf_plan *pl, const modl_somf_state *stt, const modl_somf_batch *bt, T *delta, hipStream_t st,
           bool two_phase) {
    const modl_somf_desc &d = pl->d;
    const int k = d.k, b = bt->b;
    const int64_t p = d.p, s = bt->s;
    if (!stt || !stt->d_Dt || !stt->d_code || (two_phase && !delta)) return MODL_EINVAL;
    if (!stt->d_Bt || !stt->d_C || bt->b_global <= 0) return MODL_EINVAL;
    if (d.G_agg == MODL_AGG_FULL && !stt->d_G) return MODL_EINVAL;
    if (d.G_agg == MODL_AGG_AVERAGE && !stt->d_G_average) return MODL_EINVAL;
    if (d.Dx_agg == MODL_AGG_AVERAGE && !stt->d_Dx_average) return MODL_EINVAL;
    MODL_TRY(stage_batch<T>(pl, bt, st));
    ++pl->prof_step;
    char *P = pl->dws + pl->off_params;
    const int64_t *d_idx = pl->has_idx ? reinterpret_cast<const int64_t *>(P + pl->po_idx) : nullptr;
    const int32_t *d_subset = pl->has_subset ? reinterpret_cast<const int32_t *>(P + pl->po_subset) : nullptr;
    const T *d_wsample = reinterpret_cast<const T *>(P + pl->po_wsample);
    const T *X = static_cast<const T *>(bt->d_X);
    const T *Dt = static_cast<const T *>(stt->d_Dt);
    T *code = static_cast<T *>(stt->d_code);
    T *xnorm = reinterpret_cast<T *>(pl->dws + pl->off_xnorm);
    T *Dx = reinterpret_cast<T *>(pl->dws + pl->off_Dx);
    T *H0 = reinterpret_cast<T *>(pl->dws + pl->off_H0);
    T *Gbuf = reinterpret_cast<T *>(pl->dws + pl->off_G);
    T *Fbuf = reinterpret_cast<T *>(pl->dws + pl->off_F);
    SplitWs sws{pl->dws + pl->off_split, pl->split_bytes};
    int32_t *d_sweeps = reinterpret_cast<int32_t *>(pl->dws + pl->off_sweeps);
    if (pl->hist) d_sweeps = pl->hist + (pl->hist_n++ % pl->hist_cap) * (int64_t)d.max_batch;
    pl->last_sweeps_ptr = d_sweeps;
    pl->last_b = b;
    const T red = (T)bt->reduction;

    T *Dsb = reinterpret_cast<T *>(pl->dws + pl->off_Ds);
    T *Xsb = reinterpret_cast<T *>(pl->dws + pl->off_Xs);
    T *codeb = reinterpret_cast<T *>(pl->dws + pl->off_codeb);
    const int64_t s_pad = (s + 3) / 4 * 4;
    // the minibatch's code rows, compact (warm starts now, solutions after the solve)
    const bool cd_on_compact = d.code_l1_ratio != 0.0 && d.G_agg != MODL_AGG_AVERAGE;
    T *cb = codeb;
    if (!d_idx) cb = code;                                             // rows 0..b-1 are already contiguous

    T *ws_split = reinterpret_cast<T *>(sws.ptr);
    const size_t ws_elems = sws.bytes / sizeof(T);
    bool ride = false;
    pl->ride_pending = false;
    pl->head_pending = false;
    {   // ---- Dx, G  (dict_fact.py:588-620)
        ProfScope ps(pl, st, SEC_CODE_GEMM);
        // compaction: gather once, contract dense.  Ds = Dt[subset] (whole 1 KiB feature rows),
        // Xs = X[:, subset].  With every feature sampled nothing is copied.  One launch for the row
        // norms and every gather.
        const T *Dsrc = Dt, *Xsrc = X;
        int64_t ldxs = bt->ldx;
        const bool need_sub = d_subset && (d.Dx_agg != MODL_AGG_FULL || d.G_agg != MODL_AGG_FULL);
        PrepArgs<T> pa;
        pa.X = X; pa.ldx = bt->ldx; pa.p = p; pa.b = b; pa.xnorm = xnorm; pa.n_norm = (d.code_l1_ratio != 0.0) ? b : 0;
        pa.Dt = Dt; pa.subset = d_subset; pa.s = s; pa.k = k; pa.Ds = Dsb; pa.n_rows = need_sub ? (int)cdiv(s, kPrepRows) : 0;
        pa.s_pad = s_pad; pa.Xs = Xsb; pa.gx = (int)std::min<int64_t>(cdiv(s_pad, 256), 64);
        pa.n_cols = (need_sub && d.Dx_agg != MODL_AGG_FULL) ? pa.gx * b : 0;
        // the column gather rides with the row norms when a row fits in LDS next to nothing else (<= 64 KB) and the
        // rows are 16-byte aligned
        const size_t row_bytes = sizeof(T) * (size_t)p;
        pa.fuse_cols = (pa.n_cols > 0 && pa.n_norm == b && row_bytes <= 64 * 1024 &&
                        reinterpret_cast<uintptr_t>(X) % 16 == 0 && (bt->ldx * sizeof(T)) % 16 == 0) ? 1 : 0;
        if (pa.fuse_cols) pa.n_cols = 0;
        pa.code = code; pa.idx = d_idx; pa.codeb = codeb; pa.n_code = (cd_on_compact && d_idx) ? (int)cdiv(b, kPrepRows) : 0;
        const bool proper = need_sub && s > 0 && s < p;               // a proper subset, gathered
        // only the sampled rows of B_ are needed by the dictionary update -> the rest of the B_ update is deferred
        // and rides along its launches; the sampled features are stamped so that the rider leaves them alone
        // (not for very large feature counts: at p = 200 000 the p x k product is 26 GFLOP, the launches of the
        // dictionary update would wait for their riders - it runs as its own k-wide launch instead, X read once:
        // 0.39 + 0.39 ms against 0.94 ms riding, measured)
        ride = proper && d.Dx_agg != MODL_AGG_FULL && !(d.flags & MODL_FLAG_NO_RIDER) && cdiv(p, 64) < 1024;
        if (ride) pl->step_id = (pl->step_id == 0x7fffffff) ? 1 : pl->step_id + 1;
        pa.stamp = ride ? reinterpret_cast<int32_t *>(pl->dws + pl->off_stamp) : nullptr;
        pa.pos = reinterpret_cast<int32_t *>(pl->dws + pl->off_pos);
        pa.step = pl->step_id;
        if (need_sub) {
            Dsrc = Dsb;
            if (d.Dx_agg != MODL_AGG_FULL) { Xsrc = Xsb; ldxs = s_pad; }
        }
        const int n_prep = pa.n_norm + pa.n_rows + pa.n_cols + pa.n_code;
        if (n_prep > 0) {
            hipLaunchKernelGGL((prep_kernel<T>), dim3((unsigned)n_prep), dim3(256), pa.fuse_cols ? row_bytes : 0, st, pa);
            MODL_LAUNCH_CHECK();
            ++ps.launches;
        }
        DenseOperand A, B;
        int64_t Kdim;
        T scale;
        if (d.Dx_agg == MODL_AGG_FULL) {                              // X . D^T
            A.ptr = X; A.si = bt->ldx; A.sk = 1;
            B.ptr = Dt; B.si = 1; B.sk = k;
            Kdim = p; scale = 1;
        } else {                                                       // X[:, subset] . D[:, subset]^T * reduction
            A.ptr = Xsrc; A.si = ldxs; A.sk = 1;
            B.ptr = Dsrc; B.si = 1; B.sk = k;
            Kdim = s; scale = red;
        }
        // the Dx product and the Gram product are independent: one launch (+ one for both split-K sums)
        DenseOperand Dg;
        Dg.ptr = Dsrc; Dg.si = 1; Dg.sk = k;
        const bool want_G = d.G_agg != MODL_AGG_FULL;
        bool paired = false;
        if (want_G && A.si != 1) {
            EpiStore<T> epiG{Gbuf, k, red};
            const size_t half = ws_elems / 2;
            // the two products share the chip: 2 workgroups per compute unit in total (256 each) - with 512 each the
            // split-K partials double and the minibatch at reduction 1 takes 0.352 instead of 0.344 ms (same box),
            // with 192 or 384 each 0.354 / 0.358 ms
            constexpr int kPairTarget = 256;
            // (the Gram matrix is symmetric: only its tiles on and above the diagonal are computed, plan_dense `symmetric`)
            auto PG = plan_dense<T, EpiStore<T>>(Dg, Dg, k, k, s, epiG, ws_split + half, ws_elems - half, kPairTarget, 64, 64, 64, true);
            if (d.Dx_agg == MODL_AGG_AVERAGE) {
                EpiDxAverage<T> epi{Dx, static_cast<T *>(stt->d_Dx_average), d_idx, d_wsample, k, scale};
                auto PD = plan_dense<T, EpiDxAverage<T>>(A, B, b, k, Kdim, epi, ws_split, half, kPairTarget);
                if (PD.ok && PG.ok) {
                    MODL_TRY((launch_gemm_dense_pair<T, false, true, EpiDxAverage<T>, true, true, EpiStore<T>>(st, PD, PG,
                                                                                                                &ps.launches)));
                    paired = true;
                }
            } else {
                EpiStore<T> epi{Dx, k, scale};
                auto PD = plan_dense<T, EpiStore<T>>(A, B, b, k, Kdim, epi, ws_split, half, kPairTarget);
                if (PD.ok && PG.ok) {
                    MODL_TRY((launch_gemm_dense_pair<T, false, true, EpiStore<T>, true, true, EpiStore<T>>(st, PD, PG,
                                                                                                            &ps.launches)));
                    paired = true;
                }
            }
        }
        if (!paired) {
            if (d.Dx_agg == MODL_AGG_AVERAGE) {
                EpiDxAverage<T> epi{Dx, static_cast<T *>(stt->d_Dx_average), d_idx, d_wsample, k, scale};
                MODL_TRY((launch_gemm_dense<T, EpiDxAverage<T>>(st, A, B, b, k, Kdim, epi, sws, &ps.launches)));
            } else {
                EpiStore<T> epi{Dx, k, scale};
                MODL_TRY((launch_gemm_dense<T, EpiStore<T>>(st, A, B, b, k, Kdim, epi, sws, &ps.launches)));
            }
            if (want_G) {
                EpiStore<T> epi{Gbuf, k, red};
                MODL_TRY((launch_gemm_dense<T, EpiStore<T>>(st, Dg, Dg, k, k, s, epi, sws, &ps.launches)));
            }
        }
        if (d.G_agg == MODL_AGG_AVERAGE) {
            MODL_TRY(launch_update_G_average<T>(st, static_cast<T *>(stt->d_G_average), d_idx, Gbuf, d_wsample, b, k));
            ++ps.launches;
        }
    }
    {   // ---- code solve  (dict_fact.py:636-648)
        ProfScope ps(pl, st, SEC_CODE_SOLVE);
        if (d.G_agg == MODL_AGG_AVERAGE) {                            // per-sample Gram = rows idx of G_average_
            const T *Gavg = static_cast<const T *>(stt->d_G_average);
            MODL_TRY(solve_codes<T>(pl, st, Gavg, (int64_t)k * k, d_idx, Dx, xnorm, code, d_idx, b, d_sweeps,
                                    &ps.launches, H0, Fbuf, nullptr, nullptr, bt->h_sample_idx));
        } else {
            const T *G = (d.G_agg == MODL_AGG_FULL) ? static_cast<const T *>(stt->d_G) : Gbuf;
            if (cd_on_compact) {     // solve on the compact rows, the solver also writes code_[idx]
                MODL_TRY(solve_codes<T>(pl, st, G, 0, nullptr, Dx, xnorm, cb, nullptr, b, d_sweeps, &ps.launches, H0, Fbuf,
                                        d_idx ? code : nullptr, d_idx));
            } else {
                MODL_TRY(solve_codes<T>(pl, st, G, 0, nullptr, Dx, xnorm, code, d_idx, b, d_sweeps, &ps.launches, H0, Fbuf));
            }
        }
    }
    {   // ---- statistics: C_ and B_ updated in the epilogues of code^T code and X^T code, one launch for both products
        ProfScope ps(pl, st, SEC_STATS_GEMM);
        if (!cd_on_compact && d_idx) {
            hipLaunchKernelGGL((gather_rows_T_kernel<T, int64_t>), dim3((unsigned)b), dim3(256), 0, st, code, (int64_t)k,
                               d_idx, (int64_t)b, (int64_t)b, (int64_t)k, codeb, (int64_t)k);
            MODL_LAUNCH_CHECK();
            ++ps.launches;
        }
        DenseOperand Cd;
        Cd.ptr = cb; Cd.si = 1; Cd.sk = k;                              // element (i = atom, kk = sample)
        DenseOperand Xo;
        Xo.ptr = X; Xo.si = 1; Xo.sk = bt->ldx;                         // element (i = feature, kk = sample)
        DenseOperand Xso;
        Xso.ptr = Xsb; Xso.si = 1; Xso.sk = s_pad;                      // element (i = sampled feature, kk = sample)
        const int replace = d.optimizer == MODL_OPT_SGD;
        const T beta = (T)(1.0 - bt->w), wt = (T)bt->w, bdiv = (T)bt->b_global;
        T *Bt = static_cast<T *>(stt->d_Bt);
        // two-phase step: the head [ C | rows of Bt the dictionary update reads, compact ] is mirrored into `delta`
        T *mC = two_phase ? delta : nullptr, *mB = two_phase ? delta + (size_t)k * k : nullptr;
        EpiStats<T> eC{static_cast<T *>(stt->d_C), k, beta, wt, bdiv, replace, mC};
        if (ride) {
            // C_ and the SAMPLED rows of B_ now (one small paired launch over the gathered columns of X) ...
            EpiStatsRows<T> eBs{Bt, k, d_subset, beta, wt, bdiv, replace, mB};
            const bool gstamps = (d.flags & MODL_FLAG_GEMM_STAMPS) != 0;      // (diagnostics)
            MODL_TRY((stats_pair<T>(st, Cd, Cd, k, k, eC, Xso, Cd, s, k, eBs, b, sws, &ps.launches,
                                    gstamps ? reinterpret_cast<unsigned long long *>(pl->dws + pl->off_gstamps) : nullptr)));
            // ... the other rows while the dictionary update runs
            StatsRider &R = pl->rider;
            R.X = X; R.ldx = bt->ldx; R.code = cb; R.b = b; R.p = p; R.Bt = stt->d_Bt;
            R.stamp = reinterpret_cast<const int32_t *>(pl->dws + pl->off_stamp); R.step = pl->step_id;
            R.beta = (double)beta; R.wt = (double)wt; R.bdiv = (double)bdiv; R.replace = replace; R.consumed = 0;
            pl->ride_pending = true;
        } else {
            // every row of B_ in this launch; without a subset array the head carries all of them
            EpiStats<T> eB{Bt, k, beta, wt, bdiv, replace, d_subset ? nullptr : mB};
            MODL_TRY((stats_pair<T>(st, Cd, Cd, k, k, eC, Xo, Cd, p, k, eB, b, sws, &ps.launches)));
            if (two_phase && d_subset && s > 0) {
                hipLaunchKernelGGL((gather_rows_T_kernel<T, int32_t>), dim3((unsigned)s), dim3(256), 0, st, Bt, (int64_t)k,
                                   d_subset, s, s, (int64_t)k, mB, (int64_t)k);
                MODL_LAUNCH_CHECK();
                ++ps.launches;
            }
        }
        if (two_phase) {
            pl->head_pending = true;
            pl->head_elems = (int64_t)k * k + (d_subset ? s : p) * (int64_t)k;
        }
    }
    return MODL_OK;
}

// What the persistent dictionary-update launches of this plan have reported so far (pinned words, written by the device at
// system scope; a plain read, no synchronisation).  A launch that could not run - its workgroups were not resident together -
// and was completed by one workgroup (bcd_persist.hip: persist_recover) switches the plan to one launch per block for good:
// whatever held the compute units may still be there.  An update that gave up half-way is incomplete: nothing more is
// enqueued on this plan until modl_somf_status has reported it (not only when somebody synchronises).
static int persist_gate(modl_somf_plan *pl) {
    if (!pl->pflags) return MODL_OK;
    volatile unsigned int *f = pl->pflags;
    if (f[0] != 0) return MODL_ETIMEOUT;
    const unsigned int r = f[1];
    if (r != pl->recoveries_seen) {
        pl->recoveries_seen = r;
        pl->persist_ok = false;
    }
    return MODL_OK;
}

// The dictionary update (dict_fact.py:650-715).  After a two-phase phase 1, `head` = [ C | rows of B_ ] summed over
// the ranks: C is read from it, the rows of B_ are scattered into a plan-owned [p][k] array (the rank's own B_ keeps
// its partial sums) - without a subset array the head IS that array.
template <typename T>
int phase2(modl_somf_plan *pl, const modl_somf_state *stt, const modl_somf_batch *bt, const T *head, hipStream_t st,
           const modl_somf_batch *next = nullptr) {
    const modl_somf_desc &d = pl->d;
    const int k = d.k;
    const int64_t p = d.p, s = bt->s;
    if (!pl->staged) return MODL_ESTATE;
    if (!stt || !stt->d_Dt || !stt->d_Bt || !stt->d_C || !stt->d_comp_norm) return MODL_EINVAL;
    if (pl->head_pending && !head) return MODL_EINVAL;
    MODL_TRY(persist_gate(pl));
    char *P = pl->dws + pl->off_params;
    const int32_t *d_subset = pl->has_subset ? reinterpret_cast<const int32_t *>(P + pl->po_subset) : nullptr;
    const int32_t *d_order = reinterpret_cast<const int32_t *>(P + pl->po_order);
    T *Dt = static_cast<T *>(stt->d_Dt);
    T *Bt = static_cast<T *>(stt->d_Bt);
    const T *Bu = Bt, *Cu = static_cast<const T *>(stt->d_C);
    if (pl->head_pending) {
        ProfScope ps(pl, st, SEC_STATS_APPLY);
        Cu = head;
        Bu = head + (size_t)k * k;
        if (d_subset && s > 0) {
            if (!pl->Bsum) MODL_HIP(hipMalloc(&pl->Bsum, pl->tsz * (size_t)p * k));
            hipLaunchKernelGGL((scatter_rows_T_kernel<T, int32_t>), dim3((unsigned)s), dim3(256), 0, st,
                               static_cast<T *>(pl->Bsum), (int64_t)k, d_subset, s, (int64_t)k, head + (size_t)k * k, (int64_t)k);
            MODL_LAUNCH_CHECK();
            ps.launches += 1;
            Bu = static_cast<const T *>(pl->Bsum);
        }
        pl->head_pending = false;
    }
    {
        ProfScope ps(pl, st, SEC_DICT);
        const bool track_G = d.G_agg == MODL_AGG_FULL;
        const bool partial_G = track_G && (double)s < (double)p / 2.0;   // dict_fact.py:667, 711-715
        if (partial_G) {
            EpiAxpby<T> epi{static_cast<T *>(stt->d_G), k, (T)-1, (T)1};
            MODL_TRY((gram_of_rows<T, EpiAxpby<T>>(pl, st, Dt, d_subset, s, epi, &ps.launches)));
        }
        pl->last_s = s;
        DictUpdateArgs<T> a;
        a.Dt = Dt; a.Bt = Bu; a.C = Cu; a.comp_norm = static_cast<T *>(stt->d_comp_norm);
        a.subset = d_subset; a.order = d_order; a.h_order = pl->h_order_copy.data();
        a.s = s; a.k = k; a.optimizer = d.optimizer; a.comp_pos = d.comp_pos;
        a.comp_l1_ratio = d.comp_l1_ratio; a.w = bt->w; a.step_size = d.step_size;
        a.ws = pl->dws + pl->off_du; a.ws_bytes = pl->du_bytes;
        a.rider = pl->ride_pending ? &pl->rider : nullptr;
        a.level_hint = reinterpret_cast<double *>(pl->dws + pl->off_level);
        a.persist_flags = pl->pflags_dev;
        a.allow_persist = pl->persist_ok;
        // the next minibatch of the chunk loop: its pinned slot is filled now and the copy into the OTHER device block
        // rides on the update's last launch (or follows it as a launch of its own when that path has no riders); a
        // next minibatch that does not validate is simply staged - and reported - by its own step
        StageJob njob;
        const bool have_next = next && stage_fill<T>(pl, next, st, pl->cur_par ^ 1, njob) == MODL_OK;
        a.stage = have_next ? &njob.ride : nullptr;
        const int rc_du = dict_update<T>(st, a, &ps.launches);
        if (have_next) {
            if (!njob.ride.consumed) (void)stage_launch(njob.ride, st);     // (the slot is taken: its copy must run)
            pl->ahead = true;
            pl->ahead_has_idx = njob.has_idx;
            pl->ahead_has_subset = njob.has_subset;
            pl->ahead_order.swap(njob.order);
        }
        MODL_TRY(rc_du);
        if (pl->ride_pending && !pl->rider.consumed) {                 // this dictionary update has no fused path
            const StatsRider &R = pl->rider;
            DenseOperand Xo, Cd;
            Xo.ptr = R.X; Xo.si = 1; Xo.sk = R.ldx;
            Cd.ptr = R.code; Cd.si = 1; Cd.sk = k;
            EpiStatsSkip<T> epi{Bt, k, R.stamp, R.step, (T)R.beta, (T)R.wt, (T)R.bdiv, R.replace};
            SplitWs sws{pl->dws + pl->off_split, pl->split_bytes};
            MODL_TRY((stats_pair<T>(st, Cd, Cd, 0, 0, epi, Xo, Cd, p, k, epi, R.b, sws, &ps.launches)));
        }
        pl->ride_pending = false;
        if (track_G) {
            if (partial_G) {
                EpiAxpby<T> epi{static_cast<T *>(stt->d_G), k, (T)1, (T)1};
                MODL_TRY((gram_of_rows<T, EpiAxpby<T>>(pl, st, Dt, d_subset, s, epi, &ps.launches)));
            } else {
                EpiStore<T> epi{static_cast<T *>(stt->d_G), k, (T)1};
                MODL_TRY((gram_of_rows<T, EpiStore<T>>(pl, st, Dt, nullptr, p, epi, &ps.launches)));
            }
        }
    }
    return MODL_OK;
}

template <typename T>
int transform_impl(modl_somf_plan *pl, const T *Dt, const T *G, const T *X, int64_t ldx, int64_t n, T *code_out,
                   hipStream_t st) {
    const modl_somf_desc &d = pl->d;
    const int k = d.k;
    const int64_t p = d.p;
    T *xnorm = reinterpret_cast<T *>(pl->dws + pl->off_xnorm);
    T *Dx = reinterpret_cast<T *>(pl->dws + pl->off_Dx);
    T *H0 = reinterpret_cast<T *>(pl->dws + pl->off_H0);
    T *Gbuf = reinterpret_cast<T *>(pl->dws + pl->off_G);
    T *Fbuf = reinterpret_cast<T *>(pl->dws + pl->off_F);
    SplitWs sws{pl->dws + pl->off_split, pl->split_bytes};
    int nl = 0;
    if (!G) {                                                         // dict_fact.py:67-70
        EpiStore<T> epi{Gbuf, k, (T)1};
        MODL_TRY((gram_of_rows<T, EpiStore<T>>(pl, st, Dt, nullptr, p, epi, &nl)));
        G = Gbuf;
    }
    hipLaunchKernelGGL((fill_kernel<T>), dim3((unsigned)std::min<int64_t>(cdiv(n * k, 256), 2048)), dim3(256), 0, st,
                       code_out, n * k, (T)1);                        // code = ones (:72)
    MODL_LAUNCH_CHECK();
    for (int64_t r0 = 0; r0 < n; r0 += d.max_batch) {
        const int b = (int)std::min<int64_t>(d.max_batch, n - r0);
        const T *Xb = X + r0 * ldx;
        if (d.code_l1_ratio != 0.0) MODL_TRY(launch_row_norm2<T>(st, Xb, ldx, p, b, xnorm));
        DenseOperand A, B;
        A.ptr = Xb; A.si = ldx; A.sk = 1;
        B.ptr = Dt; B.si = 1; B.sk = k;
        EpiStore<T> epi{Dx, k, (T)1};
        MODL_TRY((launch_gemm_dense<T, EpiStore<T>>(st, A, B, b, k, p, epi, sws, &nl)));
        MODL_TRY(solve_codes<T>(pl, st, G, 0, nullptr, Dx, xnorm, code_out + r0 * k, nullptr, b, nullptr, &nl, H0, Fbuf));
    }
    return MODL_OK;
}

template <typename T>
int enet_regression_abi(const T *G, int64_t g_stride, T *Dx, const T *X, int64_t ldx, int64_t p, T *code,
                        const int64_t *d_indices, int64_t b, int64_t k, T l1_ratio, T alpha, int positive, T tol,
                        int max_iter, int32_t *d_sweeps, void *d_ws, size_t ws_bytes, void *stream) {
    if (!G || !Dx || !X || !code || b < 0 || k <= 0 || k > 1024 || p < 0 || ldx < p) return MODL_EINVAL;
    if (!(l1_ratio >= 0 && l1_ratio <= 1)) return MODL_EINVAL;
    if (b == 0) return MODL_OK;
    const size_t need = modl_enet_regression_workspace(DType<T>::id, b, k, g_stride != 0);
    if (!d_ws || ws_bytes < need) return MODL_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    char *w = static_cast<char *>(d_ws);
    T *xnorm = reinterpret_cast<T *>(w);
    T *H0 = reinterpret_cast<T *>(w + align_up(sizeof(T) * (size_t)b, 256));
    T *F = reinterpret_cast<T *>(w + align_up(sizeof(T) * (size_t)b, 256) + align_up(sizeof(T) * (size_t)b * k, 256));
    if (l1_ratio == 0 && chol_blocked((int)k, sizeof(T), g_stride == 0)) {
        T *Linv = F + (size_t)k * k;
        return ridge_solve_wide<T>(st, G, g_stride, nullptr, F, Linv, Dx, (int)b, (int)k, alpha, code, d_indices);
    }
    if (l1_ratio == 0) {
        const int nmat = g_stride ? (int)b : 1;
        if (ridge_small_applies<T>((int)k))
            return launch_ridge_small<T>(st, G, g_stride, nullptr, Dx, (int)b, (int)k, alpha, code, d_indices);
        MODL_TRY(launch_cholesky<T>(st, G, g_stride, nullptr, F, (int)k, alpha, nmat));
        return launch_chol_solve<T>(st, F, g_stride ? k * k : 0, Dx, (int)b, (int)k, code, d_indices);
    }
    MODL_TRY(launch_row_norm2<T>(st, X, ldx, p, b, xnorm));
    const T *H0p = nullptr;
    if (g_stride == 0) {
        Operand A, B;
        A.ptr = code; A.si = k; A.sk = 1; A.gi = gather64(d_indices);
        B.ptr = G; B.si = k; B.sk = 1;
        EpiStore<T> epi{H0, k, (T)1};
        SplitWs none;
        MODL_TRY((launch_gemm<T, EpiStore<T>>(st, A, B, b, k, k, epi, none, nullptr, 512, 1)));
        H0p = H0;
    }
    CdArgs<T> a;
    a.G = G; a.g_stride = g_stride; a.g_idx = nullptr; a.Dx = Dx; a.xnorm2 = xnorm; a.H0 = H0p; a.code = code;
    a.idx = d_indices;
    a.sweeps = d_sweeps; a.b = (int)b; a.k = (int)k;
    // any k on the vectorised kernel: padded copy at the end of the caller's workspace.  Also for k = 512 / 1024 when the
    // caller's matrix is not 16-byte aligned: beyond 256 coefficients only the four-wavefront solver exists in the product
    // build, and it wants aligned rows (such a call used to end in MODL_EINVAL)
    const bool g_misaligned = reinterpret_cast<uintptr_t>(G) % 16 != 0;
    if (!g_stride && k <= 1024 && (cd_padded_ld((int)k) != (int)k || (k > 256 && g_misaligned))) {
        const int kq = cd_padded_ld((int)k);
        const size_t gp_bytes = align_up(sizeof(T) * ((size_t)kq + 16) * kq, 256);
        T *Gp = reinterpret_cast<T *>(w + need - gp_bytes);
        MODL_HIP(hipMemsetAsync(Gp, 0, gp_bytes, st));
        MODL_TRY(launch_cd_pad_gram<T>(st, G, (int)k, Gp, kq));
        a.G = Gp; a.ldg = kq; a.g_pad_rows = 16;
    }
    a.alpha = alpha * l1_ratio;
    a.beta = (T)((double)alpha * (1.0 - (double)l1_ratio));
    a.tol = tol; a.max_iter = max_iter; a.positive = positive;
    if (g_stride && k >= 32 && k <= 1024 && (cd_split_enabled() || !cd_one_wave_covers((int)k)) && !cd_split_applies<T>(a)) {  // one Gram matrix per sample, any k: zero-padded slots
        const size_t slot_bytes = cd_per_sample_scratch_bytes(sizeof(T), b, (int)k);   // (at the end of the workspace)
        T *slots = reinterpret_cast<T *>(w + need - align_up(slot_bytes, 256));
        MODL_HIP(hipMemsetAsync(slots, 0, slot_bytes, st));
        return launch_cd_per_sample<T>(st, a, slots, slot_bytes);
    }
    return launch_cd<T>(st, a);
}

}  // namespace

extern "C" {

int modl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t modl_enet_regression_workspace(int dtype, int64_t b, int64_t k, int multi_gram) {
    const size_t t = dtype == MODL_F32 ? 4 : 8;
    if (b < 0 || k < 0) return 0;
    const size_t kq = (!multi_gram && k > 0 && k <= 1024) ? (size_t)modl::cd_padded_ld((int)k) : 0;   // padded shared Gram (cd_padded_ld)
    const size_t slots = (multi_gram && k >= 32 && k <= 1024) ? align_up(modl::cd_per_sample_scratch_bytes(t, b, (int)k), 256) : 0;
    return align_up(t * (size_t)b, 256) + align_up(t * (size_t)b * k, 256) +
           align_up(t * ((size_t)k * k * ((multi_gram && k <= 512) ? (size_t)b : 1) + modl::chol_wide_scratch_elems((int)k)), 256) +
           ((kq && (kq != (size_t)k || k > 256)) ? align_up(t * (kq + 16) * kq, 256) : 0) + slots;   // (k > 256: a misaligned matrix is copied too)
}

#define ABI_REG(SFX, T)                                                                                            \
    int modl_enet_regression_single_gram_##SFX(const T *d_G, T *d_Dx, const T *d_X, int64_t ldx, int64_t p,        \
                                               T *d_code, const int64_t *d_indices, int64_t b, int64_t k,          \
                                               T l1_ratio, T alpha, int positive, T tol, int max_iter,             \
                                               int32_t *d_sweeps, void *d_ws, size_t ws_bytes, void *stream) {     \
        return enet_regression_abi<T>(d_G, 0, d_Dx, d_X, ldx, p, d_code, d_indices, b, k, l1_ratio, alpha,          \
                                      positive, tol, max_iter, d_sweeps, d_ws, ws_bytes, stream);                  \
    }                                                                                                              \
    int modl_enet_regression_multi_gram_##SFX(const T *d_G, T *d_Dx, const T *d_X, int64_t ldx, int64_t p,         \
                                              T *d_code, const int64_t *d_indices, int64_t b, int64_t k,           \
                                              T l1_ratio, T alpha, int positive, T tol, int max_iter,              \
                                              int32_t *d_sweeps, void *d_ws, size_t ws_bytes, void *stream) {      \
        return enet_regression_abi<T>(d_G, k * k, d_Dx, d_X, ldx, p, d_code, d_indices, b, k, l1_ratio, alpha,      \
                                      positive, tol, max_iter, d_sweeps, d_ws, ws_bytes, stream);                  \
    }
ABI_REG(f32, float)
ABI_REG(f64, double)
#undef ABI_REG

int64_t modl_somf_delta_elems(const modl_somf_desc *desc) {
    if (!desc) return 0;
    return (int64_t)desc->k * desc->k + desc->p * (int64_t)desc->k;         // [C | rows of Bt, compact]
}

int modl_somf_plan_create(const modl_somf_desc *desc, modl_somf_plan **out) {
    if (!out) return MODL_EINVAL;
    *out = nullptr;
    MODL_TRY(validate_desc(desc));
    if (modl_device_count() <= 0) return MODL_ENOGPU;
    {   // the kernels are sized for gfx950's 160 KiB of LDS per compute unit (chol.hip, bcd.hip, cd_split.hip declare up to
        // that much): on a part with less they could not launch - refuse here, with a code, instead of failing later
        int dev = 0, lds = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds < 160 * 1024)
            return MODL_ENOGPU;
    }
    modl_somf_plan *pl = new (std::nothrow) modl_somf_plan();
    if (!pl) return MODL_ENOMEM;
    pl->d = *desc;
    (void)hipGetDevice(&pl->device);
    pl->tsz = desc->dtype == MODL_F32 ? 4 : 8;
    const size_t t = pl->tsz;
    const size_t k = (size_t)desc->k, b = (size_t)desc->max_batch, p = (size_t)desc->p;
    pl->params_bytes = params_layout(pl);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    pl->off_params2[0] = take(pl->params_bytes);
    pl->off_params2[1] = take(pl->params_bytes);
    pl->off_params = pl->off_params2[0];
    pl->off_xnorm = take(t * b);
    pl->off_sweeps = take(sizeof(int32_t) * b);
    pl->off_Dx = take(t * b * k);
    pl->off_H0 = take(t * b * k);
    pl->off_G = take(t * (k + 16) * k);        // 16 readable rows behind the Gram: the solver's row prefetch runs unclamped
    if (desc->code_l1_ratio != 0.0 && cd_padded_ld(desc->k) != desc->k) {
        pl->ld_gpad = cd_padded_ld(desc->k);
        pl->off_Gpad = take(t * ((size_t)pl->ld_gpad + 16) * pl->ld_gpad);
    }
    const size_t p_pad = align_up(p, 4);
    pl->off_Ds = take(t * p_pad * k);          // compacted sampled dictionary rows
    pl->off_Xs = take(t * b * p_pad);          // compacted sampled minibatch columns
    pl->off_codeb = take(t * b * k);           // the minibatch's code rows
    pl->off_stamp = take(sizeof(int32_t) * p); // stamp[f] = last minibatch that sampled feature f
    pl->off_pos = take(sizeof(int32_t) * p);   // ... and its index in that minibatch's subset
    pl->off_gstamps = take(sizeof(unsigned long long) * 8);   // diagnostics (modl_somf_debug_gemm_stamps)
    const bool per_sample = desc->G_agg == MODL_AGG_AVERAGE;
    pl->off_F = take(t * k * k * ((per_sample && k <= 512) ? b : 1)); // Cholesky factors (one per sample for G_average_)
    pl->off_Linv = take(t * chol_wide_scratch_elems(desc->k));        // inverted diagonal blocks (wide ridge systems)
    // split-K partial tiles: the largest split product is max(b, k) x k (Dx, Gram, C increment) with
    // up to 64 splits; the p x k product (B increment) has >= 512 tiles at p >= 8k and is not split
    pl->split_bytes = t * std::max(b, k) * k * 64;
    if (p * k < std::max(b, k) * k * 64) pl->split_bytes = std::max(pl->split_bytes, t * p * k * 8);
    pl->off_split = take(pl->split_bytes);
    pl->du_bytes = dict_update_workspace(desc->dtype, (int64_t)p, desc->k);
    pl->off_du = take(pl->du_bytes);
    pl->off_level = take(sizeof(double) * k + 64);   // per-atom projection levels (warm start of the next projection)
    pl->dws_bytes = o;
    hipError_t e = hipMalloc((void **)&pl->dws, pl->dws_bytes);
    if (e != hipSuccess) { delete pl; return (int)e; }
    e = hipMemset(pl->dws + pl->off_stamp, 0, sizeof(int32_t) * p);
    if (e == hipSuccess) e = hipMemset(pl->dws + pl->off_level, 0, sizeof(double) * k + 64);
    if (e == hipSuccess && pl->ld_gpad) e = hipMemset(pl->dws + pl->off_Gpad, 0, t * ((size_t)pl->ld_gpad + 16) * pl->ld_gpad);
    if (e != hipSuccess) { modl_somf_plan_destroy(pl); return (int)e; }
    e = hipHostMalloc((void **)&pl->pflags, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&pl->pflags_dev, pl->pflags, 0);
    if (e != hipSuccess) { modl_somf_plan_destroy(pl); return (int)e; }
    std::memset(pl->pflags, 0, 64);
    for (int i = 0; i < kStageSlots; ++i) {
        e = hipHostMalloc((void **)&pl->hstage[i], align_up(pl->params_bytes, 16) + 64, hipHostMallocMapped);   // + acknowledgement word
        if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&pl->hstage_dev[i], pl->hstage[i], 0);
        if (e == hipSuccess) std::memset(pl->hstage[i] + align_up(pl->params_bytes, 16), 0, 64);
        if (e != hipSuccess) { modl_somf_plan_destroy(pl); return (int)e; }
    }
    *out = pl;
    return MODL_OK;
}

void modl_somf_plan_destroy(modl_somf_plan *pl) {
    if (!pl) return;
    if (pl->dws) (void)hipFree(pl->dws);
    if (pl->Bsum) (void)hipFree(pl->Bsum);
    if (pl->Gslots) (void)hipFree(pl->Gslots);
    if (pl->own_head) (void)hipFree(pl->own_head);
    if (pl->pflags) (void)hipHostFree(pl->pflags);
    for (int i = 0; i < kStageSlots; ++i) {
        if (pl->hstage[i]) (void)hipHostFree(pl->hstage[i]);
    }
    for (hipEvent_t ev : pl->pev) (void)hipEventDestroy(ev);
    delete pl;
}

int modl_somf_plan_update(modl_somf_plan *pl, const modl_somf_desc *desc) {
    if (!pl) return MODL_EINVAL;
    MODL_TRY(validate_desc(desc));
    const modl_somf_desc &o = pl->d;
    if (desc->dtype != o.dtype || desc->k != o.k || desc->p != o.p || desc->n_samples != o.n_samples ||
        desc->max_batch != o.max_batch)
        return MODL_EINVAL;
    if (desc->G_agg == MODL_AGG_AVERAGE && o.G_agg != MODL_AGG_AVERAGE) return MODL_EINVAL;   // scratch was not sized for it
    pl->d = *desc;
    return MODL_OK;
}

#define DISPATCH(pl, CALLF, CALLD) ((pl)->d.dtype == MODL_F32 ? (CALLF) : (CALLD))

// makes the plan's device current for the duration of an entry point (a caller may hold several estimators on
// several devices; launching into another device's stream fails with "invalid resource handle")
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(const modl_somf_plan *pl) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != pl->device && hipSetDevice(pl->device) == hipSuccess) prev = cur;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int modl_somf_code_and_partials(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt, void *d_head,
                                void *stream) {
    if (!pl || !bt) return MODL_EINVAL;
    DeviceScope dev(pl);
    pl->ahead = false;
    MODL_TRY(persist_gate(pl));
    return DISPATCH(pl, phase1<float>(pl, st, bt, static_cast<float *>(d_head), (hipStream_t)stream, true),
                    phase1<double>(pl, st, bt, static_cast<double *>(d_head), (hipStream_t)stream, true));
}

int modl_somf_head_elems(const modl_somf_plan *pl, int64_t *head_elems) {
    if (!pl || !head_elems) return MODL_EINVAL;
    if (!pl->head_pending) return MODL_ESTATE;
    *head_elems = pl->head_elems;
    return MODL_OK;
}

int modl_somf_apply_and_update_dict(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt,
                                    const void *d_head, void *stream) {
    if (!pl || !bt) return MODL_EINVAL;
    DeviceScope dev(pl);
    return DISPATCH(pl, phase2<float>(pl, st, bt, static_cast<const float *>(d_head), (hipStream_t)stream),
                    phase2<double>(pl, st, bt, static_cast<const double *>(d_head), (hipStream_t)stream));
}

// one GPU: both phases back to back, nothing travels through a head buffer; `next` (chunk loop only): staged ahead
static int somf_step_next(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt,
                          const modl_somf_batch *next, void *stream) {
    DeviceScope dev(pl);
    MODL_TRY(persist_gate(pl));
    MODL_TRY(DISPATCH(pl, phase1<float>(pl, st, bt, nullptr, (hipStream_t)stream, false),
                      phase1<double>(pl, st, bt, nullptr, (hipStream_t)stream, false)));
    return DISPATCH(pl, phase2<float>(pl, st, bt, nullptr, (hipStream_t)stream, next),
                    phase2<double>(pl, st, bt, nullptr, (hipStream_t)stream, next));
}

int modl_somf_step(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt, void *stream) {
    if (!pl || !bt) return MODL_EINVAL;
    pl->ahead = false;                              // (a minibatch staged ahead belongs to an abandoned chunk loop)
    return somf_step_next(pl, st, bt, nullptr, stream);
}

}  // extern "C"

// ---- the collective inside the library: RCCL through its C API, resolved at run time -------------------------------
// (librccl.so is looked up with dlopen: a process that already holds RCCL - e.g. through torch.distributed - gets that
// same copy; nothing links against it, so the library loads and every single-GPU entry point works without RCCL)
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, modl_comm_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommAbort)(void *) = nullptr;                  // optional (the abort path: modl_comm_abort / modl_comm_wait)
    int (*CommGetAsyncError)(void *, int *) = nullptr;   // optional
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) return a;
        a.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<int (*)(void **, int, modl_comm_id, int)>(dlsym(a.handle, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(a.handle, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(
            dlsym(a.handle, "ncclAllReduce"));
        a.CommAbort = reinterpret_cast<int (*)(void *)>(dlsym(a.handle, "ncclCommAbort"));
        a.CommGetAsyncError = reinterpret_cast<int (*)(void *, int *)>(dlsym(a.handle, "ncclCommGetAsyncError"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
        return a;
    }();
    return api;
}
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;      // rccl.h: ncclRedOp_t / ncclDataType_t
constexpr int kNcclSuccess = 0, kNcclInProgress = 7;                 // rccl.h: ncclResult_t
}  // namespace

extern "C" {

struct modl_comm {
    void *nccl = nullptr;
    int rank = 0, world = 1;
    bool dead = false;                 // aborted (an RCCL error, a peer that never arrived): every later call is MODL_ERCCL
};

// an asynchronous RCCL error on this communicator (a peer died, a link failed)?  Aborts it then: the kernels of this
// rank that wait inside a collective are released, the stream drains, and the caller sees MODL_ERCCL instead of a hang.
static int comm_poll(modl_comm *c) {
    if (c->dead) return MODL_ERCCL;
    int err = kNcclSuccess;
    if (rccl().CommGetAsyncError && rccl().CommGetAsyncError(c->nccl, &err) == 0 && err != kNcclSuccess && err != kNcclInProgress) {
        (void)modl_comm_abort(c);
        return MODL_ERCCL;
    }
    return MODL_OK;
}

int modl_comm_unique_id(modl_comm_id *out) {
    if (!out) return MODL_EINVAL;
    if (!rccl().ok) return MODL_ENORCCL;
    return rccl().GetUniqueId(out) == 0 ? MODL_OK : MODL_ERCCL;
}

int modl_comm_create(const modl_comm_id *id, int rank, int world, modl_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return MODL_EINVAL;
    *out = nullptr;
    if (!rccl().ok) return MODL_ENORCCL;
    modl_comm *c = new (std::nothrow) modl_comm();
    if (!c) return MODL_ENOMEM;
    c->rank = rank; c->world = world;
    if (rccl().CommInitRank(&c->nccl, world, *id, rank) != 0) { delete c; return MODL_ERCCL; }
    *out = c;
    return MODL_OK;
}

void modl_comm_destroy(modl_comm *c) {
    if (!c) return;
    if (c->nccl && rccl().ok && !c->dead) (void)rccl().CommDestroy(c->nccl);      // (an aborted communicator is gone already)
    delete c;
}

int modl_comm_abort(modl_comm *c) {
    if (!c) return MODL_EINVAL;
    if (!c->dead) {
        c->dead = true;
        if (c->nccl && rccl().CommAbort) (void)rccl().CommAbort(c->nccl);          // frees the communicator
        c->nccl = nullptr;
    }
    return MODL_OK;
}

int modl_comm_wait(modl_comm *c, void *stream, double timeout_s) {
    if (!c) return MODL_EINVAL;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery((hipStream_t)stream);
        if (q == hipSuccess) return c->dead ? MODL_ERCCL : MODL_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); (void)modl_comm_abort(c); return (int)q; }   // (a HIP error code, as everywhere)
        if (comm_poll(c) != MODL_OK) return MODL_ERCCL;
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_s > 0 && waited > timeout_s) {     // a rank that never arrives: do not wait for it for ever
            (void)modl_comm_abort(c);
            return MODL_ERCCL;
        }
        if (++spins < 2000) sched_yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.05 ? 20 : 500));
    }
}

int modl_comm_all_reduce_sum(modl_comm *c, void *d_buf, int64_t n, int dtype, void *stream) {
    if (!c || !d_buf || n < 0 || (dtype != MODL_F32 && dtype != MODL_F64)) return MODL_EINVAL;
    if (c->dead || !c->nccl) return c->dead ? MODL_ERCCL : MODL_EINVAL;
    if (n == 0) return MODL_OK;
    if (rccl().AllReduce(d_buf, d_buf, (size_t)n, dtype == MODL_F32 ? kNcclFloat32 : kNcclFloat64, kNcclSum, c->nccl,
                         (hipStream_t)stream) != 0) {
        (void)modl_comm_abort(c);              // whatever this rank has enqueued on the communicator is released
        return MODL_ERCCL;
    }
    return MODL_OK;
}

// several GPUs, everything on ONE stream: phase 1 (partial statistics + head), ncclAllReduce of the head in place,
// phase 2 (dictionary update from the summed head) - no cross-stream event anywhere; `next` (chunk loop only): the
// following minibatch, staged ahead by phase 2's last launch exactly as in the one-GPU step
static int somf_step_dist_next(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt,
                               const modl_somf_batch *next, modl_comm *comm, void *stream) {
    DeviceScope dev(pl);
    if (!pl->own_head) {
        const size_t n = (size_t)pl->d.k * pl->d.k + (size_t)pl->d.p * pl->d.k;
        MODL_HIP(hipMalloc(&pl->own_head, pl->tsz * n));
    }
    MODL_TRY(DISPATCH(pl, phase1<float>(pl, st, bt, static_cast<float *>(pl->own_head), (hipStream_t)stream, true),
                      phase1<double>(pl, st, bt, static_cast<double *>(pl->own_head), (hipStream_t)stream, true)));
    MODL_TRY(modl_comm_all_reduce_sum(comm, pl->own_head, pl->head_elems, pl->d.dtype, stream));
    return DISPATCH(pl, phase2<float>(pl, st, bt, static_cast<const float *>(pl->own_head), (hipStream_t)stream, next),
                    phase2<double>(pl, st, bt, static_cast<const double *>(pl->own_head), (hipStream_t)stream, next));
}

int modl_somf_step_dist(modl_somf_plan *pl, const modl_somf_state *st, const modl_somf_batch *bt, modl_comm *comm,
                        void *stream) {
    if (!pl || !bt || !comm) return MODL_EINVAL;
    pl->ahead = false;                              // (a minibatch staged ahead belongs to an abandoned chunk loop)
    return somf_step_dist_next(pl, st, bt, nullptr, comm, stream);
}

// The feature subsets of a chunk call drawn AHEAD on a worker thread (same sampler, same order of draws: the same
// streams bit for bit).  The bit-exact shuffle of p indices costs ~5 ns per feature - 1 ms at p = 200 000, as long as
// the device step there - so beyond kDrawAheadFeatures features the chunk call would be host-bound with the draw on the
// calling thread (round 4 sent such shapes back to the Python loop and its look-ahead thread: one library call per
// minibatch, and with several ranks one modl_somf_step_dist call each).  A ring of NQ buffers; minibatch u is drawn
// once minibatch u - NQ has been enqueued.  Only this thread touches the sampler while it runs; stop() joins it before
// anybody else does (the rewind of a failed call).
namespace {
constexpr int64_t kDrawAheadFeatures = 32768;
struct DrawAhead {
    static constexpr int NQ = 4;
    modl_sampler *sampler = nullptr;
    double reduction = 1.0;
    int64_t nb = 0;
    std::vector<int64_t> buf[NQ];
    int64_t len[NQ] = {0, 0, 0, 0};
    int rcs[NQ] = {0, 0, 0, 0};
    std::mutex m;
    std::condition_variable cv;
    int64_t drawn = 0, released = 0;
    bool quit = false;
    std::thread th;
    void start(modl_sampler *s, double red, int64_t p, int64_t n_batches) {
        sampler = s; reduction = red; nb = n_batches;
        for (auto &b : buf) b.resize((size_t)std::max<int64_t>(p, 1));
        th = std::thread([this] { run(); });
    }
    void run() {
        for (int64_t u = 0; u < nb; ++u) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || u - NQ < released; });
                if (quit) return;
            }
            int64_t s = 0;
            const int rc = modl_sampler_yield_subset(sampler, reduction, buf[u % NQ].data(), &s);
            {
                std::lock_guard<std::mutex> lk(m);
                len[u % NQ] = s; rcs[u % NQ] = rc; drawn = u + 1;
            }
            cv.notify_all();
            if (rc != MODL_OK) return;
        }
    }
    int take(int64_t u, const int64_t **ptr, int64_t *s) {       // blocks until minibatch u's subset is drawn
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return drawn > u; });
        *ptr = buf[u % NQ].data(); *s = len[u % NQ];
        return rcs[u % NQ];
    }
    void release(int64_t upto) {                                  // the buffers of minibatches < upto may be reused
        { std::lock_guard<std::mutex> lk(m); if (upto > released) released = upto; }
        cv.notify_all();
    }
    void stop() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        th.join();
    }
    ~DrawAhead() { stop(); }
};
}  // namespace

int modl_somf_partial_fit_chunk(modl_somf_plan *pl, const modl_somf_state *st, const void *d_X, int64_t ldx, int64_t n_rows,
                                int32_t batch_size, const int64_t *h_sample_idx, modl_sampler *sampler, modl_rk *order_rng,
                                int64_t *n_iter, double learning_rate, double reduction, const int64_t *h_b_global,
                                modl_comm *comm, int64_t *n_done, void *stream) {
    if (n_done) *n_done = 0;
    if (!pl || !st || !d_X || !sampler || !order_rng || !n_iter || n_rows < 0 || batch_size <= 0) return MODL_EINVAL;
    const modl_somf_desc &d = pl->d;
    if (batch_size > d.max_batch || ldx < d.p || !(reduction >= 1.0)) return MODL_EINVAL;
    if (d.G_agg == MODL_AGG_AVERAGE || d.Dx_agg == MODL_AGG_AVERAGE) return MODL_EINVAL;   // (needs the caller's w_sample)
    // One minibatch of look-ahead: the arrays of minibatch t + 1 are drawn (same generators, same order of draws) before
    // step t is enqueued, so that step t can stage them - their copy to HBM rides on its last launch.
    struct Prepared {
        std::vector<int64_t> subset, order, idx_local;
        modl_somf_batch bt;
    };
    Prepared prep[2];
    const int64_t n_batches = (n_rows + batch_size - 1) / batch_size;
    const bool draw_ahead = d.p >= kDrawAheadFeatures && n_batches >= 4;
    DrawAhead ahead;                               // (declared before anything that can return: its destructor joins)
    for (auto &q : prep) { q.subset.resize((size_t)(draw_ahead ? 1 : std::max<int64_t>(d.p, 1))); q.order.resize((size_t)d.k); }
    const char *X = static_cast<const char *>(d_X);
    // what a failed minibatch is rewound to: the state of both generators and of the counter at the start of the call
    // (the look-ahead has drawn for minibatch t + 1 when step t fails; a retry must continue the reference's streams
    // where the last FITTED minibatch left them, dict_fact.py:507,672)
    const int64_t n_iter0 = *n_iter;
    std::vector<char> sampler0(modl_sampler_state_bytes(sampler));
    MODL_TRY(modl_sampler_get_state(sampler, sampler0.data(), sampler0.size()));
    uint32_t order_key0[624];
    int32_t order_pos0 = 0;
    MODL_TRY(modl_rk_get_mt_state(order_rng, order_key0, &order_pos0));
    auto global_rows = [&](int64_t t, int64_t r0) -> int64_t {
        return h_b_global ? h_b_global[t] : std::min<int64_t>(batch_size, n_rows - r0);
    };
    auto fail = [&](int64_t done, int rc) -> int {
        pl->ahead = false;
        ahead.stop();                              // (nobody else draws from the sampler while it is rewound)
        if (n_done) *n_done = done;
        std::vector<int64_t> scratch((size_t)std::max<int64_t>(d.p, 1));
        // replay exactly the draws of the `done` fitted minibatches (the two generators are independent streams)
        int64_t s = 0;
        if (modl_sampler_set_state(sampler, sampler0.data(), sampler0.size()) != MODL_OK ||
            modl_rk_set_mt_state(order_rng, order_key0, order_pos0) != MODL_OK) return rc;
        int64_t it = n_iter0;
        for (int64_t t = 0; t < done; ++t) {
            (void)modl_sampler_yield_subset(sampler, reduction, scratch.data(), &s);
            (void)modl_rk_permutation(order_rng, d.k, prep[0].order.data());
            it += global_rows(t, t * (int64_t)batch_size);
        }
        *n_iter = it;
        return rc;
    };
    auto prepare = [&](int64_t t, int64_t r0, Prepared &q) -> int {
        const int32_t b = (int32_t)std::min<int64_t>(batch_size, n_rows - r0);
        int64_t s = 0;
        const int64_t *subset_ptr = q.subset.data();
        if (draw_ahead) MODL_TRY(ahead.take(t, &subset_ptr, &s));                           // dict_fact.py:507, drawn ahead
        else MODL_TRY(modl_sampler_yield_subset(sampler, reduction, q.subset.data(), &s));  // dict_fact.py:507
        const int64_t bg = global_rows(t, r0);
        if (bg < b) return MODL_EINVAL;
        *n_iter += bg;                                                                      // :510
        double w = 0;
        MODL_TRY(modl_batch_weight(*n_iter, bg, learning_rate, 0.0, &w));                   // :515
        MODL_TRY(modl_rk_permutation(order_rng, d.k, q.order.data()));                      // :672
        modl_somf_batch &bt = q.bt;
        std::memset(&bt, 0, sizeof(bt));
        bt.d_X = X + (size_t)r0 * (size_t)ldx * pl->tsz;
        bt.ldx = ldx;
        bt.b = b;
        bt.h_sample_idx = h_sample_idx ? h_sample_idx + r0 : nullptr;
        if (!h_sample_idx) {                       // rows r0 .. r0 + b - 1 of code_ (a later chunk of the same call)
            q.idx_local.resize((size_t)b);
            for (int32_t i = 0; i < b; ++i) q.idx_local[(size_t)i] = r0 + i;
            bt.h_sample_idx = q.idx_local.data();
        }
        if (s == d.p) { bt.s = (int32_t)d.p; bt.h_subset = nullptr; }   // every feature: no gather (any order is the same set)
        else { bt.s = (int32_t)s; bt.h_subset = subset_ptr; }
        bt.h_order = q.order.data();
        bt.w = w;
        bt.reduction = reduction;
        bt.b_global = bg;
        return MODL_OK;
    };
    pl->ahead = false;
    if (n_rows <= 0) return MODL_OK;
    if (draw_ahead) ahead.start(sampler, reduction, d.p, n_batches);
    { const int rc0 = prepare(0, 0, prep[0]); if (rc0 != MODL_OK) return fail(0, rc0); }
    int64_t t = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += batch_size, ++t) {
        Prepared &cur = prep[t & 1];
        const bool more = r0 + batch_size < n_rows;
        int rc_next = MODL_OK;
        if (more) rc_next = prepare(t + 1, r0 + batch_size, prep[(t + 1) & 1]);
        const modl_somf_batch *next = (more && rc_next == MODL_OK && modl::g_stage_ahead.load(std::memory_order_relaxed))
                                          ? &prep[(t + 1) & 1].bt : nullptr;
        const int rc = comm ? somf_step_dist_next(pl, st, &cur.bt, next, comm, stream)
                            : somf_step_next(pl, st, &cur.bt, next, stream);
        if (rc != MODL_OK) return fail(t, rc);
        if (rc_next != MODL_OK) return fail(t + 1, rc_next);
        if (draw_ahead) ahead.release(t + 1);      // (minibatch t's arrays were copied to the staging ring when it was enqueued)
    }
    ahead.stop();
    pl->ahead = false;
    if (n_done) *n_done = t;
    return MODL_OK;
}

int modl_somf_full_gram(modl_somf_plan *pl, const void *d_Dt, void *d_G, void *stream) {
    if (!pl || !d_Dt || !d_G) return MODL_EINVAL;
    DeviceScope dev(pl);
    int nl = 0;
    if (pl->d.dtype == MODL_F32) {
        EpiStore<float> epi{static_cast<float *>(d_G), pl->d.k, 1.0f};
        return gram_of_rows<float, EpiStore<float>>(pl, (hipStream_t)stream, static_cast<const float *>(d_Dt), nullptr,
                                                    pl->d.p, epi, &nl);
    }
    EpiStore<double> epi{static_cast<double *>(d_G), pl->d.k, 1.0};
    return gram_of_rows<double, EpiStore<double>>(pl, (hipStream_t)stream, static_cast<const double *>(d_Dt), nullptr,
                                                  pl->d.p, epi, &nl);
}

int modl_somf_transform(modl_somf_plan *pl, const void *d_Dt, const void *d_G, const void *d_X, int64_t ldx, int64_t n,
                        void *d_code_out, void *stream) {
    if (!pl || !d_Dt || !d_X || !d_code_out || n < 0 || ldx < pl->d.p) return MODL_EINVAL;
    if (n == 0) return MODL_OK;
    DeviceScope dev(pl);
    return DISPATCH(pl,
                    transform_impl<float>(pl, static_cast<const float *>(d_Dt), static_cast<const float *>(d_G),
                                          static_cast<const float *>(d_X), ldx, n, static_cast<float *>(d_code_out),
                                          (hipStream_t)stream),
                    transform_impl<double>(pl, static_cast<const double *>(d_Dt), static_cast<const double *>(d_G),
                                           static_cast<const double *>(d_X), ldx, n, static_cast<double *>(d_code_out),
                                           (hipStream_t)stream));
}

// rows d_idx[0..n) of a row-major device matrix, gathered into d_dst (the row permutations of fit / shuffle,
// dict_fact.py:309-310, fmri.py:541: data movement only, HBM-bound)
#define ABI_GATHER(SFX, T)                                                                                         \
    int modl_gather_rows_##SFX(const T *d_src, int64_t src_ld, const int64_t *d_idx, int64_t n_rows, int64_t cols,  \
                               T *d_dst, int64_t dst_ld, void *stream) {                                           \
        if (!d_src || !d_idx || !d_dst || n_rows < 0 || cols < 0 || src_ld < cols || dst_ld < cols) return MODL_EINVAL; \
        if (n_rows == 0 || cols == 0) return MODL_OK;                                                              \
        if (n_rows > 0x7fffffff) return MODL_EINVAL;                                                               \
        hipLaunchKernelGGL((gather_rows_T_kernel<T, int64_t>), dim3((unsigned)n_rows), dim3(256), 0,              \
                           (hipStream_t)stream, d_src, src_ld, d_idx, n_rows, n_rows, cols, d_dst, dst_ld);        \
        MODL_LAUNCH_CHECK();                                                                                       \
        return MODL_OK;                                                                                            \
    }
ABI_GATHER(f32, float)
ABI_GATHER(f64, double)
#undef ABI_GATHER

int modl_somf_debug_stamps(modl_somf_plan *pl, unsigned long long *h_out) {
    // diagnostics only: phase timestamps (shader clock) left by the last fused dictionary-update launch
    if (!pl || !h_out) return MODL_EINVAL;
    MODL_HIP(hipDeviceSynchronize());
    const size_t off = modl::dict_update_stamps_offset(pl->d.dtype, pl->last_s, pl->d.k);
    MODL_HIP(hipMemcpy(h_out, pl->dws + pl->off_du + off, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MODL_OK;
}

int modl_somf_status(modl_somf_plan *pl, void *stream) {
    if (!pl) return MODL_EINVAL;
    DeviceScope dev(pl);
    // The wait: the stream is POLLED for the first 20 ms, then the thread blocks.  A blocking synchronisation returns
    // 50-100 us after the stream has drained on this stack (an interrupt and a wake-up) - 2 % of a partial_fit call of 20
    // minibatches; hipStreamQuery sees it within a few microseconds.  (The words are in pinned host memory, written by the
    // device at system scope: nothing is copied.)
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery((hipStream_t)stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) MODL_HIP(q);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            MODL_HIP(hipStreamSynchronize((hipStream_t)stream));
            break;
        }
    }
    (void)persist_gate(pl);                                   // (a recovery seen now keeps the next enqueue off the persistent launch)
    volatile unsigned int *f = pl->pflags;
    if (!f || f[0] == 0) return MODL_OK;
    f[0] = 0;                                                 // (the stream is idle: nobody else writes)
    pl->persist_ok = false;
    return MODL_ETIMEOUT;
}

int modl_somf_persist_recoveries(modl_somf_plan *pl, int64_t *count) {
    if (!pl || !count) return MODL_EINVAL;
    volatile unsigned int *f = pl->pflags;
    *count = f ? (int64_t)f[1] : 0;
    return MODL_OK;
}

int modl_somf_debug_persist_stamps(modl_somf_plan *pl, unsigned long long *h_out) {
    // diagnostics only: the stamps the resolver ([0, 96)) and row workgroup 0 ([96, 192)) of the last persistent
    // dictionary-update launch left (written by the diagnostics build only)
    if (!pl || !h_out) return MODL_EINVAL;
    MODL_HIP(hipDeviceSynchronize());
    const size_t off = modl::dict_update_persist_stamps_offset(pl->d.dtype, pl->last_s, pl->d.k);
    MODL_HIP(hipMemcpy(h_out, pl->dws + pl->off_du + off, 192 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MODL_OK;
}

int modl_somf_debug_gemm_stamps(modl_somf_plan *pl, unsigned long long *h_out) {
    // diagnostics only (MODL_GEMM_STAMPS=1): stamps of the last tile of the head product of the single-GPU step
    if (!pl || !h_out) return MODL_EINVAL;
    MODL_HIP(hipDeviceSynchronize());
    MODL_HIP(hipMemcpy(h_out, pl->dws + pl->off_gstamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MODL_OK;
}

int modl_somf_last_sweeps(modl_somf_plan *pl, int32_t *h_out, int cap, int *n_out, void *stream) {
    if (!pl || !h_out || !n_out) return MODL_EINVAL;
    const int n = pl->last_b < cap ? pl->last_b : cap;
    *n_out = n;
    if (n <= 0) return MODL_OK;
    const void *src = pl->last_sweeps_ptr ? (const void *)pl->last_sweeps_ptr : (const void *)(pl->dws + pl->off_sweeps);
    MODL_HIP(hipMemcpyAsync(h_out, src, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost,
                            (hipStream_t)stream));
    MODL_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MODL_OK;
}

int modl_somf_sweeps_history(modl_somf_plan *pl, int32_t *d_buf, int64_t cap_minibatches) {
    if (!pl || (d_buf && cap_minibatches <= 0)) return MODL_EINVAL;
    pl->hist = d_buf;
    pl->hist_cap = d_buf ? cap_minibatches : 0;
    pl->hist_n = 0;
    pl->last_sweeps_ptr = nullptr;
    return MODL_OK;
}

int modl_somf_prof_enable(modl_somf_plan *pl, int enable) {
    if (!pl) return MODL_EINVAL;
    if (enable && pl->pev.empty()) {
        pl->pev.resize(2 * kProfPool, nullptr);
        pl->psec.resize(kProfPool);
        pl->plaunch.resize(kProfPool);
        for (auto &ev : pl->pev) MODL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));   // device-side timing only
    }
    if (!enable && pl->prof) MODL_TRY(prof_flush(pl));
    pl->prof = enable != 0;
    pl->prof_mask = (enable == 1 || enable == 0) ? ~0u : ((unsigned)enable >> 1);
    return MODL_OK;
}

int modl_somf_host_wait_ms(modl_somf_plan *pl, double *out, int reset) {
    if (!pl || !out) return MODL_EINVAL;
    *out = pl->wait_ms;
    if (reset) pl->wait_ms = 0;
    return MODL_OK;
}

int modl_somf_prof_stride(modl_somf_plan *pl, int every) {
    if (!pl || every < 1) return MODL_EINVAL;
    pl->prof_stride = every;
    return MODL_OK;
}

int modl_somf_prof_get(modl_somf_plan *pl, modl_prof_entry *out, int cap, int *n_out) {
    if (!pl || !out || !n_out) return MODL_EINVAL;
    MODL_TRY(prof_flush(pl));
    int n = 0;
    for (int s = 0; s < SEC_COUNT && n < cap; ++s, ++n) {
        out[n].name = kSectionNames[s];
        out[n].ms_total = pl->ms[s];
        out[n].launches = pl->launches[s];
        out[n].calls = pl->calls[s];
    }
    *n_out = n;
    return MODL_OK;
}

int modl_somf_prof_reset(modl_somf_plan *pl) {
    if (!pl) return MODL_EINVAL;
    MODL_TRY(prof_flush(pl));
    for (int s = 0; s < SEC_COUNT; ++s) { pl->ms[s] = 0; pl->launches[s] = 0; pl->calls[s] = 0; }
    return MODL_OK;
}

}  // extern "C"
