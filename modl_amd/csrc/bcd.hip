// Block-coordinate dictionary update on the sampled feature rows.
//
// Replaces DictFact._update_dict (reference: modl/decomposition/dict_fact.py:650-715;
// also the masked variant modl/decomposition/recsys.py:187-213).  The reference
// sweeps the k atoms one after the other in a random order; each step is two
// rank-1 (ger) passes over a k x s gradient plus one projection of the atom onto
// its elastic-net ball, i.e. one global reduction over the s sampled features
// per atom — k strictly sequential, bandwidth-bound steps.
//
// Three device paths, all exact restatements of that sweep (only float
// summation order differs):
//
//  * blocked path  (comp_l1_ratio == 0, no positivity: DictFact / ImageDictFact
//    defaults, recsys).  The gradient row of atom j is evaluated lazily as
//    B_j - sum_i C[i,j] D_i with the CURRENT dictionary, so no k x s gradient is
//    ever stored and the ger passes disappear.  Atoms are processed in blocks of
//    NB = 32 (in sweep order).  Inside a block the l2 projection is a pure
//    rescaling, u_j -> alpha_j u_j, hence every candidate u_j is a combination
//    T[j,:] of the block's alpha-independent vectors
//        a_j = (B_j - sum_{i not in block or i after j} C[i,j] D_i) / C[j,j],
//    which one matrix-core product (s x k) . (k x NB) delivers for all features
//    at once.  Norms follow from the NB x NB Gram matrix of the a_j (one
//    double-precision reduction over the features per BLOCK instead of per
//    atom), the alpha_j recursion runs on that Gram matrix in one wavefront, and
//    D_j = alpha_j sum_m T[j,m] a_m is applied to all features in parallel.
//    k/NB global reductions instead of k; 2 k^2 s flops on MFMA instead of
//    4 k^2 s flops of ger.
//
//  * generic path (l1 / elastic-net atoms, positivity: fMRIDictFact, NMF): one
//    "gradient row" kernel (a wavefront per sampled feature) and one projection
//    kernel (Michelot iteration, enet_block.hpp) per atom.
//
//  * sgd path (dict_fact.py:695-708).
#include "bcd_shared.hpp"

namespace modl {

std::atomic<int> g_bcd_acc{1};
std::atomic<unsigned long long *> g_atom_stamps{nullptr};
std::atomic<int> g_bcd_tiny{1};
// diagnostics (modl_debug_set(MODL_DEBUG_BCD_PERSIST, 0)): one launch per block of 32 atoms (bcd_block_kernel) instead of the
// persistent launch of bcd_persist.hip
std::atomic<int> g_bcd_persist{1};

struct DuLayout {
    size_t off_CP, off_cdiag, off_frozen, off_coef, off_a, off_partial, off_Tp, off_u, off_pold, off_Dnew, off_colp, off_BsP, off_gpartial, off_norm_in, off_pacc, off_prec, off_Sbuf, off_pflags, off_qcoef, off_pstamps, off_few, off_pipe, total;
    int64_t nslab_max, nwg_grad;
};

// row stride of the scratch of the grouped atom update: what the projecting workgroup covers with 12 / 20 / 24 elements
// per thread (s beyond that: the group path is not taken, the stride is never used)
constexpr int kFewRows = 32;                         // sampled features per workgroup of bcd_few_kernel
constexpr int kFewMaxWg = 64;                        // ... and the most workgroups it runs on (2048 features)
constexpr int kFewRec = kNB * kNB + kNB;             // a Gram record: the 32 x 32 matrix of a block's candidates + their old norms
constexpr long long kFewSentinel = 0x7ff8feed7ff8feedll;   // what an exchange slot holds until its record arrives (a NaN no sum produces)
std::atomic<int> g_bcd_few{1};                       // modl_debug_set(MODL_DEBUG_BCD_FEW, ...)

constexpr int64_t kPipeMinRows = 24 * 256;           // (= kProjEpt * 256: beyond the register-resident projection)
// row stride of the pipelined sweep's scratch: what the last workgroup covers with 40 / 64 elements per thread (atom_project_regs)
static int64_t pipe_row_stride(int64_t s) { return s <= 40 * 256 ? 40 * 256 : (s <= 64 * 256 ? 64 * 256 : s); }
static int64_t atom_row_stride(int64_t s) { return s <= 12 * 256 ? 12 * 256 : (s <= 20 * 256 ? 20 * 256 : 24 * 256); }

static DuLayout du_layout(size_t tsz, int64_t s_max, int k) {
    DuLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.nslab_max = cdiv(s_max > 0 ? s_max : 1, 32);      // the fused block kernel uses slabs of 32 or 64 rows
    L.nwg_grad = 4096;
    const size_t k4 = (size_t)cdiv(k, 4) * 4;                         // the fused path pads the atoms to a multiple of 4
    L.off_CP = take(tsz * (size_t)(cdiv(k, 32) * 32) * k4);           // (fused path: fragment order, whole tiles of 32 columns)
    L.off_cdiag = take(tsz * k4);
    L.off_frozen = take(sizeof(int32_t) * k4);
    L.off_coef = take(sizeof(double) * (size_t)kNB * k4);
    L.off_a = take(tsz * (size_t)s_max * kNB);
    L.off_partial = take(sizeof(double) * 2 * (size_t)L.nslab_max * (kNB * kNB + kNB));   // two record buffers (ping-pong)
    L.off_Tp = take(sizeof(double) * 2 * (kNB * kNB + kNB) + 256 + 512);  // two CA records (ping-pong) + arrival counters + debug stamps
    L.off_u = take(tsz * (size_t)s_max);
    L.off_pold = take(sizeof(double) * (size_t)L.nwg_grad);
    // sgd / the packed dictionary (whole tiles of 32 rows) / the scratch of a group of kAtomGroupMax atoms of the grouped
    // atom update (numerators f64, old values, two staged groups)
    L.off_Dnew = take(std::max(tsz * (size_t)(cdiv(s_max > 0 ? s_max : 1, 32) * 32) * k4,
                               (size_t)kAtomGroupMax * (size_t)atom_row_stride(s_max > 0 ? s_max : 1) * (sizeof(double) + 3 * tsz) + 64));
    L.off_colp = take(sizeof(double) * (size_t)L.nslab_max * k);
    L.off_BsP = take(tsz * (size_t)s_max * k4);                       // packed B rows (packed D shares off_Dnew)
    L.off_gpartial = take(sizeof(double) * 2 * (size_t)kCounters * (kNB * kNB + kNB));   // group sums (two-level reduction), ping-pong
    L.off_norm_in = take(tsz * k4);
    // the persistent launch (bcd_persist.hip; f32, <= 512 atoms): one accumulator (x kAccShards) of the look-ahead pieces,
    // one S buffer and one pair of flags per block of 32 atoms, two sets of per-workgroup records, Q against the block before
    const bool persist = tsz == 4 && k <= 32 * kPersistBlocksMax;
    const size_t nblk = (size_t)cdiv(k, kNB);
    L.off_pacc = take(persist ? sizeof(long long) * nblk * kAccShards * kPAccWords : 0);
    L.off_prec = take(persist ? sizeof(double) * 2 * (size_t)std::min<int64_t>(L.nslab_max, kPersistRowsMax) * kPEntries : 0);
    L.off_Sbuf = take(persist ? sizeof(double) * nblk * kNB * kNB : 0);
    L.off_pflags = take(persist ? sizeof(unsigned int) * 64 : 0);      // arrive[16], sflag[16], err
    L.off_qcoef = take(persist ? sizeof(double) * (size_t)kNB * k4 : 0);
    L.off_pstamps = take(persist ? sizeof(unsigned long long) * kPersistStampWords : 0);
    // the f64 update of a few hundred to two thousand sampled features on several workgroups (bcd_few_kernel): three exchange
    // slots of kFewMaxWg Gram records + the launch's error word
    L.off_few = take(tsz == 8 ? sizeof(double) * 3 * (size_t)kFewMaxWg * kFewRec + 64 : 0);
    // the pipelined per-atom sweep of large sampled sets (atom_corr_project_kernel, round 6): two staged groups of four atoms, three
    // sets of old values, two sets of f64 numerators (the sets are s-strided inside; s <= s_max)
    L.off_pipe = take(s_max > kPipeMinRows ? align_up(tsz * (size_t)5 * 4 * (size_t)pipe_row_stride(s_max), 16) +
                                                 sizeof(double) * (size_t)2 * 4 * (size_t)pipe_row_stride(s_max) : 0);
    L.total = o;
    return L;
}

size_t dict_update_stamps_offset(int dtype, int64_t s_max, int k) {
    return du_layout(dtype == MODL_F32 ? 4 : 8, s_max, k).off_Tp + sizeof(double) * 2 * (kNB * kNB + kNB) + 256;
}

size_t dict_update_persist_stamps_offset(int dtype, int64_t s_max, int k) {
    return du_layout(dtype == MODL_F32 ? 4 : 8, s_max, k).off_pstamps;
}

size_t dict_update_workspace(int dtype, int64_t s_max, int k) {
    return du_layout(dtype == MODL_F32 ? 4 : 8, s_max, k).total;
}


// ---------------------------------------------------------------- blocked path
template <typename T>
__global__ __launch_bounds__(256) void bcd_prepare_kernel(const T *C, const int32_t *order, int k, T *CP, T *cdiag,
                                                          int32_t *frozen, double *coef_all, unsigned int *counter,
                                                          const T *comp_norm, T *norm_in, long long *few = nullptr,
                                                          long long few_words = 0) {
    if (few) {                                       // bcd_few_kernel's exchange slots: sentinels (a slice per workgroup); its error word
        const long long sl = (few_words + gridDim.x - 1) / gridDim.x;
        const long long e1 = ((long long)(blockIdx.x + 1) * sl < few_words) ? (long long)(blockIdx.x + 1) * sl : few_words;
        for (long long e = (long long)blockIdx.x * sl + threadIdx.x; e < e1; e += 256) few[e] = kFewSentinel;
        if (blockIdx.x == 0 && threadIdx.x == 0) few[few_words] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < kCounters) counter[threadIdx.x] = 0;   // arrival tickets of the fused block kernel
    if (blockIdx.x == 0)                             // the budgets as they are before this update (fused path)
        for (int j = threadIdx.x; j < k; j += 256) norm_in[j] = comp_norm[j];
    extern __shared__ int32_t inv[];                 // position of each atom in the sweep
    for (int j = threadIdx.x; j < k; j += 256) inv[order[j]] = j;
    __syncthreads();
    const int m = blockIdx.x;                        // source atom (row of C)
    if (m < k) {
        const int pm = inv[m];
        for (int jj = threadIdx.x; jj < k; jj += 256) {
            T v = C[(int64_t)m * k + order[jj]];
            if (pm / kNB == jj / kNB && pm <= jj) v = 0;  // same block, not after jj: handled by the recursion
            CP[(int64_t)m * k + jj] = v;
            if (m == 0) {
                const T d = C[(int64_t)order[jj] * k + order[jj]];
                cdiag[jj] = d;
                frozen[jj] = !(d > (T)1e-20);         // dict_fact.py:681 "else do not update"
            }
        }
    }
    // coefficients of the in-block recursion: coef_all[jj][i] = C[o_i', o_jj] / C[o_jj, o_jj] where i' is the
    // i-th atom of jj's block and i < position of jj in its block (zero otherwise / for frozen atoms)
    if (m < kNB) {
        for (int jj = threadIdx.x; jj < k; jj += 256) {
            const int jl = jj % kNB, jb0 = jj - jl;
            double c = 0;
            if (m < jl && jb0 + m < k) {
                const int oi = order[jb0 + m], oj = order[jj];
                const T d = C[(int64_t)oj * k + oj];
                if (d > (T)1e-20) c = (double)C[(int64_t)oi * k + oj] / (double)d;
            }
            coef_all[(int64_t)jj * kNB + m] = c;          // [sweep position][atom of its block]: one contiguous row per step
        }
    }
}

// Fused path: everything the block launches need, in ONE launch.  Blocks [0, NB): recursion coefficients
// (block 0 also: arrival counters, snapshot of the norm budgets, diagonal and frozen flags; all: a slice of the Gram
// accumulators to clear); then groups of kSetupRows rows per block: C in sweep coordinates, CPP[m'][jj] = C[o_m'][o_jj]
// with the block-lower-triangular mask; then the sampled rows gathered in sweep order, DsP[f][jj] = Dt[subset[f]][order[jj]]
// and BsP likewise — every later access of the block kernels is a plain contiguous row.
template <typename T>
__global__ __launch_bounds__(256) void bcd_setup_kernel(const T *C, const int32_t *order, int k, int kp, T *CPP, T *cdiag,
                                                        int32_t *frozen, double *coef_all, unsigned int *counter,
                                                        const T *comp_norm, T *norm_in, const T *Dt, const T *Bt,
                                                        const int32_t *subset, int64_t s, T *DsP, T *BsP, long long *acc,
                                                        long long *pacc, long long pacc_words, double *qcoef, unsigned int *pflags,
                                                        long long *sbuf, long long sbuf_words) {
    // kp = k rounded up to a multiple of 4: the packed arrays carry kp - k dead atoms (zero columns, frozen), so that the
    // 16-byte fragments of the block kernel exist for every number of atoms
    int id = (int)blockIdx.x;
    if (pacc) {                                       // the persistent launch's accumulators (a slice per workgroup) and flags
        const long long sl = (pacc_words + gridDim.x - 1) / gridDim.x;
        const long long e1 = ((long long)(id + 1) * sl < pacc_words) ? (long long)(id + 1) * sl : pacc_words;
        for (long long e = (long long)id * sl + threadIdx.x; e < e1; e += 256) pacc[e] = 0;
        if (id == 0 && threadIdx.x < 64) pflags[threadIdx.x] = 0;
        // the S buffers: sentinels, which the resolver's 8-byte stores replace - the data is its own flag (bcd_persist.hip: fetch_S)
        const long long ss = (sbuf_words + gridDim.x - 1) / gridDim.x;
        const long long s1 = ((long long)(id + 1) * ss < sbuf_words) ? (long long)(id + 1) * ss : sbuf_words;
        for (long long e = (long long)id * ss + threadIdx.x; e < s1; e += 256) sbuf[e] = kPersistSentinel;
    }
    if (id < kNB) {
        const int m = id;
        if (acc) {                                    // the accumulators, a slice per workgroup of this group
            constexpr int W = 3 * kAccShards * kAccWords, SL = (W + kNB - 1) / kNB;
            for (int e = m * SL + threadIdx.x; e < (m + 1) * SL && e < W; e += 256) acc[e] = 0;
        }
        if (m == 0) {
            if (threadIdx.x < kCounters) counter[threadIdx.x] = 0;
            for (int j = threadIdx.x; j < kp; j += 256) {
                const bool real = j < k;
                const int oj = real ? order[j] : 0;
                norm_in[j] = real ? comp_norm[oj] : (T)0;   // in SWEEP order: the resolver reads its budgets without a dependent load
                const T d = real ? C[(int64_t)oj * k + oj] : (T)0;
                cdiag[j] = real ? d : (T)1;
                frozen[j] = !(d > (T)1e-20);          // dict_fact.py:681 "else do not update"
            }
        }
        for (int jj = threadIdx.x; jj < kp; jj += 256) {
            const int jl = jj % kNB, jb0 = jj - jl;
            double c = 0;
            if (jj < k && m < jl && jb0 + m < k) {
                const int oi = order[jb0 + m], oj = order[jj];
                const T d = C[(int64_t)oj * k + oj];
                if (d > (T)1e-20) c = (double)C[(int64_t)oi * k + oj] / (double)d;
            }
            coef_all[(int64_t)jj * kNB + m] = c;          // [sweep position][atom of its block]: one contiguous row per step
            if (qcoef) {                                  // ... and against atom m of the block BEFORE (look-ahead: gram_ahead)
                double qv = 0;
                if (jj < k && jb0 >= kNB) {
                    const int oi = order[jb0 - kNB + m], oj = order[jj];
                    const T d = C[(int64_t)oj * k + oj];
                    if (d > (T)1e-20) qv = (double)C[(int64_t)oi * k + oj] / (double)d;
                }
                qcoef[(int64_t)jj * kNB + m] = qv;
            }
        }
        return;
    }
    // kSetupRows rows per workgroup (one row each: k + s workgroups of one element per thread, a launch that lasted as long as
    // its grid took to start - 5.7 us at the metric's shape, 13.6 at reduction 1); a row's loads are all requested before
    // the first store, and the gathered column index serves the rows of the group
    id -= kNB;
    const int ncp = (kp + kSetupRows - 1) / kSetupRows;
    if (id < ncp) {
        int om[kSetupRows];
#pragma unroll
        for (int r = 0; r < kSetupRows; ++r) {
            const int mp = id * kSetupRows + r;
            om[r] = mp < k ? order[mp] : 0;
        }
        for (int jj = threadIdx.x; jj < kp; jj += 256) {
            const int oj = jj < k ? order[jj] : 0;
            T v[kSetupRows];
#pragma unroll
            for (int r = 0; r < kSetupRows; ++r) v[r] = C[(int64_t)om[r] * k + oj];
#pragma unroll
            for (int r = 0; r < kSetupRows; ++r) {
                const int mp = id * kSetupRows + r;
                if (mp >= kp) break;
                T x = (mp < k && jj < k) ? v[r] : (T)0;
                if (mp / kNB == jj / kNB && mp <= jj) x = 0;
                CPP[dfrag(jj, mp, kp)] = x;              // fragment order: (target position, 4 consecutive source atoms)
            }
        }
        return;
    }
    id -= ncp;
    const int64_t fb = (int64_t)id * kSetupRows;
    if (fb < s) {
        int64_t src[kSetupRows];
#pragma unroll
        for (int r = 0; r < kSetupRows; ++r) src[r] = sub_row(subset, fb + r < s ? fb + r : s - 1) * k;
        for (int jj = threadIdx.x; jj < kp; jj += 256) {
            const bool real = jj < k;
            const int o = real ? order[jj] : 0;
            T dv[kSetupRows], bv[kSetupRows];
#pragma unroll
            for (int r = 0; r < kSetupRows; ++r) { dv[r] = Dt[src[r] + o]; bv[r] = Bt[src[r] + o]; }
#pragma unroll
            for (int r = 0; r < kSetupRows; ++r) {
                const int64_t f = fb + r;
                if (f >= s) break;
                DsP[dfrag(f, jj, kp)] = real ? dv[r] : (T)0;
                BsP[f * kp + jj] = real ? bv[r] : (T)0;
            }
        }
    }
}

template <typename T> struct EpiBcdA {
    T *a; const T *Dt; const T *Bt; const T *cdiag; const int32_t *frozen; const int32_t *subset; const int32_t *order;
    int k, j0;
    __device__ __forceinline__ void operator()(int64_t f, int64_t jj, T v) const {
        const int64_t e = sub_row(subset, f) * k + order[j0 + jj];
        a[f * kNB + jj] = frozen[j0 + jj] ? Dt[e] : (Bt[e] - v) / cdiag[j0 + jj];
    }
    // the same with the operands fetched up front (gemm.hpp: epi_has_fetch)
    struct Fetched { T d, b, cd; int32_t fz; };
    __device__ __forceinline__ Fetched fetch(int64_t f, int64_t jj) const {
        const int64_t e = sub_row(subset, f) * k + order[j0 + jj];
        return Fetched{Dt[e], Bt[e], cdiag[j0 + jj], frozen[j0 + jj]};
    }
    __device__ __forceinline__ void finish(int64_t f, int64_t jj, T v, const Fetched &x) const {
        a[f * kNB + jj] = x.fz ? x.d : (x.b - v) / x.cd;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void bcd_gram_kernel(const T *a, const T *Dt, const int32_t *subset,
                                                       const int32_t *order, int64_t s, int k, int j0, int nb,
                                                       double *partial) {
    __shared__ T As[64][kNB + 1];
    __shared__ double d2red[8][kNB];
    const int64_t f_begin = (int64_t)blockIdx.x * kGramRows;
    const int64_t f_end = (f_begin + kGramRows < s) ? f_begin + kGramRows : s;
    const int i = threadIdx.x / 8, jb = (threadIdx.x % 8) * 4;
    double acc[4] = {0, 0, 0, 0};
    const int col = threadIdx.x % kNB, rg = threadIdx.x / kNB;   // for the old-atom norms
    const int ocol = order[j0 + (col < nb ? col : 0)];
    double d2 = 0;
    for (int64_t c0 = f_begin; c0 < f_end; c0 += 64) {
        __syncthreads();
        // (unconditional loads from clamped rows, selection afterwards: a per-thread branch around a load makes the
        // compiler wait for it on the spot -- the old-atom norms below were 24 serial round trips per chunk)
#pragma unroll
        for (int e = threadIdx.x; e < 64 * kNB; e += 256) {
            const int r = e / kNB, c = e % kNB;
            const T ld = a[((c0 + r < f_end) ? c0 + r : f_end - 1) * kNB + c];
            As[r][c] = (c0 + r < f_end && c < nb) ? ld : (T)0;
        }
        const int rows = (int)((f_end - c0 < 64) ? f_end - c0 : 64);
        int64_t srow[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) srow[t] = sub_row(subset, (rg + 8 * t < rows) ? c0 + rg + 8 * t : f_end - 1);
        T xo[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) xo[t] = Dt[srow[t] * k + ocol];
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const double ai = (double)As[r][i];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += ai * (double)As[r][jb + q];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const double x = (double)xo[t];
            if (col < nb && rg + 8 * t < rows) d2 += x * x;
        }
    }
    d2red[rg][col] = d2;
    __syncthreads();
    double *out = partial + (int64_t)blockIdx.x * (kNB * kNB + kNB);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[i * kNB + jb + q] = acc[q];
    if (threadIdx.x < kNB) {
        double t = 0;
        for (int g = 0; g < 8; ++g) t += d2red[g][threadIdx.x];
        out[kNB * kNB + threadIdx.x] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bcd_resolve_kernel(const double *partial, int nslab, const double *coef_all,
                                                          const int32_t *order, int k, int j0, int nb, T *comp_norm,
                                                          double *CAout) {
    __shared__ double M[kNB][kNB + 1];
    __shared__ double D2[kNB];
    __shared__ __attribute__((aligned(16))) double Cs[kNB * kNB];
    __shared__ __attribute__((aligned(16))) double scr[4 * kNB];
    stage_coef(coef_all, k, j0, Cs);
    reduce_partials(partial, nslab, M, D2);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int x = threadIdx.x & 31;
        const int jj_x = (x < nb) ? order[j0 + x] : 0;
        const double budget_x = (x < nb) ? (double)comp_norm[jj_x] : 0.0;
        resolve_wave<T>(M, D2, Cs, jj_x, budget_x, nb, comp_norm, CAout, kNB, scr);
    }
}

// D_new[f][o_j] = sum_{m <= j} Tp[j][m] a_m[f]  for the atoms j = jg, jg + NSTR, ... of one sampled feature
template <typename T, int NSTR>
__device__ __forceinline__ void apply_row(const T *ar, const double *Tps, T *row, const int32_t *order, int j0, int nb,
                                          int jg) {
    double av[kNB];
#pragma unroll
    for (int m = 0; m < kNB; ++m) {                                          // columns >= nb were never written:
        const T ld = ar[m];                                                  // loaded all the same (no branch around
        av[m] = (m < nb) ? (double)ld : 0.0;                                 // the load), then replaced by zeros
    }
    for (int j = jg; j < nb; j += NSTR) {
        double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
        for (int m = 0; m < kNB; m += 4) {
            acc0 += Tps[j * kNB + m] * av[m];
            acc1 += Tps[j * kNB + m + 1] * av[m + 1];
            acc2 += Tps[j * kNB + m + 2] * av[m + 2];
            acc3 += Tps[j * kNB + m + 3] * av[m + 3];
        }
        row[order ? order[j0 + j] : j0 + j] = (T)((acc0 + acc1) + (acc2 + acc3));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bcd_apply_kernel(const T *a, const double *__restrict__ CA, T *Dt,
                                                        const int32_t *subset, const int32_t *order, int64_t s, int k,
                                                        int j0, int nb) {
    __shared__ double CAs[kResStride];
    for (int e = threadIdx.x; e < kResStride; e += 256) CAs[e] = CA[e];
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x / 4;       // 4 threads per feature
    if (f >= s) return;
    apply_row<T, 4>(a + f * kNB, CAs, Dt + sub_row(subset, f) * k, order, j0, nb, threadIdx.x % 4);
}

// ---- the blocked update of a SMALL sampled set in ONE launch ------------------------------------------------------
// The separate-launch form of the blocked update (f64) costs five launches per block of 32 atoms - gather-GEMM, Gram,
// recursion, apply - each of them a single small workgroup's worth of work when only a handful of features are sampled:
// ImageDictFact's 8 x 8 patches at reduction 10 sample SIX features (k = 256: 8 blocks, 33 launches, 0.40 ms of a 1.6 ms
// minibatch).  Here one workgroup runs the whole sweep: per block the candidates a = (B - D CP) / diag on the f64 matrix
// cores into LDS, their Gram matrix and the old norms, the recursion on one wavefront (resolve_wave), the apply - the
// dictionary rows it changes are the ones the next block reads, in order, inside one compute unit: 0.40 -> 0.13 ms at that
// shape.  It costs ~10 us + 0.13 us per sampled feature per block against ~47 us for the separate launches, which spread
// the features over several workgroups: taken up to kTinyRows sampled features (a masked minibatch of RecsysDictFact
// touches 300-450 items and keeps the separate launches: 135 against 94 us, measured).
constexpr int kTinyRows = 192;
template <typename T>
__global__ __launch_bounds__(256) void bcd_tiny_kernel(T *Dt, const T *Bt, const T *CP, const T *cdiag, const int32_t *frozen,
                                                       const double *coef_all, const int32_t *subset, const int32_t *order,
                                                       int s, int k, T *comp_norm) {
    extern __shared__ __attribute__((aligned(16))) char tiny_smem[];
    double (*M)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(tiny_smem);             // [NB][NB + 1]
    double *D2 = reinterpret_cast<double *>(tiny_smem) + kNB * (kNB + 1);                // [NB]
    double *Cs = D2 + kNB;                                                               // [NB * NB]
    double *scr = Cs + kNB * kNB;                                                        // [4 NB]
    double *CAs = scr + 4 * kNB;                                                         // [kResStride]
    double *d2red = CAs + kResStride;                                                    // [8][NB]
    T *As = reinterpret_cast<T *>(d2red + 8 * kNB);                                      // [s][NB + 1]
    const int tid = threadIdx.x;
    for (int j0 = 0; j0 < k; j0 += kNB) {
        const int nb = (k - j0 < kNB) ? k - j0 : kNB;
        stage_coef(coef_all, k, j0, Cs);
        // (1) candidates on the f64 matrix cores: tiles of 16 features x 16 atoms, a wavefront per tile, the contraction
        // over the k atoms in chunks of 64 whose operands - dictionary rows and coefficient columns straight from L2 -
        // are all requested before the first product (a thread-per-element loop of dependent loads took 100 us per
        // block at s = 300, k = 50)
        {
            typedef double d4v __attribute__((ext_vector_type(4)));
            const int lane = tid & 63, wid = tid >> 6;
            const int ntile = ((s + 15) / 16) * 2;
            for (int t = wid; t < ntile; t += 4) {
                const int ft = t >> 1, ct = t & 1;
                const int f = ft * 16 + (lane & 15);
                const int64_t rowoff = sub_row(subset, f < s ? f : s - 1) * k;
                const int jj = ct * 16 + (lane & 15);
                const int jc = (jj < nb) ? jj : 0;
                d4v acc = {0.0, 0.0, 0.0, 0.0};
                // the epilogue's operands with the first chunk (element (row (lane >> 4) + 4 r, column lane & 15) of the tile)
                const int64_t ocolv = order[j0 + jc];
                const T cd = cdiag[j0 + jc];
                const int fz = frozen[j0 + jc];
                T eB[4], eD[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fr = ft * 16 + (lane >> 4) + 4 * r;
                    const int64_t el = sub_row(subset, fr < s ? fr : s - 1) * k + ocolv;
                    eB[r] = Bt[el];
                    eD[r] = Dt[el];
                }
                constexpr int CH = 16;                                   // products per chunk (64 atoms)
                for (int m0 = 0; m0 < k; m0 += 4 * CH) {
                    double av[CH], bv[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int m = m0 + 4 * c + (lane >> 4);
                        const int mc = (m < k) ? m : k - 1;
                        av[c] = (double)Dt[rowoff + mc];
                        bv[c] = (double)CP[(int64_t)mc * k + j0 + jc];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int m = m0 + 4 * c + (lane >> 4);
                        const double a = (m < k) ? av[c] : 0.0, b = (m < k) ? bv[c] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fr = ft * 16 + (lane >> 4) + 4 * r;
                    const T av2 = fz ? eD[r] : (T)(((double)eB[r] - acc[r]) / (double)cd);
                    if (fr < s) As[fr * (kNB + 1) + jj] = (jj < nb) ? av2 : (T)0;
                }
            }
        }
        __syncthreads();
        // (2) Gram matrix of the candidates (thread: row i, four columns) and the old squared norms
        {
            const int i = tid / 8, jb = (tid % 8) * 4;
            double acc[4] = {0, 0, 0, 0};
            for (int f = 0; f < s; ++f) {
                const double ai = (double)As[f * (kNB + 1) + i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += ai * (double)As[f * (kNB + 1) + jb + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) M[i][jb + q] = acc[q];
            const int col = tid % kNB, rg = tid / kNB;
            const int ocol = order[j0 + (col < nb ? col : 0)];
            double d2 = 0;
            for (int f = rg; f < s; f += 8) {
                const double x = (double)Dt[sub_row(subset, f) * k + ocol];
                d2 += x * x;
            }
            d2red[rg * kNB + col] = (col < nb) ? d2 : 0.0;
        }
        __syncthreads();
        if (tid < kNB) {
            double t = 0;
            for (int g = 0; g < 8; ++g) t += d2red[g * kNB + tid];
            D2[tid] = t;
        }
        __syncthreads();
        // (3) the recursion of the block, one wavefront
        if (tid < 64) {
            const int x = tid & 31;
            const int jj_x = (x < nb) ? order[j0 + x] : 0;
            const double budget_x = (x < nb) ? (double)comp_norm[jj_x] : 0.0;
            resolve_wave<T>(M, D2, Cs, jj_x, budget_x, nb, comp_norm, CAs, kNB, scr);
        }
        __syncthreads();
        // (4) apply: four threads per feature
        for (int f0 = 0; f0 < s; f0 += 64) {
            const int f = f0 + tid / 4;
            if (f < s) {
                T ar[kNB];
#pragma unroll
                for (int m = 0; m < kNB; ++m) ar[m] = As[f * (kNB + 1) + m];
                apply_row<T, 4>(ar, CAs, Dt + sub_row(subset, f) * k, order, j0, nb, tid % 4);
            }
        }
        __syncthreads();                                   // (the rows are written: the next block reads them)
    }
}
// ---- the same sweep on A FEW workgroups (round 6): 193 to 2048 sampled features, f64 - the masked minibatch of RecsysDictFact
// touches about a thousand items, and as separate launches (candidates, Gram matrix, recursion, apply: four launches of 7-19 us of
// latency per block of 32 atoms) its dictionary update was 125 of the minibatch's 197 us.  Every workgroup keeps kFewRows features:
// the candidates, their Gram contribution and the apply are the one-workgroup kernel's, on its own rows - the rows of different
// workgroups never meet - and the ONE thing the workgroups share per block is the 32 x 32 Gram matrix (+ 32 old norms): each writes
// its record through to a slot, every thread sums its entries over the workgroups in workgroup order (run-to-run identical; the
// slots hold a sentinel until the record arrives: the data is its own flag), and every workgroup runs the recursion for itself.
// One memory round trip per block.  Three slots in rotation (a workgroup restores its own record of the block BEFORE the one it has
// just summed: every workgroup has delivered this block, so every workgroup is done reading the one before).  The budgets are read
// from the snapshot the prepare launch took; workgroup 0 writes the new ones.
// The waits are bounded; with at most sixteen workgroups co-residency is not in question on a part of 256 compute units - a wait
// that nevertheless gives up raises the launch's error word and the plan's flag (MODL_ETIMEOUT from the next enqueue).
template <typename T>
__global__ __launch_bounds__(256) void bcd_few_kernel(T *Dt, const T *Bt, const T *CP, const T *cdiag, const int32_t *frozen,
                                                      const double *coef_all, const int32_t *subset, const int32_t *order,
                                                      int s, int k, T *comp_norm, const T *norm_in, double *xch,
                                                      unsigned int *err, unsigned int *flags) {
    extern __shared__ __attribute__((aligned(16))) char tiny_smem[];
    double (*M)[kNB + 1] = reinterpret_cast<double (*)[kNB + 1]>(tiny_smem);             // [NB][NB + 1]
    double *D2 = reinterpret_cast<double *>(tiny_smem) + kNB * (kNB + 1);                // [NB]
    double *Cs = D2 + kNB;                                                               // [NB * NB]
    double *scr = Cs + kNB * kNB;                                                        // [4 NB]
    double *CAs = scr + 4 * kNB;                                                         // [kResStride]
    double *d2red = CAs + kResStride;                                                    // [8][NB]
    T *As = reinterpret_cast<T *>(d2red + 8 * kNB);                                      // [kFewRows][NB + 1]
    T *sink = As + (size_t)kFewRows * (kNB + 1);                                         // [k] where the other workgroups' budgets go
    int64_t *roff = reinterpret_cast<int64_t *>(sink + k + (k & 1));                     // [kFewRows] element offsets of this workgroup's rows
    int &gave_up = *reinterpret_cast<int *>(roff + kFewRows);                            // (dynamic LDS only: the attribute below asks for all of it)
    const int tid = threadIdx.x;
    const int W = (int)gridDim.x, w = (int)blockIdx.x;
    const int r0 = w * kFewRows;
    const int rows = (s - r0 < kFewRows) ? s - r0 : kFewRows;
    if (tid == 0) gave_up = 0;
    // the rows' offsets ONCE (subset -> offset -> element was two dependent round trips in front of every tile and, in the loop of
    // the old norms, sixteen times per block: 20 of a block's 41 us)
    if (tid < kFewRows) roff[tid] = sub_row(subset, r0 + (tid < rows ? tid : rows - 1)) * k;
    __syncthreads();
    int bi = 0;
    for (int j0 = 0; j0 < k; j0 += kNB, ++bi) {
        const int nb = (k - j0 < kNB) ? k - j0 : kNB;
        stage_coef(coef_all, k, j0, Cs);
        // (1) candidates of this workgroup's rows on the f64 matrix cores (bcd_tiny_kernel)
        {
            typedef double d4v __attribute__((ext_vector_type(4)));
            const int lane = tid & 63, wid = tid >> 6;
            const int ntile = ((rows + 15) / 16) * 2;
            double dsq[2] = {0.0, 0.0};                              // this lane's share of the old squared norms of its columns
            for (int t = wid; t < ntile; t += 4) {
                const int ft = t >> 1, ct = t & 1;
                const int f = ft * 16 + (lane & 15);
                const int64_t rowoff = roff[f < rows ? f : rows - 1];
                const int jj = ct * 16 + (lane & 15);
                const int jc = (jj < nb) ? jj : 0;
                d4v acc = {0.0, 0.0, 0.0, 0.0};
                const int64_t ocolv = order[j0 + jc];
                const T cd = cdiag[j0 + jc];
                const int fz = frozen[j0 + jc];
                T eB[4], eD[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fr = ft * 16 + (lane >> 4) + 4 * r;
                    const int64_t el = roff[fr < rows ? fr : rows - 1] + ocolv;
                    eB[r] = Bt[el];
                    eD[r] = Dt[el];
                }
                constexpr int CH = 16;                                   // products per chunk (64 atoms)
                for (int m0 = 0; m0 < k; m0 += 4 * CH) {
                    double av[CH], bv[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int m = m0 + 4 * c + (lane >> 4);
                        const int mc = (m < k) ? m : k - 1;
                        av[c] = (double)Dt[rowoff + mc];
                        bv[c] = (double)CP[(int64_t)mc * k + j0 + jc];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int m = m0 + 4 * c + (lane >> 4);
                        const double a = (m < k) ? av[c] : 0.0, b = (m < k) ? bv[c] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fr = ft * 16 + (lane >> 4) + 4 * r;
                    const T av2 = fz ? eD[r] : (T)(((double)eB[r] - acc[r]) / (double)cd);
                    if (fr < rows) As[fr * (kNB + 1) + jj] = (jj < nb) ? av2 : (T)0;
                    if (fr < rows && jj < nb) dsq[ct] += (double)eD[r] * (double)eD[r];   // (every (row, atom) of the block passes here once)
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {                         // rows (lane >> 4) + 4 r of all this wavefront's tiles -> the column's sum
                double v = dsq[ct];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 16) d2red[wid * kNB + 16 * ct + lane] = v;
            }
        }
        __syncthreads();
        // (2) this workgroup's share of the Gram matrix of the candidates and of the old squared norms
        {
            const int i = tid / 8, jb = (tid % 8) * 4;
            double acc[4] = {0, 0, 0, 0};
            for (int f = 0; f < rows; ++f) {
                const double ai = (double)As[f * (kNB + 1) + i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += ai * (double)As[f * (kNB + 1) + jb + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) M[i][jb + q] = acc[q];
        }
        __syncthreads();
        if (tid < kNB) D2[tid] = (d2red[tid] + d2red[kNB + tid]) + (d2red[2 * kNB + tid] + d2red[3 * kNB + tid]);   // (the four wavefronts' sums)
        __syncthreads();
        if (W > 1) {
            // (2b) the records meet: every thread its entries e = tid + 256 q of the record
            double *slot = xch + (size_t)(bi % 3) * W * kFewRec;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the sentinels this workgroup restored two blocks ago have landed)
            for (int e = tid; e < kFewRec; e += 256) {
                const double v = (e < kNB * kNB) ? M[e / kNB][e % kNB] : D2[e - kNB * kNB];
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot + (size_t)w * kFewRec + e),
                                   (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // every load of a batch is requested before the first is looked at (one at a time - load, test, add - a thread paid
            // forty dependent memory round trips per block: 28 of the block's 47 us); a record that has not arrived yet is
            // simply asked for again
            bool ok = true;
            constexpr int NE = (kFewRec + 255) / 256;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int e = tid + 256 * q;
                const int ec = e < kFewRec ? e : kFewRec - 1;
                double sum = 0.0;
                for (int w0 = 0; w0 < W; w0 += 8) {
                    long long bits[8];
                    for (unsigned spins = 0;; ++spins) {
#pragma unroll
                        for (int u2 = 0; u2 < 8; ++u2) {
                            const int ww = (w0 + u2 < W) ? w0 + u2 : W - 1;
                            bits[u2] = (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(slot + (size_t)ww * kFewRec + ec),
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        bool have = true;
#pragma unroll
                        for (int u2 = 0; u2 < 8; ++u2) have = have && bits[u2] != kFewSentinel;
                        if (have) break;
                        if (spins > (1u << 18) || ((spins & 63) == 63 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            ok = false;
                            break;
                        }
                    }
#pragma unroll
                    for (int u2 = 0; u2 < 8; ++u2)
                        if (w0 + u2 < W) sum += __longlong_as_double(bits[u2]);
                }
                if (e < kNB * kNB) M[e / kNB][e % kNB] = sum;
                else if (e < kFewRec) D2[e - kNB * kNB] = sum;
            }
            if (!ok) {
                gave_up = 1;
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (flags) __hip_atomic_store(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
            if (gave_up) return;                                         // (blocks before this one are applied: the update is incomplete)
            // this workgroup's record of the block BEFORE back to sentinels (everybody has delivered this block: nobody reads that one any more)
            // (written through, like the records: a cached store could reach memory AFTER the record this workgroup writes to the
            //  same slot two blocks later)
            if (bi >= 1) {
                unsigned long long *old = reinterpret_cast<unsigned long long *>(xch + ((size_t)((bi + 2) % 3) * W + w) * kFewRec);
                for (int e = tid; e < kFewRec; e += 256)
                    __hip_atomic_store(old + e, (unsigned long long)kFewSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (3) the recursion of the block, one wavefront (every workgroup for itself; workgroup 0 keeps the new budgets)
        if (tid < 64) {
            const int x = tid & 31;
            const int jj_x = (x < nb) ? order[j0 + x] : 0;
            const double budget_x = (x < nb) ? (double)norm_in[jj_x] : 0.0;
            resolve_wave<T>(M, D2, Cs, jj_x, budget_x, nb, w == 0 ? comp_norm : sink, CAs, kNB, scr);
        }
        __syncthreads();
        // (4) apply: four threads per feature
        for (int f0 = 0; f0 < rows; f0 += 64) {
            const int f = f0 + tid / 4;
            if (f < rows) {
                T ar[kNB];
#pragma unroll
                for (int m = 0; m < kNB; ++m) ar[m] = As[f * (kNB + 1) + m];
                apply_row<T, 4>(ar, CAs, Dt + roff[f], order, j0, nb, tid % 4);
            }
        }
        __syncthreads();                                   // (the rows are written: the next block reads them)
    }
}
static size_t bcd_tiny_lds(size_t tsz, int64_t s) {
    return sizeof(double) * (size_t)(kNB * (kNB + 1) + kNB + kNB * kNB + 4 * kNB + kResStride + 8 * kNB) +
           tsz * (size_t)s * (kNB + 1) + 16;
}

// ---- fused block kernel (f32) -------------------------------------------------------------------
// One launch per block of NB atoms, 5 wavefronts per workgroup (4 workers + 1 resolver), RB features per
// workgroup.  Launch b
//   (A) requests everything it needs from HBM/L2 at once (one memory round trip);
//   (B) sums the Gram records of block b - 1 (written by launch b - 1) in a fixed order — EVERY workgroup
//       does this redundantly, so no cross-workgroup hand-off sits on the critical path;
//   (C) resolver wave: the alpha recursion of block b - 1 (its result stays in LDS);  meanwhile the
//       workers form a = D CP on the matrix cores with the dictionary AS IT IS IN MEMORY: dictionary rows go
//       straight from HBM/L2 into MFMA A-operands, the k x NB coefficient block sits in LDS, the four
//       waves split the contraction over the k atoms;
//   (D) apply block b - 1 to this workgroup's features: D_j = sum_m S[j][m] a_m on the f64 matrix cores;
//   (E) the 32 columns that (D) changed enter the product through a rank-32 correction (D_cur = D_mem + Delta);
//   (F) a = (B - D CP) / diag, written out and kept in LDS;
//   (G) the NB x NB Gram contribution of these RB features on the f64 matrix cores -> one record per
//       workgroup (two buffers: the records a launch reads in (B) are not the ones it writes in (G)).
//       With more than `group` workgroups the last workgroup of each group to finish pre-sums its group's
//       records (release / relaxed ticket / acquire, no spinning), so that (B) of the next launch reads at
//       most kCounters records.
// The last launch (nb == 0) only runs (A)-(D) for the final block.
template <int RT, int GPW>   // 32 * RT features per workgroup; GPW contraction groups (8 atoms) per worker wave
// (waves per SIMD pinned to the two a 384-thread workgroup needs: with the recursion on two lean wavefronts the
// largest register need of the kernel fell under 170, and the scheduler, now aiming at three waves per SIMD, serialised
// the workers' ~100 operand loads - one round trip each, 35 k cycles - to stay there)
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(2, 2)))
void bcd_block_kernel(BcdBlockArgs p, BcdRiderArgs rider) {
    constexpr int RB = 32 * RT;
    constexpr int EPT = RB / 8;                      // epilogue elements per thread
    constexpr int DLS = kNB + 4;                     // row stride of the Delta tile (16-byte aligned rows)
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef double d4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int k = p.k;
    // LDS carve (all from the dynamic region, 16-byte aligned pieces)
    double *Ms = reinterpret_cast<double *>(smem_raw);                        // [NB][64] Base rows (SinkBasePacked)
    double *D2s = Ms + kNB * 64;                                               // [NB]
    double *Cs = D2s + kNB;                                                    // [NB][NB] recursion coefficients
    double *CAs = Cs + kNB * kNB;                                              // [NB][kCaStride] S of the previous block
    double *d2red = CAs + kNB * kCaStride;                                     // [8][NB]
    float *red = reinterpret_cast<float *>(d2red + 8 * kNB);                   // [4][RB][NB + 1]; before (E): a-tile [RB][kApStride]
    float *As = red + 4 * RB * (kNB + 1);                                      // [RB][NB + 1]
    float *Dl = As + RB * (kNB + 1) + ((4 - (RB * (kNB + 1)) % 4) % 4);        // [RB][DLS] Delta of the previous block
    int *flag = reinterpret_cast<int *>(Dl + RB * DLS);
    double *CsT = reinterpret_cast<double *>(flag + 4);                        // [NB][NB] Cs transposed (resolve_helper)
    ResolveMail mail;
    mail.Pm = CsT + kNB * kNB;                                                 // [kMbox][64]
    mail.Zm = mail.Pm + kMbox * 64;                                            // [kMbox][64]
    mail.pcount = flag + 1;
    mail.zcount = flag + 2;
    float *Ap = red;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if ((int)blockIdx.x >= rider.nslab) {
        bcd_rider_tile(rider, smem_raw);
        return;
    }
    const bool worker = wid < 4;                     // wave 4: the chain of the recursion, wave 5: its helper
    const bool has_prev = p.nb_prev > 0, fin = p.nb == 0;
    if (tid == 0) { *mail.pcount = 0; *mail.zcount = 0; }
    const int64_t f0 = (int64_t)blockIdx.x * RB;
    const int nwg = rider.nslab, gsz = p.group, ngroups = (nwg + gsz - 1) / gsz;
#ifdef MODL_DIAG     // phase stamps of workgroup 0 (modl_somf_debug_stamps): the diagnostics build only - every stamp is an
                     // s_memtime behind an s_waitcnt lgkmcnt(0), i.e. it drains the LDS requests the recursion keeps in flight,
                     // and the launch lasts as long as its slowest workgroup
    unsigned long long *st = (p.stamps && blockIdx.x == 0 && !fin) ? p.stamps : nullptr;
#else
    unsigned long long *const st = nullptr;
#endif
    if (st && tid == 0) st[0] = clock64();

    // ---------------------------------------------------------------- (B) Gram of the previous block
    // First thing in the launch: the recursion is the critical path and only needs these records.
    int res_jj = 0;
    double res_budget = 0.0;
    if (has_prev) {
        const double *recs = (ngroups > 1) ? p.grec_in : p.rec_in;
        const int nrec = (ngroups > 1) ? ngroups : nwg;
        const SinkBasePacked rsink{Ms, D2s};
        if (worker) {
            if (st && tid == 0) st[16] = clock64();
            // the recursion coefficients of the block, Cs[j][i] = coef_all[j0_prev + j][i] (a contiguous block of 1024
            // doubles: two 16-byte loads per worker thread, requested together with the first records)
            typedef double d2v __attribute__((ext_vector_type(2)));
            d2v cf[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * (tid + 256 * q);
                const bool ok = p.j0_prev + e / kNB < k;
                cf[q] = *reinterpret_cast<const d2v *>(p.coef_all + (ok ? (int64_t)p.j0_prev * kNB + e : 0));   // (zeroed below)
            }
            const bool use_rec = !p.acc_in || (p.shards > 1 ? acc_load_sharded(p.acc_in, tid, true, rsink)
                                                           : acc_load(p.acc_in, tid, true, rsink));   // (out of range: the records)
            if (use_rec) reduce_records_v2<kPackStride>(recs, nrec, tid, true, rsink);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * (tid + 256 * q);
                const bool ok = p.j0_prev + e / kNB < k;
                Cs[e] = ok ? cf[q].x : 0.0;
                Cs[e + 1] = ok ? cf[q].y : 0.0;
                CsT[(e % kNB) * kNB + e / kNB] = ok ? cf[q].x : 0.0;
                CsT[(e % kNB + 1) * kNB + e / kNB] = ok ? cf[q].y : 0.0;
            }
            if (st && tid == 0) st[17] = clock64();
        } else if (wid == 4) {
            // the resolver's own inputs (atom index and norm budget of column x) and the tail of the records
            // (elements 512 ..) that the 256 worker threads do not cover
            const int x = lane & 31;
            const int res_jj_raw = p.order[p.j0_prev + ((x < p.nb_prev) ? x : 0)];   // (unconditional, clamped: no wait behind the load)
            res_jj = (x < p.nb_prev) ? res_jj_raw : 0;
            const float budget_raw = p.norm_in[(x < p.nb_prev) ? p.j0_prev + x : 0];
            const bool use_rec = !p.acc_in || (p.shards > 1 ? acc_load_sharded(p.acc_in, 256 + lane, 256 + lane < kPackStride / 2, rsink)
                                                           : acc_load(p.acc_in, 256 + lane, 256 + lane < kPackStride / 2, rsink));
            if (use_rec) reduce_records_v2<kPackStride>(recs, nrec, 256 + lane, 256 + lane < kPackStride / 2, rsink);
            res_budget = (x < p.nb_prev) ? (double)budget_raw : 0.0;
        } else {
#pragma unroll
            for (int q = 0; q < kNB * 32 / 64; ++q) {        // the identity half of the Base rows
                const int e = lane + 64 * q, m = e >> 5, xx = e & 31;
                Ms[m * 64 + xx] = (m == xx) ? 1.0 : 0.0;
            }
        }
    }
    __syncthreads();                                                                  // ---- barrier 1
    if (st && tid == 0) st[1] = clock64();
    // ---------------------------------------------------------------- (C) resolver | loads + main product
    const int col = tid % kNB, rg = (tid / kNB) % 8;             // epilogue: column, row group
    const bool col_ok = col < p.nb;
    float cdg = 1.f, eB[EPT], eD[EPT];
    int fz = 0;
    const int h = lane >> 5;
    constexpr int NA = RB * (kNB / 4) / 256;
    float dold[RT][4];
    int ocr[RT], subr[RT][4];       // destination of the applied values in the real dictionary: column, sampled row
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    f16v acc[RT];
    if (!worker) {
        // The two wavefronts of the recursion leave through their own copy of the remaining barriers: nothing of the
        // workers' state is live across their code (a common tail made the register allocator carry the workers'
        // accumulators and epilogue operands through the recursion - and spill the helper's running sums).  A
        // hardware barrier counts wavefronts, not program counters.
        __builtin_amdgcn_s_setprio(3);               // the recursion is the critical path: issue before the workers. loads
        if (wid == 4) {
            if (has_prev)
                resolve_chain<float>(D2s, Cs, res_jj, res_budget, p.nb_prev, blockIdx.x == 0 ? p.norm_out : nullptr,
                                     d2red, mail, st);
            if (st && lane == 0) st[2] = clock64();
        } else if (has_prev) {
            resolve_helper(Ms, CsT, CAs, kCaStride, mail);
            if (st && lane == 0) st[14] = clock64();
        }
        lds_barrier();                                                                // ---- barrier 2
        if (fin) return;
        lds_barrier();                                                                // ---- barrier 3
        lds_barrier();                                                                // ---- barrier 4
        lds_barrier();                                                                // ---- barrier 5
        if (ngroups > 1) {
            const int g = (int)blockIdx.x / gsz;
            const int gsize = (nwg - g * gsz < gsz) ? nwg - g * gsz : gsz;
            if (!arrive_last(p.counter + 1 + g, (unsigned int)gsize, flag)) return;
            if (wid == 4)
                reduce_records_v2<kPackStride>(p.rec_out + (int64_t)g * gsz * kPackStride, gsize, 256 + lane,
                                               256 + lane < kPackStride / 2, SinkGlobal{p.grec_out + (int64_t)g * kPackStride});
        }
        return;
    }
    {
        // every global load of the workers, requested at once: epilogue operands, the coefficient block as
        // MFMA B fragments (straight from L2: two 128-byte rows per wave instruction, no LDS staging and
        // therefore no barrier in the resolver's shadow), the dictionary rows as A fragments, the previous
        // block's old values and a-tile
        // The resolver wave shares SIMD 0 with worker wave 0; back-to-back MFMAs of that wave would slow the
        // recursion down (measured: +5 k cycles), so for k <= 256 the product is split over waves 1-3 only.
        constexpr int PW0 = (GPW == 8) ? (RT == 1 ? 2 : 1) : 0;      // first wave that takes part in the product
        constexpr int GW = (GPW == 8) ? (RT == 1 ? 16 : 11) : GPW;   // contraction groups (8 atoms) per product wave
        const bool pwave = __builtin_amdgcn_readfirstlane(wid) >= PW0;   // (wave-uniform for the compiler too)
        // 96 features per workgroup (RT == 3): the operands of the product are requested in NBATCH batches of GB
        // contraction groups, a batch behind the matrix-core instructions of the one before (whole range at once: 176
        // registers of operands, 107 of them spilled - requesting them took 24 k cycles and the workers, not the
        // recursion, were the critical path of the launch: 75.7 k cycles at p = 200 000; the extra round trips sit in
        // the recursion's shadow)
        constexpr int NBATCH = (RT == 3) ? 3 : 1;
        constexpr int GB = (GW + NBATCH - 1) / NBATCH;
        float bfr[GB][4];
        float4 av[GB][RT];
        float4 va[NA];
        if (!fin) {
            // (unconditional, clamped requests; every mask is applied behind the scheduler fence below: a select next to
            //  its load makes the compiler wait for the round trip on the spot, and three such waits sat in here)
            cdg = p.cdiag[p.j0 + (col_ok ? col : 0)];
            fz = p.frozen[p.j0 + (col_ok ? col : 0)];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int64_t f = f0 + rg + 8 * q;
                const bool ok = col_ok && f < p.s;
                const int64_t el = ok ? f * k + p.j0 + col : 0;
                eB[q] = p.Bt[el];
                eD[q] = p.Dt[ok ? dfrag(f, p.j0 + col, k) : 0];
            }
            const bool cok = (lane & 31) < p.nb;
            const float *cp0 = p.CP + dfrag(p.j0, 0, k);            // the block's 32 columns (j0 is a multiple of 32)
            const unsigned lane_off = cok ? (unsigned)(lane & 31) : 0u;
            // (wave-uniform branch: a wave outside the product requests none of its operands - wave 0 shares its SIMD
            // with the resolver, and the ~100 address computations + requests below took the recursion's issue slots
            // for its first 16 steps: ~530 cycles per atom instead of ~430)
            if (pwave)
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                const int kb = ((wid - PW0) * GW + g) * 8 + 4 * h;  // this lane's 4 consecutive atoms
                // 32-bit element offsets from one uniform base (k <= 512): two instructions of address arithmetic per
                // request instead of eight with a 64-bit multiply (92 cycles per request, measured).  k % 4 == 0, so
                // kb < k covers kb + u < k; any valid address serves the masked-out lanes.
                // (the coefficient matrix is stored in fragment order too, bcd_setup_kernel: this lane's four atoms are
                //  one 16-byte word, a wavefront's request one contiguous kilobyte; k % 4 == 0, so kb < k covers
                //  kb + u < k.  Masked below, behind the scheduler fence: a select next to its load has been compiled
                //  into load - wait - select, one round trip per request)
                const float4 b4 = *reinterpret_cast<const float4 *>(cp0 + dfrag(lane_off, kb < k ? kb : 0, k));
                bfr[g][0] = b4.x; bfr[g][1] = b4.y; bfr[g][2] = b4.z; bfr[g][3] = b4.w;
            }
            if (has_prev) {   // rank-32 correction: wave w contracts the previous block's atoms 8w .. 8w+7
                const int jb = wid * 8 + 4 * h;
                const float4 q4 = *reinterpret_cast<const float4 *>(cp0 + dfrag(lane_off, jb < p.nb_prev ? p.j0_prev + jb : 0, k));
                bq[0] = q4.x; bq[1] = q4.y; bq[2] = q4.z; bq[3] = q4.w;
            }
            // dictionary rows -> MFMA A operands, the wave's whole contraction range in flight at once
            if (pwave)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                int64_t f = f0 + t * 32 + (lane & 31);
                if (f >= p.s) f = p.s - 1;               // clamped: results of padded rows are discarded
                const float *rowp = p.Dt + dfrag(f, 0, k);
#pragma unroll
                for (int g = 0; g < GB; ++g) {
                    const int kb = ((wid - PW0) * GW + g) * 8 + 4 * h;
                    const bool ok = kb + 3 < k;
                    av[g][t] = *reinterpret_cast<const float4 *>(rowp + (ok ? kb : 0) * 32);   // (masked below)
                }
            }
        }
        // operands of the apply step, in the output layout of its matrix-core tiles: worker wave w owns tiles
        // w * RT .. w * RT + RT - 1 of the (RB / 16) x 2 grid (feature tile ft = t / 2, atom tile jt = t % 2)
        if (has_prev) {
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const int t = wid * RT + u, ft = t >> 1, jt = t & 1;
                const int cj = jt * 16 + (lane & 15);
                const int32_t *sub_src = p.subset ? p.subset : p.order;   // (any readable words when there is no subset)
                ocr[u] = p.order[p.j0_prev + ((cj < p.nb_prev) ? cj : 0)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t f = f0 + ft * 16 + (lane >> 4) + 4 * r;
                    const bool live = f < p.s && cj < p.nb_prev;
                    dold[u][r] = p.Dt[live ? dfrag(f, p.j0_prev + cj, k) : 0];
                    // (unconditional request - a branch on the null subset made the compiler wait behind each of these,
                    //  four round trips before the remaining operands were even requested; the destination offsets are
                    //  formed in (D))
                    subr[u][r] = sub_src[(p.subset && live) ? f : 0];
                }
            }
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int e = tid + 256 * q;
                const int r = e / (kNB / 4), c4 = (e % (kNB / 4)) * 4;
                const int64_t fr = (f0 + r < p.s) ? f0 + r : p.s - 1;
                va[q] = *reinterpret_cast<const float4 *>(p.a + fr * kNB + c4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!fin) {                                  // the masks of the requests above
            cdg = col_ok ? cdg : 1.f;
            fz = col_ok ? fz : 0;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const bool ok = col_ok && f0 + rg + 8 * q < p.s;
                eB[q] = ok ? eB[q] : 0.f;
                eD[q] = ok ? eD[q] : 0.f;
            }
            if (has_prev) {
                const bool cok = (lane & 31) < p.nb;
                const int jb = wid * 8 + 4 * h;
#pragma unroll
                for (int u = 0; u < 4; ++u) bq[u] = (cok && jb + u < p.nb_prev) ? bq[u] : 0.f;
            }
        }
        if (has_prev) {
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const int t = wid * RT + u, ft = t >> 1, jt = t & 1;
                const int cj = jt * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t f = f0 + ft * 16 + (lane >> 4) + 4 * r;
                    dold[u][r] = (f < p.s && cj < p.nb_prev) ? dold[u][r] : 0.f;
                }
            }
        }
        // the accumulator the NEXT launch adds to (idle during this one): cleared by a wavefront that has a SIMD to itself
        // and nothing to do until its operands arrive (at the head of the launch it sat in front of workgroup 0's record
        // loads; on wavefront 0 or 1 it takes issue slots from the recursion: 13.6 k cycles instead of 12.9 k)
        if (p.acc_zero && (int)blockIdx.x < p.shards && wid == 3)
            for (int e = lane; e < kAccWords; e += 64) p.acc_zero[(size_t)blockIdx.x * kAccWords + e] = 0;
        if (st && tid == 0) st[13] = clock64();
        if (st && tid == 64) st[20] = clock64();
        if (st && tid == 128) st[21] = clock64();            // product wave 2: operands requested
        if (!fin) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            // one 32-feature tile per wave (RT == 1): the four products of a contraction group go to four accumulators -
            // a v_mfma_f32_32x32x2_f32 issues every 16-21 cycles but its result takes 64 (scripts/micro/mfma_rate.hip).
            // (What this section really waits for is its operands: a compute unit draws ~27 bytes per cycle from the
            // fabric, and the workgroup asks for 134 KB of records, then 64 KB of operands, every launch.)
            constexpr bool kFourAcc = (RT == 1);
            f16v accx[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[q][r] = 0.f;
            if (pwave)
#pragma unroll
            for (int bt = 0; bt < NBATCH; ++bt) {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                if (bt * GB + g >= GW) continue;                     // (compile time: the last batch may be short)
                const int kb = ((wid - PW0) * GW + bt * GB + g) * 8 + 4 * h;
                const bool cok = (lane & 31) < p.nb;
#pragma unroll
                for (int u = 0; u < 4; ++u) bfr[g][u] = (cok && kb + u < k) ? bfr[g][u] : 0.f;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    if (!(kb + 3 < k)) av[g][t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kFourAcc) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].x, bfr[g][0], acc[t], 0, 0, 0);
                        accx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].y, bfr[g][1], accx[0], 0, 0, 0);
                        accx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].z, bfr[g][2], accx[1], 0, 0, 0);
                        accx[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].w, bfr[g][3], accx[2], 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].x, bfr[g][0], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].y, bfr[g][1], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].z, bfr[g][2], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][t].w, bfr[g][3], acc[t], 0, 0, 0);
                    }
                }
            }
            if constexpr (NBATCH > 1) {
                if (bt + 1 < NBATCH) {                               // the next batch of operands, into the same registers
                    __builtin_amdgcn_sched_barrier(0);
                    const bool cok = (lane & 31) < p.nb;
                    const float *cp0 = p.CP + dfrag(p.j0, 0, k);
                    const unsigned lane_off = cok ? (unsigned)(lane & 31) : 0u;
#pragma unroll
                    for (int g = 0; g < GB; ++g) {
                        if ((bt + 1) * GB + g >= GW) continue;
                        const int kb = ((wid - PW0) * GW + (bt + 1) * GB + g) * 8 + 4 * h;
                        const float4 b4 = *reinterpret_cast<const float4 *>(cp0 + dfrag(lane_off, kb < k ? kb : 0, k));
                        bfr[g][0] = b4.x; bfr[g][1] = b4.y; bfr[g][2] = b4.z; bfr[g][3] = b4.w;
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            int64_t f = f0 + t * 32 + (lane & 31);
                            if (f >= p.s) f = p.s - 1;
                            av[g][t] = *reinterpret_cast<const float4 *>(p.Dt + dfrag(f, 0, k) + (kb + 3 < k ? kb : 0) * 32);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            if (kFourAcc && pwave) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = (acc[0][r] + accx[0][r]) + (accx[1][r] + accx[2][r]);
            }
        }
        if (st && tid == 128) st[22] = clock64();            // product wave 2: product issued
        if (has_prev) {                              // a-tile of the previous block -> LDS for (D)
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int e = tid + 256 * q;
                const int r = e / (kNB / 4), c4 = (e % (kNB / 4)) * 4;
                float4 v = va[q];                        // columns >= nb_prev were never written
                v.x = (c4 + 0 < p.nb_prev) ? v.x : 0.f;
                v.y = (c4 + 1 < p.nb_prev) ? v.y : 0.f;
                v.z = (c4 + 2 < p.nb_prev) ? v.z : 0.f;
                v.w = (c4 + 3 < p.nb_prev) ? v.w : 0.f;
                *reinterpret_cast<float4 *>(Ap + r * kApStride + c4) = v;
            }
        }
        if (st && tid == 0) st[3] = clock64();
        if (st && tid == 64) st[15] = clock64();
        if (st && tid == 128) st[18] = clock64();
        if (st && tid == 192) st[19] = clock64();
    }
    lds_barrier();                                                                    // ---- barrier 2
    if (st && tid == 0) st[4] = clock64();
    // ---------------------------------------------------------------- (D) apply the previous block
    // Every A-operand load of the workgroup has been consumed by (C), so the stores below cannot overtake a
    // load of the old values.
    if (worker && has_prev) {
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int t = wid * RT + u, ft = t >> 1, jt = t & 1;
            d4v dn = {0.0, 0.0, 0.0, 0.0}, dn1 = {0.0, 0.0, 0.0, 0.0};
            const float *ap = Ap + (ft * 16 + (lane & 15)) * kApStride + (lane >> 4);
            const double *sp = CAs + (jt * 16 + (lane & 15)) * kCaStride + (lane >> 4);
            float fa[kNB / 4];                          // (every LDS operand requested before the first product; two accumulators)
            double fs[kNB / 4];
#pragma unroll
            for (int kk = 0; kk < kNB / 4; ++kk) { fa[kk] = ap[4 * kk]; fs[kk] = sp[4 * kk]; }
#pragma unroll
            for (int kk = 0; kk < kNB / 4; kk += 2) {
                dn = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk], fs[kk], dn, 0, 0, 0);
                dn1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk + 1], fs[kk + 1], dn1, 0, 0, 0);
            }
            dn += dn1;
            const int cj = jt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int frow = ft * 16 + (lane >> 4) + 4 * r;
                const int64_t f = f0 + frow;
                const bool live = f < p.s && cj < p.nb_prev;
                const float dnew = (float)dn[r];
                if (live) {
                    p.Dt[dfrag(f, p.j0_prev + cj, k)] = dnew;
                    p.Dt_out[(p.subset ? (int64_t)subr[u][r] : f) * p.kout + ocr[u]] = dnew;   // final: no unpack pass
                }
                Dl[frow * DLS + cj] = live ? dnew - dold[u][r] : 0.f;
            }
        }
    }
    if (fin) return;
    lds_barrier();                                                                    // ---- barrier 3
    if (st && tid == 0) st[5] = clock64();
    // ---------------------------------------------------------------- (E) rank-32 correction, cross-wave sum
    if (worker) {
        if (has_prev) {   // wave w contracts the previous block's atoms 8w .. 8w+7 (bq: loaded in (C))
            const int jb = wid * 8 + 4 * h;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const float4 dv = *reinterpret_cast<const float4 *>(Dl + (t * 32 + (lane & 31)) * DLS + jb);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv.x, bq[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv.y, bq[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv.z, bq[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv.w, bq[3], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, c = lane & 31;
                red[(wid * RB + row) * (kNB + 1) + c] = acc[t][r];
            }
    }
    lds_barrier();                                                                    // ---- barrier 4
    if (st && tid == 0) st[6] = clock64();
    // ---------------------------------------------------------------- (F) epilogue
    if (worker) {
        double d2 = 0;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int row = rg + 8 * q;
            const int64_t f = f0 + row;
            float val = 0.f;
            if (f < p.s && col_ok) {
                const float v = ((red[(0 * RB + row) * (kNB + 1) + col] + red[(1 * RB + row) * (kNB + 1) + col]) +
                                 red[(2 * RB + row) * (kNB + 1) + col]) + red[(3 * RB + row) * (kNB + 1) + col];
                val = fz ? eD[q] : (eB[q] - v) / cdg;
                p.a[f * kNB + col] = val;
                d2 += (double)eD[q] * (double)eD[q];
            }
            As[row * (kNB + 1) + col] = val;
        }
        d2red[rg * kNB + col] = d2;
    }
    lds_barrier();                                                                    // ---- barrier 5
    if (st && tid == 0) st[7] = clock64();
    // ---------------------------------------------------------------- (G) Gram record of this workgroup
    if (worker) {
        // tiles (0,0) (0,1) (1,1) of 16 x 16 (tile (1,0) is the transpose of (0,1)) on waves 0, 1 + 2, 3: both wave 1 and
        // wave 2 form tile (0,1), the only one without idle lanes, and add two of its four rows of entries each.  Every LDS
        // operand is requested before the first product and the contraction runs on two accumulators (as one dependent chain
        // of load - convert - product rounds the tile took 1.2 k cycles of the 4.7 k of this section).  What is left is the
        // compute unit's rate of 8-byte atomics and stores, ~1.3 cycles per lane: handing the 560 entries through LDS to all
        // 256 threads, two or three contiguous ones each, made the section LONGER (3.9 k against 3.2 k cycles).
        const int it = (wid == 3) ? 1 : 0, jt = (wid == 0) ? 0 : 1;
        const float *ai = As + (lane >> 4) * (kNB + 1) + it * 16 + (lane & 15);
        const float *aj = As + (lane >> 4) * (kNB + 1) + jt * 16 + (lane & 15);
        double *out = p.rec_out + (int64_t)blockIdx.x * kPackStride;
        long long *acc_out = p.acc_out ? p.acc_out + (size_t)((int)blockIdx.x & (p.shards - 1)) * kAccWords : nullptr;   // (1 or kAccShards = 4)
        float fa[RB / 4], fb[RB / 4];
#pragma unroll
        for (int kk = 0; kk < RB / 4; ++kk) {
            fa[kk] = ai[4 * kk * (kNB + 1)];
            fb[kk] = aj[4 * kk * (kNB + 1)];
        }
        double t = 0;
        if (wid == 2 && lane < kNB)
            for (int gq = 0; gq < 8; ++gq) t += d2red[gq * kNB + lane];
        d4v g = {0.0, 0.0, 0.0, 0.0}, g1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk += 2) {
            g = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk], (double)fb[kk], g, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk + 1], (double)fb[kk + 1], g1, 0, 0, 0);
        }
        g += g1;
        if (st && tid == 0) st[30] = clock64() + (unsigned long long)(g[0] * 0.0);
        bool bad = false;
        const int c15 = lane & 15;
        if (it != jt) {                                              // tile (0,1): full; wave 1 rows r = 0, 1, wave 2 rows 2, 3
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = (wid == 2) ? 2 + rr : rr;
                const int e = kTri + ((lane >> 4) + 4 * r) * 16 + c15;
                const double v = (wid == 2) ? (rr ? g[3] : g[2]) : (rr ? g[1] : g[0]);
                if (acc_out) acc_add(acc_out, e, v, false, bad);
                out[e] = v;
            }
            if (wid == 2 && lane < kNB) {                            // + the old squared norms
                if (acc_out) acc_add(acc_out, 2 * kTri + 256 + lane, t, true, bad);
                out[2 * kTri + 256 + lane] = t;
            }
        } else {                                                     // diagonal tiles: triangle
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) + 4 * r;
                if (row <= c15) {
                    const int e = (it ? kTri + 256 : 0) + tri_index(row, c15);
                    if (acc_out) acc_add(acc_out, e, g[r], row == c15, bad);
                    out[e] = g[r];
                }
            }
        }
        if (acc_out) acc_flag(acc_out, bad);
    }
    if (st && tid == 0) st[12] = clock64();
    if (st && tid == 64) st[33] = clock64();
    if (st && tid == 128) st[35] = clock64();
    if (st && tid == 192) st[34] = clock64();
    if (ngroups > 1) {   // pre-sum this group's records: the last workgroup of the group to arrive does it
        const int g = (int)blockIdx.x / gsz;
        const int gsize = (nwg - g * gsz < gsz) ? nwg - g * gsz : gsz;
        if (!arrive_last(p.counter + 1 + g, (unsigned int)gsize, flag)) return;
        const double *grecs = p.rec_out + (int64_t)g * gsz * kPackStride;
        const SinkGlobal gsink{p.grec_out + (int64_t)g * kPackStride};
        reduce_records_v2<kPackStride>(grecs, gsize, tid, true, gsink);       // (the record's tail: wave 4, above)
    }
}

static size_t bcd_block_lds(int gpw, int RT) {
    const int kpad = gpw * 32, RB = 32 * RT;
    const size_t dbl = (size_t)kNB * 64 + kNB + kNB * kNB + (size_t)kNB * kCaStride + 8 * kNB;
    const size_t fl = 4 * (size_t)RB * (kNB + 1) + RB * (kNB + 1) + 4 + RB * (kNB + 4) + 4;
    const size_t mail = (size_t)kNB * kNB + 2 * kMbox * 64;     // Cs transposed + the two mailboxes of the recursion
    (void)kpad;
    return dbl * 8 + fl * 4 + 16 + mail * 8 + 16;
}

// ---------------------------------------------------------------- generic path
template <typename T, int KPL>
__global__ __launch_bounds__(256) void atom_grad_kernel(const T *Dt, const T *Bt, const T *C, const int32_t *subset,
                                                        int64_t s, int k, int j, int pos, double rho, T *u,
                                                        double *partial_old) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int e0 = lane * KPL;
    T cc[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {                                     // row j == column j (unconditional, clamped loads)
        const T cv = C[(int64_t)j * k + (e0 + c < k ? e0 + c : k - 1)];
        cc[c] = (e0 + c < k) ? cv : (T)0;
    }
    const T Cjj = C[(int64_t)j * k + j];
    const bool frozen = !(Cjj > (T)1e-20);
    double old = 0;
    for (int64_t f = (int64_t)blockIdx.x * 4 + wid; f < s; f += (int64_t)gridDim.x * 4) {
        const T *row = Dt + sub_row(subset, f) * k;
        double dot = 0;
#pragma unroll
        for (int c = 0; c < KPL; ++c) dot += (double)row[e0 + c < k ? e0 + c : k - 1] * (double)cc[c];   // cc = 0 beyond k
        dot = wave_sum(dot);
        if (lane == 0) {
            const T dj = row[j];
            T val = dj;
            if (!frozen) val = (T)((((double)Bt[sub_row(subset, f) * k + j] - dot) + (double)Cjj * (double)dj) / (double)Cjj);
            if (pos && val < (T)0) val = 0;            // dict_fact.py:684-685
            u[f] = val;
            const double a = fabs((double)dj);
            old += a * (rho + (1.0 - rho) * a);
        }
    }
    old = block_sum(old, red);
    if (threadIdx.x == 0) partial_old[blockIdx.x] = old;
}

// The projection of one atom, by one workgroup: old-norm partials -> radius, Michelot, write-back.
// Up to kProjEpt * 256 elements it is run by the FIRST FOUR wavefronts with the vector in registers (the caller
// retires the other threads first: see block_enet_project_reg); beyond that every thread takes part and the
// passes scan the vector (from its LDS copy `ul` when given, else from L2).  red: >= 32 doubles.
constexpr int kProjEpt = 24;
template <typename T, bool REGS = true>   // REGS = false: no register-resident projection (callers beyond its reach: 1024-thread workgroups)
__device__ __forceinline__ void atom_project(T *u, T *ul, const double *partial_old, int nparts, T *Dt,
                                             const int32_t *subset, int64_t s, int k, int j, double rho, T *comp_norm,
                                             double *red, unsigned long long *dbg = nullptr, double *level_hint = nullptr,
                                             T *stage_out = nullptr) {
    const bool in_regs = REGS && s <= (int64_t)kProjEpt * 256 && blockDim.x >= 256;
    const int nthr = in_regs ? 256 : (int)blockDim.x;
    if ((int)threadIdx.x >= nthr) return;
    double old = 0, dummy = 0;
    for (int i = threadIdx.x; i < nparts; i += nthr) old += partial_old[i];
    T *w = u;
    constexpr int NQ = 8;                                            // elements per thread in flight (a dependent load - store loop
    if (!in_regs && ul) {                                            //  took one round trip per element: 40 of them at 10 000 features)
        for (int64_t f0 = 0; f0 < s; f0 += (int64_t)NQ * nthr) {
            T v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + (int64_t)q * nthr;
                v[q] = u[f < s ? f : s - 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + (int64_t)q * nthr;
                if (f < s) ul[f] = v[q];
            }
        }
        w = ul;
    }
    block_sum2(old, dummy, red, nthr);                               // (also orders the LDS copy)
    const double radius = (double)(T)((double)comp_norm[j] + old);   // comp_norm_[k] += subset_norm (:676-678)
    if (dbg && threadIdx.x == 0) dbg[7] = clock64();
    double nrm;
    if (REGS && in_regs) {                                           // projected values go straight to the dictionary
        nrm = block_enet_project_reg<T, kProjEpt>(u, Dt + j, subset, (int64_t)k, s, radius, rho, red, nthr, dbg,
                                                  level_hint ? level_hint + j : nullptr);
    } else {
        if (ul && rho == 1.0)                                        // (l1 ball, vector in LDS: fused sums, warm start - same bits)
            nrm = block_l1_project_inplace<T>(ul, s, radius, red, nthr, level_hint ? level_hint + j : nullptr, dbg);
        else
            nrm = block_enet_project<T>(w, 1, w, 1, s, radius, rho, red, dbg);
        __syncthreads();
        if (dbg && threadIdx.x == 0) dbg[5] = clock64();
        if (stage_out && ul) {
            // the projected atom leaves as a COMPACT row: the workgroups of the NEXT atom's launch put its values where they
            // belong while they read their dictionary rows anyway (atom_grad4_kernel: stage_prev) - written from here they are
            // 10 000 scattered 4-byte stores by one workgroup, 42 k of the 160 k cycles an atom costs at config 6's shape
            typedef __attribute__((address_space(3))) T lds_T;
            for (int64_t f = threadIdx.x; f < s; f += nthr) stage_out[f] = ((lds_T *)ul)[f];
        } else
        for (int64_t f0 = 0; f0 < s; f0 += (int64_t)NQ * nthr) {      // (the row indices of a batch in one round trip)
            int64_t row[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + (int64_t)q * nthr;
                row[q] = sub_row(subset, f < s ? f : s - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            typedef __attribute__((address_space(3))) T lds_T;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t f = f0 + threadIdx.x + (int64_t)q * nthr;
                if (f < s) Dt[row[q] * k + j] = ul ? (T)((lds_T *)ul)[f] : w[f];
            }
        }
    }
    if (threadIdx.x == 0) comp_norm[j] = (T)(radius - nrm);          // :690-692
}

template <typename T>
__global__ __launch_bounds__(1024) void atom_project_kernel(T *u, const double *partial_old, int nparts, T *Dt,
                                                            const int32_t *subset, int64_t s, int k, int j,
                                                            double rho, T *comp_norm) {
    __shared__ double red[32];
    atom_project<T>(u, nullptr, partial_old, nparts, Dt, subset, s, k, j, rho, comp_norm, red);
}

// One atom in ONE launch: every workgroup evaluates its share of the gradient row (a wavefront per sampled
// feature, coalesced), the LAST workgroup to finish (release / relaxed ticket / acquire, no spinning) projects.
template <typename T, int KPL>
__global__ __launch_bounds__(256) void atom_step_kernel(T *Dt, const T *Bt, const T *C, const int32_t *subset, int64_t s,
                                                        int k, int j, int pos, double rho, T *u, double *partial_old,
                                                        T *comp_norm, unsigned int *counter, int u_in_lds,
                                                        unsigned long long *dbg, double *level_hint) {
    extern __shared__ __attribute__((aligned(16))) char step_smem[];   // the s-vector for the projection, if it fits
    __shared__ double red[32];
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long t0 = clock64();
    // lane l takes elements l, l + 64, ...: a wavefront's load is 256 contiguous bytes (with KPL consecutive elements per
    // lane the KPL loads of a row walked the same cache lines KPL times and counted on the L1 to keep them: four rows of
    // eight wavefronts in flight evicted each other - the batched rows below were SLOWER that way, 89 k against 69 k cycles)
    T cc[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {                                     // row j == column j (unconditional, clamped loads)
        const int e = lane + 64 * c;
        const T cv = C[(int64_t)j * k + (e < k ? e : k - 1)];
        cc[c] = (e < k) ? cv : (T)0;
    }
    const T Cjj = C[(int64_t)j * k + j];
    const bool frozen = !(Cjj > (T)1e-20);
    double old = 0;
    const int nwv = (int)(blockDim.x >> 6);
    // A wavefront's rows in batches of kStepRows: the batch's row indices in one round trip, every load of its rows in a
    // second one, then the reductions (a row at a time - index, then row, then the next index - a wavefront had ONE 4 KB row
    // in flight and ten dependent round trips at config 6's shape, 10 000 rows of 1024 atoms over 2048 wavefronts: 29 us of
    // the 86 an atom costs there, 1.4 TB/s).  Same sums in the same order.
    constexpr int kStepRows = 3;                                         // (3 x (KPL + 2) loads in flight: under the 6-bit counter at KPL = 16)
    const int64_t fstride = (int64_t)gridDim.x * nwv;
    for (int64_t fb = (int64_t)blockIdx.x * nwv + wid; fb < s; fb += fstride * kStepRows) {
        int64_t r[kStepRows];
#pragma unroll
        for (int q = 0; q < kStepRows; ++q) {
            const int64_t f = fb + q * fstride;
            r[q] = sub_row(subset, f < s ? f : fb) * k;                  // (clamped to the batch's first row: no branch around a load)
        }
        T rv[kStepRows][KPL], djv[kStepRows], bjv[kStepRows];
#pragma unroll
        for (int q = 0; q < kStepRows; ++q) {
            const T *row = Dt + r[q];
            djv[q] = row[j];
            bjv[q] = Bt[r[q] + j];
#pragma unroll
            for (int c = 0; c < KPL; ++c) rv[q][c] = row[lane + 64 * c < k ? lane + 64 * c : k - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < kStepRows; ++q) {
            const int64_t f = fb + q * fstride;
            if (f >= s) break;                                           // (wavefront-uniform)
            double dot = 0;
#pragma unroll
            for (int c = 0; c < KPL; ++c) dot += (double)rv[q][c] * (double)cc[c];   // cc = 0 beyond k
            dot = wave_sum(dot);
            if (lane == 0) {
                const T dj = djv[q], bj = bjv[q];
                T val = dj;
                if (!frozen) val = (T)((((double)bj - dot) + (double)Cjj * (double)dj) / (double)Cjj);
                if (pos && val < (T)0) val = 0;            // dict_fact.py:684-685
                u[f] = val;
                const double a = fabs((double)dj);
                old += a * (rho + (1.0 - rho) * a);
            }
        }
    }
    old = block_sum(old, red);
    if (threadIdx.x == 0) partial_old[blockIdx.x] = old;
    const unsigned long long t1 = clock64();
    if (!arrive_last(counter, gridDim.x, &flag)) return;
    const unsigned long long t2 = clock64();
    atom_project<T>(u, u_in_lds ? reinterpret_cast<T *>(step_smem) : nullptr, partial_old, (int)gridDim.x, Dt, subset, s, k, j,
                    rho, comp_norm, red, dbg, level_hint);
    if (dbg && threadIdx.x == 0) { dbg[0] = t0; dbg[1] = t1; dbg[2] = t2; dbg[3] = clock64(); }
}

template <int G> struct AtomGroupN { int j[G]; int n; };

// ---- The one-launch-per-atom path for GROUPS of kStepGroup consecutive atoms of the sweep, sampled sets beyond the register-resident
// projection (more than 6144 features: the reference's HCP run samples 10 000 of 200 000 for 1024 atoms).  The gradient row
// is what an atom costs there - 41 MB of dictionary per atom - so the launch of a group's FIRST atom evaluates the
// numerators of all its atoms against the dictionary as it is then (one read serves the group; they are kept in double
// with the old values), and the launches of the other atoms only correct theirs for what the atoms before them in the
// group changed, from compact rows:
//     num_a[f] -= sum_{a' < a} C[j_a', j_a] (D_new[j_a'][f] - D_old[j_a'][f])
// (the difference between the sequential sweep's gradient row and the stale one - the grouped atom update's identity, below).
// Every atom's launch ends like atom_step_kernel: the last workgroup to arrive projects from LDS - and leaves a compact
// row; the gradient launch of the NEXT group puts the group's rows where they belong while it reads the dictionary.
constexpr int kStepGroup = 4;
// (1) the group's gradient rows: numerators and old values of all its atoms, the old-norm partial sums, and the rows of
// the group BEFORE put where they belong
// vblock of nvblocks: the workgroup's place in the gradient launch's grid (a launch of its own: blockIdx.x of gridDim.x; riding on
// the launches of the group before, round 6: a quarter of the virtual grid per launch)
template <typename T, int KPL>
__device__ __forceinline__ void atom_grad4_body(T *Dt, const T *Bt, const T *C, const int32_t *subset, int64_t s, int k,
                                                const AtomGroupN<kStepGroup> &g, const AtomGroupN<kStepGroup> &gp, const T *stage_prev,
                                                double rho, double *num, T *dold, int64_t ldr, double *partial_old,
                                                int part_stride, int vblock, int nvblocks, double (*s_oldw)[kStepGroup]) {
    constexpr int G = kStepGroup;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int ja[G];
    T cc[G][KPL], Cjj[G];
#pragma unroll
    for (int b = 0; b < G; ++b) {
        ja[b] = g.j[b < g.n ? b : 0];                               // (a short last group repeats its first atom: not stored)
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = lane + 64 * c;
            const T cv = C[(int64_t)ja[b] * k + (e < k ? e : k - 1)];
            cc[b][c] = (e < k) ? cv : (T)0;
        }
        Cjj[b] = C[(int64_t)ja[b] * k + ja[b]];
    }
    double old = 0;                                                 // lane b: atom b
    const int nwv = (int)(blockDim.x >> 6);
    constexpr int kRows = 2;                                        // (2 x (KPL + 3 G) loads in flight at KPL = 16)
    const int64_t fstride = (int64_t)nvblocks * nwv;
    for (int64_t fb = (int64_t)vblock * nwv + wid; fb < s; fb += fstride * kRows) {
        int64_t r[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const int64_t f = fb + q * fstride;
            r[q] = sub_row(subset, f < s ? f : fb) * k;
        }
        T rv[kRows][KPL], dj[kRows][G], bj[kRows][G], pv[kRows][G];
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const T *row = Dt + r[q];
            const int64_t f = fb + q * fstride;
#pragma unroll
            for (int c = 0; c < KPL; ++c) rv[q][c] = row[lane + 64 * c < k ? lane + 64 * c : k - 1];
#pragma unroll
            for (int b = 0; b < G; ++b) {
                dj[q][b] = row[ja[b]];
                bj[q][b] = Bt[r[q] + ja[b]];
                pv[q][b] = stage_prev[(int64_t)(b < gp.n ? b : 0) * ldr + (f < s ? f : 0)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const int64_t f = fb + q * fstride;
            if (f >= s) break;                                       // (wavefront-uniform)
            T pmine = 0;                                             // lane b: the staged value of the previous group's atom b
            int jmine = -1;
#pragma unroll
            for (int b = 0; b < G; ++b) {
                if (b < gp.n) {                                      // the previous group's atoms: into the row, and home
#pragma unroll
                    for (int c = 0; c < KPL; ++c) rv[q][c] = (lane + 64 * c == gp.j[b]) ? pv[q][b] : rv[q][c];
                    if (lane == b) { pmine = pv[q][b]; jmine = gp.j[b]; }
                }
            }
            if (jmine >= 0) Dt[r[q] + jmine] = pmine;
            double nmine = 0;
            T dmine = 0;
#pragma unroll
            for (int b = 0; b < G; ++b) {
                double dot = 0;
#pragma unroll
                for (int c = 0; c < KPL; ++c) dot += (double)rv[q][c] * (double)cc[b][c];   // cc = 0 beyond k
                dot = wave_sum(dot);
                const double nv = ((double)bj[q][b] - dot) + (double)Cjj[b] * (double)dj[q][b];
                if (lane == b) { nmine = nv; dmine = dj[q][b]; }
            }
            if (lane < g.n) {
                num[(int64_t)lane * ldr + f] = nmine;
                dold[(int64_t)lane * ldr + f] = dmine;
                const double ab = fabs((double)dmine);
                old += ab * (rho + (1.0 - rho) * ab);
            }
        }
    }
    if (lane < G) s_oldw[wid][lane] = old;
    __syncthreads();
    if (threadIdx.x < G)
        partial_old[(int64_t)threadIdx.x * part_stride + vblock] =
            (s_oldw[0][threadIdx.x] + s_oldw[1][threadIdx.x]) + (s_oldw[2][threadIdx.x] + s_oldw[3][threadIdx.x]);
}

template <typename T, int KPL>
__global__ __launch_bounds__(256) void atom_grad4_kernel(T *Dt, const T *Bt, const T *C, const int32_t *subset, int64_t s, int k,
                                                         AtomGroupN<kStepGroup> g, AtomGroupN<kStepGroup> gp, const T *stage_prev,
                                                         double rho, double *num, T *dold, int64_t ldr, double *partial_old,
                                                         int part_stride) {
    __shared__ double s_oldw[4][kStepGroup];
    atom_grad4_body<T, KPL>(Dt, Bt, C, subset, s, k, g, gp, stage_prev, rho, num, dold, ldr, partial_old, part_stride, (int)blockIdx.x,
                            (int)gridDim.x, s_oldw);
}

// The gradient rows of the NEXT group riding on a launch of this group's atoms (round 6, the pipelined sweep below): workgroups
// beyond the launch's own do the part [vb0, vb0 + their number) of that group's gradient launch.
template <typename T> struct GradRide {
    const T *Bt;
    AtomGroupN<kStepGroup> gn, gfold;     // the group whose numerators are formed; the finished group whose rows are put home on the way
    const T *stage_fold;
    double *num;
    T *dold;
    double *pold;
    int part_stride, vb0, nvb, nride;     // nride riders do the workgroups [vb0, vb0 + nride) of the gradient launch's nvb
    int ept;                              // elements per thread of the launch's own workgroups (1, or 2: half as many workgroups)
    int regs;                             // 1: no spread projection - the launch's last workgroup projects from registers (atom_project_regs)
};

// (2) one atom of the group: its candidate from the stored numerator minus what the atoms before it in the group changed (a
// thread per feature), then the last workgroup to arrive projects from LDS (atom_project) and leaves the compact row.
// (256 threads: the projection is 22 of the 24 us of such a launch, scans of the 10 000-element vector and block-wide
// sums - and with 1024 threads, sixteen wavefronts, it took 29: the sums and their barriers grow faster than the scans
// shrink.)
// ---- the l1 projection of a vector SPREAD OVER THE LAUNCH'S WORKGROUPS (round 6; the reference's HCP configuration: 10 000 sampled
// features, 1024 positive l1 atoms, one launch per atom).  The launch's last workgroup used to project the whole vector alone from
// LDS - 22 of the launch's 24 us, 40 elements per thread and Michelot pass.  Here every thread keeps ITS element in a register and a
// pass is: the workgroup's two sums (wave sums, LDS), written through to a slot of this pass, one lane per workgroup polls the
// slots (sentinels: the data is its own flag, bcd_persist.hip), wave sum, broadcast - one memory round trip per pass instead of a
// scan of the vector.  Michelot's iteration with the closed form of enet.pyx:119, warm-started at the atom's last level (the
// register-resident projection's rule, enet_project_slim), the sums in a fixed order: run-to-run identical.
// MEASURED at the HCP shape (39 workgroups, scripts/diag_atom_stamps_c6.py): an exchange costs 5.5 k cycles (2.3 us: a write-through
// store, its way to memory, a polled load); a two-minibatch-old dictionary needs ten of them per atom, the launch 21.5 us against
// 24.3 with the last workgroup's projection (minibatch 35.7 -> 33.9 ms).  Not kept: EIGHT levels per exchange (every Newton step
// is a lower bound, every level with f <= 0 an upper one; the next exchange brackets between them) - five exchanges instead of
// ten, but 11.7 k cycles each (sixteen wave sums on either side of the round trip, sixteen polled words per lane): 39.3 ms.
// Every wait is bounded; a wait that gives up (a workgroup of the launch that is not resident: another process on the GPU) raises
// the launch's abort word and the launch ends as it always has - every workgroup has left its candidates in `u`, the last one to
// arrive projects them alone: nothing has been written that this would not overwrite (the budget and the level hint are written
// by workgroup 0 only once EVERY workgroup has delivered its last sum).
constexpr int kMwgMaxWg = 64, kMwgMaxPass = 23;
constexpr int kMwgWords = (kMwgMaxPass + 1) * 2 * kMwgMaxWg;          // doubles of one exchange buffer (two in rotation)
constexpr unsigned int kMwgSentinel32 = 0x7ff8deadu;                  // both halves of the sentinel (a NaN): hipMemsetD32Async fills it
constexpr long long kMwgSentinel = ((long long)kMwgSentinel32 << 32) | kMwgSentinel32;
std::atomic<int> g_atom_mwg{1};                                       // modl_debug_set(MODL_DEBUG_ATOM_MWG, ...)
std::atomic<int> g_atom_pipe{1};                                      // modl_debug_set(MODL_DEBUG_ATOM_PIPE, ...): 0 = a gradient launch per group

__device__ __forceinline__ void mwg_store(double *ptr, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ptr), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double mwg_load(const double *ptr) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
struct MwgPlace { int wg, nwg; };                               // this workgroup's place among the nwg that take part (a launch may carry riders)

// the launch-wide sums of (S, cnt); false: a wait gave up.  red: >= 24 doubles of LDS.  Called by every thread (256).
__device__ __forceinline__ bool mwg_sum2(double &S, double &cnt, double *xb, int slot, unsigned int *abort_word, double *red,
                                         const MwgPlace &pl, bool withhold = false) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wg = pl.wg, nwg = pl.nwg;                          // (a launch may carry riders: not gridDim.x)
    S = wave_sum(S);
    cnt = wave_sum(cnt);
    if (lane == 0) { red[2 * wid] = S; red[2 * wid + 1] = cnt; }
    __syncthreads();
    if (wid == 0) {
        double *sl = xb + (size_t)slot * 2 * kMwgMaxWg;
        if (lane == 0 && !withhold) {                            // (withhold: diagnostics build - the workgroup that never delivers)
            mwg_store(sl + 2 * wg, (red[0] + red[2]) + (red[4] + red[6]));
            mwg_store(sl + 2 * wg + 1, (red[1] + red[3]) + (red[5] + red[7]));
        }
        double gs = 0.0, gc = 0.0;
        bool ok = true;
        if (lane < nwg) {
            for (unsigned spins = 0;; ++spins) {
                gs = mwg_load(sl + 2 * lane);
                gc = mwg_load(sl + 2 * lane + 1);
                if (__double_as_longlong(gs) != kMwgSentinel && __double_as_longlong(gc) != kMwgSentinel) break;
                if (spins > (1u << 17) || ((spins & 63) == 63 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const bool all_ok = __all(ok);
        gs = wave_sum((lane < nwg && all_ok) ? gs : 0.0);
        gc = wave_sum((lane < nwg && all_ok) ? gc : 0.0);
        if (lane == 0) {
            if (!all_ok) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[16] = gs; red[17] = gc; red[18] = all_ok ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    S = red[16];
    cnt = red[17];
    const bool ok = red[18] != 0.0;
    __syncthreads();
    return ok;
}
// x: this thread's element (0 beyond the vector).  On success: out = the projected element, nrm = the l1 norm of the projected
// vector (every workgroup has it), level_out / searched = the level the search ended at.
template <typename T>
__device__ __forceinline__ bool mwg_l1_project(T x, T x1, double radius, double l_prev, double *xb, unsigned int *abort_word, double *red,
                                               T &out, T &out1, double &nrm, double &level_out, bool &searched, bool withhold, MwgPlace pl) {
    // (x1, out1: a second element of this thread - zero when it has one; a zero adds 0.0 to every sum: the same bits)
    searched = false;
    level_out = 0.0;
    if (!(radius > 0.0)) {                                   // enet.pyx:57-59 (radius == 0 -> zeros); the same on every workgroup
        out = 0;
        out1 = 0;
        nrm = 0.0;
        return true;
    }
    const double R = radius, a = fabs((double)x), a1 = fabs((double)x1);
    int slot = 0;
    double S, cnt, level = 0.0, prev_cnt = -1.0;
    bool warm = false, inside = false;
    auto scan = [&](double lv) {
        S = (a > lv ? a : 0.0) + (a1 > lv ? a1 : 0.0);
        cnt = (a > lv ? 1.0 : 0.0) + (a1 > lv ? 1.0 : 0.0);
        return mwg_sum2(S, cnt, xb, slot++, abort_word, red, pl, withhold);
    };
    // warm start: the level the atom ended with at the previous minibatch ITSELF - f(l) = sum_{|x| > l} (|x| - l) - R is convex and
    // decreasing, so the Newton step that Michelot's update is lands at or left of the root from EITHER side and the iteration
    // rises monotonically from there (enet_project_slim: no verification pass, no safety factor); a step that lands at or below
    // zero proves nothing and falls back to the cold start
    if (l_prev > 0.0 && l_prev < 1e300) {
        if (!scan(l_prev)) return false;
        const double l1 = (S - R) / cnt;
        if (cnt != 0.0 && l1 > 0.0) { warm = true; prev_cnt = cnt; level = l1; }
    }
    if (!warm) {
        if (!scan(0.0)) return false;
        if (S <= R) inside = true;                           // inside the ball: nothing moves
        else if (cnt != 0.0) { prev_cnt = cnt; level = (S - R) / cnt; }
    }
    double total = S;
    if (!inside) {
        searched = true;
        for (;;) {
            if (cnt == 0.0) break;
            if (slot >= kMwgMaxPass) return false;           // (never seen: the old path takes any number of passes)
            if (!scan(level)) return false;
            if (cnt == prev_cnt || cnt == 0.0) break;
            prev_cnt = cnt;
            level = (S - R) / cnt;                           // enet.pyx:119
        }
        const double lT = (double)(T)level;
        double pos = a - lT, pos1 = a1 - lT;
        pos = pos > 0 ? pos : 0;
        pos1 = pos1 > 0 ? pos1 : 0;
        out = (T)(((double)x >= 0) ? pos : -pos);            // enet.pyx:121, sign(0) = +1
        out1 = (T)(((double)x1 >= 0) ? pos1 : -pos1);
        double mine = fabs((double)out) + fabs((double)out1), dummy = 0.0;
        if (!mwg_sum2(mine, dummy, xb, kMwgMaxPass, abort_word, red, pl)) return false;
        total = mine;
    } else {
        out = x;
        out1 = x1;
        double mine = 0.0, dummy = 0.0;                      // (the last slot is the "everybody is done" exchange in every case)
        if (!mwg_sum2(mine, dummy, xb, kMwgMaxPass, abort_word, red, pl)) return false;
    }
    nrm = total;
    level_out = level;
    if (threadIdx.x == 0) red[19] = (double)slot;            // (diagnostics: exchanges of the search)
    return true;
}

// Round 6, the PIPELINED sweep (KPL > 0): the gradient launch of group g + 1 (25 us: one read of the 41 MB sampled dictionary)
// used to sit between the last atom of group g and the first of group g + 1, with 217 of the 256 compute units idle during the
// 4 x 22 us of atom launches around it.  Now it RIDES on those launches: nride workgroups beyond the launch's own nwg_main do a
// quarter of it each (atom_grad4_body on a part of its virtual grid).  They read the dictionary as it is while group g is still
// being projected, so what group g changes is missing from group g + 1's numerators as well: the launches of group g + 1
// subtract it with the same identity that handles their own group, from the PREVIOUS group's compact rows (gprev, stage_prev,
// dold_prev).  The riders put the rows of group g - 1 home (the newest finished one), so the host flushes the last TWO groups.
// (defined with the grouped atom update below)
template <typename T, int EPT, bool L1>
__device__ __forceinline__ double enet_project_slim(double (&x)[EPT], T *out, double radius, double l1_ratio, double *red4, int &par,
                                                    double l_prev, double *level_out, unsigned long long *dbg);
__device__ __forceinline__ void block_sum1_pp(double &a, double *red4, int &par);
__device__ __forceinline__ void block_sum2_pp(double &a, double &b, double *red4, int &par);
template <int EPT> __device__ __forceinline__ void store_row4(double *row, const double (&v)[EPT]);
template <int EPT> __device__ __forceinline__ void store_row4(float *row, const float (&v)[EPT]);
template <int EPT> __device__ __forceinline__ void load_row4(const double *row, double (&v)[EPT]);
template <int EPT> __device__ __forceinline__ void load_row4(const float *row, float (&v)[EPT]);

__device__ __forceinline__ void store_wt(float *ptr, float v) {       // a write-through store (a relaxed agent-scope atomic store IS one)
    __hip_atomic_store(reinterpret_cast<unsigned int *>(ptr), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt(double *ptr, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ptr), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// load_row4 past the caches (16-byte sc1 loads): the row was written by OTHER workgroups of this launch, write-through
template <typename T, int EPT>
__device__ __forceinline__ void load_row4_sc1(const T *row, T (&v)[EPT]) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    constexpr int PER = 16 / (int)sizeof(T);                         // elements per 16-byte load
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(row), 0, EPT * 256 * (int)sizeof(T), 0x00020000);
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q)
#pragma unroll
        for (int h = 0; h < 4 / PER; ++h) {
            const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)sizeof(T) * (4 * (int)threadIdx.x + 1024 * q + PER * h), 0, 16);   // (aux 16: sc1)
            __builtin_memcpy(&v[4 * q + PER * h], &raw, 16);
        }
}

// block_sum2_pp for a sum and a small COUNT (per thread at most 64: a wavefront's total is exact in f32, whose lane exchanges are
// one instruction a stage instead of three).  (Measured and not kept: the wavefront's count as scalar population counts of the
// compares - the scalar unit waits for every compare: 2.3 k cycles per pass instead of 2.1 k; the first passes with f32 sums,
// guarded by a relative 4e-6 - a pass costs the same, its instruction count is not what bounds it.)
__device__ __forceinline__ void block_sum_count_pp(double &a, double &cnt, double *red4, int &par) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    a = wave_sum(a);
    const float c = wave_sum((float)cnt);
    double *r = red4 + 8 * par;
    par ^= 1;
    if (lane == 0) { r[2 * wid] = a; r[2 * wid + 1] = (double)c; }
    __syncthreads();
    a = (r[0] + r[2]) + (r[4] + r[6]);
    cnt = (r[1] + r[3]) + (r[5] + r[7]);
}

// The l1 projection of the launch's candidates by its LAST workgroup with the whole vector in REGISTERS (round 6; EPT = 40 or 64
// elements per thread: up to 16 384 sampled features): a Michelot pass is EPT compare-select-adds and one LDS exchange, ~1.7 k
// cycles, against a 2 us round trip through memory per pass of the spread projection and 5-7 k cycles per scan of the LDS copy.
// Writes the compact row (whole 1024-element chunks: the rows of the pipelined sweep are padded to that), the budget, the level.
template <typename T, int EPT>
__device__ __forceinline__ void atom_project_regs(const T *u, int64_t s, int j, const double *partial_old, int nparts, T *comp_norm,
                                                  double *level_hint, T *stage_row, double *red, unsigned long long *dbg) {
    double old = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) old += partial_old[i];
    T v[EPT], ax[EPT];                                               // the candidates and their magnitudes, in T: a register each in f32
    load_row4_sc1<T, EPT>(u, v);                                     // (beyond s: whatever follows in the workspace, masked)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        v[e] = (4 * (int)threadIdx.x + (e & 3) + 1024 * (e >> 2) < s) ? v[e] : (T)0;
        ax[e] = v[e] < (T)0 ? -v[e] : v[e];
    }
    int par = 0;
    block_sum1_pp(old, red, par);
    const double radius = (double)(T)((double)comp_norm[j] + old);   // comp_norm_[k] += subset_norm (:676-678)
    if (dbg && threadIdx.x == 0) dbg[7] = clock64();
    // Michelot's iteration with the closed form of enet.pyx:119 (enet_project_slim's rules: the previous level ITSELF as the warm
    // start - the Newton step lands at or left of the root from either side -, the cold start settles "inside the ball").  A
    // pass compares in T: for a T-representable magnitude, a > level (in double) <=> a > the level rounded DOWN to T.
    const bool zero = !(radius > 0.0);                               // enet.pyx:57-59 (radius == 0 -> zeros)
    const double R = radius;
    auto pass = [&](double lv, double &S, double &cnt) {
        T lt = (T)lv;
        if ((double)lt > lv) lt = sizeof(T) == 4 ? (T)__uint_as_float(__float_as_uint((float)lt) - 1u) : (T)__longlong_as_double(__double_as_longlong((double)lt) - 1ll);   // (lv > 0 here)
        double S0 = 0, S1 = 0, S2 = 0, S3 = 0;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {                           // selects, no branches; four chains
            const bool i0 = ax[e] > lt, i1 = ax[e + 1] > lt, i2 = ax[e + 2] > lt, i3 = ax[e + 3] > lt;
            S0 += (double)(i0 ? ax[e] : (T)0);
            S1 += (double)(i1 ? ax[e + 1] : (T)0);
            S2 += (double)(i2 ? ax[e + 2] : (T)0);
            S3 += (double)(i3 ? ax[e + 3] : (T)0);
            c0 += (i0 ? 1 : 0) + (i2 ? 1 : 0);
            c1 += (i1 ? 1 : 0) + (i3 ? 1 : 0);
        }
        S = (S0 + S1) + (S2 + S3);
        cnt = (double)(c0 + c1);
        block_sum_count_pp(S, cnt, red, par);
    };
    double level = 0.0, prev_cnt = -1.0;
    bool warm = false, search = !zero;
    int npass = 0;
    const double l_prev = level_hint ? level_hint[j] : 0.0;
    if (search && l_prev > 0.0 && l_prev < 1e300) {
        double S, cnt;
        pass(l_prev, S, cnt);
        ++npass;
        const double l1 = (S - R) / cnt;
        if (cnt != 0.0 && l1 > 0.0) { warm = true; prev_cnt = cnt; level = l1; }
    }
    if (search && !warm) {
        double tot = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) tot += (double)ax[e];
        block_sum1_pp(tot, red, par);
        if (tot <= R) search = false;                                // inside the ball: level 0 is the identity
    }
    if (search) {
        for (int p = 0; p < 256; ++p) {
            double S, cnt;
            if (level > 0.0) pass(level, S, cnt);
            else {                                                   // (level 0 - the cold start's first pass: every non-zero counts)
                double S0 = 0;
                int c0 = 0;
#pragma unroll
                for (int e = 0; e < EPT; ++e) { S0 += (double)ax[e]; c0 += ax[e] > (T)0 ? 1 : 0; }
                S = S0; cnt = (double)c0;
                block_sum_count_pp(S, cnt, red, par);
            }
            ++npass;
            if (cnt == prev_cnt || cnt == 0.0) break;
            prev_cnt = cnt;
            level = (S - R) / cnt;                                   // enet.pyx:119
        }
    }
    if (dbg && threadIdx.x == 0) { dbg[4] = (unsigned)npass | (warm ? 1u << 16 : 0u); dbg[5] = clock64(); }
    if (level_hint && threadIdx.x == 0 && search) level_hint[j] = level;
    const double lT = (double)(T)level;
    double mine = 0;
    T o[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        double pos = (double)ax[e] - lT;
        pos = pos > 0 ? pos : 0;
        o[e] = (T)((v[e] >= (T)0) ? pos : -pos);                     // enet.pyx:121, sign(0) = +1
        o[e] = zero ? (T)0 : o[e];
        mine += fabs((double)o[e]);
    }
    store_row4<EPT>(stage_row, o);
    block_sum1_pp(mine, red, par);
    if (threadIdx.x == 0) comp_norm[j] = (T)(radius - mine);         // :690-692
}

template <typename T, int KPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void atom_corr_project_kernel(T *Dt, const T *C, const int32_t *subset, int64_t s, int k,
                                                                AtomGroupN<kStepGroup> g, int a, int pos, double rho, T *u,
                                                                const double *num, const T *dold, T *stage_cur, int64_t ldr,
                                                                const double *partial_old, int part_stride, T *comp_norm,
                                                                unsigned int *counter, unsigned long long *dbg, double *level_hint,
                                                                double *xch, unsigned int *xabort, int parity, int nwg_main,
                                                                AtomGroupN<kStepGroup> gprev, const T *stage_prev, const T *dold_prev,
                                                                GradRide<T> ride) {
    constexpr int G = kStepGroup;
    extern __shared__ __attribute__((aligned(16))) char step_smem[];   // the s-vector for the projection
    __shared__ double red[32];
    __shared__ int flag;
    const int wg = (int)blockIdx.x;                                  // one of the launch's own nwg_main workgroups, or a rider behind them
    if constexpr (KPL > 0) {
        __shared__ double s_oldw[4][G];
        if (wg >= nwg_main) {
            if (wg - nwg_main < ride.nride)
                atom_grad4_body<T, KPL>(Dt, ride.Bt, C, subset, s, k, ride.gn, ride.gfold, ride.stage_fold, rho, ride.num, ride.dold, ldr,
                                        ride.pold, ride.part_stride, ride.vb0 + wg - nwg_main, ride.nvb, s_oldw);
            return;
        }
    }
    const unsigned long long t0 = clock64();
    const int j = g.j[a];
    if (xch) {
        // ---- the projection spread over the launch's workgroups (a thread per feature: gridDim.x * 256 >= s, at most 64 workgroups)
        double *xb = xch + (size_t)(parity & 1) * kMwgWords;
        {   // the buffer of the NEXT launch back to sentinels (its last user was the launch before this one)
            long long *xo = reinterpret_cast<long long *>(xch + (size_t)((parity & 1) ^ 1) * kMwgWords);
            const int per = (kMwgWords + nwg_main - 1) / nwg_main;
            for (int e = wg * per + threadIdx.x; e < (wg + 1) * per && e < kMwgWords; e += 256) xo[e] = kMwgSentinel;
        }
        const double cjj = (double)C[(int64_t)j * k + j];
        const bool frozen = !((T)cjj > (T)1e-20);
        double cb[G], cp[G];
#pragma unroll
        for (int b = 0; b < G; ++b) cb[b] = (b < a) ? (double)C[(int64_t)g.j[b] * k + j] : 0.0;
#pragma unroll
        for (int b = 0; b < G; ++b) cp[b] = (KPL > 0 && b < gprev.n) ? (double)C[(int64_t)gprev.j[b] * k + j] : 0.0;
        // one or (ride.ept == 2) two elements per thread: feature f0 and f0 + 256 of the workgroup's 256 ept
        const int ept = (KPL > 0 && ride.ept == 2) ? 2 : 1;
        const int64_t fq[2] = {(int64_t)wg * 256 * ept + threadIdx.x, ept == 2 ? (int64_t)wg * 512 + 256 + threadIdx.x : s};
        // radius = the budget + the atom's old norm on the sampled features: the partial sums of the group's gradient launch,
        // summed by every workgroup in the same order (block_sum2: fixed association)
        double old = 0, dummy = 0;
        for (int i = threadIdx.x; i < part_stride; i += 256) old += partial_old[(int64_t)a * part_stride + i];
        const double cn = (double)comp_norm[j], lprev = level_hint ? level_hint[j] : 0.0;
        T val[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t f = fq[q];
            const int64_t fc = f < s ? f : s - 1;
            double x = num[(int64_t)a * ldr + fc];
            const T dj = dold[(int64_t)a * ldr + fc];
            T sn[G], so[G], pn[G], po[G];
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const int bc = b < a ? b : 0;
                sn[b] = stage_cur[(int64_t)bc * ldr + fc];
                so[b] = dold[(int64_t)bc * ldr + fc];
            }
            if constexpr (KPL > 0) {
#pragma unroll
                for (int b = 0; b < G; ++b) {                        // (pipelined sweep: what the group before changed)
                    const int bc = b < gprev.n ? b : 0;
                    pn[b] = stage_prev[(int64_t)bc * ldr + fc];
                    po[b] = dold_prev[(int64_t)bc * ldr + fc];
                }
#pragma unroll
                for (int b = 0; b < G; ++b)
                    if (b < gprev.n) x -= cp[b] * ((double)pn[b] - (double)po[b]);
            }
#pragma unroll
            for (int b = 0; b < G; ++b)
                if (b < a) x -= cb[b] * ((double)sn[b] - (double)so[b]);
            T v = dj;
            if (!frozen) v = (T)(x / cjj);
            if (pos && v < (T)0) v = 0;                              // dict_fact.py:684-685
            if (f >= s) v = 0;
            if (f < s) u[f] = v;                                     // (what the last workgroup projects if this attempt gives up)
            val[q] = v;
        }
        block_sum2(old, dummy, red, 256);
        const double radius = (double)(T)(cn + old);                 // comp_norm_[k] += subset_norm (:676-678)
        T outv[2];
        double nrm, level;
        bool searched;
#ifdef MODL_DIAG
        const bool withhold = (parity & 2) && wg == 1;               // (MODL_DEBUG_ATOM_MWG = 2: the fallback path, tests)
#else
        const bool withhold = false;
#endif
        const unsigned long long t1 = clock64();
        const bool ok = mwg_l1_project<T>(val[0], val[1], radius, lprev, xb, xabort, red, outv[0], outv[1], nrm, level, searched, withhold,
                                          MwgPlace{wg, nwg_main});
        const unsigned long long t2 = clock64();
        if (ok) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (fq[q] < s) stage_cur[(int64_t)a * ldr + fq[q]] = outv[q];
            if (wg == 0 && threadIdx.x == 0) {
                comp_norm[j] = (T)(radius - nrm);                    // :690-692
                if (level_hint && searched) level_hint[j] = level;
            }
        }
        if (dbg && wg == 0 && threadIdx.x == 0) { dbg[8] = t0; dbg[9] = t1; dbg[10] = t2; dbg[11] = clock64(); dbg[12] = (unsigned long long)red[19]; }
        // every workgroup arrives; the last one looks at the abort word and, if it is raised, projects everything the old way
        if (!arrive_last(counter, (unsigned)nwg_main, &flag)) return;
        if (__hip_atomic_load(xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        if (threadIdx.x == 0) __hip_atomic_store(xabort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atom_project<T, false>(u, reinterpret_cast<T *>(step_smem), partial_old + (int64_t)a * part_stride, part_stride, Dt, subset, s, k, j,
                               rho, comp_norm, red, dbg, level_hint, stage_cur + (int64_t)a * ldr);
        return;
    }
    // the numerators of the group's launch, minus what the atoms before this one changed
    const double cjj = (double)C[(int64_t)j * k + j];
    const bool frozen = !((T)cjj > (T)1e-20);
    // (rows padded to whole passes: pipe_row_stride)
    const bool use_regs = KPL > 0 && ride.regs && rho == 1.0 && s <= 64 * 256 && ldr >= (s <= 40 * 256 ? 40 : 64) * 256;
    double cb[G], cp[G];
#pragma unroll
    for (int b = 0; b < G; ++b) cb[b] = (b < a) ? (double)C[(int64_t)g.j[b] * k + j] : 0.0;
#pragma unroll
    for (int b = 0; b < G; ++b) cp[b] = (KPL > 0 && b < gprev.n) ? (double)C[(int64_t)gprev.j[b] * k + j] : 0.0;
    for (int64_t f = (int64_t)wg * blockDim.x + threadIdx.x; f < s; f += (int64_t)nwg_main * blockDim.x) {
        double x = num[(int64_t)a * ldr + f];
        const T dj = dold[(int64_t)a * ldr + f];
        T sn[G], so[G];
#pragma unroll
        for (int b = 0; b < G; ++b) {
            const int bc = b < a ? b : 0;
            sn[b] = stage_cur[(int64_t)bc * ldr + f];
            so[b] = dold[(int64_t)bc * ldr + f];
        }
        if constexpr (KPL > 0) {                                     // (pipelined sweep: what the group before changed)
            T pn[G], po[G];
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const int bc = b < gprev.n ? b : 0;
                pn[b] = stage_prev[(int64_t)bc * ldr + f];
                po[b] = dold_prev[(int64_t)bc * ldr + f];
            }
#pragma unroll
            for (int b = 0; b < G; ++b)
                if (b < gprev.n) x -= cp[b] * ((double)pn[b] - (double)po[b]);
        }
#pragma unroll
        for (int b = 0; b < G; ++b)
            if (b < a) x -= cb[b] * ((double)sn[b] - (double)so[b]);
        T val = dj;
        if (!frozen) val = (T)(x / cjj);
        if (pos && val < (T)0) val = 0;                              // dict_fact.py:684-685
        if (use_regs) store_wt(u + f, val);                          // (write-through: the hand-off below has no fence)
        else u[f] = val;
    }
    const unsigned long long t1 = clock64();
    if (use_regs) {
        // hand-off without fences (bcd_persist.hip: signal_word): the write-through stores drained, one relaxed ticket; the last
        // workgroup reads the candidates past its caches (atom_project_regs: sc1 loads).  An agent-scope release would write back
        // the XCD's whole L2 - the riders' dictionary rows included - and the acquire invalidate it: 4 k cycles per atom
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)nwg_main - 1;
            if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            flag = last;
        }
        __syncthreads();
        if (!flag) return;
    } else if (!arrive_last(counter, (unsigned)nwg_main, &flag)) return;
    const unsigned long long t2 = clock64();
    bool done = false;
    if constexpr (KPL > 0) {
        if (use_regs) {
            if (s <= 40 * 256)
                atom_project_regs<T, 40>(u, s, j, partial_old + (int64_t)a * part_stride, part_stride, comp_norm, level_hint,
                                         stage_cur + (int64_t)a * ldr, red, dbg);
            else
                atom_project_regs<T, 64>(u, s, j, partial_old + (int64_t)a * part_stride, part_stride, comp_norm, level_hint,
                                         stage_cur + (int64_t)a * ldr, red, dbg);
            done = true;
        }
    }
    if (!done)
        atom_project<T, false>(u, reinterpret_cast<T *>(step_smem), partial_old + (int64_t)a * part_stride, part_stride, Dt, subset, s, k,
                               j, rho, comp_norm, red, dbg, level_hint, stage_cur + (int64_t)a * ldr);
    if (dbg && threadIdx.x == 0) { dbg[0] = t0; dbg[1] = t1; dbg[2] = t2; dbg[3] = clock64(); }
}

template <int... As, class F>
__device__ __forceinline__ void for_each_int(std::integer_sequence<int, As...>, F &&f) {
    (f(std::integral_constant<int, As>{}), ...);
}

// ---- The grouped atom update (l1 / elastic-net atoms, s <= 24 * 256 sampled features): a GROUP of up to eight consecutive
// atoms of the sweep per pair of launches.  Of the 23 us an atom costs with one launch each (atom_step_kernel), the
// launch boundary, the gradient row and the hand-off to the projecting workgroup are 60 %; they are shared by the group.
// First launch: many small workgroups evaluate, for their features, the numerators of the candidates of ALL atoms of
// the group against the dictionary as it is at the start of the launch (one read of the dictionary row serves the
// group).  Second launch: ONE workgroup - a projection is a chain of block-wide reductions, which a single compute
// unit does fastest - projects the atoms one after the other, the numerator of atom a corrected for what the atoms
// before it in the group just changed:
//     num_a[f] -= sum_{a' < a} C[j_a', j_a] (D_new[j_a'][f] - D_old[j_a'][f])
// which is exactly the difference between the gradient row of the sequential sweep and the stale one (the own term
// C_jj D_old[j][f] does not depend on the other atoms).  Same update as atom_step_kernel up to the rounding of that
// double-precision correction.  It runs alone on the chip and may use the whole register file of its compute unit (256
// VGPRs + 256 AGPRs per wavefront): the changes of the group's earlier atoms stay in registers.
// Round 2 did this in ONE launch per group of four (the last workgroup to arrive projected): the launch boundary costs
// what the release / ticket / acquire hand-off cost (1.5 against 1.7 us), but the single kernel had to keep the register
// budget of its many gradient workgroups, so the changes travelled through memory (two dependent round trips per atom
// for a lone workgroup: 18 k of its 108 k cycles on the fMRI shape) and eight atoms did not fit; fixed cost per group
// (gradient phase, hand-off, prologue) 33 k cycles.  fMRI shape: 48.8 us per 4 atoms -> 43 us per 8.
// The projected atoms leave the projecting workgroup as COMPACT rows (stage[a][feature], 16-byte stores): written
// straight into the dictionary they would be scattered 4-byte stores, one cache line per lane.  The workgroups of the
// NEXT gradient launch put them where they belong while they read their dictionary rows anyway (gp, stage_in: the
// previous group of this sweep); atom_stage_flush_group_kernel does it for the last group.

// A wavefront takes kGradRows feature rows (grid = ceil(s / (4 kGradRows)) workgroups: s <= 24 * 256 fits 512) and requests
// EVERYTHING it needs - the group's columns of C, its rows of the dictionary, the entries of B_, the staged atoms of
// the previous group - before the first use: one memory round trip per launch instead of one for C and one per row.
// After the wave-level sums every lane holds them: lane a writes atom a.
constexpr int kGradRows = 3;
template <typename T, int KPL, int G>
__global__ __launch_bounds__(256) void atom_grad_group_kernel(T *Dt, const T *Bt, const T *C, const int32_t *subset, int64_t s,
                                                              int k, AtomGroupN<G> g, AtomGroupN<G> gp, const T *stage_in,
                                                              double rho, double *num, T *dold, double *partial_old,
                                                              int64_t ldr) {
    // ldr: stride of the rows of num / dold / stage (s rounded up to whole passes of the projecting workgroup, whose
    // loads and stores are then unconditional)
    static_assert(G <= 64, "one lane per atom of the group");
    __shared__ double s_oldw[4][G];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int e0 = lane * KPL;
    int ja[G];
    T cc[G][KPL], Cjj[G];
#pragma unroll
    for (int a = 0; a < G; ++a) {
        ja[a] = g.j[a < g.n ? a : 0];                                   // (a short last group repeats its first atom: not stored)
#pragma unroll
        for (int c = 0; c < KPL; ++c) cc[a][c] = C[(int64_t)ja[a] * k + (e0 + c < k ? e0 + c : k - 1)];
        Cjj[a] = C[(int64_t)ja[a] * k + ja[a]];
    }
    const int64_t f0 = ((int64_t)blockIdx.x * 4 + wid) * kGradRows;
    int64_t r[kGradRows];
    T rv[kGradRows][KPL], dj[kGradRows][G], bj[kGradRows][G], pv[kGradRows][G];
#pragma unroll
    for (int q = 0; q < kGradRows; ++q) {
        const int64_t f = f0 + q < s ? f0 + q : s - 1;                   // (clamped: no branch around a load)
        r[q] = sub_row(subset, f) * k;
    }
#pragma unroll
    for (int q = 0; q < kGradRows; ++q) {
        const int64_t f = f0 + q < s ? f0 + q : s - 1;
        const T *row = Dt + r[q];
#pragma unroll
        for (int c = 0; c < KPL; ++c) rv[q][c] = row[e0 + c < k ? e0 + c : k - 1];
#pragma unroll
        for (int a = 0; a < G; ++a) { dj[q][a] = row[ja[a]]; bj[q][a] = Bt[r[q] + ja[a]]; }
#pragma unroll
        for (int a = 0; a < G; ++a) pv[q][a] = stage_in[(int64_t)(a < gp.n ? a : 0) * ldr + f];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < G; ++a)
#pragma unroll
        for (int c = 0; c < KPL; ++c) cc[a][c] = (e0 + c < k) ? cc[a][c] : (T)0;
    double old = 0;                                                      // lane a: atom a
#pragma unroll
    for (int q = 0; q < kGradRows; ++q) {
        const bool live = f0 + q < s;                                    // wavefront-uniform
        T pmine = 0;                                                     // lane a: the staged value of the previous group's atom a
        int jmine = -1;
#pragma unroll
        for (int a = 0; a < G; ++a) {
            if (a < gp.n) {                                              // the previous group's atoms: into the row
#pragma unroll
                for (int c = 0; c < KPL; ++c) rv[q][c] = (e0 + c == gp.j[a]) ? pv[q][a] : rv[q][c];
                if (lane == a) { pmine = pv[q][a]; jmine = gp.j[a]; }
            }
        }
        if (live && jmine >= 0) Dt[r[q] + jmine] = pmine;
        double nmine = 0;
        T dmine = 0;
#pragma unroll
        for (int a = 0; a < G; ++a) {
            double dot = 0;
#pragma unroll
            for (int c = 0; c < KPL; ++c) dot += (double)rv[q][c] * (double)cc[a][c];
            dot = wave_sum(dot);
            const double nv = ((double)bj[q][a] - dot) + (double)Cjj[a] * (double)dj[q][a];
            if (lane == a) { nmine = nv; dmine = dj[q][a]; }
        }
        if (live && lane < g.n) {
            num[(int64_t)lane * ldr + f0 + q] = nmine;
            dold[(int64_t)lane * ldr + f0 + q] = dmine;
            const double ab = fabs((double)dmine);
            old += ab * (rho + (1.0 - rho) * ab);
        }
    }
    if (lane < G) s_oldw[wid][lane] = old;
    __syncthreads();
    if (threadIdx.x < G)
        partial_old[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] =
            (s_oldw[0][threadIdx.x] + s_oldw[1][threadIdx.x]) + (s_oldw[2][threadIdx.x] + s_oldw[3][threadIdx.x]);
}

// two block-wide sums of a four-wavefront workgroup with ONE barrier: the exchange slots alternate (red4: 2 x 8 doubles;
// a slot is rewritten two exchanges later, when every wavefront has passed the barrier behind its last read).  Same
// association as block_sum2.
__device__ __forceinline__ void block_sum2_pp(double &a, double &b, double *red4, int &par) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    double *r = red4 + 8 * par;
    par ^= 1;
    if (lane == 0) { r[2 * wid] = a; r[2 * wid + 1] = b; }
    __syncthreads();
    a = (r[0] + r[2]) + (r[4] + r[6]);
    b = (r[1] + r[3]) + (r[5] + r[7]);
}
__device__ __forceinline__ void block_sum1_pp(double &a, double *red4, int &par) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    a = wave_sum(a);
    double *r = red4 + 8 * par;
    par ^= 1;
    if (lane == 0) r[2 * wid] = a;
    __syncthreads();
    a = (r[0] + r[2]) + (r[4] + r[6]);
}

// Element e of a thread of the projecting workgroup is element atom_elem(e) of the row: chunks of four consecutive
// elements per thread, so that numerators, old values and the output row move as 16-byte accesses - 20 memory
// operations per atom instead of 80.  (With one operation per element the 40 prefetch loads of the next atom behind the
// 20 output stores of the last one overflowed the 6-bit vmcnt counter, and the compiler drained the stores - a full
// write round trip, exposed - before it could issue the loads.)
__device__ __forceinline__ int atom_elem(int e) { return 4 * (int)threadIdx.x + (e & 3) + 1024 * (e >> 2); }
template <int EPT>
__device__ __forceinline__ void load_row4(const double *row, double (&v)[EPT]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
        const d2 *p = reinterpret_cast<const d2 *>(row + 4 * threadIdx.x + 1024 * q);
        const d2 lo = p[0], hi = p[1];
        v[4 * q] = lo[0]; v[4 * q + 1] = lo[1]; v[4 * q + 2] = hi[0]; v[4 * q + 3] = hi[1];
    }
}
template <int EPT>
__device__ __forceinline__ void load_row4(const float *row, float (&v)[EPT]) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(row + 4 * threadIdx.x + 1024 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}
template <int EPT>
__device__ __forceinline__ void store_row4(float *row, const float (&v)[EPT]) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q)
        *reinterpret_cast<float4 *>(row + 4 * threadIdx.x + 1024 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int EPT>
__device__ __forceinline__ void store_row4(double *row, const double (&v)[EPT]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
        d2 *p = reinterpret_cast<d2 *>(row + 4 * threadIdx.x + 1024 * q);
        d2 lo, hi;
        lo[0] = v[4 * q]; lo[1] = v[4 * q + 1]; hi[0] = v[4 * q + 2]; hi[1] = v[4 * q + 3];
        p[0] = lo; p[1] = hi;
    }
}

// block_enet_project_vals (enet_block.hpp) for the lone projecting workgroup of atom_project_group_kernel: the same
// level equation and iteration, but |x| and its term are recomputed in every pass instead of kept (2 x EPT doubles of
// registers that hold the group's changes instead), one barrier per exchange, ONE output code path (inside the ball is
// level 0, radius 0 is a flag: every merge of two paths costs register copies in a kernel this large), the output
// row written as 16-byte stores, and the warm start of l1 atoms (L1: l1_ratio == 1, gamma == 0) is the previous level
// ITSELF: f(l) = sum_{|x| > l} (|x| - l) - R is convex and decreasing, so the Newton step that Michelot's update is
// lands at or left of the root from EITHER side, and the iteration rises monotonically from there - no check, no safety
// factor (4.4 -> 3.3 passes per atom on the fMRI shape).  A step that lands at or below zero proves nothing and falls
// back to the cold start (which also settles "inside the ball").  The output row has room for EPT * 256 elements (x is
// zero beyond n, and so is the output).  Leaves the result in x; returns THIS THREAD'S share of its enet norm (the
// block-wide sum is off the chain of the projections: the caller forms it for all atoms of a group in one exchange).
// dbg[4] = passes | warm << 16, dbg[5] = clock when the level is known.
template <typename T, int EPT, bool L1>
__device__ __forceinline__ double enet_project_slim(double (&x)[EPT], T *out, double radius, double l1_ratio,
                                                    double *red4, int &par, double l_prev, double *level_out,
                                                    unsigned long long *dbg) {
    if (!L1 && l1_ratio == 0.0 && radius > 0.0) {           // enet.pyx:62-70, radius in squared-norm units
        double s2 = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) s2 += x[e] * x[e];
        block_sum1_pp(s2, red4, par);
        const T scale = (s2 <= radius) ? (T)1 : (T)sqrt(s2 / radius);
        T o[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            o[e] = (T)x[e] / scale;
            x[e] = (double)o[e];
        }
        store_row4<EPT>(out, o);
        return threadIdx.x == 0 ? ((s2 <= radius) ? s2 : radius) : 0.0;   // (as a partial sum: thread 0 carries the value)
    }
    const bool zero = !(radius > 0.0);                       // enet.pyx:57-59 (radius == 0 -> zeros)
    const double gamma = L1 ? 0.0 : 2.0 / l1_ratio - 2.0;
    const double R = radius / l1_ratio;
    const double hg = 0.5 * gamma;
    auto term = [&](double ax) { return L1 ? ax : ax * (1.0 + hg * ax); };
    auto pass = [&](double lv, double &S, double &cnt) {     // selects, no branches; two chains (block_enet_project_vals)
        double S0 = 0, S1 = 0;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int e = 0; e < EPT; e += 2) {
            const double a0 = fabs(x[e]), a1 = fabs(x[e + 1]);
            const bool i0 = a0 > lv, i1 = a1 > lv;
            S0 += i0 ? term(a0) : 0.0;
            S1 += i1 ? term(a1) : 0.0;
            c0 += i0 ? 1 : 0;
            c1 += i1 ? 1 : 0;
        }
        S = S0 + S1;
        cnt = (double)(c0 + c1);
        block_sum2_pp(S, cnt, red4, par);
    };
    auto solve = [&](double S, double cnt) {
        if (!L1) {                                           // enet.pyx:113-117
            const double qa = gamma * gamma * R + gamma * cnt * 0.5;
            const double qd = 2.0 * R * gamma + cnt;
            const double qc = R - S;
            return (-qd + sqrt(qd * qd - 4.0 * qa * qc)) / (2.0 * qa);
        }
        return (S - R) / cnt;                                // :119
    };
    double level = 0.0, prev_cnt = -1.0;
    bool warm = false, search = !zero;
    int npass = 0;
    if (search && l_prev > 0.0 && l_prev < 1e300) {
        const double l0 = L1 ? l_prev : 0.9 * l_prev;
        double S, cnt;
        if (L1) {
            pass(l0, S, cnt);
            ++npass;
            const double l1 = (S - R) / cnt;
            if (cnt != 0.0 && l1 > 0.0) { warm = true; prev_cnt = cnt; level = l1; }
        } else {                                             // the verified guess of block_enet_project_vals
            double P = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) P += (fabs(x[e]) > l0) ? fabs(x[e]) : 0.0;
            pass(l0, S, cnt);
            ++npass;
            block_sum1_pp(P, red4, par);
            const double d = 1.0 + l0 * gamma;
            const double sum_u = (P - cnt * l0) / d;
            const double sum_a2 = (S - P) / (0.5 * gamma);
            const double sum_u2 = (sum_a2 - 2.0 * l0 * P + cnt * l0 * l0) / (d * d);
            const double h0 = sum_u + 0.5 * gamma * sum_u2;
            if (h0 >= R * (1.0 + 1e-9) && cnt != 0.0) { warm = true; prev_cnt = cnt; level = solve(S, cnt); }
        }
    }
    if (search && !warm) {
        double tot = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) tot += term(fabs(x[e]));
        block_sum1_pp(tot, red4, par);
        if (tot <= R) search = false;                        // inside the ball: level 0 is the identity
    }
    if (search) {
        for (int p = 0; p < 256; ++p) {
            double S, cnt;
            pass(level, S, cnt);
            ++npass;
            if (cnt == prev_cnt || cnt == 0.0) break;
            prev_cnt = cnt;
            level = solve(S, cnt);
        }
    }
    if (dbg && threadIdx.x == 0) { dbg[4] = (unsigned)npass | (warm ? 1u << 16 : 0u); dbg[5] = clock64(); }
    if (level_out && threadIdx.x == 0 && search) *level_out = level;
    const double lT = (double)(T)level;
    const double inv_den = zero ? 0.0 : 1.0 / (1.0 + lT * gamma);   // one division (exactly 1 for l1 atoms), EPT products
    double nrm = 0;
    T o[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        double pos = fabs(x[e]) - lT;
        pos = pos > 0 ? pos : 0;
        o[e] = (T)(((x[e] >= 0) ? pos : -pos) * inv_den);   // enet.pyx:121, sign(0) = +1
        o[e] = zero ? (T)0 : o[e];
        x[e] = (double)o[e];
        const double a = fabs(x[e]);
        nrm += a * (l1_ratio + (1.0 - l1_ratio) * a);
    }
    store_row4<EPT>(out, o);
    return nrm;                                              // this thread's share: the caller sums a whole group's at once
}

// The projections of a group, in sweep order, by ONE workgroup of four wavefronts (one per SIMD).  Per atom: the
// numerators (prefetched while the atom before was projected) minus what the group's earlier atoms changed, from
// registers, in sweep order; the candidate; the projection; the output row; the change stays in registers for the atoms
// behind.  Every scalar of the group (coefficients, budgets, old-norm sums, level hints) is fetched in the prologue, all
// requests before the first wait: a lone workgroup pays a full memory round trip for every dependent load.
// dbg: [0] launches, [3] prologue, [4 + 4a] atom a (corrections + projection), [5 + 4a] its passes, [7 + 4a] warm
// starts taken (a < 4: the first four atoms of the group), [20] whole kernel, over all atoms: [24] corrections and
// candidate, [25] level search, [26] output row and norm, [28] atoms; [32..39] scratch.  (The stamps cost a memory
// round trip per atom on wavefront 0: they inflate what follows them.)
template <typename T, int EPT, int G, bool L1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void atom_project_group_kernel(const T *C, int64_t s, int k, AtomGroupN<G> g, T *stage_out, int pos, double rho,
                               const double *num, const T *dold, const double *partial_old, int nparts, T *comp_norm,
                               double *level_hint, unsigned long long *dbg) {
    static_assert(EPT % 4 == 0 && G % 4 == 0, "chunks of four elements, four wavefronts");
    constexpr int64_t ldr = (int64_t)EPT * 256;                         // row stride of num / dold / stage_out
    __shared__ double red4[16];
    __shared__ double s_coef[G][G], s_cjj[G], s_cn[G], s_old[G], s_lvl[G];
    __shared__ double s_nrm[G][256];                                    // the threads' shares of the atoms' norms
    const unsigned long long t0 = dbg ? clock64() : 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double X[2][EPT];
    T Dd[2][EPT];
    auto load_atom = [&](int a, double (&xd)[EPT], T (&dd)[EPT]) {      // (elements beyond s: whatever the scratch holds,
        const int ac = a < g.n ? a : g.n - 1;                            //  masked where the candidate is formed)
        load_row4<EPT>(num + (int64_t)ac * ldr, xd);
        load_row4<EPT>(dold + (int64_t)ac * ldr, dd);
    };
    load_atom(0, X[0], Dd[0]);
    {
        // old-norm sums: wavefront w sums the partials of atoms w, w + 4, ... (no exchange between the wavefronts); up to
        // 512 partials per atom, every load requested before the first sum (clamped, masked)
        constexpr int NP = 8;
        double pv[G / 4][NP];
#pragma unroll
        for (int h = 0; h < G / 4; ++h) {
            const int a = 4 * h + wid;
            const int ac = a < g.n ? a : 0;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int i = lane + 64 * u;
                pv[h][u] = partial_old[(int64_t)ac * nparts + (i < nparts ? i : nparts - 1)];
            }
        }
        const int t = threadIdx.x;                                        // (G * G <= 64 < 256: one element per thread)
        const int b = (t / G) % G, a = t % G;
        const int jb = g.j[b < g.n ? b : 0], jaa = g.j[a < g.n ? a : 0];   // (from the kernel arguments)
        const double cba = (double)C[(int64_t)jb * k + jaa], caa = (double)C[(int64_t)jaa * k + jaa];
        const double cn = (double)comp_norm[jaa], lv = level_hint ? level_hint[jaa] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        if (t < G * G) {
            s_coef[b][a] = cba;
            if (b == 0) { s_cjj[a] = caa; s_cn[a] = cn; s_lvl[a] = lv; }
        }
#pragma unroll
        for (int h = 0; h < G / 4; ++h) {
            double o = 0;
#pragma unroll
            for (int u = 0; u < NP; ++u) o += (lane + 64 * u < nparts) ? pv[h][u] : 0.0;
            o = wave_sum(o);
            if (lane == 0) s_old[4 * h + wid] = o;
        }
    }
    __syncthreads();
    if (dbg && threadIdx.x == 0) { dbg[0] += 1; dbg[3] += clock64() - t0; }
    int par = 0;
    double dl[G > 1 ? G - 1 : 1][EPT];                                  // what the atoms of the group changed
    auto step = [&](auto A_) {
        constexpr int a = decltype(A_)::value;
        if (a >= g.n) return;                                            // workgroup-uniform
        const unsigned long long ta = dbg ? clock64() : 0;
        if (dbg && threadIdx.x == 0) dbg[36] = 0;
        double (&x)[EPT] = X[a & 1];
        T (&dd)[EPT] = Dd[a & 1];
        if (a + 1 < G) load_atom(a + 1, X[(a + 1) & 1], Dd[(a + 1) & 1]);   // lands while this atom is projected
#pragma unroll
        for (int b = 0; b < a; ++b) {
            const double c = s_coef[b][a];
#pragma unroll
            for (int e = 0; e < EPT; ++e) x[e] -= c * dl[b][e];
        }
        const int j = g.j[a];
        const double radius = (double)(T)(s_cn[a] + s_old[a]);          // comp_norm_[k] += subset_norm (:676-678)
        const double cjj = s_cjj[a];
        const bool frozen = !((T)cjj > (T)1e-20);
        const double icjj = 1.0 / cjj;                                   // one reciprocal, then products
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            T val = dd[e];
            if (!frozen) val = (T)(x[e] * icjj);
            if (pos && val < (T)0) val = 0;                              // dict_fact.py:684-685
            x[e] = (atom_elem(e) < s) ? (double)val : 0.0;
        }
        const unsigned long long tb = dbg ? clock64() : 0;
        s_nrm[a][threadIdx.x] = enet_project_slim<T, EPT, L1>(x, stage_out + (int64_t)a * ldr, radius, rho, red4, par, s_lvl[a],
                                                              level_hint ? level_hint + j : nullptr, dbg ? dbg + 32 : nullptr);
        if (a + 1 < G) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) dl[a < G - 1 ? a : 0][e] = (atom_elem(e) < s) ? x[e] - (double)dd[e] : 0.0;
        }
        if (dbg && threadIdx.x == 0) {
            const unsigned long long tc = clock64();
            if (a < 4) {
                dbg[4 + 4 * a] += tc - ta;
                dbg[5 + 4 * a] += dbg[36] & 0xffff;
                dbg[7 + 4 * a] += dbg[36] >> 16;
            }
            dbg[24] += tb - ta; dbg[25] += dbg[37] - tb; dbg[26] += tc - dbg[37]; dbg[28] += 1;
        }
    };
    for_each_int(std::make_integer_sequence<int, G>{}, step);
    // the norms of the projected atoms, all at once (one barrier and two wave sums per wavefront instead of a block
    // exchange per atom on the chain of the projections): wavefront w takes atoms w, w + 4, ...
    __syncthreads();
#pragma unroll
    for (int h = 0; h < G / 4; ++h) {
        const int a = 4 * h + __builtin_amdgcn_readfirstlane(wid);
        if (a < g.n) {                                                   // (wavefront-uniform)
            double v = (s_nrm[a][lane] + s_nrm[a][lane + 64]) + (s_nrm[a][lane + 128] + s_nrm[a][lane + 192]);
            v = wave_sum(v);
            const double radius = (double)(T)(s_cn[a] + s_old[a]);
            if (lane == 0) comp_norm[g.j[a]] = (T)(radius - v);          // :690-692
        }
    }
    if (dbg && threadIdx.x == 0) dbg[20] += clock64() - t0;
}

// the last group's staged atoms -> dictionary
template <typename T, int G>
__global__ __launch_bounds__(256) void atom_stage_flush_group_kernel(T *Dt, const int32_t *subset, int64_t s, int k,
                                                                     AtomGroupN<G> g, const T *stage, int64_t ldr) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= s) return;
    const int64_t r = sub_row(subset, f) * k;
#pragma unroll
    for (int a = 0; a < G; ++a)
        if (a < g.n) Dt[r + g.j[a]] = stage[(int64_t)a * ldr + f];
}

// The whole sweep in ONE launch, by one workgroup, for TINY problems (u = the s-vector of the atom in flight
// lives in LDS): the generic path is a chain of k strictly sequential atoms, each needing a few global
// reductions over the sampled features (Michelot passes); for a handful of kilobytes one workgroup beats 2 k
// launches.  Its thread-per-row reads are uncoalesced, so anything larger uses one multi-workgroup launch
// per atom (atom_step_kernel).  Same arithmetic as atom_grad_kernel + atom_project_kernel (the dot product is accumulated
// in double, sequentially over the atoms instead of lane-wise).
template <typename T>
__global__ __launch_bounds__(1024) void atom_sweep_kernel(T *Dt, const T *Bt, const T *C, const int32_t *subset,
                                                          const int32_t *order, int64_t s, int k, int pos, double rho,
                                                          T *comp_norm) {
    extern __shared__ __attribute__((aligned(16))) char sweep_smem[];
    double *red = reinterpret_cast<double *>(sweep_smem);             // [16]
    double *ccol = red + 16;                                          // [k] column j of C (C is symmetric)
    T *u = reinterpret_cast<T *>(ccol + k);                           // [s]
    for (int t = 0; t < k; ++t) {
        const int j = order[t];
        for (int c = threadIdx.x; c < k; c += blockDim.x) ccol[c] = (double)C[(int64_t)j * k + c];
        __syncthreads();
        const T Cjj = C[(int64_t)j * k + j];
        const bool frozen = !(Cjj > (T)1e-20);                        // dict_fact.py:681
        double old = 0;
        for (int64_t f = threadIdx.x; f < s; f += blockDim.x) {
            const int64_t r = sub_row(subset, f) * k;
            const T *row = Dt + r;
            double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            int c = 0;
            for (; c + 3 < k; c += 4) {
                d0 += (double)row[c] * ccol[c];
                d1 += (double)row[c + 1] * ccol[c + 1];
                d2 += (double)row[c + 2] * ccol[c + 2];
                d3 += (double)row[c + 3] * ccol[c + 3];
            }
            for (; c < k; ++c) d0 += (double)row[c] * ccol[c];
            const double dot = (d0 + d1) + (d2 + d3);
            const T dj = row[j];
            T val = dj;
            if (!frozen) val = (T)((((double)Bt[r + j] - dot) + (double)Cjj * (double)dj) / (double)Cjj);
            if (pos && val < (T)0) val = 0;                           // dict_fact.py:684-685
            u[f] = val;
            const double a = fabs((double)dj);
            old += a * (rho + (1.0 - rho) * a);
        }
        old = block_sum(old, red);
        const double radius = (double)(T)((double)comp_norm[j] + old);   // comp_norm_[k] += subset_norm (:676-678)
        const double nrm = block_enet_project<T>(u, 1, u, 1, s, radius, rho, red);
        __syncthreads();
        for (int64_t f = threadIdx.x; f < s; f += blockDim.x) Dt[sub_row(subset, f) * k + j] = u[f];
        if (threadIdx.x == 0) comp_norm[j] = (T)(radius - nrm);          // :690-692
        __syncthreads();
    }
}

// -------------------------------------------------------------------- sgd path
template <typename T>
__global__ __launch_bounds__(256) void col_norm_partial_kernel(const T *Dt, const int32_t *subset, int64_t s, int k,
                                                               double rho, double *partial) {
    const int64_t f_begin = (int64_t)blockIdx.x * kGramRows;
    const int64_t f_end = (f_begin + kGramRows < s) ? f_begin + kGramRows : s;
    for (int j = threadIdx.x; j < k; j += 256) {
        double acc = 0;
        for (int64_t f = f_begin; f < f_end; ++f) {
            const double a = fabs((double)Dt[sub_row(subset, f) * k + j]);
            acc += a * (rho + (1.0 - rho) * a);
        }
        partial[(int64_t)blockIdx.x * k + j] = acc;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void col_norm_add_kernel(const double *partial, int nslab, int k, T *comp_norm) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    double acc = 0;
    for (int z = 0; z < nslab; ++z) acc += partial[(int64_t)z * k + j];
    comp_norm[j] = (T)((double)comp_norm[j] + acc);
}
template <typename T> struct EpiSgd {
    T *Dnew; const T *Dt; const T *Bt; const int32_t *subset; int k; T ws;
    __device__ __forceinline__ void operator()(int64_t f, int64_t j, T v) const {
        const int64_t e = sub_row(subset, f) * k + j;
        Dnew[f * k + j] = Dt[e] + ws * (Bt[e] - v);
    }
};
template <typename T>
__global__ __launch_bounds__(256) void sgd_project_kernel(T *Dnew, T *Dt, const int32_t *subset, int64_t s, int k,
                                                          double rho, T *comp_norm) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const double radius = (double)comp_norm[j];
    const double nrm = block_enet_project<T>(Dnew + j, k, Dnew + j, k, s, radius, rho, red);
    __syncthreads();
    for (int64_t f = threadIdx.x; f < s; f += 256) Dt[sub_row(subset, f) * k + j] = Dnew[f * k + j];
    if (threadIdx.x == 0) comp_norm[j] = (T)(radius - nrm);
}

// ---------------------------------------------------------------------- driver
template <typename T> int dict_update_generic(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches);

template <typename T>
int dict_update(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches) {
    const int k = a.k;
    const int64_t s = a.s;
    if (s <= 0 || k <= 0) return MODL_OK;
    if (k > 1024) return MODL_EINVAL;
    const DuLayout L = du_layout(sizeof(T), s, k);
    if (L.total > a.ws_bytes) return MODL_ENOMEM;
    char *ws = static_cast<char *>(a.ws);
    int nl = 0;

    if (a.optimizer == MODL_OPT_SGD) {
        double *colp = reinterpret_cast<double *>(ws + L.off_colp);
        T *Dnew = reinterpret_cast<T *>(ws + L.off_Dnew);
        const int nslab = (int)cdiv(s, kGramRows);
        hipLaunchKernelGGL((col_norm_partial_kernel<T>), dim3(nslab), dim3(256), 0, stream, a.Dt, a.subset, s, k,
                           a.comp_l1_ratio, colp);
        MODL_LAUNCH_CHECK();
        hipLaunchKernelGGL((col_norm_add_kernel<T>), dim3((unsigned)cdiv(k, 256)), dim3(256), 0, stream, colp, nslab, k,
                           a.comp_norm);
        MODL_LAUNCH_CHECK();
        nl += 2;
        Operand A, B;
        A.ptr = a.Dt; A.si = k; A.sk = 1; A.gi = gather32(a.subset);
        B.ptr = a.C; B.si = k; B.sk = 1;                 // B(n = j, kk = m) = C[j][m]
        EpiSgd<T> epi{Dnew, a.Dt, a.Bt, a.subset, k, (T)(a.w * a.step_size)};
        SplitWs none;
        MODL_TRY((launch_gemm<T, EpiSgd<T>>(stream, A, B, s, k, k, epi, none, &nl, 512, 1)));
        hipLaunchKernelGGL((sgd_project_kernel<T>), dim3(k), dim3(256), 0, stream, Dnew, a.Dt, a.subset, s, k,
                           a.comp_l1_ratio, a.comp_norm);
        MODL_LAUNCH_CHECK();
        ++nl;
    } else if (a.comp_l1_ratio == 0.0 && !a.comp_pos) {
        T *CP = reinterpret_cast<T *>(ws + L.off_CP);
        T *cdiag = reinterpret_cast<T *>(ws + L.off_cdiag);
        int32_t *frozen = reinterpret_cast<int32_t *>(ws + L.off_frozen);
        T *abuf = reinterpret_cast<T *>(ws + L.off_a);
        double *partial = reinterpret_cast<double *>(ws + L.off_partial);
        double *Tp = reinterpret_cast<double *>(ws + L.off_Tp);
        double *coef_all = reinterpret_cast<double *>(ws + L.off_coef);
        double *CA[2] = {Tp, Tp + kResStride};
        unsigned int *counter = reinterpret_cast<unsigned int *>(Tp + 2 * kResStride);
        const bool fused = std::is_same<T, float>::value && k <= 512;
        const int kp = (int)cdiv(k, 4) * 4;                      // atoms of the packed arrays (dead ones behind the k real)
        constexpr int64_t rt1_max = MODL_RT1_MAX;
        int RT = (s <= rt1_max || k > 256) ? 1 : 2;              // k > 256: 64-row tiles would spill registers
        {   // one workgroup per compute unit (registers): a grid that is a few workgroups larger than the chip runs in
            // two rounds, i.e. every block launch takes twice as long (s = 16.7 k: 261 workgroups of 64 features ->
            // 174 of 96 features)
            static const int ncu_blk = [] {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                    return prop.multiProcessorCount;
                return 256;
            }();
            if (RT == 2 && cdiv(s, 64) > ncu_blk && cdiv(s, 96) <= ncu_blk) RT = 3;
        }
        // ONE persistent launch per dictionary update (bcd_persist.hip) when its workgroups - one per 32 or 64 sampled rows
        // plus the resolver - can all be resident at once; otherwise one launch per block of 32 atoms
        int persist_rt = 0;
        if (fused && a.allow_persist && g_bcd_persist.load(std::memory_order_relaxed) && g_bcd_acc.load(std::memory_order_relaxed) &&
            cdiv(k, kNB) <= kPersistBlocksMax) {
            // (every workgroup must be resident: the resolver + one per 32 or 64 sampled rows - asked of the runtime for the
            //  kernel's own footprint on this device, bcd_persist_fits.  Reduction 1 at the metric's shape: 157 workgroups of
            //  64 rows, 0.173 -> 0.138 ms per dictionary update against one launch per block; config 5's 16.7 k sampled rows
            //  do not fit and keep one launch per block.  The LDS of riding k-wide tiles is part of the question: a launch
            //  that cannot carry them must not have marked them consumed)
            const size_t xl = (a.rider && a.rider->p >= 2048) ? wide_lds_bytes<32, 128>() : 0;
            const int64_t n1 = cdiv(s, 32), n2 = cdiv(s, 64);
            if (n1 <= kPersistRowsMax && bcd_persist_fits(kp, 1, (int)n1, xl)) persist_rt = 1;
            else if (kp <= 256 && n2 <= kPersistRowsMax && bcd_persist_fits(kp, 2, (int)n2, xl)) persist_rt = 2;
        }
        if (persist_rt) RT = persist_rt;
        const int nslab = fused ? (int)cdiv(s, 32 * RT) : (int)cdiv(s, kGramRows);
        const int GPW = (kp <= 256) ? 8 : 16;
        void (*blk)(BcdBlockArgs, BcdRiderArgs) = nullptr;
        T *DsP = reinterpret_cast<T *>(ws + L.off_Dnew), *BsP = reinterpret_cast<T *>(ws + L.off_BsP), *CPP = CP;
        // the Gram accumulators of the fused path (three in rotation; in the space of the group sums, which they replace)
        long long *fused_acc = (fused && !persist_rt && g_bcd_acc.load(std::memory_order_relaxed)) ? reinterpret_cast<long long *>(ws + L.off_gpartial) : nullptr;
        const int pshards = nslab > MODL_ACC_SHARD_MIN ? kAccShards : 1;
        long long *pacc = persist_rt ? reinterpret_cast<long long *>(ws + L.off_pacc) : nullptr;
        unsigned int *pflags = reinterpret_cast<unsigned int *>(ws + L.off_pflags);
        double *qcoef = persist_rt ? reinterpret_cast<double *>(ws + L.off_qcoef) : nullptr;
        static_assert(sizeof(long long) * 3 * kAccShards * kAccWords <= sizeof(double) * 2 * (size_t)kCounters * (kNB * kNB + kNB), "accumulators fit");
        if (fused) {
            blk = (RT == 1) ? (GPW == 8 ? bcd_block_kernel<1, 8> : bcd_block_kernel<1, 16>)
                            : (RT == 2 ? bcd_block_kernel<2, 8> : bcd_block_kernel<3, 8>);
            MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(blk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
            hipLaunchKernelGGL((bcd_setup_kernel<T>), dim3((unsigned)(kNB + cdiv(kp, kSetupRows) + cdiv(s, kSetupRows))), dim3(256), 0, stream, a.C, a.order, k, kp, CPP,
                               cdiag, frozen, coef_all, counter, a.comp_norm, reinterpret_cast<T *>(ws + L.off_norm_in), a.Dt,
                               a.Bt, a.subset, s, DsP, BsP, fused_acc, pacc,
                               (long long)cdiv(k, kNB) * pshards * kPAccWords, qcoef, pflags,
                               reinterpret_cast<long long *>(ws + L.off_Sbuf), persist_rt ? (long long)cdiv(k, kNB) * kNB * kNB : 0);
            MODL_LAUNCH_CHECK();
            ++nl;
        } else {
            const int nfew = (int)cdiv(s, kFewRows);
            const bool few = std::is_same<T, double>::value && s > kTinyRows && nfew <= kFewMaxWg && g_bcd_tiny.load(std::memory_order_relaxed) &&
                             g_bcd_few.load(std::memory_order_relaxed);
            long long *few_x = few ? reinterpret_cast<long long *>(ws + L.off_few) : nullptr;
            const long long few_words = 3LL * nfew * kFewRec;             // (the error word sits right behind them)
            hipLaunchKernelGGL((bcd_prepare_kernel<T>), dim3(k > kNB ? k : kNB), dim3(256), sizeof(int32_t) * (size_t)k,
                               stream, a.C, a.order, k, CP, cdiag, frozen, coef_all, counter, a.comp_norm,
                               reinterpret_cast<T *>(ws + L.off_norm_in), few_x, few_words);
            MODL_LAUNCH_CHECK();
            ++nl;
            if (few) {                                       // the whole sweep in one launch on a few workgroups
                const size_t lds = bcd_tiny_lds(sizeof(T), kFewRows) + sizeof(T) * ((size_t)k + 1) + sizeof(int64_t) * kFewRows + 32;
                MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bcd_few_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((bcd_few_kernel<T>), dim3(nfew), dim3(256), lds, stream, a.Dt, a.Bt, CP, cdiag, frozen, coef_all,
                                   a.subset, a.order, (int)s, k, a.comp_norm, reinterpret_cast<const T *>(ws + L.off_norm_in),
                                   reinterpret_cast<double *>(few_x), reinterpret_cast<unsigned int *>(few_x + few_words),
                                   a.persist_flags);
                MODL_LAUNCH_CHECK();
                ++nl;
                if (launches) *launches += nl;
                return MODL_OK;
            }
            if (std::is_same<T, double>::value && s <= kTinyRows && g_bcd_tiny.load(std::memory_order_relaxed)) {   // the whole sweep by one workgroup
                MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bcd_tiny_kernel<T>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((bcd_tiny_kernel<T>), dim3(1), dim3(256), bcd_tiny_lds(sizeof(T), s), stream, a.Dt, a.Bt, CP,
                                   cdiag, frozen, coef_all, a.subset, a.order, (int)s, k, a.comp_norm);
                MODL_LAUNCH_CHECK();
                ++nl;
                if (launches) *launches += nl;
                return MODL_OK;
            }
        }
        const size_t rec_half = (size_t)L.nslab_max * kResStride, grec_half = (size_t)kCounters * kResStride;   // >= the packed sizes
        double *gpart = reinterpret_cast<double *>(ws + L.off_gpartial);
        BcdBlockArgs base;
        base.acc_out = nullptr; base.acc_in = nullptr; base.acc_zero = nullptr; base.shards = 1;
        if (fused) {
            base.Dt = reinterpret_cast<float *>(DsP); base.Bt = reinterpret_cast<const float *>(BsP);
            base.CP = reinterpret_cast<const float *>(CPP); base.cdiag = reinterpret_cast<const float *>(cdiag);
            base.frozen = frozen; base.order = a.order; base.a = reinterpret_cast<float *>(abuf);
            base.coef_all = coef_all; base.norm_in = reinterpret_cast<const float *>(ws + L.off_norm_in);
            base.norm_out = reinterpret_cast<float *>(a.comp_norm); base.counter = counter;
            base.stamps = reinterpret_cast<unsigned long long *>(counter + kCounters); base.s = s; base.k = kp; base.kout = k;
            // up to kCounters workgroups: every workgroup sums all records itself in (B) - the pre-summing tail (a
            // release fence + a serial reduction behind the slowest workgroup of a group) costs more than the second
            // round of record loads it saves (measured at 32 workgroups: 17.7 -> 15.9 us per launch); beyond, groups
            base.group = nslab <= kCounters ? nslab : ((nslab + 31) / 32 > kGroup ? (nslab + 31) / 32 : kGroup);
            base.Dt_out = reinterpret_cast<float *>(a.Dt); base.subset = a.subset;
            base.shards = nslab > MODL_ACC_SHARD_MIN ? kAccShards : 1;       // (one accumulator: 30 workgroups at the metric's shape)
        }
        // deferred statistics product riding along (launches 1 .. nblk carry cdiv(tiles, nblk) tiles each)
        BcdRiderArgs rid;
        rid.nslab = persist_rt ? nslab + 1 : nslab;                 // (the persistent launch: the resolver in front of the row workgroups)
#ifdef MODL_DIAG
        if (fused) rid.dbg = reinterpret_cast<unsigned long long *>(counter + kCounters) + 40;     // (stamps of the first riding tile)
#endif
        int ride_tiles = 0, ride_per = 0, ride_next = 0;
        if (fused && a.rider) {
            const StatsRider &R = *a.rider;
            DenseOperand Xo, Cd;
            Xo.ptr = R.X; Xo.si = 1; Xo.sk = R.ldx;                    // element (i = feature, kk = sample)
            Cd.ptr = R.code; Cd.si = 1; Cd.sk = k;                     // element (i = atom, kk = sample)
            EpiStatsSkip<float> epi{static_cast<float *>(R.Bt), k, R.stamp, R.step, (float)R.beta, (float)R.wt,
                                    (float)R.bdiv, R.replace};
            rid.P = plan_stats<EpiStatsSkip<float>>(Xo, Cd, R.p, k, R.b, epi);
            // wide tiles (gemm_wide.hpp) keep their X tile in LDS for 128 atoms: X is fetched twice instead of eight
            // times, and a tile is still short enough for the shadow of a block step.  (Very large feature counts do
            // not ride at all: somf_step.hip runs their product as its own k-wide launch.)
            const int wbm = 32;
            if (R.p >= 2048) rid.W = plan_wide<32, EpiStatsSkip<float>, 128>(Xo, Cd, R.p, k, R.b, epi);
            rid.wide = rid.W.ok ? wbm : 0;
            if ((rid.P.ok || rid.wide) && R.p > 0) {
                ride_tiles = rid.wide ? rid.W.tm * rid.W.tn : rid.P.tn * rid.P.tm;
                // as few carrier launches as the idle compute units allow (a tile needs a compute unit to itself for
                // most of a block step; more tiles than free units would queue and stretch the launch)
                static const int ncu = [] {                             // one process per GPU: queried once
                    int dev = 0;
                    hipDeviceProp_t prop;
                    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                        return prop.multiProcessorCount;
                    return 256;
                }();
                const int free_cu = (ncu - rid.nslab > 64) ? ncu - rid.nslab : 64;
                const int nblk_all = (int)cdiv(k, kNB);
                int carriers = (int)cdiv(ride_tiles, free_cu);
                if (carriers > nblk_all) carriers = nblk_all;
                if (carriers < 1) carriers = 1;
                ride_per = (int)cdiv(ride_tiles, carriers);
                a.rider->consumed = 1;
            }
        }
        // a launch that carries k-wide tiles needs their LDS (one workgroup per compute unit then)
        auto lds_bytes = [&](int extra) {
            size_t n = bcd_block_lds(GPW, RT);
            if (extra > 0 && rid.wide) n = std::max(n, wide_lds_bytes<32, 128>());
            return n;
        };
        auto ride = [&](BcdRiderArgs &r) {                             // the next share of tiles; returns their number
            r = rid;
            r.t0 = ride_next;
            r.t1 = (ride_next + ride_per < ride_tiles) ? ride_next + ride_per : ride_tiles;
            ride_next = r.t1;
            return r.t1 - r.t0;
        };
        if (persist_rt) {
            BcdPersistArgs pa;
            pa.DsP = reinterpret_cast<const float *>(DsP); pa.BsP = reinterpret_cast<const float *>(BsP);
            pa.CPP = reinterpret_cast<const float *>(CPP); pa.cdiag = reinterpret_cast<const float *>(cdiag);
            pa.frozen = frozen; pa.order = a.order; pa.subset = a.subset;
            pa.coef_all = coef_all; pa.qcoef = qcoef;
            pa.norm_in = reinterpret_cast<const float *>(ws + L.off_norm_in);
            pa.norm_out = reinterpret_cast<float *>(a.comp_norm);
            pa.Dt_out = reinterpret_cast<float *>(a.Dt);
            pa.acc = pacc;
            pa.rec = reinterpret_cast<double *>(ws + L.off_prec);
            pa.Sbuf = reinterpret_cast<double *>(ws + L.off_Sbuf);
            pa.arrive = pflags; pa.sflag = pflags + kPersistBlocksMax; pa.err = pflags + 2 * kPersistBlocksMax;
            pa.flags = a.persist_flags;
            pa.C = reinterpret_cast<const float *>(a.C);
            pa.expect = nslab;
#ifdef MODL_DIAG
            // diagnostics build only: 3 makes the resolver wait for a workgroup that never comes (the recovery path), 4 makes it
            // wait for it from the SECOND block on (the unrecoverable path), tests/test_gpu_step.py
            pa.inject = g_bcd_persist.load(std::memory_order_relaxed);
#endif
            pa.stamps = reinterpret_cast<unsigned long long *>(ws + L.off_pstamps);
            pa.s = s; pa.k = kp; pa.kout = k; pa.nblk = (int)cdiv(k, kNB); pa.nrow = nslab; pa.shards = pshards;
            BcdRiderArgs r = rid;                                       // every riding tile with the one launch
            r.t0 = 0; r.t1 = ride_tiles;
            int extra = ride_tiles;
            if (a.stage && a.stage->src) {                              // + one workgroup: the next minibatch's parameters
                r.stage = *a.stage;
                a.stage->consumed = 1;
                ++extra;
            }
            MODL_TRY(launch_bcd_persist(stream, pa, r, extra, (ride_tiles > 0 && rid.wide) ? wide_lds_bytes<32, 128>() : 0, persist_rt));
            ++nl;
            if (launches) *launches += nl;
            return MODL_OK;
        }
        int blk_i = 0, j0_prev = 0, nb_prev = 0;
        for (int j0 = 0; j0 < k; j0 += kNB, ++blk_i) {
            const int nb = (k - j0 < kNB) ? k - j0 : kNB;
            double *CAcur = CA[blk_i & 1], *CAprev = blk_i ? CA[(blk_i - 1) & 1] : nullptr;
            if (fused) {
                BcdBlockArgs ba = base;
                ba.rec_out = partial + (size_t)(blk_i & 1) * rec_half; ba.rec_in = partial + (size_t)((blk_i + 1) & 1) * rec_half;
                ba.grec_out = gpart + (size_t)(blk_i & 1) * grec_half; ba.grec_in = gpart + (size_t)((blk_i + 1) & 1) * grec_half;
                ba.j0 = j0; ba.nb = nb; ba.j0_prev = j0_prev; ba.nb_prev = blk_i ? nb_prev : 0;
                if (fused_acc) {
                    ba.acc_out = fused_acc + (size_t)(blk_i % 3) * kAccShards * kAccWords;
                    ba.acc_in = fused_acc + (size_t)((blk_i + 2) % 3) * kAccShards * kAccWords;
                    ba.acc_zero = fused_acc + (size_t)((blk_i + 1) % 3) * kAccShards * kAccWords;
                    ba.group = nslab;                                   // (no pre-summed groups: one accumulator)
                }
                BcdRiderArgs r = rid;
                const int extra = blk_i ? ride(r) : 0;                 // (launch 0 is short: no resolver)
                hipLaunchKernelGGL(blk, dim3(nslab + extra), dim3(384), lds_bytes(extra), stream, ba, r);
                MODL_LAUNCH_CHECK();
                ++nl;
            } else {
                if (CAprev) {
                    hipLaunchKernelGGL((bcd_apply_kernel<T>), dim3((unsigned)cdiv(s, 64)), dim3(256), 0, stream, abuf,
                                       CAprev, a.Dt, a.subset, a.order, s, k, j0_prev, nb_prev);
                    MODL_LAUNCH_CHECK();
                    ++nl;
                }
                Operand A, B;
                A.ptr = a.Dt; A.si = k; A.sk = 1; A.gi = gather32(a.subset);
                B.ptr = CP + j0; B.si = 1; B.sk = k;         // B(n = jj, kk = m) = CP[m][j0 + jj]
                EpiBcdA<T> epi{abuf, a.Dt, a.Bt, cdiag, frozen, a.subset, a.order, k, j0};
                SplitWs none;
                MODL_TRY((launch_gemm<T, EpiBcdA<T>>(stream, A, B, s, nb, k, epi, none, &nl, 512, 1)));
                hipLaunchKernelGGL((bcd_gram_kernel<T>), dim3(nslab), dim3(256), 0, stream, abuf, a.Dt, a.subset,
                                   a.order, s, k, j0, nb, partial);
                MODL_LAUNCH_CHECK();
                hipLaunchKernelGGL((bcd_resolve_kernel<T>), dim3(1), dim3(256), 0, stream, partial, nslab, coef_all,
                                   a.order, k, j0, nb, a.comp_norm, CAcur);
                MODL_LAUNCH_CHECK();
                nl += 2;
            }
            j0_prev = j0; nb_prev = nb;
        }
        // the last block's atoms
        if (fused) {
            BcdBlockArgs ba = base;                  // nb == 0: resolve and apply only
            ba.rec_out = nullptr; ba.grec_out = nullptr;
            ba.rec_in = partial + (size_t)((blk_i + 1) & 1) * rec_half; ba.grec_in = gpart + (size_t)((blk_i + 1) & 1) * grec_half;
            ba.j0 = 0; ba.nb = 0; ba.j0_prev = j0_prev; ba.nb_prev = nb_prev;
            if (fused_acc) {
                ba.acc_out = nullptr; ba.acc_zero = nullptr;
                ba.acc_in = fused_acc + (size_t)((blk_i + 2) % 3) * kAccShards * kAccWords;     // (blk_i: the blocks launched so far)
                ba.group = nslab;
            }
            BcdRiderArgs r = rid;
            ride_per = ride_tiles - ride_next;                         // whatever is left
            int extra = ride(r);
            if (a.stage && a.stage->src) {                             // + one workgroup: the next minibatch's parameters
                r.stage = *a.stage;
                a.stage->consumed = 1;
                hipLaunchKernelGGL(blk, dim3(nslab + extra + 1), dim3(384), lds_bytes(extra), stream, ba, r);
            } else
            hipLaunchKernelGGL(blk, dim3(nslab + extra), dim3(384), lds_bytes(extra), stream, ba, r);
            MODL_LAUNCH_CHECK();
            nl += 1;
        } else {
            hipLaunchKernelGGL((bcd_apply_kernel<T>), dim3((unsigned)cdiv(s, 64)), dim3(256), 0, stream, abuf,
                               CA[(blk_i - 1) & 1], a.Dt, a.subset, a.order, s, k, j0_prev, nb_prev);
            MODL_LAUNCH_CHECK();
            ++nl;
        }
    } else {
        return dict_update_generic<T>(stream, a, launches);
    }
    if (launches) *launches += nl;
    return MODL_OK;
}

// the last atom of a sweep of atom_step_kernel launches with staged rows: its values into the dictionary
template <typename T>
__global__ __launch_bounds__(256) void atom_stage_flush_kernel(T *Dt, const int32_t *subset, int64_t s, int k, int j,
                                                               const T *stage) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f < s) Dt[sub_row(subset, f) * k + j] = stage[f];
}

// generic path: the sweep order is needed on the host (one launch pair per atom)
template <typename T>
int dict_update_generic(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches) {
    const int64_t *h_order = a.h_order;
    if (!h_order) return MODL_EINVAL;
    const int k = a.k;
    const int64_t s = a.s;
    if (s <= 0 || k <= 0) return MODL_OK;
    if (k > 1024) return MODL_EINVAL;
    const DuLayout L = du_layout(sizeof(T), s, k);
    if (L.total > a.ws_bytes) return MODL_ENOMEM;
    char *ws = static_cast<char *>(a.ws);
    T *u = reinterpret_cast<T *>(ws + L.off_u);
    double *pold = reinterpret_cast<double *>(ws + L.off_pold);
    // the whole sweep order is checked BEFORE the first launch: an index out of range in a later group would otherwise
    // come back as MODL_EINVAL with earlier groups' norm budgets already updated and their projected atoms still in the
    // staging rows
    for (int t = 0; t < k; ++t)
        if (h_order[t] < 0 || h_order[t] >= k) return MODL_EINVAL;
    // small problems: the whole sweep in one launch (u in LDS, see atom_sweep_kernel)
    const size_t lds = sizeof(double) * (16 + (size_t)k) + sizeof(T) * (size_t)s + 16;
    if (a.order && lds <= 64 * 1024 && (double)s * k <= 32e3) {     // thread-per-row reads: only worth it when tiny
        auto kern = atom_sweep_kernel<T>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, stream, a.Dt, a.Bt, a.C, a.subset, a.order, s, k, a.comp_pos,
                           a.comp_l1_ratio, a.comp_norm);
        MODL_LAUNCH_CHECK();
        if (launches) *launches += 1;
        return MODL_OK;
    }
    int nwg = (int)cdiv(s, 4);                       // 4 waves per workgroup (the projection wants registers: 256 threads), one feature per wave while the grid lasts
    if (nwg > 512) nwg = 512;
    if (nwg > L.nwg_grad) nwg = (int)L.nwg_grad;
    if (nwg < 1) nwg = 1;
    unsigned int *counter = reinterpret_cast<unsigned int *>(reinterpret_cast<double *>(ws + L.off_Tp) + 2 * kResStride);
    MODL_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned int), stream));   // the last arriver re-arms it after each atom
    const size_t u_lds = (sizeof(T) * (size_t)s <= 60 * 1024) ? sizeof(T) * (size_t)s : 0;
    // groups of atoms per launch pair while the vector fits the registers of the projecting workgroup
    if (s <= (int64_t)kProjEpt * 256 && k <= 512) {
        const bool l1 = a.comp_l1_ratio == 1.0;
        const int G = (s <= 20 * 256) ? 8 : 4;                           // (24 elements per thread: the changes of 3 atoms fit, not of 7)
        nwg = (int)cdiv(s, 4 * kGradRows);                               // kGradRows rows per wavefront (<= 512: s <= 24 * 256)
        const int64_t ldr = atom_row_stride(s);       // rows of the group's scratch: whole passes of the projecting workgroup
        char *gb = ws + L.off_Dnew;
        double *num = reinterpret_cast<double *>(gb);
        T *dold = reinterpret_cast<T *>(num + (size_t)G * ldr);
        T *stage[2] = {dold + (size_t)G * ldr, dold + (size_t)2 * G * ldr};
        unsigned long long *dbg = g_atom_stamps.load(std::memory_order_relaxed);
        auto run = [&](auto G_) -> int {
            constexpr int GG = decltype(G_)::value;
            AtomGroupN<GG> gp;                            // the group whose projected atoms are staged (none yet)
            gp.n = 0;
            for (int x = 0; x < GG; ++x) gp.j[x] = 0;
            int gi = 0;
            for (int t = 0; t < k; t += GG, ++gi) {
                AtomGroupN<GG> g;
                g.n = (k - t < GG) ? k - t : GG;
                for (int x = 0; x < GG; ++x) {
                    const int64_t j = h_order[t + (x < g.n ? x : 0)];
                    if (j < 0 || j >= k) return MODL_EINVAL;
                    g.j[x] = (int)j;
                }
#define MODL_GRAD(KPL)                                                                                                        \
    hipLaunchKernelGGL((atom_grad_group_kernel<T, KPL, GG>), dim3(nwg), dim3(256), 0, stream, a.Dt, a.Bt, a.C, a.subset, s, k, g, \
                       gp, stage[(gi + 1) & 1], a.comp_l1_ratio, num, dold, pold, ldr)
                if (k <= 64) MODL_GRAD(1);
                else if (k <= 128) MODL_GRAD(2);
                else if (k <= 256) MODL_GRAD(4);
                else MODL_GRAD(8);
#undef MODL_GRAD
                MODL_LAUNCH_CHECK();
#define MODL_PROJ(EPT, L1)                                                                                                    \
    hipLaunchKernelGGL((atom_project_group_kernel<T, EPT, GG, L1>), dim3(1), dim3(256), 0, stream, a.C, s, k, g, stage[gi & 1],   \
                       a.comp_pos, a.comp_l1_ratio, num, dold, pold, nwg, a.comp_norm, a.level_hint, dbg)
                if constexpr (GG == 8) {
                    if (s <= 12 * 256) { if (l1) MODL_PROJ(12, true); else MODL_PROJ(12, false); }
                    else { if (l1) MODL_PROJ(20, true); else MODL_PROJ(20, false); }
                } else {
                    if (l1) MODL_PROJ(kProjEpt, true); else MODL_PROJ(kProjEpt, false);
                }
#undef MODL_PROJ
                MODL_LAUNCH_CHECK();
                if (launches) *launches += 2;
                gp = g;
            }
            hipLaunchKernelGGL((atom_stage_flush_group_kernel<T, GG>), dim3((unsigned)cdiv(s, 256)), dim3(256), 0, stream, a.Dt,
                               a.subset, s, k, gp, stage[(gi + 1) & 1], ldr);
            MODL_LAUNCH_CHECK();
            if (launches) *launches += 1;
            return MODL_OK;
        };
        return G == 8 ? run(std::integral_constant<int, 8>{}) : run(std::integral_constant<int, 4>{});
    }
    // beyond the register-resident projection (24 elements per thread): groups of atoms, compact rows (atom_grad4_kernel)
    const bool staged = u_lds != 0 && s > (int64_t)kProjEpt * 256;
    if (staged) {
        // groups of kStepGroup atoms (atom_step_group_kernel): one read of the dictionary per group; scratch in the a-tile's space
        constexpr int G = kStepGroup;
        static_assert(sizeof(T) * 3 * G + sizeof(double) * G <= sizeof(T) * kNB, "the group's rows fit the a-tile's space");
        const int64_t ldr = s;
        T *base = reinterpret_cast<T *>(ws + L.off_a);
        T *stage[2] = {base, base + (size_t)G * ldr};
        T *dold = base + (size_t)2 * G * ldr;
        double *num = reinterpret_cast<double *>(ws + L.off_a + align_up(sizeof(T) * (size_t)3 * G * ldr, 16));
        if (align_up(sizeof(T) * (size_t)3 * G * ldr, 16) + sizeof(double) * (size_t)G * ldr > sizeof(T) * (size_t)s * kNB) return MODL_ENOMEM;
        if ((int64_t)G * nwg > L.nwg_grad) return MODL_ENOMEM;
        const int nwg_corr = (int)cdiv(s, 256);                           // a thread per feature
        // l1 atoms on at most 64 workgroups: the projection spread over the launch (mwg_l1_project); its two exchange buffers
        // (sentinels) and the abort word live in the blocked path's record space, unused here
        const bool mwg = a.comp_l1_ratio == 1.0 && nwg_corr <= kMwgMaxWg && g_atom_mwg.load(std::memory_order_relaxed) &&
                         sizeof(double) * 2 * (size_t)L.nslab_max * (kNB * kNB + kNB) >= sizeof(double) * 2 * kMwgWords + 64;
        double *xch = reinterpret_cast<double *>(ws + L.off_partial);
        unsigned int *xabort = reinterpret_cast<unsigned int *>(xch + 2 * kMwgWords);
        unsigned long launch_no = 0;
        if (mwg) {
            MODL_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(xch), (int)kMwgSentinel32, 2 * (size_t)kMwgWords * 2, stream));
            MODL_HIP(hipMemsetAsync(xabort, 0, sizeof(unsigned int), stream));
        }
        AtomGroupN<G> g, gp;
        gp.n = 0;
        for (int b = 0; b < G; ++b) gp.j[b] = 0;
        GradRide<T> no_ride{};
        // ---- the pipelined sweep (round 6; atom_corr_project_kernel): the gradient launch of group g + 1 rides on the atom launches of
        // group g.  Scratch of its own (DuLayout::off_pipe): two staged groups, THREE sets of old values
        // (the riders write group g + 1's while the launches of group g still read group g - 1's), two sets of numerators and of
        // old-norm partial sums.
        const int ngroups = (int)cdiv(k, G);
        static_assert(G == 4 && kPipeMinRows == (int64_t)kProjEpt * 256, "du_layout sizes off_pipe for groups of four");
        // MODL_DEBUG_ATOM_MWG = 1 (default): the launch's LAST workgroup projects the vector from registers (atom_project_regs: 20.2 ms
        // per minibatch at the HCP shape); 3: the projection spread over the launch's workgroups (mwg_l1_project: 27.0 ms)
        const bool regs = a.comp_l1_ratio == 1.0 && g_atom_mwg.load(std::memory_order_relaxed) == 1 && s <= 64 * 256;
        const bool pipelined = (mwg || regs) && ngroups >= 2 && g_atom_pipe.load(std::memory_order_relaxed) != 0 && s > kPipeMinRows &&
                               (int64_t)2 * G * nwg <= L.nwg_grad;
        if (pipelined) {
            const int64_t ldr = pipe_row_stride(s);                       // (shadows the a-tile's stride: this scratch is its own)
            // two elements per thread beyond 32 workgroups: half as many slots to poll and sum per exchange (C6: 26.8 against 27.3 ms)
            const int ept = (nwg_corr > 32 && !regs) ? 2 : 1;
            const int nwg_p = (int)cdiv(s, 256 * ept);
            T *pb = reinterpret_cast<T *>(ws + L.off_pipe);
            T *pstage[2] = {pb, pb + (size_t)G * ldr};
            T *pdold[3] = {pb + (size_t)2 * G * ldr, pb + (size_t)3 * G * ldr, pb + (size_t)4 * G * ldr};
            double *pnum0 = reinterpret_cast<double *>(ws + L.off_pipe + align_up(sizeof(T) * (size_t)5 * G * ldr, 16));
            double *pnum[2] = {pnum0, pnum0 + (size_t)G * ldr};
            double *ppold[2] = {pold, pold + (size_t)G * nwg};
            auto group_of = [&](int gi_, AtomGroupN<G> &out) -> bool {
                const int t0 = gi_ * G;
                out.n = (k - t0 < G) ? k - t0 : G;
                for (int b = 0; b < G; ++b) {
                    out.j[b] = (int)h_order[t0 + (b < out.n ? b : 0)];
                    if (out.j[b] < 0 || out.j[b] >= k) return false;
                }
                return true;
            };
            AtomGroupN<G> none;
            none.n = 0;
            for (int b = 0; b < G; ++b) none.j[b] = 0;
            AtomGroupN<G> gcur, gnext, gprev = none;
            if (!group_of(0, gcur)) return MODL_EINVAL;
#define MODL_GRAD4(KPL)                                                                                               \
    hipLaunchKernelGGL((atom_grad4_kernel<T, KPL>), dim3(nwg), dim3(256), 0, stream, a.Dt, a.Bt, a.C, a.subset, s, k, gcur, none,     \
                       (const T *)pstage[1], a.comp_l1_ratio, pnum[0], pdold[0], ldr, ppold[0], nwg)
            if (k <= 64) MODL_GRAD4(1);
            else if (k <= 128) MODL_GRAD4(2);
            else if (k <= 256) MODL_GRAD4(4);
            else if (k <= 512) MODL_GRAD4(8);
            else MODL_GRAD4(16);
#undef MODL_GRAD4
            MODL_LAUNCH_CHECK();
            for (int gi = 0; gi < ngroups; ++gi) {
                const bool has_next = gi + 1 < ngroups;
                if (has_next && !group_of(gi + 1, gnext)) return MODL_EINVAL;
                for (int ai = 0; ai < gcur.n; ++ai) {
                    GradRide<T> ride{};
                    int nride = 0;
                    if (has_next) {                                       // (then this group is a full one: G launches, a part each)
                        const int lo = (int)((int64_t)ai * nwg / G), hi = (int)((int64_t)(ai + 1) * nwg / G);
                        nride = hi - lo;
                        ride.Bt = a.Bt; ride.gn = gnext; ride.gfold = gprev; ride.stage_fold = pstage[(gi + 1) & 1];
                        ride.num = pnum[(gi + 1) & 1]; ride.dold = pdold[(gi + 1) % 3]; ride.pold = ppold[(gi + 1) & 1];
                        ride.part_stride = nwg; ride.vb0 = lo; ride.nvb = nwg; ride.nride = nride;
                    }
                    ride.ept = ept;
                    ride.regs = regs ? 1 : 0;
                    const int grid = nwg_p + nride;
#define MODL_CORR(KPL)                                                                                                            \
    hipLaunchKernelGGL((atom_corr_project_kernel<T, KPL>), dim3(grid), dim3(256), u_lds, stream, a.Dt, a.C, a.subset, s, k,       \
                       gcur, ai, a.comp_pos, a.comp_l1_ratio, u, (const double *)pnum[gi & 1], (const T *)pdold[gi % 3], pstage[gi & 1], ldr,  \
                       (const double *)ppold[gi & 1], nwg, a.comp_norm, counter,                                                                \
                       reinterpret_cast<unsigned long long *>(counter + kCounters), a.level_hint, regs ? nullptr : xch, xabort,                \
                       (int)(launch_no++ & 1) | (g_atom_mwg.load(std::memory_order_relaxed) == 2 ? 2 : 0), nwg_p, gprev,                       \
                       (const T *)pstage[(gi + 1) & 1], (const T *)pdold[(gi + 2) % 3], ride)
                    if (k <= 64) MODL_CORR(1);
                    else if (k <= 128) MODL_CORR(2);
                    else if (k <= 256) MODL_CORR(4);
                    else if (k <= 512) MODL_CORR(8);
                    else MODL_CORR(16);
#undef MODL_CORR
                    MODL_LAUNCH_CHECK();
                }
                if (has_next) { gprev = gcur; gcur = gnext; }
            }
            // home: the last two groups (the riders of a group's launches put the group BEFORE it home; the last group has none)
            int nflush = 0;
            for (int gi = ngroups - 2; gi < ngroups; ++gi) {
                const AtomGroupN<G> &gf = (gi == ngroups - 1) ? gcur : gprev;
                for (int b = 0; b < gf.n; ++b, ++nflush) {
                    hipLaunchKernelGGL((atom_stage_flush_kernel<T>), dim3((unsigned)cdiv(s, 256)), dim3(256), 0, stream, a.Dt, a.subset, s, k,
                                       gf.j[b], (const T *)(pstage[gi & 1] + (size_t)b * ldr));
                    MODL_LAUNCH_CHECK();
                }
            }
            if (launches) *launches += k + 1 + nflush;
            return MODL_OK;
        }
        int gi = 0;
        for (int t0 = 0; t0 < k; t0 += G, ++gi) {
            g.n = (k - t0 < G) ? k - t0 : G;
            for (int b = 0; b < G; ++b) {
                g.j[b] = (int)h_order[t0 + (b < g.n ? b : 0)];
                if (g.j[b] < 0 || g.j[b] >= k) return MODL_EINVAL;
            }
#define MODL_GRAD4(KPL)                                                                                               \
    hipLaunchKernelGGL((atom_grad4_kernel<T, KPL>), dim3(nwg), dim3(256), 0, stream, a.Dt, a.Bt, a.C, a.subset, s, k, g, gp,          \
                       (const T *)stage[(gi + 1) & 1], a.comp_l1_ratio, num, dold, ldr, pold, nwg)
            if (k <= 64) MODL_GRAD4(1);
            else if (k <= 128) MODL_GRAD4(2);
            else if (k <= 256) MODL_GRAD4(4);
            else if (k <= 512) MODL_GRAD4(8);
            else MODL_GRAD4(16);
#undef MODL_GRAD4
            MODL_LAUNCH_CHECK();
            for (int ai = 0; ai < g.n; ++ai) {
                hipLaunchKernelGGL((atom_corr_project_kernel<T, 0>), dim3(nwg_corr), dim3(256), u_lds, stream, a.Dt, a.C, a.subset, s, k,
                                   g, ai, a.comp_pos, a.comp_l1_ratio, u, (const double *)num, (const T *)dold, stage[gi & 1], ldr,
                                   (const double *)pold, nwg, a.comp_norm, counter,
                                   reinterpret_cast<unsigned long long *>(counter + kCounters), a.level_hint,
                                   mwg ? xch : nullptr, xabort,
                                   (int)(launch_no++ & 1) | (g_atom_mwg.load(std::memory_order_relaxed) == 2 ? 2 : 0), nwg_corr, gp,
                                   (const T *)nullptr, (const T *)nullptr, no_ride);
                MODL_LAUNCH_CHECK();
            }
            gp = g;
        }
        for (int b = 0; b < gp.n; ++b) {                                 // the last group's rows
            hipLaunchKernelGGL((atom_stage_flush_kernel<T>), dim3((unsigned)cdiv(s, 256)), dim3(256), 0, stream, a.Dt, a.subset, s, k,
                               gp.j[b], (const T *)(stage[(gi + 1) & 1] + (size_t)b * ldr));
            MODL_LAUNCH_CHECK();
        }
        if (launches) *launches += k + gi + gp.n;
        return MODL_OK;
    }
    for (int t = 0; t < k; ++t) {
        const int j = (int)h_order[t];
        if (j < 0 || j >= k) return MODL_EINVAL;
#define MODL_STEP(KPL)                                                                                            \
    hipLaunchKernelGGL((atom_step_kernel<T, KPL>), dim3(nwg), dim3(256), u_lds, stream, a.Dt, a.Bt, a.C, a.subset, s, k, j, \
                       a.comp_pos, a.comp_l1_ratio, u, pold, a.comp_norm, counter, u_lds ? 1 : 0, \
                       reinterpret_cast<unsigned long long *>(counter + kCounters), a.level_hint)
        if (k <= 64) MODL_STEP(1);
        else if (k <= 128) MODL_STEP(2);
        else if (k <= 256) MODL_STEP(4);
        else if (k <= 512) MODL_STEP(8);
        else MODL_STEP(16);
#undef MODL_STEP
        MODL_LAUNCH_CHECK();
    }
    if (launches) *launches += k;
    return MODL_OK;
}

template int dict_update<float>(hipStream_t, const DictUpdateArgs<float> &, int *);
template int dict_update<double>(hipStream_t, const DictUpdateArgs<double> &, int *);

}  // namespace modl
