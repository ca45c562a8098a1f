// Ridge code solve: (G + alpha I) code = Dx  by Cholesky factorisation and two
// triangular substitutions.
//
// Replaces the LAPACK ?posv calls of the reference's ridge branch
//   (modl/decomposition/dict_fact_fast.pyx:82-94 per-sample Gram, :174-197
//   shared Gram).  As there, `positive`, `tol` and `max_iter` do not apply and
//   no failure is reported for a non positive-definite system (the factor then
//   holds NaNs, like an unchecked LAPACK info).
//
// gfx950 mapping: one workgroup factors one k x k system in LDS when it fits
// (k*k*sizeof(T) <= 96 KiB: k <= 156 in f32, 110 in f64; fMRI k = 70, recsys
// k = 50), otherwise in its global scratch (L2-resident).  The factor is stored
// symmetrically, F[i][j] = L[max(i,j)][min(i,j)], so that both substitutions read
// row j of F contiguously.  The substitutions run one right-hand side per
// wavefront with the same register layout as the coordinate-descent solver.
// Wide systems (k > 512, e.g. the 1024 maps of the reference's HCP experiment, exps/hcp/decompose_hcp.py:50-60): a
// right-looking BLOCKED factorisation over the whole chip - per 64-column block one small workgroup factors the
// diagonal block in LDS and inverts it, the panel below it and the symmetric trailing update are matrix-core
// products (launch_gemm) - and blocked substitutions for all right-hand sides at once, the diagonal blocks applied
// through their inverses (two products per block and direction).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "gemm.hpp"
#include "chol_small.hpp"
#include "kernels.hpp"

namespace modl {

constexpr size_t kCholLdsBytes = 160 * 1024 - 512;   // the matrix of the one-workgroup kernel lives in LDS up to this size

// code[idx[r]] (or code[r]) <- rows[r]
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_any_kernel(T *dst, int64_t ld, const int64_t *idx, int n_rows, int cols,
                                                               const T *src) {
    const int r = blockIdx.x;
    if (r >= n_rows) return;
    T *d = dst + (idx ? idx[r] : (int64_t)r) * ld;
    for (int c = threadIdx.x; c < cols; c += 256) d[c] = src[(int64_t)r * cols + c];
}

template <typename T>
__device__ void cholesky_inplace(T *W, int k) {
    // right-looking, symmetric trailing update so every access is row-contiguous
    for (int j = 0; j < k; ++j) {
        __syncthreads();                                   // trailing update of column j - 1 done
        const T d = sqrt(W[(int64_t)j * k + j]);
        for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) {
            const T l = W[(int64_t)j * k + i] / d;
            W[(int64_t)j * k + i] = l;
            W[(int64_t)i * k + j] = l;
        }
        __syncthreads();                                   // scaled row visible; every read of W[j][j] is behind us,
        if (threadIdx.x == 0) W[(int64_t)j * k + j] = d;   // so the pivot can go in now (two barriers per column, not three)
        const int n = k - j - 1;
        const T *lrow = W + (int64_t)j * k + (j + 1);
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
            const int i = e / n, m = e % n;
            T *t = W + (int64_t)(j + 1 + i) * k + (j + 1 + m);
            *t = fma(-lrow[i], lrow[m], *t);
        }
    }
    __syncthreads();
}

template <typename T, bool LDS>
__global__ __launch_bounds__(1024) void cholesky_kernel(const T *G, int64_t g_stride, const int64_t *g_idx, T *F, int k,
                                                        T alpha) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const T *g = G + (g_idx ? g_idx[blockIdx.x] : (int64_t)blockIdx.x) * g_stride;
    T *f = F + (int64_t)blockIdx.x * k * k;
    T *W = LDS ? reinterpret_cast<T *>(smem_raw) : f;
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
        T v = g[e];
        if (e / k == e % k) v += alpha;
        W[e] = v;
    }
    cholesky_inplace<T>(W, k);
    if (LDS) {
        for (int e = threadIdx.x; e < k * k; e += blockDim.x) f[e] = W[e];
    }
}

template <typename T>
int launch_cholesky(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *g_idx, T *F, int k, T alpha,
                    int nmat) {
    if (nmat <= 0 || k <= 0) return MODL_OK;
    const size_t bytes = (size_t)k * k * sizeof(T);
    const int threads = k <= 64 ? 256 : 1024;
    if (bytes <= kCholLdsBytes) {
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cholesky_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCholLdsBytes));
        hipLaunchKernelGGL((cholesky_kernel<T, true>), dim3(nmat), dim3(threads), bytes, stream, G, g_stride, g_idx, F,
                           k, alpha);
    } else {
        hipLaunchKernelGGL((cholesky_kernel<T, false>), dim3(nmat), dim3(threads), 0, stream, G, g_stride, g_idx, F, k,
                           alpha);
    }
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_cholesky<float>(hipStream_t, const float *, int64_t, const int64_t *, float *, int, float, int);
template int launch_cholesky<double>(hipStream_t, const double *, int64_t, const int64_t *, double *, int, double, int);

template <typename T, int KPL>
__global__ __launch_bounds__(256) void chol_solve_kernel(const T *F, int64_t f_stride, T *rhs, int b, int k, T *code,
                                                         const int64_t *idx) {
    const int lane = threadIdx.x & 63;
    const int smp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (smp >= b) return;
    const T *__restrict__ Fm = F + (int64_t)smp * f_stride;
    T *r = rhs + (int64_t)smp * k;
    const int e0 = lane * KPL;
    T y[KPL], dg[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool in = e0 + c < k;
        const int ec = in ? e0 + c : k - 1;
        const T yv = r[ec], dv = Fm[(int64_t)ec * k + ec];
        y[c] = in ? yv : (T)0;
        dg[c] = in ? dv : (T)1;
    }
    const int n_li = (k + KPL - 1) / KPL;
    // forward: L y = rhs, column-oriented (row j of F right of the diagonal = column j of L)
    for (int li = 0; li < n_li; ++li) {
        T rows[KPL][KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            const T *row = Fm + (int64_t)(j < k ? j : 0) * k;
#pragma unroll
            for (int c2 = 0; c2 < KPL; ++c2) {               // unconditional clamped load, then the selection
                const T rv = row[e0 + c2 < k ? e0 + c2 : k - 1];
                rows[c][c2] = (e0 + c2 < k) ? rv : (T)0;
            }
        }
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            if (j < k) {
                const T yj = bcast_lane(y[c], li) / bcast_lane(dg[c], li);
                if (lane == li) y[c] = yj;
#pragma unroll
                for (int c2 = 0; c2 < KPL; ++c2)
                    if (e0 + c2 > j) y[c2] = fma(-yj, rows[c][c2], y[c2]);
            }
        }
    }
    // backward: L^T x = y (row j of F left of the diagonal = row j of L)
    for (int li = n_li - 1; li >= 0; --li) {
        T rows[KPL][KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            const T *row = Fm + (int64_t)(j < k ? j : 0) * k;
#pragma unroll
            for (int c2 = 0; c2 < KPL; ++c2) {               // unconditional clamped load, then the selection
                const T rv = row[e0 + c2 < k ? e0 + c2 : k - 1];
                rows[c][c2] = (e0 + c2 < k) ? rv : (T)0;
            }
        }
#pragma unroll
        for (int c = KPL - 1; c >= 0; --c) {
            const int j = li * KPL + c;
            if (j < k) {
                const T xj = bcast_lane(y[c], li) / bcast_lane(dg[c], li);
                if (lane == li) y[c] = xj;
#pragma unroll
                for (int c2 = 0; c2 < KPL; ++c2)
                    if (e0 + c2 < j) y[c2] = fma(-xj, rows[c][c2], y[c2]);
            }
        }
    }
    T *out = code + (idx ? idx[smp] : (int64_t)smp) * k;
#pragma unroll
    for (int c = 0; c < KPL; ++c)
        if (e0 + c < k) {
            r[e0 + c] = y[c];      // the reference leaves the solution in Dx (:188-197)
            out[e0 + c] = y[c];
        }
}

template <typename T>
int launch_chol_solve(hipStream_t stream, const T *F, int64_t f_stride, T *rhs, int b, int k, T *code,
                      const int64_t *idx) {
    if (b <= 0 || k <= 0) return MODL_OK;
    if (k > 512) return MODL_EINVAL;
    dim3 grid((unsigned)cdiv(b, 4)), block(256);
    if (k <= 64) hipLaunchKernelGGL((chol_solve_kernel<T, 1>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else if (k <= 128) hipLaunchKernelGGL((chol_solve_kernel<T, 2>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else if (k <= 256) hipLaunchKernelGGL((chol_solve_kernel<T, 4>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else hipLaunchKernelGGL((chol_solve_kernel<T, 8>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_chol_solve<float>(hipStream_t, const float *, int64_t, float *, int, int, float *, const int64_t *);
template int launch_chol_solve<double>(hipStream_t, const double *, int64_t, double *, int, int, double *, const int64_t *);


// ---- small systems (k <= 128): factor and solve in ONE launch, the matrix in LDS ---------------------------------------
// fMRI k = 70, recsys k = 50.  cholesky_kernel + chol_solve_kernel took 60 + 25 us at k = 70 (two workgroup barriers per
// column with 16 wavefronts, the substitutions reading the factor from L2 one row per round trip).  Here the matrix is
// factored left-looking, four columns per pass: lane l owns rows l and l + 64 and forms
//     c_i = A[i][j] - sum_{m < j} L[i][m] L[j][m]
// for the four columns of the pass with its own rows read once (stride ld between lanes: odd, conflict-free) and the
// four pivot rows as broadcasts; the 4 x 4 diagonal block is resolved with lane broadcasts.  The substitutions run
// one wavefront per batch of NR right-hand sides, column-oriented, the factor's column (forward) or row (backward) read
// once per step for the whole batch, one step ahead of its use.  Same operations as the textbook loops up to the order
// of the sums and the reciprocal pivots.
constexpr int kRidgeSmallMax = 128;
extern std::atomic<unsigned long long *> g_atom_stamps;   // bcd.hip (diagnostics)

// blockIdx.x: the system (g_stride == 0: one shared matrix and b right-hand sides; else one matrix and ONE right-hand side
// per workgroup).  rhs rows are solved in place and also written to code[idx ? idx[r] : r].
template <typename T, int RPL, int NR>
__global__ __launch_bounds__(256) void ridge_small_kernel(const T *G, int64_t g_stride, const int64_t *g_idx, T *rhs, int b,
                                                          int k, T alpha, T *code, const int64_t *idx,
                                                          unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) char ridge_smem[];
    const unsigned long long t0 = clock64();
    T *W = reinterpret_cast<T *>(ridge_smem);
    const int ld = k | 1;                                               // odd row stride: a column is conflict-free
    T *dinv = W + (size_t)ld * k;
    T *part = dinv + ((k + 3) & ~3);                                    // [3][RPL * 4][64]: partial sums of wavefronts 1-3
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const T *g = G + (g_stride ? (g_idx ? g_idx[blockIdx.x] : (int64_t)blockIdx.x) * g_stride : 0);
    // the right-hand sides of this wavefront's first batch are requested with the matrix (one round trip for both)
    const int first = g_stride ? (int)blockIdx.x : 0, count = g_stride ? 1 : b;
    T y[RPL][NR];
    auto load_rhs = [&](int r0) {
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int r = first + ((r0 + t < count) ? r0 + t : count - 1);
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int e = lane + 64 * q;
                y[q][t] = rhs[(int64_t)r * k + (e < k ? e : k - 1)];
            }
        }
    };
    load_rhs(wid * NR);
    constexpr int NL = 8;                                               // matrix elements per thread in flight
    for (int base = 0; base < k * k; base += NL * 256) {
        T v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int e = base + threadIdx.x + 256 * u;
            v[u] = g[e < k * k ? e : k * k - 1];
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int e = base + threadIdx.x + 256 * u;
            if (e < k * k) {
                const int i = e / k, j = e - i * k;
                W[i * ld + j] = (i == j) ? v[u] + alpha : v[u];
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    chol_block_lds<T, RPL>(W, k, ld, dinv, part);
    const unsigned long long t2 = clock64();
    for (int r0 = wid * NR; r0 < count; r0 += 4 * NR) {
        if (r0 != wid * NR) load_rhs(r0);
#pragma unroll
        for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int q = 0; q < RPL; ++q) y[q][t] = (lane + 64 * q < k) ? y[q][t] : (T)0;
        chol_solve_wave_lds<T, RPL, NR>(W, dinv, k, ld, y);
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            if (r0 + t >= count) break;
            const int r = first + r0 + t;
            T *out = code + (idx ? idx[r] : (int64_t)r) * k;
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int e = lane + 64 * q;
                if (e < k) { rhs[(int64_t)r * k + e] = y[q][t]; out[e] = y[q][t]; }
            }
        }
    }
    if (dbg && threadIdx.x == 0) { dbg[40] += 1; dbg[41] += t1 - t0; dbg[42] += t2 - t1; dbg[43] += clock64() - t2; }
}

static size_t ridge_small_lds(int k, size_t tsz) { return ((size_t)(k | 1) * k + ((k + 3) & ~3) + 3 * (k <= 64 ? 4 : 8) * 64) * tsz; }
template <typename T>
bool ridge_small_applies(int k) { return k <= kRidgeSmallMax && ridge_small_lds(k, sizeof(T)) <= 150 * 1024; }

template <typename T>
int launch_ridge_small(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *g_idx, T *rhs, int b, int k, T alpha,
                       T *code, const int64_t *idx) {
    if (b <= 0 || k <= 0) return MODL_OK;
    if (!ridge_small_applies<T>(k)) return MODL_EINVAL;
    const size_t lds = ridge_small_lds(k, sizeof(T));                    // the matrix, the reciprocal pivots, the partial sums
    const unsigned grid = g_stride ? (unsigned)b : 1u;
#define MODL_RIDGE_SMALL(RPL, NR)                                                                                          \
    do {                                                                                                                   \
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ridge_small_kernel<T, RPL, NR>),                       \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                             \
        hipLaunchKernelGGL((ridge_small_kernel<T, RPL, NR>), dim3(grid), dim3(256), lds, stream, G, g_stride, g_idx, rhs, b, \
                           k, alpha, code, idx, g_atom_stamps.load());                                                     \
    } while (0)
    // right-hand sides per wavefront: the whole minibatch in one pass of the four wavefronts while it fits 6 each (the
    // chains of a batch interleave; beyond that a step is bound by its instruction count and further passes follow)
    const int per_wave = g_stride ? 1 : (int)cdiv(b, 4);
    // (a step of the substitutions costs its instruction count, i.e. grows with the batch: the smallest batch that covers
    // the minibatch in one pass - 20 records: 5 per wavefront, not 6 + 6 + 6 + 2)
    if (k <= 64) {
        if (per_wave <= 1) MODL_RIDGE_SMALL(1, 1);
        else if (per_wave <= 2) MODL_RIDGE_SMALL(1, 2);
        else if (per_wave <= 3) MODL_RIDGE_SMALL(1, 3);
        else if (per_wave <= 4) MODL_RIDGE_SMALL(1, 4);
        else if (per_wave <= 5) MODL_RIDGE_SMALL(1, 5);
        else MODL_RIDGE_SMALL(1, 6);
    } else {
        if (per_wave <= 1) MODL_RIDGE_SMALL(2, 1);
        else if (per_wave <= 2) MODL_RIDGE_SMALL(2, 2);
        else if (per_wave <= 3) MODL_RIDGE_SMALL(2, 3);
        else if (per_wave <= 4) MODL_RIDGE_SMALL(2, 4);
        else if (per_wave <= 5) MODL_RIDGE_SMALL(2, 5);
        else MODL_RIDGE_SMALL(2, 6);
    }
#undef MODL_RIDGE_SMALL
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template bool ridge_small_applies<float>(int);
template bool ridge_small_applies<double>(int);
template int launch_ridge_small<float>(hipStream_t, const float *, int64_t, const int64_t *, float *, int, int, float, float *,
                                       const int64_t *);
template int launch_ridge_small<double>(hipStream_t, const double *, int64_t, const int64_t *, double *, int, int, double,
                                        double *, const int64_t *);

// ---- wide systems: blocked factorisation / substitutions ----------------------------------------------------
constexpr int kCholNB = 64;

// out(i, r) -> W[row0 + i][col0 + r] and its mirror image (the factor is stored symmetrically)
template <typename T> struct EpiPanelSym {
    T *W; int64_t ld, row0, col0;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const {
        W[(row0 + m) * ld + col0 + n] = v;
        W[(col0 + n) * ld + row0 + m] = v;
    }
};

// factor the nb x nb diagonal block at (j0, j0) of W in LDS, write it back (symmetric storage) and write the
// inverse of its lower-triangular factor to Linv[nb][nb] (row-major, zeros above the diagonal)
// Diagonal block of the blocked factorisation on ONE wavefront, entirely in registers: lane i owns row i of the block (kCholNB values).  Column j: the pivot
// by v_readlane, every lane writes its own l_i = a_i[j] / d to LDS - by symmetry that is also the scaled row j every
// lane needs for its rank-1 update - reads it back as broadcast 16-byte words and updates its row: no workgroup
// barrier, no integer division per element (a 256-thread version with two barriers per column took 2.5 us per column, 162 us per
// block, 57 % of a ridge minibatch at k = 256).  Then L^-1 column by column (lane t owns column t), the rows of L
// as LDS broadcasts.  Rows / columns beyond nb are an identity, so ragged blocks need no special case.
template <typename T>
__global__ __launch_bounds__(64) void chol_diag_wave_kernel(T *W, int k, int j0, int nb, T *Linv) {
    constexpr int N = kCholNB;
    // (one struct: `col` at LDS offset 0, so that its broadcast reads are immediate offsets from one base register -
    // with the array behind Ls the compiler kept a base register per pair of reads and spilled)
    struct Lds { T col[N]; T invd[N]; T L[N][N + 1]; };     // invd: 1 / L[i][i]
    __shared__ __attribute__((aligned(16))) Lds lds;
    T (&col)[N] = lds.col;
    T (&invd)[N] = lds.invd;
    T (&Ls)[N][N + 1] = lds.L;
    const int lane = threadIdx.x;
    // the block -> LDS, row by row (coalesced; an identity beyond nb), then row `lane` -> registers
    // (two passes, both unrolled: all requests first - as one loop the compiler waits for every row's round trip in turn,
    // 64 of them)
    {
        T raw[N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const bool in = r < nb && lane < nb;
            raw[r] = W[(int64_t)(j0 + (in ? r : 0)) * k + j0 + (in ? lane : 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < N; ++r) Ls[r][lane] = (r < nb && lane < nb) ? raw[r] : ((r == lane) ? (T)1 : (T)0);
    }
    __builtin_amdgcn_wave_barrier();
    T a[N];
#pragma unroll
    for (int m = 0; m < N; ++m) a[m] = Ls[lane][m];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // (scheduler fences per column: in one basic block of 64 unrolled columns the scheduler hoists the LDS reads
        // of many columns and spills - 1400 scratch accesses, 127 us per block)
        __builtin_amdgcn_sched_barrier(0);
        const T d = sqrt(bcast_lane(a[j], j));
        const T id = (T)1 / d;                             // one division per column (wave-uniform), products after it
        const T l = a[j] * id;                             // lane i: L[i][j] (lanes < j hold already final values there)
        col[lane] = l;
        if (lane == j) invd[j] = id;
        __builtin_amdgcn_wave_barrier();                   // (LDS operations of one wavefront complete in order)
        a[j] = (lane == j) ? d : l;
#pragma unroll
        for (int m = j + 1; m < N; ++m) {                  // rows <= j: garbage in columns > j, never read
            a[m] = fma(-l, col[m], a[m]);
            asm volatile("" : "+v"(a[m]));                 // (computed HERE: the compiler otherwise sinks every update to
        }                                                  //  the column that reads it and keeps 64 columns of operands live)
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    // L (lower triangle of row `lane` in a[0 .. lane]) -> LDS, then the symmetric block back to memory
#pragma unroll
    for (int m = 0; m < N; ++m) Ls[lane][m] = (m <= lane) ? a[m] : (T)0;
    __builtin_amdgcn_wave_barrier();
    if (lane < nb) {
        T *row = W + (int64_t)(j0 + lane) * k + j0;
#pragma unroll
        for (int m = 0; m < N; ++m)
            if (m < nb) row[m] = (m <= lane) ? Ls[lane][m] : Ls[m][lane];
    }
    // X = L^-1: lane t owns column t; x_i = ([i == t] - sum_{m < i} L[i][m] x_m) / L[i][i]  (zero above the diagonal)
    T x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        T acc = (lane == i) ? (T)1 : (T)0;
        const T *Lr = &Ls[i][0];
#pragma unroll
        for (int m = 0; m < i; ++m) acc = fma(-Lr[m], x[m], acc);
        x[i] = acc * invd[i];
        asm volatile("" : "+v"(x[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (lane < nb) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < nb) Linv[i * nb + lane] = x[i];
    }
}


template <typename T>
__global__ __launch_bounds__(256) void chol_load_kernel(const T *g, T *W, int k, T alpha) {
    const int64_t n = (int64_t)k * k, stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        T v = g[e];
        if (e / k == e % k) v += alpha;
        W[e] = v;
    }
}

// F <- factor of (g + alpha I), Linv <- inverses of its diagonal blocks ([ceil(k / 64)][64 * 64])
template <typename T>
int cholesky_blocked(hipStream_t stream, const T *g, T *F, T *Linv, int k, T alpha) {
    hipLaunchKernelGGL((chol_load_kernel<T>), dim3((unsigned)std::min<int64_t>(cdiv((int64_t)k * k, 256), 2048)), dim3(256), 0,
                       stream, g, F, k, alpha);
    MODL_LAUNCH_CHECK();
    SplitWs none;
    for (int j0 = 0, blk = 0; j0 < k; j0 += kCholNB, ++blk) {
        const int nb = std::min(kCholNB, k - j0);
        T *Li = Linv + (size_t)blk * kCholNB * kCholNB;
        hipLaunchKernelGGL((chol_diag_wave_kernel<T>), dim3(1), dim3(64), 0, stream, F, k, j0, nb, Li);
        MODL_LAUNCH_CHECK();
        const int64_t rest = k - j0 - nb;
        if (rest <= 0) break;
        // panel: L[i][r] = sum_c A[i][j0 + c] Linv[r][c]  (in place: a workgroup reads its own rows before it writes them)
        Operand A, B;
        A.ptr = F + (size_t)(j0 + nb) * k + j0; A.si = k; A.sk = 1;
        B.ptr = Li; B.si = nb; B.sk = 1;
        EpiPanelSym<T> ep{F, k, j0 + nb, j0};
        MODL_TRY((launch_gemm<T, EpiPanelSym<T>>(stream, A, B, rest, nb, nb, ep, none, nullptr, 512, 1)));
        // trailing update: W[i][m] -= sum_r L[i][r] L[m][r]  (both triangles: the storage stays symmetric)
        Operand P;
        P.ptr = F + (size_t)(j0 + nb) * k + j0; P.si = k; P.sk = 1;
        EpiAxpby<T> et{F + (size_t)(j0 + nb) * k + (j0 + nb), k, (T)-1, (T)1};
        MODL_TRY((launch_gemm<T, EpiAxpby<T>>(stream, P, P, rest, rest, nb, et, none, nullptr, 512, 1)));
    }
    return MODL_OK;
}

// rhs[b][k] <- solutions of F F^T x = rhs (in place), block substitutions on the matrix cores
template <typename T>
int chol_solve_blocked(hipStream_t stream, const T *F, const T *Linv, T *rhs, int b, int k) {
    SplitWs none;
    const int nblk = (int)cdiv(k, kCholNB);
    for (int blk = 0; blk < nblk; ++blk) {                             // forward: L y = rhs
        const int j0 = blk * kCholNB, nb = std::min(kCholNB, k - j0);
        const T *Li = Linv + (size_t)blk * kCholNB * kCholNB;
        Operand R, B;
        R.ptr = rhs + j0; R.si = k; R.sk = 1;                          // (s, c) -> rhs[s][j0 + c]
        B.ptr = Li; B.si = nb; B.sk = 1;                               // (r, c) -> Linv[r][c]
        EpiStore<T> ey{rhs + j0, k, (T)1};
        MODL_TRY((launch_gemm<T, EpiStore<T>>(stream, R, B, b, nb, nb, ey, none, nullptr, 512, 1)));
        const int64_t rest = k - j0 - nb;
        if (rest <= 0) break;
        Operand Lp;
        Lp.ptr = F + (size_t)(j0 + nb) * k + j0; Lp.si = k; Lp.sk = 1;  // (m, r) -> L[j0 + nb + m][j0 + r]
        EpiAxpby<T> eu{rhs + j0 + nb, k, (T)-1, (T)1};
        MODL_TRY((launch_gemm<T, EpiAxpby<T>>(stream, R, Lp, b, rest, nb, eu, none, nullptr, 512, 1)));
    }
    for (int blk = nblk - 1; blk >= 0; --blk) {                        // backward: L^T x = y
        const int j0 = blk * kCholNB, nb = std::min(kCholNB, k - j0);
        const T *Li = Linv + (size_t)blk * kCholNB * kCholNB;
        Operand R, B;
        R.ptr = rhs + j0; R.si = k; R.sk = 1;
        B.ptr = Li; B.si = 1; B.sk = nb;                               // (r, c) -> Linv[c][r]
        EpiStore<T> ex{rhs + j0, k, (T)1};
        MODL_TRY((launch_gemm<T, EpiStore<T>>(stream, R, B, b, nb, nb, ex, none, nullptr, 512, 1)));
        if (j0 == 0) break;
        Operand Lr;
        Lr.ptr = F + j0; Lr.si = k; Lr.sk = 1;                          // (m, r) -> L[j0 + r][m] = F[m][j0 + r]
        EpiAxpby<T> eu{rhs, k, (T)-1, (T)1};
        MODL_TRY((launch_gemm<T, EpiAxpby<T>>(stream, R, Lr, b, j0, nb, eu, none, nullptr, 512, 1)));
    }
    return MODL_OK;
}

size_t chol_wide_scratch_elems(int k) { return k > 128 ? (size_t)cdiv(k, kCholNB) * kCholNB * kCholNB : 0; }
// A SHARED Gram beyond ~160 atoms goes the blocked way whatever k is: the one-workgroup kernel pays two barriers per column
// (two memory round trips per column once the matrix no longer fits its LDS: 1.7 ms at k = 256, measured - the whole
// minibatch with l1 codes takes 0.25 ms).  One Gram per sample stays with one workgroup per sample up to k = 512.
bool chol_blocked(int k, size_t tsz, bool shared) {
    // measured (minibatch of 256, ridge codes): k = 150 0.50 ms in one workgroup / 0.60 blocked, k = 200 0.87 / 0.80
    return k > 512 || (shared && (k > 160 || (size_t)k * k * tsz > kCholLdsBytes));
}

// The ridge solve for k > 512: shared Gram (f_stride == 0: one factorisation, all right-hand sides at once) or one
// Gram per sample (a factorisation and a single right-hand side each - as slow, relatively, as the reference's b
// separate posv calls, dict_fact_fast.pyx:82-94).  F: k*k (or b*k*k is NOT needed: one factor is reused), Linv:
// chol_wide_scratch_elems(k).  Solutions are left in rhs and scattered to code rows idx.
template <typename T>
int ridge_solve_wide(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *h_gidx, T *F, T *Linv, T *rhs, int b,
                     int k, T alpha, T *code, const int64_t *d_idx) {
    if (g_stride == 0) {
        MODL_TRY(cholesky_blocked<T>(stream, G, F, Linv, k, alpha));
        MODL_TRY(chol_solve_blocked<T>(stream, F, Linv, rhs, b, k));
    } else {
        for (int i = 0; i < b; ++i) {
            const T *g = G + (h_gidx ? h_gidx[i] : (int64_t)i) * g_stride;
            MODL_TRY(cholesky_blocked<T>(stream, g, F, Linv, k, alpha));
            MODL_TRY(chol_solve_blocked<T>(stream, F, Linv, rhs + (size_t)i * k, 1, k));
        }
    }
    hipLaunchKernelGGL((scatter_rows_any_kernel<T>), dim3((unsigned)b), dim3(256), 0, stream, code, (int64_t)k, d_idx, b, k, rhs);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int ridge_solve_wide<float>(hipStream_t, const float *, int64_t, const int64_t *, float *, float *, float *, int, int,
                                     float, float *, const int64_t *);
template int ridge_solve_wide<double>(hipStream_t, const double *, int64_t, const int64_t *, double *, double *, double *, int,
                                      int, double, double *, const int64_t *);

}  // namespace modl
