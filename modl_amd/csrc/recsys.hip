// Masked (missing-data) path of RecsysDictFact on CSR ratings.
//
// Replaces the per-sample Python loop of the reference
//   (modl/decomposition/recsys.py:147-185 _single_batch_fit / _single_sample_update,
//    :254-265 _refit, and recsys_fast.pyx:10-38 _predict).
// For a row i with observed columns S_i and values x_S:
//     G = D_S D_S^T + (alpha |S_i| / p) I,   code_i = G^{-1} D_S x_S               (:176-181)
//     w_B[f] = min(1, w n_iter / feature_n_iter[f]),  B[:, f] = (1 - w_B) B[:, f] + code_i (x_f w_B)   (:182-185)
// The dictionary is fixed inside a minibatch, so the b code solves are independent (one workgroup per
// row: Gram accumulated in LDS from the feature-major dictionary rows — a rated item is one contiguous
// k-vector — then an in-LDS Cholesky); the B update is order-dependent per feature (two rows of the
// batch rating the same item do not commute), so it runs one wavefront per touched feature, walking
// that feature's entries in batch order.  C_ and the block-coordinate update on the union of the
// batch's columns (recsys.py:159-165,187-213: squared-l2 budget, clip-only projection) reuse the
// kernels of the dense path (csrc/bcd.hip blocked path is exactly that update).
#include "gemm.hpp"
#include "kernels.hpp"
#include "chol_small.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <vector>

namespace modl {

// RPL: rows of the system per lane of the factorising wavefront (1: k <= 64, 2: k <= 128; chol_small.hpp), 0: any k that
// fits LDS, two workgroup barriers per column
template <typename T, int RPL>
__global__ __launch_bounds__(256) void recsys_code_kernel(const T *Dt, int64_t p, int k, const int32_t *indptr,
                                                          const int32_t *indices, const T *data, const int64_t *row_ids,
                                                          const int64_t *code_rows, double alpha, T *code) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int ld = RPL ? (k | 1) : k;                       // (odd row stride: a column of the factor is conflict-free)
    T *G = reinterpret_cast<T *>(smem_raw);                 // [k][ld]
    T *rhs = G + (size_t)k * ld;                            // [k]
    T *rows = rhs + k;                                      // [32][k] staged dictionary rows
    T *xv = rows + 32 * (size_t)k;                          // [32]
    const int64_t r = row_ids ? row_ids[blockIdx.x] : (int64_t)blockIdx.x;
    const int32_t beg = indptr[r], end = indptr[r + 1];
    const int nnz = end - beg;
    if (nnz == 0) return;                                   // recsys.py:170: rows without ratings keep their code
    for (int e = threadIdx.x; e < k * ld + k; e += 256) G[e] = 0;  // G and rhs are contiguous
    int *ids = reinterpret_cast<int *>(xv + 32);              // [32] item ids of the chunk
    for (int c0 = 0; c0 < nnz; c0 += 32) {
        const int nc = (nnz - c0 < 32) ? nnz - c0 : 32;
        __syncthreads();
        // the chunk's item ids and ratings first (one round trip), then its dictionary rows with every load of a
        // thread in flight together (unconditional, clamped; a dependent id -> row chain per element made the staging
        // a dozen serial round trips per chunk)
        if (threadIdx.x < 32) {
            const int e = beg + c0 + (threadIdx.x < nc ? (int)threadIdx.x : nc - 1);
            ids[threadIdx.x] = indices[e];
            xv[threadIdx.x] = data[e];
        }
        __syncthreads();
        const int tot = nc * k;
        for (int e0 = threadIdx.x; e0 < tot; e0 += 4 * 256) {
            T ld[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 < tot ? e0 + u * 256 : tot - 1;
                ld[u] = Dt[(int64_t)ids[e / k] * k + e % k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 256 < tot) rows[e0 + u * 256] = ld[u];     // rows[j * k + c], e = j * k + c
        }
        __syncthreads();
        for (int e = threadIdx.x; e < k * k; e += 256) {
            const int a = e / k, c = e % k;
            T acc = G[a * ld + c];
            for (int j = 0; j < nc; ++j) acc = fma(rows[j * k + a], rows[j * k + c], acc);
            G[a * ld + c] = acc;
        }
        for (int a = threadIdx.x; a < k; a += 256) {
            T acc = rhs[a];
            for (int j = 0; j < nc; ++j) acc = fma(rows[j * k + a], xv[j], acc);
            rhs[a] = acc;
        }
    }
    __syncthreads();
    const T ridge = (T)(alpha * (double)nnz / (double)p);   // alpha / reduction, reduction = p / |S_i| (:175,179)
    for (int a = threadIdx.x; a < k; a += 256) G[a * ld + a] += ridge;
    if constexpr (RPL > 0) {
        // factor and substitutions of chol_small.hpp (four columns per pass, two barriers per pass; one wavefront solves)
        T *dinv = reinterpret_cast<T *>(ids + 32);            // [k rounded up to 4]
        T *part = dinv + ((k + 3) & ~3);                     // [3][RPL * 4][64]
        __syncthreads();
        chol_block_lds<T, RPL>(G, k, ld, dinv, part);
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            T y[RPL][1];
#pragma unroll
            for (int q = 0; q < RPL; ++q) y[q][0] = (lane + 64 * q < k) ? rhs[lane + 64 * q] : (T)0;
            chol_solve_wave_lds<T, RPL, 1>(G, dinv, k, ld, y);
            T *out = code + (code_rows ? code_rows[blockIdx.x] : r) * k;
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                if (lane + 64 * q < k) out[lane + 64 * q] = y[q][0];
        }
        return;
    }
    // Cholesky G = L L^T in place (symmetric storage; the diagonal of L goes to dg so that nobody overwrites an entry
    // others still read: two barriers per column instead of three), then L y = rhs, L^T x = y with ONE barrier per
    // step (every thread forms the pivot value itself; results go to their own arrays).  Same operations in the same
    // order as the textbook loops: identical bits.
    T *dg = rows, *ys = rows + k, *xs = rows + 2 * k;       // (the staging buffer is free now: 32 k >= 3 k)
    for (int j = 0; j < k; ++j) {
        __syncthreads();                                    // trailing update of column j - 1 (or the ridge) done
        const T d = sqrt(G[j * k + j]);
        for (int i = j + 1 + threadIdx.x; i < k; i += 256) {
            const T l = G[j * k + i] / d;
            G[j * k + i] = l;
            G[i * k + j] = l;
        }
        if (threadIdx.x == 0) dg[j] = d;
        __syncthreads();
        const int n = k - j - 1;
        for (int e = threadIdx.x; e < n * n; e += 256) {
            const int i = e / n, m = e % n;
            T *t = G + (size_t)(j + 1 + i) * k + (j + 1 + m);
            *t = fma(-G[j * k + j + 1 + i], G[j * k + j + 1 + m], *t);
        }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {                          // forward
        const T yj = rhs[j] / dg[j];
        for (int i = j + 1 + threadIdx.x; i < k; i += 256) rhs[i] = fma(-yj, G[j * k + i], rhs[i]);
        if (threadIdx.x == 0) ys[j] = yj;
        __syncthreads();
    }
    for (int j = k - 1; j >= 0; --j) {                     // backward
        const T xj = ys[j] / dg[j];
        for (int i = threadIdx.x; i < j; i += 256) ys[i] = fma(-xj, G[j * k + i], ys[i]);
        if (threadIdx.x == 0) xs[j] = xj;
        __syncthreads();
    }
    T *out = code + (code_rows ? code_rows[blockIdx.x] : r) * k;
    for (int a = threadIdx.x; a < k; a += 256) out[a] = xs[a];
}

// One wavefront per touched feature; entries of that feature in batch order.
template <typename T>
__global__ __launch_bounds__(256) void recsys_update_B_kernel(T *Bt, int k, int64_t *feature_n_iter, const int32_t *subset,
                                                              const int32_t *fptr, const int32_t *entry_sample,
                                                              const T *entry_val, const T *code_b,
                                                              const int64_t *code_rows, double w_n_iter, int64_t u) {
    const int lane = threadIdx.x & 63;
    const int64_t fi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fi >= u) return;
    const int32_t f = subset[fi];
    int64_t n = feature_n_iter[f];
    T *brow = Bt + (int64_t)f * k;
    for (int32_t e = fptr[fi]; e < fptr[fi + 1]; ++e) {
        n += 1;                                              // recsys.py:175 feature_n_iter_[subset] += 1
        double wB = w_n_iter / (double)n;                    // :182-183
        wB = wB < 1.0 ? wB : 1.0;
        const double xw = (double)entry_val[e] * wB;
        const int64_t cpos = entry_sample[e];                // position in the batch; code_rows: its row of code_
        const T *cr = code_b + (code_rows ? code_rows[cpos] : cpos) * k;
        for (int c = lane; c < k; c += 64) {
            T b = (T)((double)brow[c] * (1.0 - wB));         // B_[:, subset] *= 1 - w_B
            b = (T)((double)b + (double)cr[c] * xw);          // += outer(code, X_subset * w_B)
            brow[c] = b;
        }
    }
    if (lane == 0) feature_n_iter[f] = n;
}

// data[ii] = sum_c code[u][c] * Dt[indices[ii]][c]     (recsys_fast.pyx:10-38 with the feature-major dictionary)
template <typename T>
__global__ __launch_bounds__(256) void recsys_predict_kernel(double *out, const int32_t *indices, const int32_t *indptr,
                                                             const T *code, int64_t n_rows, int k, const T *Dt) {
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_rows) return;
    const T *cu = code + u * k;
    for (int32_t ii = indptr[u] + lane; ii < indptr[u + 1]; ii += 64) {
        const T *dr = Dt + (int64_t)indices[ii] * k;
        double dot = 0;
        for (int c = 0; c < k; ++c) dot += (double)cu[c] * (double)dr[c];
        out[ii] = dot;
    }
}

// ---- the masked minibatch as ONE launch (round 6) -------------------------------------------------------------------------
// recsys.py:147-213 for a batch of at most 64 rows, at most 64 (f64: 56) atoms: codes, B_, C_ and the dictionary update in one kernel
// (it was ten launches per minibatch of ten rows: a code launch that lasted as long as the batch's heaviest row, the B_ and C_
// updates, then the blocked dictionary update as separate launches).
//   * CHUNK workgroups: a row's ratings are cut into chunks of 128 (the work is balanced by ratings, not by rows); a chunk's
//     Gram contribution D_S^T D_S and right-hand side are accumulated from the feature-major dictionary rows in 4 x 4 register
//     tiles (the next 32 rows requested while the current 32 are contracted).  A row of several chunks meets in a scratch
//     record per chunk: write-through stores, one ticket per workgroup, the LAST to arrive sums the records in chunk order
//     (a fixed order: run-to-run identical) - then factors and solves the row's k x k system in LDS (chol_small.hpp) and
//     writes code_[row].
//   * the LAST row to finish (a second ticket) goes on alone, 512 threads: C_ = (1 - w) C_ + (w / b) code_[batch]^T code_[batch]
//     (recsys.py:159-160), the per-item B_ update in batch order (:175, :182-185: a wavefront per touched item), and the
//     dictionary update on the touched items (:187-213) with ONE ITEM PER THREAD: the item's k dictionary entries stay in
//     registers in sweep order for the whole sweep; per atom the candidate is (B_j - sum_{i != j} C[i][j] D_i) / C[j][j]
//     against the registers as they are (the reference's two rank-1 passes per atom over a k x u gradient, evaluated lazily),
//     its norm and the atom's old norm are ONE workgroup-wide reduction (wave sums, eight partial sums through LDS, one
//     barrier per atom), the clip to the atom's budget, next atom.  The atom loop is unrolled (the register of atom j is a
//     compile-time index), KP = 32 or 64 (f64: 56) registers per item.
// More than 512 touched items: the kernel stops after B_ and the host runs the blocked dictionary update's launches.
constexpr int kRfMaxChunks = 64, kRfMaxBatch = 64, kRfChunk = 128;
constexpr int kRfMaxSweep = 4;                          // sweep workgroups of 512 items each
constexpr long long kRfSentinel = 0x7ff8dead0000beefll; // a NaN no sum produces
typedef unsigned int rf_u4 __attribute__((ext_vector_type(4)));   // (a native vector: an array of HIP's uint4 structs went to scratch)
struct RecsysChunk { int32_t pos, beg, cnt, nch, ci, part0; };   // row of the batch, first CSR entry, entries, chunks of the row, index among them, first record
template <typename T> struct RecsysFusedArgs {
    const int32_t *indptr, *indices;
    const T *data;
    T *Dt, *Bt, *C, *code, *comp_norm;
    int64_t *feature_n_iter;
    const int32_t *order, *subset, *fptr, *esample;      // the staged minibatch (device copy of the pinned slot)
    const T *eval;
    T *part;                                             // [chunks][k * k + k] records of rows with several chunks
    unsigned int *tickets;                               // [kRfMaxBatch + 1], zero between launches (the last workgroup clears them)
    double *xch;                                         // [(2 KP + KP) x kRfMaxSweep] exchange slots of the sweep workgroups, sentinels between launches
    double alpha, w, w_n_iter;
    int64_t p;
    int k, b, u, n_solve, do_dict, nsweep;
    int cap2;                                            // items of a sweep workgroup's LDS tier (a multiple of 64, at most 512)
    unsigned long long *dbg;                             // optional: 100 MHz wall-clock stamps of the LAST workgroup (modl_recsys_plan_stamps)
    // the NEXT minibatch's pinned slot -> the other device staging buffer, by one more workgroup at the end of the grid (the
    // run of minibatches of modl_recsys_fit_batches_*: no staging launch between two minibatches); stage_n16 == 0: none
    const rf_u4 *stage_src;
    rf_u4 *stage_dst;
    size_t stage_n16;
    unsigned long long *stage_ack, stage_use;
    int64_t rows[kRfMaxBatch];
    RecsysChunk ch[kRfMaxChunks];
};

template <typename T> __device__ __forceinline__ void rf_store(T *ptr, T v);
template <> __device__ __forceinline__ void rf_store<float>(float *ptr, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int *>(ptr), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <> __device__ __forceinline__ void rf_store<double>(double *ptr, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ptr), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float rf_load(const float *ptr) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double rf_load(const double *ptr) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

// the widest variant: an item's atoms in registers for the whole sweep, 512 threads = two waves per SIMD = 256 registers per
// thread: 64 floats, or 56 doubles (64 doubles spill; a masked minibatch with more atoms takes the separate launches)
template <typename T> struct kRfWide { static constexpr int value = sizeof(T) == 4 ? 64 : 56; };

template <int J, int N, typename F> __device__ __forceinline__ void rf_static_for(F &&f) {
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        rf_static_for<J + 1, N>(f);
    }
}

// a workgroup barrier that orders LDS traffic only (global requests stay in flight across it)
__device__ __forceinline__ void rf_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// D[jj] = v for a wavefront-uniform jj, every register index a compile-time constant: a uniform branch to the group of eight
// registers, eight selects
template <typename T, int KP> __device__ __forceinline__ void rf_set(T (&D)[KP], int jj, T v) {
    static_assert(KP % 8 == 0, "groups of eight registers");
    const int g = jj >> 3, r = jj & 7;
    rf_static_for<0, KP / 8>([&](auto G) {
        constexpr int gg = decltype(G)::value;
        if (g == gg) {
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                D[8 * gg + x] = (r == x) ? v : D[8 * gg + x];
                // (opaque: the optimiser recognises the chain of selects as an insertion at a variable index and moves the
                //  whole array to scratch)
                asm volatile("" : "+v"(D[8 * gg + x]));
            }
        }
    });
}

// The row's k x k system (k <= KC) factored and solved by ONE wavefront with the matrix in REGISTERS: lane i holds row i, column j
// is a compile-time register index (the loops are unrolled), the pivot row's entries reach the other lanes as v_readlane
// broadcasts: right-looking, G[i][m] -= L[i][j] L[m][j] - no LDS traffic, no barrier.  (chol_block_lds + chol_solve_wave_lds, four
// wavefronts working on the matrix in LDS with two barriers per four columns, took 18.5 + 9.2 us at k = 50; this one 23.8.)  Rows and columns
// beyond k are the identity: no branch anywhere.  The reciprocal pivot: rsq + two Newton steps (f64) / one (f32), d = pivot * inv.
// Backward substitution: x_j = (y_j - sum_{m > j} L[m][j] x_m) / L[j][j] with the sum over the lanes (a wave sum per column).
// (Measured and not kept: the pivot column through LDS - one store, then broadcast reads - instead of a v_readlane pair per entry:
//  half the instructions, 23.8 -> 30.6 us: each read waits for its LDS round trip on the lone wavefront.)
template <typename T, int KC>
__device__ __forceinline__ T chol_solve_wave_reg(const T *G, int ld, int k, const T *rhs) {
    const int lane = threadIdx.x & 63;
    T R[KC];
#pragma unroll
    for (int m = 0; m < KC; ++m) {
        const T v = G[(lane < k ? lane : 0) * ld + (m < k ? m : 0)];
        R[m] = (lane < k && m < k) ? v : ((lane == m) ? (T)1 : (T)0);
    }
    T b = (lane < k) ? rhs[lane] : (T)0;
    T myinv = 1;
    rf_static_for<0, KC>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const T piv = bcast_lane(R[j], j);
        T inv;
        if constexpr (sizeof(T) == 8) {
            double y0 = __builtin_amdgcn_rsq(piv);
            y0 = y0 * fma(-0.5 * piv * y0, y0, 1.5);
            inv = y0 * fma(-0.5 * piv * y0, y0, 1.5);
        } else {
            const float y0 = __builtin_amdgcn_rsqf(piv);
            inv = y0 * fmaf(-0.5f * piv * y0, y0, 1.5f);
        }
        const T l = R[j] * inv;                                  // lane i: L[i][j] (lane j: the pivot's square root)
        R[j] = l;
        myinv = (lane == j) ? inv : myinv;
        rf_static_for<j + 1, KC>([&](auto M) {
            constexpr int m = decltype(M)::value;
            R[m] = fma(-l, bcast_lane(l, m), R[m]);
        });
    });
    // L y = b (column j of L is register j of the lanes below j)
    rf_static_for<0, KC>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const T yj = bcast_lane(b * myinv, j);
        b = (lane == j) ? yj : ((lane > j) ? fma(-R[j], yj, b) : b);
    });
    // L^T x = y
    T x = 0;
    rf_static_for<0, KC>([&](auto J) {
        constexpr int j = KC - 1 - decltype(J)::value;
        const T t = wave_sum((lane > j) ? R[j] * x : (T)0);      // sum_{m > j} L[m][j] x_m
        const T xj = bcast_lane((b - t) * myinv, j);
        x = (lane == j) ? xj : x;
    });
    return x;
}

template <typename T>
size_t recsys_fused_lds(int k, int b, int KP, int cap2) {
    const size_t KS = (size_t)((k + 3) & ~3), ld = (size_t)(k | 1);
    const size_t chunk = sizeof(T) * ((size_t)k * ld + 4 + 2 * KS + 768 + 2 * 32 * (size_t)((k + 15) & ~15) + kRfChunk) + sizeof(int) * (kRfChunk + 8);
    const size_t fin = sizeof(T) * ((size_t)k * KP + (size_t)b * k + 3 * (size_t)KP) + sizeof(double) * (2 * 8 * 2 + 8 * (size_t)KP) +
                       ((sizeof(int) * ((size_t)KP + 8) + 15) & ~(size_t)15) + sizeof(T) * (size_t)KP * cap2 + 16;
    return (chunk > fin ? chunk : fin) + 64;
}

template <typename T, int KP>
__global__ __launch_bounds__(512) void recsys_fused_kernel(const RecsysFusedArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int k = a.k;
    int me = 0;                                                 // which of the sweep workgroups this one becomes (0: the last to finish)
    unsigned long long stp[8];                                  // stamps of the chunk phase (kept only by the last workgroup)
    stp[0] = a.dbg ? wall_clock64() : 0;
    if (a.stage_n16 && blockIdx.x == gridDim.x - 1) {           // the rider: the next minibatch's staged arrays
        for (size_t i0 = tid; i0 < a.stage_n16; i0 += 8 * 512) {
            rf_u4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = a.stage_src[(i0 + 512 * q < a.stage_n16) ? i0 + 512 * q : a.stage_n16 - 1];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + 512 * q < a.stage_n16) a.stage_dst[i0 + 512 * q] = v[q];
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.stage_ack, a.stage_use, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (a.n_solve > 0) {
        // ---------------------------------------------------------------- a chunk of a row
        const int ld = k | 1, KS = (k + 3) & ~3;
        T *G = reinterpret_cast<T *>(smem_raw);                 // [k][ld]
        T *rhs = G + (((size_t)k * ld + 3) & ~(size_t)3);       // [KS] (16-byte aligned, like everything behind it)
        T *dinv = rhs + KS;                                     // [KS]
        T *cpart = dinv + KS;                                   // [3][4][64]
        T *rowsb = cpart + 768;                                 // [2][32][KQ], KQ = k rounded up to 16
        T *xv = rowsb + 2 * 32 * ((k + 15) & ~15);              // [kRfChunk]
        int *ids = reinterpret_cast<int *>(xv + kRfChunk);      // [kRfChunk]
        int *flag = ids + kRfChunk;
        const RecsysChunk c = a.ch[blockIdx.x];
        const int64_t r = a.rows[c.pos];
        if (tid < kRfChunk) {
            const int e = c.beg + (tid < c.cnt ? tid : c.cnt - 1);
            ids[tid] = a.indices[e];
            xv[tid] = a.data[e];
        }
        __syncthreads();
        // The chunk's Gram contribution on the MATRIX CORES (16 x 16 x 4 tiles, f64 or f32: G[i][j] += sum_items D[item][i] D[item][j];
        // both operands of a tile are columns of the staged rows, four items per instruction), two tiles per wavefront and
        // sub-chunk of 32 items.  (First version: 4 x 4 register tiles on 169 threads, 1.7 us per sub-chunk, plus - eight times
        // per sub-chunk and thread - an integer division by the row stride in the staging loops: 4.5 us per sub-chunk, measured.)
        const int KT = (k + 15) >> 4, KQ = KT * 16;             // tiles per side, padded row stride of the staged rows
        const int ra = tid - 256;                               // the right-hand side's threads: 256 .. 256 + k - 1
        typedef typename std::conditional<sizeof(T) == 8, double, float>::type AT;
        typedef AT acc_t __attribute__((ext_vector_type(4)));
        acc_t acc[2];
        acc[0] = (acc_t){0, 0, 0, 0};
        acc[1] = (acc_t){0, 0, 0, 0};
        const int nt = KT * KT;
        const int t0 = wid, t1 = wid + 8;                       // this wavefront's tiles (k <= 64: at most 16)
        T racc = 0;
        const int nsub = (c.cnt + 31) >> 5;
        const int tot = 32 * KQ;
        T pre[8];
        int jq[8], cq[8];                                       // (item of the sub-chunk, atom) of this thread's staged elements
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            jq[q] = (e < tot) ? e / KQ : 0;
            cq[q] = (e < tot) ? e % KQ : KQ;                    // (KQ: beyond the row - stored as zero)
        }
        auto request = [&](int sb) {                            // 32 dictionary rows -> registers (waves 0-3), clamped addresses
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int jj = sb * 32 + jq[q];
                const bool ok = jj < c.cnt && cq[q] < k;
                pre[q] = a.Dt[(int64_t)ids[ok ? jj : 0] * k + (ok ? cq[q] : 0)];
            }
        };
        auto deposit = [&](int sb, T *buf) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + 256 * q;
                if (e < tot) buf[e] = (sb * 32 + jq[q] < c.cnt && cq[q] < k) ? pre[q] : (T)0;
            }
        };
        stp[1] = a.dbg ? wall_clock64() : 0;
        if (tid < 256) request(0);
        for (int sb = 0; sb < nsub; ++sb) {
            T *buf = rowsb + (size_t)(sb & 1) * tot;
            if (tid < 256) deposit(sb, buf);
            __syncthreads();
            if (tid < 256 && sb + 1 < nsub) request(sb + 1);
            const int ncs = (c.cnt - 32 * sb < 32) ? c.cnt - 32 * sb : 32;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = u ? t1 : t0;
                if (t < nt) {                                   // (wavefront-uniform)
                    const int ta = t / KT, tb = t % KT;
                    const T *pa = buf + (lane >> 4) * KQ + 16 * ta + (lane & 15);
                    const T *pb = buf + (lane >> 4) * KQ + 16 * tb + (lane & 15);
#pragma unroll
                    for (int st = 0; st < 8; ++st) {            // (items beyond the chunk are zero rows)
                        const AT av = pa[4 * st * KQ], bv = pb[4 * st * KQ];
                        if constexpr (sizeof(T) == 8) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[u], 0, 0, 0);
                        else acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u], 0, 0, 0);
                    }
                }
            }
            if (ra >= 0 && ra < k)
                for (int j = 0; j < ncs; ++j) racc = fma(buf[j * KQ + ra], xv[32 * sb + j], racc);
        }
        stp[2] = a.dbg ? wall_clock64() : 0;
        const int rec = k * k + k;
        // element (row, column) of register r of a tile's accumulator (f64: row = (lane >> 4) + 4 r; f32: row = 4 (lane >> 4) + r)
        auto tile_rc = [&](int t, int r, int &row, int &col) {
            const int ta = t / KT, tb = t % KT;
            row = 16 * ta + (sizeof(T) == 8 ? (lane >> 4) + 4 * r : 4 * (lane >> 4) + r);
            col = 16 * tb + (lane & 15);
        };
        if (c.nch > 1) {                                        // several chunks: meet in the records, the last to arrive goes on
            T *pp = a.part + (size_t)(c.part0 + c.ci) * rec;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = u ? t1 : t0;
                if (t < nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row, col;
                        tile_rc(t, r, row, col);
                        if (row < k && col < k) rf_store(pp + row * k + col, (T)acc[u][r]);
                    }
            }
            if (ra >= 0 && ra < k) rf_store(pp + k * k + ra, racc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) flag[0] = (int)__hip_atomic_fetch_add(a.tickets + c.pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (flag[0] != c.nch - 1) return;
            // (every chunk's record has landed - written through before its ticket - so these are plain loads behind an acquire:
            //  one at a time through the atomic path they were a memory round trip each, 9-30 us for two to eight chunks)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int e = tid; e < rec; e += 512) {
                T t = 0;
                for (int z0 = 0; z0 < c.nch; z0 += 8) {
                    T v[8];
#pragma unroll
                    for (int z = 0; z < 8; ++z) v[z] = a.part[(size_t)(c.part0 + (z0 + z < c.nch ? z0 + z : c.nch - 1)) * rec + e];
#pragma unroll
                    for (int z = 0; z < 8; ++z) t += (z0 + z < c.nch) ? v[z] : (T)0;
                }
                if (e < k * k) G[(e / k) * ld + e % k] = t;
                else rhs[e - k * k] = t;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = u ? t1 : t0;
                if (t < nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row, col;
                        tile_rc(t, r, row, col);
                        if (row < k && col < k) G[row * ld + col] = (T)acc[u][r];
                    }
            }
            if (ra >= 0 && ra < k) rhs[ra] = racc;
        }
        __syncthreads();
        {
            const int nnz = a.indptr[r + 1] - a.indptr[r];
            const T ridge = (T)(a.alpha * (double)nnz / (double)a.p);   // alpha / reduction, reduction = p / |S_i| (recsys.py:175,179)
            if (tid < k) G[tid * ld + tid] += ridge;
        }
        __syncthreads();
        stp[3] = a.dbg ? wall_clock64() : 0;
        stp[4] = a.dbg ? wall_clock64() : 0;
        if (wid == 0) {                                          // one wavefront, the matrix in its registers (chol_solve_wave_reg)
            T xs;
            xs = chol_solve_wave_reg<T, KP>(G, ld, k, rhs);         // (KP = 32: k <= 32; KP = 56 / 64: 32 < k <= KP)
            if (lane < k) rf_store(a.code + r * k + lane, xs);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) flag[1] = (int)__hip_atomic_fetch_add(a.tickets + kRfMaxBatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stp[5] = a.dbg ? wall_clock64() : 0;
        me = a.n_solve - 1 - flag[1];                           // 0: the last row to finish, 1 .. : the ones just before it
        if (me >= a.nsweep) return;
        if (me > 0) {
            // a helper of the sweep: every code is ready when the last row has taken its ticket.  (It WILL: every workgroup of
            // this launch that has not run yet needs one free compute unit, and the helpers hold at most three.)
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(a.tickets + kRfMaxBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.n_solve &&
                       ++spins < (1u << 20))
                    __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
    }
    // ------------------------------------------- the last nsweep workgroups: C_, B_ and the dictionary on 512 items each
    const int b = a.b, u = a.u, W = a.nsweep;
    T *Csw = reinterpret_cast<T *>(smem_raw);                   // [k][KP] C in sweep coordinates, zero diagonal, zero padding
    T *codeb = Csw + (size_t)k * KP;                            // [b][k]
    T *cdg = codeb + (size_t)b * k;                             // [KP] reciprocals of the diagonal of C (0: frozen), sweep order
    T *cn = cdg + KP;                                           // [KP] norm budgets, sweep order
    T *bud = cn + KP;                                           // [KP] budget each atom was clipped to
    double *red = reinterpret_cast<double *>(bud + KP);         // [2][8][2]
    double *red2 = red + 32;                                    // [8][KP]
    int *ord = reinterpret_cast<int *>(red2 + 8 * KP);          // [KP]
    const bool stamp = a.dbg && me == 0 && tid == 0;
    if (stamp) {
        for (int x = 0; x < 6; ++x) a.dbg[x] = stp[x];
        a.dbg[6] = wall_clock64();
    }
    for (int e = tid; e < b * k; e += 512) codeb[e] = rf_load(a.code + a.rows[e / k] * k + e % k);
    if (tid < KP) ord[tid] = (tid < k) ? a.order[tid] : 0;
    __syncthreads();
    // C_ = (1 - w) C_ + (w / b) code_[batch]^T code_[batch] (recsys.py:159-160) - formed by every sweep workgroup for itself, in
    // sweep coordinates, from the C_ in memory, which the first of them replaces only at the very end (below)
    auto c_new = [&](int oj, int oi) {
        T dot = 0;
        for (int rr = 0; rr < b; ++rr) dot = fma(codeb[rr * k + oj], codeb[rr * k + oi], dot);
        return (T)((1.0 - a.w) * (double)a.C[(int64_t)oj * k + oi]) + (T)(a.w / (double)b) * dot;
    };
    for (int e = tid; e < k * KP; e += 512) {
        const int jj = e / KP, ii = e % KP;
        T v = 0;
        if (ii < k) {
            v = c_new(ord[jj], ord[ii]);
            if (ii == jj) { cdg[jj] = (v > (T)1e-20) ? (T)1 / v : (T)0; v = 0; }     // (the reciprocal; 0: recsys.py:201 "else do not update")
        }
        Csw[e] = v;
    }
    if (tid < KP) cn[tid] = (tid < k) ? a.comp_norm[ord[tid]] : (T)0;
    // B_: ONE ITEM PER THREAD (recsys.py:175, 182-185: the item's entries in batch order), the items of this workgroup's slice -
    // or, without an in-kernel sweep, all of them in passes.  Every load a thread needs for its item is independent of the other
    // threads': 512 items in flight per pass (a wavefront per item, as a launch of its own did it, is a chain of three
    // dependent round trips per item - 94 items per wavefront took 100 us here).
    if (stamp) a.dbg[7] = wall_clock64();
    const int per = 512 + a.cap2;                               // items of a sweep workgroup: 512 in registers + cap2 in LDS
    // (without an in-kernel sweep B_ is not updated here either: one workgroup walking every item's row takes 50 us for the
    //  thousand items of a MovieLens minibatch, the launch of its own that the host enqueues behind this one - a wavefront per
    //  item over the whole chip - 6 us)
    const int it0 = a.do_dict ? per * me : 0, it1 = a.do_dict ? ((u < it0 + per) ? u : it0 + per) : 0;
    for (int base = it0; base < it1; base += 512) {
        const int fi = base + tid;
        const bool on = fi < it1;
        const int32_t f = a.subset[on ? fi : it0];
        const int e0 = a.fptr[on ? fi : it0], e1 = on ? a.fptr[fi + 1] : e0;
        int64_t n = a.feature_n_iter[f];
        T *brow = a.Bt + (int64_t)f * k;
        T bv[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) bv[c] = brow[c < k ? c : 0];
        for (int e = e0; e < e1; ++e) {
            n += 1;                                             // recsys.py:175
            double wB = a.w_n_iter / (double)n;                 // :182-183
            wB = wB < 1.0 ? wB : 1.0;
            const double xw = (double)a.eval[e] * wB;
            const T *cr = codeb + a.esample[e] * k;
#pragma unroll
            for (int c = 0; c < KP; ++c) {
                if (c < k) {
                    T t = (T)((double)bv[c] * (1.0 - wB));      // B_[:, subset] *= 1 - w_B
                    bv[c] = (T)((double)t + (double)cr[c] * xw); // += outer(code, X_subset * w_B)
                }
            }
        }
        if (on && e1 > e0) {
#pragma unroll
            for (int c = 0; c < KP; ++c)
                if (c < k) brow[c] = bv[c];
            a.feature_n_iter[f] = n;
        }
    }
    __syncthreads();
    if (stamp) a.dbg[8] = wall_clock64();
    if (a.do_dict) {
        const int fi = it0 + tid;
        const bool live = fi < it1;
        const int64_t fo = (int64_t)a.subset[live ? fi : 0] * k;
        // the LDS tier: a SECOND item for the first cap2 threads, its atoms in LDS (D2[atom][thread]: a wavefront reads
        // consecutive words) - a minibatch of ten MovieLens rows touches 700-800 items, more than 512 threads hold in registers
        const int cap2 = a.cap2;
        T *D2 = reinterpret_cast<T *>(reinterpret_cast<char *>(ord) + ((sizeof(int) * (KP + 8) + 15) & ~(size_t)15));   // [KP][cap2]
        const int fi2 = it0 + 512 + tid;
        const bool live2 = tid < cap2 && fi2 < it1;
        const bool wave2 = (wid * 64 < cap2) && (it0 + 512 + wid * 64 < it1);     // (wave-uniform: this wave has second items)
        const int64_t fo2 = (int64_t)a.subset[live2 ? fi2 : 0] * k;
        T bn2 = 0;
        if (wave2) {                                                     // (before the first item's registers are live)
#pragma unroll 8
            for (int ii = 0; ii < KP; ++ii) {
                const T v = a.Dt[fo2 + ord[ii]];
                if (tid < cap2) D2[ii * cap2 + tid] = (live2 && ii < k) ? v : (T)0;
            }
            bn2 = a.Bt[fo2 + ord[0]];
        }
        __builtin_amdgcn_sched_barrier(0);
        T Dr[KP];
#pragma unroll
        for (int ii = 0; ii < KP; ++ii) Dr[ii] = a.Dt[fo + ord[ii]];
#pragma unroll
        for (int ii = 0; ii < KP; ++ii) Dr[ii] = (live && ii < k) ? Dr[ii] : (T)0;
        T bn = a.Bt[fo + ord[0]];
        // The atom loop is a RUNTIME loop with one shared body.  (Round 6, first version: unrolled by recursion so that the atom's
        // register was a compile-time index - 56 copies of a 300-instruction body, 134 KB of straight-line code executed once:
        // the sweep was bound by INSTRUCTION FETCH, 1.8 us per atom whatever the number of items, 89 of a minibatch's 172 us.)
        // (Measured and not kept: the row of C read with SCALAR loads from a copy in memory - no LDS traffic, but seven dependent
        //  scalar-memory round trips per wavefront and atom: 88 -> 128 us.)
        // What needs the atom's index: the dot product does not (C in sweep coordinates has a zero diagonal); the atom's old
        // value comes from memory with the B_ entry, one atom ahead; the new value goes into its register through a uniform
        // switch over groups of eight registers (rf_set: a scalar branch + eight selects).
        // The B_ entry and the old value of an atom are requested TWO atoms ahead, and the barriers inside the loop order LDS
        // traffic only (rf_lds_barrier): __syncthreads() also waits for every outstanding global load - it drained the requests
        // it was meant to overlap, a memory round trip per atom (4.5 k cycles per atom, 93 of a minibatch's 175 us, measured).
        T dn = a.Dt[fo + ord[0]];
        T dn2 = wave2 ? a.Dt[fo2 + ord[0]] : (T)0;
        T bm = a.Bt[fo + ord[k > 1 ? 1 : 0]], dm = a.Dt[fo + ord[k > 1 ? 1 : 0]];
        T bm2 = wave2 ? a.Bt[fo2 + ord[k > 1 ? 1 : 0]] : (T)0, dm2 = wave2 ? a.Dt[fo2 + ord[k > 1 ? 1 : 0]] : (T)0;
        for (int jj = 0; jj < k; ++jj) {
            {
                const int jn = (jj + 2 < k) ? jj + 2 : k - 1;
                const T bvj = live ? bn : (T)0;
                const T dold = live ? dn : (T)0;
                bn = bm; dn = dm;
                bm = a.Bt[fo + ord[jn]];
                dm = a.Dt[fo + ord[jn]];
                // the row of C as explicit 16-byte LDS reads (KP / VW of them: the LDS pipe is what bounds this loop - eight
                // wavefronts broadcast-read every row)
                constexpr int VW = 16 / (int)sizeof(T);
                typedef T rf_vec __attribute__((ext_vector_type(VW)));
                const rf_vec *crow = reinterpret_cast<const rf_vec *>(Csw + jj * KP);
                T dot = 0, dot2 = 0;
                const int t2 = tid < cap2 ? tid : 0;
                if (!wave2) {
#pragma unroll
                    for (int iv = 0; iv < KP / VW; ++iv) {
                        const rf_vec c = crow[iv];
#pragma unroll
                        for (int x = 0; x < VW; ++x) dot = fma(c[x], Dr[iv * VW + x], dot);
                    }
                } else {                                         // (both items of the thread on one read of the row of C)
#pragma unroll
                    for (int iv = 0; iv < KP / VW; ++iv) {
                        const rf_vec c = crow[iv];
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            dot = fma(c[x], Dr[iv * VW + x], dot);
                            dot2 = fma(c[x], D2[(iv * VW + x) * cap2 + t2], dot2);
                        }
                        if ((iv & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (a few LDS reads in flight, not all KP: registers)
                    }
                }
                const T icd = cdg[jj];                                   // 1 / C[j][j], or 0: "else do not update"
                T un = (icd != (T)0) ? (bvj - dot) * icd : dold;       // recsys.py:201-203
                un = live ? un : (T)0;
                double o2 = (double)dold * (double)dold, n2 = (double)un * (double)un;
                T un2 = 0;
                if (wave2) {
                    const T bv2 = live2 ? bn2 : (T)0;
                    const T dold2 = live2 ? dn2 : (T)0;
                    bn2 = bm2; dn2 = dm2;
                    bm2 = a.Bt[fo2 + ord[jn]];
                    dm2 = a.Dt[fo2 + ord[jn]];
                    un2 = (icd != (T)0) ? (bv2 - dot2) * icd : dold2;
                    un2 = live2 ? un2 : (T)0;
                    o2 += (double)dold2 * (double)dold2;
                    n2 += (double)un2 * (double)un2;
                }
                o2 = wave_sum(o2);
                n2 = wave_sum(n2);
                double *rd = red + (jj & 1) * 16;
                if (lane == 0) { rd[2 * wid] = o2; rd[2 * wid + 1] = n2; }
                rf_lds_barrier();
                double so = 0, sn = 0;
#pragma unroll
                for (int x = 0; x < 8; ++x) { so += rd[2 * x]; sn += rd[2 * x + 1]; }
                if (W > 1) {
                    // the other sweep workgroups' sums of this atom: written through, read past the caches, the data is its own
                    // flag (a NaN no sum can be: the first workgroup restores it at the end of the launch) - one memory round trip
                    double *slot = a.xch + (size_t)jj * 2 * kRfMaxSweep;
                    if (tid < 2) rf_store(slot + 2 * me + tid, tid == 0 ? so : sn);
                    double got = 0.0;
                    if (tid < 2 * W) {
                        unsigned spins = 0;
                        do got = rf_load(slot + tid);
                        while (__double_as_longlong(got) == kRfSentinel && ++spins < (1u << 18));
                    }
                    // (red2 is free until the end of the sweep; it is rewritten behind the NEXT atom's barrier, which every
                    //  thread reaches only after it has read these)
                    if (tid < 2 * kRfMaxSweep) red2[tid] = (tid < 2 * W) ? got : 0.0;
                    rf_lds_barrier();
                    so = (red2[0] + red2[2]) + (red2[4] + red2[6]);
                    sn = (red2[1] + red2[3]) + (red2[5] + red2[7]);
                }
                const T budget = cn[jj] + (T)so;                         // :197-198 comp_norm_ += subset_norm
                // :205-208 norm = sqrt(sum), lim = sqrt(budget), "if norm > lim: atom /= norm / lim" as ONE factor
                // sqrt(budget / sum) (a negative or NaN budget never clips, as there)
                const bool clip = (T)sn > budget && budget >= (T)0;
                const T scale = clip ? (T)sqrt((double)budget / sn) : (T)1;
                un *= scale;
                rf_set<T, KP>(Dr, jj, un);
                if (wave2) {
                    un2 *= scale;
                    if (tid < cap2) D2[jj * cap2 + tid] = un2;
                }
                if (tid == 0) bud[jj] = budget;
            }
        }
        // the projected atoms' norms leave the budgets (:211-212), the rows go back
        __syncthreads();                                                 // (red2 held the last atom's exchanged sums)
        if (stamp) a.dbg[9] = wall_clock64();
#pragma unroll
        for (int ii = 0; ii < KP; ++ii) {
            double q2 = (double)Dr[ii] * (double)Dr[ii];
            if (wave2) {
                const T d2 = D2[ii * cap2 + (tid < cap2 ? tid : 0)];
                q2 += live2 ? (double)d2 * (double)d2 : 0.0;
                if (live2 && ii < k) a.Dt[fo2 + ord[ii]] = d2;
            }
            const double s2 = wave_sum(q2);
            if (lane == 0) red2[wid * KP + ii] = s2;
        }
        if (live)
#pragma unroll
            for (int ii = 0; ii < KP; ++ii)
                if (ii < k) a.Dt[fo + ord[ii]] = Dr[ii];
        __syncthreads();
        double s2 = 0;
        if (tid < KP) {
#pragma unroll
            for (int x = 0; x < 8; ++x) s2 += red2[x * KP + tid];
        }
        if (W > 1) {
            double *slot = a.xch + (size_t)KP * 2 * kRfMaxSweep;           // [kRfMaxSweep][KP] behind the atoms' slots
            if (me > 0) {
                if (tid < k) rf_store(slot + (size_t)me * KP + tid, s2);
                return;                                                  // (a helper is done)
            }
            for (int h = 1; h < W; ++h) {
                double got = 0.0;
                if (tid < k) {
                    unsigned spins = 0;
                    do got = rf_load(slot + (size_t)h * KP + tid);
                    while (__double_as_longlong(got) == kRfSentinel && ++spins < (1u << 18));
                }
                s2 += got;
            }
        }
        if (tid < k) a.comp_norm[ord[tid]] = bud[tid] - (T)s2;
        if (W > 1) {                                                     // the slots back to sentinels: every helper has left them
            __syncthreads();                                             // (... once wave 0 has READ the helpers' last ones)
            long long *xs = reinterpret_cast<long long *>(a.xch);
            for (int e2 = tid; e2 < (KP * 2 + KP) * kRfMaxSweep; e2 += 512) xs[e2] = kRfSentinel;
        }
    }
    // (only the first sweep workgroup is left) C_ in memory, natural order: the same expression, the same bits as Csw
    __syncthreads();
    for (int e = tid; e < k * k; e += 512) {
        a.part[e] = c_new(e / k, e % k);                                 // (staged: every C_[oj][oi] is still read by other threads' c_new)
    }
    __syncthreads();
    for (int e = tid; e < k * k; e += 512) a.C[e] = a.part[e];
    if (stamp) { a.dbg[10] = wall_clock64(); a.dbg[11] = (unsigned long long)a.u; a.dbg[12] = (unsigned long long)a.nsweep; }
    if (tid <= kRfMaxBatch) a.tickets[tid] = 0;                 // for the next launch
}

template <typename T>
int recsys_codes(const T *Dt, int64_t p, int k, const int32_t *indptr, const int32_t *indices, const T *data,
                 const int64_t *row_ids, const int64_t *code_rows, int64_t b, double alpha, T *code, hipStream_t st) {
    if (b <= 0) return MODL_OK;
    const int rpl = k <= 64 ? 1 : (k <= 128 ? 2 : 0);
    const size_t extra = rpl ? (size_t)((k + 3) & ~3) + 3 * (size_t)rpl * 4 * 64 : 0;     // reciprocal pivots + partial sums
    size_t lds = sizeof(T) * ((size_t)k * (rpl ? (k | 1) : k) + k + 32 * (size_t)k + 32 + extra) + 32 * sizeof(int) + 16;
    int use = rpl;
    if (lds > 160 * 1024 && rpl) {                           // (k near 128 in f64: the padded matrix no longer fits)
        use = 0;
        lds = sizeof(T) * ((size_t)k * k + k + 32 * (size_t)k + 32) + 32 * sizeof(int) + 16;
    }
    if (lds > 160 * 1024) return MODL_EINVAL;
#define MODL_RECSYS_CODE(RPL)                                                                                              \
    do {                                                                                                                   \
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&recsys_code_kernel<T, RPL>),                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                             \
        hipLaunchKernelGGL((recsys_code_kernel<T, RPL>), dim3((unsigned)b), dim3(256), lds, st, Dt, p, k, indptr, indices,  \
                           data, row_ids, code_rows, alpha, code);                                                         \
    } while (0)
    if (use == 1) MODL_RECSYS_CODE(1);
    else if (use == 2) MODL_RECSYS_CODE(2);
    else MODL_RECSYS_CODE(0);
#undef MODL_RECSYS_CODE
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

template <typename T> struct EpiAxpbyC {
    T *out; int64_t ld; T alpha, beta;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const {
        T *o = out + m * ld + n;
        *o = (*o) * beta + alpha * v;
    }
};


// ---- one call per minibatch ---------------------------------------------------------------------------------------
// The host side of recsys.py:147-165 for a batch of CSR rows: the batch's ratings grouped by item (counting sort, batch
// order kept inside an item: two rows rating the same item do not commute in the B_ update), written with the row ids
// and the atom order into a pinned slot, moved to HBM by a kernel reading the device-mapped slot (no copy-engine hop
// between kernels), then the four launches of the step.  Nothing is synchronised: a ring of slots + events.
constexpr int kRecsysSlots = 8;
std::atomic<int> g_recsys_fused{1};        // modl_debug_set(MODL_DEBUG_RECSYS_FUSED, ...): 0 = the separate launches of rounds 2-5

}  // namespace modl

struct modl_recsys_plan {
    int dtype = 0, k = 0;
    int64_t p = 0, max_entries = 0, max_batch = 0;
    char *h[modl::kRecsysSlots] = {nullptr};
    char *hdev[modl::kRecsysSlots] = {nullptr};
    int slot = 0;
    char *dstage = nullptr;                    // two staging buffers (dstage2[cur]: the minibatch being launched)
    char *dstage2[2] = {nullptr, nullptr};
    int cur = 0;
    size_t stage_bytes = 0;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<int32_t> cnt, touched;
    // the fused minibatch kernel (k <= 64, batches of at most 64 rows): chunk records, tickets; the slots are protected by an
    // acknowledgement word the staging kernel writes into the slot itself (its use count; no stream event)
    char *part = nullptr;
    unsigned int *tickets = nullptr;
    double *xch = nullptr;                     // exchange slots of the sweep workgroups (sentinels between launches)
    unsigned long long *dbg = nullptr;         // modl_recsys_plan_stamps(plan, 1, ...): stamps of the last workgroup of each launch
    unsigned long long uses[modl::kRecsysSlots] = {0};
    size_t ack_off = 0;
    long fused_calls = 0, split_calls = 0;     // minibatches through the one-launch path / through the separate launches
    double wait_ms = 0;                        // host time spent waiting for a staging slot (the host is eight minibatches ahead)
    // raised (pinned, system scope) by a dictionary-update launch whose cross-workgroup wait gave up (bcd_few_kernel): the update is
    // incomplete, every further minibatch of the plan is refused with MODL_ETIMEOUT until modl_recsys_plan_status has reported it
    unsigned int *pflags = nullptr, *pflags_dev = nullptr;
};

namespace modl {

struct RecsysLayout { size_t rows, order, subset, fptr, esample, eval, total; };
static RecsysLayout recsys_layout(size_t tsz, int64_t b, int k, int64_t u, int64_t m) {
    RecsysLayout L;
    size_t o = 0;
    L.rows = o; o = align_up(o + sizeof(int64_t) * (size_t)b, 16);
    L.order = o; o = align_up(o + sizeof(int32_t) * (size_t)k, 16);
    L.subset = o; o = align_up(o + sizeof(int32_t) * (size_t)u, 16);
    L.fptr = o; o = align_up(o + sizeof(int32_t) * (size_t)(u + 1), 16);
    L.esample = o; o = align_up(o + sizeof(int32_t) * (size_t)m, 16);
    L.eval = o; o = align_up(o + tsz * (size_t)m, 16);
    L.total = o;
    return L;
}

// ONE workgroup: the used part of the pinned slot -> HBM (eight 16-byte loads per thread in flight), then the slot's use count
// into its acknowledgement word (host memory): the host checks that word before it refills the slot - no stream event (a
// hipEventRecord between two kernels is a ~5 us bubble on this part)
__global__ __launch_bounds__(256) void recsys_stage_kernel(const rf_u4 *__restrict__ src, rf_u4 *__restrict__ dst, size_t n16,
                                                           unsigned long long *ack, unsigned long long use) {
    for (size_t i0 = threadIdx.x; i0 < n16; i0 += 8 * 256) {
        rf_u4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(i0 + 256 * q < n16) ? i0 + 256 * q : n16 - 1];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i0 + 256 * q < n16) dst[i0 + 256 * q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ack, use, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// What the host prepares for one minibatch: its pinned slot filled (row ids, atom order, the batch's ratings grouped by item).
struct RecsysPrep { int slot = -1; int64_t b = 0, u = 0, m = 0; RecsysLayout L{}; unsigned long long use = 0; };

template <typename T>
int recsys_prepare(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices, const T *h_data, int64_t n_rows,
                   const int64_t *h_rows, int64_t b, const int64_t *h_order, hipStream_t st, RecsysPrep &out) {
    const int k = pl->k;
    if (b <= 0 || b > pl->max_batch) return MODL_EINVAL;
    if (pl->pflags && *reinterpret_cast<volatile unsigned int *>(pl->pflags) != 0) return MODL_ETIMEOUT;
    int64_t m = 0;
    for (int64_t i = 0; i < b; ++i) {
        const int64_t r = h_rows[i];
        if (r < 0 || r >= n_rows) return MODL_EINVAL;
        m += h_indptr[r + 1] - h_indptr[r];
    }
    if (m > pl->max_entries) return MODL_ENOMEM;
    for (int i = 0; i < k; ++i)
        if (h_order[i] < 0 || h_order[i] >= k) return MODL_EINVAL;
    // items touched by the batch, ascending; entries per item
    std::vector<int32_t> &cnt = pl->cnt, &touched = pl->touched;
    touched.clear();
    for (int64_t i = 0; i < b; ++i)
        for (int32_t e = h_indptr[h_rows[i]]; e < h_indptr[h_rows[i] + 1]; ++e) {
            const int32_t c = h_indices[e];
            if (c < 0 || c >= pl->p) {
                for (int32_t t : touched) cnt[t] = 0;
                return MODL_EINVAL;
            }
            if (cnt[c]++ == 0) touched.push_back(c);
        }
    std::sort(touched.begin(), touched.end());
    const int64_t u = (int64_t)touched.size();
    const RecsysLayout L = recsys_layout(sizeof(T), b, k, u, m);
    if (L.total > pl->stage_bytes) {
        for (int32_t t : touched) cnt[t] = 0;
        return MODL_ENOMEM;
    }
    const int slot = pl->slot;
    pl->slot = (slot + 1) % kRecsysSlots;
    char *h = pl->h[slot];
    if (pl->uses[slot]) {                                    // the copy of the slot's last use has read it?
        // (the host is eight minibatches ahead: the device is the bottleneck then, and this wait is where the host idles)
        volatile unsigned long long *ack = reinterpret_cast<volatile unsigned long long *>(h + pl->ack_off);
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0; *ack < pl->uses[slot]; ++spins)
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
                MODL_HIP(hipStreamSynchronize(st));           // (another stream than the slot's last use, or a stuck device: settle it)
                break;
            }
        pl->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    std::memcpy(h + L.rows, h_rows, sizeof(int64_t) * (size_t)b);
    int32_t *ho = reinterpret_cast<int32_t *>(h + L.order);
    for (int i = 0; i < k; ++i) ho[i] = (int32_t)h_order[i];
    int32_t *hs = reinterpret_cast<int32_t *>(h + L.subset), *hf = reinterpret_cast<int32_t *>(h + L.fptr);
    int32_t *hes = reinterpret_cast<int32_t *>(h + L.esample);
    T *hev = reinterpret_cast<T *>(h + L.eval);
    hf[0] = 0;
    for (int64_t i = 0; i < u; ++i) {
        hs[i] = touched[(size_t)i];
        hf[i + 1] = hf[i] + cnt[touched[(size_t)i]];
        cnt[touched[(size_t)i]] = hf[i];                    // becomes the write cursor of the item
    }
    for (int64_t i = 0; i < b; ++i)
        for (int32_t e = h_indptr[h_rows[i]]; e < h_indptr[h_rows[i] + 1]; ++e) {
            const int32_t at = cnt[h_indices[e]]++;
            hes[at] = (int32_t)i;
            hev[at] = h_data[e];
        }
    for (int32_t t : touched) cnt[t] = 0;
    out.slot = slot; out.b = b; out.u = u; out.m = m; out.L = L; out.use = ++pl->uses[slot];
    return MODL_OK;
}

// The launches of a prepared minibatch.  staged: its arrays are already in the current device staging buffer (the previous
// minibatch's launch carried them); next: the minibatch after this one, prepared - its arrays ride on this one's launch
// (*next_staged says whether they did).
template <typename T>
int recsys_launch(modl_recsys_plan *pl, const RecsysPrep &pr, bool staged, const RecsysPrep *next, bool *next_staged,
                  const int32_t *h_indptr, const int32_t *d_indptr, const int32_t *d_indices, const T *d_data,
                  const int64_t *h_rows, const int64_t *h_order, double alpha, double w, double n_iter,
                  T *Dt, T *Bt, T *C, T *code, T *comp_norm, int64_t *feature_n_iter, hipStream_t st) {
    const int k = pl->k;
    const int64_t b = pr.b, u = pr.u;
    const RecsysLayout &L = pr.L;
    if (next_staged) *next_staged = false;
    char *dst = pl->dstage2[pl->cur];
    if (!staged) {
        hipLaunchKernelGGL(recsys_stage_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const rf_u4 *>(pl->hdev[pr.slot]),
                           reinterpret_cast<rf_u4 *>(dst), L.total / 16,
                           reinterpret_cast<unsigned long long *>(pl->hdev[pr.slot] + pl->ack_off), pr.use);
        MODL_LAUNCH_CHECK();
    }
    const int64_t *d_rows = reinterpret_cast<const int64_t *>(dst + L.rows);
    const int32_t *d_order = reinterpret_cast<const int32_t *>(dst + L.order);
    const int32_t *d_subset = reinterpret_cast<const int32_t *>(dst + L.subset);
    const int32_t *d_fptr = reinterpret_cast<const int32_t *>(dst + L.fptr);
    const int32_t *d_es = reinterpret_cast<const int32_t *>(dst + L.esample);
    const T *d_ev = reinterpret_cast<const T *>(dst + L.eval);
    // ---- ONE launch for the minibatch (recsys_fused_kernel) when it applies: at most 64 atoms, 64 rows, 64 chunks of 128 ratings
    if (k <= kRfWide<T>::value && b <= kRfMaxBatch && pl->part && g_recsys_fused.load(std::memory_order_relaxed)) {
        RecsysFusedArgs<T> fa;
        int nc = 0, n_solve = 0;
        bool fits = true;
        for (int64_t i = 0; i < b && fits; ++i) {
            const int32_t beg = h_indptr[h_rows[i]], nnz = h_indptr[h_rows[i] + 1] - beg;
            fa.rows[i] = h_rows[i];
            if (nnz == 0) continue;
            ++n_solve;
            const int nch = (nnz + kRfChunk - 1) / kRfChunk;
            if (nc + nch > kRfMaxChunks) { fits = false; break; }
            for (int ci = 0; ci < nch; ++ci) {
                RecsysChunk &c = fa.ch[nc + ci];
                c.pos = (int32_t)i; c.beg = beg + ci * kRfChunk; c.cnt = std::min(kRfChunk, nnz - ci * kRfChunk);
                c.nch = nch; c.ci = ci; c.part0 = nc;
            }
            nc += nch;
        }
        if (fits) {
            fa.indptr = d_indptr; fa.indices = d_indices; fa.data = d_data;
            fa.Dt = Dt; fa.Bt = Bt; fa.C = C; fa.code = code; fa.comp_norm = comp_norm; fa.feature_n_iter = feature_n_iter;
            fa.order = d_order; fa.subset = d_subset; fa.fptr = d_fptr; fa.esample = d_es; fa.eval = d_ev;
            fa.part = reinterpret_cast<T *>(pl->part); fa.tickets = pl->tickets;
            fa.alpha = alpha; fa.w = w; fa.w_n_iter = w * n_iter; fa.p = pl->p;
            fa.k = k; fa.b = (int)b; fa.u = (int)u; fa.n_solve = n_solve;
            constexpr int KPW0 = kRfWide<T>::value;
            const int KP0 = k <= 32 ? 32 : KPW0;
            {   // the LDS tier: what 160 KB leave next to the last phase's other arrays, in whole wavefronts
                const size_t base = recsys_fused_lds<T>(k, (int)b, KP0, 0);
                const size_t room = base < 156 * 1024 ? 156 * 1024 - base : 0;
                int cap = (int)(room / (sizeof(T) * KP0)) & ~63;
                if (cap > 512) cap = 512;
                if (g_recsys_fused.load(std::memory_order_relaxed) == 3) cap = 0;      // (registers only)
                fa.cap2 = cap;
            }
            const int per = 512 + fa.cap2;
            fa.nsweep = (int)std::max<int64_t>(1, (u + per - 1) / per);
            fa.do_dict = (u > 0 && fa.nsweep <= kRfMaxSweep && fa.nsweep <= std::max(n_solve, 1)) ? 1 : 0;
            // Where B_ and the dictionary update run.  MEASURED (scripts/ab_recsys_fused.sh, MovieLens-10M-shaped rows, k = 50,
            // b = 10, f64; profiles/r06_ab_recsys_fused.txt): as launches of their own behind this one - a wavefront per item, the
            // blocked update over the whole chip - 197 us per minibatch at 109 ratings per row, 169 at 36, 130 at 13; inside this
            // kernel on ONE workgroup (value 2; 3 without the LDS tier) 251-333 / 203 / 172 us - one compute unit walks a thousand
            // item rows and broadcasts every row of C to eight wavefronts (1.5 us per atom, bound by the LDS pipe) while 255 units
            // idle; on up to four workgroups that exchange every atom's sums through memory (value 4) 345 us.  So the default (1)
            // keeps this launch to the codes and C_; the in-kernel variants stay selectable (and tested: they are correct).
            const int mode = g_recsys_fused.load(std::memory_order_relaxed);
            if (mode == 1 || (fa.nsweep > 1 && mode != 4)) fa.do_dict = 0;
            if (!fa.do_dict) fa.nsweep = 1;
            fa.xch = pl->xch;
            fa.dbg = pl->dbg;
            fa.stage_src = nullptr; fa.stage_dst = nullptr; fa.stage_n16 = 0; fa.stage_ack = nullptr; fa.stage_use = 0;
            if (next) {                                      // the next minibatch's arrays ride along, into the other buffer
                fa.stage_src = reinterpret_cast<const rf_u4 *>(pl->hdev[next->slot]);
                fa.stage_dst = reinterpret_cast<rf_u4 *>(pl->dstage2[pl->cur ^ 1]);
                fa.stage_n16 = next->L.total / 16;
                fa.stage_ack = reinterpret_cast<unsigned long long *>(pl->hdev[next->slot] + pl->ack_off);
                fa.stage_use = next->use;
            }
            constexpr int KPW = kRfWide<T>::value;             // registers per item of the wide variant: 64 (f32) / 56 (f64)
            const int KP = k <= 32 ? 32 : KPW;
            const size_t lds = recsys_fused_lds<T>(k, (int)b, KP, fa.cap2);
            const unsigned grid = (unsigned)(nc > 0 ? nc : 1) + (fa.stage_n16 ? 1u : 0u);
            if (KP == 32) {
                MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&recsys_fused_kernel<T, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((recsys_fused_kernel<T, 32>), dim3(grid), dim3(512), lds, st, fa);
            } else {
                MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&recsys_fused_kernel<T, KPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((recsys_fused_kernel<T, KPW>), dim3(grid), dim3(512), lds, st, fa);
            }
            MODL_LAUNCH_CHECK();
            ++pl->fused_calls;
            if (fa.stage_n16) {
                pl->cur ^= 1;
                if (next_staged) *next_staged = true;
            }
            if (u > 0 && !fa.do_dict) {                      // more touched items than the sweep's workgroups hold: B_ and the
                                                             // blocked dictionary update as launches of their own
                hipLaunchKernelGGL((recsys_update_B_kernel<T>), dim3((unsigned)cdiv(u, 4)), dim3(256), 0, st, Bt, k, feature_n_iter,
                                   d_subset, d_fptr, d_es, d_ev, (const T *)code, d_rows, w * n_iter, u);
                MODL_LAUNCH_CHECK();
                DictUpdateArgs<T> a;
                a.Dt = Dt; a.Bt = Bt; a.C = C; a.comp_norm = comp_norm; a.subset = d_subset; a.order = d_order;
                a.h_order = h_order; a.s = u; a.k = k; a.optimizer = 0; a.comp_pos = 0;
                a.comp_l1_ratio = 0.0; a.w = w; a.step_size = 1.0; a.ws = pl->ws; a.ws_bytes = pl->ws_bytes;
                a.persist_flags = pl->pflags_dev;
                int nl = 0;
                MODL_TRY(dict_update<T>(st, a, &nl));
            }
            return MODL_OK;
        }
    }
    ++pl->split_calls;
    // codes of the batch's rows (recsys.py:176-181), written to code_[rows]
    MODL_TRY(recsys_codes<T>(Dt, pl->p, k, d_indptr, d_indices, d_data, d_rows, nullptr, b, alpha, code, st));
    if (u > 0) {                                             // :175, :182-185
        hipLaunchKernelGGL((recsys_update_B_kernel<T>), dim3((unsigned)cdiv(u, 4)), dim3(256), 0, st, Bt, k, feature_n_iter,
                           d_subset, d_fptr, d_es, d_ev, (const T *)code, d_rows, w * n_iter, u);
        MODL_LAUNCH_CHECK();
    }
    {                                                        // :159-160  C_ = (1 - w) C_ + (w / b) code_[batch]^T code_[batch]
        Operand A;
        A.ptr = code; A.si = 1; A.sk = k; A.gk = gather64(d_rows);
        EpiAxpbyC<T> epi{C, k, (T)(w / (double)b), (T)(1.0 - w)};
        SplitWs none;
        MODL_TRY((launch_gemm<T, EpiAxpbyC<T>>(st, A, A, k, k, b, epi, none, nullptr, 512, 1)));
    }
    if (u > 0) {                                             // :187-213
        DictUpdateArgs<T> a;
        a.Dt = Dt; a.Bt = Bt; a.C = C; a.comp_norm = comp_norm; a.subset = d_subset; a.order = d_order;
        a.h_order = h_order; a.s = u; a.k = k; a.optimizer = 0; a.comp_pos = 0;
        a.comp_l1_ratio = 0.0; a.w = w; a.step_size = 1.0; a.ws = pl->ws; a.ws_bytes = pl->ws_bytes;
        a.persist_flags = pl->pflags_dev;
        int nl = 0;
        MODL_TRY(dict_update<T>(st, a, &nl));
    }
    return MODL_OK;
}

template <typename T>
int recsys_minibatch(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices, const T *h_data,
                     int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const T *d_data,
                     const int64_t *h_rows, int64_t b, const int64_t *h_order, double alpha, double w, double n_iter,
                     T *Dt, T *Bt, T *C, T *code, T *comp_norm, int64_t *feature_n_iter, hipStream_t st) {
    RecsysPrep pr;
    MODL_TRY(recsys_prepare<T>(pl, h_indptr, h_indices, h_data, n_rows, h_rows, b, h_order, st, pr));
    return recsys_launch<T>(pl, pr, false, nullptr, nullptr, h_indptr, d_indptr, d_indices, d_data, h_rows, h_order, alpha, w, n_iter,
                            Dt, Bt, C, code, comp_norm, feature_n_iter, st);
}

// A run of minibatches (modl_recsys_fit_batches_*): minibatch t + 1 is drawn and prepared BEFORE minibatch t is launched, and its
// staged arrays ride on t's launch - one launch per minibatch.  Same draws in the same order as one minibatch at a time.
template <typename T>
int recsys_fit_batches(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices, const T *h_data, int64_t n_rows,
                       const int32_t *d_indptr, const int32_t *d_indices, const T *d_data, const int64_t *h_rows,
                       int64_t n_rows_fit, int64_t batch_size, modl_rk *order_rng, double alpha, double learning_rate,
                       int64_t *n_iter, T *Dt, T *Bt, T *C, T *code, T *comp_norm, int64_t *feature_n_iter, hipStream_t st,
                       int64_t *n_done) {
    struct Drawn { RecsysPrep pr; std::vector<int64_t> order; double w = 0; int64_t bb = 0, r0 = 0, n_after = 0;
                   uint32_t key[624]; int32_t pos = 0; int rc = MODL_OK; };
    Drawn q[2];
    q[0].order.resize((size_t)pl->k);
    q[1].order.resize((size_t)pl->k);
    int64_t n_virtual = *n_iter;                             // n_iter_ behind the last DRAWN minibatch
    auto draw = [&](Drawn &d, int64_t r0) {                  // weights, order, the slot: everything but the launch
        d.r0 = r0;
        d.bb = std::min<int64_t>(batch_size, n_rows_fit - r0);
        d.rc = modl_rk_get_mt_state(order_rng, d.key, &d.pos);
        if (d.rc == MODL_OK) d.rc = modl_batch_weight(n_virtual + d.bb, d.bb, learning_rate, 0.0, &d.w);   // recsys.py:153-154
        if (d.rc == MODL_OK) d.rc = modl_rk_permutation(order_rng, pl->k, d.order.data());                 // :196
        if (d.rc == MODL_OK)
            d.rc = recsys_prepare<T>(pl, h_indptr, h_indices, h_data, n_rows, h_rows + r0, d.bb, d.order.data(), st, d.pr);
        if (d.rc != MODL_OK) (void)modl_rk_set_mt_state(order_rng, d.key, d.pos);     // (a minibatch that fails drew nothing)
        else { n_virtual += d.bb; d.n_after = n_virtual; }
    };
    int64_t done = 0;
    if (n_done) *n_done = 0;
    if (n_rows_fit <= 0) return MODL_OK;
    int cur = 0;
    draw(q[0], 0);
    if (q[0].rc != MODL_OK) return q[0].rc;
    bool staged = false;
    for (;;) {
        Drawn &d = q[cur];
        const int64_t r1 = d.r0 + d.bb;
        Drawn *nx = nullptr;
        if (r1 < n_rows_fit) {
            nx = &q[cur ^ 1];
            draw(*nx, r1);
        }
        bool next_staged = false;
        const int rc = recsys_launch<T>(pl, d.pr, staged, (nx && nx->rc == MODL_OK) ? &nx->pr : nullptr, &next_staged, h_indptr,
                                        d_indptr, d_indices, d_data, h_rows + d.r0, d.order.data(), alpha, d.w, (double)d.n_after,
                                        Dt, Bt, C, code, comp_norm, feature_n_iter, st);
        if (rc != MODL_OK) {                                 // leave the generator behind the last ENQUEUED minibatch
            (void)modl_rk_set_mt_state(order_rng, d.key, d.pos);
            return rc;
        }
        *n_iter = d.n_after;
        ++done;
        if (n_done) *n_done = done;
        if (!nx) break;
        if (nx->rc != MODL_OK) return nx->rc;                // (its draws are rewound already)
        staged = next_staged;
        cur ^= 1;
    }
    return MODL_OK;
}

}  // namespace modl

using namespace modl;

extern "C" {

#define ABI_RECSYS(SFX, T)                                                                                          \
    int modl_recsys_codes_##SFX(const T *d_Dt, int64_t p, int k, const int32_t *d_indptr, const int32_t *d_indices,   \
                                const T *d_data, const int64_t *d_row_ids, const int64_t *d_code_rows, int64_t b,     \
                                double alpha, T *d_code, void *stream) {                                            \
        if (!d_Dt || !d_indptr || !d_indices || !d_data || !d_code || k <= 0 || p <= 0 || b < 0) return MODL_EINVAL;  \
        return recsys_codes<T>(d_Dt, p, k, d_indptr, d_indices, d_data, d_row_ids, d_code_rows, b, alpha, d_code,     \
                               (hipStream_t)stream);                                                                \
    }                                                                                                               \
    int modl_recsys_update_B_##SFX(T *d_Bt, int k, int64_t *d_feature_n_iter, const int32_t *d_subset,                \
                                   const int32_t *d_fptr, const int32_t *d_entry_sample, const T *d_entry_val,        \
                                   const T *d_code_b, double w_times_n_iter, int64_t u, void *stream) {             \
        if (!d_Bt || !d_feature_n_iter || !d_subset || !d_fptr || !d_entry_sample || !d_entry_val || !d_code_b ||     \
            k <= 0 || u < 0)                                                                                        \
            return MODL_EINVAL;                                                                                     \
        if (u == 0) return MODL_OK;                                                                                 \
        hipLaunchKernelGGL((recsys_update_B_kernel<T>), dim3((unsigned)cdiv(u, 4)), dim3(256), 0, (hipStream_t)stream, \
                           d_Bt, k, d_feature_n_iter, d_subset, d_fptr, d_entry_sample, d_entry_val, d_code_b,        \
                           (const int64_t *)nullptr, w_times_n_iter, u);                                                                      \
        MODL_LAUNCH_CHECK();                                                                                        \
        return MODL_OK;                                                                                             \
    }                                                                                                               \
    int modl_recsys_predict_##SFX(double *d_out, const int32_t *d_indices, const int32_t *d_indptr, const T *d_code,  \
                                  int64_t n_rows, int k, const T *d_Dt, void *stream) {                             \
        if (!d_out || !d_indices || !d_indptr || !d_code || !d_Dt || n_rows < 0 || k <= 0) return MODL_EINVAL;        \
        if (n_rows == 0) return MODL_OK;                                                                            \
        hipLaunchKernelGGL((recsys_predict_kernel<T>), dim3((unsigned)cdiv(n_rows, 4)), dim3(256), 0,                 \
                           (hipStream_t)stream, d_out, d_indices, d_indptr, d_code, n_rows, k, d_Dt);                \
        MODL_LAUNCH_CHECK();                                                                                        \
        return MODL_OK;                                                                                             \
    }                                                                                                               \
    /* C = beta C + alpha rows^T rows  (recsys.py:159-160 with beta = 1 - w, alpha = w / b) */                      \
    int modl_gram_axpby_##SFX(const T *d_rows, int64_t b, int k, T *d_C, T beta, T alpha, void *stream) {            \
        if (!d_rows || !d_C || b < 0 || k <= 0) return MODL_EINVAL;                                                  \
        Operand A;                                                                                                  \
        A.ptr = d_rows; A.si = 1; A.sk = k;                                                                         \
        EpiAxpbyC<T> epi{d_C, k, alpha, beta};                                                                      \
        SplitWs none;                                                                                               \
        return launch_gemm<T, EpiAxpbyC<T>>((hipStream_t)stream, A, A, k, k, b, epi, none, nullptr, 512, 1);         \
    }                                                                                                               \
    /* _update_dict on device-resident state (dict_fact.py:650-715 / recsys.py:187-213) */                         \
    int modl_dict_update_##SFX(T *d_Dt, const T *d_Bt, const T *d_C, T *d_comp_norm, const int32_t *d_subset,         \
                               int64_t s, const int32_t *d_order, const int64_t *h_order, int k, int optimizer,       \
                               int comp_pos, double comp_l1_ratio, double w, double step_size, void *d_ws,          \
                               size_t ws_bytes, void *stream) {                                                     \
        if (!d_Dt || !d_Bt || !d_C || !d_comp_norm || !d_order || !h_order || s < 0 || k <= 0) return MODL_EINVAL;    \
        DictUpdateArgs<T> a;                                                                                        \
        a.Dt = d_Dt; a.Bt = d_Bt; a.C = d_C; a.comp_norm = d_comp_norm; a.subset = d_subset; a.order = d_order;       \
        a.h_order = h_order; a.s = s; a.k = k; a.optimizer = optimizer; a.comp_pos = comp_pos;                        \
        a.comp_l1_ratio = comp_l1_ratio; a.w = w; a.step_size = step_size; a.ws = d_ws; a.ws_bytes = ws_bytes;        \
        int nl = 0;                                                                                                 \
        return dict_update<T>((hipStream_t)stream, a, &nl);                                                         \
    }
ABI_RECSYS(f32, float)
ABI_RECSYS(f64, double)
#undef ABI_RECSYS

size_t modl_dict_update_workspace(int dtype, int64_t s_max, int k) { return dict_update_workspace(dtype, s_max, k); }

int modl_recsys_plan_create(int dtype, int64_t p, int k, int64_t max_batch, int64_t max_entries, modl_recsys_plan **out) {
    if (!out || (dtype != MODL_F32 && dtype != MODL_F64) || p <= 0 || k <= 0 || max_batch <= 0 || max_entries < 0)
        return MODL_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MODL_ENOGPU;
    modl_recsys_plan *pl = new (std::nothrow) modl_recsys_plan();
    if (!pl) return MODL_ENOMEM;
    pl->dtype = dtype; pl->p = p; pl->k = k; pl->max_batch = max_batch; pl->max_entries = max_entries;
    const size_t tsz = dtype == MODL_F32 ? 4 : 8;
    const int64_t u_max = max_entries < p ? max_entries : p;
    pl->stage_bytes = recsys_layout(tsz, max_batch, k, u_max, max_entries).total;
    pl->ack_off = pl->stage_bytes;                           // the acknowledgement word sits behind the staged arrays
    pl->ws_bytes = dict_update_workspace(dtype, p, k);
    pl->cnt.assign((size_t)p, 0);
    hipError_t e = hipMalloc((void **)&pl->dstage, 2 * align_up(pl->stage_bytes, 256));
    pl->dstage2[0] = pl->dstage;
    pl->dstage2[1] = pl->dstage ? pl->dstage + align_up(pl->stage_bytes, 256) : nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&pl->ws, pl->ws_bytes > 0 ? pl->ws_bytes : 16);
    if (e == hipSuccess && k <= 64) {
        e = hipMalloc((void **)&pl->part, tsz * (size_t)kRfMaxChunks * ((size_t)k * k + k));
        if (e == hipSuccess) e = hipMalloc((void **)&pl->tickets, sizeof(unsigned int) * (kRfMaxBatch + 1));
        if (e == hipSuccess) e = hipMemset(pl->tickets, 0, sizeof(unsigned int) * (kRfMaxBatch + 1));
        if (e == hipSuccess) e = hipMalloc((void **)&pl->xch, sizeof(double) * 3 * 64 * kRfMaxSweep);
        if (e == hipSuccess) {
            std::vector<long long> fill((size_t)3 * 64 * kRfMaxSweep, kRfSentinel);
            e = hipMemcpy(pl->xch, fill.data(), sizeof(long long) * fill.size(), hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&pl->pflags, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&pl->pflags_dev, pl->pflags, 0);
    if (e == hipSuccess) std::memset(pl->pflags, 0, 64);
    for (int i = 0; i < kRecsysSlots && e == hipSuccess; ++i) {
        e = hipHostMalloc((void **)&pl->h[i], pl->stage_bytes + 64, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&pl->hdev[i], pl->h[i], 0);
        if (e == hipSuccess) std::memset(pl->h[i] + pl->stage_bytes, 0, 64);
    }
    if (e != hipSuccess) {
        modl_recsys_plan_destroy(pl);
        return (int)e;
    }
    *out = pl;
    return MODL_OK;
}

void modl_recsys_plan_destroy(modl_recsys_plan *pl) {
    if (!pl) return;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < kRecsysSlots; ++i) {
        if (pl->h[i]) (void)hipHostFree(pl->h[i]);
    }
    if (pl->pflags) (void)hipHostFree(pl->pflags);
    if (pl->part) (void)hipFree(pl->part);
    if (pl->tickets) (void)hipFree(pl->tickets);
    if (pl->xch) (void)hipFree(pl->xch);
    if (pl->dbg) (void)hipFree(pl->dbg);
    if (pl->dstage) (void)hipFree(pl->dstage);
    if (pl->ws) (void)hipFree(pl->ws);
    delete pl;
}

#define MODL_RECSYS_MB_ARGS_OK                                                                                       \
    (pl && h_indptr && h_indices && h_data && d_indptr && d_indices && d_data && h_rows && h_order && d_Dt && d_Bt && \
     d_C && d_code && d_comp_norm && d_feature_n_iter && n_rows >= 0)
int modl_recsys_minibatch_f32(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices, const float *h_data,
                              int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const float *d_data,
                              const int64_t *h_rows, int64_t b, const int64_t *h_order, double alpha, double w,
                              double n_iter, float *d_Dt, float *d_Bt, float *d_C, float *d_code, float *d_comp_norm,
                              int64_t *d_feature_n_iter, void *stream) {
    if (!MODL_RECSYS_MB_ARGS_OK || pl->dtype != MODL_F32) return MODL_EINVAL;
    return recsys_minibatch<float>(pl, h_indptr, h_indices, h_data, n_rows, d_indptr, d_indices, d_data, h_rows, b, h_order,
                                   alpha, w, n_iter, d_Dt, d_Bt, d_C, d_code, d_comp_norm, d_feature_n_iter,
                                   (hipStream_t)stream);
}
int modl_recsys_minibatch_f64(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices, const double *h_data,
                              int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const double *d_data,
                              const int64_t *h_rows, int64_t b, const int64_t *h_order, double alpha, double w,
                              double n_iter, double *d_Dt, double *d_Bt, double *d_C, double *d_code,
                              double *d_comp_norm, int64_t *d_feature_n_iter, void *stream) {
    if (!MODL_RECSYS_MB_ARGS_OK || pl->dtype != MODL_F64) return MODL_EINVAL;
    return recsys_minibatch<double>(pl, h_indptr, h_indices, h_data, n_rows, d_indptr, d_indices, d_data, h_rows, b, h_order,
                                    alpha, w, n_iter, d_Dt, d_Bt, d_C, d_code, d_comp_norm, d_feature_n_iter,
                                    (hipStream_t)stream);
}

/* The per-minibatch host loop of RecsysDictFact.fit (recsys.py:135-139 + :147-165) for a run of minibatches in ONE call:
 * h_rows[n_rows_fit] = the (permuted) row ids of the run, cut into minibatches of batch_size (the last one ragged); per
 * minibatch n_iter += its rows, w = _batch_weight(n_iter, rows, learning_rate, 0), order = legacy permutation(k) of order_rng
 * (load it with numpy's state, modl_rk_set_mt_state), then modl_recsys_minibatch.  *n_done = minibatches enqueued (on an
 * error: before the failing one; n_iter and the generator are left behind the last enqueued one). */
#define ABI_RECSYS_FIT(SFX, T)                                                                                        \
    int modl_recsys_fit_batches_##SFX(modl_recsys_plan *pl, const int32_t *h_indptr, const int32_t *h_indices,          \
                                      const T *h_data, int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, \
                                      const T *d_data, const int64_t *h_rows, int64_t n_rows_fit, int64_t batch_size,   \
                                      modl_rk *order_rng, double alpha, double learning_rate, int64_t *n_iter,          \
                                      T *d_Dt, T *d_Bt, T *d_C, T *d_code, T *d_comp_norm, int64_t *d_feature_n_iter,   \
                                      void *stream, int64_t *n_done) {                                                \
        if (!pl || !h_indptr || !h_indices || !h_data || !d_indptr || !d_indices || !d_data || !h_rows || !d_Dt || !d_Bt ||  \
            !d_C || !d_code || !d_comp_norm || !d_feature_n_iter || n_rows < 0 || !order_rng || !n_iter ||              \
            batch_size <= 0 || n_rows_fit < 0 || pl->dtype != (sizeof(T) == 4 ? MODL_F32 : MODL_F64))                   \
            return MODL_EINVAL;                                                                                       \
        return recsys_fit_batches<T>(pl, h_indptr, h_indices, h_data, n_rows, d_indptr, d_indices, d_data, h_rows, n_rows_fit, \
                                     batch_size, order_rng, alpha, learning_rate, n_iter, d_Dt, d_Bt, d_C, d_code, d_comp_norm, \
                                     d_feature_n_iter, (hipStream_t)stream, n_done);                                    \
    }
ABI_RECSYS_FIT(f32, float)
ABI_RECSYS_FIT(f64, double)
#undef ABI_RECSYS_FIT

/* diagnostics: minibatches of this plan that ran as one launch / as separate launches */
int modl_recsys_plan_counts(const modl_recsys_plan *pl, int64_t *fused, int64_t *split) {
    if (!pl || !fused || !split) return MODL_EINVAL;
    *fused = pl->fused_calls;
    *split = pl->split_calls;
    return MODL_OK;
}

/* Synchronises `stream`; MODL_ETIMEOUT if a dictionary-update launch of this plan gave up since the last call (the update is
 * incomplete; the flag is cleared), MODL_OK otherwise. */
int modl_recsys_plan_status(modl_recsys_plan *pl, void *stream) {
    if (!pl) return MODL_EINVAL;
    MODL_HIP(hipStreamSynchronize((hipStream_t)stream));
    volatile unsigned int *f = pl->pflags;
    if (!f || f[0] == 0) return MODL_OK;
    f[0] = 0;
    return MODL_ETIMEOUT;
}

/* diagnostics: on = 1 allocates 16 words the last workgroup of every one-launch minibatch stamps with the 100 MHz wall clock
 * ([0] entry, [1] ids, [2] Gram chunk, [3] records summed, [4] factor, [5] ticket, [6] last phase, [7] C_, [8] B_, [9] sweep,
 * [10] end, [11] items, [12] sweep workgroups); h_out[16] (may be NULL) receives the last launch's (synchronises the device) */
int modl_recsys_plan_stamps(modl_recsys_plan *pl, int on, unsigned long long *h_out) {
    if (!pl) return MODL_EINVAL;
    if (on && !pl->dbg) {
        MODL_HIP(hipMalloc((void **)&pl->dbg, 16 * sizeof(unsigned long long)));
        MODL_HIP(hipMemset(pl->dbg, 0, 16 * sizeof(unsigned long long)));
    }
    if (h_out && pl->dbg) {
        MODL_HIP(hipDeviceSynchronize());
        MODL_HIP(hipMemcpy(h_out, pl->dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (!on && pl->dbg) { (void)hipFree(pl->dbg); pl->dbg = nullptr; }
    return MODL_OK;
}

/* diagnostics: host time (ms) this plan has spent waiting for a staging slot to be read by the device - zero while the host
 * is the bottleneck, the device's lead otherwise */
int modl_recsys_plan_wait_ms(const modl_recsys_plan *pl, double *ms) {
    if (!pl || !ms) return MODL_EINVAL;
    *ms = pl->wait_ms;
    return MODL_OK;
}

}  // extern "C"
