// Declarations shared between the translation units of libmodl_hip.
#pragma once
#include "common.hpp"

namespace modl {

// ---- cd_solver.hip ----------------------------------------------------------
template <typename T>
struct CdArgs {
    const T *G;            // [k][k] shared, or [b][k][k] when g_stride != 0
    int64_t g_stride;
    const int64_t *g_idx;  // per-sample Gram index (rows of G_average_), or null: sample ii uses G[ii]
    const T *Dx;           // [b][k]
    const T *xnorm2;       // [b] squared norm of the full sample row
    const T *H0;           // [b][k] initial Q w (from a matrix-core product), or null: computed in-kernel
    T *code;               // [n][k]
    const int64_t *idx;    // [b] or null (identity)
    T *code2 = nullptr;    // optional second destination: code2[idx2[i]] = solution i
    int g_pad_rows = 0;    // rows readable behind G (shared Gram only): >= 16 lets the row prefetch run unclamped
    int ldg = 0;           // row stride of G (0: k).  A shared Gram padded with zeros to ldg = 64 * ceil(k / 64 rounded to
                           // a power of two) columns and ldg + 16 rows runs the vectorised kernel whatever k is
                           // (launch_cd_pad_gram / cd_padded_ld)
    const int64_t *idx2 = nullptr;
    int32_t *sweeps;       // [b] or null
    int b, k;
    T alpha, beta, tol;
    int max_iter, positive;
    int sparse_pct = 20;   // a sweep with at most sparse_pct % of the coordinates active runs as a sparse sweep
                           // (measured at k = 256: a sparse step costs 0.23 us, a dense sweep 14.3 us (f32) / 18.9 us (f64))
};
template <typename T> int launch_cd(hipStream_t stream, const CdArgs<T> &a);
// cd_split.hip: the same solver with the chain and the k-wide update on two wavefronts (shared Gram, 64 < k <= 512)
template <typename T> bool cd_split_applies(const CdArgs<T> &a);
template <typename T> int launch_cd_split(hipStream_t stream, const CdArgs<T> &a);
// a Gram matrix PER SAMPLE whose size is not one of the solver's strides: slices of the minibatch through zero-padded
// copies (slots: scratch of cd_per_sample_scratch_bytes, zero outside the k x k corners - zero-filled once per k)
bool cd_split_enabled();
int cd_per_sample_ld(int k);
size_t cd_per_sample_scratch_bytes(size_t tsz, int64_t b, int k);
template <typename T> int launch_cd_per_sample(hipStream_t stream, const CdArgs<T> &a, T *slots, size_t slot_bytes);
bool cd_one_wave_covers(int k);   // the one-wavefront kernel exists for k coefficients (k <= 256; any k in the diagnostics library)
int cd_padded_ld(int k);   // the row stride the vectorised solver wants for k coefficients (k itself when it fits already)
// Gp[ldg + 16][ldg] (zero-filled once by the caller) <- G[k][k]
template <typename T> int launch_cd_pad_gram(hipStream_t stream, const T *G, int k, T *Gp, int ldg);
template <typename T> int launch_row_norm2(hipStream_t stream, const T *X, int64_t ldx, int64_t p, int64_t b, T *out);

// ---- chol.hip ---------------------------------------------------------------
// factor nmat matrices (G[m] + alpha I) into F[m] (symmetric storage of the
// Cholesky factor: F[i][j] = L[max(i,j)][min(i,j)])
template <typename T>
int launch_cholesky(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *g_idx, T *F, int k, T alpha,
                    int nmat);
// solve F[m] F[m]^T x = rhs for b right-hand sides (rows of rhs, in place);
// f_stride == 0: one shared factor.  Results also scattered to code rows.
// k <= 128: factor (G[m] + alpha I) and solve in one launch, the matrix in LDS (shared matrix: all b right-hand sides;
// g_stride != 0: one matrix and one right-hand side per workgroup).  rhs solved in place, also written to the code rows.
template <typename T> bool ridge_small_applies(int k);
template <typename T>
int launch_ridge_small(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *g_idx, T *rhs, int b, int k, T alpha,
                       T *code, const int64_t *idx);
template <typename T>
int launch_chol_solve(hipStream_t stream, const T *F, int64_t f_stride, T *rhs, int b, int k, T *code,
                      const int64_t *idx);

// wide systems (k > 512): blocked factorisation + substitutions on the matrix cores.  G: shared (g_stride == 0) or
// one Gram per sample at G + h_gidx[i] * g_stride (HOST index array, or null = i); F: k*k scratch; Linv:
// chol_wide_scratch_elems(k) scratch; rhs[b][k]: right-hand sides, overwritten by the solutions, which are also
// written to code rows d_idx (device, or null = row i).
size_t chol_wide_scratch_elems(int k);
bool chol_blocked(int k, size_t tsz, bool shared);   // which ridge systems take the blocked factorisation
template <typename T>
int ridge_solve_wide(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *h_gidx, T *F, T *Linv, T *rhs, int b,
                     int k, T alpha, T *code, const int64_t *d_idx);

// ---- enet.hip ---------------------------------------------------------------
template <typename T>
int launch_enet_norm(hipStream_t stream, const T *v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                     T *out);
template <typename T>
int launch_enet_projection(hipStream_t stream, const T *v, T *out, int64_t rows, int64_t n, int64_t ld,
                           int64_t inc, const T *radius, T l1_ratio);
template <typename T>
int launch_enet_scale(hipStream_t stream, T *v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                      T radius);
template <typename T>
int launch_update_G_average(hipStream_t stream, T *G_average, const int64_t *idx, const T *G, const T *w_sample,
                            int64_t b, int64_t k);
template <typename T>
int launch_transpose(hipStream_t stream, const T *in, T *out, int64_t rows, int64_t cols);

// ---- bcd.hip ----------------------------------------------------------------
// Work that rides along the block launches of the fused dictionary update, on the compute units that update
// leaves idle: the statistics update of the rows of Bt that were NOT sampled,
//   Bt[f] <- beta Bt[f] + wt (X^T code)[f] / bdiv   for stamp[f] != step,
// a p x k x b product whose result the dictionary update does not need.  dict_update sets `consumed` when it
// took the work; otherwise the caller runs it on its own.
struct StatsRider {
    const void *X; int64_t ldx;           // minibatch rows [b][ldx]
    const void *code;                     // [b][k] the minibatch's code rows (compact)
    int b; int64_t p;
    void *Bt;                             // [p][k]
    const int32_t *stamp; int32_t step;
    double beta, wt, bdiv; int replace;
    int consumed;
};

// The parameter block of the NEXT minibatch (sample indices, subset, order, sample weights: a few KB of a pinned host
// slot) copied to HBM by one extra workgroup of the dictionary update's last launch instead of a launch of its own at
// the head of the next step (4.4 us of PCIe round trip on an otherwise idle chip, every step).  off16 / n16: up to four
// ranges of 16-byte words; the acknowledgement word tells the host the slot may be refilled.
struct StageRide {
    const uint4 *src = nullptr;
    uint4 *dst = nullptr;
    unsigned int off16[4] = {0, 0, 0, 0}, n16[4] = {0, 0, 0, 0};
    unsigned long long *ack = nullptr;
    unsigned long long use = 0;
    int consumed = 0;
};
// the copy itself, by `nthr` threads of ONE workgroup (every load of a round requested before the first store: the
// reads cross the host link)
__device__ __forceinline__ void stage_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, const unsigned int (&off16)[4],
                                           const unsigned int (&n16)[4], unsigned long long *ack, unsigned long long use,
                                           int tid, int nthr) {
    constexpr int kMax = 4;
    const size_t c1 = n16[0], c2 = c1 + n16[1], c3 = c2 + n16[2], total = c3 + n16[3];
    for (size_t base = 0; base < total; base += (size_t)kMax * nthr) {
        uint4 v[kMax];
        size_t at[kMax];
#pragma unroll
        for (int u = 0; u < kMax; ++u) {
            size_t e = base + tid + (size_t)u * nthr;
            e = e < total ? e : total - 1;                        // (clamped, no branch around the load)
            at[u] = e < c1 ? off16[0] + e : (e < c2 ? off16[1] + (e - c1) : (e < c3 ? off16[2] + (e - c2) : off16[3] + (e - c3)));
            v[u] = src[at[u]];
        }
#pragma unroll
        for (int u = 0; u < kMax; ++u)
            if (base + tid + (size_t)u * nthr < total) dst[at[u]] = v[u];
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ack, use, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
struct DictUpdateArgs {
    T *Dt;                    // [p][k] dictionary, feature-major, updated in place on the subset rows
    const T *Bt;              // [p][k]
    const T *C;               // [k][k] (bitwise symmetric)
    T *comp_norm;             // [k]
    const int32_t *subset;    // [s] device, or null = rows 0..s-1
    const int32_t *order;     // [k] device
    const int64_t *h_order;   // [k] host copy (the generic path issues one launch pair per atom)
    int64_t s;
    int k;
    int optimizer, comp_pos;
    double comp_l1_ratio, w, step_size;
    void *ws;                 // scratch, dict_update_workspace() bytes
    size_t ws_bytes;
    StatsRider *rider = nullptr;   // optional (f32 fused path only)
    StageRide *stage = nullptr;    // optional (f32 fused path only): rides the last launch; `consumed` says whether it did
    unsigned int *persist_flags = nullptr;   // optional, pinned host memory, PERSISTENT across calls (zero-initialised), BcdPersistArgs::flags:
                                   // [0] a persistent launch gave up half-way (update incomplete), [1] launches that could not
                                   // run and were completed by one workgroup
    bool allow_persist = true;     // false: one launch per block (a plan whose persistent launch could not run once)
    double *level_hint = nullptr;  // optional [k], PERSISTENT across calls (zero-initialised): the soft-threshold level
                                   // each atom's l1 / elastic-net projection ended with, warm start of the next one
};
size_t dict_update_workspace(int dtype, int64_t s_max, int k);
size_t dict_update_stamps_offset(int dtype, int64_t s_max, int k);
size_t dict_update_persist_stamps_offset(int dtype, int64_t s_max, int k);
template <typename T> int dict_update(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches);

}  // namespace modl
