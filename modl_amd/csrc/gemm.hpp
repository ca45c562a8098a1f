// Tiled matrix-core contraction used by every dense product of the SOMF step
// (Dx, Gram, C_/B_ increments, the blocked dictionary update).
//
//   out(m, n) = epilogue( sum_kk  A(m, kk) * B(n, kk) )
//
// Both operands are described the same way: element (i, kk) lives at
// ptr[gi(i) * si + gk(kk) * sk] where gi / gk are optional gather indices
// (feature subset, sample indices, atom order).  That covers X[:, subset],
// D[:, subset] held feature-major, code_[idx] ... without materialising any
// gathered copy: the subsampled feature rows are read straight from Dt / Bt.
//
// gfx950 mapping: 256-thread workgroups = 4 wavefronts in a 2x2 grid, each wave
// owning RM x RN matrix-core tiles (f32: v_mfma_f32_32x32x2_f32, exact fp32 FMA
// chains; f64: v_mfma_f64_16x16x4_f64).  Operand tiles are staged k-major in LDS
// (+1 padding) so fragment reads are conflict-free 32-lane rows.  K can be split
// over gridDim.z; partial tiles are then summed in a fixed order by
// gemm_reduce_kernel, so results are run-to-run deterministic.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace modl {

struct Operand {
    const void *ptr = nullptr;
    int64_t si = 0, sk = 0;   // element strides of the free index and of the contraction index
    Gather gi, gk;
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static constexpr int TM = 32, TN = 32, TK = 2, NACC = 16;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int frag_i(int lane) { return lane & 31; }
    static __device__ __forceinline__ int frag_k(int lane) { return lane >> 5; }
    static __device__ __forceinline__ int acc_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int acc_col(int lane, int) { return lane & 31; }
};
// 16 x 16 x 4 f32 tiles: a quarter of the work of the 32 x 32 x 2 tile per instruction stream, for products whose
// 64 x 64 workgroup tiles would leave most of the chip idle (the statistics products: K = minibatch)
struct Mma16f {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static constexpr int TM = 16, TN = 16, TK = 4, NACC = 4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int frag_i(int lane) { return lane & 15; }
    static __device__ __forceinline__ int frag_k(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int acc_row(int lane, int r) { return 4 * (lane >> 4) + r; }
    static __device__ __forceinline__ int acc_col(int lane, int) { return lane & 15; }
};
template <> struct Mma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static constexpr int TM = 16, TN = 16, TK = 4, NACC = 4;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int frag_i(int lane) { return lane & 15; }
    static __device__ __forceinline__ int frag_k(int lane) { return lane >> 4; }
    static __device__ __forceinline__ int acc_row(int lane, int r) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ int acc_col(int lane, int) { return lane & 15; }
};

// an epilogue with `typedef ... Fetched; Fetched fetch(m, n) const; void finish(m, n, v, const Fetched &) const;` gets its
// per-output operands requested for all outputs before any is consumed (see gemm_kernel)
template <class Epi, class = void> struct epi_has_fetch { static constexpr bool value = false; };
template <class Epi> struct epi_has_fetch<Epi, std::void_t<typename Epi::Fetched>> { static constexpr bool value = true; };

// Staging of a tile whose rows mostly exist (tall configuration only), without a gather on the contraction index: the
// gather mode is resolved OUTSIDE the loops and every lane loads, from clamped coordinates, in two straight-line rounds
// (all index loads, then all data loads), the selection of zeros follows.  The general staging below keeps its loads
// behind per-thread branches (and the gather functor's own), after each of which the compiler waits for the load: 2
// serial memory round trips per element, 20 us of the 27 us of a 350 x 32 x 50 product.
template <typename T, int BI, int BK, int GI>
__device__ __forceinline__ void stage_tile_dense(T (*S)[BI + 1], const Operand &op, int64_t i0, int64_t I, int64_t k0,
                                                 int64_t k_end) {
    const T *base = static_cast<const T *>(op.ptr);
    constexpr int NL = BI * BK / 256;
    static_assert(BI * BK % 256 == 0, "tile elements per thread");
    const bool walk_i = op.si == 1;                            // workgroup-uniform
    int64_t off[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int e = threadIdx.x + t * 256;
        const int il = walk_i ? e % BI : e / BK, kl = walk_i ? e / BI : e % BK;
        const int64_t i = i0 + il, kk = k0 + kl;
        const int64_t ic = i < I ? i : I - 1, kc = kk < k_end ? kk : k_end - 1;
        const int64_t gi = GI == 0 ? ic : (GI == 1 ? (int64_t)op.gi.i32[ic] : op.gi.i64[ic]);
        off[t] = gi * op.si + kc * op.sk;
    }
    T v[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) v[t] = base[off[t]];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int e = threadIdx.x + t * 256;
        const int il = walk_i ? e % BI : e / BK, kl = walk_i ? e / BI : e % BK;
        S[kl][il] = (i0 + il < I && k0 + kl < k_end) ? v[t] : (T)0;
    }
}

template <typename T, int BI, int BK>
__device__ __forceinline__ void stage_tile(T (*S)[BI + 1], const Operand &op, int64_t i0, int64_t I, int64_t k0,
                                           int64_t k_end, bool dense = false) {
    if (dense && op.gk.identity()) {                           // workgroup-uniform
        if (op.gi.i32) stage_tile_dense<T, BI, BK, 1>(S, op, i0, I, k0, k_end);
        else if (op.gi.i64) stage_tile_dense<T, BI, BK, 2>(S, op, i0, I, k0, k_end);
        else stage_tile_dense<T, BI, BK, 0>(S, op, i0, I, k0, k_end);
        return;
    }
    const T *base = static_cast<const T *>(op.ptr);
    constexpr int kElems = BI * BK;
    if (op.si == 1) {   // free index contiguous: lanes walk i
#pragma unroll 4
        for (int e = threadIdx.x; e < kElems; e += 256) {
            const int il = e % BI, kl = e / BI;
            const int64_t i = i0 + il, kk = k0 + kl;
            T v = 0;
            if (i < I && kk < k_end) v = base[op.gi(i) * op.si + op.gk(kk) * op.sk];
            S[kl][il] = v;
        }
    } else {            // contraction index contiguous (or neither): lanes walk kk
#pragma unroll 4
        for (int e = threadIdx.x; e < kElems; e += 256) {
            const int kl = e % BK, il = e / BK;
            const int64_t i = i0 + il, kk = k0 + kl;
            T v = 0;
            if (i < I && kk < k_end) v = base[op.gi(i) * op.si + op.gk(kk) * op.sk];
            S[kl][il] = v;
        }
    }
}

template <typename T, int RM, int RN, int WGM, int WGN, int BK, class Epi>
__global__ __launch_bounds__(256) void gemm_kernel(Operand A, Operand B, int64_t M, int64_t N, int64_t K,
                                                   int64_t k_per_split, T *partial, Epi epi) {
    using MT = Mma<T>;
    static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
    constexpr int WTM = MT::TM * RM, WTN = MT::TN * RN, BM = WGM * WTM, BN = WGN * WTN;
    __shared__ T As[BK][BM + 1];
    __shared__ T Bs[BK][BN + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_split;
    const int64_t k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;

    typename MT::acc_t acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MT::NACC; ++r) acc[i][j][r] = 0;

    // tall configuration (short products, one workgroup per 128 rows): tiles at least half full take the dense staging
    constexpr bool kTall = (WGM == 4 && WGN == 1);
    const bool denseA = kTall && 2 * (M - m0) >= BM, denseB = kTall && 2 * (N - n0) >= BN;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
        stage_tile<T, BM, BK>(As, A, m0, M, k0, k_end, denseA);
        stage_tile<T, BN, BK>(Bs, B, n0, N, k0, k_end, denseB);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += MT::TK) {
            T af[RM], bf[RN];
            const int kr = kk + MT::frag_k(lane);
#pragma unroll
            for (int i = 0; i < RM; ++i) af[i] = As[kr][wm * WTM + i * MT::TM + MT::frag_i(lane)];
#pragma unroll
            for (int j = 0; j < RN; ++j) bf[j] = Bs[kr][wn * WTN + j * MT::TN + MT::frag_i(lane)];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = MT::mma(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    const bool direct = (gridDim.z == 1);
    if constexpr (epi_has_fetch<Epi>::value) {
        // Epilogues that READ per-output operands (gathered rows, per-column scalars): every output's operands are
        // requested first, unconditionally, from clamped coordinates (two dependent rounds at most: indices, then
        // data), and only the stores are guarded.  With the loads inside the per-output guard each output pays its
        // own serial memory round trips (16 outputs x 2 rounds per lane: 14 of the 29 us of the k x 32 product of
        // the f64 dictionary update).
        if (direct) {
            typename Epi::Fetched fetched[RM][RN][MT::NACC];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int r = 0; r < MT::NACC; ++r) {
                        const int64_t m = m0 + wm * WTM + i * MT::TM + MT::acc_row(lane, r);
                        const int64_t n = n0 + wn * WTN + j * MT::TN + MT::acc_col(lane, r);
                        fetched[i][j][r] = epi.fetch(m < M ? m : M - 1, n < N ? n : N - 1);
                    }
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int r = 0; r < MT::NACC; ++r) {
                        const int64_t m = m0 + wm * WTM + i * MT::TM + MT::acc_row(lane, r);
                        const int64_t n = n0 + wn * WTN + j * MT::TN + MT::acc_col(lane, r);
                        if (m < M && n < N) epi.finish(m, n, acc[i][j][r], fetched[i][j][r]);
                    }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MT::NACC; ++r) {
                const int64_t m = m0 + wm * WTM + i * MT::TM + MT::acc_row(lane, r);
                const int64_t n = n0 + wn * WTN + j * MT::TN + MT::acc_col(lane, r);
                if (m < M && n < N) {
                    if (direct) epi(m, n, acc[i][j][r]);
                    else partial[((int64_t)blockIdx.z * M + m) * N + n] = acc[i][j][r];
                }
            }
}

template <typename T, class Epi>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const T *partial, int splits, int64_t M, int64_t N,
                                                          Epi epi) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    T v = 0;
    for (int z = 0; z < splits; ++z) v += partial[(int64_t)z * M * N + e];
    epi(e / N, e % N, v);
}

// ---- epilogues --------------------------------------------------------------
template <typename T> struct EpiStore {          // out[m][n] = alpha * v
    T *out; int64_t ld; T alpha;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { out[m * ld + n] = alpha * v; }
};
// the statistics update  out <- (1 - w) out + w v / b  as an epilogue, leaving the rows stamped `step` alone
// (they were updated earlier from the head of the increment)
template <typename T> struct EpiStatsSkip {
    static constexpr bool rmw = true;
    T *out; int64_t ld; const int32_t *stamp; int32_t step; T beta, wt, bdiv; int replace;
    // load() is unconditional (no branch around a load: see gemm_tile_epilogue) and fetches the row's stamp along with
    // the old value; store() skips the stamped rows from registers (with the stamp read inside store() every element
    // of a tile's epilogue was a memory round trip of its own: 9 k cycles of a 31 k-cycle riding tile, measured)
    struct Old { T v; int32_t st; };
    __device__ __forceinline__ Old load(int64_t m, int64_t n) const { return Old{out[m * ld + n], stamp[m]}; }
    __device__ __forceinline__ void store(int64_t m, int64_t n, T v, const Old &old) const {
        if (old.st == step) return;
        out[m * ld + n] = replace ? v / bdiv : old.v * beta + (wt * v) / bdiv;
    }
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { store(m, n, v, load(m, n)); }
};
template <typename T> struct EpiAxpby {          // out[m][n] = beta * out[m][n] + alpha * v
    T *out; int64_t ld; T alpha, beta;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const {
        T *o = out + m * ld + n;
        *o = beta * (*o) + alpha * v;
    }
};

// ---- launcher ---------------------------------------------------------------
struct SplitWs {      // scratch for split-K partial tiles
    void *ptr = nullptr;
    size_t bytes = 0;
};

// Tile configurations (block tile BM x BN): "big" 128x128 (2x2 waves, most operand reuse), "small"
// 64x64 (2x2 waves, more workgroups for mid-size outputs), "tall" 128x32 (4x1 waves, for the narrow
// N = 32 products of the blocked dictionary update).
template <typename T> struct TileCfg;
// BKT: K-tile of the tall configuration.  Its products are short and launched one workgroup per 128 rows: every K-tile
// is a full memory round trip (no prefetch in this kernel), so fewer, deeper K-tiles (41 KB of LDS in f64).
template <> struct TileCfg<float> { static constexpr int BK = 16, BKT = 32, RB = 2, RS = 1, RT = 1; };
template <> struct TileCfg<double> { static constexpr int BK = 16, BKT = 32, RB = 4, RS = 2, RT = 2; };

// target_wgs: how many workgroups we would like in flight (256 CUs, a few per CU)
template <typename T, class Epi>
int launch_gemm(hipStream_t stream, const Operand &A, const Operand &B, int64_t M, int64_t N, int64_t K,
                const Epi &epi, const SplitWs &ws, int *launches = nullptr, int target_wgs = 512,
                int max_splits = 64) {
    if (M <= 0 || N <= 0) return MODL_OK;
    using C = TileCfg<T>;
    int cfg, bm, bn;                              // 0 big, 1 small, 2 tall
    if (N <= 32) { cfg = 2; bm = 128; bn = 32; }
    else if (cdiv(M, 128) * cdiv(N, 128) >= 512) { cfg = 0; bm = 128; bn = 128; }
    else { cfg = 1; bm = 64; bn = 64; }
    const int64_t tm = cdiv(M, bm), tn = cdiv(N, bn);
    int64_t splits = 1;
    if (K > 0 && tm * tn < target_wgs && max_splits > 1 && ws.ptr) {
        splits = target_wgs / (tm * tn);
        const int64_t max_by_k = K / (8 * C::BK) > 0 ? K / (8 * C::BK) : 1;   // >= 8 k-tiles per split
        if (splits > max_by_k) splits = max_by_k;
        if (splits > max_splits) splits = max_splits;
        const int64_t max_by_ws = (int64_t)(ws.bytes / sizeof(T)) / (M * N);
        if (splits > max_by_ws) splits = max_by_ws;
        if (splits < 1) splits = 1;
    }
    const int64_t Kp = K > 0 ? K : 1;
    const int bk = (cfg == 2) ? C::BKT : C::BK;
    const int64_t kps = cdiv(cdiv(Kp, splits), bk) * bk;
    splits = cdiv(Kp, kps);
    dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)splits);
    T *partial = static_cast<T *>(ws.ptr);
    if (cfg == 0)
        hipLaunchKernelGGL((gemm_kernel<T, C::RB, C::RB, 2, 2, C::BK, Epi>), grid, dim3(256), 0, stream, A, B, M, N, K,
                           kps, partial, epi);
    else if (cfg == 1)
        hipLaunchKernelGGL((gemm_kernel<T, C::RS, C::RS, 2, 2, C::BK, Epi>), grid, dim3(256), 0, stream, A, B, M, N, K,
                           kps, partial, epi);
    else
        hipLaunchKernelGGL((gemm_kernel<T, C::RT, C::RT, 4, 1, C::BKT, Epi>), grid, dim3(256), 0, stream, A, B, M, N, K,
                           kps, partial, epi);
    MODL_LAUNCH_CHECK();
    if (launches) ++*launches;
    if (splits > 1) {
        hipLaunchKernelGGL((gemm_reduce_kernel<T, Epi>), dim3((unsigned)cdiv(M * N, 256)), dim3(256), 0, stream,
                           partial, (int)splits, M, N, epi);
        MODL_LAUNCH_CHECK();
        if (launches) ++*launches;
    }
    return MODL_OK;
}

}  // namespace modl
