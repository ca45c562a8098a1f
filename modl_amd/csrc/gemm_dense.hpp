// Dense, software-pipelined matrix-core contraction (no gathers):
//
//   out(m, n) = epilogue( sum_kk A(m, kk) * B(n, kk) ),   X(i, kk) = ptr[i * si + kk * sk], si == 1 or sk == 1
//
// This is the fast path of the SOMF step: the step first compacts what it
// gathers (sampled dictionary rows, sampled columns of the minibatch, the
// minibatch's code rows), then every product is a plain strided GEMM.
//
// gfx950 mapping: 64x64 block tile, BK = 32, 4 wavefronts (2x2) each owning one
// 32x32 v_mfma_f32_32x32x2_f32 tile (f64: 2x2 v_mfma_f64_16x16x4_f64 tiles).
// Global -> register -> LDS staging with 16-byte loads along the contiguous
// dimension; the loads of K-tile t+1 are issued before the MFMA loop of tile t
// and written to the other LDS buffer afterwards (one barrier per K-tile).
// Operands are k-major in LDS (row stride BI + 4: 16-byte aligned vector stores,
// conflict-free fragment reads).  Split-K partial tiles are reduced in a fixed
// order (gemm_reduce_kernel) -> deterministic results.
#pragma once
#include "gemm.hpp"

namespace modl {

struct DenseOperand {
    const void *ptr = nullptr;
    int64_t si = 0, sk = 0;
};

template <typename T> struct Vec4;   // 16-byte vector of T
template <> struct Vec4<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec4<double> { typedef double2 type; static constexpr int N = 2; };

// Stage one BI x BK operand tile global -> registers (16-byte vectors along the contiguous dim).
// NV vectors per thread.  FAST = tile fully inside the matrix; otherwise element-wise guarded.
template <typename T, int BI, int BK, bool IFAST>
struct TileLoader {
    static constexpr int VN = Vec4<T>::N;
    static constexpr int NV = BI * BK / VN / 256;
    typedef typename Vec4<T>::type V;
    V r[NV];

    __device__ __forceinline__ void load(const DenseOperand &op, int64_t i0, int64_t I, int64_t k0, int64_t k_end) {
        const T *base = static_cast<const T *>(op.ptr);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = threadIdx.x + 256 * q;
            int il, kl;
            if (IFAST) { il = (e % (BI / VN)) * VN; kl = e / (BI / VN); }
            else { kl = (e % (BK / VN)) * VN; il = e / (BK / VN); }
            const int64_t i = i0 + il, kk = k0 + kl;
            const T *p = base + i * op.si + kk * op.sk;
            const bool full = IFAST ? (i + VN <= I && kk < k_end) : (i < I && kk + VN <= k_end);
            if (full) {
                r[q] = *reinterpret_cast<const V *>(p);
            } else {
                T t[VN];
#pragma unroll
                for (int c = 0; c < VN; ++c) {
                    const bool in = IFAST ? (i + c < I && kk < k_end) : (i < I && kk + c < k_end);
                    t[c] = in ? p[c] : (T)0;
                }
                V v;
                T *vp = reinterpret_cast<T *>(&v);
#pragma unroll
                for (int c = 0; c < VN; ++c) vp[c] = t[c];
                r[q] = v;
            }
        }
    }
    __device__ __forceinline__ void store(T (*S)[BI + 4]) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = threadIdx.x + 256 * q;
            if (IFAST) {
                const int il = (e % (BI / VN)) * VN, kl = e / (BI / VN);
                *reinterpret_cast<V *>(&S[kl][il]) = r[q];
            } else {
                const int kl = (e % (BK / VN)) * VN, il = e / (BK / VN);
                const T *t = reinterpret_cast<const T *>(&r[q]);
#pragma unroll
                for (int c = 0; c < VN; ++c) S[kl + c][il] = t[c];
            }
        }
    }
};

// One output tile of one (possibly K-split) product.  (bx, by, bz) = tile column, tile row, split index.
template <typename T, bool AIFAST, bool BIFAST, class Epi, int BM, int BN, int BK>
__device__ __forceinline__ void gemm_dense_tile(const DenseOperand &A, const DenseOperand &B, int64_t M, int64_t N,
                                                int64_t K, int64_t k_per_split, T *partial, const Epi &epi, int bx, int by,
                                                int bz, int nsplit, T (*As)[BK][BM + 4], T (*Bs)[BK][BN + 4]) {
    using MT = Mma<T>;
    constexpr int WM = BM / 2, WN = BN / 2;               // 2 x 2 waves
    constexpr int RM = WM / MT::TM, RN = WN / MT::TN;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
    const int64_t k_begin = (int64_t)bz * k_per_split;
    const int64_t k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;

    typename MT::acc_t acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MT::NACC; ++r) acc[i][j][r] = 0;

    TileLoader<T, BM, BK, AIFAST> la;
    TileLoader<T, BN, BK, BIFAST> lb;
    la.load(A, m0, M, k_begin, k_end);
    lb.load(B, n0, N, k_begin, k_end);
    la.store(As[0]);
    lb.store(Bs[0]);
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) {
            la.load(A, m0, M, k0 + BK, k_end);
            lb.load(B, n0, N, k0 + BK, k_end);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += MT::TK) {
            T af[RM], bf[RN];
            const int kr = kk + MT::frag_k(lane);
#pragma unroll
            for (int i = 0; i < RM; ++i) af[i] = As[cur][kr][wm * WM + i * MT::TM + MT::frag_i(lane)];
#pragma unroll
            for (int j = 0; j < RN; ++j) bf[j] = Bs[cur][kr][wn * WN + j * MT::TN + MT::frag_i(lane)];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = MT::mma(af[i], bf[j], acc[i][j]);
        }
        if (more) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
        cur ^= 1;
    }

    const bool direct = (nsplit == 1);
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < MT::NACC; ++r) {
                const int64_t m = m0 + wm * WM + i * MT::TM + MT::acc_row(lane, r);
                const int64_t n = n0 + wn * WN + j * MT::TN + MT::acc_col(lane, r);
                if (m < M && n < N) {
                    if (direct) epi(m, n, acc[i][j][r]);
                    else partial[((int64_t)bz * M + m) * N + n] = acc[i][j][r];
                }
            }
}

template <typename T, bool AIFAST, bool BIFAST, class Epi>
__global__ __launch_bounds__(256) void gemm_dense_kernel(DenseOperand A, DenseOperand B, int64_t M, int64_t N, int64_t K,
                                                         int64_t k_per_split, T *partial, Epi epi) {
    constexpr int BM = 64, BN = 64, BK = (sizeof(T) == 4) ? 32 : 16;
    __shared__ __attribute__((aligned(16))) T As[2][BK][BM + 4];
    __shared__ __attribute__((aligned(16))) T Bs[2][BK][BN + 4];
    gemm_dense_tile<T, AIFAST, BIFAST, Epi, BM, BN, BK>(A, B, M, N, K, k_per_split, partial, epi, (int)blockIdx.x,
                                                        (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.z, As, Bs);
}

// ---- two independent products in ONE launch (and their split-K reductions in one more) -------------
// At the metric's shape the products of a minibatch are tiny (0.1 GFLOP): a launch costs more than the
// arithmetic, so the independent ones travel together.
template <typename T, class Epi> struct DenseProblem {
    DenseOperand A, B;
    int64_t M = 0, N = 0, K = 0, kps = 0;
    T *partial = nullptr;
    Epi epi;
    int tn = 0, tm = 0, splits = 1;
    bool ok = false;                                  // aligned: eligible for the dense kernels
};

template <typename T, class Epi>
DenseProblem<T, Epi> plan_dense(const DenseOperand &A, const DenseOperand &B, int64_t M, int64_t N, int64_t K, const Epi &epi,
                                T *ws, size_t ws_elems, int target_wgs = 512, int max_splits = 64, int bm = 64, int bn = 64) {
    DenseProblem<T, Epi> P;
    P.A = A; P.B = B; P.M = M; P.N = N; P.K = K; P.epi = epi;
    constexpr int VN = Vec4<T>::N;
    auto aligned = [](const DenseOperand &o) {
        if (o.si != 1 && o.sk != 1) return false;
        const int64_t ld = (o.si == 1) ? o.sk : o.si;
        return (reinterpret_cast<uintptr_t>(o.ptr) % 16 == 0) && (ld % VN == 0);
    };
    P.ok = aligned(A) && aligned(B) && K > 0 && M > 0 && N > 0;
    if (!P.ok) return P;
    constexpr int BK = (sizeof(T) == 4) ? 32 : 16;
    const int64_t tm = cdiv(M, bm), tn = cdiv(N, bn);
    int64_t splits = 1;
    if (tm * tn < target_wgs && max_splits > 1 && ws) {
        splits = target_wgs / (tm * tn);
        const int64_t max_by_k = K / (4 * BK) > 0 ? K / (4 * BK) : 1;       // >= 4 k-tiles per split
        if (splits > max_by_k) splits = max_by_k;
        if (splits > max_splits) splits = max_splits;
        const int64_t max_by_ws = (int64_t)ws_elems / (M * N);
        if (splits > max_by_ws) splits = max_by_ws;
        if (splits < 1) splits = 1;
    }
    P.kps = cdiv(cdiv(K, splits), BK) * BK;
    P.splits = (int)cdiv(K, P.kps);
    P.tn = (int)tn; P.tm = (int)tm;
    P.partial = ws;
    return P;
}

template <typename T, bool AI0, bool BI0, class Epi0, bool AI1, bool BI1, class Epi1, int BM1 = 64, int BN1 = 64, int BK1 = 0>
__global__ __launch_bounds__(256) void gemm_dense_pair_kernel(DenseProblem<T, Epi0> P0, DenseProblem<T, Epi1> P1) {
    constexpr int BM = 64, BN = 64, BK = (sizeof(T) == 4) ? 32 : 16;
    extern __shared__ __attribute__((aligned(16))) char gd_smem[];   // sized for the larger of the two tilings
    int id = (int)blockIdx.x;
    const int t0 = P0.tn * P0.tm * P0.splits;
    if (id < t0) {
        T (*As)[BK][BM + 4] = reinterpret_cast<T (*)[BK][BM + 4]>(gd_smem);
        T (*Bs)[BK][BN + 4] = reinterpret_cast<T (*)[BK][BN + 4]>(gd_smem + sizeof(T) * 2 * BK * (BM + 4));
        const int bx = id % P0.tn, by = (id / P0.tn) % P0.tm, bz = id / (P0.tn * P0.tm);
        gemm_dense_tile<T, AI0, BI0, Epi0, BM, BN, BK>(P0.A, P0.B, P0.M, P0.N, P0.K, P0.kps, P0.partial, P0.epi, bx, by, bz,
                                                       P0.splits, As, Bs);
    } else {
        id -= t0;
        // XCD-aware order for the large problem: workgroups are dealt round-robin to the 8 XCDs (each with its
        // own L2), so the tiles that share an A row-tile are renumbered to land on ONE XCD and fetch it once
        const int t1 = P1.tn * P1.tm * P1.splits, chunk = (t1 + 7) / 8;
        id = (id % 8) * chunk + id / 8;
        if (id >= t1) return;
        constexpr int BKB = BK1 > 0 ? BK1 : BK;
        T (*As)[BKB][BM1 + 4] = reinterpret_cast<T (*)[BKB][BM1 + 4]>(gd_smem);
        T (*Bs)[BKB][BN1 + 4] = reinterpret_cast<T (*)[BKB][BN1 + 4]>(gd_smem + sizeof(T) * 2 * BKB * (BM1 + 4));
        const int bx = id % P1.tn, by = (id / P1.tn) % P1.tm, bz = id / (P1.tn * P1.tm);
        gemm_dense_tile<T, AI1, BI1, Epi1, BM1, BN1, BKB>(P1.A, P1.B, P1.M, P1.N, P1.K, P1.kps, P1.partial, P1.epi, bx, by, bz,
                                                         P1.splits, As, Bs);
    }
}

template <typename T, class Epi0, class Epi1>
__global__ __launch_bounds__(256) void gemm_reduce_pair_kernel(const T *partial0, int splits0, int64_t M0, int64_t N0, Epi0 epi0,
                                                               int nblk0, const T *partial1, int splits1, int64_t M1,
                                                               int64_t N1, Epi1 epi1) {
    if ((int)blockIdx.x < nblk0) {
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= M0 * N0) return;
        T v = 0;
        for (int z = 0; z < splits0; ++z) v += partial0[(int64_t)z * M0 * N0 + e];
        epi0(e / N0, e % N0, v);
    } else {
        const int64_t e = (int64_t)((int)blockIdx.x - nblk0) * 256 + threadIdx.x;
        if (e >= M1 * N1) return;
        T v = 0;
        for (int z = 0; z < splits1; ++z) v += partial1[(int64_t)z * M1 * N1 + e];
        epi1(e / N1, e % N1, v);
    }
}

// Both problems must be `ok` (see plan_dense) and their operand orientations known statically; the second
// problem may use a larger block tile (BM1 x BN1, planned with the same values).
template <typename T, bool AI0, bool BI0, class Epi0, bool AI1, bool BI1, class Epi1, int BM1 = 64, int BN1 = 64, int BK1 = 0>
int launch_gemm_dense_pair(hipStream_t stream, const DenseProblem<T, Epi0> &P0, const DenseProblem<T, Epi1> &P1,
                           int *launches = nullptr) {
    constexpr int BK = (sizeof(T) == 4) ? 32 : 16;
    constexpr int BKB = BK1 > 0 ? BK1 : BK;
    constexpr size_t lds0 = sizeof(T) * 2 * BK * ((64 + 4) + (64 + 4)), lds1 = sizeof(T) * 2 * BKB * ((BM1 + 4) + (BN1 + 4));
    constexpr size_t lds = lds0 > lds1 ? lds0 : lds1;
    const int total = P0.tn * P0.tm * P0.splits + 8 * ((P1.tn * P1.tm * P1.splits + 7) / 8);   // second problem padded to 8 XCD chunks
    auto kern = gemm_dense_pair_kernel<T, AI0, BI0, Epi0, AI1, BI1, Epi1, BM1, BN1, BK1>;
    if (lds > 64 * 1024)
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, stream, P0, P1);
    MODL_LAUNCH_CHECK();
    if (launches) ++*launches;
    const int nb0 = P0.splits > 1 ? (int)cdiv(P0.M * P0.N, 256) : 0, nb1 = P1.splits > 1 ? (int)cdiv(P1.M * P1.N, 256) : 0;
    if (nb0 + nb1 > 0) {
        hipLaunchKernelGGL((gemm_reduce_pair_kernel<T, Epi0, Epi1>), dim3((unsigned)(nb0 + nb1)), dim3(256), 0, stream,
                           P0.partial, P0.splits, P0.M, P0.N, P0.epi, nb0, P1.partial, P1.splits, P1.M, P1.N, P1.epi);
        MODL_LAUNCH_CHECK();
        if (launches) ++*launches;
    }
    return MODL_OK;
}

// Dense launcher.  Requires 16-byte aligned bases and leading strides that keep every vector
// aligned; otherwise falls back to the generic gather kernel (gemm.hpp).
template <typename T, class Epi>
int launch_gemm_dense(hipStream_t stream, const DenseOperand &A, const DenseOperand &B, int64_t M, int64_t N, int64_t K,
                      const Epi &epi, const SplitWs &ws, int *launches = nullptr, int target_wgs = 512,
                      int max_splits = 64) {
    if (M <= 0 || N <= 0) return MODL_OK;
    const DenseProblem<T, Epi> P = plan_dense<T, Epi>(A, B, M, N, K, epi, static_cast<T *>(ws.ptr), ws.bytes / sizeof(T),
                                                      target_wgs, max_splits);
    if (!P.ok) {
        Operand a, b;
        a.ptr = A.ptr; a.si = A.si; a.sk = A.sk;
        b.ptr = B.ptr; b.si = B.si; b.sk = B.sk;
        return launch_gemm<T, Epi>(stream, a, b, M, N, K, epi, ws, launches, target_wgs, max_splits);
    }
    dim3 grid((unsigned)P.tn, (unsigned)P.tm, (unsigned)P.splits);
    const bool ai = A.si == 1, bi = B.si == 1;
#define MODL_GD(AI, BI) \
    hipLaunchKernelGGL((gemm_dense_kernel<T, AI, BI, Epi>), grid, dim3(256), 0, stream, A, B, M, N, K, P.kps, P.partial, epi)
    if (ai && bi) MODL_GD(true, true);
    else if (ai) MODL_GD(true, false);
    else if (bi) MODL_GD(false, true);
    else MODL_GD(false, false);
#undef MODL_GD
    MODL_LAUNCH_CHECK();
    if (launches) ++*launches;
    if (P.splits > 1) {
        hipLaunchKernelGGL((gemm_reduce_kernel<T, Epi>), dim3((unsigned)cdiv(M * N, 256)), dim3(256), 0, stream,
                           P.partial, P.splits, M, N, epi);
        MODL_LAUNCH_CHECK();
        if (launches) ++*launches;
    }
    return MODL_OK;
}

// ---- compaction kernels -------------------------------------------------------
// dst[r][0..cols) = src[idx[r]][0..cols)   (rows r >= n_rows of the padded destination are zero)
template <typename T, typename I>
__global__ __launch_bounds__(256) void gather_rows_T_kernel(const T *src, int64_t src_ld, const I *idx, int64_t n_rows,
                                                            int64_t n_rows_pad, int64_t cols, T *dst, int64_t dst_ld) {
    const int64_t r = blockIdx.x;
    if (r >= n_rows_pad) return;
    T *d = dst + r * dst_ld;
    if (r < n_rows) {
        const T *s = src + (int64_t)idx[r] * src_ld;
        for (int64_t c = threadIdx.x; c < cols; c += 256) d[c] = s[c];
    } else {
        for (int64_t c = threadIdx.x; c < cols; c += 256) d[c] = 0;
    }
}
// src[idx[r]][0..cols) = dst-compact[r][0..cols)
template <typename T, typename I>
__global__ __launch_bounds__(256) void scatter_rows_T_kernel(T *dst, int64_t dst_ld, const I *idx, int64_t n_rows,
                                                             int64_t cols, const T *src, int64_t src_ld) {
    const int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    T *d = dst + (int64_t)idx[r] * dst_ld;
    const T *s = src + r * src_ld;
    for (int64_t c = threadIdx.x; c < cols; c += 256) d[c] = s[c];
}
}  // namespace modl
