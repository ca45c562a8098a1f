// The p x k statistics product  X^T code  (dict_fact.py:573: B_ = (1 - w) B_ + (w / b) code^T X, held here
// feature-major) as a k-WIDE tile: one workgroup owns BM features and ALL atoms (up to 256 per chunk), so its slice
// of the minibatch X is fetched from HBM exactly once.
//
//   out(m = feature, n = atom) = epi( sum_kk X[kk][m] * code[kk][n] ),   kk = sample, K = minibatch <= 256
//
// The 32 x 32 tiles of gemm_stats_tile (gemm_dense.hpp) read their X tile once per 32-atom tile, i.e. k / 32 = 8
// times: 82 MB of HBM traffic for 10 MB of X at the metric's shape, and HBM-bound at config 5's shape (p = 200 000:
// 1.6 GB per minibatch).  Here: X once (coalesced 4 BM-byte row segments), the code matrix from L2 (256 KB, shared
// by every workgroup), B_ once each way.
//
// gfx950 mapping: 4 wavefronts, each all BM features x 64 atoms = (BM / 16) x 4 tiles of v_mfma_f32_16x16x4_f32;
// the whole X tile (K x BM) is staged in LDS once, the code matrix streams through a double-buffered 32-sample
// LDS tile (next tile requested before the matrix cores start on the current one, one barrier per tile); the old
// values of the read-modify-write epilogue are requested before the contraction starts.  LDS: BM = 64: 150 KB,
// BM = 32: 118 KB (one workgroup per compute unit).  Matrix-core bound: 2 BM 256 K flops per tile at 614 GFLOP/s
// per compute unit = 13.7 us (BM = 64, K = 256).
#pragma once
#include "gemm_dense.hpp"

namespace modl {

constexpr int kWideBK = 32, kWideKmax = 256;
// LDS row padding of 16 floats: a fragment read takes 4 consecutive rows x 16 consecutive floats; with a row stride of
// 16 (mod 32) banks the four rows fall on banks 0-15 / 16-31 alternately (2 lanes per bank, the minimum for 64 lanes);
// a stride of 4 (mod 32) makes it a 4-way conflict
constexpr int kWidePad = 16;

template <int BM, int BN = 256, int BK = kWideBK, int PAD = kWidePad> constexpr size_t wide_lds_bytes() {
    return sizeof(float) * ((size_t)kWideKmax * (BM + PAD) + 2 * (size_t)BK * (BN + PAD));
}

template <class Epi> struct WideProblem {
    const float *X = nullptr; int64_t ldx = 0;       // (kk, m) -> X[kk * ldx + m]
    const float *Cd = nullptr; int64_t ldc = 0;      // (kk, n) -> Cd[kk * ldc + n]
    int64_t M = 0;
    int N = 0, K = 0;
    Epi epi;
    int tm = 0, tn = 0;                              // feature tiles, atom chunks of BN
    bool vec4 = false;                               // (gemm_resident.hpp: the epilogue takes 16 bytes at a time)
    unsigned long long *dbg = nullptr;               // (diagnostics, gemm_resident.hpp: shader-clock stamps of workgroup 0)
    bool ok = false;
};

// eligible: 16-byte aligned operands, M, N multiples of 4, K <= 256
template <int BM, class Epi, int BN = 256>
WideProblem<Epi> plan_wide(const DenseOperand &A, const DenseOperand &B, int64_t M, int64_t N, int64_t K, const Epi &epi) {
    WideProblem<Epi> P;
    P.epi = epi;
    P.X = static_cast<const float *>(A.ptr); P.ldx = A.sk;
    P.Cd = static_cast<const float *>(B.ptr); P.ldc = B.sk;
    P.M = M; P.N = (int)N; P.K = (int)K;
    P.tm = (int)cdiv(M, BM); P.tn = (int)cdiv(N, BN);
    P.ok = A.si == 1 && B.si == 1 && M > 0 && N > 0 && K > 0 && K <= kWideKmax && M % 4 == 0 && N % 4 == 0 &&
           A.sk % 4 == 0 && B.sk % 4 == 0 && reinterpret_cast<uintptr_t>(A.ptr) % 16 == 0 &&
           reinterpret_cast<uintptr_t>(B.ptr) % 16 == 0;
    return P;
}

// one tile (256 threads: callers with larger workgroups retire the other threads first)
// BN atoms per tile (256: every atom of the metric's shape, X fetched once; 128: twice, half the tile time): each of
// the 4 wavefronts takes BN / 4 atoms
// SWZ: no padding; column c of row kk lives at c ^ ((kk & 1) << 4) instead - odd rows swap the two 16-float halves of
// every 32-float group, so the four rows of a fragment read fall on the two bank halves alternately exactly as with a
// row stride of 16 (mod 32), and a float4 stays a float4.  (PAD = 4 - what fits two workgroups per compute unit at
// BN = 256 - is the 4-way conflict of the note above.)
template <int BM, class Epi, int BN = 256, int BK = kWideBK, int PAD = kWidePad, bool SWZ = false>
__device__ __forceinline__ void gemm_wide_tile(const WideProblem<Epi> &P, int tile, char *smem, unsigned long long *dbg = nullptr) {
    auto sw = [](int kk, int c) { return SWZ ? (c ^ ((kk & 1) << 4)) : c; };
    if (dbg && threadIdx.x == 0) dbg[0] = clock64();   // diagnostics (scripts/diag_stamps.py): phases of one tile
    constexpr int TI = BM / 16, WN = BN / 4, TJ = WN / 16;
    constexpr int NA = kWideKmax * BM / 4 / 256;           // float4 of the X tile per thread
    constexpr int NB = BK * BN / 4 / 256;                  // float4 of a code tile per thread (8)
    typedef float f4v __attribute__((ext_vector_type(4)));
    float (*As)[BM + PAD] = reinterpret_cast<float (*)[BM + PAD]>(smem);
    float (*Bs)[BK][BN + PAD] = reinterpret_cast<float (*)[BK][BN + PAD]>(smem + sizeof(float) * kWideKmax * (BM + PAD));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bm = tile % P.tm, bn = tile / P.tm;
    const int64_t m0 = (int64_t)bm * BM;
    const int n0 = bn * BN;
    const int K = P.K, nkt = (K + BK - 1) / BK, kpad = nkt * BK;
    const int64_t M = P.M;
    const int N = P.N;

    // ---- requests: the whole X tile, then the first code tile (clamped addresses, no branches around loads)
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    {
        f4v a[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int e = tid + 256 * q, kk = e / (BM / 4), fv = (e % (BM / 4)) * 4;
            const int kc = kk < K ? kk : K - 1;
            const int64_t mc = (m0 + fv < M) ? m0 + fv : M - 4;
            a[q] = *reinterpret_cast<const f4v *>(P.X + (int64_t)kc * P.ldx + mc);
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int e = tid + 256 * q, kk = e / (BM / 4), fv = (e % (BM / 4)) * 4;
            const bool in = kk < K && m0 + fv < M;
            if (kk < kpad) *reinterpret_cast<f4v *>(&As[kk][sw(kk, fv)]) = in ? a[q] : zero4;
        }
    }
    f4v bq[NB];
    auto request_b = [&](int kt) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int e = tid + 256 * q, kl = e / (BN / 4), jv = (e % (BN / 4)) * 4;
            const int kk = kt * BK + kl, kc = kk < K ? kk : K - 1;
            const int jc = (n0 + jv < N) ? n0 + jv : N - 4;
            bq[q] = *reinterpret_cast<const f4v *>(P.Cd + (int64_t)kc * P.ldc + jc);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int e = tid + 256 * q, kl = e / (BN / 4), jv = (e % (BN / 4)) * 4;
            *reinterpret_cast<f4v *>(&Bs[buf][kl][sw(kl, jv)]) = (n0 + jv < N) ? bq[q] : zero4;   // (rows >= K meet zero rows of X)
        }
    };
    request_b(0);
    // the old values of a read-modify-write epilogue do not depend on the product: with 32 features per tile they are
    // requested now and wait in registers; with 64 (64 more registers: the tile would spill) after the contraction,
    // one more round trip behind 13 us of matrix-core work
    constexpr bool kRmw = EpiIsRmw<Epi>::value;
    constexpr bool kEarlyOld = kRmw && BM <= 32;
    EpiOld<Epi, float> old[TI][TJ][4];
    auto request_old = [&]() {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + 16 * ti + 4 * (lane >> 4) + r;
                    const int n = n0 + WN * wid + 16 * tj + (lane & 15);
                    old[ti][tj][r] = P.epi.load(m < M ? m : M - 1, n < N ? n : N - 1);
                }
    };
    if constexpr (kEarlyOld) request_old();
    if (dbg && threadIdx.x == 0) dbg[1] = clock64();
    store_b(0);
    gemm_lds_barrier();
    if (dbg && threadIdx.x == 0) dbg[2] = clock64();

    f4v acc[TI][TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = f4v{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) request_b(kt + 1);
        const int buf = kt & 1;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kr = ks * 4 + (lane >> 4);
            float af[TI], bf[TJ];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) af[ti] = As[kt * BK + kr][sw(kt * BK + kr, 16 * ti + (lane & 15))];
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) bf[tj] = Bs[buf][kr][sw(kr, WN * wid + 16 * tj + (lane & 15))];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_b(buf ^ 1);               // last read in step kt - 1, before that step's barrier
        gemm_lds_barrier();
    }
    if (dbg && threadIdx.x == 0) dbg[3] = clock64();
    if constexpr (kRmw && !kEarlyOld) request_old();
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + 16 * ti + 4 * (lane >> 4) + r;
                const int n = n0 + WN * wid + 16 * tj + (lane & 15);
                if (m < M && n < N) {
                    if constexpr (kRmw) P.epi.store(m, n, acc[ti][tj][r], old[ti][tj][r]);
                    else P.epi(m, n, acc[ti][tj][r]);
                }
            }
    if (dbg && threadIdx.x == 0) dbg[4] = clock64();
}

// the small problem (code^T code -> C_, 32 x 32 tiles of gemm_stats_tile) and the wide one in ONE launch
template <int BM, class Epi0, class Epi1, int BN = 256, int BK = kWideBK, int PAD = kWidePad, bool SWZ = false>
__global__ __launch_bounds__(256) void gemm_stats_wide_pair_kernel(DenseProblem<float, Epi0> P0, WideProblem<Epi1> P1) {
    extern __shared__ __attribute__((aligned(16))) char wide_smem[];
    int id = (int)blockIdx.x;
    const int t0 = P0.tn * P0.tm;
    if (id < t0) {
        gemm_stats_tile<Epi0>(P0, id, wide_smem);
        return;
    }
    id -= t0;
    if (id >= P1.tm * P1.tn) return;
    gemm_wide_tile<BM, Epi1, BN, BK, PAD, SWZ>(P1, id, wide_smem);
}

template <int BM, class Epi0, class Epi1, int BN = 256, int BK = kWideBK, int PAD = kWidePad, bool SWZ = false>
int launch_gemm_stats_wide_pair(hipStream_t stream, const DenseProblem<float, Epi0> &P0, const WideProblem<Epi1> &P1,
                                int *launches = nullptr) {
    const int total = P0.tn * P0.tm + P1.tm * P1.tn;
    if (total <= 0) return MODL_OK;
    constexpr size_t lds = wide_lds_bytes<BM, BN, BK, PAD>() > kStatsLds ? wide_lds_bytes<BM, BN, BK, PAD>() : kStatsLds;
    auto kern = gemm_stats_wide_pair_kernel<BM, Epi0, Epi1, BN, BK, PAD, SWZ>;
    MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, stream, P0, P1);
    MODL_LAUNCH_CHECK();
    if (launches) ++*launches;
    return MODL_OK;
}

}  // namespace modl
