// The p x k statistics product  X^T code  (dict_fact.py:573: B_ = (1 - w) B_ + (w / b) code^T X, feature-major here)
// with the CODE matrix resident in registers.
//
//   out(m = feature, n = atom) = epi( sum_kk X[kk][m] * code[kk][n] ),   kk = sample, K = minibatch <= 256, n < 256
//
// gemm_wide.hpp keeps the X tile of a workgroup in LDS and streams the code matrix through LDS once per 32 features: a
// barrier and an L2 round trip every 8 samples, 48 % of the f32 matrix rate at p = 200 000 with three workgroups per
// compute unit hiding each other's waits; the 32 x 32 tiles of gemm_stats_tile read X eight times (27.7 us at p = 10 000,
// 30 %).  Here the operand every tile shares never moves again: a workgroup is EIGHT wavefronts, each owning 32 atoms,
// and a wavefront holds its 32 columns of the code matrix as matrix-core B fragments - 64 k-steps x 2 tiles = 128
// registers - for the whole launch.  The workgroups are persistent (one per compute unit, two wavefronts per SIMD) and
// walk over feature tiles of FT = 16 or 32 features: the X tile (K x FT) is the only thing that passes through LDS
// (double-buffered, requested a whole tile ahead, ONE barrier per tile), the old values of the read-modify-write epilogue
// are requested at the top of the tile and wait in registers under its 128 / 256 matrix-core instructions per wavefront.
// Per tile and wavefront the loop is one or two LDS reads and two or four v_mfma_f32_16x16x4_f32 per k-step, on two or
// four independent accumulators; two wavefronts per SIMD keep its matrix pipe fed.
//
// Roof: 2 FT 256 K flops per tile at 256 flop/cycle per compute unit (four SIMDs x 2048 flop / 32 cycles) = 8192 cycles
// per 16 features at K = 256.  p = 10 000: 625 tiles of 16 on 256 compute units = 3 rounds = 24.6 k cycles = 10.2 us (the
// quantisation costs 19 %: 2.44 tiles per unit); p = 200 000: 6250 tiles of 32, 25 rounds, 171 us.
// Summation order: the k-steps in sample order on ONE accumulator, as gemm_wide_tile (identical bits).
#pragma once
#include "gemm_wide.hpp"
#ifndef RES_EXP
#define RES_EXP 0      // (scripts/micro/res_gemm.hip: timing experiments that drop a part of the tile loop; 0 in the library)
#endif

namespace modl {

constexpr int kResThreads = 512;
template <int FT> constexpr size_t resident_lds_bytes() { return sizeof(float) * 2 * (size_t)kWideKmax * FT; }

// an epilogue that can take four consecutive n at once (load4 / store4 / vec4_ok: somf_step.hip's EpiStats)
template <class E, class = void> struct EpiHasVec4 : std::false_type {};
template <class E> struct EpiHasVec4<E, std::void_t<decltype(&E::vec4_ok)>> : std::true_type {};

// eligible (plan_wide's conditions, N <= 256, K % 4 == 0): see launch_gemm_stats_resident_pair.
// The matrix instruction computes the TRANSPOSED tile (A operand = code fragment, B operand = X fragment): a lane then holds
// four consecutive ATOMS of one feature - 16 contiguous bytes of the feature-major B_ - instead of one atom of four
// consecutive features, four scattered words.  VEC: the epilogue takes them as one 16-byte load and one 16-byte store.
template <int FT, class Epi, bool VEC>
__device__ __forceinline__ void gemm_resident_loop(const WideProblem<Epi> &P, int wg, int nwg, char *smem) {
    constexpr int NP = FT / 16;                              // panels of 16 features
    constexpr int NKS = kWideKmax / 4;                       // k-steps of the matrix-core instruction
    constexpr int NX = kWideKmax * FT / 4 / kResThreads;     // float4 of an X tile per thread (2 / 4)
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float panel_t[kWideKmax][16];
    panel_t *Xs = reinterpret_cast<panel_t *>(smem);         // [2 buffers][NP panels][K][16]: a fragment read is 64 consecutive floats
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = P.K, N = P.N;
    const int64_t M = P.M;
    const int ntile = (int)((M + FT - 1) / FT);
    // workgroup wg runs on XCD wg % 8: the workgroups of an XCD take NEIGHBOURING tiles (both halves of a 128-byte line of an
    // X row - a tile's row segment is 64 bytes at FT = 16 - then come through the same L2)
    const int first = (nwg % 8 == 0) ? (wg % 8) * (nwg / 8) + wg / 8 : wg;
    if (first >= ntile) return;

    // ---- requests: the first X tile, then this wavefront's 32 columns of the code matrix
    f4v xr[NX];
    auto request_x = [&](int t) {
        const int64_t m0 = (int64_t)t * FT;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int e = tid + kResThreads * q, kk = e / (FT / 4), fv = (e % (FT / 4)) * 4;
            const int kc = kk < K ? kk : K - 1;
            const int64_t mc = (m0 + fv < M) ? m0 + fv : M - 4;
            xr[q] = *reinterpret_cast<const f4v *>(P.X + (int64_t)kc * P.ldx + mc);
        }
    };
    auto store_x = [&](int t, int buf) {                     // (zeros beyond K and beyond M)
        const int64_t m0 = (int64_t)t * FT;
        const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int e = tid + kResThreads * q, kk = e / (FT / 4), fv = (e % (FT / 4)) * 4;
            const bool in = kk < K && m0 + fv < M;
            *reinterpret_cast<f4v *>(&Xs[buf * NP + (fv >> 4)][kk][fv & 15]) = in ? xr[q] : zero4;
        }
    };
    request_x(first);
    const int nks = K >> 2;                                  // (K % 4 == 0: a k-step is inside K or outside, for every lane)
    // row i = lane & 15 of the fragment of tile tj is atom 32 wid + 2 i + tj: a lane's two tiles are NEIGHBOURS in memory (one
    // 8-byte load per k-step instead of two 4-byte ones - the prologue is bound by the number of load instructions, 1024
    // per compute unit of 64 scattered words each took 6 us), and what it holds of the product is eight consecutive atoms
    float bf[NKS][2];                                        // code[4 ks + (lane >> 4)][32 wid + 2 (lane & 15) + tj]
    {
        typedef float f2v __attribute__((ext_vector_type(2)));
        const int kq = lane >> 4, n = 32 * wid + 2 * (lane & 15);
        const float *cp = P.Cd + (int64_t)kq * P.ldc + (n < N ? n : N - 2);          // (N % 4 == 0, the rows 8-byte aligned)
        // (no selection after the loads - the compiler turns `cond ? loaded : 0` into a branch around the load and waits for
        //  each one on the spot, 128 serial L2 round trips, 21 us, measured.  None is needed: k-steps beyond K are skipped
        //  (and meet zero rows of X), columns beyond N are never stored; the clamped addresses hold finite code values)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int64_t off = (int64_t)(4 * (ks < nks ? ks : 0)) * P.ldc;   // (uniform)
            const f2v v = *reinterpret_cast<const f2v *>(cp + off);
            bf[ks][0] = v[0];
            bf[ks][1] = v[1];
        }
    }
    store_x(first, 0);
    gemm_lds_barrier();

    constexpr bool kRmw = EpiIsRmw<Epi>::value;
    // what a lane holds of panel pn: feature m0 + 16 pn + (lane & 15), atoms nb + 0 .. 7 (acc[pn][tj][r] is atom nb + 2 r + tj),
    // as two groups h = 0, 1 of four consecutive atoms
    const int fl = lane & 15, nb = 32 * wid + 8 * (lane >> 4);
    typedef EpiOld<Epi, float> old_t;
    struct OldTile { f4v v[NP][2]; old_t s[NP][2][4]; };     // (one of the two is used)
    auto load_old = [&](OldTile &o, int64_t m0) {
        if constexpr (kRmw) {
#pragma unroll
            for (int pn = 0; pn < NP; ++pn)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int64_t m = m0 + 16 * pn + fl;
                    const int n = nb + 4 * h;
                    if constexpr (VEC) {
                        o.v[pn][h] = P.epi.load4(m < M ? m : M - 1, n < N ? n : N - 4);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o.s[pn][h][c] = P.epi.load(m < M ? m : M - 1, n + c < N ? n + c : N - 1);
                    }
                }
        }
    };
    auto use_old = [&](OldTile &o) {                         // an unconditional use: see the loop
        if constexpr (kRmw) {
#pragma unroll
            for (int pn = 0; pn < NP; ++pn)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (VEC) {
                        asm volatile("" : "+v"(o.v[pn][h]));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(o.s[pn][h][c]));
                    }
                }
        }
    };
    auto epilogue = [&](f4v (&a)[NP][2], OldTile &o, int64_t m0) {
#pragma unroll
        for (int pn = 0; pn < NP; ++pn)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t m = m0 + 16 * pn + fl;
                const int n = nb + 4 * h;
                const f4v val = {a[pn][0][2 * h], a[pn][1][2 * h], a[pn][0][2 * h + 1], a[pn][1][2 * h + 1]};   // atoms n .. n + 3
                if constexpr (VEC) {
                    if (m < M && n < N) P.epi.store4(m, n, val, o.v[pn][h]);             // (N % 4 == 0)
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (m < M && n + c < N) {
                            if constexpr (kRmw) P.epi.store(m, n + c, val[c], o.s[pn][h][c]);
                            else P.epi(m, n + c, val[c]);
                        }
                }
            }
    };
    // The two wavefronts of a SIMD (w and w + 4) run HALF A TILE APART: wavefronts 0-3 keep a tile's accumulators and old
    // values and run its epilogue at the top of the NEXT tile, while wavefronts 4-7 are in their matrix instructions;
    // wavefronts 4-7 run theirs before the barrier, while 0-3 are in the second half of their matrix instructions.  (A
    // SIMD issues ONE vector instruction at a time, matrix or not: stamps show an epilogue of ~200 instructions taking 5-6 k
    // cycles next to a wavefront in its matrix instructions - one slot per 32-cycle v_mfma.  What counts is the number of
    // other instructions per tile; the stagger keeps the matrix pipe busy while they trickle through.)
    const bool lag = (FT == 16) && wid < 4;                  // (wave-uniform; tiles of 32 features: no registers for the second set)
    f4v pacc[NP][2];
    OldTile pold;
    int64_t pm0 = -1;
    unsigned long long *dbg = (P.dbg && wg == 0 && lane == 0 && (wid & 3) == 0) ? P.dbg + (wid >> 2) * 64 : nullptr;
    int ns = 0;
    auto stamp = [&]() { if (dbg && ns < 64) dbg[ns++] = clock64(); };
    if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    int buf = 0;
    for (int t = first; t < ntile; t += nwg, buf ^= 1) {
        const int64_t m0 = (int64_t)t * FT;
        stamp();
        // (the next tile - clamped to this one after the last - is requested and stored WITHOUT a branch, and the old values
        //  get an unconditional use: a load whose only use sits behind a branch counts as possibly outstanding at the
        //  loop header, and with stores in flight the compiler then waits for EVERYTHING there - vmcnt(0) right after the
        //  requests of the next tile, a memory round trip per tile, 6.3 instead of 3.4 us per tile, measured)
        const int tn = t + nwg < ntile ? t + nwg : t;
#if RES_EXP != 2
        request_x(tn);
#endif
#if RES_EXP != 1
        if (lag && pm0 >= 0) epilogue(pacc, pold, pm0);
#endif
        stamp();
        OldTile old;
#if RES_EXP != 1
        load_old(old, m0);
#endif
        f4v acc[NP][2];
#pragma unroll
        for (int pn = 0; pn < NP; ++pn)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[pn][tj] = f4v{0.f, 0.f, 0.f, 0.f};
        const float *xs = &Xs[buf * NP][0][0] + lane;            // fragment of k-step ks, panel pn: xs[pn * K * 16 + 64 ks]
        auto steps = [&](auto lo_, auto hi_) {
            constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
            auto one = [&](int ks) {
                float af[NP];
#pragma unroll
                for (int pn = 0; pn < NP; ++pn) af[pn] = xs[pn * kWideKmax * 16 + 64 * ks];
#pragma unroll
                for (int pn = 0; pn < NP; ++pn)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[pn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ks][tj], af[pn], acc[pn][tj], 0, 0, 0);
            };
            if (nks == NKS) {                                // K = 256: straight-line code, the LDS reads run ahead of the matrix instructions
#pragma unroll
                for (int ks = lo; ks < hi; ++ks) one(ks);
            } else {                                         // k-steps in groups of eight: one uniform test per group (rows K .. of X and of the fragments are zeros)
#pragma unroll
                for (int g = lo; g < hi; g += 8) {
                    if (g < nks) {
#pragma unroll
                        for (int ks = g; ks < g + 8; ++ks) one(ks);
                    }
                }
            }
        };
        // the next tile goes to the other buffer (last read in the previous tile, before that tile's barrier) in the middle
        // of this one: its loads have had half a tile to arrive
        stamp();
        steps(std::integral_constant<int, 0>{}, std::integral_constant<int, NKS / 2>{});
        stamp();
#if RES_EXP != 2
        store_x(tn, buf ^ 1);
#endif
        stamp();
        steps(std::integral_constant<int, NKS / 2>{}, std::integral_constant<int, NKS>{});
        stamp();
#if RES_EXP != 1
        use_old(old);
#endif
        if (lag) {
#pragma unroll
            for (int pn = 0; pn < NP; ++pn)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) pacc[pn][tj] = acc[pn][tj];
            pold = old;
            pm0 = m0;
        } else {
#if RES_EXP != 1
            epilogue(acc, old, m0);
#else
            if (acc[0][0][0] == 123.456f) epilogue(acc, old, m0);
#endif
        }
        stamp();
#if RES_EXP != 3
        gemm_lds_barrier();
#endif
    }
    stamp();
    if (lag && pm0 >= 0) epilogue(pacc, pold, pm0);
    stamp();
}

// the small problem (code^T code -> C_, 32 x 32 tiles of gemm_stats_tile) rides the same launch: its tiles go, after the
// feature tiles, to the first four wavefronts of the workgroups that had the FEWEST feature tiles (the last ones in tile
// order: at p = 10 000 143 of the 256 workgroups have two tiles instead of three, a third of their time to spare).  As
// workgroups of their own they would take a compute unit each from the persistent ones for their duration: the register
// allocation is the kernel's, three of these wavefronts do not fit a SIMD.
template <int FT, class Epi0, class Epi1>
__global__ __launch_bounds__(kResThreads) void gemm_stats_resident_pair_kernel(DenseProblem<float, Epi0> P0, WideProblem<Epi1> P1) {
    extern __shared__ __attribute__((aligned(16))) char res_smem[];
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    unsigned long long *kd = (P1.dbg && threadIdx.x == 0 && (wg == 0 || wg == nwg - 1)) ? P1.dbg + (wg == 0 ? 112 : 120) : nullptr;
    if (kd) kd[0] = clock64();
    if constexpr (EpiHasVec4<Epi1>::value) {
        if (P1.vec4) gemm_resident_loop<FT, Epi1, true>(P1, wg, nwg, res_smem);
        else gemm_resident_loop<FT, Epi1, false>(P1, wg, nwg, res_smem);
    } else {
        gemm_resident_loop<FT, Epi1, false>(P1, wg, nwg, res_smem);
    }
    if (kd) kd[1] = clock64();
    const int t0 = P0.tn * P0.tm;
    if (t0 <= 0 || threadIdx.x >= 256) return;
    const int first = (nwg % 8 == 0) ? (wg % 8) * (nwg / 8) + wg / 8 : wg;
    for (int i = nwg - 1 - first; i < t0; i += nwg) gemm_stats_tile<Epi0>(P0, i, res_smem);
    if (kd) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); kd[2] = clock64(); }
}

// ncu: compute units of the device (one persistent workgroup each)
template <int FT, class Epi0, class Epi1>
int launch_gemm_stats_resident_pair(hipStream_t stream, const DenseProblem<float, Epi0> &P0, WideProblem<Epi1> P1, int ncu,
                                    int *launches = nullptr) {
    if constexpr (EpiHasVec4<Epi1>::value) P1.vec4 = P1.epi.vec4_ok();
    const int ntile = (int)cdiv(P1.M, FT);
    int nwg = ntile < ncu ? ntile : ncu;
    if (nwg >= 8) nwg &= ~7;                                  // (whole XCD rounds: the tile order above)
    if (nwg <= 0) return MODL_OK;
    constexpr size_t lds = resident_lds_bytes<FT>() > kStatsLds ? resident_lds_bytes<FT>() : kStatsLds;
    auto kern = gemm_stats_resident_pair_kernel<FT, Epi0, Epi1>;
    MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kResThreads), lds, stream, P0, P1);
    MODL_LAUNCH_CHECK();
    if (launches) ++*launches;
    return MODL_OK;
}

}  // namespace modl
