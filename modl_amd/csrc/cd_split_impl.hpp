// Batched elastic-net code solver, shared Gram matrix, 32 <= k <= 1024: the cyclic coordinate descent of
//   enet_coordinate_descent_gram (reference: modl/decomposition/dict_fact_fast.pyx:270-427)
// with the Gauss-Seidel chain taken OFF the k-wide update and every load taken off both.
//
// cd_kernel (cd_solver.hip) runs a sample on one wavefront: every coordinate is 5 chain instructions, two
// v_readlane, four v_pk_fma_f32 (the k-wide H <- H - w_old Q_i + w_new Q_i), a select and a 16-byte row load:
// 14 instructions, ~110 cycles at one wavefront per SIMD, and the next coordinate cannot start before all of them
// have issued - although it only needs ONE element of the k-wide update.  Here a sample is a WORKGROUP of four
// wavefronts, one per SIMD of a compute unit:
//   * the CHAIN wave owns the coordinates in blocks of 64 (lane l of block b <-> coordinate 64 b + l).  Inside a
//     block it keeps ONE register: Z[m] = q[m] - (H[m] - Q[m][m] w_old[m]) - sum over the block's coordinates j < m
//     already visited of Q[j][m] (w_new[j] - w_old[j]), i.e. the `tmp` of dict_fact_fast.pyx:367 of every coordinate
//     of the block AS IT WILL BE at that coordinate's turn once the steps before it have been applied.  What it needs
//     of the Gram matrix is the STRICTLY UPPER triangle of the block's 64 x 64 diagonal tile (LDS): a step only moves
//     the Z of the coordinates AFTER it, so the Z of a visited coordinate is frozen at the value its step used, and
//     re-evaluating the step formula on all 64 lanes (which every step does) reproduces the visited coordinates'
//     results bit for bit - no per-coordinate write-back (v_writelane) of the result, no separate w_old pass.  Per
//     coordinate: v_med3, v_sub, v_fma (delta = w_new - w_old), v_readlane, v_fma (Z), ds_read: SIX instructions
//     (round 3: ten).  Every 8 coordinates it publishes the block's new coefficients and a monotonic counter to LDS;
//   * two UPDATE waves (a column half each) apply the published steps to all k entries of H = Q w, in coordinate
//     order, with the reference's two fused multiply-adds per element (H <- fma(w_new, Q_j, fma(-w_old, Q_j, H)): a single
//     fma with the rounded difference is MORE accurate per operation and was measured 1.7 x NOISIER on the codes), a chunk
//     of 8 behind the chain, from a register ring of Gram rows requested a ring ahead; the counter and the chunk's pairs
//     come in ONE LDS round trip.  They hand the next block its 64 entries of H through LDS: a snapshot taken 32
//     coordinates before the block's end - the chain wave applies those last 32 steps to the next block itself (LOOK-AHEAD:
//     one more fma per coordinate with the delta it has just broadcast, on a strip of rows the tile loader brings), so it
//     no longer waits ~1200 cycles at every block boundary for the update waves to catch up;
//   * the TILE loader brings the diagonal tile of the chain's next block (plain 16-byte loads), clears its lower
//     triangle and diagonal in registers and stores it to one of two LDS buffers.
// Same sweep order, skip rule, step formula and both stopping tests as the reference, the same two roundings per element
// on H; what differs from its operation order is the rounding INSIDE a block of 64 coordinates (the chain wave's private
// Z takes one fma with the rounded difference per earlier coordinate of the block), so the iterates agree with
// cd_kernel / the oracle to rounding noise, not bit for bit (tests/test_gpu_kernels.py:
// test_cd_two_solvers_agree: f64 <= 1e-12 with identical sweep counts, f32 within the f32 noise rule).  The gap test
// reads w, q and H in cd_kernel's element order (lane l <-> elements KPL l ..).
//
// Why four waves (measured with in-kernel stamps, scripts/diag_cd_split_stamps.py, k = 256, f32): a wave at one per
// SIMD issues an instruction every ~4.5-6 cycles whether or not it depends on the one before (a dependent fma chain
// runs at 4.2-5.8 cycles per link, scripts/micro/launch_gap.hip), so what a coordinate costs the chain wave is its
// INSTRUCTION COUNT; a memory instruction costs the issuing wave 18 (global_load) to 60+ cycles (direct-to-LDS load).
// Counters in LDS are monotonic; data is stored before its counter and the LDS executes a wavefront's operations in
// order; spins are bounded by the workgroup's own progress (all four waves are resident together).  One sample per
// workgroup: a minibatch of 256 fills the chip's 256 compute units.
#pragma once
#include "kernels.hpp"
#include "cd_common.hpp"
#include <atomic>
#include <type_traits>

namespace modl {


typedef __attribute__((address_space(3))) volatile int lds_vi32;
typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;

constexpr int kStop = 0x7fffffff;
extern std::atomic<unsigned long long *> g_cd_stamps;   // diagnostics, cd_split.hip

// Spin until the LDS counter at `p` is >= need; returns the value read (or `cached` if that already suffices).  The
// loop is ONE assembly block: the compiler sees straight-line code around it, so its s_waitcnt bookkeeping for the
// row loads in flight stays exact (with a C loop here it waited for ALL outstanding rows at every join: measured).
__device__ __forceinline__ int spin_until(int cached, const int *p_lds, int need) {
    const unsigned int addr = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const int *)p_lds;
    int v = cached, sv;
    asm volatile(
        "v_readfirstlane_b32 %1, %0\n\t"
        "s_cmp_ge_i32 %1, %3\n\t"
        "s_cbranch_scc1 modl_spin_done%=\n"
        "modl_spin_loop%=:\n\t"
        "ds_read_b32 %0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %1, %0\n\t"
        "s_cmp_lt_i32 %1, %3\n\t"
        "s_cbranch_scc1 modl_spin_loop%=\n"
        "modl_spin_done%=:"
        : "+&v"(v), "=&s"(sv) : "v"(addr), "s"(need) : "memory", "scc");
    return v;
}

// NB = row stride of the Gram matrix / 64 (the matrix is padded with dead coordinates up to it, cd_padded_ld)
// FULL: k == 64 NB (every block has its 64 coordinates): the sweep is unrolled without a branch, see the update waves
template <typename T, int NB, bool POSITIVE, bool FULL>
__global__ __launch_bounds__(256) void cd_split_kernel(CdArgs<T> a, unsigned long long *stamps) {
    constexpr int K = 64 * NB;                        // row stride of the Gram matrix
    constexpr int KPL = NB;                           // cd_kernel's layout: element e <-> register e % KPL of lane e / KPL
    constexpr unsigned int TRB = 64 * sizeof(T);      // bytes of a row of a diagonal tile
    constexpr unsigned int TB = 64 * TRB;             // bytes of a tile: 16 / 32 KiB
    // LA: look-ahead.  The chain wave applies the block's LAST LA coordinates to the NEXT block's entries itself (one
    // extra fma per coordinate, with the delta it has just broadcast, on a strip of rows the tile loader brings), so
    // what it needs from the update waves at a block boundary is H as it was LA coordinates before the block's end:
    // they run ~1200 cycles (two LDS round trips and a chunk) behind, which the chain used to wait for, four times a sweep.
#ifndef MODL_CD_LA
#define MODL_CD_LA 32
#endif
    constexpr int LA = FULL ? MODL_CD_LA : 0;         // (a multiple of 8, < 64; A/B builds override it: scripts/build_variant.sh)
    // MR: f32 with k <= 256 - a column half of a row is 256 / 512 bytes, i.e. 4 / 8 bytes per lane of an update wave: such
    // accesses run at 0.5-0.7 of the 16-byte rate and every memory instruction costs the compute unit's address pipe 16
    // cycles whatever its width - the update waves, not the chain, bounded the sweep (57 cycles per coordinate against
    // 48, stamps).  There ONE 16-byte-per-lane request fetches RPI consecutive rows (a group of 64 / RPI lanes per row, 4
    // columns per lane) and every group of lanes accumulates the rows of its residue class: RPI partial sums per entry of
    // H, added (in a fixed order) when H is handed over.
    constexpr bool MR = FULL && sizeof(T) == 4 && NB <= 4;
    constexpr int RPI = MR ? 8 / NB : 1;              // rows per request: 4 (k = 128), 2 (k = 256)
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[2 * TB];
    __shared__ __attribute__((aligned(16))) T s_strip[LA > 0 ? 2 * LA * 64 : 4];   // rows 64 b + 64 - LA .. of the columns of block b + 1
    __shared__ __attribute__((aligned(16))) T s_wn[2 * 64];   // the chain wave's current block: new coefficients, by lane (blocks alternate) ...
    __shared__ __attribute__((aligned(16))) T s_wo[2 * 64];   // ... and the coefficients the block had before the sweep touched it
    __shared__ __attribute__((aligned(16))) T s_H[K]; // H after a whole block (cd_kernel's element order = plain order)
    __shared__ __attribute__((aligned(16))) T s_Hs[K];// H after the first 64 - LA coordinates of a block (look-ahead hand-off)
    __shared__ T s_w[K];                              // the chain wave's coefficients, for the gap test
    __shared__ int s_cnt[12];
    typedef __attribute__((address_space(3))) volatile T lds_vT;
    lds_vT *wns = (lds_vT *)s_wn;
    lds_vT *wos = (lds_vT *)s_wo;
    lds_vi32 *sver = (lds_vi32 *)&s_cnt[8];           // update waves 0 / 1 (s_cnt[8], [9]): 2 + the block whose snapshot is in s_Hs
    lds_vi32 *prog = (lds_vi32 *)&s_cnt[0];           // chain: 64 * (blocks finished) + coordinates published of the current one
    lds_vi32 *ver = (lds_vi32 *)&s_cnt[1];            // update waves 0 / 1 (s_cnt[1], [2]): 1 + blocks applied completely
                                                      // (their halves of the next block's H are in s_H)
    lds_vi32 *tiles = (lds_vi32 *)&s_cnt[4];          // tile loader: tiles in LDS (tile t serves the chain's block t)
    lds_vi32 *cblk = (lds_vi32 *)&s_cnt[5];           // chain: blocks finished (their tile buffers are free)
    lds_vi32 *stopf = (lds_vi32 *)&s_cnt[6];          // chain: the solve is over

    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;                 // 0 chain, 1 and 2 update (column halves), 3 tile loader
    const int smp = blockIdx.x;
    if (smp != 0) stamps = nullptr;
    int nst = 0;
    (void)nst;
#ifdef MODL_DIAG     // shader-clock stamps of sample 0: the diagnostics build only (the product kernels carry no stamp code)
#define MODL_STAMP(off) do { if (stamps && lane == 0 && nst < 250) stamps[(off) + nst++] = clock64(); } while (0)
#else
#define MODL_STAMP(off) do { } while (0)
#endif
    const int k = a.k;
    const int kc = FULL ? K : (k + 31) / 32 * 32;     // coordinates a sweep visits (the padding is dead: inv = 0)
    const int nblk = FULL ? NB : (kc + 63) / 64;
    // (a Gram matrix per sample, G_agg = 'average': the workgroup's own matrix; the index load is scalar, Q stays uniform)
    const T *__restrict__ Q = a.G + (a.g_stride ? (a.g_idx ? a.g_idx[smp] : (int64_t)smp) * a.g_stride : 0);
    const int64_t row_out = a.idx ? a.idx[smp] : (int64_t)smp;
    T *wptr = a.code + row_out * k;
    const T *qptr = a.Dx + (int64_t)smp * k;
    const T alpha = a.alpha, beta = a.beta;
    if (threadIdx.x < 12) s_cnt[threadIdx.x] = 0;
    __syncthreads();

    if (wid == 3) {
        // ------------------------------------------------------------------ tile loader
        // tile t = rows and columns 64 b .. 64 b + 63 of the matrix (b = t mod nblk) -> buffer t % 2, row-major, as
        // soon as the chain has finished block t - 2.  Only its STRICTLY UPPER triangle is kept (the rest is zero): a
        // step of the chain wave moves the coordinates after it and leaves the visited ones exactly as they were.
        constexpr int VE = (int)(16 / sizeof(T));                // elements per 16-byte unit: 4 (f32), 2 (f64)
        constexpr int UPR = 64 / VE;                             // units per tile row
        constexpr int RPP = 64 / UPR;                            // tile rows per piece of 64 units: 4 (f32), 2 (f64)
        constexpr int TP = 64 / RPP;                             // pieces per tile
        typedef T tvec_t __attribute__((ext_vector_type(VE)));
        const int r_l = lane / UPR, c_l = (lane % UPR) * VE;     // this lane's row within a piece, first column
        constexpr int SP = LA > 0 ? LA / RPP : 1;
        tvec_t pc[TP], sp[SP];
        // the loads of a tile (and of its strip) into registers: requested for tile t + 1 as soon as tile t is in LDS, i.e.
        // a whole block of the chain wave before its buffer is free - when the chain finishes a block only the masking and
        // the LDS stores remain (round 4 stamps: with the loads requested AFTER the buffer had become free the chain wave
        // waited ~490 cycles for its tile at every block boundary)
        auto request_tile = [&](int t) {
            const int b = t % nblk;
            const T *src = Q + (int64_t)(64 * b + r_l) * K + 64 * b + c_l;
#pragma unroll
            for (int i = 0; i < TP; ++i) pc[i] = *reinterpret_cast<const tvec_t *>(src + (int64_t)(i * RPP) * K);
            if constexpr (LA > 0) {
                // the strip of the transition b -> b + 1 (the sweep's last block -> block 0): the block's last LA rows,
                // the NEXT block's columns, as they are
                const int nb2 = (b + 1 == nblk) ? 0 : b + 1;
                const T *ssrc = Q + (int64_t)(64 * b + 64 - LA + r_l) * K + 64 * nb2 + c_l;
#pragma unroll
                for (int i = 0; i < SP; ++i) sp[i] = *reinterpret_cast<const tvec_t *>(ssrc + (int64_t)(i * RPP) * K);
            }
        };
        request_tile(0);
        int fin = 0;
        for (int t = 0;; ++t) {
            if (t - __builtin_amdgcn_readfirstlane(fin) >= 2) {
                int st = 0;
                do {
                    fin = *cblk;
                    st = *stopf;
                    __builtin_amdgcn_s_sleep(2);
                } while (t - __builtin_amdgcn_readfirstlane(fin) >= 2 && __builtin_amdgcn_readfirstlane(st) == 0);
                if (__builtin_amdgcn_readfirstlane(st) != 0) break;
            }
            T *dst = reinterpret_cast<T *>(s_tile + (unsigned int)(t & 1) * TB) + r_l * 64 + c_l;
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                const int row = i * RPP + r_l;
                tvec_t v = pc[i];
#pragma unroll
                for (int e = 0; e < VE; ++e) v[e] = (c_l + e > row) ? v[e] : (T)0;
                *reinterpret_cast<tvec_t *>(dst + i * RPP * 64) = v;
            }
            if constexpr (LA > 0) {
                T *sdst = s_strip + (t & 1) * (LA * 64) + r_l * 64 + c_l;
#pragma unroll
                for (int i = 0; i < SP; ++i) *reinterpret_cast<tvec_t *>(sdst + i * RPP * 64) = sp[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *tiles = t + 1;
            request_tile(t + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (wid == 1 || wid == 2) {
        // ------------------------------------------------------------------ update waves: a column half each
        if constexpr (MR) {
            constexpr int LPR = 64 / RPI;                        // lanes per row: 16 (k = 128), 32 (k = 256)
            constexpr int NI = K / RPI;                          // requests per sweep
            constexpr int RR = 32 / RPI;                         // requests in flight (32 rows, as in the generic path)
            constexpr int CQ = 8 / RPI;                          // requests per chunk of 8 coordinates
            const int uh = wid - 1;
            const int rsub = lane / LPR;                         // this lane's row inside a request
            const int cb = uh * (K / 2) + (lane % LPR) * 4;      // its four columns
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef float f2v __attribute__((ext_vector_type(2)));
            const float *const mine0 = Q + (int64_t)rsub * K + cb;   // this lane's slice of request 0
            const float *mine = mine0;
            f4v ring[RR];
#pragma unroll
            for (int j = 0; j < RR; ++j) ring[j] = *reinterpret_cast<const f4v *>(mine + (int64_t)(j * RPI) * K);
            {
                int64_t opaque = 0;
                asm volatile("" : "+s"(opaque));
                mine = mine0 + opaque;
            }
            f2v H01 = {0.f, 0.f}, H23 = {0.f, 0.f};              // partial sums of this lane's four entries of H
            if (a.H0) {
                if (rsub == 0) {                                 // (the first group of lanes carries H0, the others zero)
                    const f4v h = *reinterpret_cast<const f4v *>(a.H0 + (int64_t)smp * k + cb);
                    H01 = {h[0], h[1]};
                    H23 = {h[2], h[3]};
                }
            } else {
                // H0 = Q w: the coefficient of a lane's row comes from LDS (the chain wave stores its w there first)
                lds_vi32 *wflag = (lds_vi32 *)&s_cnt[10];
                int wf = *wflag;
                while (__builtin_amdgcn_readfirstlane(wf) == 0) wf = *wflag;
                asm volatile("" ::: "memory");
                lds_vT *sw = (lds_vT *)s_w + rsub;
                float wq[8], wqn[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wq[q] = sw[q * RPI];
                static_for<NI / 8>([&](auto GG) {
                    constexpr int g8 = decltype(GG)::value * 8;
                    if constexpr (g8 + 8 < NI) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) wqn[q] = sw[(g8 + 8 + q) * RPI];
                    }
                    static_for<8>([&](auto QQ) {
                        constexpr int jj = g8 + decltype(QQ)::value;
                        constexpr int j = jj % RR;
                        const f2v w2 = {wq[jj - g8], wq[jj - g8]};
                        const f2v r01 = {ring[j][0], ring[j][1]}, r23 = {ring[j][2], ring[j][3]};
                        H01 = __builtin_elementwise_fma(w2, r01, H01);
                        H23 = __builtin_elementwise_fma(w2, r23, H23);
                        asm volatile("" : "+v"(H01), "+v"(H23));
                        __builtin_amdgcn_sched_barrier(0);
                        ring[j] = *reinterpret_cast<const f4v *>(mine + (int64_t)(((jj + RR) % NI) * RPI) * K);
                        __builtin_amdgcn_sched_barrier(0);
                    });
#pragma unroll
                    for (int q = 0; q < 8; ++q) wq[q] = wqn[q];
                });
            }
            // this lane's four entries of H: the sum of the RPI partial sums, in a fixed order (every group gets it)
            auto total = [&](f2v x) -> f2v {
                f2v r;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v = x[e], lo, hi;
                    if constexpr (RPI == 4) { lane_swap<true>(v, lo, hi); v = lo + hi; }
                    lane_swap<false>(v, lo, hi);
                    r[e] = lo + hi;
                }
                return r;
            };
            auto publish_to = [&](float *dst, lds_vi32 *cnt, int version) {
                const f2v t01 = total(H01), t23 = total(H23);
                if (rsub == 0) *reinterpret_cast<f4v *>(dst + cb) = f4v{t01[0], t01[1], t23[0], t23[1]};
                asm volatile("" ::: "memory");
                cnt[uh] = version;
            };
            publish_to(s_H, ver, 1);
            if (uh == 0) MODL_STAMP(512);
            int ready = 0;
            // the pairs of this lane's rows of a chunk of 8 coordinates (the chunk after the one being applied is requested
            // before its steps start; the counter first: the pairs are good if the counter that came with them was)
            struct Pref { float pn[CQ], po[CQ]; int cnt; };
            auto prefetch = [&](int dbo_, int c, Pref &P) {
                P.cnt = *prog;
                lds_vT *pn_ = wns + dbo_ + c + rsub, *po_ = wos + dbo_ + c + rsub;
#pragma unroll
                for (int q = 0; q < CQ; ++q) { P.pn[q] = pn_[q * RPI]; P.po[q] = po_[q * RPI]; }
            };
            Pref cur;
            prefetch(0, 0, cur);
            for (int sw = 0;; ++sw) {
                {
                    int64_t opaque = 0;                          // (see the generic path: keeps the row loads inside the sweep)
                    asm volatile("" : "+s"(opaque));
                    mine = mine0 + opaque;
                }
                static_for<NB>([&](auto BB) {
                    constexpr int bb = decltype(BB)::value;
                    const int t = sw * NB + bb, base = 64 * t;
                    const int dbo = (t & 1) * 64;
                    if (uh == 0) MODL_STAMP(512);
                    static_for<8>([&](auto CC) {
                        constexpr int c8 = decltype(CC)::value * 8;
                        constexpr int js = (bb * 64 + c8) / RPI;                 // first request of the chunk, in the sweep
                        if constexpr (c8 == 32) { if (uh == 0) MODL_STAMP(512); }
                        Pref nxt;
                        if constexpr (c8 < 56) prefetch(dbo, c8 + 8, nxt);
                        else prefetch(dbo ^ 64, 0, nxt);
                        ready = cur.cnt;
                        if (__builtin_amdgcn_readfirstlane(ready) < base + c8 + 8) {
                            ready = spin_until(ready, s_cnt, base + c8 + 8);
                            prefetch(dbo, c8, cur);
                        }
                        if constexpr (c8 == 0) { if (__builtin_amdgcn_readfirstlane(ready) == kStop) asm volatile("s_endpgm"); }
                        static_for<CQ>([&](auto QQ) {
                            constexpr int q = decltype(QQ)::value;
                            constexpr int j = (js + q) % RR;
                            // H <- fma(w_new, Q_i, fma(-w_old, Q_i, H)) on this lane's row (dict_fact_fast.pyx:361-365, :375-378)
                            const f2v mo = {-cur.po[q], -cur.po[q]}, pn2 = {cur.pn[q], cur.pn[q]};
                            const f2v r01 = {ring[j][0], ring[j][1]}, r23 = {ring[j][2], ring[j][3]};
                            H01 = __builtin_elementwise_fma(pn2, r01, __builtin_elementwise_fma(mo, r01, H01));
                            H23 = __builtin_elementwise_fma(pn2, r23, __builtin_elementwise_fma(mo, r23, H23));
                            asm volatile("" : "+v"(H01), "+v"(H23));
                            __builtin_amdgcn_sched_barrier(0);
                            ring[j] = *reinterpret_cast<const f4v *>(mine + (int64_t)(((js + q + RR) % NI) * RPI) * K);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                        cur = nxt;
                        if constexpr (c8 + 8 == 64 - LA) publish_to(s_Hs, sver, t + 2);
                    });
                    publish_to(s_H, ver, t + 2);
                    if (uh == 0) MODL_STAMP(512);
                });
            }
        }
        if constexpr (!MR) {
        constexpr int KU = KPL / 2;                              // elements per lane (NB >= 2)
        constexpr int R = (KU * (int)sizeof(T) <= 16) ? 32 : 16; // Gram rows in flight (a row comes from L2 in ~600 ns)
        const int uh = wid - 1;
        const int e0 = uh * (K / 2) + lane * KU;                 // this lane's elements e0 .. e0 + KU - 1
        T H[KU], w[KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) {                          // (all of w: H0 = Q w broadcasts every coefficient)
            const int e = lane * KPL + c;
            const T wv = wptr[e < k ? e : 0];
            w[c] = e < k ? wv : (T)0;
        }
#pragma unroll
        for (int c = 0; c < KU; ++c) H[c] = 0;
        constexpr int V = (KU * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : KU;   // elements per load
        typedef T vec_t __attribute__((ext_vector_type(V)));
        T ring[R][KU];
        auto request = [&](T (&dst)[KU], const T *p) {
            if constexpr (V == 1) {
                dst[0] = *p;
            } else {
                const vec_t *rp = reinterpret_cast<const vec_t *>(p);
#pragma unroll
                for (int v = 0; v < KU / V; ++v) {
                    const vec_t x = rp[v];
#pragma unroll
                    for (int c = 0; c < V; ++c) dst[v * V + c] = x[c];
                }
            }
        };
        const T *const mine0 = Q + e0;                // this lane's slice of row 0
        const T *mine = mine0;
#pragma unroll
        for (int j = 0; j < R; ++j) request(ring[j], mine + (int64_t)(j % kc) * K);
        {
            int64_t opaque = 0;
            asm volatile("" : "+s"(opaque));
            mine = mine0 + opaque;
        }
        int nrow = R % kc;                            // the row the next request fetches (rows cycle in sweep order)
        if (a.H0) {
            const T *hp = a.H0 + (int64_t)smp * k;
#pragma unroll
            for (int c = 0; c < KU; ++c) {
                const T hv = hp[e0 + c < k ? e0 + c : 0];
                H[c] = (e0 + c < k) ? hv : (T)0;
            }
        } else {
            // H = Q w as a combination of rows, ascending (cd_kernel's order: same bits).  FULL: no loop - a back edge
            // with row loads in flight makes the compiler wait for ALL of them (the values live in loop-carried
            // registers it may have to copy): one full memory round trip per iteration, measured.
            if constexpr (FULL) {
                static_for<K>([&](auto JJ) {
                    constexpr int jj = decltype(JJ)::value;
                    constexpr int j = jj % R;
                    const T wj = bcast_lane(w[jj % KPL], jj / KPL);
#pragma unroll
                    for (int c2 = 0; c2 < KU; ++c2) {
                        H[c2] = fma(wj, ring[j][c2], H[c2]);
                        asm volatile("" : "+v"(H[c2]));          // (pinned: see `step`)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    request(ring[j], mine + (int64_t)((jj + R) % K) * K);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                for (int j0 = 0; j0 < kc; j0 += R) {
                    const T *next = mine + (int64_t)nrow * K;
                    static_for<R>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        const T wj = bcast_lane(w[j % KPL], (j0 / KPL + j / KPL) & 63);
#pragma unroll
                        for (int c2 = 0; c2 < KU; ++c2) H[c2] = fma(wj, ring[j][c2], H[c2]);
                        __builtin_amdgcn_sched_barrier(0);
                        request(ring[j], next + (int64_t)j * K);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    nrow += R;
                    if (nrow >= kc) nrow -= kc;
                }
            }
        }
        typedef T hvec_t __attribute__((ext_vector_type(KU)));
        auto publish_H = [&](int version) {
            if constexpr (KU == 1) {
                s_H[e0] = H[0];
            } else {
                hvec_t hv;
#pragma unroll
                for (int c = 0; c < KU; ++c) hv[c] = H[c];
                *reinterpret_cast<hvec_t *>(&s_H[e0]) = hv;
            }
            asm volatile("" ::: "memory");
            ver[uh] = version;
        };
        auto publish_snap = [&](int version) {        // (look-ahead hand-off: H after the first 64 - LA coordinates of a block)
            if constexpr (KU == 1) {
                s_Hs[e0] = H[0];
            } else {
                hvec_t hv;
#pragma unroll
                for (int c = 0; c < KU; ++c) hv[c] = H[c];
                *reinterpret_cast<hvec_t *>(&s_Hs[e0]) = hv;
            }
            asm volatile("" ::: "memory");
            sver[uh] = version;
        };
        publish_H(1);
        if (uh == 0) MODL_STAMP(512);
        int ready = 0;                                // cached value of the chain wave's counter
        // one coordinate: H <- fma(w_new, Q_i, fma(-w_old, Q_i, H)) on this wave's entries (dict_fact_fast.pyx:361-365 and
        // :375-378: the reference's two roundings.  ONE fma with the rounded difference w_new - w_old was measured NOISIER -
        // 1.35 x the reference algorithm's own f32 noise on the codes against 0.8 x, scripts/diag_f32_noise.py - although
        // each single operation is more accurate: DESIGN 3.2), then the next request
        auto step = [&](T (&row)[KU], T dn, T dold, const T *nextp) {
            // every result is pinned where it is computed: left alone, the compiler sinks the whole chain of
            // updates to its first use (the block's end) and keeps every row and pair live until then: hundreds of
            // spills.  f32 pairs go through v_pk_fma_f32 (half the issue slots of this wave's busiest loop).
            if constexpr (sizeof(T) == 4 && KU % 2 == 0) {
                typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < KU; r += 2) {
                    f2v h = {H[r], H[r + 1]};
                    const f2v q2 = {row[r], row[r + 1]}, mo = {-dold, -dold}, pn2 = {dn, dn};
                    h = __builtin_elementwise_fma(pn2, q2, __builtin_elementwise_fma(mo, q2, h));
                    asm volatile("" : "+v"(h));
                    H[r] = h[0];
                    H[r + 1] = h[1];
                }
            } else {
#pragma unroll
                for (int r = 0; r < KU; ++r) {
                    H[r] = fma(dn, row[r], fma(-dold, row[r], H[r]));
                    asm volatile("" : "+v"(H[r]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            request(row, nextp);
            __builtin_amdgcn_sched_barrier(0);
        };
        // The chain wave's counter and N published (w_new, w_old) pairs starting at coordinate c of the block, in ONE LDS
        // round trip: the counter is requested first, the LDS executes a wavefront's operations in order and the chain
        // wave stores data before counter, so the pairs are good if the counter that came with them was; repeated until
        // it is.  (Round 3 spun on the counter and only then asked for the pairs: two round trips, ~130 cycles each, in
        // front of every chunk of 8 - the update waves never made up the lag they start a block with.)  One assembly block:
        // the compiler sees straight-line code around it, so its s_waitcnt bookkeeping for the row loads in flight
        // stays exact.
        auto fetch = [&](auto N_, int need, int dbo, int c, T (&pn)[8], T (&po)[8]) {
            constexpr int N = decltype(N_)::value;                // 4 or 8
            const unsigned int a_cnt = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const int *)s_cnt;
            const unsigned int a_n = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const T *)(s_wn + dbo + c);
            const unsigned int a_o = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) const T *)(s_wo + dbo + c);
            int v, sv;
            if constexpr (sizeof(T) == 4) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v n0, n1, o0, o1;
                if constexpr (N == 8) {
                    asm volatile(
                        "modl_fetch%=:\n\t"
                        "ds_read_b32 %0, %6\n\t"
                        "ds_read_b128 %2, %7\n\t"
                        "ds_read_b128 %3, %7 offset:16\n\t"
                        "ds_read_b128 %4, %8\n\t"
                        "ds_read_b128 %5, %8 offset:16\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %1, %0\n\t"
                        "s_cmp_lt_i32 %1, %9\n\t"
                        "s_cbranch_scc1 modl_fetch%="
                        : "=&v"(v), "=&s"(sv), "=&v"(n0), "=&v"(n1), "=&v"(o0), "=&v"(o1)
                        : "v"(a_cnt), "v"(a_n), "v"(a_o), "s"(need) : "memory", "scc");
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pn[i] = n0[i]; pn[4 + i] = n1[i]; po[i] = o0[i]; po[4 + i] = o1[i]; }
                } else {
                    asm volatile(
                        "modl_fetch%=:\n\t"
                        "ds_read_b32 %0, %4\n\t"
                        "ds_read_b128 %2, %5\n\t"
                        "ds_read_b128 %3, %6\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %1, %0\n\t"
                        "s_cmp_lt_i32 %1, %7\n\t"
                        "s_cbranch_scc1 modl_fetch%="
                        : "=&v"(v), "=&s"(sv), "=&v"(n0), "=&v"(o0)
                        : "v"(a_cnt), "v"(a_n), "v"(a_o), "s"(need) : "memory", "scc");
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pn[i] = n0[i]; po[i] = o0[i]; }
                }
            } else {
                typedef double d2v __attribute__((ext_vector_type(2)));
                d2v n0, n1, n2, n3, o0, o1, o2, o3;
                if constexpr (N == 8) {
                    asm volatile(
                        "modl_fetch%=:\n\t"
                        "ds_read_b32 %0, %10\n\t"
                        "ds_read_b128 %2, %11\n\t"
                        "ds_read_b128 %3, %11 offset:16\n\t"
                        "ds_read_b128 %4, %11 offset:32\n\t"
                        "ds_read_b128 %5, %11 offset:48\n\t"
                        "ds_read_b128 %6, %12\n\t"
                        "ds_read_b128 %7, %12 offset:16\n\t"
                        "ds_read_b128 %8, %12 offset:32\n\t"
                        "ds_read_b128 %9, %12 offset:48\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %1, %0\n\t"
                        "s_cmp_lt_i32 %1, %13\n\t"
                        "s_cbranch_scc1 modl_fetch%="
                        : "=&v"(v), "=&s"(sv), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                        : "v"(a_cnt), "v"(a_n), "v"(a_o), "s"(need) : "memory", "scc");
                    pn[0] = n0[0]; pn[1] = n0[1]; pn[2] = n1[0]; pn[3] = n1[1]; pn[4] = n2[0]; pn[5] = n2[1]; pn[6] = n3[0]; pn[7] = n3[1];
                    po[0] = o0[0]; po[1] = o0[1]; po[2] = o1[0]; po[3] = o1[1]; po[4] = o2[0]; po[5] = o2[1]; po[6] = o3[0]; po[7] = o3[1];
                } else {
                    asm volatile(
                        "modl_fetch%=:\n\t"
                        "ds_read_b32 %0, %6\n\t"
                        "ds_read_b128 %2, %7\n\t"
                        "ds_read_b128 %3, %7 offset:16\n\t"
                        "ds_read_b128 %4, %8\n\t"
                        "ds_read_b128 %5, %8 offset:16\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %1, %0\n\t"
                        "s_cmp_lt_i32 %1, %9\n\t"
                        "s_cbranch_scc1 modl_fetch%="
                        : "=&v"(v), "=&s"(sv), "=&v"(n0), "=&v"(n1), "=&v"(o0), "=&v"(o1)
                        : "v"(a_cnt), "v"(a_n), "v"(a_o), "s"(need) : "memory", "scc");
                    pn[0] = n0[0]; pn[1] = n0[1]; pn[2] = n1[0]; pn[3] = n1[1];
                    po[0] = o0[0]; po[1] = o0[1]; po[2] = o1[0]; po[3] = o1[1];
                }
            }
            return v;
        };
        // One chunk of 8 coordinates starting at coordinate c8 of the block (ring slots s0 ..): the chain wave publishes
        // in eights, the last eight of a block as 4 + 4 (`halves`).  The waits are assembly blocks, the ring slots
        // static: straight-line code.  (Publishing a block's last coordinates one by one was measured SLOWER: every
        // single coordinate costs this wave two LDS round trips - counter, then pair - ~300 cycles against the 62 the
        // chain needs for it; the block's H was out 1500 cycles after the chain's last coordinate instead of 1200.
        // Also measured and NOT kept: a look-ahead - the update waves hand over H after a block's first 40 coordinates
        // and the chain wave applies the last 24 steps to the next block's entries itself (a strip of rows fetched by
        // the tile loader; same bits).  The wait only moved: the update waves run ~1100 cycles (18 coordinates) behind,
        // so their snapshot "after 40" is not there when the chain reaches coordinate 40, and the 24 extra steps cost
        // the chain as much as the wait they were meant to remove (block 5200 cycles against 4870).)
        auto chunk = [&](auto S0, int cbase, int c8, int halves, auto &&next_of) {
            constexpr int s0 = decltype(S0)::value;
            T pn[8], po[8];
            const int dbo = (cbase >> 6 & 1) * 64;    // (the pairs of consecutive blocks alternate between two buffers)
            if (halves) {
                ready = fetch(std::integral_constant<int, 4>{}, cbase + c8 + 4, dbo, c8, pn, po);
                static_for<4>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    step(ring[s0 + i], pn[i], po[i], next_of(s0 + i));
                });
                T qn4[8], qo4[8];
                ready = fetch(std::integral_constant<int, 4>{}, cbase + c8 + 8, dbo, c8 + 4, qn4, qo4);
                static_for<4>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    step(ring[s0 + 4 + i], qn4[i], qo4[i], next_of(s0 + 4 + i));
                });
            } else {
                ready = fetch(std::integral_constant<int, 8>{}, cbase + c8 + 8, dbo, c8, pn, po);
                static_for<8>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    step(ring[s0 + i], pn[i], po[i], next_of(s0 + i));
                });
            }
        };
        if constexpr (FULL) {
            // The chunk AFTER the one being applied is requested (counter first, then its pairs: plain LDS reads, the
            // compiler places the wait) before the current chunk's steps start, so the LDS round trip runs under them;
            // only if the counter that came with the pairs falls short - the chain wave has not published that chunk yet -
            // does the wave fall back to the blocking `fetch`.  (Round 4 stamps: with the round trip in front of every
            // chunk the update waves needed 57 cycles per coordinate against the chain wave's 48 - the chain waited.)
            constexpr int VN = (int)(16 / sizeof(T));                 // coefficients per 16-byte LDS read: 4 / 2
            typedef T cvec_t __attribute__((ext_vector_type(VN)));
            typedef __attribute__((address_space(3))) volatile cvec_t lds_cvec;
            struct Pref { T pn[8], po[8]; int cnt; };
            auto prefetch = [&](int dbo_, int c, Pref &P) {
                P.cnt = *prog;
                lds_cvec *pn_ = (lds_cvec *)(s_wn + dbo_ + c), *po_ = (lds_cvec *)(s_wo + dbo_ + c);
#pragma unroll
                for (int v = 0; v < 8 / VN; ++v) {
                    const cvec_t a_ = pn_[v], b_ = po_[v];
#pragma unroll
                    for (int e = 0; e < VN; ++e) { P.pn[v * VN + e] = a_[e]; P.po[v * VN + e] = b_[e]; }
                }
            };
            Pref cur;
            prefetch(0, 0, cur);
            // the whole sweep unrolled (NB blocks of 8 chunks): the only back edge is the sweep's
            for (int sw = 0;; ++sw) {
                // (an opaque pointer per sweep: the matrix is read-only and every sweep reads the same addresses - left
                // alone, the compiler hoists the loads out of the loop and keeps the whole slice in registers: spills)
                {
                    int64_t opaque = 0;                          // (an offset, not the pointer: that would turn the
                    asm volatile("" : "+s"(opaque));             //  loads into flat ones, which also count as LDS traffic)
                    mine = mine0 + opaque;
                }
                static_for<NB>([&](auto BB) {
                    constexpr int bb = decltype(BB)::value;
                    const int t = sw * NB + bb, base = 64 * t;
                    const int dbo = (t & 1) * 64;
                    if (uh == 0) MODL_STAMP(512);
                    static_for<8>([&](auto CC) {
                        constexpr int c8 = decltype(CC)::value * 8;
                        constexpr int cs = bb * 64 + c8;                         // coordinate of the sweep
                        if constexpr (c8 == 32) { if (uh == 0) MODL_STAMP(512); }
                        // (the slot of coordinate c of the sweep is c % R; it is refilled with row (c + R) mod K)
                        auto row_after = [&](int c) { return mine + (int64_t)((c + R) % K) * K; };
                        Pref nxt;
                        if constexpr (c8 < 56) prefetch(dbo, c8 + 8, nxt);
                        else prefetch(dbo ^ 64, 0, nxt);                         // (the next block's first chunk)
                        ready = cur.cnt;
                        if (__builtin_amdgcn_readfirstlane(ready) < base + c8 + 8)
                            ready = fetch(std::integral_constant<int, 8>{}, base + c8 + 8, dbo, c8, cur.pn, cur.po);
                        // the first coordinates of a block that never comes: the chain wave has ended the solve.  (The
                        // wave ends inside the assembly block: no join for the compiler; loads in flight die with it.)
                        if constexpr (c8 == 0) { if (__builtin_amdgcn_readfirstlane(ready) == kStop) asm volatile("s_endpgm"); }
                        static_for<8>([&](auto I) {
                            constexpr int i = decltype(I)::value;
                            step(ring[(cs + i) % R], cur.pn[i], cur.po[i], row_after(cs + i));
                        });
                        cur = nxt;
                        if constexpr (LA > 0 && c8 + 8 == 64 - LA) publish_snap(t + 2);
                    });
                    publish_H(t + 2);
                    if (uh == 0) MODL_STAMP(512);
                });
            }
        } else {
            for (int t = 0;; ++t) {                   // blocks, across sweeps
                const int bsw = t % nblk;
                const int len = (bsw == nblk - 1) ? kc - 64 * (nblk - 1) : 64;
                const int base = 64 * t;
                ready = spin_until(ready, s_cnt, base + 8);
                if (__builtin_amdgcn_readfirstlane(ready) == kStop) asm volatile("s_endpgm");
                for (int h = 0; h < len; h += 32) {
                    if (uh == 0) MODL_STAMP(512);
                    const int fine = (h + 32 >= len) ? 1 : 0;           // the block's last chunk comes as 4 + 4
                    static_for<32 / R>([&](auto GG) {
                        constexpr int gg = decltype(GG)::value;
                        const T *next = mine + (int64_t)nrow * K;
                        auto next_of = [&](int slot) { return next + (int64_t)slot * K; };
                        static_for<R / 8>([&](auto CC) {
                            constexpr int cc = decltype(CC)::value;
                            chunk(std::integral_constant<int, cc * 8>{}, base, h + gg * R + cc * 8,
                                  (gg * R + cc * 8 == 24) ? fine : 0, next_of);
                        });
                        nrow += R;
                        if (nrow >= kc) nrow -= kc;
                    });
                }
                publish_H(t + 2);
                if (uh == 0) MODL_STAMP(512);
            }
        }
        }   // !MR
    }

    // ---------------------------------------------------------------------- chain wave
    // coordinate 64 b + lane in register b; the same vectors once more in cd_kernel's element order for the gap test
    T w[NB], q[NB], inv[NB], wfix[NB], qdg[NB];
    T qe[KPL];
#pragma unroll
    for (int bI = 0; bI < NB; ++bI) {
        const int e = 64 * bI + lane;
        const bool in = e < k;
        const int ec = in ? e : 0;
        const T wv = wptr[ec], qv = qptr[ec], dv = Q[(int64_t)ec * K + ec];
        q[bI] = in ? qv : (T)0;
        const T dg = in ? dv : (T)0;
        qdg[bI] = dg;
        inv[bI] = (dg != (T)0) ? (T)1 / (dg + beta) : (T)0;   // reciprocal of the step denominator (:373); 0 = skipped (:357)
        const bool lv = inv[bI] != (T)0;
        const T w_in = in ? wv : (T)0;
        wfix[bI] = lv ? (T)0 : w_in;                            // a skipped coordinate keeps its value here ...
        w[bI] = lv ? w_in : (T)0;                               // ... and carries 0 in the sweeps
    }
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const int e = lane * KPL + c;
        const T qv = qptr[e < k ? e : 0];
        qe[c] = e < k ? qv : (T)0;
    }
    if constexpr (MR) {                            // (the update waves form H0 = Q w with a coefficient per LANE: from LDS)
        if (!a.H0) {
#pragma unroll
            for (int bI = 0; bI < NB; ++bI) s_w[64 * bI + lane] = w[bI];
            asm volatile("" ::: "memory");
            s_cnt[10] = 1;
        }
    }
    const T y_norm2 = a.xnorm2[smp];
    const T tol_abs = a.tol * y_norm2;             // :336
    const T d_w_tol = a.tol;

    int tblk = 0;                                   // blocks started, across sweeps
    int n_iter = 0;
    int have_tiles = 0;
    T Hb;                                           // H of the current block's coordinates
    {
        // (counter, then data, in one round trip; the data is good if the counter it followed was)
        lds_vT *sH = (lds_vT *)s_H;
        int v0 = ver[0], v1 = ver[1];
        Hb = sH[lane];
        while (__builtin_amdgcn_readfirstlane(v0) < 1 || __builtin_amdgcn_readfirstlane(v1) < 1) {
            v0 = ver[0]; v1 = ver[1];
            Hb = sH[lane];
        }
    }
    // the step formula on the block's 64 lanes at once (:367-373 with tmp = z): returns w_new - w_old
    auto delta_of = [&](T z, T ri, T wold) -> T {
        const T cl = POSITIVE ? (z < alpha ? z : alpha) : clamp3(z, -alpha, alpha);   // as cd_coordinate
        return fma(z - cl, ri, -wold);
    };
    auto wnew_of = [&](T z, T ri) -> T {            // the new coefficient itself (:372-373), the product rounded once
        const T cl = POSITIVE ? (z < alpha ? z : alpha) : clamp3(z, -alpha, alpha);
        return (z - cl) * ri;
    };
    T Zcarry = 0;                                   // look-ahead: what the previous block's last LA steps did to this block
    // (look-ahead) requested under a block's last eight coordinates, for the NEXT block: the first eight rows of its tile
    // (good if the tile loader's counter, requested first, already covered it) - the block boundary then costs one check
    // instead of three dependent LDS round trips (versions + H, tile counter, first rows: ~480 cycles per block, stamps)
    T qpre[8];
    bool pre_ok = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) qpre[i] = 0;
    bool done = false;
    for (; n_iter < a.max_iter && !done; ++n_iter) {
        T w0[NB];
#pragma unroll
        for (int bI = 0; bI < NB; ++bI) w0[bI] = w[bI];
        static_for<NB>([&](auto BI) {
            constexpr int bI = decltype(BI)::value;
            if constexpr (!FULL) { if (bI >= nblk) return; }
            const int len = FULL ? 64 : ((bI == nblk - 1) ? kc - 64 * bI : 64);
            const T wob = w[bI];                    // the block's coefficients before the sweep touches them
            const T rib = inv[bI];
            if (__builtin_amdgcn_readfirstlane(have_tiles) <= tblk) {        // the block's diagonal tile is in LDS
                do { have_tiles = *tiles; } while (__builtin_amdgcn_readfirstlane(have_tiles) <= tblk);   // (the tile loader never stops first)
            }
            MODL_STAMP(0);
            const int base = 64 * tblk;
            const int dbo = (tblk & 1) * 64;
            // Z[m] = q[m] - (H[m] - Q[m][m] w_old[m]): the tmp of :367 of every coordinate of the block, before the
            // steps of the block's own coordinates (H = Q w with the coefficients as they are at the block's start)
            T Z = (fma(qdg[bI], wob, q[bI]) - Hb) + Zcarry;
            T Zla = 0;                              // look-ahead: this block's last LA steps on the NEXT block's entries
            // row L of the tile, this lane's column: Q[64 b + L][64 b + lane] for lane > L, else 0
            // (plain, not volatile, LDS pointers: rows L and L + n of a lane's column are 256 n bytes apart - what
            //  ds_read2st64_b32 fetches in ONE instruction, and an LDS instruction costs the chain wave an issue slot
            //  like any other; the compiler barrier keeps the reads behind the wait for the tile, and after it the
            //  buffer is not written until the chain wave itself releases it)
            typedef __attribute__((address_space(3))) const T lds_cT;
            asm volatile("" ::: "memory");
            lds_cT *tile = (lds_cT *)(s_tile + (unsigned int)(tblk & 1) * TB) + lane;
            lds_cT *strip = (lds_cT *)(s_strip + (LA > 0 ? (tblk & 1) * (LA * 64) : 0)) + lane;
            // groups of 8 coordinates, everything unrolled (the lane of a coordinate is an immediate); the slices of
            // the tile (and of the strip) for the NEXT group are requested before the current one starts
            T qd[8], qn[8], sd[8], sn[8];
            if (pre_ok) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { qd[i] = qpre[i]; sd[i] = 0; sn[i] = 0; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { qd[i] = tile[i * 64]; sd[i] = 0; sn[i] = 0; }
            }
            T dl = 0;
            // (look-ahead) the hand-off of the next block, requested under this block's last group
            const int nb_off = (bI + 1 < nblk) ? 64 * (bI + 1) : 0;
            int spec_t = 0, sv0 = 0, sv1 = 0;
            T Hspec = 0;
            wos[dbo + lane] = wob;                  // (read by the update waves behind the first counter of the block)
            // what the step of coordinate LP (delta in lane LP of dlv) does to the coordinates after it in the block (:361-365
            // and :375-378 as one fused multiply-add with the rounded difference) and, look-ahead, to the next block
            auto apply = [&](auto LP_, T dlv, T rowv, T spv) {
                constexpr int LP = decltype(LP_)::value;
                const T dn = bcast_lane(dlv, LP);
                Z = fma(-dn, rowv, Z);
                if constexpr (LA > 0 && LP >= 64 - LA) {
                    Zla = fma(-dn, spv, Zla);
                    // (pinned to its step: left alone, the compiler sinks the whole chain of LA fused multiply-adds to its
                    //  first use - the block's end - and keeps the LA broadcast deltas and strip rows live until then: 64
                    //  instructions in a row at every block boundary, the ~460 cycles the stamps showed there)
                    asm volatile("" : "+v"(Zla));
                }
            };
            static_for<8>([&](auto GG) {
                constexpr int g = decltype(GG)::value;
                if constexpr (!FULL) { if (g >= 4 && len == 32) return; }   // (a sweep of 64 m + 32 coordinates: short last block)
                if constexpr (g < 7) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) qn[i] = tile[((g + 1) * 8 + i) * 64];
                    if constexpr (LA > 0 && (g + 1) * 8 >= 64 - LA) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) sn[i] = strip[((g + 1) * 8 + i - (64 - LA)) * 64];
                    }
                } else if constexpr (LA > 0) {
                    // counters first, then what they guard (the LDS executes a wavefront's operations in order)
                    spec_t = *tiles;
                    asm volatile("" ::: "memory");               // (the rows stay behind their counter)
                    lds_cT *ntile = (lds_cT *)(s_tile + (unsigned int)((tblk + 1) & 1) * TB) + lane;
#pragma unroll
                    for (int i = 0; i < 8; ++i) qpre[i] = ntile[i * 64];
                    asm volatile("" ::: "memory");
                    sv0 = sver[0]; sv1 = sver[1];
                    Hspec = ((lds_vT *)s_Hs)[nb_off + lane];
                }
                static_for<8>([&](auto II) {
                    constexpr int i = decltype(II)::value;
                    constexpr int L = g * 8 + i;                 // lane of the coordinate
                    // every lane evaluates the step on its own Z: lane L's is this coordinate's (:367-373), the lanes
                    // before it reproduce what their own steps found (their Z has not moved since), the lanes after it
                    // are not there yet
                    dl = delta_of(Z, rib, wob);
                    apply(std::integral_constant<int, L>{}, dl, qd[i], sd[i]);
                    // (!FULL) the block's last eight are published as 4 + 4: the update waves then have four coordinates
                    // left when the chain needs the next block's H
                    if constexpr (!FULL && (g == 7 || g == 3) && i == 3) {
                        if (g == 7 || len == 32) {
                            wns[dbo + lane] = wnew_of(Z, rib);
                            asm volatile("" ::: "memory");
                            *prog = base + L + 1;
                            asm volatile("" ::: "memory");
                        }
                    }
                });
                // publish: every 8 coordinates.  The lanes <= 8 g + 7 are final: their frozen Z reproduces their step, and
                // their new coefficients are re-evaluated from it (what the block's end stores in w)
                wns[dbo + lane] = wnew_of(Z, rib);
                asm volatile("" ::: "memory");
                *prog = base + g * 8 + 8;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) { qd[i] = qn[i]; sd[i] = sn[i]; }
            });
            w[bI] = wnew_of(Z, rib);               // the block's new coefficients, from the frozen Z of every coordinate
            Zcarry = Zla;
            ++tblk;
            *cblk = tblk;
            MODL_STAMP(0);
            // the next block's entries of H (the first block's, after the sweep's last one): complete, or (look-ahead)
            // as they were LA coordinates before the end of the block just finished
            lds_vT *sH = (lds_vT *)(LA > 0 ? s_Hs : s_H);
            lds_vi32 *vv = LA > 0 ? sver : ver;
            if constexpr (LA > 0) {
                Hb = Hspec;
                if (__builtin_amdgcn_readfirstlane(sv0) < tblk + 1 || __builtin_amdgcn_readfirstlane(sv1) < tblk + 1) {
                    int v0 = vv[0], v1 = vv[1];
                    Hb = sH[nb_off + lane];
                    while (__builtin_amdgcn_readfirstlane(v0) < tblk + 1 || __builtin_amdgcn_readfirstlane(v1) < tblk + 1) {
                        v0 = vv[0]; v1 = vv[1];
                        Hb = sH[nb_off + lane];
                    }
                }
                pre_ok = __builtin_amdgcn_readfirstlane(spec_t) > tblk;      // the next block's tile was in LDS when its rows were read
                if (pre_ok) have_tiles = spec_t;
            } else {
                int v0 = vv[0], v1 = vv[1];
                Hb = sH[nb_off + lane];
                while (__builtin_amdgcn_readfirstlane(v0) < tblk + 1 || __builtin_amdgcn_readfirstlane(v1) < tblk + 1) {
                    v0 = vv[0]; v1 = vv[1];
                    Hb = sH[nb_off + lane];
                }
            }
            asm volatile("" ::: "memory");          // (the gap test below reads s_H with plain loads)
        });
        T dmx = 0, wmx = 0;                        // skipped coordinates do not count (:357): their w is 0 here
#pragma unroll
        for (int bI = 0; bI < NB; ++bI) {
            const T d = fabs(w[bI] - w0[bI]), aw = fabs(w[bI]);
            dmx = d > dmx ? d : dmx;
            wmx = aw > wmx ? aw : wmx;
        }
        const T d_w_max = wave_max_nn(dmx), w_max = wave_max_nn(wmx);
        if (w_max == (T)0 || d_w_max / w_max < d_w_tol || n_iter == a.max_iter - 1) {   // :388
            if constexpr (LA > 0) {                 // (look-ahead: nobody has waited for the sweep's complete H yet)
                int v0 = ver[0], v1 = ver[1];
                while (__builtin_amdgcn_readfirstlane(v0) < tblk + 1 || __builtin_amdgcn_readfirstlane(v1) < tblk + 1) {
                    v0 = ver[0]; v1 = ver[1];
                }
                asm volatile("" ::: "memory");
            }
            // cd_kernel's element order (lane l <-> elements KPL l ..): the reductions round as they do there
#pragma unroll
            for (int bI = 0; bI < NB; ++bI) s_w[64 * bI + lane] = w[bI] + wfix[bI];     // one of the two is zero
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            T s_qw = 0, s_wH = 0, s_ww = 0, s_l1 = 0;
            T xmax = POSITIVE ? -INFINITY : (T)0;
            const int e0 = lane * KPL;
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const T wt = s_w[e0 + c], Hc = s_H[e0 + c];
                s_qw += wt * qe[c];
                s_wH += wt * Hc;
                s_ww += wt * wt;
                s_l1 += fabs(wt);
                if (e0 + c < k) {
                    const T x = (qe[c] - Hc) - beta * wt;                             // :397
                    const T mx = POSITIVE ? x : fabs(x);
                    xmax = mx > xmax ? mx : xmax;
                }
            }
            const T q_dot_w = wave_sum(s_qw);
            const T wH = wave_sum(s_wH);
            const T w_norm2 = wave_sum(s_ww);
            const T l1 = wave_sum(s_l1);
            const T dual = wave_max(xmax);
            const double R_norm2 = (double)(y_norm2 + wH) - 2.0 * (double)q_dot_w;      // :404
            double cst;
            T gap;
            if (dual > alpha) {
                cst = (double)(alpha / dual);
                gap = (T)(0.5 * (R_norm2 + R_norm2 * cst * cst));
            } else {
                cst = 1.0;
                gap = (T)R_norm2;
            }
            gap = (T)((double)gap + (((double)(alpha * l1) - cst * (double)y_norm2) + cst * (double)q_dot_w +
                                     ((0.5 * (double)beta) * (1.0 + cst * cst)) * (double)w_norm2));   // :421-423
            if (gap < tol_abs) done = true;                                           // :425
        }
    }
    *stopf = 1;                                     // the loaders leave ...
    asm volatile("" ::: "memory");
    *prog = kStop;                                  // ... and the update wave
    T *w2 = a.code2 ? a.code2 + (a.idx2 ? a.idx2[smp] : (int64_t)smp) * k : nullptr;
#pragma unroll
    for (int bI = 0; bI < NB; ++bI) {
        const int e = 64 * bI + lane;
        if (e < k) {
            wptr[e] = w[bI] + wfix[bI];
            if (w2) w2[e] = w[bI] + wfix[bI];
        }
    }
    if (a.sweeps && lane == 0) a.sweeps[smp] = n_iter;
}

template <typename T, int NB>
void launch_split_nb(hipStream_t stream, const CdArgs<T> &a) {
    dim3 grid((unsigned)a.b), block(256);
    unsigned long long *st = g_cd_stamps.load();
    const bool full = a.k == 64 * NB;
#define MODL_SPLIT_LAUNCH(POS, FULL) hipLaunchKernelGGL((cd_split_kernel<T, NB, POS, FULL>), grid, block, 0, stream, a, st)
    if (full) { if (a.positive) MODL_SPLIT_LAUNCH(true, true); else MODL_SPLIT_LAUNCH(false, true); }
    else { if (a.positive) MODL_SPLIT_LAUNCH(true, false); else MODL_SPLIT_LAUNCH(false, false); }
#undef MODL_SPLIT_LAUNCH
}


// (the kernels are large - a whole sweep unrolled - and are instantiated in four translation units that compile side
// by side: cd_split.hip f32 k <= 256, cd_split_b.hip f32 k <= 1024, cd_split_c.hip f64 k <= 512, cd_split_d.hip f64 k <= 1024)
#define MODL_SPLIT_EXTERN(T, NB) extern template void launch_split_nb<T, NB>(hipStream_t, const CdArgs<T> &)
MODL_SPLIT_EXTERN(float, 2); MODL_SPLIT_EXTERN(float, 4); MODL_SPLIT_EXTERN(float, 8); MODL_SPLIT_EXTERN(float, 16);
MODL_SPLIT_EXTERN(double, 2); MODL_SPLIT_EXTERN(double, 4); MODL_SPLIT_EXTERN(double, 8); MODL_SPLIT_EXTERN(double, 16);
#undef MODL_SPLIT_EXTERN

}  // namespace modl
