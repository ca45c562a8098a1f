// Batched elastic-net code solver: cyclic coordinate descent on a Gram system
// with the duality-gap stopping rule, one sample per wavefront.
//
// Replaces enet_coordinate_descent_gram and its two batch drivers
//   (reference: modl/decomposition/dict_fact_fast.pyx:270-427, 33-113, 125-215).
// The sweep order (0..k-1), the skip rules (Q[ii,ii] == 0, zero coefficients),
// the update formula and both stopping tests are those of the reference, so a
// sample performs the same number of sweeps; only float summation order inside
// the gap evaluation differs (the reference uses BLAS dot/asum there).
//
// gfx950 mapping: the k-vectors H = Q w, w, q and 1 / (diag(Q) + beta) live in
// registers, KPL consecutive coefficients per lane (coefficient e in register
// e % KPL of lane e / KPL, k <= 64 * KPL), so a row of the Gram matrix is ONE
// 16-byte load per lane at k = 256 (f32).  Two kinds of sweep, chosen per sweep
// from the number of active coordinates:
//   * dense: the update formula of a coordinate is evaluated by EVERY lane on its
//     own coefficient (the values of the other lanes are discarded), so no scalar
//     has to be fetched before the arithmetic; the only cross-lane traffic is the
//     pair (w_new, w_old) of the owning lane, read with v_readlane after it, then
//     the k-wide update H <- fma(w_new, Q_ii, fma(-w_old, Q_ii, H)) (v_pk_fma_f32).
//     Rows are prefetched through a ring of 8 row buffers;
//   * sparse (active set): a coordinate with w_ii == 0 whose update stays 0
//     (|q_ii - H_ii| <= alpha) is a no-op of the reference's sweep (nothing is
//     written, d_w_max / w_max are unaffected), and that test is evaluated for ALL
//     coordinates at once (one compare + ballot per register).  The ordered list of
//     the active coordinates is built once per sweep (prefix counts of the ballots,
//     through LDS), their Gram rows are prefetched through a ring, and a coordinate
//     that a step ACTIVATES ahead of the cursor is picked up exactly (the list is
//     rebuilt from the cursor).  With l1-penalised codes most coordinates are
//     inactive after the first sweeps: 0.23 us per active coordinate against
//     14 us for a dense sweep at k = 256.
// Both produce the reference's iterates bit for bit.  The five reductions of the
// gap test are wave shuffles.  b independent problems -> b wavefronts, 4 per
// workgroup.
#include "kernels.hpp"
#include "cd_common.hpp"
#include <algorithm>
#include <atomic>
#ifndef MODL_CD_SPARSE_RING
#define MODL_CD_SPARSE_RING 4          /* (tuning) ring depth of the sparse sweep for k <= 256 */
#endif
#include <utility>

namespace modl {

// Row loader: coefficients lane * KPL .. lane * KPL + KPL - 1 of row ii in r[].  VEC (k == 64 * KPL, 16-byte
// aligned rows): 16-byte loads; otherwise element loads with a clamped address and a select — no branches
// either way, so all requests of a group stay in flight together.
template <typename T, int KPL, bool VEC>
__device__ __forceinline__ void load_row(const T *__restrict__ Q, int ii, int k, int lane, T (&r)[KPL]) {
    const unsigned int base = (unsigned int)ii * (unsigned int)k;      // k <= 1024: 32-bit element offsets
    if constexpr (VEC) {
        constexpr int V = (KPL * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : KPL;   // elements per load
        typedef T vec_t __attribute__((ext_vector_type(V)));
        const vec_t *rp = reinterpret_cast<const vec_t *>((Q + lane * KPL) + base);
#pragma unroll
        for (int v = 0; v < KPL / V; ++v) {
            const vec_t x = rp[v];
#pragma unroll
            for (int c = 0; c < V; ++c) r[v * V + c] = x[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = lane * KPL + c;
            const T v = Q[base + (unsigned int)(e < k ? e : k - 1)];
            r[c] = e < k ? v : (T)0;
        }
    }
}

// Gram rows in flight in the dense sweep.  A row comes from L2 (the 256 KB matrix does not fit a compute unit's L1):
// ~600 ns, i.e. more than 8 coordinates of ~45 ns - with 8 rows in flight the wave stalled on every row (30 % of
// its cycles at s_waitcnt); 16 cover the latency (code solve 90.5 -> 84.5 us at the metric's shape), 32 buy nothing.
constexpr int kCdRing = 16;

// One coordinate ii = L * KPL + C of a dense sweep.  A zero diagonal skips the coordinate (:357): its
// working coefficient is 0 and so is its step (inv = 0), both multipliers vanish and H is unchanged bit
// for bit.
template <typename T, int KPL, int C, bool POSITIVE>
__device__ __forceinline__ void cd_coord(int L, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL], const T (&inv)[KPL],
                                         const T (&row)[KPL], T alpha) {
    const T wv = w[C];
    const T xv = cd_coordinate<T, POSITIVE>(H[C], wv, q[C], inv[C], row[C], alpha);
    const T dn = bcast_lane(xv, L), dold = bcast_lane(wv, L);
#pragma unroll
    for (int r = 0; r < KPL; ++r) H[r] = fma(dn, row[r], fma(-dold, row[r], H[r]));   // :361-365, :375-378
    w[C] = __builtin_amdgcn_inverse_ballot_w64(1ull << L) ? xv : wv;   // lane L only (scalar mask)
}

// A dense sweep: groups of U = max(8, KPL) coordinates, straight-line code with static register indices;
// row ii + 8 is requested as soon as row ii has been consumed (ring of 8 buffers).  The scheduler barriers
// keep every request where it is: otherwise the scheduler gathers the group's loads at the end of the loop
// body and the wait for a row sits right behind its own request.
// PAD: at least kCdRing rows are readable behind the matrix, so the prefetch needs no clamp and the row
// address is ONE 64-bit add per coordinate on a per-lane pointer (instead of min / mul / add on the scalar
// unit plus the lane offset: the wave spends 2/3 of its cycles issuing instructions, every one counts).
template <typename T, int KPL, bool VEC, bool POSITIVE, bool PAD>
__device__ __forceinline__ void cd_dense_sweep(int lane, int kq, int k, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                               const T (&inv)[KPL], const T *__restrict__ Q, T alpha) {
    // kq: row stride of Q; k: coordinates to visit (VEC: a multiple of U, the rows up to it exist - zero rows of a
    // padded Gram are dead coordinates, inv = 0)
    constexpr int U = KPL > kCdRing ? KPL : kCdRing;
    T ring[kCdRing][KPL];
#pragma unroll
    for (int j = 0; j < kCdRing; ++j) load_row<T, KPL, VEC>(Q, j < k ? j : k - 1, kq, lane, ring[j]);
    const int groups = k / U;
    if constexpr (PAD && VEC) {
        constexpr int V = (KPL * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : KPL;
        typedef T vec_t __attribute__((ext_vector_type(V)));
        constexpr int K = 64 * KPL;                                   // VEC: k == 64 * KPL, the row stride is a constant
        const T *next = Q + (int64_t)kCdRing * K + lane * KPL;       // this lane's slice of row ii + kCdRing
        // (the compiler waits for ALL outstanding rows at the top of every iteration of this loop - s_waitcnt vmcnt(0)
        // behind the back edge, i.e. also for the row requested one coordinate earlier; unrolling the K / U groups
        // gives exact waits but 25 KB of straight-line code per sweep that ran 45 % SLOWER, measured)
        for (int g = 0; g < groups; ++g) {
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                cd_coord<T, KPL, j % KPL, POSITIVE>(g * (U / KPL) + j / KPL, w, H, q, inv, ring[j % kCdRing], alpha);
                __builtin_amdgcn_sched_barrier(0);
                const vec_t *rp = reinterpret_cast<const vec_t *>(next + j * K);   // immediate offsets within the group
#pragma unroll
                for (int v = 0; v < KPL / V; ++v) {
                    const vec_t x = rp[v];
#pragma unroll
                    for (int c = 0; c < V; ++c) ring[j % kCdRing][v * V + c] = x[c];
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            next += U * K;
        }
        return;
    }
    for (int g = 0; g < groups; ++g) {
        static_for<U>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int ii = g * U + j;
            cd_coord<T, KPL, j % KPL, POSITIVE>(g * (U / KPL) + j / KPL, w, H, q, inv, ring[j % kCdRing], alpha);
            __builtin_amdgcn_sched_barrier(0);
            const int nx = (ii + kCdRing < k) ? ii + kCdRing : k - 1;
            load_row<T, KPL, VEC>(Q, nx, kq, lane, ring[j % kCdRing]);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    if constexpr (!VEC) {                          // ragged end: fewer than U coordinates, rows (clamped) are in the ring
        const int done = groups * U, tail = k - done;
        static_for<U - 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if (j < tail) {
                cd_coord<T, KPL, j % KPL, POSITIVE>((done + j) / KPL, w, H, q, inv, ring[j % kCdRing], alpha);
                if (j + kCdRing < tail) load_row<T, KPL, VEC>(Q, done + j + kCdRing, kq, lane, ring[j % kCdRing]);
            }
        });
    }
}

// Active coordinates as one 64-bit mask per register (bit L of m[c] = coordinate L * KPL + c): live and
// (w != 0 or the update leaves zero).
template <typename T, int KPL, bool POSITIVE>
__device__ __forceinline__ void cd_active(const T (&w)[KPL], const T (&H)[KPL], const T (&q)[KPL], const T (&inv)[KPL],
                                          T alpha, unsigned long long (&m)[KPL]) {
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const T tmp = q[r] - H[r];                            // w == 0 there: H[ii] needs no correction
        bool act = (fabs(tmp) - alpha) > (T)0;
        if (POSITIVE) act = act && !(tmp < (T)0);
        act = (act || w[r] != (T)0) && inv[r] != (T)0;
        m[r] = __ballot(act);
    }
}
// first active coordinate >= pos, or a value >= 64 * KPL
template <int KPL>
__device__ __forceinline__ int cd_next(const unsigned long long (&m)[KPL], int pos) {
    int res = 64 * KPL;
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const int lo = (pos - c + KPL - 1) / KPL;             // first lane whose coordinate lane * KPL + c is >= pos
        const unsigned long long mm = (lo <= 0) ? m[c] : (lo >= 64 ? 0ull : (m[c] & (~0ull << lo)));
        const int cand = mm ? __builtin_ctzll(mm) * KPL + c : 64 * KPL;
        res = cand < res ? cand : res;
    }
    return res;
}

// one active coordinate ii = L * KPL + C with its Gram row in `row`
template <typename T, int KPL, int C, bool POSITIVE>
__device__ __forceinline__ void cd_step(int L, int lane, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                        const T (&inv)[KPL], const T (&row)[KPL], T alpha) {
    const T h = bcast_lane(H[C], L), wo = bcast_lane(w[C], L), qq = bcast_lane(q[C], L), ri = bcast_lane(inv[C], L);
    const T Qcc = bcast_lane(row[C], L);
    const T x = cd_coordinate<T, POSITIVE>(h, wo, qq, ri, Qcc, alpha);
#pragma unroll
    for (int r = 0; r < KPL; ++r) H[r] = fma(x, row[r], fma(-wo, row[r], H[r]));
    if (lane == L) w[C] = x;
}

// One coordinate of a sparse sweep; the register index is resolved by a wave-uniform switch whose cases only form the
// candidate, broadcast (w_new, w_old) and store the coefficient -- the k-wide update of H is common code behind it (a
// switch around the whole step makes every case end in a dozen register moves).  !valid: a null step (both
// multipliers 0, no lane selected), so that the unrolled ring needs no exit in the middle of a group.
template <typename T, int KPL, bool POSITIVE>
__device__ __forceinline__ void cd_coord_any(unsigned int ii, bool valid, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                             const T (&inv)[KPL], const T (&row)[KPL], T alpha) {
    const int L = (int)(ii / KPL);
    const unsigned long long lm = (unsigned long long)valid << L;
    T dn = 0, dold = 0;
    switch (ii % KPL) {
#define MODL_CD_CASE(C)                                                                                  \
    case C:                                                                                              \
        if constexpr (C < KPL) {                                                                         \
            const T wv = w[C];                                                                           \
            const T xv = cd_coordinate<T, POSITIVE>(H[C], wv, q[C], inv[C], row[C], alpha);              \
            dn = bcast_lane(xv, L);                                                                      \
            dold = bcast_lane(wv, L);                                                                    \
            w[C] = __builtin_amdgcn_inverse_ballot_w64(lm) ? xv : wv;                                    \
        }                                                                                                \
        break;
        MODL_CD_CASE(0) MODL_CD_CASE(1) MODL_CD_CASE(2) MODL_CD_CASE(3) MODL_CD_CASE(4) MODL_CD_CASE(5)
        MODL_CD_CASE(6) MODL_CD_CASE(7) MODL_CD_CASE(8) MODL_CD_CASE(9) MODL_CD_CASE(10) MODL_CD_CASE(11)
        MODL_CD_CASE(12) MODL_CD_CASE(13) MODL_CD_CASE(14) MODL_CD_CASE(15)
#undef MODL_CD_CASE
    }
    if (!valid) { dn = 0; dold = 0; }
#pragma unroll
    for (int r = 0; r < KPL; ++r) H[r] = fma(dn, row[r], fma(-dold, row[r], H[r]));   // as cd_coord
}

// lo[c] = the bits of the coordinates <= ii (coordinate L * KPL + c is bit L of mask c); ii may be -1
template <int KPL>
__device__ __forceinline__ void cd_low_masks(int ii, unsigned long long (&lo)[KPL]) {
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const int cnt = ii >= c ? (ii - c) / KPL + 1 : 0;     // lanes L with L * KPL + c <= ii
        lo[c] = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
    }
}

// Gram rows of the next active coordinates in flight (register budget: R * KPL values)
template <int KPL> constexpr int cd_sparse_ring() { return KPL <= 4 ? MODL_CD_SPARSE_RING : (KPL == 8 ? 4 : 2); }
constexpr int kCdSparseRingMax = MODL_CD_SPARSE_RING > 4 ? MODL_CD_SPARSE_RING : 4;

// A sparse sweep.  At one wavefront per SIMD the sweep is bound by its INSTRUCTION COUNT (an instruction issues every
// ~5 cycles), so the per-coordinate work is kept minimal: the ordered list of the active coordinates is built ONCE
// (prefix counts of the masks, through `list` in LDS), the rows of the next R list entries are in flight in
// a ring with static slots (the loop is unrolled over the ring: a register copy would wait for the load it moves), and
// after each step the only question asked is "did this step activate a coordinate that is not known yet?" -- one
// compare + ballot + andn2 per register against `mq` (known active, or dead, or behind the cursor).  If so, the exact
// masks are re-evaluated; an activation AHEAD of the cursor rebuilds the list from the cursor (rare after the first
// sweeps), one behind it belongs to the next sweep.  A listed coordinate that went inactive meanwhile is a no-op step
// (w == 0 there and the update stays 0: H and w unchanged).  So the coordinates that change anything are visited in
// increasing order with the H of the moment, exactly as in the reference's sweep.
template <typename T, int KPL, bool VEC, bool POSITIVE>
__device__ __forceinline__ int cd_sparse_sweep(int lane, int kq, int k, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                               const T (&inv)[KPL], unsigned long long (&m)[KPL],
                                               const T *__restrict__ Q, T alpha, unsigned short *list) {
    constexpr int R = cd_sparse_ring<KPL>(), NONE = 64 * KPL;
    int start = 0;                                            // coordinates >= start remain to be visited
    for (int rebuilds = 0;; ++rebuilds) {                     // returns the number of list rebuilds
        unsigned long long lo[KPL], mq[KPL];
        cd_low_masks<KPL>(start - 1, lo);
        int n = 0, pos = 0;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            m[c] &= ~lo[c];
            n += __builtin_popcountll(m[c]);
            pos += __builtin_amdgcn_mbcnt_hi((unsigned int)(m[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m[c], 0u));
            mq[c] = m[c] | lo[c] | __ballot(inv[c] == (T)0);   // listed, behind the cursor, or dead (never turns active)
        }
        if (n == 0) return rebuilds;
#pragma unroll
        for (int c = 0; c < KPL; ++c)
            if ((m[c] >> lane) & 1ull) list[pos++] = (unsigned short)(lane * KPL + c);
        if (lane < 2 * R) list[n + lane] = (unsigned short)NONE;
        __builtin_amdgcn_wave_barrier();                      // LDS operations of one wavefront complete in order

        T ring[R][KPL];
        int cq[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            cq[j] = __builtin_amdgcn_readfirstlane((int)list[j]);
            load_row<T, KPL, VEC>(Q, cq[j] < k ? cq[j] : k - 1, kq, lane, ring[j]);   // unconditional: no branch around a load
        }
        const unsigned short *lp = list + R;
        unsigned int pending = *lp;                           // list entry of the next refill, read one step ahead
        bool stop = false;                                    // an activation ahead of the cursor: rebuild from `start`
        while (!stop && cq[0] < k) {                          // at the top of a group slot 0 holds the oldest entry
            static_for<R>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const bool valid = !stop && cq[s] < k;
                const int ii = valid ? cq[s] : 0;
                cd_coord_any<T, KPL, POSITIVE>((unsigned int)ii, valid, w, H, q, inv, ring[s], alpha);
                unsigned long long nb = 0;
#pragma unroll
                for (int c = 0; c < KPL; ++c) {               // a superset of cd_active's "turns non-zero" test
                    const T tmp = q[c] - H[c];
                    nb |= __ballot(POSITIVE ? tmp > alpha : fabs(tmp) > alpha) & ~mq[c];
                }
                if (valid && nb != 0ull) {
                    cd_active<T, KPL, POSITIVE>(w, H, q, inv, alpha, m);
                    cd_low_masks<KPL>(ii, lo);
                    unsigned long long ahead = 0;
#pragma unroll
                    for (int c = 0; c < KPL; ++c) ahead |= m[c] & ~lo[c] & ~mq[c];
                    if (ahead != 0ull) {
                        start = ii + 1;
                        stop = true;
                    } else {
#pragma unroll
                        for (int c = 0; c < KPL; ++c) mq[c] |= m[c] | lo[c];
                    }
                }
                cq[s] = __builtin_amdgcn_readfirstlane((int)pending);
                ++lp;
                pending = *lp;
                load_row<T, KPL, VEC>(Q, cq[s] < k ? cq[s] : k - 1, kq, lane, ring[s]);
            });
        }
        if (!stop) return rebuilds;
    }
}

template <typename T, int KPL, bool VEC, bool POSITIVE, bool PAD>
__global__ __launch_bounds__(256) void cd_kernel(CdArgs<T> a) {
    __shared__ unsigned short s_list[4][64 * KPL + 2 * kCdSparseRingMax];   // active coordinates of a sparse sweep, per wave
    const int lane = threadIdx.x & 63;
    const int smp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (smp >= a.b) return;                        // whole wave exits together
    const int k = a.k;
    const int kq = a.ldg ? a.ldg : k;              // row stride of the Gram matrix
    constexpr int UU = KPL > kCdRing ? KPL : kCdRing;
    const int kc = VEC ? (k + UU - 1) / UU * UU : k;   // coordinates a dense pass visits (VEC: the padding is dead)
    const T *__restrict__ Q = a.G + (a.g_idx ? a.g_idx[smp] : (int64_t)smp) * a.g_stride;
    const int64_t row_out = a.idx ? a.idx[smp] : (int64_t)smp;
    T *wptr = a.code + row_out * k;
    const T *qptr = a.Dx + (int64_t)smp * k;
    const T alpha = a.alpha, beta = a.beta;
    constexpr bool positive = POSITIVE;
    const int e0 = lane * KPL;

    T w[KPL], H[KPL], q[KPL], inv[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool in = e0 + c < k;
        const int e = in ? e0 + c : 0;
        const T wv = wptr[e], qv = qptr[e], dv = Q[(int64_t)e * kq + e];
        w[c] = in ? wv : (T)0;
        q[c] = in ? qv : (T)0;
        const T dg = in ? dv : (T)0;
        inv[c] = (dg != (T)0) ? (T)1 / (dg + beta) : (T)0;   // reciprocal of the step denominator (:373); 0 = skipped
        H[c] = 0;
    }
    const T y_norm2 = a.xnorm2[smp];
    const T tol_abs = a.tol * y_norm2;             // :336
    const T d_w_tol = a.tol;

    if (a.H0) {
        const T *hp = a.H0 + (int64_t)smp * k;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const T hv = hp[e0 + c < k ? e0 + c : 0];
            H[c] = (e0 + c < k) ? hv : (T)0;
        }
    } else {
        // H = Q w as a combination of rows (Q is symmetric, as the solver itself assumes), rows prefetched
        // through the same ring as the sweeps: cheaper than a separate product launch for one minibatch
        constexpr int U = KPL > kCdRing ? KPL : kCdRing;
        T ring[kCdRing][KPL];
#pragma unroll
        for (int j = 0; j < kCdRing; ++j) load_row<T, KPL, VEC>(Q, j < k ? j : k - 1, kq, lane, ring[j]);
        if constexpr (PAD && VEC) {
            constexpr int V = (KPL * sizeof(T) >= 16) ? (int)(16 / sizeof(T)) : KPL;
            typedef T vec_t __attribute__((ext_vector_type(V)));
            const T *next = Q + (int64_t)kCdRing * kq + lane * KPL;
            for (int j0 = 0; j0 < kc; j0 += U) {
                static_for<U>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const T wj = bcast_lane(w[j % KPL], (j0 / KPL + j / KPL) & 63);
#pragma unroll
                    for (int c2 = 0; c2 < KPL; ++c2) H[c2] = fma(wj, ring[j % kCdRing][c2], H[c2]);
                    __builtin_amdgcn_sched_barrier(0);
                    const vec_t *rp = reinterpret_cast<const vec_t *>(next);
#pragma unroll
                    for (int v = 0; v < KPL / V; ++v) {
                        const vec_t x = rp[v];
#pragma unroll
                        for (int c = 0; c < V; ++c) ring[j % kCdRing][v * V + c] = x[c];
                    }
                    next += kq;
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        } else {
            for (int j0 = 0; j0 < kc; j0 += U) {
                static_for<U>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const int jj = j0 + j;
                    T wj = bcast_lane(w[j % KPL], (j0 / KPL + j / KPL) & 63);   // coefficient jj (zero beyond k)
                    if (!VEC && jj >= k) wj = 0;
#pragma unroll
                    for (int c2 = 0; c2 < KPL; ++c2) H[c2] = fma(wj, ring[j % kCdRing][c2], H[c2]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int nx = (jj + kCdRing < kc) ? jj + kCdRing : kc - 1;
                    load_row<T, KPL, VEC>(Q, nx, kq, lane, ring[j % kCdRing]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
    }

    // Working coefficients: a skipped coordinate (zero diagonal, :357) keeps its value in wfix and carries 0
    // in w, so that the sweeps need no per-coordinate liveness test (its step inv is 0 as well).
    T wfix[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool lv = inv[c] != (T)0;
        wfix[c] = lv ? (T)0 : w[c];
        w[c] = lv ? w[c] : (T)0;
    }

    int n_iter = 0, churn = 0;
    for (; n_iter < a.max_iter; ++n_iter) {
        unsigned long long m[KPL];
        cd_active<T, KPL, POSITIVE>(w, H, q, inv, alpha, m);
        int n_act = 0;
#pragma unroll
        for (int r = 0; r < KPL; ++r) n_act += __builtin_popcountll(m[r]);
        T w0[KPL];                                 // a coefficient changes once per sweep: d_w_ii = |w - w0| (:380-384)
#pragma unroll
        for (int r = 0; r < KPL; ++r) w0[r] = w[r];
        // a sparse step costs about four dense coordinates and a list rebuild (a coordinate activated ahead of the
        // cursor) about eight sparse steps: while the active set keeps churning (e.g. a rank-deficient Gram matrix
        // that never lets the sweeps settle) the sweeps stay dense
        if (100 * (n_act + 8 * churn) > a.sparse_pct * k) {
            cd_dense_sweep<T, KPL, VEC, POSITIVE, PAD>(lane, kq, kc, w, H, q, inv, Q, alpha);
            churn >>= 1;
        } else {
            churn = cd_sparse_sweep<T, KPL, VEC, POSITIVE>(lane, kq, k, w, H, q, inv, m, Q, alpha, s_list[threadIdx.x >> 6]);
        }
        T dmx = 0, wmx = 0;                        // skipped coordinates do not count (:357): their w is 0 here
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const T d = fabs(w[r] - w0[r]), aw = fabs(w[r]);
            dmx = d > dmx ? d : dmx;
            wmx = aw > wmx ? aw : wmx;
        }
        const T d_w_max = wave_max(dmx), w_max = wave_max(wmx);
        if (w_max == (T)0 || d_w_max / w_max < d_w_tol || n_iter == a.max_iter - 1) {   // :388
            T s_qw = 0, s_wH = 0, s_ww = 0, s_l1 = 0;
            T xmax = positive ? -INFINITY : (T)0;
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const T wt = w[c] + wfix[c];                                          // one of the two is zero
                s_qw += wt * q[c];
                s_wH += wt * H[c];
                s_ww += wt * wt;
                s_l1 += fabs(wt);
                if (e0 + c < k) {
                    const T x = (q[c] - H[c]) - beta * wt;                            // :397
                    const T mx = positive ? x : fabs(x);
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
            if (gap < tol_abs) { ++n_iter; break; }                                   // :425
        }
    }
    T *w2 = a.code2 ? a.code2 + (a.idx2 ? a.idx2[smp] : (int64_t)smp) * k : nullptr;
#pragma unroll
    for (int c = 0; c < KPL; ++c)
        if (e0 + c < k) {
            wptr[e0 + c] = w[c] + wfix[c];
            if (w2) w2[e0 + c] = w[c] + wfix[c];
        }
    if (a.sweeps && lane == 0) a.sweeps[smp] = n_iter;
}

template <typename T, int KPL>
static void launch_cd_kpl(hipStream_t stream, const CdArgs<T> &a, dim3 grid, dim3 block) {
    constexpr size_t kRowAlign = (KPL * sizeof(T) >= 16) ? 16 : KPL * sizeof(T);
    const bool vec = ((a.ldg ? a.ldg : a.k) == 64 * KPL) && (reinterpret_cast<uintptr_t>(a.G) % kRowAlign == 0) &&
                     ((a.g_stride * sizeof(T)) % kRowAlign == 0);
    const bool pad = vec && a.g_pad_rows >= kCdRing && a.g_stride == 0;
#define MODL_CD_LAUNCH(VEC, POS, PAD) hipLaunchKernelGGL((cd_kernel<T, KPL, VEC, POS, PAD>), grid, block, 0, stream, a)
    if (pad) {
        if (a.positive) MODL_CD_LAUNCH(true, true, true);
        else MODL_CD_LAUNCH(true, false, true);
    } else if (vec) {
        if (a.positive) MODL_CD_LAUNCH(true, true, false);
        else MODL_CD_LAUNCH(true, false, false);
    } else {
        if (a.positive) MODL_CD_LAUNCH(false, true, false);
        else MODL_CD_LAUNCH(false, false, false);
    }
#undef MODL_CD_LAUNCH
}

// diagnostics (modl_debug_set): >= 0 overrides CdArgs::sparse_pct for every launch of this process
std::atomic<int> g_cd_sparse_pct{-1};
std::atomic<int> g_cd_split{1};
extern std::atomic<unsigned long long *> g_cd_stamps;   // cd_split.hip
extern std::atomic<unsigned long long *> g_atom_stamps; // bcd.hip
extern std::atomic<int> g_bcd_acc;                      // bcd.hip
extern std::atomic<int> g_bcd_tiny;                     // bcd.hip
extern std::atomic<int> g_bcd_persist;                  // bcd.hip
extern std::atomic<int> g_stats_resident;               // somf_step.hip
extern std::atomic<int> g_recsys_fused;                 // recsys.hip
extern std::atomic<int> g_atom_mwg;                     // bcd.hip
extern std::atomic<int> g_bcd_few;                      // bcd.hip
extern std::atomic<int> g_atom_pipe;                    // bcd.hip
extern std::atomic<int> g_stage_ahead;                  // somf_step.hip

template <typename T>
int launch_cd(hipStream_t stream, const CdArgs<T> &a0) {
    if (a0.b <= 0 || a0.k <= 0) return MODL_OK;
    if (a0.k > 1024) return MODL_EINVAL;
    CdArgs<T> a = a0;
    const int sparse_pct = g_cd_sparse_pct.load(std::memory_order_relaxed);   // diagnostics: modl_debug_set
    if (sparse_pct >= 0) a.sparse_pct = sparse_pct;
    // a shared Gram matrix with k >= 32: a workgroup per sample, the chain on one wavefront and the k-wide update on
    // two others (cd_split_impl.hpp); cd_kernel keeps k < 32 and is the reference-order implementation the tests compare
    // the four-wavefront solver with (diagnostics: modl_debug_set(MODL_DEBUG_CD_SPLIT, 0) forces it - for k > 256 only in
    // the diagnostics library: its 8 / 16-coefficients-per-lane variants spill hundreds of registers and are not built
    // into the product)
#ifdef MODL_DIAG
    const bool want_split = g_cd_split.load(std::memory_order_relaxed) != 0;
#else
    const bool want_split = g_cd_split.load(std::memory_order_relaxed) != 0 || a.k > 256;
#endif
    if (want_split && cd_split_applies<T>(a)) return launch_cd_split<T>(stream, a);
    dim3 grid((unsigned)cdiv(a.b, 4)), block(256);
    if (a.k <= 64) launch_cd_kpl<T, 1>(stream, a, grid, block);
    else if (a.k <= 128) launch_cd_kpl<T, 2>(stream, a, grid, block);
    else if (a.k <= 256) launch_cd_kpl<T, 4>(stream, a, grid, block);
#ifdef MODL_DIAG
    else if (a.k <= 512) launch_cd_kpl<T, 8>(stream, a, grid, block);
    else launch_cd_kpl<T, 16>(stream, a, grid, block);
#else
    else return MODL_EINVAL;     // (k > 256 needs a Gram matrix the four-wavefront solver takes: 16-byte aligned, row
                                 //  stride 512 / 1024 - every caller in this library pads and aligns it)
#endif
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

template int launch_cd<float>(hipStream_t, const CdArgs<float> &);
template int launch_cd<double>(hipStream_t, const CdArgs<double> &);

// Any number of coefficients on the vectorised kernel: the element-wise row loader of the general kernel costs 4.6x at
// k = 200 and 40x at k = 320 (385 us / 3.6 ms per minibatch of 256 against 84 us at k = 256, measured), so a shared Gram
// whose size is not 64 / 128 / 256 / 512 / 1024 is copied into a zero-padded matrix of that stride; the padding is dead
// coordinates (zero diagonal), which the sweeps skip bit for bit.
int cd_padded_ld(int k) { return k <= 64 ? 64 : (k <= 128 ? 128 : (k <= 256 ? 256 : (k <= 512 ? 512 : 1024))); }
template <typename T>
__global__ __launch_bounds__(256) void cd_pad_gram_kernel(const T *G, int k, T *Gp, int ldg) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < k; c += 256) Gp[(int64_t)r * ldg + c] = G[(int64_t)r * k + c];
}
template <typename T>
int launch_cd_pad_gram(hipStream_t stream, const T *G, int k, T *Gp, int ldg) {
    hipLaunchKernelGGL((cd_pad_gram_kernel<T>), dim3(k), dim3(256), 0, stream, G, k, Gp, ldg);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_cd_pad_gram<float>(hipStream_t, const float *, int, float *, int);
template int launch_cd_pad_gram<double>(hipStream_t, const double *, int, double *, int);

// ---- a Gram matrix per sample (G_agg = 'average') whose size is not one of the solver's strides --------------------
// The general one-wavefront kernel loads such rows element by element (4.6x slower at k = 200, 40x at k = 320, and
// its 16-coefficients-per-lane variants spill).  Instead the matrices of a SLICE of the minibatch are copied into
// zero-padded slots of stride kq (the padding is dead coordinates, as for a shared matrix) and the slice goes to the
// four-wavefront solver; slices of at most kCdSliceBytes, two launches each.  The slots must be zero outside the k x k
// corners (the caller zero-fills the scratch once per k).  Bit-identical to cd_kernel.
constexpr size_t kCdSliceBytes = (size_t)64 << 20;
bool cd_split_enabled() { return g_cd_split.load(std::memory_order_relaxed) != 0; }   // (diagnostics switch MODL_DEBUG_CD_SPLIT)
// does the one-wavefront kernel exist for k coefficients?  (beyond 256 only in the diagnostics library: with MODL_DEBUG_CD_SPLIT
// = 0 the product library still solves such systems on the four-wavefront solver - every switch selects between correct paths)
bool cd_one_wave_covers(int k) {
#ifdef MODL_DIAG
    (void)k; return true;
#else
    return k <= 256;
#endif
}
int cd_per_sample_ld(int k) { const int kq = cd_padded_ld(k); return kq < 128 ? 128 : kq; }
size_t cd_per_sample_scratch_bytes(size_t tsz, int64_t b, int k) {
    const size_t per = (size_t)cd_per_sample_ld(k) * cd_per_sample_ld(k) * tsz;
    size_t n = kCdSliceBytes / per;
    if (n < 1) n = 1;
    if ((int64_t)n > b) n = (size_t)(b > 0 ? b : 1);
    return n * per;
}
template <typename T>
__global__ __launch_bounds__(256) void cd_pad_gram_batch_kernel(const T *G, int64_t g_stride, const int64_t *g_idx, int first,
                                                                int k, T *Gp, int kq) {
    const int r = blockIdx.x, i = blockIdx.y;                        // row, sample of the slice
    const T *g = G + (g_idx ? g_idx[first + i] : (int64_t)(first + i)) * g_stride + (int64_t)r * k;
    T *d = Gp + ((int64_t)i * kq + r) * kq;
    for (int c = threadIdx.x; c < k; c += 256) d[c] = g[c];
}
template <typename T>
int launch_cd_per_sample(hipStream_t stream, const CdArgs<T> &a, T *slots, size_t slot_bytes) {
    if (!a.g_stride || !slots) return MODL_EINVAL;
    const int k = a.k, kq = cd_per_sample_ld(k);
    const size_t per = (size_t)kq * kq * sizeof(T);
    const int ns_max = (int)std::min<size_t>(slot_bytes / per, (size_t)a.b);
    if (ns_max < 1) return MODL_ENOMEM;
    for (int s0 = 0; s0 < a.b; s0 += ns_max) {
        const int ns = a.b - s0 < ns_max ? a.b - s0 : ns_max;
        hipLaunchKernelGGL((cd_pad_gram_batch_kernel<T>), dim3((unsigned)k, (unsigned)ns), dim3(256), 0, stream, a.G, a.g_stride,
                           a.g_idx, s0, k, slots, kq);
        MODL_LAUNCH_CHECK();
        CdArgs<T> a2 = a;
        a2.G = slots; a2.g_stride = (int64_t)kq * kq; a2.g_idx = nullptr; a2.ldg = kq; a2.g_pad_rows = 0;
        a2.b = ns;
        a2.Dx = a.Dx + (int64_t)s0 * k;
        a2.xnorm2 = a.xnorm2 + s0;
        if (a.H0) a2.H0 = a.H0 + (int64_t)s0 * k;
        if (a.idx) a2.idx = a.idx + s0; else a2.code = a.code + (int64_t)s0 * k;
        if (a.code2) { if (a.idx2) a2.idx2 = a.idx2 + s0; else a2.code2 = a.code2 + (int64_t)s0 * k; }
        if (a.sweeps) a2.sweeps = a.sweeps + s0;
        if (!cd_split_applies<T>(a2)) return MODL_EINVAL;
        MODL_TRY(launch_cd_split<T>(stream, a2));
    }
    return MODL_OK;
}
template int launch_cd_per_sample<float>(hipStream_t, const CdArgs<float> &, float *, size_t);
template int launch_cd_per_sample<double>(hipStream_t, const CdArgs<double> &, double *, size_t);

// squared row norms: out[i] = sum_f X[i][f]^2   (dict_fact_fast.pyx:334 uses dot(y, y))
template <typename T>
__global__ __launch_bounds__(256) void row_norm2_kernel(const T *X, int64_t ldx, int64_t p, int64_t b, T *out) {
    __shared__ double red[4];
    const int64_t i = blockIdx.x;
    if (i >= b) return;
    const T *x = X + i * ldx;
    // 16-byte loads, four independent partial sums per thread (a dependent load -> add chain of p / 256
    // single elements is latency bound)
    constexpr int V = 16 / sizeof(T);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    const int64_t nv = vec ? p / V : 0;
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const vec_t *xv = reinterpret_cast<const vec_t *>(x);
    int64_t f = threadIdx.x;
    for (; f + 768 < nv; f += 1024) {
        const vec_t a0 = xv[f], a1 = xv[f + 256], a2 = xv[f + 512], a3 = xv[f + 768];
#pragma unroll
        for (int c = 0; c < V; ++c) {
            s0 += (double)a0[c] * (double)a0[c];
            s1 += (double)a1[c] * (double)a1[c];
            s2 += (double)a2[c] * (double)a2[c];
            s3 += (double)a3[c] * (double)a3[c];
        }
    }
    for (; f < nv; f += 256) {
        const vec_t a0 = xv[f];
#pragma unroll
        for (int c = 0; c < V; ++c) s0 += (double)a0[c] * (double)a0[c];
    }
    for (int64_t e = nv * V + threadIdx.x; e < p; e += 256) s1 += (double)x[e] * (double)x[e];
    double s = (s0 + s1) + (s2 + s3);
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[i] = (T)s;
}

template <typename T>
int launch_row_norm2(hipStream_t stream, const T *X, int64_t ldx, int64_t p, int64_t b, T *out) {
    if (b <= 0) return MODL_OK;
    hipLaunchKernelGGL((row_norm2_kernel<T>), dim3((unsigned)b), dim3(256), 0, stream, X, ldx, p, b, out);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_row_norm2<float>(hipStream_t, const float *, int64_t, int64_t, int64_t, float *);
template int launch_row_norm2<double>(hipStream_t, const double *, int64_t, int64_t, int64_t, double *);

}  // namespace modl

extern "C" int modl_is_diag_build(void) {
#ifdef MODL_DIAG
    return 1;
#else
    return 0;
#endif
}

extern "C" int modl_debug_set(int what, int64_t value) {
    if (what == MODL_DEBUG_CD_SPARSE_PCT) {
        modl::g_cd_sparse_pct.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
#ifdef MODL_DIAG     // shader-clock stamps into a caller-supplied buffer: the diagnostics library only
    if (what == MODL_DEBUG_CD_STAMPS) {
        modl::g_cd_stamps.store(reinterpret_cast<unsigned long long *>((uintptr_t)value));
        return MODL_OK;
    }
    if (what == MODL_DEBUG_ATOM_STAMPS) {
        modl::g_atom_stamps.store(reinterpret_cast<unsigned long long *>((uintptr_t)value));
        return MODL_OK;
    }
#endif
    if (what == MODL_DEBUG_STAGE_AHEAD) {
        modl::g_stage_ahead.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_BCD_TINY) {
        modl::g_bcd_tiny.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_BCD_PERSIST) {
        modl::g_bcd_persist.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_BCD_FEW) {
        modl::g_bcd_few.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_ATOM_PIPE) {
        modl::g_atom_pipe.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_ATOM_MWG) {
        modl::g_atom_mwg.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_RECSYS_FUSED) {
        modl::g_recsys_fused.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_STATS_RESIDENT) {
        modl::g_stats_resident.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_BCD_ACC) {
        modl::g_bcd_acc.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    if (what == MODL_DEBUG_CD_SPLIT) {
        modl::g_cd_split.store((int)value, std::memory_order_relaxed);
        return MODL_OK;
    }
    return MODL_EINVAL;
}
