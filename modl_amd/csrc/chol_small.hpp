// In-LDS Cholesky factorisation and substitutions for small systems (k <= 128), shared by the ridge code solve
// (chol.hip: ridge_small_kernel) and the masked minibatch of RecsysDictFact (recsys.hip: recsys_code_kernel).
// The matrix W[k][ld] (ld odd: a column is conflict-free) lives in LDS; four wavefronts work (a workgroup with more -
// recsys.hip's fused minibatch kernel has eight - brings them along to the barriers, idle).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace modl {

// Called by all four wavefronts.  Pass j0 (four columns): the contraction over the finished columns m < j0 is split
// between the wavefronts (a quarter of the range each, whole steps of four), wavefronts 1-3 hand their partial sums to
// wavefront 0 through `part` ([3][RPL * 4][64] elements of LDS), which resolves the 4 x 4 diagonal block with lane
// broadcasts and writes the four columns: two workgroup barriers per FOUR columns, and the LDS latency of the
// contraction - the bulk of the work, on one wavefront 54 k of 120 k cycles at k = 70 - is paid by four wavefronts side
// by side.
template <typename T, int RPL>
__device__ __forceinline__ void chol_block_lds(T *W, int k, int ld, T *dinv, T *part) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int row[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) row[q] = (lane + 64 * q < k) ? lane + 64 * q : k - 1;    // (clamped: idle lanes repeat the last row)
    for (int j0 = 0; j0 < k; j0 += 4) {
        T c[RPL][4];
        int pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = (j0 + u < k) ? j0 + u : k - 1;
#pragma unroll
        for (int q = 0; q < RPL; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) c[q][u] = (wid == 0) ? W[row[q] * ld + pr[u]] : (T)0;
        const int steps = j0 / 4;                                       // steps of four finished columns
        const bool shared = steps >= 4;                                 // (fewer: wavefront 0 takes them all)
        const bool worker = wid < 4;
        const int s_lo = (shared && worker) ? steps * wid / 4 : 0;
        const int s_hi = !worker ? 0 : (shared ? steps * (wid + 1) / 4 : (wid == 0 ? steps : 0));
        for (int m = 4 * s_lo; m < 4 * s_hi; m += 4) {                  // every read of the four columns first
            T lj[4][4], w[RPL][4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int u = 0; u < 4; ++u) lj[v][u] = W[pr[u] * ld + m + v];
#pragma unroll
                for (int q = 0; q < RPL; ++q) w[q][v] = W[row[q] * ld + m + v];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int q = 0; q < RPL; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[q][u] = fma(-w[q][v], lj[v][u], c[q][u]);
        }
        if (shared && wid > 0 && worker) {
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) part[((wid - 1) * RPL * 4 + q * 4 + u) * 64 + lane] = c[q][u];
        }
        if (shared) __syncthreads();
        if (wid == 0) {
            if (shared) {
#pragma unroll
                for (int x = 0; x < 3; ++x)
#pragma unroll
                    for (int q = 0; q < RPL; ++q)
#pragma unroll
                        for (int u = 0; u < 4; ++u) c[q][u] += part[(x * RPL * 4 + q * 4 + u) * 64 + lane];
            }
            // the 4 x 4 diagonal block: pivots and the rows below them by lane broadcasts (row r lives in lane r % 64, slot
            // r / 64; j0 is a multiple of 4, so the four pivot rows share a slot)
            const int slot = (RPL > 1 && j0 >= 64) ? 1 : 0;             // (wave-uniform)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                if (j >= k) break;                                      // (wave-uniform)
                const T piv = slot ? bcast_lane(c[RPL - 1][u], j & 63) : bcast_lane(c[0][u], j & 63);
                // the reciprocal pivot: f32 from the hardware reciprocal square root (1 ulp) + one Newton step, the
                // pivot as its product with the radicand - two dependent instructions instead of the ~25 of sqrt + a
                // division on the chain of every column; f64 keeps sqrt and one division
                T d, inv;
                if constexpr (std::is_same<T, float>::value) {
                    const float y0 = __builtin_amdgcn_rsqf(piv);
                    inv = y0 * fmaf(-0.5f * piv * y0, y0, 1.5f);
                    d = piv * inv;
                } else {
                    d = sqrt(piv);
                    inv = (T)1 / d;
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) c[q][u] = (row[q] == j) ? d : c[q][u] * inv;   // rows < j: never read
                if (lane == 0) dinv[j] = inv;
#pragma unroll
                for (int u2 = u + 1; u2 < 4; ++u2) {
                    if (j0 + u2 >= k) break;
                    const T lr = slot ? bcast_lane(c[RPL - 1][u], (j0 + u2) & 63) : bcast_lane(c[0][u], (j0 + u2) & 63);   // L[j0 + u2][j]
#pragma unroll
                    for (int q = 0; q < RPL; ++q) c[q][u2] = fma(-c[q][u], lr, c[q][u2]);
                }
            }
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lane + 64 * q < k && j0 + u < k && lane + 64 * q >= j0 + u) W[(lane + 64 * q) * ld + j0 + u] = c[q][u];
        }
        __syncthreads();                                                // the four columns are in place
    }
}

// L y = b, L^T x = y for NR right-hand sides held by one wavefront (lane l: elements l and l + 64 of each).
// L: lower triangle of W; dinv: the reciprocals of its diagonal.  Column-oriented: step j reads column j (forward) or
// row j (backward) of L once for the whole batch - requested one step ahead, off the chain y_j -> update -> y_{j+1} -
// and multiplies by the reciprocal pivot.
template <typename T, int RPL, int NR>
__device__ __forceinline__ void chol_solve_wave_lds(const T *W, const T *dinv, int k, int ld, T (&y)[RPL][NR]) {
    const int lane = threadIdx.x & 63;
    int row[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) row[q] = (lane + 64 * q < k) ? lane + 64 * q : k - 1;
    auto step = [&](auto Q_, int j, T di, const T (&lv)[RPL], bool fwd) {
        constexpr int Q = decltype(Q_)::value;              // the slot that owns row j
        T lm[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int r = lane + 64 * q;
            lm[q] = ((fwd ? r > j : r < j) && r < k) ? lv[q] : (T)0;
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const T yj = bcast_lane(y[Q][t], j & 63) * di;
#pragma unroll
            for (int q = 0; q < RPL; ++q) y[q][t] = (lane + 64 * q == j) ? yj : fma(-lm[q], yj, y[q][t]);
        }
    };
    {   // forward: column j of L
        T dn = dinv[0], ln[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) ln[q] = W[row[q] * ld];
        for (int j = 0; j < k; ++j) {
            const T di = dn;
            T lv[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) lv[q] = ln[q];
            const int jn = (j + 1 < k) ? j + 1 : j;
            dn = dinv[jn];
#pragma unroll
            for (int q = 0; q < RPL; ++q) ln[q] = W[row[q] * ld + jn];
            if (RPL > 1 && j >= 64) step(std::integral_constant<int, RPL - 1>{}, j, di, lv, true);
            else step(std::integral_constant<int, 0>{}, j, di, lv, true);
        }
    }
    {   // backward: row j of L
        T dn = dinv[k - 1], ln[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) ln[q] = W[(k - 1) * ld + row[q]];
        for (int j = k - 1; j >= 0; --j) {
            const T di = dn;
            T lv[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) lv[q] = ln[q];
            const int jn = (j > 0) ? j - 1 : 0;
            dn = dinv[jn];
#pragma unroll
            for (int q = 0; q < RPL; ++q) ln[q] = W[jn * ld + row[q]];
            if (RPL > 1 && j >= 64) step(std::integral_constant<int, RPL - 1>{}, j, di, lv, false);
            else step(std::integral_constant<int, 0>{}, j, di, lv, false);
        }
    }
}

}  // namespace modl
