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
// gfx950 mapping: the k-vectors H = Q w, w, q and diag(Q) live in registers,
// KPL consecutive coefficients per lane (k <= 64 * KPL).  A coordinate step reads
// its four scalars with v_readlane, updates H with two fused multiply-adds per
// register against row ii of the Gram matrix, which is streamed from L2 as one
// coalesced row (16 B per lane at k = 256) and prefetched one chunk of rows ahead
// so that the load latency sits under the previous chunk's arithmetic.  The five
// reductions of the gap test are wave shuffles.  b independent problems -> b
// wavefronts, 4 per workgroup.
#include "kernels.hpp"

namespace modl {

// Row loader: KPL consecutive coefficients per lane.  VEC = rows are KPL-aligned (k % KPL == 0 and
// an aligned base): one 16-byte load per lane at k = 256.  No branches: lanes past the end read a
// clamped address and select zero, so the loads of a chunk stay in flight together.
template <typename T, int KPL, bool VEC>
__device__ __forceinline__ void load_row(const T *__restrict__ row, int k, int lane, int n_li, T (&r)[KPL]) {
    if constexpr (VEC) {
        const int lc = lane < n_li ? lane : n_li - 1;
        constexpr int kAlign = (KPL * sizeof(T) >= 16) ? 16 : (int)(KPL * sizeof(T));
        const T *p = static_cast<const T *>(__builtin_assume_aligned(row + lc * KPL, kAlign));
#pragma unroll
        for (int c = 0; c < KPL; ++c) r[c] = p[c];
        if (lane >= n_li) {
#pragma unroll
            for (int c = 0; c < KPL; ++c) r[c] = 0;
        }
    } else {
        const int e0 = lane * KPL;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int e = e0 + c;
            const T v = row[e < k ? e : k - 1];
            r[c] = e < k ? v : (T)0;
        }
    }
}

// A chunk of PG consecutive coordinates (dict_fact_fast.pyx:354-386).  They all live in lane `li`
// (registers C0 .. C0+PG-1), so the chunk's Gauss-Seidel recurrence is run on wave-uniform scalars:
// the PG entries of H it needs are read once (v_readlane), and after each coordinate the later ones
// are advanced with the same two fused multiply-adds the vector update applies to lane li — the
// values stay bit-identical to the vector H.  The k-wide H update (two FMAs per register and
// coordinate) is off the dependency chain.  A zero diagonal skips the coordinate (:357).
template <typename T, int KPL, int PG, int C0, bool VEC, bool POSITIVE>
__device__ __forceinline__ void cd_chunk(int li, int n_li, int lane, int k, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                         const T (&inv)[KPL], T (&cur)[PG][KPL], T (&nxt)[PG][KPL],
                                         const T *__restrict__ Q, T alpha, T &w_max, T &d_w_max) {
    const int ii0 = li * KPL + C0;
    // prefetch the rows of the next chunk (wraps to row 0 for the next sweep)
    int nxt0 = ii0 + PG;
    if (nxt0 >= n_li * KPL) nxt0 = 0;
#pragma unroll
    for (int j = 0; j < PG; ++j) {
        const int rn = (nxt0 + j < k) ? nxt0 + j : 0;
        load_row<T, KPL, VEC>(Q + (int64_t)rn * k, k, lane, n_li, nxt[j]);
    }
    T h[PG], wo[PG], qq[PG], ri[PG], wn[PG], Qb[PG][PG];
#pragma unroll
    for (int c = 0; c < PG; ++c) {
        h[c] = bcast_lane(H[C0 + c], li);
        wo[c] = bcast_lane(w[C0 + c], li);
        qq[c] = bcast_lane(q[C0 + c], li);
        ri[c] = (ii0 + c < k) ? bcast_lane(inv[C0 + c], li) : (T)0;      // 0: skipped coordinate
#pragma unroll
        for (int c2 = c; c2 < PG; ++c2) Qb[c][c2] = bcast_lane(cur[c][C0 + c2], li);
    }
#pragma unroll
    for (int c = 0; c < PG; ++c) {
        const bool live = ri[c] != (T)0;
        const T Hii = fma(-wo[c], Qb[c][c], h[c]);                // H[ii] after "H -= w_ii * Q[ii]" (:361-365)
        const T tmp = qq[c] - Hii;                                // :367
        T mag = fabs(tmp) - alpha;                                // :372 soft threshold
        mag = mag > (T)0 ? mag : (T)0;
        T x = copysign(mag * ri[c], tmp);
        if (POSITIVE && tmp < (T)0) x = 0;
        wn[c] = live ? x : wo[c];
        const T dn = live ? x : (T)0, dold = live ? wo[c] : (T)0;
#pragma unroll
        for (int c2 = c + 1; c2 < PG; ++c2) h[c2] = fma(dn, Qb[c][c2], fma(-dold, Qb[c][c2], h[c2]));
#pragma unroll
        for (int r = 0; r < KPL; ++r) H[r] = fma(dn, cur[c][r], fma(-dold, cur[c][r], H[r]));   // :361-365, :375-378
        if (live) {
            const T d = fabs(x - wo[c]);
            d_w_max = d > d_w_max ? d : d_w_max;
            const T aw = fabs(x);
            w_max = aw > w_max ? aw : w_max;
        }
    }
    if (lane == li) {
#pragma unroll
        for (int c = 0; c < PG; ++c) w[C0 + c] = wn[c];
    }
#pragma unroll
    for (int j = 0; j < PG; ++j)
#pragma unroll
        for (int c = 0; c < KPL; ++c) cur[j][c] = nxt[j][c];
}

template <typename T, int KPL, bool VEC, bool POSITIVE>
__global__ __launch_bounds__(256) void cd_kernel(CdArgs<T> a) {
    constexpr int PG = (KPL >= 4) ? 4 : KPL;     // rows per prefetch chunk
    const int lane = threadIdx.x & 63;
    const int smp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (smp >= a.b) return;                        // whole wave exits together
    const int k = a.k;
    const T *__restrict__ Q = a.G + (a.g_idx ? a.g_idx[smp] : (int64_t)smp) * a.g_stride;
    const int64_t row_out = a.idx ? a.idx[smp] : (int64_t)smp;
    T *wptr = a.code + row_out * k;
    const T *qptr = a.Dx + (int64_t)smp * k;
    const int e0 = lane * KPL;
    const int n_li = (k + KPL - 1) / KPL;          // lanes that own coefficients
    const T alpha = a.alpha, beta = a.beta;
    constexpr bool positive = POSITIVE;

    T w[KPL], H[KPL], q[KPL], dg[KPL], inv[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool in = e0 + c < k;
        const int e = in ? e0 + c : 0;
        const T wv = wptr[e], qv = qptr[e], dv = Q[(int64_t)e * k + e];
        w[c] = in ? wv : (T)0;
        q[c] = in ? qv : (T)0;
        dg[c] = in ? dv : (T)0;
        inv[c] = (dg[c] != (T)0) ? (T)1 / (dg[c] + beta) : (T)0;   // reciprocal of the step denominator (:373)
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
        // H = Q w as a combination of rows (Q is symmetric, as the solver itself assumes)
        for (int li = 0; li < n_li; ++li) {
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int j = li * KPL + c;
                const T wj = (j < k) ? bcast_lane(w[c], li) : (T)0;
                T r[KPL];
                load_row<T, KPL, VEC>(Q + (int64_t)(j < k ? j : 0) * k, k, lane, n_li, r);
#pragma unroll
                for (int c2 = 0; c2 < KPL; ++c2) H[c2] = fma(wj, r[c2], H[c2]);
            }
        }
    }

    T cur[PG][KPL], nxt[PG][KPL];
#pragma unroll
    for (int j = 0; j < PG; ++j) load_row<T, KPL, VEC>(Q + (int64_t)(j < k ? j : 0) * k, k, lane, n_li, cur[j]);

    int n_iter = 0;
    for (; n_iter < a.max_iter; ++n_iter) {
        T w_max = 0, d_w_max = 0;
        for (int li = 0; li < n_li; ++li) {
            cd_chunk<T, KPL, PG, 0, VEC, POSITIVE>(li, n_li, lane, k, w, H, q, inv, cur, nxt, Q, alpha, w_max, d_w_max);
            if constexpr (KPL > 4)
                cd_chunk<T, KPL, PG, 4, VEC, POSITIVE>(li, n_li, lane, k, w, H, q, inv, cur, nxt, Q, alpha, w_max, d_w_max);
            if constexpr (KPL > 8) {
                cd_chunk<T, KPL, PG, 8, VEC, POSITIVE>(li, n_li, lane, k, w, H, q, inv, cur, nxt, Q, alpha, w_max, d_w_max);
                cd_chunk<T, KPL, PG, 12, VEC, POSITIVE>(li, n_li, lane, k, w, H, q, inv, cur, nxt, Q, alpha, w_max, d_w_max);
            }
        }
        if (w_max == (T)0 || d_w_max / w_max < d_w_tol || n_iter == a.max_iter - 1) {   // :388
            T s_qw = 0, s_wH = 0, s_ww = 0, s_l1 = 0;
            T xmax = positive ? -INFINITY : (T)0;
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                s_qw += w[c] * q[c];
                s_wH += w[c] * H[c];
                s_ww += w[c] * w[c];
                s_l1 += fabs(w[c]);
                if (e0 + c < k) {
                    const T x = (q[c] - H[c]) - beta * w[c];                          // :397
                    const T m = positive ? x : fabs(x);
                    xmax = m > xmax ? m : xmax;
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
#pragma unroll
    for (int c = 0; c < KPL; ++c)
        if (e0 + c < k) wptr[e0 + c] = w[c];
    if (a.sweeps && lane == 0) a.sweeps[smp] = n_iter;
}

template <typename T, int KPL>
static void launch_cd_kpl(hipStream_t stream, const CdArgs<T> &a, dim3 grid, dim3 block) {
    constexpr size_t kRowAlign = (KPL * sizeof(T) >= 16) ? 16 : KPL * sizeof(T);
    const bool vec = (a.k % KPL == 0) && (reinterpret_cast<uintptr_t>(a.G) % kRowAlign == 0) &&
                     ((a.g_stride * sizeof(T)) % kRowAlign == 0) && (((size_t)a.k * sizeof(T)) % kRowAlign == 0);
    if (vec) {
        if (a.positive) hipLaunchKernelGGL((cd_kernel<T, KPL, true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((cd_kernel<T, KPL, true, false>), grid, block, 0, stream, a);
    } else {
        if (a.positive) hipLaunchKernelGGL((cd_kernel<T, KPL, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((cd_kernel<T, KPL, false, false>), grid, block, 0, stream, a);
    }
}

template <typename T>
int launch_cd(hipStream_t stream, const CdArgs<T> &a) {
    if (a.b <= 0 || a.k <= 0) return MODL_OK;
    if (a.k > 1024) return MODL_EINVAL;
    dim3 grid((unsigned)cdiv(a.b, 4)), block(256);
    if (a.k <= 64) launch_cd_kpl<T, 1>(stream, a, grid, block);
    else if (a.k <= 128) launch_cd_kpl<T, 2>(stream, a, grid, block);
    else if (a.k <= 256) launch_cd_kpl<T, 4>(stream, a, grid, block);
    else if (a.k <= 512) launch_cd_kpl<T, 8>(stream, a, grid, block);
    else launch_cd_kpl<T, 16>(stream, a, grid, block);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

template int launch_cd<float>(hipStream_t, const CdArgs<float> &);
template int launch_cd<double>(hipStream_t, const CdArgs<double> &);

// squared row norms: out[i] = sum_f X[i][f]^2   (dict_fact_fast.pyx:334 uses dot(y, y))
template <typename T>
__global__ __launch_bounds__(256) void row_norm2_kernel(const T *X, int64_t ldx, int64_t p, int64_t b, T *out) {
    __shared__ double red[4];
    const int64_t i = blockIdx.x;
    if (i >= b) return;
    const T *x = X + i * ldx;
    double s = 0;
    for (int64_t f = threadIdx.x; f < p; f += 256) s += (double)x[f] * (double)x[f];
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
