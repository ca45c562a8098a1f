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

template <typename T, int KPL>
__device__ __forceinline__ void load_row(const T *__restrict__ row, int k, int lane, T (&r)[KPL], bool vec_ok) {
    const int e0 = lane * KPL;
    if (vec_ok) {
        if (e0 < k) {
            constexpr int kAlign = (KPL * sizeof(T) >= 16) ? 16 : (int)(KPL * sizeof(T));
            const T *p = static_cast<const T *>(__builtin_assume_aligned(row + e0, kAlign));
#pragma unroll
            for (int c = 0; c < KPL; ++c) r[c] = p[c];
        } else {
#pragma unroll
            for (int c = 0; c < KPL; ++c) r[c] = 0;
        }
    } else {
#pragma unroll
        for (int c = 0; c < KPL; ++c) r[c] = (e0 + c < k) ? row[e0 + c] : (T)0;
    }
}

template <typename T, int KPL, int C>
__device__ __forceinline__ void cd_coordinate(int li, int lane, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                              const T (&dg)[KPL], const T (&row)[KPL], T alpha, T beta,
                                              bool positive, T &w_max, T &d_w_max) {
    const T Qii = bcast_lane(dg[C], li);
    if (Qii == (T)0) return;                                  // dict_fact_fast.pyx:357
    const T w_ii = bcast_lane(w[C], li);
    T Hii = bcast_lane(H[C], li);
    const T qii = bcast_lane(q[C], li);
    Hii = fma(-w_ii, Qii, Hii);                               // H[ii] after "H -= w_ii * Q[ii]" (:361-365)
    const T tmp = qii - Hii;                                  // :367
    T wn;
    if (positive && tmp < (T)0) {
        wn = 0;
    } else {                                                  // :372 soft threshold
        T mag = fabs(tmp) - alpha;
        mag = mag > (T)0 ? mag : (T)0;
        const T sg = (tmp > (T)0) ? (T)1 : ((tmp < (T)0) ? (T)-1 : (T)0);
        wn = sg * mag / (Qii + beta);
    }
    if (w_ii != (T)0 || wn != (T)0) {                         // the two axpys of :361-365 and :375-378
#pragma unroll
        for (int c = 0; c < KPL; ++c) H[c] = fma(wn, row[c], fma(-w_ii, row[c], H[c]));
    }
    if (lane == li) w[C] = wn;
    const T d = fabs(wn - w_ii);
    d_w_max = d > d_w_max ? d : d_w_max;
    const T aw = fabs(wn);
    w_max = aw > w_max ? aw : w_max;
}

template <typename T, int KPL, int PG, int C0>
__device__ __forceinline__ void cd_chunk(int li, int n_li, int lane, int k, T (&w)[KPL], T (&H)[KPL], const T (&q)[KPL],
                                         const T (&dg)[KPL], T (&cur)[PG][KPL], T (&nxt)[PG][KPL],
                                         const T *__restrict__ Q, bool vec_ok, T alpha, T beta, bool positive,
                                         T &w_max, T &d_w_max) {
    const int ii0 = li * KPL + C0;
    // prefetch the rows of the next chunk (wraps to row 0 for the next sweep)
    int nxt0 = ii0 + PG;
    if (nxt0 >= n_li * KPL) nxt0 = 0;
#pragma unroll
    for (int j = 0; j < PG; ++j) {
        const int rn = (nxt0 + j < k) ? nxt0 + j : 0;
        load_row<T, KPL>(Q + (int64_t)rn * k, k, lane, nxt[j], vec_ok);
    }
    if (ii0 + 0 < k) cd_coordinate<T, KPL, C0 + 0>(li, lane, w, H, q, dg, cur[0], alpha, beta, positive, w_max, d_w_max);
    if constexpr (PG > 1) {
        if (ii0 + 1 < k) cd_coordinate<T, KPL, C0 + 1>(li, lane, w, H, q, dg, cur[1], alpha, beta, positive, w_max, d_w_max);
    }
    if constexpr (PG > 2) {
        if (ii0 + 2 < k) cd_coordinate<T, KPL, C0 + 2>(li, lane, w, H, q, dg, cur[2], alpha, beta, positive, w_max, d_w_max);
        if (ii0 + 3 < k) cd_coordinate<T, KPL, C0 + 3>(li, lane, w, H, q, dg, cur[3], alpha, beta, positive, w_max, d_w_max);
    }
#pragma unroll
    for (int j = 0; j < PG; ++j)
#pragma unroll
        for (int c = 0; c < KPL; ++c) cur[j][c] = nxt[j][c];
}

template <typename T, int KPL>
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
    constexpr int kRowAlign = (KPL * sizeof(T) >= 16) ? 16 : (int)(KPL * sizeof(T));
    const bool vec_ok = (k % KPL == 0) && ((reinterpret_cast<uintptr_t>(Q) % kRowAlign) == 0);
    const int e0 = lane * KPL;
    const int n_li = (k + KPL - 1) / KPL;          // lanes that own coefficients

    T w[KPL], H[KPL], q[KPL], dg[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool in = e0 + c < k;
        w[c] = in ? wptr[e0 + c] : (T)0;
        q[c] = in ? qptr[e0 + c] : (T)0;
        dg[c] = in ? Q[(int64_t)(e0 + c) * k + (e0 + c)] : (T)0;
        H[c] = 0;
    }
    const T y_norm2 = a.xnorm2[smp];
    const T tol_abs = a.tol * y_norm2;             // :336
    const T d_w_tol = a.tol;
    const T alpha = a.alpha, beta = a.beta;
    const bool positive = a.positive != 0;

    if (a.H0) {
        const T *hp = a.H0 + (int64_t)smp * k;
#pragma unroll
        for (int c = 0; c < KPL; ++c) H[c] = (e0 + c < k) ? hp[e0 + c] : (T)0;
    } else {
        // H = Q w as a combination of rows (Q is symmetric, as the solver itself assumes)
        for (int li = 0; li < n_li; ++li) {
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int j = li * KPL + c;
                const T wj = (j < k) ? bcast_lane(w[c], li) : (T)0;
                if (wj != (T)0) {
                    T r[KPL];
                    load_row<T, KPL>(Q + (int64_t)j * k, k, lane, r, vec_ok);
#pragma unroll
                    for (int c2 = 0; c2 < KPL; ++c2) H[c2] = fma(wj, r[c2], H[c2]);
                }
            }
        }
    }

    T cur[PG][KPL], nxt[PG][KPL];
#pragma unroll
    for (int j = 0; j < PG; ++j) load_row<T, KPL>(Q + (int64_t)(j < k ? j : 0) * k, k, lane, cur[j], vec_ok);

    int n_iter = 0;
    for (; n_iter < a.max_iter; ++n_iter) {
        T w_max = 0, d_w_max = 0;
        for (int li = 0; li < n_li; ++li) {
            cd_chunk<T, KPL, PG, 0>(li, n_li, lane, k, w, H, q, dg, cur, nxt, Q, vec_ok, alpha, beta, positive, w_max, d_w_max);
            if constexpr (KPL > 4)
                cd_chunk<T, KPL, PG, 4>(li, n_li, lane, k, w, H, q, dg, cur, nxt, Q, vec_ok, alpha, beta, positive, w_max, d_w_max);
            if constexpr (KPL > 8) {
                cd_chunk<T, KPL, PG, 8>(li, n_li, lane, k, w, H, q, dg, cur, nxt, Q, vec_ok, alpha, beta, positive, w_max, d_w_max);
                cd_chunk<T, KPL, PG, 12>(li, n_li, lane, k, w, H, q, dg, cur, nxt, Q, vec_ok, alpha, beta, positive, w_max, d_w_max);
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

template <typename T>
int launch_cd(hipStream_t stream, const CdArgs<T> &a) {
    if (a.b <= 0 || a.k <= 0) return MODL_OK;
    if (a.k > 1024) return MODL_EINVAL;
    dim3 grid((unsigned)cdiv(a.b, 4)), block(256);
    if (a.k <= 64) hipLaunchKernelGGL((cd_kernel<T, 1>), grid, block, 0, stream, a);
    else if (a.k <= 128) hipLaunchKernelGGL((cd_kernel<T, 2>), grid, block, 0, stream, a);
    else if (a.k <= 256) hipLaunchKernelGGL((cd_kernel<T, 4>), grid, block, 0, stream, a);
    else if (a.k <= 512) hipLaunchKernelGGL((cd_kernel<T, 8>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((cd_kernel<T, 16>), grid, block, 0, stream, a);
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
