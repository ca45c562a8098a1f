// Workgroup-level elastic-net ball geometry for one atom (vector of n entries
// with element stride `inc`): norm, rescale-to-radius and Euclidean projection.
//
// Replaces modl/utils/math/enet.pyx (enet_norm :125-148, enet_projection
// :38-122, enet_scale :150-167).  The projection onto
//   { u : sum_i |u_i| (rho + (1 - rho) |u_i|) <= radius }
// is unique, so instead of the reference's sequential pivot ("quickselect")
// search for the soft-threshold level l we run Michelot's active-set iteration:
// solve the level equation on the current support, drop the entries at or below
// the level, repeat until the support is stable.  Each pass is a block reduction
// (double accumulation), typically < 10 passes; the fixed point satisfies the
// same closed form (enet.pyx:112-119) the reference evaluates on its (s, rho).
#pragma once
#include "common.hpp"

#ifndef MODL_MAX_PASS
#define MODL_MAX_PASS 256
#endif
namespace modl {

template <typename T>
__device__ __forceinline__ double block_enet_norm(const T *v, int64_t inc, int64_t n, double l1_ratio, double *red) {
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double a = fabs((double)v[i * inc]);
        s += a * (l1_ratio + (1.0 - l1_ratio) * a);
    }
    return block_sum(s, red);
}

// v -> out (may alias).  Returns the enet norm of the result (every thread).
template <typename T>
__device__ double block_enet_project(const T *v, int64_t inc_v, T *out, int64_t inc_o, int64_t n, double radius,
                                     double l1_ratio, double *red, unsigned long long *dbg = nullptr) {
    if (!(radius > 0.0)) {                                   // enet.pyx:57-59 (radius == 0 -> zeros)
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[i * inc_o] = 0;
        return 0.0;
    }
    if (l1_ratio == 0.0) {                                   // enet.pyx:62-70, radius in squared-norm units
        double s = 0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const double x = (double)v[i * inc_v];
            s += x * x;
        }
        s = block_sum(s, red);
        const T scale = (s <= radius) ? (T)1 : (T)sqrt(s / radius);
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[i * inc_o] = v[i * inc_v] / scale;
        return (s <= radius) ? s : radius;
    }
    const double gamma = 2.0 / l1_ratio - 2.0;
    const double R = radius / l1_ratio;
    double tot = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double a = fabs((double)v[i * inc_v]);
        tot += a * (1.0 + 0.5 * gamma * a);
    }
    tot = block_sum(tot, red);
    if (tot <= R) {                                          // inside the ball: copy
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[i * inc_o] = v[i * inc_v];
        return tot * l1_ratio;
    }
    double level = 0.0, prev_cnt = -1.0;
    int pass = 0;
    for (; pass < MODL_MAX_PASS; ++pass) {
        double S = 0, cnt = 0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = fabs((double)v[i * inc_v]);
            if (a > level) { S += a * (1.0 + 0.5 * gamma * a); cnt += 1.0; }
        }
        S = block_sum(S, red);
        cnt = block_sum(cnt, red);
        if (cnt == prev_cnt || cnt == 0.0) break;
        prev_cnt = cnt;
        if (gamma != 0.0) {                                  // enet.pyx:113-117
            const double qa = gamma * gamma * R + gamma * cnt * 0.5;
            const double qd = 2.0 * R * gamma + cnt;
            const double qc = R - S;
            level = (-qd + sqrt(qd * qd - 4.0 * qa * qc)) / (2.0 * qa);
        } else {                                             // :119
            level = (S - R) / cnt;
        }
    }
    if (dbg && threadIdx.x == 0) { dbg[4] = (unsigned long long)pass; dbg[6] = clock64(); }
    const double lT = (double)(T)level;
    const double den = 1.0 + lT * gamma;
    double nrm = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)v[i * inc_v];
        double pos = fabs(x) - lT;
        pos = pos > 0 ? pos : 0;
        const T o = (T)(((x >= 0) ? pos : -pos) / den);      // enet.pyx:121, sign(0) = +1
        out[i * inc_o] = o;
        const double a = fabs((double)o);
        nrm += a * (l1_ratio + (1.0 - l1_ratio) * a);
    }
    return block_sum(nrm, red);
}

// two block-wide sums with ONE exchange; red2 = LDS scratch of >= 2 * blockDim / 64 doubles
__device__ __forceinline__ void block_sum2(double &a, double &b, double *red2, int nthreads) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (nthreads + 63) >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if (lane == 0) { red2[2 * wid] = a; red2[2 * wid + 1] = b; }
    __syncthreads();
    if (nw == 4) {
        // four wavefronts (every register-resident projection): the four partial sums as broadcast reads, added in
        // the association of the wave reduction, (r0 + r1) + (r2 + r3) - same bits, a quarter of its instructions
        a = (red2[0] + red2[2]) + (red2[4] + red2[6]);
        b = (red2[1] + red2[3]) + (red2[5] + red2[7]);
        return;
    }
    a = wave_sum(lane < nw ? red2[2 * lane] : 0.0);
    b = wave_sum(lane < nw ? red2[2 * lane + 1] : 0.0);
}
// one block-wide sum (the same exchange, without a second value travelling along)
__device__ __forceinline__ void block_sum1(double &a, double *red2, int nthreads) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (nthreads + 63) >> 6;
    a = wave_sum(a);
    __syncthreads();
    if (lane == 0) red2[2 * wid] = a;
    __syncthreads();
    if (nw == 4) {
        a = (red2[0] + red2[2]) + (red2[4] + red2[6]);
        return;
    }
    a = wave_sum(lane < nw ? red2[2 * lane] : 0.0);
}

// The same projection for vectors of at most EPT * nthreads elements (unit-stride input), run by the first
// `nthreads` threads of the workgroup (the others must have left), with every thread's elements held in
// REGISTERS across the Michelot passes: a pass is a handful of compares and one block exchange instead of a scan
// of the vector.  Four wavefronts (one per SIMD) beat sixteen here: the passes are reductions, and wavefronts
// sharing a SIMD serialise them.  red2: >= 2 * nthreads / 64 doubles.
// element i is written to out[(rows ? rows[i] : i) * row_stride]: the scatter offsets are fetched up front, together with
// the vector, so that the write-back is not a chain of index load -> store round trips.
// (Every load is UNCONDITIONAL, from a clamped index, the selection follows: a per-thread branch around a load makes the
// compiler wait for that load on the spot, and 2 * EPT requests become as many serial memory round trips -- 22 k of the
// 43 k cycles of a projection before this was removed.)
template <int EPT>
__device__ __forceinline__ void enet_scatter_offsets(const int32_t *rows, int64_t row_stride, int64_t n, int nthreads,
                                                     int64_t (&dst)[EPT]) {
    if (rows) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t i = threadIdx.x + (int64_t)e * nthreads;
            const int64_t r = (int64_t)rows[i < n ? i : n - 1];
            dst[e] = (i < n) ? r * row_stride : 0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t i = threadIdx.x + (int64_t)e * nthreads;
            dst[e] = (i < n) ? i * row_stride : 0;
        }
    }
}

// The projection of the register-resident vector x (element threadIdx.x + e * nthreads in x[e], zeros beyond n; the values
// are T-representable).  Writes the result to out[dst[e]] and LEAVES IT IN x; returns its enet norm (every thread).
template <typename T, int EPT>
__device__ __forceinline__ double block_enet_project_vals(double (&x)[EPT], const int64_t (&dst)[EPT], T *out, int64_t n, double radius,
                                          double l1_ratio, double *red2, int nthreads, unsigned long long *dbg = nullptr,
                                          double *level_io = nullptr) {
    if (!(radius > 0.0)) {                                   // enet.pyx:57-59 (radius == 0 -> zeros)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            if (threadIdx.x + (int64_t)e * nthreads < n) out[dst[e]] = 0;
            x[e] = 0.0;
        }
        return 0.0;
    }
    if (l1_ratio == 0.0) {                                   // enet.pyx:62-70, radius in squared-norm units
        double s = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) s += x[e] * x[e];
        block_sum1(s, red2, nthreads);
        const T scale = (s <= radius) ? (T)1 : (T)sqrt(s / radius);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t i = threadIdx.x + (int64_t)e * nthreads;
            const T o = (T)x[e] / scale;
            if (i < n) out[dst[e]] = o;
            x[e] = (i < n) ? (double)o : 0.0;
        }
        return (s <= radius) ? s : radius;
    }
    const double gamma = 2.0 / l1_ratio - 2.0;
    const double R = radius / l1_ratio;
    double ax[EPT], term[EPT];                              // |x| and its contribution, computed once
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        ax[e] = fabs(x[e]);
        term[e] = ax[e] * (1.0 + 0.5 * gamma * ax[e]);
    }
    // WARM START (level_io: the level this atom ended with at its previous projection, 0 = none).  Michelot's
    // iteration converges to the level l* from ANY starting level l0 <= l* (levels only rise, entries at or below a
    // level <= l* are never in the final support), and it ends on the same support, hence on the same sums in the
    // same order: identical bits.  l0 <= l* holds iff the thresholded vector at l0 still lies outside the ball,
    // h(l0) = sum_{|x|>l0} u (1 + gamma u / 2) >= R with u = (|x| - l0) / (1 + l0 gamma) (h decreases in l, and
    // h(0) is the norm of the vector itself: a verified guess also settles the inside-the-ball test); h(l0) follows
    // from the sums of the first Michelot pass (plus sum |x| when gamma != 0), so the check costs nothing when it holds
    // and one pass when it does not (cold start).  Between two minibatches an atom's level moves little: 8 passes -> 3
    // on the fMRI shape.
    double level = 0.0, prev_cnt = -1.0;
    bool warm = false;
    if (level_io) {
        const double l0 = 0.9 * *level_io;
        if (l0 > 0.0 && l0 < 1e300) {
            double S = 0, S1 = 0, P = 0;
            int c0 = 0, c1 = 0;
            if (gamma != 0.0) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) P += (ax[e] > l0) ? ax[e] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < EPT; e += 2) {
                const bool i0 = ax[e] > l0, i1 = ax[e + 1] > l0;
                S += i0 ? term[e] : 0.0;
                S1 += i1 ? term[e + 1] : 0.0;
                c0 += i0 ? 1 : 0;
                c1 += i1 ? 1 : 0;
            }
            S += S1;
            double cnt = (double)(c0 + c1);
            block_sum2(S, cnt, red2, nthreads);
            double h0;
            if (gamma != 0.0) {
                block_sum1(P, red2, nthreads);
                const double d = 1.0 + l0 * gamma;
                const double sum_u = (P - cnt * l0) / d;
                const double sum_a2 = (S - P) / (0.5 * gamma);
                const double sum_u2 = (sum_a2 - 2.0 * l0 * P + cnt * l0 * l0) / (d * d);
                h0 = sum_u + 0.5 * gamma * sum_u2;
            } else {
                h0 = S - cnt * l0;
            }
            if (h0 >= R * (1.0 + 1e-9) && cnt != 0.0) {      // (otherwise the guess overshoots, or cannot be told apart)
                warm = true;
                if (dbg && threadIdx.x == 0) { dbg[4] = 1 | (1 << 16); dbg[5] = (unsigned long long)cnt; }
                prev_cnt = cnt;
                if (gamma != 0.0) {
                    const double qa = gamma * gamma * R + gamma * cnt * 0.5;
                    const double qd = 2.0 * R * gamma + cnt;
                    const double qc = R - S;
                    level = (-qd + sqrt(qd * qd - 4.0 * qa * qc)) / (2.0 * qa);
                } else {
                    level = (S - R) / cnt;
                }
            }
        }
    }
    if (!warm) {
        double tot = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) tot += term[e];
        block_sum1(tot, red2, nthreads);
        if (tot <= R) {                                      // inside the ball: copy
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int64_t i = threadIdx.x + (int64_t)e * nthreads;
                if (i < n) out[dst[e]] = (T)x[e];
            }
            return tot * l1_ratio;
        }
    }
    for (int pass = 0; pass < 256; ++pass) {
        double S = 0, S1 = 0;
        int c0 = 0, c1 = 0;
#pragma unroll
        for (int e = 0; e < EPT; e += 2) {                  // selects, no branches; two chains
            const bool i0 = ax[e] > level, i1 = ax[e + 1] > level;
            S += i0 ? term[e] : 0.0;
            S1 += i1 ? term[e + 1] : 0.0;
            c0 += i0 ? 1 : 0;
            c1 += i1 ? 1 : 0;
        }
        S += S1;
        double cnt = (double)(c0 + c1);
        block_sum2(S, cnt, red2, nthreads);
        if (dbg && threadIdx.x == 0) { dbg[4] = (dbg[4] & (1 << 16)) | (unsigned)(pass + 1 + (warm ? 1 : 0)); dbg[5] = (unsigned long long)cnt; }
        if (cnt == prev_cnt || cnt == 0.0) break;
        prev_cnt = cnt;
        if (gamma != 0.0) {                                  // enet.pyx:113-117
            const double qa = gamma * gamma * R + gamma * cnt * 0.5;
            const double qd = 2.0 * R * gamma + cnt;
            const double qc = R - S;
            level = (-qd + sqrt(qd * qd - 4.0 * qa * qc)) / (2.0 * qa);
        } else {                                             // :119
            level = (S - R) / cnt;
        }
    }
    if (dbg && threadIdx.x == 0) dbg[6] = clock64();
    if (level_io && threadIdx.x == 0) *level_io = level;
    const double lT = (double)(T)level;
    const double inv_den = 1.0 / (1.0 + lT * gamma);       // one division (exactly 1 for l1 atoms, gamma = 0), EPT products
    double nrm = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + (int64_t)e * nthreads;
        double pos = fabs(x[e]) - lT;
        pos = pos > 0 ? pos : 0;
        const T o = (T)(((x[e] >= 0) ? pos : -pos) * inv_den);   // enet.pyx:121, sign(0) = +1
        if (i < n) out[dst[e]] = o;
        x[e] = (i < n) ? (double)o : 0.0;
        const double a = fabs(x[e]);
        nrm += a * (l1_ratio + (1.0 - l1_ratio) * a);
    }
    block_sum1(nrm, red2, nthreads);
    return nrm;
}

// block_enet_project for an l1 ball (l1_ratio == 1), in place on a unit-stride vector too long for the registers of
// nthreads threads (the s-vector of an atom beyond 24 elements per thread, kept in LDS by the caller): the same level
// equation on the same supports - hence the same sums in the same order and the same bits - with
//   * both sums of a pass in ONE exchange (block_sum2: same association as two block_sum calls),
//   * the cold start's first pass doubling as the inside-the-ball test (level 0: S is the vector's l1 norm),
//   * the warm start of block_enet_project_vals (level_io: the level this atom ended with at its previous projection).
// 9 scans of 7.4 k cycles each at 10 000 elements before (profiles/r04_atom_step_c6_stamps.txt).  red2: >= 2 * nthreads / 64.
// w_generic MUST point into LDS: it is read and written through an LDS-typed pointer (through the generic pointer the
// caller has - a select of the global and the LDS copy - every access was a flat load: 4.4 k cycles per scan).
template <typename T>
__device__ double block_l1_project_inplace(T *w_generic, int64_t n, double radius, double *red2, int nthreads, double *level_io,
                                           unsigned long long *dbg = nullptr) {
    typedef __attribute__((address_space(3))) T lds_T;
    lds_T *w = (lds_T *)w_generic;
    if (!(radius > 0.0)) {                                   // enet.pyx:57-59 (radius == 0 -> zeros)
        for (int64_t i = threadIdx.x; i < n; i += nthreads) w[i] = 0;
        return 0.0;
    }
    const double R = radius;
    // (eight elements per thread requested before the first is used: two at a time, each iteration waited for its own LDS
    //  round trip - 5.2 k cycles per scan of 40 elements per thread)
    constexpr int NQ = 8;
    auto scan = [&](double level, double &S, double &cnt) {
        double S0 = 0;
        int c0 = 0;
        for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)NQ * nthreads) {
            T v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t i = i0 + (int64_t)q * nthreads;
                v[q] = w[i < n ? i : i0];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {                   // (in the order of the plain loop: same sums)
                const int64_t i = i0 + (int64_t)q * nthreads;
                const double a = (i < n) ? fabs((double)v[q]) : 0.0;
                const bool in = a > level;                   // (level >= 0: the padding never counts)
                S0 += in ? a : 0.0;
                c0 += in ? 1 : 0;
            }
        }
        S = S0;
        cnt = (double)c0;
        block_sum2(S, cnt, red2, nthreads);
    };
    double level = 0.0, prev_cnt = -1.0, S, cnt;
    bool warm = false;
    int pass = 0;
    if (level_io) {
        const double l0 = 0.9 * *level_io;
        if (l0 > 0.0 && l0 < 1e300) {
            scan(l0, S, cnt);
            ++pass;
            if (S - cnt * l0 >= R * (1.0 + 1e-9) && cnt != 0.0) {   // the guess lies at or below the level: Michelot continues from it
                warm = true;
                prev_cnt = cnt;
                level = (S - R) / cnt;
            }
        }
    }
    if (!warm) {
        scan(0.0, S, cnt);                                   // S = the l1 norm (zeros add nothing), cnt = the non-zeros
        ++pass;
        if (S <= R) return S;                                // inside the ball: nothing moves
        if (cnt != 0.0) { prev_cnt = cnt; level = (S - R) / cnt; }
    }
    for (; pass < MODL_MAX_PASS && cnt != 0.0; ++pass) {
        scan(level, S, cnt);
        if (cnt == prev_cnt || cnt == 0.0) break;
        prev_cnt = cnt;
        level = (S - R) / cnt;                               // enet.pyx:119
    }
    if (dbg && threadIdx.x == 0) { dbg[4] = (unsigned long long)pass | (warm ? 1u << 16 : 0u); dbg[6] = clock64(); }
    if (level_io && threadIdx.x == 0) *level_io = level;
    const double lT = (double)(T)level;
    double nrm = 0;
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += (int64_t)NQ * nthreads) {
        T v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t i = i0 + (int64_t)q * nthreads;
            v[q] = w[i < n ? i : i0];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t i = i0 + (int64_t)q * nthreads;
            const double x = (double)v[q];
            double pos = fabs(x) - lT;
            pos = pos > 0 ? pos : 0;
            const T o = (T)((x >= 0) ? pos : -pos);          // enet.pyx:121, sign(0) = +1 (denominator 1 for an l1 ball)
            if (i < n) {
                w[i] = o;
                nrm += fabs((double)o);
            }
        }
    }
    block_sum1(nrm, red2, nthreads);
    return nrm;
}

template <typename T, int EPT>
__device__ double block_enet_project_reg(const T *v, T *out, const int32_t *rows, int64_t row_stride, int64_t n,
                                         double radius, double l1_ratio, double *red2, int nthreads,
                                         unsigned long long *dbg = nullptr, double *level_io = nullptr) {
    int64_t dst[EPT];
    enet_scatter_offsets<EPT>(rows, row_stride, n, nthreads, dst);
    double x[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + (int64_t)e * nthreads;
        const T val = v[i < n ? i : n - 1];
        x[e] = (i < n) ? (double)val : 0.0;
    }
    return block_enet_project_vals<T, EPT>(x, dst, out, n, radius, l1_ratio, red2, nthreads, dbg, level_io);
}

}  // namespace modl
