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
                                     double l1_ratio, double *red) {
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
    for (int pass = 0; pass < 256; ++pass) {
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

}  // namespace modl
