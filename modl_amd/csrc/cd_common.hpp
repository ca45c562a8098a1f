// Pieces shared by the two coordinate-descent kernels (cd_solver.hip, cd_split.hip).
#pragma once
#include <utility>

#include "common.hpp"

namespace modl {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int... Js, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Js...>, F &&f) {
    (f(std::integral_constant<int, Js>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

__device__ __forceinline__ float clamp3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double clamp3(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// One coordinate (dict_fact_fast.pyx:354-386), elementwise or on wave-uniform scalars; returns the new
// coefficient.
template <typename T, bool POSITIVE>
__device__ __forceinline__ T cd_coordinate(T h, T wo, T qq, T ri, T Qcc, T alpha) {
    const T Hii = fma(-wo, Qcc, h);                            // H[ii] after "H -= w_ii * Q[ii]" (:361-365)
    const T tmp = qq - Hii;                                    // :367
    // :372 soft threshold sign(tmp) max(|tmp| - alpha, 0) as tmp - clamp(tmp, -alpha, alpha): the same
    // rounded difference, two operations shorter (a zero result may carry the other sign)
    const T cl = POSITIVE ? (tmp < alpha ? tmp : alpha) : clamp3(tmp, -alpha, alpha);
    return (tmp - cl) * ri;
}

}  // namespace modl
