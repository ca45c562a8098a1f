// Shared device/host helpers for libmodl_hip (gfx950 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/modl_hip.h"

#define MODL_HIP(call)                                   \
    do {                                                 \
        hipError_t modl_e_ = (call);                     \
        if (modl_e_ != hipSuccess) return (int)modl_e_;  \
    } while (0)
#define MODL_TRY(call)                    \
    do {                                  \
        int modl_r_ = (call);             \
        if (modl_r_ != MODL_OK) return modl_r_; \
    } while (0)
#define MODL_LAUNCH_CHECK() MODL_HIP(hipGetLastError())

namespace modl {

constexpr int kWave = 64;

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// --- wave-level primitives -------------------------------------------------
__device__ __forceinline__ float bcast_lane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ double bcast_lane(double v, int src_lane) {
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int bcast_lane(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }

// Wave-wide reductions without the LDS crossbar: four DPP stages inside each row of 16 lanes
// (quad_perm, quad_perm, row_half_mirror, row_mirror), then v_permlane16_swap (rows 0/1 and 2/3) and
// v_permlane32_swap (the two halves).  Every lane ends with the result; ~20 cheap instructions instead of six
// ds_bpermute round trips (~700 cycles).  Fixed association, so results are run-to-run deterministic.
template <int CTRL> __device__ __forceinline__ float dpp_perm(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_perm(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (a, b) = (value of the even / lower partner, value of the odd / upper partner), for rows (SWAP16) or halves
template <bool SWAP16> __device__ __forceinline__ void lane_swap(float x, float &a, float &b) {
    const unsigned int w = (unsigned int)__float_as_int(x);
    const auto r = SWAP16 ? __builtin_amdgcn_permlane16_swap(w, w, false, false) : __builtin_amdgcn_permlane32_swap(w, w, false, false);
    a = __int_as_float((int)r[0]);
    b = __int_as_float((int)r[1]);
}
template <bool SWAP16> __device__ __forceinline__ void lane_swap(double x, double &a, double &b) {
    const long long v = __double_as_longlong(x);
    const unsigned int w0 = (unsigned int)(v & 0xffffffffll), w1 = (unsigned int)(v >> 32);
    const auto r0 = SWAP16 ? __builtin_amdgcn_permlane16_swap(w0, w0, false, false) : __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
    const auto r1 = SWAP16 ? __builtin_amdgcn_permlane16_swap(w1, w1, false, false) : __builtin_amdgcn_permlane32_swap(w1, w1, false, false);
    a = __longlong_as_double(((long long)r1[0] << 32) | r0[0]);
    b = __longlong_as_double(((long long)r1[1] << 32) | r0[1]);
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    v += dpp_perm<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);      // row_half_mirror
    v += dpp_perm<0x140>(v);      // row_mirror
    T a, b;
    lane_swap<true>(v, a, b);
    v = a + b;
    lane_swap<false>(v, a, b);
    return a + b;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
    T u;
    u = dpp_perm<0xB1>(v); v = (u > v) ? u : v;
    u = dpp_perm<0x4E>(v); v = (u > v) ? u : v;
    u = dpp_perm<0x141>(v); v = (u > v) ? u : v;
    u = dpp_perm<0x140>(v); v = (u > v) ? u : v;
    T a, b;
    lane_swap<true>(v, a, b);
    v = (a > b) ? a : b;
    lane_swap<false>(v, a, b);
    return (a > b) ? a : b;
}

// the same for values that are >= 0 and not NaN (|differences|, |coefficients|): v_max with a DPP operand, one
// instruction per stage instead of move + compare + select
__device__ __forceinline__ float wave_max_nn(float v) {
    v = fmaxf(v, dpp_perm<0xB1>(v));
    v = fmaxf(v, dpp_perm<0x4E>(v));
    v = fmaxf(v, dpp_perm<0x141>(v));
    v = fmaxf(v, dpp_perm<0x140>(v));
    float a, b;
    lane_swap<true>(v, a, b);
    v = fmaxf(a, b);
    lane_swap<false>(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ double wave_max_nn(double v) {
    v = fmax(v, dpp_perm<0xB1>(v));
    v = fmax(v, dpp_perm<0x4E>(v));
    v = fmax(v, dpp_perm<0x141>(v));
    v = fmax(v, dpp_perm<0x140>(v));
    double a, b;
    lane_swap<true>(v, a, b);
    v = fmax(a, b);
    lane_swap<false>(v, a, b);
    return fmax(a, b);
}

// block-wide sum of doubles; `red` = LDS scratch of >= blockDim/64 doubles.
// Every thread gets the result.  Deterministic.
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    // the <= 16 per-wave sums are combined by another wave-level reduction (a serial loop over LDS reads is a
    // chain of ~150-cycle round trips)
    return wave_sum(lane < nw ? red[lane] : 0.0);
}
__device__ __forceinline__ double block_max(double v, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return wave_max(lane < nw ? red[lane] : red[0]);
}

// optional gather index: identity when both pointers are null
struct Gather {
    const int32_t *i32 = nullptr;
    const int64_t *i64 = nullptr;
    __host__ __device__ __forceinline__ int64_t operator()(int64_t i) const {
        return i64 ? i64[i] : (i32 ? (int64_t)i32[i] : i);
    }
    __host__ __device__ bool identity() const { return !i32 && !i64; }
};
static inline Gather gather32(const int32_t *p) { Gather g; g.i32 = p; return g; }
static inline Gather gather64(const int64_t *p) { Gather g; g.i64 = p; return g; }

template <typename T> struct DType;
template <> struct DType<float> { static constexpr int id = MODL_F32; };
template <> struct DType<double> { static constexpr int id = MODL_F64; };

}  // namespace modl
