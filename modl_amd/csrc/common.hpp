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

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T u = __shfl_xor(v, o, 64);
        v = (u > v) ? u : v;
    }
    return v;
}

// block-wide sum of doubles; `red` = LDS scratch of >= blockDim/64 doubles.
// Every thread gets the result.  Deterministic.
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ double block_max(double v, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < nw; ++i) t = red[i] > t ? red[i] : t;
    return t;
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
