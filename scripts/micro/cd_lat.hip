// Micro-benchmark: latency of the primitives on the critical chain of the coordinate-descent sweep (cd_solver.hip),
// one wavefront alone on a SIMD, every instruction dependent on the previous one.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/cd_lat.hip -o scripts/micro/cd_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int REP = 256;
typedef float f2v __attribute__((ext_vector_type(2)));

template <int WHICH>
__global__ __launch_bounds__(64) void k_chain(float *out, unsigned long long *cyc, float seed) {
    const int lane = threadIdx.x;
    float x = seed + lane * 1e-3f, y = 1.0f + lane * 1e-4f;
    f2v h = {x, y}, q = {y, x};
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    asm volatile("" : "+v"(x), "+v"(y));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        if constexpr (WHICH == 0) x = __builtin_fmaf(x, 0.999f, y);                               // v_fma_f32
        if constexpr (WHICH == 1) { h = __builtin_elementwise_fma(h, q, q); }                      // v_pk_fma_f32
        if constexpr (WHICH == 2) {                                                                // readlane -> VALU with the SGPR
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 17));
            x = __builtin_fmaf(s, 0.999f, y);
        }
        if constexpr (WHICH == 3) {                                                                // readlane -> v_pk_fma with SGPR operand
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.x), 17));
            const f2v sv = {s, s};
            h = __builtin_elementwise_fma(sv, q, h);
        }
        if constexpr (WHICH == 4) x = __builtin_amdgcn_fmed3f(x, -1.f, 1.f) + y;                   // v_med3 + add
        if constexpr (WHICH == 5) {                                                                // the coordinate formula: fma, sub, med3, sub, mul
            const float Hii = __builtin_fmaf(-y, 0.5f, x);
            const float tmp = y - Hii;
            const float cl = __builtin_amdgcn_fmed3f(tmp, -1.f, 1.f);
            x = (tmp - cl) * 0.7f;
        }
        if constexpr (WHICH == 6) {                                                                // one whole coordinate as in cd_coord (KPL = 4)
            const float Hii = __builtin_fmaf(-y, q.x, h.x);
            const float tmp = y - Hii;
            const float cl = __builtin_amdgcn_fmed3f(tmp, -1.f, 1.f);
            const float xv = (tmp - cl) * 0.7f;
            const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), 17));
            const float dold = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 17));
            const f2v dnv = {dn, dn}, dov = {-dold, -dold};
            h = __builtin_elementwise_fma(dnv, q, __builtin_elementwise_fma(dov, q, h));
            y = (lane == 17) ? xv : y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(x), "+v"(h));
    const unsigned long long t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    out[lane] = x + h.x + h.y + y;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int W> int run(const char *name, float *out, unsigned long long *cyc) {
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k_chain<W>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5f);
        CK(hipDeviceSynchronize());
        unsigned long long h;
        CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("%-64s %7.1f cycles per repetition\n", name, (double)best / REP);
    return 0;
}

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&cyc, 8));
    run<0>("v_fma_f32 (dependent)", out, cyc);
    run<1>("v_pk_fma_f32 (dependent)", out, cyc);
    run<2>("v_readlane_b32 -> v_fma_f32 with the SGPR", out, cyc);
    run<3>("v_readlane_b32 -> v_pk_fma_f32 with the SGPR", out, cyc);
    run<4>("v_med3_f32 + v_add_f32", out, cyc);
    run<5>("coordinate formula (fma, sub, med3, sub, mul)", out, cyc);
    run<6>("whole coordinate, one H register pair (formula, 2 readlane, 2 pk_fma, select)", out, cyc);
    return 0;
}
