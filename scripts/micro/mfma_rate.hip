// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 (the block kernel's main product): one wave, one or two
// accumulator chains; 1-4 waves of a workgroup at once.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_rate.hip -o scripts/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int N = 64;

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *out, unsigned long long *cyc, float a0) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f16v acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + lane, b = 1.f + 0.001f * lane;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    asm volatile("" : "+v"(a), "+v"(b));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float s = 0;
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    asm volatile("" : "+v"(s));
    const unsigned long long t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    out[threadIdx.x] = s;
    if (lane == 0) cyc[wid] = t1 - t0;
}

template <int NACC> int run(int waves, float *out, unsigned long long *cyc) {
    unsigned long long h[4], best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k_mfma<NACC>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.5f);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long m = 0;
        for (int w = 0; w < waves; ++w) m = h[w] > m ? h[w] : m;
        best = m < best ? m : best;
    }
    printf("%d wave(s), %d accumulator chain(s): %.1f cycles per v_mfma_f32_32x32x2_f32 (slowest wave)\n", waves, NACC, (double)best / N);
    return 0;
}

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 4)); CK(hipMalloc(&cyc, 32));
    run<1>(1, out, cyc); run<2>(1, out, cyc); run<4>(1, out, cyc);
    run<2>(2, out, cyc); run<2>(4, out, cyc);
    return 0;
}
