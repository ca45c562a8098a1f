// Micro-benchmark: what does it cost to take the parameter staging kernel (a read of pinned host memory, one PCIe
// round trip) off the main stream?  (a) today: staging kernel + work kernels in ONE stream; (b) staging kernel on a
// side stream, hipEventRecord there, hipStreamWaitEvent on the main stream in front of the work kernels; (c) as (b)
// plus an event recorded at the END of every step on the main stream that the side stream waits for before it
// reuses a parameter slot (what a ring of device parameter blocks needs to stay safe).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_wait.hip -o scripts/micro/stream_wait
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_stage(const uint4 *src, uint4 *dst, int n16) {
    for (int e = threadIdx.x; e < n16; e += 1024) dst[e] = src[e];
}

__global__ __launch_bounds__(256) void k_work(float *x, const uint4 *params, int spin) {
    float v = x[blockIdx.x * 256 + threadIdx.x] + (float)params[threadIdx.x & 63].x;
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 0.999f, 0.001f);
    x[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    constexpr int kSlots = 8, kIter = 2000, kWork = 4, n16 = 512;       // 8 KB of parameters, 4 kernels of ~6 us per step
    hipStream_t main_s, side_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking));
    char *h[kSlots]; uint4 *hd[kSlots];
    for (int i = 0; i < kSlots; ++i) {
        CK(hipHostMalloc((void **)&h[i], n16 * 16, hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void **)&hd[i], h[i], 0));
        for (int j = 0; j < n16 * 16; ++j) h[i][j] = (char)(i + j);
    }
    uint4 *dparams; float *x;
    CK(hipMalloc(&dparams, (size_t)kSlots * n16 * 16)); CK(hipMalloc(&x, 256 * 256 * 4)); CK(hipMemset(x, 0, 256 * 256 * 4));
    hipEvent_t staged[kSlots], done[kSlots];
    for (int i = 0; i < kSlots; ++i) {
        CK(hipEventCreateWithFlags(&staged[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    const int spin = 1500;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < kIter; ++it) {
                const int slot = it % kSlots;
                uint4 *dst = dparams + (size_t)(mode == 0 ? 0 : slot) * n16;
                if (mode == 0) {
                    hipLaunchKernelGGL(k_stage, dim3(1), dim3(1024), 0, main_s, hd[slot], dst, n16);
                } else {
                    if (mode == 2 && it >= kSlots) CK(hipStreamWaitEvent(side_s, done[slot], 0));
                    hipLaunchKernelGGL(k_stage, dim3(1), dim3(1024), 0, side_s, hd[slot], dst, n16);
                    CK(hipEventRecord(staged[slot], side_s));
                    CK(hipStreamWaitEvent(main_s, staged[slot], 0));
                }
                for (int w = 0; w < kWork; ++w) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, main_s, x, dst, spin);
                if (mode == 2) CK(hipEventRecord(done[slot], main_s));
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / kIter;
            if (rep == 1)
                printf("%-78s %7.2f us per step\n",
                       mode == 0 ? "(a) staging kernel in the main stream" :
                       mode == 1 ? "(b) staging on a side stream + event + wait on the main stream" :
                                   "(c) as (b) + end-of-step event on the main stream the side stream waits for", us);
        }
    }
    {   // the work kernels alone
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < kIter; ++it)
            for (int w = 0; w < kWork; ++w) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, main_s, x, dparams, spin);
        CK(hipDeviceSynchronize());
        printf("%-78s %7.2f us per step\n", "(d) the work kernels alone",
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / kIter);
    }
    return 0;
}
