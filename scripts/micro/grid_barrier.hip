// Micro-benchmark: cost of a grid-wide barrier (hand-rolled, agent-scope release/acquire) per iteration,
// for grids of 17 and 157 workgroups of 320 threads — the shapes of the dictionary-update block launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned int *counter, unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(320) void bench_kernel(unsigned int *counter, double *buf, int iters, unsigned long long *cycles) {
    const unsigned long long t0 = clock64();
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        // every workgroup writes a record, then all read all records (like the Gram records)
        buf[(size_t)(it & 1) * gridDim.x * 1024 + blockIdx.x * 1024 + threadIdx.x] = (double)(it + blockIdx.x);
        grid_barrier(counter, (unsigned int)(it + 1) * gridDim.x);
        for (unsigned int z = 0; z < gridDim.x && z < 16; ++z) acc += buf[(size_t)(it & 1) * gridDim.x * 1024 + z * 1024 + threadIdx.x];
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; buf[0] += acc * 0; }
}

int main() {
    unsigned int *counter; double *buf; unsigned long long *cyc;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&buf, 2 * 160 * 1024 * 8)); CK(hipMalloc(&cyc, 160 * 8));
    for (int nwg : {17, 157}) {
        for (int iters : {1, 101}) {
            CK(hipMemset(counter, 0, 4));
            void *args[] = {&counter, &buf, &iters, &cyc};
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void *)bench_kernel, dim3(nwg), dim3(320), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(nwg);
            CK(hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost));
            printf("nwg %3d iters %3d: kernel %.1f us, wg0 %llu cycles -> %.0f cycles per iteration\n", nwg, iters, ms * 1e3, h[0], (double)h[0] / iters);
        }
    }
    return 0;
}
