// Micro-benchmark: what does a launch boundary between two DEPENDENT kernels of one stream cost on this part,
// and what does it depend on?  (the block launches of the dictionary update pay ~4.6 us each, the
// guide measures 1.1-1.9 us.)  Variants of a chain of N launches on one stream:
//   grid x block, dynamic LDS, kernarg bytes, dirty bytes written per launch (spread over the workgroups),
//   code size (a long straight-line body executed once), and a fence-free in-kernel grid barrier
//   (relaxed agent-scope atomics on the ticket, sc1 stores/loads for the payload only) for comparison.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/launch_gap.hip -o scripts/micro/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { unsigned long long pad[48]; };   // 384 bytes of by-value kernel arguments

__global__ __launch_bounds__(320) void k_empty(float *buf, int dirty_per_thread, unsigned long long *stamps, int it) {
    extern __shared__ char smem[];
    if (stamps && threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * it] = wall_clock64();
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    for (int i = 0; i < dirty_per_thread; ++i) buf[base + (size_t)i * gridDim.x * blockDim.x] = (float)(it + i);
    if (dirty_per_thread < 0) smem[threadIdx.x] = 1;   // keep the LDS allocation alive
    if (stamps && threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * it + 1] = wall_clock64();
}

__global__ __launch_bounds__(320) void k_bigargs(Big a, Big b, float *buf, int it) {
    if (a.pad[it & 31] == 0xdeadbeef && b.pad[3] == 1) buf[0] = 1.f;
}

// a long straight-line body (~24 KB of code), every instruction executed once: cold instruction cache per launch?
template <int N>
__global__ __launch_bounds__(64) void k_code(float *buf, float x0, int it) {
    float x = x0 + threadIdx.x, y = 1.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) { x = __builtin_fmaf(x, 1.0001f + i * 1e-7f, y); y = __builtin_fmaf(y, 0.9999f, x * 1e-9f); }
    if (x == 12345.678f) buf[threadIdx.x] = x + y;
}

// fence-free grid barrier: payload through sc1 (write-through / L2-bypassing) accesses only
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(320) void k_persistent(unsigned int *ticket, double *rec, int iters, int rec_doubles,
                                                     unsigned long long *cycles, int fenced) {
    const unsigned long long t0 = wall_clock64();
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double *mine = rec + ((size_t)(it & 1) * gridDim.x + blockIdx.x) * rec_doubles;
        for (int e = threadIdx.x; e < rec_doubles; e += blockDim.x) {
            if (fenced) mine[e] = (double)(it + e); else st_sc1(mine + e, (double)(it + e));
        }
        if (fenced) { __syncthreads(); if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = (unsigned int)(it + 1) * gridDim.x;
            while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // every workgroup reads every record (like the Gram records of the block update)
        for (unsigned int z = 0; z < gridDim.x; ++z) {
            const double *r = rec + ((size_t)(it & 1) * gridDim.x + z) * rec_doubles;
            for (int e = threadIdx.x; e < rec_doubles; e += blockDim.x) acc += fenced ? r[e] : ld_sc1(r + e);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; if (acc == 1.2345) rec[0] = acc; }
}

// keeps the stream busy for `ticks` of the 100 MHz wall clock, so that the launches enqueued behind it are all in
// the queue when they start: the chain is then timed on the GPU side (the host needs ~3 us per hipLaunchKernelGGL,
// which is what a chain of EMPTY kernels measures otherwise)
__global__ void k_blocker(unsigned long long ticks, float *buf) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks == 1) buf[0] = 0;
}

static float chain_us(int n, const std::function<void(int)> &launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(i);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_blocker, dim3(1), dim3(64), 0, 0, (unsigned long long)(n * 6 * 100 + 200000), (float *)nullptr);   // 6 us per launch + 2 ms
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}

// dependent-issue latency of the VALU: the same dependent chain as a loop (small code) and fully unrolled (every
// instruction fetched once: instruction-cache / fetch bound?)
__global__ __launch_bounds__(64) void k_chain_loop(float *buf, float x0, int n, unsigned long long *cyc) {
    float x = x0 + threadIdx.x;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        x = __builtin_fmaf(x, 1.0001f, 0.5f); x = __builtin_fmaf(x, 0.9999f, 0.25f); x = __builtin_fmaf(x, 1.0002f, 0.125f); x = __builtin_fmaf(x, 0.9998f, 0.0625f);
        x = __builtin_fmaf(x, 1.0001f, 0.5f); x = __builtin_fmaf(x, 0.9999f, 0.25f); x = __builtin_fmaf(x, 1.0002f, 0.125f); x = __builtin_fmaf(x, 0.9998f, 0.0625f);
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (x == 12345.678f) buf[threadIdx.x] = x;
}
template <int N>
__global__ __launch_bounds__(64) void k_chain_unrolled(float *buf, float x0, unsigned long long *cyc) {
    float x = x0 + threadIdx.x;
    const unsigned long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, 1.0001f + (i & 7) * 1e-6f, 0.5f);
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (x == 12345.678f) buf[threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void k_chain_f64_loop(double *buf, double x0, int n, unsigned long long *cyc) {
    double x = x0 + threadIdx.x;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        x = __builtin_fma(x, 1.0001, 0.5); x = __builtin_fma(x, 0.9999, 0.25); x = __builtin_fma(x, 1.0002, 0.125); x = __builtin_fma(x, 0.9998, 0.0625);
        x = __builtin_fma(x, 1.0001, 0.5); x = __builtin_fma(x, 0.9999, 0.25); x = __builtin_fma(x, 1.0002, 0.125); x = __builtin_fma(x, 0.9998, 0.0625);
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (x == 12345.678) buf[threadIdx.x] = x;
}

#include <functional>
int main() {
    float *buf; CK(hipMalloc(&buf, (size_t)64 << 20));
    unsigned long long *stamps; CK(hipMalloc(&stamps, 4096 * 16));
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int N = 2000;
    struct V { int grid, block, lds, dirty; };
    for (V v : {V{1, 64, 0, 0}, V{32, 320, 0, 0}, V{32, 320, 53 * 1024, 0}, V{256, 320, 53 * 1024, 0}, V{32, 320, 53 * 1024, 8},
                V{32, 320, 53 * 1024, 100}, V{256, 256, 0, 38}, V{256, 256, 0, 150}}) {
        float us = chain_us(N, [&](int i) {
            hipLaunchKernelGGL(k_empty, dim3(v.grid), dim3(v.block), v.lds, 0, buf, v.dirty, (unsigned long long *)nullptr, i);
        });
        printf("chain k_empty grid %3d block %3d lds %5d dirty %6.2f MB/launch : %.2f us per launch\n", v.grid, v.block, v.lds,
               (double)v.dirty * v.grid * v.block * 4 / 1e6, us);
    }
    {
        Big a{}, b{};
        float us = chain_us(N, [&](int i) { hipLaunchKernelGGL(k_bigargs, dim3(32), dim3(320), 0, 0, a, b, buf, i); });
        printf("chain k_bigargs (768 B of by-value kernargs), grid 32 block 320 : %.2f us per launch\n", us);
    }
    {
        float us = chain_us(N, [&](int i) { hipLaunchKernelGGL((k_code<3000>), dim3(32), dim3(64), 0, 0, buf, 1.0f, i); });
        float us2 = chain_us(N, [&](int i) { hipLaunchKernelGGL((k_code<30>), dim3(32), dim3(64), 0, 0, buf, 1.0f, i); });
        printf("chain k_code: 6000 dependent fma straight-line (48 KB of code) %.2f us, 60 fma %.2f us per launch  (6000 fma at 2.4 GHz, 4-8 cycles each: 10-20 us)\n", us, us2);
    }
    {
        unsigned long long *cyc; CK(hipMalloc(&cyc, 64 * 8));
        unsigned long long h[4];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_chain_loop, dim3(1), dim3(64), 0, 0, buf, 1.0f, 1000, cyc);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL((k_chain_unrolled<8000>), dim3(1), dim3(64), 0, 0, buf, 1.0f, cyc);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(&h[1], cyc, 8, hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(k_chain_f64_loop, dim3(1), dim3(64), 0, 0, (double *)buf, 1.0, 1000, cyc);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(&h[2], cyc, 8, hipMemcpyDeviceToHost));
            printf("dependent fma chain, one wave (clock64 ticks per fma): f32 loop %.2f, f32 unrolled x8000 (64 KB of code) %.2f, f64 loop %.2f\n",
                   (double)h[0] / 8000, (double)h[1] / 8000, (double)h[2] / 8000);
        }
        // clock64 vs wall clock: how many clock64 ticks per microsecond?
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain_loop, dim3(1), dim3(64), 0, 0, buf, 1.0f, 200000, cyc);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        CK(hipMemcpy(&h[0], cyc, 8, hipMemcpyDeviceToHost));
        printf("clock64: %.1f ticks per us (kernel of %.1f us)\n", (double)h[0] / (ms * 1e3), ms * 1e3);
    }
    {   // gap between the last instruction of launch i and the first of launch i + 1, on the 100 MHz wall clock
        const int M = 512;
        for (V v : {V{32, 320, 53 * 1024, 0}, V{32, 320, 53 * 1024, 8}}) {
            for (int i = 0; i < M; ++i) hipLaunchKernelGGL(k_empty, dim3(v.grid), dim3(v.block), v.lds, 0, buf, v.dirty, stamps, i);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(2 * M);
            CK(hipMemcpy(h.data(), stamps, 2 * M * 8, hipMemcpyDeviceToHost));
            double gap = 0, body = 0;
            for (int i = 100; i < M; ++i) { gap += (double)(h[2 * i] - h[2 * i - 1]); body += (double)(h[2 * i + 1] - h[2 * i]); }
            printf("wall-clock stamps (10 ns ticks), dirty %.2f MB: end(i-1) -> start(i) %.2f us, body %.2f us\n",
                   (double)v.dirty * v.grid * v.block * 4 / 1e6, gap / (M - 100) * 0.01, body / (M - 100) * 0.01);
        }
    }
    {
        unsigned int *ticket; double *rec; unsigned long long *cyc;
        CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&rec, 2 * 160 * 640 * 8)); CK(hipMalloc(&cyc, 160 * 8));
        for (int nwg : {32, 157}) for (int fenced : {1, 0}) {
            int iters = 200, recd = 560;
            CK(hipMemset(ticket, 0, 4));
            void *args[] = {&ticket, &rec, &iters, &recd, &cyc, &fenced};
            CK(hipLaunchCooperativeKernel((const void *)k_persistent, dim3(nwg), dim3(320), args, 0, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(nwg);
            CK(hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost));
            printf("persistent grid %3d, %s barrier + all-read-all of 560-double records: %.2f us per iteration\n", nwg,
                   fenced ? "FENCED (release/acquire agent)" : "fence-free (sc1 payload, relaxed ticket)", (double)h[0] * 0.01 / iters);
        }
    }
    return 0;
}
