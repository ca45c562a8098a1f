// Micro-benchmark: what do the integer atomics of the Gram accumulator cost a launch of the blocked dictionary update,
// and can they retire in the shadow of other work of the same launch?  (a block launch adds ~1100 64-bit
// atomics per workgroup at its very end and cannot end before they have retired, ~2 us.)
//   G workgroups x 256 threads; every thread adds E 64-bit values (no return) to E * 256 addresses that ALL workgroups
//   share (address = tid + 256 e, like the accumulator's entries), then spins for `busy` ticks of the 100 MHz wall clock.
//   Reported: time per launch of a chain of dependent launches (events around 200 launches), for
//     - E = 0 (the boundary itself), 4, 8, 12, 24, 48 with no busy loop: the drain as a function of the count;
//     - the same with a busy loop of ~5 us AFTER the atomics: if they retire in its shadow, the launch costs busy + boundary;
//     - the same with the busy loop BEFORE the atomics (today's order): busy + drain + boundary.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/atomic_drain.hip -o scripts/micro/atomic_drain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_atomics(unsigned long long *acc, int E, long long busy, int order, int it) {
    const unsigned long long t0 = wall_clock64();
    if (order == 1 && busy > 0) while ((long long)(wall_clock64() - t0) < busy) __builtin_amdgcn_s_sleep(1);
    for (int e = 0; e < E; ++e) atomicAdd(acc + threadIdx.x + 256 * e, (unsigned long long)(it + e + 1));
    if (order == 0 && busy > 0) while ((long long)(wall_clock64() - t0) < busy) __builtin_amdgcn_s_sleep(1);
}

int main() {
    unsigned long long *acc;
    CK(hipMalloc(&acc, sizeof(unsigned long long) * 256 * 64));
    CK(hipMemset(acc, 0, sizeof(unsigned long long) * 256 * 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int N = 200;
    // wall_clock64() runs at 100 MHz: 500 ticks = 5 us
    const long long busy5 = 500;
    for (int G : {30, 157}) {
        for (int order : {-1, 0, 1}) {
            for (int E : {0, 4, 8, 12, 24, 48}) {
                const long long busy = order < 0 ? 0 : busy5;
                for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_atomics, dim3(G), dim3(256), 0, 0, acc, E, busy, order, w);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_atomics, dim3(G), dim3(256), 0, 0, acc, E, busy, order, i);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("workgroups %3d  %s  atomics/thread %2d (%5d per workgroup): %.2f us per launch\n", G,
                       order < 0 ? "no busy loop      " : (order == 0 ? "atomics, then 5 us" : "5 us, then atomics"), E, E * 256,
                       1e3 * ms / N);
            }
        }
    }
    return 0;
}
