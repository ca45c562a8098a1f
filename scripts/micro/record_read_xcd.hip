// Micro-benchmark: the 30 Gram records (4480 B each) of the blocked dictionary update, written by the workgroups of one
// launch and summed by every workgroup of the next - does it matter whether writers and readers sit on ONE XCD (block b
// runs on XCD b % 8: grids of 8 x 31 workgroups in which only the multiples of 8 work) or are spread over the eight?
// Same-XCD readers find the records in their own L2 if a kernel boundary leaves them there.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/record_read_xcd.hip -o scripts/micro/record_read_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NREC = 30, STRIDE = 560, NWG = 31;
typedef double d2v __attribute__((ext_vector_type(2)));

// writer: workgroup w (of NWG working ones) writes record w; `one_xcd`: the working workgroups are blockIdx = 8 w
__global__ __launch_bounds__(320) void k_write(double *rec, int it, int one_xcd, int *xcc) {
    int w = (int)blockIdx.x;
    if (one_xcd) { if (w % 8) return; w /= 8; }
    if (w >= NREC) return;
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[w] = (int)(id & 0xf); }
    for (int e = threadIdx.x; e < STRIDE; e += 320) rec[(size_t)w * STRIDE + e] = (double)((e + w) % 97) + it;
}
__global__ __launch_bounds__(320) void k_read(const double *rec, double *out, unsigned long long *cyc, int one_xcd, int *xcc) {
    int w = (int)blockIdx.x;
    if (one_xcd) { if (w % 8) return; w /= 8; }
    if (w >= NWG) return;
    const int tid = threadIdx.x;
    const unsigned long long t0 = clock64();
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[32 + w] = (int)(id & 0xf); }
    constexpr int PAIRS = STRIDE / 2;
    d2v v[NREC];
    const int e2 = tid < PAIRS ? tid : 0;
    const d2v *base = reinterpret_cast<const d2v *>(rec) + e2;
#pragma unroll
    for (int u = 0; u < NREC; ++u) v[u] = base[(size_t)u * (STRIDE / 2)];
    __builtin_amdgcn_sched_barrier(0);
    d2v tot = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < NREC; ++u) tot += v[u];
    out[w * 320 + tid] = tot.x + tot.y;
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (tid == 0) cyc[w] = t1 - t0;
}

int main() {
    double *rec, *out; unsigned long long *cyc; int *xcc;
    CK(hipMalloc(&rec, NREC * STRIDE * 8)); CK(hipMalloc(&out, 32 * 320 * 8)); CK(hipMalloc(&cyc, 32 * 8)); CK(hipMalloc(&xcc, 64 * 4));
    for (int one = 0; one < 2; ++one) {
        unsigned long long best = ~0ull, h[32];
        int hx[64];
        for (int r = 0; r < 8; ++r) {
            hipLaunchKernelGGL(k_write, dim3(one ? 8 * NWG : NWG), dim3(320), 0, 0, rec, r, one, xcc);
            hipLaunchKernelGGL(k_read, dim3(one ? 8 * NWG : NWG), dim3(320), 0, 0, rec, out, cyc, one, xcc);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
            unsigned long long m = 0;
            for (int w = 0; w < NWG; ++w) m = h[w] > m ? h[w] : m;
            if (m < best) best = m;
        }
        CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
        int wx = 0, rx = 0;
        for (int w = 0; w < NREC; ++w) wx |= 1 << hx[w];
        for (int w = 0; w < NWG; ++w) rx |= 1 << hx[32 + w];
        printf("%-44s %8llu cycles (slowest of %d workgroups, best of 8); XCD masks: writers 0x%02x readers 0x%02x\n",
               one ? "writers and readers on ONE XCD" : "writers and readers spread over the XCDs", best, NWG, wx, rx);
    }
    return 0;
}
