// Micro-benchmark: what one Michelot pass of the lone projecting workgroup costs (bcd.hip: enet_project_slim) - 256
// threads (one wavefront per SIMD), 20 elements per thread in registers, the level of pass n + 1 depends on the sums
// of pass n.  Cycles = clock64() ticks per pass.
// Build: hipcc -O3 --offload-arch=gfx950 -I modl_amd/csrc scripts/micro/michelot_pass.hip -o scripts/micro/michelot_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
using namespace modl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int REP = 64;

template <int NW>
__device__ __forceinline__ void block_sum2_pp(double &a, double &b, double *red4, int &par) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    double *r = red4 + 2 * NW * par;
    par ^= 1;
    if (lane == 0) { r[2 * wid] = a; r[2 * wid + 1] = b; }
    __syncthreads();
    if (NW == 4) {
        a = (r[0] + r[2]) + (r[4] + r[6]);
        b = (r[1] + r[3]) + (r[5] + r[7]);
    } else {
        a = ((r[0] + r[2]) + (r[4] + r[6])) + ((r[8] + r[10]) + (r[12] + r[14]));
        b = ((r[1] + r[3]) + (r[5] + r[7])) + ((r[9] + r[11]) + (r[13] + r[15]));
    }
}

// WHICH: 0 full pass, 1 no exchange between the wavefronts (wave sums only), 2 element loop only, 3 exchange only
// (wave sums + LDS + barrier), 4 LDS + barrier only, 5 full pass, sums packed: count carried in the low bits? no:
// 5 = full pass with float count
template <int WHICH, int NW, int EPT>
__global__ __launch_bounds__(64 * NW) void k_pass(const double *in, double *out, unsigned long long *cyc, double R) {
    __shared__ double red4[32];
    double x[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = in[threadIdx.x + e * 64 * NW];
    int par = 0;
    double level = 1e-3;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    for (int p = 0; p < REP; ++p) {
        double S = level, cnt = 1.0;
        if constexpr (WHICH == 0 || WHICH == 1 || WHICH == 2) {
            double S0 = 0, S1 = 0;
            int c0 = 0, c1 = 0;
#pragma unroll
            for (int e = 0; e < EPT; e += 2) {
                const double a0 = fabs(x[e]), a1 = fabs(x[e + 1]);
                const bool i0 = a0 > level, i1 = a1 > level;
                S0 += i0 ? a0 : 0.0;
                S1 += i1 ? a1 : 0.0;
                c0 += i0 ? 1 : 0;
                c1 += i1 ? 1 : 0;
            }
            S = S0 + S1;
            cnt = (double)(c0 + c1);
        }
        if constexpr (WHICH == 0 || WHICH == 3) block_sum2_pp<NW>(S, cnt, red4, par);
        if constexpr (WHICH == 1) { S = wave_sum(S); cnt = wave_sum(cnt); }
        if constexpr (WHICH == 4) {
            double *r = red4 + 2 * NW * par;
            par ^= 1;
            if ((threadIdx.x & 63) == 0) { r[2 * (threadIdx.x >> 6)] = S; r[2 * (threadIdx.x >> 6) + 1] = cnt; }
            __syncthreads();
            S = (r[0] + r[2]) + (r[4] + r[6]);
            cnt = (r[1] + r[3]) + (r[5] + r[7]);
        }
        level = (S - R) / (cnt + 1.0) * 1e-6 + 1e-3;         // (a division on the chain, as in the solver; kept near 1e-3)
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; }
    out[threadIdx.x] = level;
}

template <int WHICH, int NW, int EPT>
int run(const char *name, const double *d_in, double *d_out, unsigned long long *d_cyc) {
    unsigned long long best = ~0ull;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((k_pass<WHICH, NW, EPT>), dim3(1), dim3(64 * NW), 0, 0, d_in, d_out, d_cyc, 0.5);
        CK(hipDeviceSynchronize());
        unsigned long long c;
        CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
        if (c < best) best = c;
    }
    printf("%-60s %8.0f cycles per pass\n", name, (double)best / REP);
    return 0;
}

int main() {
    double *d_in, *d_out;
    unsigned long long *d_cyc;
    constexpr int EPT = 20;
    CK(hipMalloc(&d_in, sizeof(double) * 256 * EPT));
    CK(hipMalloc(&d_out, sizeof(double) * 1024));
    CK(hipMalloc(&d_cyc, 64));
    double h[256 * EPT];
    for (int i = 0; i < 256 * EPT; ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-5 - 5e-3;
    CK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
    if (run<0, 4, 20>("full pass (selects, wave sums, LDS exchange, division)", d_in, d_out, d_cyc)) return 1;
    if (run<1, 4, 20>("selects + wave sums, no exchange", d_in, d_out, d_cyc)) return 1;
    if (run<2, 4, 20>("selects only", d_in, d_out, d_cyc)) return 1;
    if (run<3, 4, 20>("wave sums + LDS exchange only", d_in, d_out, d_cyc)) return 1;
    if (run<4, 4, 20>("LDS exchange + barrier only", d_in, d_out, d_cyc)) return 1;
    printf("eight wavefronts (two per SIMD), 10 elements per thread:\n");
    if (run<0, 8, 10>("full pass", d_in, d_out, d_cyc)) return 1;
    if (run<2, 8, 10>("selects only", d_in, d_out, d_cyc)) return 1;
    if (run<3, 8, 10>("wave sums + LDS exchange only", d_in, d_out, d_cyc)) return 1;
    printf("sixteen wavefronts (four per SIMD), 5 elements per thread... as 6:\n");
    if (run<0, 16, 6>("full pass (4 of the 16 partial sums read: timing only)", d_in, d_out, d_cyc)) return 1;
    return 0;
}
