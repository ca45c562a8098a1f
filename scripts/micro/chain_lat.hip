// Micro-benchmark: latency of the primitives on the critical chain of the alpha recursion (bcd.hip: resolve_wave),
// one wavefront alone on a SIMD, every instruction dependent on the previous one.  Cycles = s_memtime ticks of
// clock64() per repetition (the same clock as the in-kernel stamps of scripts/diag_stamps.py).
// Build: hipcc -O3 --offload-arch=gfx950 -I modl_amd/csrc scripts/micro/chain_lat.hip -o scripts/micro/chain_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
using namespace modl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int REP = 256;

template <int WHICH>
__global__ __launch_bounds__(64) void k_chain(double *out, unsigned long long *cyc, double seed) {
    __shared__ double lds[128];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-4;
    lds[lane] = x;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    asm volatile("" : "+v"(x), "+v"(y));            // the chain starts after the first stamp ...
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        if constexpr (WHICH == 0) x = __builtin_fma(x, 0.999, y);                                  // v_fma_f64
        if constexpr (WHICH == 1) x = x + y;                                                       // v_add_f64
        if constexpr (WHICH == 2) x += dpp_perm<0xB1>(x);                                          // 2 mov_dpp + add (quad_perm)
        if constexpr (WHICH == 3) x += dpp_perm<0x140>(x);                                         // row_mirror
        if constexpr (WHICH == 4) { double a, b; lane_swap<true>(x, a, b); x = a + b; }           // permlane16_swap pair + add
        if constexpr (WHICH == 5) { double a, b; lane_swap<false>(x, a, b); x = a * b; }          // permlane32_swap pair + mul
        if constexpr (WHICH == 6) x = __builtin_amdgcn_rsq(x) + 1.0;                               // v_rsq_f64 + add
        if constexpr (WHICH == 7) {                                                                // the alpha tail
            const double yy = __builtin_amdgcn_rsq(x);
            const double r = __builtin_fma(-(0.5 * yy), x * yy, 0.5);
            const double yn = __builtin_fma(yy, r, yy);
            const double sy = y * yn;
            double al;
            asm("v_min_f64 %0, %1, %2" : "=v"(al) : "v"(sy), "v"(y));
            x = al + 1.0;
        }
        if constexpr (WHICH == 8) { lds[lane] = x; __builtin_amdgcn_wave_barrier(); x = lds[lane ^ 1] + 1.0; }   // LDS round trip
        if constexpr (WHICH == 9) x = (double)((float)x * 0.999f);                                 // cvt f64->f32, mul, cvt back
        if constexpr (WHICH == 10) {                                                               // whole 32-lane sum as in resolve_wave
            double pr = x * y;
            pr += dpp_perm<0xB1>(pr);
            pr += dpp_perm<0x4E>(pr);
            pr += dpp_perm<0x141>(pr);
            pr += dpp_perm<0x140>(pr);
            double r0, r1;
            lane_swap<true>(pr, r0, r1);
            x = (r0 + r1) * 0.03;
        }
        if constexpr (WHICH == 11) {                                                               // row sum through row_shr DPP adds of f32 halves? no: readlane chain
            const double v = bcast_lane(x, 17);
            x = v + y;
        }
        if constexpr (WHICH == 12) {                                                               // 32-lane sum with the f64 matrix core: ones^T x
            typedef double d4v __attribute__((ext_vector_type(4)));
            d4v acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0, acc, 0, 0, 0);                      // D[i][j] = sum_k x(i + 16 k): 4 row sums
            x = (acc[0] + acc[1]) * 0.01 + y;
        }
        if constexpr (WHICH == 13) {                                                               // v_mov_b64 (register copy of a double)
            double c;
            asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "v"(x));
            x = c + y;
        }
        if constexpr (WHICH == 14) {                                                               // ds_swizzle-free: DPP row_bcast15 add (rows 0->1, 2->3)
            x += dpp_perm<0x142>(x);
        }
        if constexpr (WHICH == 16) {                                                               // 16-lane sum: two v_mfma_f64_4x4x4 (4 blocks)
            double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
            double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, 1.0, 0.0, 0, 0, 0);
            x = s2 * 0.05 + y;
        }
        if constexpr (WHICH == 17) {                                                               // the same with the first result as B operand
            double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
            double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s1, 0.0, 0, 0, 0);
            x = s2 * 0.05 + y;
        }
        if constexpr (WHICH == 15) {                                                               // DPP wave_shr / row_bcast31
            x += dpp_perm<0x143>(x);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(x));                      // ... and ends before the second one
    const unsigned long long t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    out[lane] = x;
    if (lane == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_sumtest(double *out) {
    const int lane = threadIdx.x;
    const double x = 1.0 + 0.01 * lane * lane;
    double d = x;
    d += dpp_perm<0xB1>(d);
    d += dpp_perm<0x4E>(d);
    d += dpp_perm<0x141>(d);
    d += dpp_perm<0x140>(d);
    const double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
    const double a = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, 1.0, 0.0, 0, 0, 0);
    const double b = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s1, 0.0, 0, 0, 0);
    out[lane] = d; out[64 + lane] = a; out[128 + lane] = b;
}

template <int W> int run(const char *name, double *out, unsigned long long *cyc) {
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k_chain<W>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5);
        CK(hipDeviceSynchronize());
        unsigned long long h;
        CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("%-64s %7.1f cycles per repetition\n", name, (double)best / REP);
    return 0;
}

int main() {
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 64 * 8 * 3)); CK(hipMalloc(&cyc, 8));
    run<0>("v_fma_f64 (dependent)", out, cyc);
    run<1>("v_add_f64", out, cyc);
    run<2>("2 v_mov_b32_dpp quad_perm + v_add_f64", out, cyc);
    run<3>("2 v_mov_b32_dpp row_mirror + v_add_f64", out, cyc);
    run<4>("2 v_permlane16_swap (+ copies) + v_add_f64", out, cyc);
    run<5>("2 v_permlane32_swap (+ copies) + v_mul_f64", out, cyc);
    run<6>("v_rsq_f64 + v_add_f64", out, cyc);
    run<7>("alpha tail: rsq, mul, mul, fma, fma, mul, min, add", out, cyc);
    run<8>("ds_write_b64 -> ds_read_b64 + add", out, cyc);
    run<9>("cvt f64->f32, v_mul_f32, cvt f32->f64", out, cyc);
    run<10>("mul + 4 DPP stages + permlane16 stage + mul (the 32-lane dot)", out, cyc);
    run<11>("2 v_readlane_b32 + v_add_f64", out, cyc);
    run<12>("v_mfma_f64_16x16x4 + add + fma", out, cyc);
    run<13>("v_mov_b64 + v_add_f64", out, cyc);
    run<14>("2 v_mov_b32_dpp row_bcast15 + v_add_f64", out, cyc);
    run<15>("2 v_mov_b32_dpp row_bcast31 + v_add_f64", out, cyc);
    run<16>("2 v_mfma_f64_4x4x4 (A, then A) + fma", out, cyc);
    run<17>("2 v_mfma_f64_4x4x4 (A, then B) + fma", out, cyc);
    {   // which arrangement sums the 16 lanes of a block?
        hipLaunchKernelGGL(k_sumtest, dim3(1), dim3(64), 0, 0, out);
        CK(hipDeviceSynchronize());
        double h[64 * 3];
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        printf("lane: dpp 16-lane sum | mfma A,A | mfma A,B\n");
        for (int l = 0; l < 64; l += 5) printf("%2d: %10.4f %10.4f %10.4f\n", l, h[l], h[64 + l], h[128 + l]);
    }
    return 0;
}
