// Micro-benchmark: the alpha recursion of one 32-atom block (bcd.hip: resolve_wave) on one wavefront alone on its SIMD:
// (A) the single-wave form (chain + partial sums), (B) the chain alone, partial sums arriving through an LDS mailbox
// (flags pre-set: what the chain wave would cost if a helper wave always delivered on time).
// Build: hipcc -O3 --offload-arch=gfx950 -I modl_amd/csrc scripts/micro/resolve_lat.hip -o scripts/micro/resolve_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "common.hpp"
using namespace modl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kNB = 32;

__device__ __forceinline__ void halves(double z, double &low, double &high) {
    const long long b = __double_as_longlong(z);
    const unsigned int w0 = (unsigned int)(b & 0xffffffffll), w1 = (unsigned int)(b >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(w1, w1, false, false);
    low = __longlong_as_double(((long long)r1[0] << 32) | r0[0]);
    high = __longlong_as_double(((long long)r1[1] << 32) | r0[1]);
}

// LDS mailbox words: volatile accesses that keep their address space (a generic volatile pointer turns into flat
// accesses with sc0 sc1 and a wait behind each)
typedef __attribute__((address_space(3))) volatile double lds_vf64;
typedef __attribute__((address_space(3))) volatile int lds_vi32;
// (B) chain only
__device__ __forceinline__ void chain_wave(const double *Cs, const double *scr_in, double *scr_out, double *Pm_,
                                           int *pcount_, double *Zm_, int *zcount_, double *CAout) {
    lds_vf64 *Pm = (lds_vf64 *)Pm_;
    lds_vf64 *Zm = (lds_vf64 *)Zm_;
    lds_vi32 *pcount = (lds_vi32 *)pcount_;
    lds_vi32 *zcount = (lds_vi32 *)zcount_;
    const int lane = threadIdx.x & 63;
    double al_prev = 0.0, q_prev = 0.0, z_prev = 0.0, Zm1 = 0.0, Zm2 = 0.0;
    const double lmask = lane < 32 ? 1.0 : 0.0;
    (void)lmask;
    int rn = *pcount;
    double Pn = Pm[lane];
    while (__builtin_amdgcn_readfirstlane(rn) < 1) { rn = *pcount; Pn = Pm[lane]; }
    double srn = scr_in[0], capn = scr_in[1];
    double c2n = 0.0, c1nn = Cs[1 * kNB + 0];
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        const double P = Pn, sr = srn, cap = capn, c2 = c2n, c1n = c1nn;
        if (j + 1 < kNB) {                       // everything step j + 1 needs from LDS is requested now
            rn = *pcount;
            Pn = Pm[((j + 1) & 7) * 64 + lane];
            srn = scr_in[2 * (j + 1)]; capn = scr_in[2 * (j + 1) + 1];
            c2n = (j + 1 >= 2) ? Cs[(j + 1) * kNB + j - 1] : 0.0;
            c1nn = (j + 2 < kNB) ? Cs[(j + 2) * kNB + j + 1] : 0.0;
        }
        const double part = __builtin_fma(-c2, Zm2, P);
        const double z = __builtin_fma(-al_prev, q_prev, part);
        double t, w;
        halves(z, t, w);
#ifdef MFMA_SUM
        // 64-lane sum: two 4x4x4 f64 matrix-core products (sum over lane % 4 and the four rows of 16 lanes; the B operand
        // of the first one is 1 in rows 0-1 and 0 in rows 2-3, which hold duplicates), then the four quads of a row
        double pr = t * w;
        pr = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, lmask, 0.0, 0, 0, 0);
        pr = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, 1.0, 0.0, 0, 0, 0);
        pr += dpp_perm<0x141>(pr);
        pr += dpp_perm<0x140>(pr);
        const double nrm = pr;
#else
        double pr = t * w;
        pr += dpp_perm<0xB1>(pr);
        pr += dpp_perm<0x4E>(pr);
        pr += dpp_perm<0x141>(pr);
        pr += dpp_perm<0x140>(pr);
        double r0, r1;
        lane_swap<true>(pr, r0, r1);
        const double nrm = r0 + r1;
#endif
        if (j > 0) {
            const double Zp = al_prev * z_prev;
            Zm[((j - 1) & 7) * 64 + lane] = Zp;
            *zcount = j;
            Zm2 = Zm1; Zm1 = Zp;
            (void)CAout;
        }
        const double q = c1n * z;
        const double y = __builtin_amdgcn_rsq(nrm);
        const double r = __builtin_fma(-(0.5 * y), nrm * y, 0.5);
        const double yn = __builtin_fma(y, r, y);
        double al;
        const double sy = sr * yn;
        asm("v_min_f64 %0, %1, %2" : "=v"(al) : "v"(sy), "v"(cap));
        scr_out[2 * j] = al;
        scr_out[2 * j + 1] = nrm;
        al_prev = al; q_prev = q; z_prev = z;
        if (j + 1 < kNB)
            while (__builtin_amdgcn_readfirstlane(rn) < j + 2) { rn = *pcount; Pn = Pm[((j + 1) & 7) * 64 + lane]; }
    }
    Zm[((kNB - 1) & 7) * 64 + lane] = al_prev * z_prev;
    *zcount = kNB;
}

__global__ __launch_bounds__(64) void k_chain(double *out, unsigned long long *cyc) {
    __shared__ double Cs[kNB * kNB], scr[4 * kNB], Pm[8 * 64], Zm[8 * 64];
    __shared__ int pc, zc;
    const int lane = threadIdx.x;
    for (int e = lane; e < kNB * kNB; e += 64) Cs[e] = ((e / kNB) > (e % kNB)) ? 0.01 * ((e * 7) % 13 - 6) : 0.0;
    for (int e = lane; e < 2 * kNB; e += 64) scr[e] = (e & 1) ? 1.0 : 0.9;
    for (int e = lane; e < 8 * 64; e += 64) Pm[e] = 1.0 + 0.001 * e;
    if (lane == 0) { pc = 1000; zc = 0; }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    chain_wave(Cs, scr, scr + 2 * kNB, Pm, &pc, Zm, &zc, nullptr);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = clock64();
    out[lane] = Zm[lane] + scr[2 * kNB + lane];
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, cyc);
        CK(hipDeviceSynchronize());
        unsigned long long h;
        CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    double h[64];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("chain wave alone, mailbox always ready: %llu cycles per block = %.1f per atom   (out[0] = %.15g, out[37] = %.15g)\n", best, (double)best / kNB, h[0], h[37]);
    return 0;
}
