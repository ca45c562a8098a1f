// Micro-benchmark of the register-resident statistics product (modl_amd/csrc/gemm_resident.hpp) on its own: time per launch,
// shader-clock stamps of workgroup 0 (wavefronts 0 and 4), result against a plain kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I modl_amd/csrc -I include scripts/micro/res_gemm.hip -o scripts/micro/res_gemm
// Run:   scripts/micro/res_gemm [p] [k] [b]
#include "gemm_resident.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace modl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> struct EpiS {             // (somf_step.hip's EpiStats)
    static constexpr bool rmw = true;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    T *out; int64_t ld; T beta, wt, bdiv; int replace; T *mirror;
    __device__ __forceinline__ static bool pow2(T b) { return (__float_as_uint(b) & 0x007fffffu) == 0 && b > (T)0 && b < (T)1e30; }
    __device__ __forceinline__ T value(T v, T old, bool p2, T rinv) const {
        const T x = replace ? v : wt * v;
        const T q = p2 ? x * rinv : x / bdiv;
        return replace ? q : old * beta + q;
    }
    __device__ __forceinline__ T load(int64_t m, int64_t n) const { return out[m * ld + n]; }
    __device__ __forceinline__ void store(int64_t m, int64_t n, T v, T old) const {
        const T nv = value(v, old, pow2(bdiv), (T)1 / bdiv);
        out[m * ld + n] = nv;
        if (mirror) mirror[m * ld + n] = nv;
    }
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { store(m, n, v, load(m, n)); }
    bool vec4_ok() const { return ld % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(mirror) % 16 == 0; }
    __device__ __forceinline__ vec4 load4(int64_t m, int64_t n) const { return *reinterpret_cast<const vec4 *>(out + m * ld + n); }
    __device__ __forceinline__ void store4(int64_t m, int64_t n, vec4 v, vec4 old) const {
        const bool p2 = pow2(bdiv);
        const T rinv = (T)1 / bdiv;
        vec4 nv;
#pragma unroll
        for (int c = 0; c < 4; ++c) nv[c] = value(v[c], old[c], p2, rinv);
        *reinterpret_cast<vec4 *>(out + m * ld + n) = nv;
        if (mirror) *reinterpret_cast<vec4 *>(mirror + m * ld + n) = nv;
    }
};

__global__ void ref_kernel(const float *X, int64_t ldx, const float *Cd, int ldc, int64_t M, int N, int K, const float *old, float *out,
                           float beta, float wt, float bdiv) {
    const int64_t m = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double s = 0;
        for (int kk = 0; kk < K; ++kk) s += (double)X[kk * ldx + m] * (double)Cd[kk * ldc + n];
        out[m * N + n] = old[m * N + n] * beta + (wt * (float)s) / bdiv;
    }
}

int main(int argc, char **argv) {
    const int64_t p = argc > 1 ? atoll(argv[1]) : 10000;
    const int k = argc > 2 ? atoi(argv[2]) : 256, b = argc > 3 ? atoi(argv[3]) : 256, ft = argc > 4 ? atoi(argv[4]) : 16;
    std::vector<float> hX((size_t)b * p), hC((size_t)b * k), hB((size_t)p * k);
    srand(1);
    for (auto &v : hX) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hC) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hB) v = (float)rand() / RAND_MAX - 0.5f;
    float *X, *Cd, *B0, *B, *R, *Cc;
    unsigned long long *dbg;
    CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&Cd, hC.size() * 4)); CK(hipMalloc(&B0, hB.size() * 4));
    CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&R, hB.size() * 4)); CK(hipMalloc(&Cc, (size_t)k * k * 4));
    CK(hipMalloc(&dbg, 128 * 8));
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Cd, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B0, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Cc, 0, (size_t)k * k * 4));
    CK(hipMemset(dbg, 0, 128 * 8));
    const float beta = 0.9f, wt = 0.1f, bdiv = (float)b;
    DenseOperand A1; A1.ptr = X; A1.si = 1; A1.sk = p;
    DenseOperand B1; B1.ptr = Cd; B1.si = 1; B1.sk = k;
    EpiS<float> eB{B, k, beta, wt, bdiv, 0, nullptr}, eC{Cc, k, beta, wt, bdiv, 0, nullptr};
    auto P0 = plan_stats<EpiS<float>>(B1, B1, k, k, b, eC);
    auto W = plan_wide<16, EpiS<float>>(A1, B1, p, k, b, eB);
    if (!W.ok || !P0.ok) { printf("not eligible\n"); return 1; }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)p), dim3(256), 0, 0, X, p, Cd, k, p, k, b, B0, R, beta, wt, bdiv);
    CK(hipMemcpy(B, B0, hB.size() * 4, hipMemcpyDeviceToDevice));
    W.dbg = dbg;
    if ((ft == 16 ? launch_gemm_stats_resident_pair<16>(0, P0, W, ncu) : launch_gemm_stats_resident_pair<32>(0, P0, W, ncu)) != 0) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hR(hB.size()), hO(hB.size());
    CK(hipMemcpy(hR.data(), R, hB.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hO.data(), B, hB.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < hR.size(); ++i) { num += (double)(hR[i] - hO[i]) * (hR[i] - hO[i]); den += (double)hR[i] * hR[i]; }
    printf("p=%lld k=%d b=%d: rel error against the plain kernel %.2e\n", (long long)p, k, b, sqrt(num / den));
    std::vector<unsigned long long> st(128);
    CK(hipMemcpy(st.data(), dbg, 128 * 8, hipMemcpyDeviceToHost));
    for (int w = 0; w < 2; ++w) {
        printf("wavefront %d of workgroup 0 (cycles since its first stamp):", 4 * w);
        for (int i = 0; i < 64 && st[64 * w + i]; ++i) printf(" %llu", st[64 * w + i] - st[64 * w]);
        printf("\n");
    }
    printf("workgroup 0: kernel entry -> first stamp of wavefront 0 %lld, -> loop done %lld, -> C_ tiles done %lld;  last workgroup: loop done %lld, C_ tile done %lld\n",
           (long long)(st[0] - st[112]), (long long)(st[113] - st[112]), (long long)(st[114] ? st[114] - st[112] : 0), (long long)(st[121] - st[120]),
           (long long)(st[122] ? st[122] - st[120] : 0));
    W.dbg = nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) { if (ft == 16) launch_gemm_stats_resident_pair<16>(0, P0, W, ncu); else launch_gemm_stats_resident_pair<32>(0, P0, W, ncu); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.2f us per launch (%.1f TFLOP/s)\n", ms / 20 * 1e3, 2.0 * p * k * b / (ms / 20 * 1e-3) * 1e-12);
    }
    return 0;
}
