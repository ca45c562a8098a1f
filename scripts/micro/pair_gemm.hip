// Micro-benchmark of the code step's product pair (modl_amd/csrc/gemm_dense.hpp: Dx = Xs Ds^T and the symmetric Gram matrix
// Ds^T Ds, split along K, one launch + one for both split-K sums) with 64 x 64 and with 128 x 128 workgroup tiles.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I modl_amd/csrc -I include scripts/micro/pair_gemm.hip -o scripts/micro/pair_gemm
// Run:   scripts/micro/pair_gemm [K] [k] [b]
#include "gemm_dense.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace modl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BMT> int run(const float *Xs, int64_t ldx, const float *Ds, int b, int k, int64_t K, float *Dx, float *G, float *ws, size_t ws_elems,
                           int target, int reps, float *ms_out) {
    DenseOperand A; A.ptr = Xs; A.si = ldx; A.sk = 1;             // (i = sample, kk = feature)
    DenseOperand B; B.ptr = Ds; B.si = 1; B.sk = k;               // (i = atom, kk = feature)
    EpiStore<float> eD{Dx, k, 1.0f}, eG{G, k, 1.0f};
    const size_t half = ws_elems / 2;
    auto PD = plan_dense<float, EpiStore<float>>(A, B, b, k, K, eD, ws, half, target, 64, BMT, BMT);
    auto PG = plan_dense<float, EpiStore<float>>(B, B, k, k, K, eG, ws + half, ws_elems - half, target, 64, BMT, BMT, true);
    if (!PD.ok || !PG.ok) { printf("not eligible\n"); return 1; }
    printf("  tiles %d: Dx %d tiles x %d splits, Gram %d tiles x %d splits (%d + %d workgroups)\n", BMT, PD.tiles(), PD.splits, PG.tiles(), PG.splits,
           PD.tiles() * PD.splits, PG.tiles() * PG.splits);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i)
            if (launch_gemm_dense_pair<float, false, true, EpiStore<float>, true, true, EpiStore<float>, BMT, BMT, 0, BMT, BMT>(0, PD, PG) != 0) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms_out, e0, e1));
        *ms_out /= reps;
    }
    return 0;
}

int main(int argc, char **argv) {
    const int64_t K = argc > 1 ? atoll(argv[1]) : 10000;
    const int k = argc > 2 ? atoi(argv[2]) : 256, b = argc > 3 ? atoi(argv[3]) : 256;
    const int target = argc > 4 ? atoi(argv[4]) : 256;
    const int64_t ldx = (K + 3) / 4 * 4;
    std::vector<float> hX((size_t)b * ldx), hD((size_t)K * k);
    srand(1);
    for (auto &v : hX) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    for (auto &v : hD) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    float *X, *D, *Dx[2], *G[2], *ws;
    const size_t ws_elems = (size_t)64 << 20;
    CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&D, hD.size() * 4)); CK(hipMalloc(&ws, ws_elems * 4));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&Dx[v], (size_t)b * k * 4)); CK(hipMalloc(&G[v], (size_t)k * k * 4)); }
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice));
    float ms64 = 0, ms128 = 0;
    printf("K=%lld k=%d b=%d, %d workgroups per product\n", (long long)K, k, b, target);
    if (run<64>(X, ldx, D, b, k, K, Dx[0], G[0], ws, ws_elems, target, 20, &ms64)) return 1;
    if (run<128>(X, ldx, D, b, k, K, Dx[1], G[1], ws, ws_elems, target, 20, &ms128)) return 1;
    CK(hipDeviceSynchronize());
    std::vector<float> a((size_t)b * k), c((size_t)b * k), g0((size_t)k * k), g1((size_t)k * k);
    CK(hipMemcpy(a.data(), Dx[0], a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), Dx[1], c.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g0.data(), G[0], g0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), G[1], g1.size() * 4, hipMemcpyDeviceToHost));
    // reference of a few entries in double
    double num = 0, den = 0, numg = 0, deng = 0, asym = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += (double)(a[i] - c[i]) * (a[i] - c[i]); den += (double)a[i] * a[i]; }
    for (size_t i = 0; i < g0.size(); ++i) { numg += (double)(g0[i] - g1[i]) * (g0[i] - g1[i]); deng += (double)g0[i] * g0[i]; }
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) asym += fabs((double)g1[i * k + j] - g1[j * k + i]);
    double ref = 0; for (int64_t f = 0; f < K; ++f) ref += (double)hX[5 * ldx + f] * hD[f * k + 7];
    printf("  128 against 64: Dx %.2e, Gram %.2e; Gram asymmetry (128) %.1e; Dx[5][7] %.6f / %.6f, f64 %.6f\n", sqrt(num / den), sqrt(numg / deng), asym,
           a[5 * k + 7], c[5 * k + 7], ref);
    const double fl = 2.0 * b * k * K + 2.0 * k * k * K;
    printf("  64 x 64: %.2f us per pair + reduce (%.1f TFLOP/s of the full products);  128 x 128: %.2f us (%.1f)\n", ms64 * 1e3, fl / (ms64 * 1e-3) * 1e-12,
           ms128 * 1e3, fl / (ms128 * 1e-3) * 1e-12);
    return 0;
}
