// Micro-check: global_load_lds_dwordx4 (gfx950) from inline assembly - one wave-instruction moves one 1 KiB row
// into LDS at M0 + 16 * lane.  Prints PASS when four rows read back as written.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_row lds_dma_row.hip && ./lds_dma_row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *Q, float *out) {
    __shared__ __attribute__((aligned(16))) float ring[4][256];
    const int lane = threadIdx.x;
    unsigned int off = lane * 16;
    // the instruction offset moves BOTH addresses: one M0 / address set-up serves four consecutive 1 KiB pieces
    unsigned int lds = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) float *)&ring[0][0];
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(lds), "v"(off), "s"(Q) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    typedef __attribute__((address_space(3))) volatile float lv;
    lv *r = (lv *)&ring[0][0];
    for (int j = 0; j < 4; ++j)
        for (int c = 0; c < 4; ++c) out[j * 256 + lane * 4 + c] = r[j * 256 + lane * 4 + c];
}
int main() {
    std::vector<float> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i * 0.5f + 1.0f;
    float *dQ, *dO;
    hipMalloc(&dQ, 4096); hipMalloc(&dO, 4096);
    hipMemcpy(dQ, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dQ, dO);
    hipMemcpy(o.data(), dO, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += o[i] != h[i];
    printf(bad ? "FAIL %d\n" : "PASS %d\n", bad);
    return bad != 0;
}
