// Batched atom-geometry kernels (one workgroup per atom) and small elementwise
// helpers of the SOMF path.
//   enet_norm / enet_projection / enet_scale : modl/utils/math/enet.pyx:125,38,150
//   _update_G_average                        : modl/decomposition/dict_fact_fast.pyx:217-228
//   _predict                                 : modl/decomposition/recsys_fast.pyx:10-38
#include "enet_block.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <cstdlib>

namespace modl {

template <typename T>
__global__ __launch_bounds__(256) void enet_norm_kernel(const T *v, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                                                        T *out) {
    __shared__ double red[4];
    const double s = block_enet_norm<T>(v + (int64_t)blockIdx.x * ld, inc, n, (double)l1_ratio, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (T)s;
}

template <typename T>
__global__ __launch_bounds__(256) void enet_projection_kernel(const T *v, T *out, int64_t n, int64_t ld, int64_t inc,
                                                              const T *radius, T l1_ratio) {
    __shared__ double red[4];
    block_enet_project<T>(v + (int64_t)blockIdx.x * ld, inc, out + (int64_t)blockIdx.x * ld, inc, n,
                          (double)radius[blockIdx.x], (double)l1_ratio, red);
}

template <typename T>
__global__ __launch_bounds__(256) void enet_scale_kernel(T *v, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                                                         T radius) {
    __shared__ double red[4];
    T *x = v + (int64_t)blockIdx.x * ld;
    double l1 = 0, l2 = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double a = (double)x[i * inc];
        l1 += fabs(a);
        l2 += a * a;
    }
    l1 = block_sum(l1, red) * (double)l1_ratio;
    l2 = block_sum(l2, red) * (1.0 - (double)l1_ratio);
    double S = 0;                                            // enet.pyx:157-167
    if (l2 != 0.0) S = (-l1 + sqrt(l1 * l1 + 4.0 * (double)radius * l2)) / (2.0 * l2);
    else if (l1 != 0.0) S = (double)radius / l1;
    const T St = (T)S;
    for (int64_t i = threadIdx.x; i < n; i += 256) x[i * inc] *= St;
}

template <typename T>
int launch_enet_norm(hipStream_t stream, const T *v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                     T *out) {
    if (rows <= 0) return MODL_OK;
    hipLaunchKernelGGL((enet_norm_kernel<T>), dim3((unsigned)rows), dim3(256), 0, stream, v, n, ld, inc, l1_ratio, out);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template <typename T>
int launch_enet_projection(hipStream_t stream, const T *v, T *out, int64_t rows, int64_t n, int64_t ld,
                           int64_t inc, const T *radius, T l1_ratio) {
    if (rows <= 0) return MODL_OK;
    hipLaunchKernelGGL((enet_projection_kernel<T>), dim3((unsigned)rows), dim3(256), 0, stream, v, out, n, ld, inc,
                       radius, l1_ratio);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template <typename T>
int launch_enet_scale(hipStream_t stream, T *v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio,
                      T radius) {
    if (rows <= 0) return MODL_OK;
    hipLaunchKernelGGL((enet_scale_kernel<T>), dim3((unsigned)rows), dim3(256), 0, stream, v, n, ld, inc, l1_ratio,
                       radius);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

// G_average[idx[ii]] = (1 - w[ii]) * G_average[idx[ii]] + w[ii] * G     (idx null: rows 0..b-1)
template <typename T>
__global__ __launch_bounds__(256) void g_average_kernel(T *G_average, const int64_t *idx, const T *G, const T *w_sample,
                                                        int64_t kk) {
    const int64_t ii = blockIdx.y;
    T *ga = G_average + (idx ? idx[ii] : ii) * kk;
    const T w = w_sample[ii];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kk; e += (int64_t)gridDim.x * 256) {
        T g = ga[e];
        g = g * ((T)1 - w);
        ga[e] = g + G[e] * w;
    }
}
template <typename T>
int launch_update_G_average(hipStream_t stream, T *G_average, const int64_t *idx, const T *G, const T *w_sample,
                            int64_t b, int64_t k) {
    if (b <= 0 || k <= 0) return MODL_OK;
    const int64_t kk = k * k;
    dim3 grid((unsigned)std::min<int64_t>(cdiv(kk, 256), 64), (unsigned)b);
    hipLaunchKernelGGL((g_average_kernel<T>), grid, dim3(256), 0, stream, G_average, idx, G, w_sample, kk);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

// out[c][r] = in[r][c] through a padded LDS tile (both sides coalesced)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *in, T *out, int64_t rows, int64_t cols) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[(c0 + j) * rows + r0 + tx] = tile[tx][j];
}
template <typename T>
int launch_transpose(hipStream_t stream, const T *in, T *out, int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return MODL_OK;
    dim3 grid((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
    hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, stream, in, out, rows, cols);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

__global__ __launch_bounds__(256) void predict_csr_kernel(double *data, const int32_t *indices, const int32_t *indptr,
                                                          const double *P, int64_t n_rows, int64_t k, const double *Q,
                                                          int64_t n_cols) {
    // one wavefront per CSR row; lanes walk the row's stored entries
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_rows) return;
    for (int32_t ii = indptr[u] + lane; ii < indptr[u + 1]; ii += 64) {
        const int32_t i = indices[ii];
        double dot = 0;
        for (int64_t c = 0; c < k; ++c) dot += P[u * k + c] * Q[c * n_cols + i];
        data[ii] = dot;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t *src, uint32_t *dst, const int64_t *perm,
                                                          int64_t n, int64_t words) {
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t *s = src + perm[i] * words;
        uint32_t *d = dst + i * words;
        for (int64_t w = threadIdx.x; w < words; w += 256) d[w] = s[w];
    }
}

#define INST(T)                                                                                                   \
    template int launch_enet_norm<T>(hipStream_t, const T *, int64_t, int64_t, int64_t, int64_t, T, T *);         \
    template int launch_enet_projection<T>(hipStream_t, const T *, T *, int64_t, int64_t, int64_t, int64_t,       \
                                           const T *, T);                                                        \
    template int launch_enet_scale<T>(hipStream_t, T *, int64_t, int64_t, int64_t, int64_t, T, T);                \
    template int launch_update_G_average<T>(hipStream_t, T *, const int64_t *, const T *, const T *, int64_t,     \
                                            int64_t);                                                            \
    template int launch_transpose<T>(hipStream_t, const T *, T *, int64_t, int64_t);
INST(float)
INST(double)
#undef INST

}  // namespace modl

using namespace modl;

extern "C" {

#define ABI_ENET(SFX, T)                                                                                          \
    int modl_enet_norm_##SFX(const T *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio,          \
                             T *d_out_norm, void *stream) {                                                      \
        if (!d_v || !d_out_norm || rows < 0 || n < 0) return MODL_EINVAL;                                         \
        return launch_enet_norm<T>((hipStream_t)stream, d_v, rows, n, ld, inc, l1_ratio, d_out_norm);             \
    }                                                                                                             \
    int modl_enet_projection_##SFX(const T *d_v, T *d_out, int64_t rows, int64_t n, int64_t ld, int64_t inc,      \
                                   const T *d_radius, T l1_ratio, void *stream) {                                \
        if (!d_v || !d_out || !d_radius || rows < 0 || n < 0) return MODL_EINVAL;                                 \
        return launch_enet_projection<T>((hipStream_t)stream, d_v, d_out, rows, n, ld, inc, d_radius, l1_ratio); \
    }                                                                                                             \
    int modl_enet_scale_##SFX(T *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, T l1_ratio, T radius,     \
                              void *stream) {                                                                    \
        if (!d_v || rows < 0 || n < 0) return MODL_EINVAL;                                                        \
        return launch_enet_scale<T>((hipStream_t)stream, d_v, rows, n, ld, inc, l1_ratio, radius);                \
    }                                                                                                             \
    int modl_update_G_average_##SFX(T *d_G_average, const T *d_G, const T *d_w_sample, int64_t b, int64_t k,      \
                                    void *stream) {                                                              \
        if (!d_G_average || !d_G || !d_w_sample || b < 0 || k < 0) return MODL_EINVAL;                            \
        return launch_update_G_average<T>((hipStream_t)stream, d_G_average, nullptr, d_G, d_w_sample, b, k);      \
    }                                                                                                             \
    int modl_transpose_##SFX(const T *d_in, T *d_out, int64_t rows, int64_t cols, void *stream) {                 \
        if (!d_in || !d_out || rows < 0 || cols < 0) return MODL_EINVAL;                                          \
        return launch_transpose<T>((hipStream_t)stream, d_in, d_out, rows, cols);                                 \
    }
ABI_ENET(f32, float)
ABI_ENET(f64, double)
#undef ABI_ENET

int modl_predict_csr(double *d_data, const int32_t *d_indices, const int32_t *d_indptr, const double *d_P,
                     int64_t n_rows, int64_t k, const double *d_Q, int64_t n_cols, void *stream) {
    if (!d_data || !d_indices || !d_indptr || !d_P || !d_Q || n_rows < 0 || k < 0) return MODL_EINVAL;
    if (n_rows == 0) return MODL_OK;
    hipLaunchKernelGGL(predict_csr_kernel, dim3((unsigned)cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, d_data,
                       d_indices, d_indptr, d_P, n_rows, k, d_Q, n_cols);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

int modl_apply_swaps_rows_device(void *d_base, int64_t n, size_t row_bytes, const int64_t *h_swaps, void *stream) {
    // The swap sequence (i = n-1 .. 1, random_fast.pyx:105-119) is composed into one permutation on
    // the host; the rows are then gathered once on the device: new[i] = old[perm[i]].
    if (n < 0 || (n > 1 && (!d_base || !h_swaps))) return MODL_EINVAL;
    if (n < 2 || row_bytes == 0) return MODL_OK;
    if (row_bytes % 4 != 0) return MODL_EINVAL;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    if (!perm) return MODL_ENOMEM;
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    for (int64_t i = n - 1; i > 0; --i) {
        const int64_t j = h_swaps[i];
        if (j < 0 || j > i) { free(perm); return MODL_EINVAL; }
        const int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    void *tmp = nullptr;
    int64_t *d_perm = nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMalloc(&tmp, (size_t)n * row_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_perm, sizeof(int64_t) * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_perm, perm, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, d_base, (size_t)n * row_bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {
        const int64_t words = (int64_t)(row_bytes / 4);
        dim3 grid((unsigned)std::min<int64_t>(n, 1 << 20));
        hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, st, (const uint32_t *)tmp, (uint32_t *)d_base,
                           d_perm, n, words);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (tmp) (void)hipFree(tmp);
    if (d_perm) (void)hipFree(d_perm);
    free(perm);
    return e == hipSuccess ? MODL_OK : (int)e;
}

}  // extern "C"
