// what sits either side of the SOMF step.
//   image patch pipeline  : modl/feature_extraction/image.py:54-63 (LazyCleanPatchExtractor.partial_transform: a fancy
//                           index into the sliding-window view of the image) fused with
//                           modl/input_data/image.py:4-23 (scale_patches, channel-wise) and the flattening of
//                           modl/decomposition/image.py:190-199 -- the image stays resident in HBM, a minibatch
//                           buffer of flattened, centred, normalised patches is produced by one launch;
//   fill / clean_mask     : modl/input_data/image_fast.pyx:12-74 (host, integer work);
//   objective terms       : modl/decomposition/dict_fact.py:94-114 (CodingMixin.score) on device-resident X, codes
//                           and dictionary, so that scoring callbacks need no D2H copy of the dictionary.
#include "gemm.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <vector>

namespace modl {

// ---- patches ------------------------------------------------------------------------------------------------------
// One wavefront per patch.  A patch is the window image[i : i + x, j : j + y, c0 : c0 + z]; its flattened row is
// (x, y, z) in C order.  Channel statistics are taken over the x * y positions of each channel (numpy: axis=(1, 2)):
//   with_mean: v -= mean_c;   with_std: v /= (sqrt(sum_c v^2) or 1 if 0) * sqrt(z)
// The image (a few MB) is L2-resident; the row is written once, coalesced, in flat element order.
constexpr int kPatchMaxChannels = 1024;

template <typename T>
__global__ __launch_bounds__(256) void image_patches_kernel(const T *__restrict__ img, int64_t W, int64_t C,
                                                            const int64_t *__restrict__ idx3, int64_t n, int x, int y,
                                                            int z, int with_mean, int with_std, T sqrt_z,
                                                            T *__restrict__ out, int64_t ldo) {
    __shared__ T s_mean[4][kPatchMaxChannels];
    __shared__ T s_den[4][kPatchMaxChannels];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    if (row >= n) return;                                     // wave-uniform; no block barrier below
    const int64_t i0 = idx3[row * 3 + 0], j0 = idx3[row * 3 + 1], c0 = idx3[row * 3 + 2];
    const T *base = img + (i0 * W + j0) * C + c0;
    const int xy = x * y;
    if (with_mean || with_std) {
        for (int c = 0; c < z; ++c) {
            T mean = 0;
            if (with_mean) {
                T s = 0;
                for (int pos = lane; pos < xy; pos += 64) s += base[((int64_t)(pos / y) * W + pos % y) * C + c];
                mean = wave_sum(s) / (T)xy;
            }
            T den = 1;
            if (with_std) {
                T s = 0;
                for (int pos = lane; pos < xy; pos += 64) {
                    const T v = base[((int64_t)(pos / y) * W + pos % y) * C + c] - mean;
                    s += v * v;
                }
                T sd = sqrt(wave_sum(s));
                if (sd == (T)0) sd = 1;
                den = sd * sqrt_z;
            }
            if (lane == 0) { s_mean[wid][c] = mean; s_den[wid][c] = den; }
        }
    }
    __builtin_amdgcn_wave_barrier();                          // LDS operations of one wavefront complete in order
    T *o = out + row * ldo;
    const int P = xy * z;
    const int yz = y * z;
    for (int e = lane; e < P; e += 64) {
        const int xi = e / yz, rem = e - xi * yz;             // rem = yi * z + c: contiguous in the image row
        const int c = rem % z;
        T v = base[(int64_t)xi * W * C + (int64_t)(rem / z) * C + c];
        if (with_mean) v -= s_mean[wid][c];
        if (with_std) v /= s_den[wid][c];
        o[e] = v;
    }
}

template <typename T>
int launch_image_patches(hipStream_t stream, const T *img, int64_t H, int64_t W, int64_t C, const int64_t *idx3,
                         int64_t n, int x, int y, int z, int with_mean, int with_std, T *out, int64_t ldo) {
    if (n <= 0) return MODL_OK;
    hipLaunchKernelGGL((image_patches_kernel<T>), dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, img, W, C, idx3, n,
                       x, y, z, with_mean, with_std, (T)sqrt((double)z), out, ldo);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

// ---- objective ----------------------------------------------------------------------------------------------------
template <typename T> struct EpiResidual {     // R[m][n] = X[m][n] - v
    const T *X; int64_t ldx; T *R; int64_t ldr;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const { R[m * ldr + n] = X[m * ldx + n] - v; }
};

constexpr int kObjBlocks = 1024;

// part[b] = sum over the elements owned by block b of f(a): MODE 0 a^2, 1 |a|.  f64 accumulation, fixed order.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void obj_partial_kernel(const T *a, int64_t rows, int64_t cols, int64_t ld,
                                                          double *part) {
    __shared__ double red[4];
    const int64_t total = rows * cols;
    double s = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const double v = (double)a[(e / cols) * ld + e % cols];
        s += MODE == 0 ? v * v : fabs(v);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void obj_final_kernel(const double *part, int m, double *out, int accumulate) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < m; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = accumulate ? *out + s : s;
}

template <typename T>
int objective_impl(hipStream_t stream, const T *X, int64_t ldx, int64_t n, int64_t p, const T *Dt, int k, const T *code,
                   void *ws, size_t ws_bytes, double *out3) {
    if (ws_bytes < sizeof(double) * kObjBlocks + sizeof(T) * (size_t)p) return MODL_ENOMEM;
    double *part = static_cast<double *>(ws);
    T *R = reinterpret_cast<T *>(part + kObjBlocks);
    const int64_t chunk = (int64_t)((ws_bytes - sizeof(double) * kObjBlocks) / (sizeof(T) * (size_t)p));
    MODL_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(double), stream));
    if (n == 0) return MODL_OK;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = (n - r0 < chunk) ? n - r0 : chunk;
        Operand A, B;
        A.ptr = code + r0 * k; A.si = k; A.sk = 1;
        B.ptr = Dt; B.si = k; B.sk = 1;
        EpiResidual<T> epi{X + r0 * ldx, ldx, R, p};
        MODL_TRY((launch_gemm<T, EpiResidual<T>>(stream, A, B, rows, p, k, epi, SplitWs{}, nullptr, 512, 1)));
        hipLaunchKernelGGL((obj_partial_kernel<T, 0>), dim3(kObjBlocks), dim3(256), 0, stream, R, rows, p, p, part);
        hipLaunchKernelGGL(obj_final_kernel, dim3(1), dim3(256), 0, stream, part, kObjBlocks, out3, 1);
        MODL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((obj_partial_kernel<T, 1>), dim3(kObjBlocks), dim3(256), 0, stream, code, n, (int64_t)k, (int64_t)k, part);
    hipLaunchKernelGGL(obj_final_kernel, dim3(1), dim3(256), 0, stream, part, kObjBlocks, out3 + 1, 0);
    hipLaunchKernelGGL((obj_partial_kernel<T, 0>), dim3(kObjBlocks), dim3(256), 0, stream, code, n, (int64_t)k, (int64_t)k, part);
    hipLaunchKernelGGL(obj_final_kernel, dim3(1), dim3(256), 0, stream, part, kObjBlocks, out3 + 2, 0);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

}  // namespace modl

using namespace modl;

extern "C" {

int modl_image_fill(int64_t p, int64_t q, int64_t r, int64_t *h_out) {
    if (p < 0 || q < 0 || r < 0 || (!h_out && p * q * r > 0)) return MODL_EINVAL;
    int64_t l = 0;
    for (int64_t pp = 0; pp < p; ++pp)
        for (int64_t qq = 0; qq < q; ++qq)
            for (int64_t rr = 0; rr < r; ++rr) {
                h_out[3 * l] = pp; h_out[3 * l + 1] = qq; h_out[3 * l + 2] = rr;
                ++l;
            }
    return MODL_OK;
}

}  // extern "C"

namespace {
// image_fast.pyx:36-56.  A missing pixel (value -1) at (pp, qq, rr) clears every patch origin whose window holds it.
// The third range uses the patch WIDTH y where one would expect the depth z (image_fast.pyx:46): kept, it is what the
// reference selects (no difference whenever the patch spans all channels and z <= y, the only use in the reference).
template <typename T>
int clean_mask_impl(const T *image, int64_t H, int64_t W, int64_t C, int64_t x, int64_t y, int64_t z, int64_t *out,
                    int64_t *n_out) {
    if (!image || !n_out || x <= 0 || y <= 0 || z <= 0 || x > H || y > W || z > C) return MODL_EINVAL;
    const int64_t p = H - x + 1, q = W - y + 1, r = C - z + 1;
    std::vector<unsigned char> take((size_t)(p * q * r), 1);
    for (int64_t pp = 0; pp < H; ++pp)
        for (int64_t qq = 0; qq < W; ++qq)
            for (int64_t rr = 0; rr < C; ++rr) {
                if (image[(pp * W + qq) * C + rr] != (T)-1) continue;
                const int64_t x0 = std::max<int64_t>(0, pp - x + 1), x1 = std::min<int64_t>(p, pp + 1);
                const int64_t y0 = std::max<int64_t>(0, qq - y + 1), y1 = std::min<int64_t>(q, qq + 1);
                const int64_t z0 = std::max<int64_t>(0, rr - y + 1), z1 = std::min<int64_t>(r, rr + 1);
                for (int64_t xx = x0; xx < x1; ++xx)
                    for (int64_t yy = y0; yy < y1; ++yy)
                        for (int64_t zz = z0; zz < z1; ++zz) take[(size_t)((xx * q + yy) * r + zz)] = 0;
            }
    int64_t l = 0;
    for (int64_t pp = 0; pp < p; ++pp)
        for (int64_t qq = 0; qq < q; ++qq)
            for (int64_t rr = 0; rr < r; ++rr)
                if (take[(size_t)((pp * q + qq) * r + rr)]) {
                    if (out) { out[3 * l] = pp; out[3 * l + 1] = qq; out[3 * l + 2] = rr; }
                    ++l;
                }
    *n_out = l;
    return MODL_OK;
}
}  // namespace

extern "C" {

int modl_image_clean_mask_f32(const float *h_image, int64_t H, int64_t W, int64_t C, int64_t x, int64_t y, int64_t z,
                              int64_t *h_out, int64_t *n_out) {
    return clean_mask_impl<float>(h_image, H, W, C, x, y, z, h_out, n_out);
}
int modl_image_clean_mask_f64(const double *h_image, int64_t H, int64_t W, int64_t C, int64_t x, int64_t y, int64_t z,
                              int64_t *h_out, int64_t *n_out) {
    return clean_mask_impl<double>(h_image, H, W, C, x, y, z, h_out, n_out);
}

#define MODL_PATCH_ARGS_OK                                                                                          \
    (d_image && d_idx3 && d_out && n >= 0 && x > 0 && y > 0 && z > 0 && x <= H && y <= W && z <= C &&              \
     z <= kPatchMaxChannels && ldo >= (int64_t)x * y * z)

int modl_image_patches_f32(const float *d_image, int64_t H, int64_t W, int64_t C, const int64_t *d_idx3, int64_t n, int x,
                           int y, int z, int with_mean, int with_std, float *d_out, int64_t ldo, void *stream) {
    if (!MODL_PATCH_ARGS_OK) return MODL_EINVAL;
    return launch_image_patches<float>((hipStream_t)stream, d_image, H, W, C, d_idx3, n, x, y, z, with_mean, with_std, d_out, ldo);
}
int modl_image_patches_f64(const double *d_image, int64_t H, int64_t W, int64_t C, const int64_t *d_idx3, int64_t n, int x,
                           int y, int z, int with_mean, int with_std, double *d_out, int64_t ldo, void *stream) {
    if (!MODL_PATCH_ARGS_OK) return MODL_EINVAL;
    return launch_image_patches<double>((hipStream_t)stream, d_image, H, W, C, d_idx3, n, x, y, z, with_mean, with_std, d_out, ldo);
}

size_t modl_objective_workspace(int dtype, int64_t n, int64_t p) {
    const size_t e = dtype == MODL_F32 ? 4 : 8;
    int64_t rows = n < 1 ? 1 : n;
    if (rows > 8192) rows = 8192;
    return sizeof(double) * kObjBlocks + e * (size_t)p * (size_t)rows;
}
int modl_objective_f32(const float *d_X, int64_t ldx, int64_t n, int64_t p, const float *d_Dt, int k, const float *d_code,
                       void *d_ws, size_t ws_bytes, double *d_out3, void *stream) {
    if (!d_X || !d_Dt || !d_code || !d_ws || !d_out3 || n < 0 || p <= 0 || k <= 0 || ldx < p) return MODL_EINVAL;
    return objective_impl<float>((hipStream_t)stream, d_X, ldx, n, p, d_Dt, k, d_code, d_ws, ws_bytes, d_out3);
}
int modl_objective_f64(const double *d_X, int64_t ldx, int64_t n, int64_t p, const double *d_Dt, int k, const double *d_code,
                       void *d_ws, size_t ws_bytes, double *d_out3, void *stream) {
    if (!d_X || !d_Dt || !d_code || !d_ws || !d_out3 || n < 0 || p <= 0 || k <= 0 || ldx < p) return MODL_EINVAL;
    return objective_impl<double>((hipStream_t)stream, d_X, ldx, n, p, d_Dt, k, d_code, d_ws, ws_bytes, d_out3);
}

}  // extern "C"
