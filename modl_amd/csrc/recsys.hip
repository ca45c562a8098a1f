// Masked (missing-data) path of RecsysDictFact on CSR ratings.
//
// Replaces the per-sample Python loop of the reference
//   (modl/decomposition/recsys.py:147-185 _single_batch_fit / _single_sample_update,
//    :254-265 _refit, and recsys_fast.pyx:10-38 _predict).
// For a row i with observed columns S_i and values x_S:
//     G = D_S D_S^T + (alpha |S_i| / p) I,   code_i = G^{-1} D_S x_S               (:176-181)
//     w_B[f] = min(1, w n_iter / feature_n_iter[f]),  B[:, f] = (1 - w_B) B[:, f] + code_i (x_f w_B)   (:182-185)
// The dictionary is fixed inside a minibatch, so the b code solves are independent (one workgroup per
// row: Gram accumulated in LDS from the feature-major dictionary rows — a rated item is one contiguous
// k-vector — then an in-LDS Cholesky); the B update is order-dependent per feature (two rows of the
// batch rating the same item do not commute), so it runs one wavefront per touched feature, walking
// that feature's entries in batch order.  C_ and the block-coordinate update on the union of the
// batch's columns (recsys.py:159-165,187-213: squared-l2 budget, clip-only projection) reuse the
// kernels of the dense path (csrc/bcd.hip blocked path is exactly that update).
#include "gemm.hpp"
#include "kernels.hpp"

namespace modl {

template <typename T>
__global__ __launch_bounds__(256) void recsys_code_kernel(const T *Dt, int64_t p, int k, const int32_t *indptr,
                                                          const int32_t *indices, const T *data, const int64_t *row_ids,
                                                          const int64_t *code_rows, double alpha, T *code) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);                 // [k][k]
    T *rhs = G + (size_t)k * k;                             // [k]
    T *rows = rhs + k;                                      // [32][k] staged dictionary rows
    T *xv = rows + 32 * (size_t)k;                          // [32]
    const int64_t r = row_ids ? row_ids[blockIdx.x] : (int64_t)blockIdx.x;
    const int32_t beg = indptr[r], end = indptr[r + 1];
    const int nnz = end - beg;
    if (nnz == 0) return;                                   // recsys.py:170: rows without ratings keep their code
    for (int e = threadIdx.x; e < k * k + k; e += 256) G[e] = 0;   // G and rhs are contiguous
    for (int c0 = 0; c0 < nnz; c0 += 32) {
        const int nc = (nnz - c0 < 32) ? nnz - c0 : 32;
        __syncthreads();
        for (int e = threadIdx.x; e < nc * k; e += 256) {
            const int j = e / k, c = e % k;
            rows[j * k + c] = Dt[(int64_t)indices[beg + c0 + j] * k + c];
        }
        if (threadIdx.x < nc) xv[threadIdx.x] = data[beg + c0 + threadIdx.x];
        __syncthreads();
        for (int e = threadIdx.x; e < k * k; e += 256) {
            const int a = e / k, c = e % k;
            T acc = G[e];
            for (int j = 0; j < nc; ++j) acc = fma(rows[j * k + a], rows[j * k + c], acc);
            G[e] = acc;
        }
        for (int a = threadIdx.x; a < k; a += 256) {
            T acc = rhs[a];
            for (int j = 0; j < nc; ++j) acc = fma(rows[j * k + a], xv[j], acc);
            rhs[a] = acc;
        }
    }
    __syncthreads();
    const T ridge = (T)(alpha * (double)nnz / (double)p);   // alpha / reduction, reduction = p / |S_i| (:175,179)
    for (int a = threadIdx.x; a < k; a += 256) G[a * k + a] += ridge;
    __syncthreads();
    // Cholesky G = L L^T in place (symmetric storage), then L y = rhs, L^T x = y
    for (int j = 0; j < k; ++j) {
        __syncthreads();
        const T d = sqrt(G[j * k + j]);
        __syncthreads();
        for (int i = j + 1 + threadIdx.x; i < k; i += 256) {
            const T l = G[j * k + i] / d;
            G[j * k + i] = l;
            G[i * k + j] = l;
        }
        if (threadIdx.x == 0) G[j * k + j] = d;
        __syncthreads();
        const int n = k - j - 1;
        for (int e = threadIdx.x; e < n * n; e += 256) {
            const int i = e / n, m = e % n;
            T *t = G + (size_t)(j + 1 + i) * k + (j + 1 + m);
            *t = fma(-G[j * k + j + 1 + i], G[j * k + j + 1 + m], *t);
        }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {                          // forward
        if (threadIdx.x == 0) rhs[j] = rhs[j] / G[j * k + j];
        __syncthreads();
        const T yj = rhs[j];
        for (int i = j + 1 + threadIdx.x; i < k; i += 256) rhs[i] = fma(-yj, G[j * k + i], rhs[i]);
        __syncthreads();
    }
    for (int j = k - 1; j >= 0; --j) {                     // backward
        if (threadIdx.x == 0) rhs[j] = rhs[j] / G[j * k + j];
        __syncthreads();
        const T xj = rhs[j];
        for (int i = threadIdx.x; i < j; i += 256) rhs[i] = fma(-xj, G[j * k + i], rhs[i]);
        __syncthreads();
    }
    T *out = code + (code_rows ? code_rows[blockIdx.x] : r) * k;
    for (int a = threadIdx.x; a < k; a += 256) out[a] = rhs[a];
}

// One wavefront per touched feature; entries of that feature in batch order.
template <typename T>
__global__ __launch_bounds__(256) void recsys_update_B_kernel(T *Bt, int k, int64_t *feature_n_iter, const int32_t *subset,
                                                              const int32_t *fptr, const int32_t *entry_sample,
                                                              const T *entry_val, const T *code_b, double w_n_iter,
                                                              int64_t u) {
    const int lane = threadIdx.x & 63;
    const int64_t fi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fi >= u) return;
    const int32_t f = subset[fi];
    int64_t n = feature_n_iter[f];
    T *brow = Bt + (int64_t)f * k;
    for (int32_t e = fptr[fi]; e < fptr[fi + 1]; ++e) {
        n += 1;                                              // recsys.py:175 feature_n_iter_[subset] += 1
        double wB = w_n_iter / (double)n;                    // :182-183
        wB = wB < 1.0 ? wB : 1.0;
        const double xw = (double)entry_val[e] * wB;
        const T *cr = code_b + (int64_t)entry_sample[e] * k;
        for (int c = lane; c < k; c += 64) {
            T b = (T)((double)brow[c] * (1.0 - wB));         // B_[:, subset] *= 1 - w_B
            b = (T)((double)b + (double)cr[c] * xw);          // += outer(code, X_subset * w_B)
            brow[c] = b;
        }
    }
    if (lane == 0) feature_n_iter[f] = n;
}

// data[ii] = sum_c code[u][c] * Dt[indices[ii]][c]     (recsys_fast.pyx:10-38 with the feature-major dictionary)
template <typename T>
__global__ __launch_bounds__(256) void recsys_predict_kernel(double *out, const int32_t *indices, const int32_t *indptr,
                                                             const T *code, int64_t n_rows, int k, const T *Dt) {
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_rows) return;
    const T *cu = code + u * k;
    for (int32_t ii = indptr[u] + lane; ii < indptr[u + 1]; ii += 64) {
        const T *dr = Dt + (int64_t)indices[ii] * k;
        double dot = 0;
        for (int c = 0; c < k; ++c) dot += (double)cu[c] * (double)dr[c];
        out[ii] = dot;
    }
}

template <typename T>
int recsys_codes(const T *Dt, int64_t p, int k, const int32_t *indptr, const int32_t *indices, const T *data,
                 const int64_t *row_ids, const int64_t *code_rows, int64_t b, double alpha, T *code, hipStream_t st) {
    if (b <= 0) return MODL_OK;
    const size_t lds = sizeof(T) * ((size_t)k * k + k + 32 * (size_t)k + 32) + 16;
    if (lds > 160 * 1024) return MODL_EINVAL;
    MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&recsys_code_kernel<T>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((recsys_code_kernel<T>), dim3((unsigned)b), dim3(256), lds, st, Dt, p, k, indptr, indices, data,
                       row_ids, code_rows, alpha, code);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

template <typename T> struct EpiAxpbyC {
    T *out; int64_t ld; T alpha, beta;
    __device__ __forceinline__ void operator()(int64_t m, int64_t n, T v) const {
        T *o = out + m * ld + n;
        *o = (*o) * beta + alpha * v;
    }
};

}  // namespace modl

using namespace modl;

extern "C" {

#define ABI_RECSYS(SFX, T)                                                                                          \
    int modl_recsys_codes_##SFX(const T *d_Dt, int64_t p, int k, const int32_t *d_indptr, const int32_t *d_indices,   \
                                const T *d_data, const int64_t *d_row_ids, const int64_t *d_code_rows, int64_t b,     \
                                double alpha, T *d_code, void *stream) {                                            \
        if (!d_Dt || !d_indptr || !d_indices || !d_data || !d_code || k <= 0 || p <= 0 || b < 0) return MODL_EINVAL;  \
        return recsys_codes<T>(d_Dt, p, k, d_indptr, d_indices, d_data, d_row_ids, d_code_rows, b, alpha, d_code,     \
                               (hipStream_t)stream);                                                                \
    }                                                                                                               \
    int modl_recsys_update_B_##SFX(T *d_Bt, int k, int64_t *d_feature_n_iter, const int32_t *d_subset,                \
                                   const int32_t *d_fptr, const int32_t *d_entry_sample, const T *d_entry_val,        \
                                   const T *d_code_b, double w_times_n_iter, int64_t u, void *stream) {             \
        if (!d_Bt || !d_feature_n_iter || !d_subset || !d_fptr || !d_entry_sample || !d_entry_val || !d_code_b ||     \
            k <= 0 || u < 0)                                                                                        \
            return MODL_EINVAL;                                                                                     \
        if (u == 0) return MODL_OK;                                                                                 \
        hipLaunchKernelGGL((recsys_update_B_kernel<T>), dim3((unsigned)cdiv(u, 4)), dim3(256), 0, (hipStream_t)stream, \
                           d_Bt, k, d_feature_n_iter, d_subset, d_fptr, d_entry_sample, d_entry_val, d_code_b,        \
                           w_times_n_iter, u);                                                                      \
        MODL_LAUNCH_CHECK();                                                                                        \
        return MODL_OK;                                                                                             \
    }                                                                                                               \
    int modl_recsys_predict_##SFX(double *d_out, const int32_t *d_indices, const int32_t *d_indptr, const T *d_code,  \
                                  int64_t n_rows, int k, const T *d_Dt, void *stream) {                             \
        if (!d_out || !d_indices || !d_indptr || !d_code || !d_Dt || n_rows < 0 || k <= 0) return MODL_EINVAL;        \
        if (n_rows == 0) return MODL_OK;                                                                            \
        hipLaunchKernelGGL((recsys_predict_kernel<T>), dim3((unsigned)cdiv(n_rows, 4)), dim3(256), 0,                 \
                           (hipStream_t)stream, d_out, d_indices, d_indptr, d_code, n_rows, k, d_Dt);                \
        MODL_LAUNCH_CHECK();                                                                                        \
        return MODL_OK;                                                                                             \
    }                                                                                                               \
    /* C = beta C + alpha rows^T rows  (recsys.py:159-160 with beta = 1 - w, alpha = w / b) */                      \
    int modl_gram_axpby_##SFX(const T *d_rows, int64_t b, int k, T *d_C, T beta, T alpha, void *stream) {            \
        if (!d_rows || !d_C || b < 0 || k <= 0) return MODL_EINVAL;                                                  \
        Operand A;                                                                                                  \
        A.ptr = d_rows; A.si = 1; A.sk = k;                                                                         \
        EpiAxpbyC<T> epi{d_C, k, alpha, beta};                                                                      \
        SplitWs none;                                                                                               \
        return launch_gemm<T, EpiAxpbyC<T>>((hipStream_t)stream, A, A, k, k, b, epi, none, nullptr, 512, 1);         \
    }                                                                                                               \
    /* _update_dict on device-resident state (dict_fact.py:650-715 / recsys.py:187-213) */                         \
    int modl_dict_update_##SFX(T *d_Dt, const T *d_Bt, const T *d_C, T *d_comp_norm, const int32_t *d_subset,         \
                               int64_t s, const int32_t *d_order, const int64_t *h_order, int k, int optimizer,       \
                               int comp_pos, double comp_l1_ratio, double w, double step_size, void *d_ws,          \
                               size_t ws_bytes, void *stream) {                                                     \
        if (!d_Dt || !d_Bt || !d_C || !d_comp_norm || !d_order || !h_order || s < 0 || k <= 0) return MODL_EINVAL;    \
        DictUpdateArgs<T> a;                                                                                        \
        a.Dt = d_Dt; a.Bt = d_Bt; a.C = d_C; a.comp_norm = d_comp_norm; a.subset = d_subset; a.order = d_order;       \
        a.h_order = h_order; a.s = s; a.k = k; a.optimizer = optimizer; a.comp_pos = comp_pos;                        \
        a.comp_l1_ratio = comp_l1_ratio; a.w = w; a.step_size = step_size; a.ws = d_ws; a.ws_bytes = ws_bytes;        \
        int nl = 0;                                                                                                 \
        return dict_update<T>((hipStream_t)stream, a, &nl);                                                         \
    }
ABI_RECSYS(f32, float)
ABI_RECSYS(f64, double)
#undef ABI_RECSYS

size_t modl_dict_update_workspace(int dtype, int64_t s_max, int k) { return dict_update_workspace(dtype, s_max, k); }

}  // extern "C"
