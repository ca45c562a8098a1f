// Ridge code solve: (G + alpha I) code = Dx  by Cholesky factorisation and two
// triangular substitutions.
//
// Replaces the LAPACK ?posv calls of the reference's ridge branch
//   (modl/decomposition/dict_fact_fast.pyx:82-94 per-sample Gram, :174-197
//   shared Gram).  As there, `positive`, `tol` and `max_iter` do not apply and
//   no failure is reported for a non positive-definite system (the factor then
//   holds NaNs, like an unchecked LAPACK info).
//
// gfx950 mapping: one workgroup factors one k x k system in LDS when it fits
// (k*k*sizeof(T) <= 96 KiB: k <= 156 in f32, 110 in f64; fMRI k = 70, recsys
// k = 50), otherwise in its global scratch (L2-resident).  The factor is stored
// symmetrically, F[i][j] = L[max(i,j)][min(i,j)], so that both substitutions read
// row j of F contiguously.  The substitutions run one right-hand side per
// wavefront with the same register layout as the coordinate-descent solver.
#include "kernels.hpp"

namespace modl {

template <typename T>
__device__ void cholesky_inplace(T *W, int k) {
    // right-looking, symmetric trailing update so every access is row-contiguous
    for (int j = 0; j < k; ++j) {
        __syncthreads();                                   // trailing update of column j - 1 done
        const T d = sqrt(W[(int64_t)j * k + j]);
        for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) {
            const T l = W[(int64_t)j * k + i] / d;
            W[(int64_t)j * k + i] = l;
            W[(int64_t)i * k + j] = l;
        }
        __syncthreads();                                   // scaled row visible; every read of W[j][j] is behind us,
        if (threadIdx.x == 0) W[(int64_t)j * k + j] = d;   // so the pivot can go in now (two barriers per column, not three)
        const int n = k - j - 1;
        const T *lrow = W + (int64_t)j * k + (j + 1);
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
            const int i = e / n, m = e % n;
            T *t = W + (int64_t)(j + 1 + i) * k + (j + 1 + m);
            *t = fma(-lrow[i], lrow[m], *t);
        }
    }
    __syncthreads();
}

template <typename T, bool LDS>
__global__ __launch_bounds__(1024) void cholesky_kernel(const T *G, int64_t g_stride, const int64_t *g_idx, T *F, int k,
                                                        T alpha) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const T *g = G + (g_idx ? g_idx[blockIdx.x] : (int64_t)blockIdx.x) * g_stride;
    T *f = F + (int64_t)blockIdx.x * k * k;
    T *W = LDS ? reinterpret_cast<T *>(smem_raw) : f;
    for (int e = threadIdx.x; e < k * k; e += blockDim.x) {
        T v = g[e];
        if (e / k == e % k) v += alpha;
        W[e] = v;
    }
    cholesky_inplace<T>(W, k);
    if (LDS) {
        for (int e = threadIdx.x; e < k * k; e += blockDim.x) f[e] = W[e];
    }
}

template <typename T>
int launch_cholesky(hipStream_t stream, const T *G, int64_t g_stride, const int64_t *g_idx, T *F, int k, T alpha,
                    int nmat) {
    if (nmat <= 0 || k <= 0) return MODL_OK;
    const size_t bytes = (size_t)k * k * sizeof(T);
    const int threads = k <= 64 ? 256 : 1024;
    if (bytes <= 96 * 1024) {
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cholesky_kernel<T, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        hipLaunchKernelGGL((cholesky_kernel<T, true>), dim3(nmat), dim3(threads), bytes, stream, G, g_stride, g_idx, F,
                           k, alpha);
    } else {
        hipLaunchKernelGGL((cholesky_kernel<T, false>), dim3(nmat), dim3(threads), 0, stream, G, g_stride, g_idx, F, k,
                           alpha);
    }
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_cholesky<float>(hipStream_t, const float *, int64_t, const int64_t *, float *, int, float, int);
template int launch_cholesky<double>(hipStream_t, const double *, int64_t, const int64_t *, double *, int, double, int);

template <typename T, int KPL>
__global__ __launch_bounds__(256) void chol_solve_kernel(const T *F, int64_t f_stride, T *rhs, int b, int k, T *code,
                                                         const int64_t *idx) {
    const int lane = threadIdx.x & 63;
    const int smp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (smp >= b) return;
    const T *__restrict__ Fm = F + (int64_t)smp * f_stride;
    T *r = rhs + (int64_t)smp * k;
    const int e0 = lane * KPL;
    T y[KPL], dg[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        const bool in = e0 + c < k;
        const int ec = in ? e0 + c : k - 1;
        const T yv = r[ec], dv = Fm[(int64_t)ec * k + ec];
        y[c] = in ? yv : (T)0;
        dg[c] = in ? dv : (T)1;
    }
    const int n_li = (k + KPL - 1) / KPL;
    // forward: L y = rhs, column-oriented (row j of F right of the diagonal = column j of L)
    for (int li = 0; li < n_li; ++li) {
        T rows[KPL][KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            const T *row = Fm + (int64_t)(j < k ? j : 0) * k;
#pragma unroll
            for (int c2 = 0; c2 < KPL; ++c2) {               // unconditional clamped load, then the selection
                const T rv = row[e0 + c2 < k ? e0 + c2 : k - 1];
                rows[c][c2] = (e0 + c2 < k) ? rv : (T)0;
            }
        }
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            if (j < k) {
                const T yj = bcast_lane(y[c], li) / bcast_lane(dg[c], li);
                if (lane == li) y[c] = yj;
#pragma unroll
                for (int c2 = 0; c2 < KPL; ++c2)
                    if (e0 + c2 > j) y[c2] = fma(-yj, rows[c][c2], y[c2]);
            }
        }
    }
    // backward: L^T x = y (row j of F left of the diagonal = row j of L)
    for (int li = n_li - 1; li >= 0; --li) {
        T rows[KPL][KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int j = li * KPL + c;
            const T *row = Fm + (int64_t)(j < k ? j : 0) * k;
#pragma unroll
            for (int c2 = 0; c2 < KPL; ++c2) {               // unconditional clamped load, then the selection
                const T rv = row[e0 + c2 < k ? e0 + c2 : k - 1];
                rows[c][c2] = (e0 + c2 < k) ? rv : (T)0;
            }
        }
#pragma unroll
        for (int c = KPL - 1; c >= 0; --c) {
            const int j = li * KPL + c;
            if (j < k) {
                const T xj = bcast_lane(y[c], li) / bcast_lane(dg[c], li);
                if (lane == li) y[c] = xj;
#pragma unroll
                for (int c2 = 0; c2 < KPL; ++c2)
                    if (e0 + c2 < j) y[c2] = fma(-xj, rows[c][c2], y[c2]);
            }
        }
    }
    T *out = code + (idx ? idx[smp] : (int64_t)smp) * k;
#pragma unroll
    for (int c = 0; c < KPL; ++c)
        if (e0 + c < k) {
            r[e0 + c] = y[c];      // the reference leaves the solution in Dx (:188-197)
            out[e0 + c] = y[c];
        }
}

template <typename T>
int launch_chol_solve(hipStream_t stream, const T *F, int64_t f_stride, T *rhs, int b, int k, T *code,
                      const int64_t *idx) {
    if (b <= 0 || k <= 0) return MODL_OK;
    if (k > 512) return MODL_EINVAL;
    dim3 grid((unsigned)cdiv(b, 4)), block(256);
    if (k <= 64) hipLaunchKernelGGL((chol_solve_kernel<T, 1>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else if (k <= 128) hipLaunchKernelGGL((chol_solve_kernel<T, 2>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else if (k <= 256) hipLaunchKernelGGL((chol_solve_kernel<T, 4>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    else hipLaunchKernelGGL((chol_solve_kernel<T, 8>), grid, block, 0, stream, F, f_stride, rhs, b, k, code, idx);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}
template int launch_chol_solve<float>(hipStream_t, const float *, int64_t, float *, int, int, float *, const int64_t *);
template int launch_chol_solve<double>(hipStream_t, const double *, int64_t, double *, int, int, double *, const int64_t *);

}  // namespace modl
