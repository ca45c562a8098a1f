// Block-coordinate dictionary update on the sampled feature rows.
//
// Replaces DictFact._update_dict (reference: modl/decomposition/dict_fact.py:650-715;
// also the masked variant modl/decomposition/recsys.py:187-213).  The reference
// sweeps the k atoms one after the other in a random order; each step is two
// rank-1 (ger) passes over a k x s gradient plus one projection of the atom onto
// its elastic-net ball, i.e. one global reduction over the s sampled features
// per atom — k strictly sequential, bandwidth-bound steps.
//
// Three device paths, all exact restatements of that sweep (only float
// summation order differs):
//
//  * blocked path  (comp_l1_ratio == 0, no positivity: DictFact / ImageDictFact
//    defaults, recsys).  The gradient row of atom j is evaluated lazily as
//    B_j - sum_i C[i,j] D_i with the CURRENT dictionary, so no k x s gradient is
//    ever stored and the ger passes disappear.  Atoms are processed in blocks of
//    NB = 32 (in sweep order).  Inside a block the l2 projection is a pure
//    rescaling, u_j -> alpha_j u_j, hence every candidate u_j is a combination
//    T[j,:] of the block's alpha-independent vectors
//        a_j = (B_j - sum_{i not in block or i after j} C[i,j] D_i) / C[j,j],
//    which one matrix-core product (s x k) . (k x NB) delivers for all features
//    at once.  Norms follow from the NB x NB Gram matrix of the a_j (one
//    double-precision reduction over the features per BLOCK instead of per
//    atom), the alpha_j recursion runs on that Gram matrix in one wavefront, and
//    D_j = alpha_j sum_m T[j,m] a_m is applied to all features in parallel.
//    k/NB global reductions instead of k; 2 k^2 s flops on MFMA instead of
//    4 k^2 s flops of ger.
//
//  * generic path (l1 / elastic-net atoms, positivity: fMRIDictFact, NMF): one
//    "gradient row" kernel (a wavefront per sampled feature) and one projection
//    kernel (Michelot iteration, enet_block.hpp) per atom.
//
//  * sgd path (dict_fact.py:695-708).
#include "enet_block.hpp"
#include "gemm.hpp"
#include "kernels.hpp"

namespace modl {

constexpr int kNB = 32;            // atoms per block of the blocked path
constexpr int kGramRows = 128;     // feature rows per Gram slab

struct DuLayout {
    size_t off_CP, off_cdiag, off_frozen, off_a, off_partial, off_Tp, off_u, off_pold, off_Dnew, off_colp, total;
    int64_t nslab_max, nwg_grad;
};

static DuLayout du_layout(size_t tsz, int64_t s_max, int k) {
    DuLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.nslab_max = cdiv(s_max > 0 ? s_max : 1, kGramRows);
    L.nwg_grad = 512;
    L.off_CP = take(tsz * (size_t)k * k);
    L.off_cdiag = take(tsz * (size_t)k);
    L.off_frozen = take(sizeof(int32_t) * (size_t)k);
    L.off_a = take(tsz * (size_t)s_max * kNB);
    L.off_partial = take(sizeof(double) * (size_t)L.nslab_max * (kNB * kNB + kNB));
    L.off_Tp = take(sizeof(double) * kNB * kNB);
    L.off_u = take(tsz * (size_t)s_max);
    L.off_pold = take(sizeof(double) * (size_t)L.nwg_grad);
    L.off_Dnew = take(tsz * (size_t)s_max * k);                       // sgd only, but sized once
    L.off_colp = take(sizeof(double) * (size_t)L.nslab_max * k);
    L.total = o;
    return L;
}

size_t dict_update_workspace(int dtype, int64_t s_max, int k) {
    return du_layout(dtype == MODL_F32 ? 4 : 8, s_max, k).total;
}

__device__ __forceinline__ int64_t sub_row(const int32_t *subset, int64_t f) { return subset ? (int64_t)subset[f] : f; }

// ---------------------------------------------------------------- blocked path
template <typename T>
__global__ __launch_bounds__(256) void bcd_prepare_kernel(const T *C, const int32_t *order, int k, T *CP, T *cdiag,
                                                          int32_t *frozen) {
    extern __shared__ int32_t inv[];                 // position of each atom in the sweep
    for (int j = threadIdx.x; j < k; j += 256) inv[order[j]] = j;
    __syncthreads();
    const int m = blockIdx.x;                        // source atom (row of C)
    const int pm = inv[m];
    for (int jj = threadIdx.x; jj < k; jj += 256) {
        T v = C[(int64_t)m * k + order[jj]];
        if (pm / kNB == jj / kNB && pm <= jj) v = 0;  // same block, not after jj: handled by the recursion
        CP[(int64_t)m * k + jj] = v;
        if (m == 0) {
            const T d = C[(int64_t)order[jj] * k + order[jj]];
            cdiag[jj] = d;
            frozen[jj] = !(d > (T)1e-20);             // dict_fact.py:681 "else do not update"
        }
    }
}

template <typename T> struct EpiBcdA {
    T *a; const T *Dt; const T *Bt; const T *cdiag; const int32_t *frozen; const int32_t *subset; const int32_t *order;
    int k, j0;
    __device__ __forceinline__ void operator()(int64_t f, int64_t jj, T v) const {
        const int64_t e = sub_row(subset, f) * k + order[j0 + jj];
        a[f * kNB + jj] = frozen[j0 + jj] ? Dt[e] : (Bt[e] - v) / cdiag[j0 + jj];
    }
};

template <typename T>
__global__ __launch_bounds__(256) void bcd_gram_kernel(const T *a, const T *Dt, const int32_t *subset,
                                                       const int32_t *order, int64_t s, int k, int j0, int nb,
                                                       double *partial) {
    __shared__ T As[64][kNB + 1];
    __shared__ double d2red[8][kNB];
    const int64_t f_begin = (int64_t)blockIdx.x * kGramRows;
    const int64_t f_end = (f_begin + kGramRows < s) ? f_begin + kGramRows : s;
    const int i = threadIdx.x / 8, jb = (threadIdx.x % 8) * 4;
    double acc[4] = {0, 0, 0, 0};
    const int col = threadIdx.x % kNB, rg = threadIdx.x / kNB;   // for the old-atom norms
    double d2 = 0;
    for (int64_t c0 = f_begin; c0 < f_end; c0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * kNB; e += 256) {
            const int r = e / kNB, c = e % kNB;
            As[r][c] = (c0 + r < f_end && c < nb) ? a[(c0 + r) * kNB + c] : (T)0;
        }
        __syncthreads();
        const int rows = (int)((f_end - c0 < 64) ? f_end - c0 : 64);
        for (int r = 0; r < rows; ++r) {
            const double ai = (double)As[r][i];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += ai * (double)As[r][jb + q];
        }
        if (col < nb)
            for (int r = rg; r < rows; r += 8) {
                const double x = (double)Dt[sub_row(subset, c0 + r) * k + order[j0 + col]];
                d2 += x * x;
            }
    }
    d2red[rg][col] = d2;
    __syncthreads();
    double *out = partial + (int64_t)blockIdx.x * (kNB * kNB + kNB);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[i * kNB + jb + q] = acc[q];
    if (threadIdx.x < kNB) {
        double t = 0;
        for (int g = 0; g < 8; ++g) t += d2red[g][threadIdx.x];
        out[kNB * kNB + threadIdx.x] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bcd_resolve_kernel(const double *partial, int nslab, const T *C,
                                                          const int32_t *order, const int32_t *frozen, int k, int j0,
                                                          int nb, T *comp_norm, double *Tp) {
    __shared__ double M[kNB][kNB + 1];
    __shared__ double D2[kNB];
    __shared__ double Tm[kNB][kNB + 1];
    __shared__ double coef[kNB][kNB + 1];
    __shared__ double alph[kNB];
    constexpr int kStride = kNB * kNB + kNB;
    for (int e = threadIdx.x; e < kStride; e += 256) {
        double sum = 0;
        for (int z = 0; z < nslab; ++z) sum += partial[(int64_t)z * kStride + e];   // fixed order: deterministic
        if (e < kNB * kNB) M[e / kNB][e % kNB] = sum;
        else D2[e - kNB * kNB] = sum;
    }
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
        const int i = e / kNB, j = e % kNB;
        double c = 0;
        if (i < j && j < nb && !frozen[j0 + j]) {
            const int oi = order[j0 + i], oj = order[j0 + j];
            c = (double)C[(int64_t)oi * k + oj] / (double)C[(int64_t)oj * k + oj];
        }
        coef[i][j] = c;
        Tm[i][j] = 0;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;                    // the recursion runs in wavefront 0 (no block barrier below)
    const int m = threadIdx.x;
    for (int j = 0; j < nb; ++j) {
        double t = 0;
        if (m < kNB) {
            t = (m == j) ? 1.0 : 0.0;
            for (int i = m; i < j; ++i) t -= coef[i][j] * alph[i] * Tm[i][m];
            Tm[j][m] = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double v = 0;
        if (m <= j)
            for (int m2 = 0; m2 <= j; ++m2) v += M[m2][m] * Tm[j][m2];
        const double nrm = wave_sum(t * v);
        const int jj = order[j0 + j];
        const double radius = (double)comp_norm[jj] + D2[j];
        double al;
        if (!(radius > 0.0)) al = 0.0;                // enet.pyx:57 (radius == 0 -> zero atom)
        else if (nrm <= radius) al = 1.0;             // enet.pyx:65
        else al = 1.0 / sqrt(nrm / radius);
        if (m == 0) {
            alph[j] = al;
            comp_norm[jj] = (T)(radius - al * al * nrm);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (m < kNB)
        for (int j = 0; j < kNB; ++j) Tp[j * kNB + m] = (j < nb) ? alph[j] * Tm[j][m] : 0.0;
}

template <typename T>
__global__ __launch_bounds__(256) void bcd_apply_kernel(const T *a, const double *Tp, T *Dt, const int32_t *subset,
                                                        const int32_t *order, int64_t s, int k, int j0, int nb) {
    __shared__ T As[64][kNB + 1];
    __shared__ double Ts[kNB][kNB + 1];
    const int64_t f0 = (int64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * kNB; e += 256) {
        const int r = e / kNB, c = e % kNB;
        As[r][c] = (f0 + r < s) ? a[(f0 + r) * kNB + c] : (T)0;
    }
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) Ts[e / kNB][e % kNB] = Tp[e];
    __syncthreads();
    const int r = threadIdx.x / 4, jg = threadIdx.x % 4;
    if (f0 + r >= s) return;
    T *row = Dt + sub_row(subset, f0 + r) * k;
    for (int j = jg; j < nb; j += 4) {
        double acc = 0;
        for (int m = 0; m <= j; ++m) acc += Ts[j][m] * (double)As[r][m];
        row[order[j0 + j]] = (T)acc;
    }
}

// ---------------------------------------------------------------- generic path
template <typename T, int KPL>
__global__ __launch_bounds__(256) void atom_grad_kernel(const T *Dt, const T *Bt, const T *C, const int32_t *subset,
                                                        int64_t s, int k, int j, int pos, double rho, T *u,
                                                        double *partial_old) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int e0 = lane * KPL;
    T cc[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) cc[c] = (e0 + c < k) ? C[(int64_t)j * k + e0 + c] : (T)0;   // row j == column j
    const T Cjj = C[(int64_t)j * k + j];
    const bool frozen = !(Cjj > (T)1e-20);
    double old = 0;
    for (int64_t f = (int64_t)blockIdx.x * 4 + wid; f < s; f += (int64_t)gridDim.x * 4) {
        const T *row = Dt + sub_row(subset, f) * k;
        double dot = 0;
#pragma unroll
        for (int c = 0; c < KPL; ++c)
            if (e0 + c < k) dot += (double)row[e0 + c] * (double)cc[c];
        dot = wave_sum(dot);
        if (lane == 0) {
            const T dj = row[j];
            T val = dj;
            if (!frozen) val = (T)((((double)Bt[sub_row(subset, f) * k + j] - dot) + (double)Cjj * (double)dj) / (double)Cjj);
            if (pos && val < (T)0) val = 0;            // dict_fact.py:684-685
            u[f] = val;
            const double a = fabs((double)dj);
            old += a * (rho + (1.0 - rho) * a);
        }
    }
    old = block_sum(old, red);
    if (threadIdx.x == 0) partial_old[blockIdx.x] = old;
}

template <typename T>
__global__ __launch_bounds__(1024) void atom_project_kernel(T *u, const double *partial_old, int nparts, T *Dt,
                                                            const int32_t *subset, int64_t s, int k, int j,
                                                            double rho, T *comp_norm) {
    __shared__ double red[16];
    double old = 0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) old += partial_old[i];
    old = block_sum(old, red);
    const double radius = (double)(T)((double)comp_norm[j] + old);   // comp_norm_[k] += subset_norm (:676-678)
    const double nrm = block_enet_project<T>(u, 1, u, 1, s, radius, rho, red);
    __syncthreads();
    for (int64_t f = threadIdx.x; f < s; f += blockDim.x) Dt[sub_row(subset, f) * k + j] = u[f];
    if (threadIdx.x == 0) comp_norm[j] = (T)(radius - nrm);          // :690-692
}

// -------------------------------------------------------------------- sgd path
template <typename T>
__global__ __launch_bounds__(256) void col_norm_partial_kernel(const T *Dt, const int32_t *subset, int64_t s, int k,
                                                               double rho, double *partial) {
    const int64_t f_begin = (int64_t)blockIdx.x * kGramRows;
    const int64_t f_end = (f_begin + kGramRows < s) ? f_begin + kGramRows : s;
    for (int j = threadIdx.x; j < k; j += 256) {
        double acc = 0;
        for (int64_t f = f_begin; f < f_end; ++f) {
            const double a = fabs((double)Dt[sub_row(subset, f) * k + j]);
            acc += a * (rho + (1.0 - rho) * a);
        }
        partial[(int64_t)blockIdx.x * k + j] = acc;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void col_norm_add_kernel(const double *partial, int nslab, int k, T *comp_norm) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k) return;
    double acc = 0;
    for (int z = 0; z < nslab; ++z) acc += partial[(int64_t)z * k + j];
    comp_norm[j] = (T)((double)comp_norm[j] + acc);
}
template <typename T> struct EpiSgd {
    T *Dnew; const T *Dt; const T *Bt; const int32_t *subset; int k; T ws;
    __device__ __forceinline__ void operator()(int64_t f, int64_t j, T v) const {
        const int64_t e = sub_row(subset, f) * k + j;
        Dnew[f * k + j] = Dt[e] + ws * (Bt[e] - v);
    }
};
template <typename T>
__global__ __launch_bounds__(256) void sgd_project_kernel(T *Dnew, T *Dt, const int32_t *subset, int64_t s, int k,
                                                          double rho, T *comp_norm) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    const double radius = (double)comp_norm[j];
    const double nrm = block_enet_project<T>(Dnew + j, k, Dnew + j, k, s, radius, rho, red);
    __syncthreads();
    for (int64_t f = threadIdx.x; f < s; f += 256) Dt[sub_row(subset, f) * k + j] = Dnew[f * k + j];
    if (threadIdx.x == 0) comp_norm[j] = (T)(radius - nrm);
}

// ---------------------------------------------------------------------- driver
template <typename T> int dict_update_generic(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches);

template <typename T>
int dict_update(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches) {
    const int k = a.k;
    const int64_t s = a.s;
    if (s <= 0 || k <= 0) return MODL_OK;
    if (k > 1024) return MODL_EINVAL;
    const DuLayout L = du_layout(sizeof(T), s, k);
    if (L.total > a.ws_bytes) return MODL_ENOMEM;
    char *ws = static_cast<char *>(a.ws);
    int nl = 0;

    if (a.optimizer == MODL_OPT_SGD) {
        double *colp = reinterpret_cast<double *>(ws + L.off_colp);
        T *Dnew = reinterpret_cast<T *>(ws + L.off_Dnew);
        const int nslab = (int)cdiv(s, kGramRows);
        hipLaunchKernelGGL((col_norm_partial_kernel<T>), dim3(nslab), dim3(256), 0, stream, a.Dt, a.subset, s, k,
                           a.comp_l1_ratio, colp);
        MODL_LAUNCH_CHECK();
        hipLaunchKernelGGL((col_norm_add_kernel<T>), dim3((unsigned)cdiv(k, 256)), dim3(256), 0, stream, colp, nslab, k,
                           a.comp_norm);
        MODL_LAUNCH_CHECK();
        nl += 2;
        Operand A, B;
        A.ptr = a.Dt; A.si = k; A.sk = 1; A.gi = gather32(a.subset);
        B.ptr = a.C; B.si = k; B.sk = 1;                 // B(n = j, kk = m) = C[j][m]
        EpiSgd<T> epi{Dnew, a.Dt, a.Bt, a.subset, k, (T)(a.w * a.step_size)};
        SplitWs none;
        MODL_TRY((launch_gemm<T, EpiSgd<T>>(stream, A, B, s, k, k, epi, none, &nl, 512, 1)));
        hipLaunchKernelGGL((sgd_project_kernel<T>), dim3(k), dim3(256), 0, stream, Dnew, a.Dt, a.subset, s, k,
                           a.comp_l1_ratio, a.comp_norm);
        MODL_LAUNCH_CHECK();
        ++nl;
    } else if (a.comp_l1_ratio == 0.0 && !a.comp_pos) {
        T *CP = reinterpret_cast<T *>(ws + L.off_CP);
        T *cdiag = reinterpret_cast<T *>(ws + L.off_cdiag);
        int32_t *frozen = reinterpret_cast<int32_t *>(ws + L.off_frozen);
        T *abuf = reinterpret_cast<T *>(ws + L.off_a);
        double *partial = reinterpret_cast<double *>(ws + L.off_partial);
        double *Tp = reinterpret_cast<double *>(ws + L.off_Tp);
        hipLaunchKernelGGL((bcd_prepare_kernel<T>), dim3(k), dim3(256), sizeof(int32_t) * (size_t)k, stream, a.C,
                           a.order, k, CP, cdiag, frozen);
        MODL_LAUNCH_CHECK();
        ++nl;
        const int nslab = (int)cdiv(s, kGramRows);
        for (int j0 = 0; j0 < k; j0 += kNB) {
            const int nb = (k - j0 < kNB) ? k - j0 : kNB;
            Operand A, B;
            A.ptr = a.Dt; A.si = k; A.sk = 1; A.gi = gather32(a.subset);
            B.ptr = CP + j0; B.si = 1; B.sk = k;         // B(n = jj, kk = m) = CP[m][j0 + jj]
            EpiBcdA<T> epi{abuf, a.Dt, a.Bt, cdiag, frozen, a.subset, a.order, k, j0};
            SplitWs none;
            MODL_TRY((launch_gemm<T, EpiBcdA<T>>(stream, A, B, s, nb, k, epi, none, &nl, 512, 1)));
            hipLaunchKernelGGL((bcd_gram_kernel<T>), dim3(nslab), dim3(256), 0, stream, abuf, a.Dt, a.subset, a.order,
                               s, k, j0, nb, partial);
            MODL_LAUNCH_CHECK();
            hipLaunchKernelGGL((bcd_resolve_kernel<T>), dim3(1), dim3(256), 0, stream, partial, nslab, a.C, a.order,
                               frozen, k, j0, nb, a.comp_norm, Tp);
            MODL_LAUNCH_CHECK();
            hipLaunchKernelGGL((bcd_apply_kernel<T>), dim3((unsigned)cdiv(s, 64)), dim3(256), 0, stream, abuf, Tp, a.Dt,
                               a.subset, a.order, s, k, j0, nb);
            MODL_LAUNCH_CHECK();
            nl += 3;
        }
    } else {
        return dict_update_generic<T>(stream, a, launches);
    }
    if (launches) *launches += nl;
    return MODL_OK;
}

// generic path: the sweep order is needed on the host (one launch pair per atom)
template <typename T>
int dict_update_generic(hipStream_t stream, const DictUpdateArgs<T> &a, int *launches) {
    const int64_t *h_order = a.h_order;
    if (!h_order) return MODL_EINVAL;
    const int k = a.k;
    const int64_t s = a.s;
    if (s <= 0 || k <= 0) return MODL_OK;
    if (k > 1024) return MODL_EINVAL;
    const DuLayout L = du_layout(sizeof(T), s, k);
    if (L.total > a.ws_bytes) return MODL_ENOMEM;
    char *ws = static_cast<char *>(a.ws);
    T *u = reinterpret_cast<T *>(ws + L.off_u);
    double *pold = reinterpret_cast<double *>(ws + L.off_pold);
    int nwg = (int)cdiv(s, 16);
    if (nwg > L.nwg_grad) nwg = (int)L.nwg_grad;
    if (nwg < 1) nwg = 1;
    for (int t = 0; t < k; ++t) {
        const int j = (int)h_order[t];
        if (j < 0 || j >= k) return MODL_EINVAL;
#define MODL_GRAD(KPL)                                                                                          \
    hipLaunchKernelGGL((atom_grad_kernel<T, KPL>), dim3(nwg), dim3(256), 0, stream, a.Dt, a.Bt, a.C, a.subset, s, \
                       k, j, a.comp_pos, a.comp_l1_ratio, u, pold)
        if (k <= 64) MODL_GRAD(1);
        else if (k <= 128) MODL_GRAD(2);
        else if (k <= 256) MODL_GRAD(4);
        else if (k <= 512) MODL_GRAD(8);
        else MODL_GRAD(16);
#undef MODL_GRAD
        MODL_LAUNCH_CHECK();
        hipLaunchKernelGGL((atom_project_kernel<T>), dim3(1), dim3(1024), 0, stream, u, pold, nwg, a.Dt, a.subset, s, k,
                           j, a.comp_l1_ratio, a.comp_norm);
        MODL_LAUNCH_CHECK();
    }
    if (launches) *launches += 2 * k;
    return MODL_OK;
}

template int dict_update<float>(hipStream_t, const DictUpdateArgs<float> &, int *);
template int dict_update<double>(hipStream_t, const DictUpdateArgs<double> &, int *);

}  // namespace modl
