/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the floating-point native kernels of MODL's hot path, in
 * f32 and f64 (the reference instantiates both through Cython's fused
 * `floating`):
 *   - elastic-net coordinate descent on a Gram system with duality-gap stop
 *       (reference: modl/decomposition/dict_fact_fast.pyx:270-427)
 *   - batch drivers for a shared / per-sample Gram, l1 branch only (the ridge
 *     branch is LAPACK posv, restated in oracle/somf_oracle.py with scipy's
 *     own posv)      (reference: dict_fact_fast.pyx:33-113, 125-215)
 *   - per-sample Gram running average (dict_fact_fast.pyx:217-228)
 *   - minibatch weight (dict_fact_fast.pyx:115-122)
 *   - elastic-net norm / projection / rescale of one atom
 *       (reference: modl/utils/math/enet.pyx:38-122, 125-148, 150-167)
 *   - CSR-pattern predict (reference: modl/decomposition/recsys_fast.pyx:10-38)
 *
 * The reference reaches BLAS (dot/axpy/asum/gemv) for the vector ops; BLAS
 * summation order is not part of its contract, so they are plain loops here.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  Pinned by tests/golden/*.npz generated from the real reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* dict_fact_fast.pyx:115-122 */
double osf_batch_weight(long count, long batch_size, double learning_rate, double offset)
{
    double w = 1;
    for (long i = count + 1 - batch_size; i < count + 1; i++)
        w *= (1 - pow((1 + offset) / (offset + i), learning_rate));
    return 1 - w;
}

/* recsys_fast.pyx:10-38 */
void osf_predict_csr(double *data, const int *indices, const int *indptr,
                     const double *P, long n_rows, long k, const double *Q, long n_cols)
{
    for (long u = 0; u < n_rows; u++)
        for (int ii = indptr[u]; ii < indptr[u + 1]; ii++) {
            int i = indices[ii];
            double dot = 0;
            for (long c = 0; c < k; c++) dot += P[u * k + c] * Q[c * n_cols + i];
            data[ii] = dot;
        }
}

#define T float
#define SFX(name) name##_f32
#include "somf_oracle_impl.inc"
#undef T
#undef SFX

#define T double
#define SFX(name) name##_f64
#include "somf_oracle_impl.inc"
#undef T
#undef SFX
