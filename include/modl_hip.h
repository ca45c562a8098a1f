/*
 * modl_hip.h — C-ABI of libmodl_hip.so, the MI355X (gfx950) implementation of
 * MODL's per-minibatch SOMF hot path.
 *
 * This library takes the place of the reference's Cython extensions
 *   modl/decomposition/dict_fact_fast.pyx, modl/utils/math/enet.pyx,
 *   modl/utils/randomkit/{random_fast,sampler}.pyx (+ randomkit.c, distributions.c),
 *   modl/decomposition/recsys_fast.pyx
 * and of the numpy/BLAS calls of modl/decomposition/dict_fact.py:495-715.
 * Each entry point cites the reference interface it replaces (paths relative to
 * the reference repository).
 *
 * Conventions
 *  - every function returns int: 0 = ok, < 0 = MODL_E* (bad argument / state),
 *    > 0 = hipError_t of a failed HIP call.  No exceptions cross the ABI.
 *  - `_f32` / `_f64` suffixes mirror Cython's fused `floating` type.
 *  - pointers named d_* are DEVICE pointers (hipMalloc'd or torch CUDA tensors),
 *    h_* are HOST pointers.  `stream` is a hipStream_t passed as void*
 *    (NULL = default stream).  Calls are asynchronous w.r.t. the host unless
 *    stated otherwise; the caller owns all buffers.
 *  - matrices are row-major (C order), leading dimension = number of columns
 *    unless an explicit ld is given.
 *  - the dictionary and the surrogate statistic B are held FEATURE-MAJOR on the
 *    device: d_Dt[p][k] = components_.T, d_Bt[p][k] = B_.T, so that a sampled
 *    feature is one contiguous k-vector (coalesced gather of subsampled columns).
 */
#ifndef MODL_HIP_H
#define MODL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODL_ABI_VERSION 5

#define MODL_OK 0
#define MODL_EINVAL (-1)   /* bad argument */
#define MODL_ENOMEM (-2)   /* workspace too small / allocation failed */
#define MODL_ESTATE (-3)   /* object used in the wrong state */
#define MODL_ENOGPU (-4)   /* no HIP device available, or not a gfx950-class part (160 KiB of LDS per compute unit) */
#define MODL_ENORCCL (-5)  /* librccl.so could not be loaded (modl_comm_*) */
#define MODL_ERCCL (-6)    /* an RCCL call failed */
#define MODL_ETIMEOUT (-7) /* a persistent dictionary-update launch gave up half-way (the update is incomplete): modl_somf_status */

#define MODL_F32 0
#define MODL_F64 1

#define MODL_AGG_MASKED 0
#define MODL_AGG_FULL 1
#define MODL_AGG_AVERAGE 2

#define MODL_OPT_VARIATIONAL 0
#define MODL_OPT_SGD 1

/* modl_somf_desc.flags (diagnostics; 0 in production) */
#define MODL_FLAG_NO_RIDER 1      /* the B_ update of the rows that were not sampled runs as its own launch */
#define MODL_FLAG_GEMM_STAMPS 2   /* the head statistics product leaves shader-clock stamps (modl_somf_debug_gemm_stamps) */

int modl_abi_version(void);
/* number of visible HIP devices (0 without a GPU); never fails */
int modl_device_count(void);
const char *modl_error_string(int code);
/* process-wide diagnostics switches (the test-suite forces code paths with them; nothing reads the environment; every
 * switch of the product library selects between CORRECT code paths).
 * MODL_DEBUG_CD_SPARSE_PCT: value >= 0 = share of active coordinates (percent) below which a sweep of the one-wavefront
 * coordinate-descent kernel runs as an active-set sweep (0: always dense, 100: always sparse), -1 = the default rule.
 * MODL_DEBUG_CD_SPLIT: 1 (default) = shared-Gram solves with 32 <= k <= 1024 run the four-wavefront solver
 * (cd_split_impl.hpp), 0 = the one-wavefront solver (cd_solver.hip: the reference's operation order; the two agree to
 * rounding, the tests compare them).  Its variants for k > 256 spill registers and are only built into the
 * diagnostics library (libmodl_hip_diag.so, -DMODL_DIAG): the product library keeps the four-wavefront solver there.
 * MODL_DEBUG_CD_STAMPS / MODL_DEBUG_ATOM_STAMPS (shader-clock stamps written to a caller-supplied device buffer) exist
 * in the diagnostics library only: the product library answers MODL_EINVAL. */
#define MODL_DEBUG_CD_SPARSE_PCT 1
#define MODL_DEBUG_CD_SPLIT 2
#define MODL_DEBUG_CD_STAMPS 3        /* (diagnostics library) value = device pointer to 1024 uint64 (0: off): shader-clock stamps of sample 0 of the four-wavefront solver */
#define MODL_DEBUG_BCD_ACC 5          /* 1 (default): the blocked dictionary update sums its Gram contributions with integer atomics into one fixed-point record; 0: one record per workgroup, summed by every workgroup of the next launch */
#define MODL_DEBUG_ATOM_STAMPS 6      /* (diagnostics library) value = device pointer to 64 uint64, zeroed by the caller (0: off): cycle sums of the projecting workgroup of the grouped atom update (bcd.hip: atom_project_group_kernel) */
#define MODL_DEBUG_BCD_TINY 7         /* 1 (default): the f64 blocked dictionary update of at most 192 sampled features runs as ONE one-workgroup launch; 0: five launches per block of 32 atoms */
#define MODL_DEBUG_STAGE_AHEAD 8      /* 1 (default): inside modl_somf_partial_fit_chunk the next minibatch's parameters are copied to HBM by a workgroup of the dictionary update's last launch; 0: a staging launch at the head of every step */
#define MODL_DEBUG_BCD_PERSIST 9      /* 1 (default): the f32 blocked dictionary update runs as ONE persistent launch (a resolver workgroup + one workgroup per 32 / 64 sampled rows resident in LDS, look-ahead Gram matrices: csrc/bcd_persist.hip) whenever its workgroups fit the chip (hipOccupancyMaxActiveBlocksPerMultiprocessor for the kernel's own footprint); 0: one launch per block of 32 atoms.  Diagnostics library only (the product library treats them as 1): 3 = the resolver waits for a workgroup that never comes before the first block (the launch cannot run: the resolver completes the update alone, modl_somf_persist_recoveries counts it), 4 = the same from the second block on (the update is incomplete: MODL_ETIMEOUT) */
#define MODL_DEBUG_STATS_RESIDENT 10  /* 1 (default): the f32 statistics product X^T code over at least 4096 features and at most 256 atoms runs on persistent workgroups that keep the code matrix in registers (csrc/gemm_resident.hpp, tiles of 16 features); 0: the 32 x 32 tiles or the k-wide tiles of rounds 2-4 */
#define MODL_DEBUG_RECSYS_FUSED 11    /* 1 (default): a masked minibatch of RecsysDictFact with at most 64 (f64: 56) atoms and 64 rows runs its codes (rating chunks of 128, ticketed fixed-order sums, Cholesky) and C_ as ONE launch (csrc/recsys.hip: recsys_fused_kernel), B_ and the blocked dictionary update as launches of their own behind it; 2: B_ and the dictionary sweep inside that launch when one workgroup holds every touched item (512 in registers + what fits LDS), 3: the same without the LDS tier, 4: the sweep also on up to four workgroups that exchange an atom's sums through memory (all three measured slower, kept selectable and tested); 0: the separate launches of rounds 2-5 for everything */
#define MODL_DEBUG_ATOM_MWG 12        /* the l1 projection of an atom with more than 6144 and at most 16 384 sampled features (one launch per atom: the reference's HCP configuration).  1 (default): by the launch's last workgroup with the vector in registers, 40 or 64 elements per thread (csrc/bcd.hip: atom_project_regs; needs MODL_DEBUG_ATOM_PIPE != 0, else as 3); 3: spread over the launch's workgroups - an element or two per thread, a memory round trip per Michelot pass (mwg_l1_project: measured slower, kept selectable and tested); 0: the last workgroup projects from an LDS copy; diagnostics library: 2 = as 3 with a workgroup that withholds its sums (the attempt gives up and the last workgroup projects: the fallback path) */
#define MODL_DEBUG_BCD_FEW 13         /* 1 (default): the f64 blocked dictionary update of 193 to 2048 sampled features runs as ONE launch on up to sixteen workgroups of 128 features that exchange a block's Gram record through memory (csrc/bcd.hip: bcd_few_kernel); 0: four launches per block of 32 atoms */
#define MODL_DEBUG_ATOM_PIPE 14       /* 1 (default): in the per-atom sweep of MODL_DEBUG_ATOM_MWG the gradient rows of the NEXT group of four atoms are formed by workgroups riding on the launches of this group's atoms (against the dictionary as it is then; what this group changes is subtracted afterwards from its compact rows - csrc/bcd.hip: atom_corr_project_kernel, GradRide); 0: a gradient launch of its own between two groups */
int modl_debug_set(int what, int64_t value);
/* 1 in libmodl_hip_diag.so (built with -DMODL_DIAG: the same sources plus the A/B-only kernel variants and the stamp
 * switches), 0 in the product library */
int modl_is_diag_build(void);

/* ------------------------------------------------------------------------- *
 * Host-side RNG — replaces modl/utils/randomkit/random_fast.pyx (RandomState,
 * :49-150) over randomkit.c / distributions.c.  Streams are bit-identical to
 * the reference for the same seed.  Pure host code: usable without a GPU.
 * ------------------------------------------------------------------------- */
typedef struct modl_rk modl_rk;
int modl_rk_create(uint64_t seed, modl_rk **out);            /* RandomState(seed), rk_seed randomkit.c:138 */
void modl_rk_destroy(modl_rk *rk);
int modl_rk_seed(modl_rk *rk, uint64_t seed);                 /* random_fast.pyx:64 */
int modl_rk_random(modl_rk *rk, uint32_t *out);               /* rk_random, randomkit.c:212 */
int modl_rk_randint(modl_rk *rk, uint64_t high, int64_t *out);/* random_fast.pyx:76 -> rk_interval :260 */
int modl_rk_double(modl_rk *rk, double *out);                 /* rk_double, randomkit.c:292 */
int modl_rk_binomial(modl_rk *rk, int64_t n, double p, int64_t *out); /* random_fast.pyx:146 -> distributions.c:442 */
int modl_rk_permutation(modl_rk *rk, int64_t n, int64_t *h_out);      /* random_fast.pyx:79 */
/* raw MT19937 state (key[624], pos), the layout of numpy's legacy RandomState.get_state()[1:3]: a modl_rk loaded with
 * numpy's state continues numpy's stream (legacy permutation(n) == modl_rk_permutation: dict_fact.py:672) */
int modl_rk_get_mt_state(const modl_rk *rk, uint32_t *h_key624, int32_t *pos);
int modl_rk_set_mt_state(modl_rk *rk, const uint32_t *h_key624, int32_t pos);
int modl_rk_shuffle_i64(modl_rk *rk, int64_t *h_x, int64_t n);        /* random_fast.pyx:87 (1-D) */
/* random_fast.pyx:127: draws one swap sequence; h_trace[n] receives the
 * permutation, h_swaps[n] the swap targets to replay on other arrays. */
int modl_rk_shuffle_trace(modl_rk *rk, int64_t n, int64_t *h_trace, int64_t *h_swaps);
/* replays a swap sequence on the rows of a host array (random_fast.pyx:105-119) */
int modl_apply_swaps_rows(void *h_base, int64_t n, size_t row_bytes, const int64_t *h_swaps);
/* same on a DEVICE array (epoch shuffle of code_ / averages, dict_fact.py:359-379) */
int modl_apply_swaps_rows_device(void *d_base, int64_t n, size_t row_bytes, const int64_t *h_swaps,
                                 void *stream);

/* Feature sampler — replaces modl/utils/randomkit/sampler.pyx:9-70 */
typedef struct modl_sampler modl_sampler;
int modl_sampler_create(int64_t range, int rand_size, int replacement, uint64_t seed, modl_sampler **out);
void modl_sampler_destroy(modl_sampler *s);
/* yield_subset(reduction), sampler.pyx:41.  h_out must hold `range` entries;
 * *n_out receives the subset length.  Indices are unsorted, as in the reference. */
int modl_sampler_yield_subset(modl_sampler *s, double reduction, int64_t *h_out, int64_t *n_out);
/* public attributes of the reference class (sampler.pxd:4-14) */
int modl_sampler_get(modl_sampler *s, int64_t *range, int64_t *lim_inf, int64_t *lim_sup, int64_t *h_box);
/* full state save / restore (box, limits, MT key + position, binomial cache) */
size_t modl_sampler_state_bytes(const modl_sampler *s);
int modl_sampler_get_state(const modl_sampler *s, void *h_buf, size_t bytes);
int modl_sampler_set_state(modl_sampler *s, const void *h_buf, size_t bytes);

/* _batch_weight, dict_fact_fast.pyx:115-122 */
int modl_batch_weight(int64_t count, int64_t batch_size, double learning_rate, double offset, double *out);

/* ------------------------------------------------------------------------- *
 * Fine-grained device entry points, 1:1 with the Cython functions
 * (used by the parity tests and by transform()).
 * ------------------------------------------------------------------------- */

/* _enet_regression_single_gram, dict_fact_fast.pyx:125-215.
 * d_G[k][k], d_Dx[b][k] (overwritten by the solution when l1_ratio == 0, as in
 * the reference), d_X[b][ldx] (only the squared row norms are used, :334),
 * d_code[n][k] (rows d_indices[ii] are warm starts and results), d_indices[b]
 * int64 (NULL = 0..b-1).  d_sweeps (optional, int32[b]) receives the number of
 * coordinate-descent sweeps per sample.  d_ws / ws_bytes: scratch, see
 * modl_enet_regression_workspace(). */
size_t modl_enet_regression_workspace(int dtype, int64_t b, int64_t k, int multi_gram);
int modl_enet_regression_single_gram_f32(const float *d_G, float *d_Dx, const float *d_X, int64_t ldx,
                                         int64_t p, float *d_code, const int64_t *d_indices, int64_t b,
                                         int64_t k, float l1_ratio, float alpha, int positive, float tol,
                                         int max_iter, int32_t *d_sweeps, void *d_ws, size_t ws_bytes,
                                         void *stream);
int modl_enet_regression_single_gram_f64(const double *d_G, double *d_Dx, const double *d_X, int64_t ldx,
                                         int64_t p, double *d_code, const int64_t *d_indices, int64_t b,
                                         int64_t k, double l1_ratio, double alpha, int positive, double tol,
                                         int max_iter, int32_t *d_sweeps, void *d_ws, size_t ws_bytes,
                                         void *stream);
/* _enet_regression_multi_gram, dict_fact_fast.pyx:33-113: d_G[b][k][k] */
int modl_enet_regression_multi_gram_f32(const float *d_G, float *d_Dx, const float *d_X, int64_t ldx,
                                        int64_t p, float *d_code, const int64_t *d_indices, int64_t b,
                                        int64_t k, float l1_ratio, float alpha, int positive, float tol,
                                        int max_iter, int32_t *d_sweeps, void *d_ws, size_t ws_bytes,
                                        void *stream);
int modl_enet_regression_multi_gram_f64(const double *d_G, double *d_Dx, const double *d_X, int64_t ldx,
                                        int64_t p, double *d_code, const int64_t *d_indices, int64_t b,
                                        int64_t k, double l1_ratio, double alpha, int positive, double tol,
                                        int max_iter, int32_t *d_sweeps, void *d_ws, size_t ws_bytes,
                                        void *stream);
/* _update_G_average, dict_fact_fast.pyx:217-228: d_G_average[b][k][k] in place */
int modl_update_G_average_f32(float *d_G_average, const float *d_G, const float *d_w_sample, int64_t b,
                              int64_t k, void *stream);
int modl_update_G_average_f64(double *d_G_average, const double *d_G, const double *d_w_sample, int64_t b,
                              int64_t k, void *stream);

/* enet_norm / enet_projection / enet_scale, modl/utils/math/enet.pyx:125,38,150,
 * batched over `rows` vectors of length n laid out with element stride `inc`
 * and vector stride `ld` (so atoms can be rows of D or columns of Dt).
 * d_radius[rows] (projection, scale), d_out_norm[rows] (norm). */
int modl_enet_norm_f32(const float *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, float l1_ratio,
                       float *d_out_norm, void *stream);
int modl_enet_norm_f64(const double *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, double l1_ratio,
                       double *d_out_norm, void *stream);
int modl_enet_projection_f32(const float *d_v, float *d_out, int64_t rows, int64_t n, int64_t ld, int64_t inc,
                             const float *d_radius, float l1_ratio, void *stream);
int modl_enet_projection_f64(const double *d_v, double *d_out, int64_t rows, int64_t n, int64_t ld,
                             int64_t inc, const double *d_radius, double l1_ratio, void *stream);
int modl_enet_scale_f32(float *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, float l1_ratio,
                        float radius, void *stream);
int modl_enet_scale_f64(double *d_v, int64_t rows, int64_t n, int64_t ld, int64_t inc, double l1_ratio,
                        double radius, void *stream);

/* _predict, recsys_fast.pyx:10-38: d_data[nnz] = (P Q) at the CSR pattern;
 * P[n_rows][k], Q[k][n_cols] (f64, int32 CSR as in the reference). */
int modl_predict_csr(double *d_data, const int32_t *d_indices, const int32_t *d_indptr, const double *d_P,
                     int64_t n_rows, int64_t k, const double *d_Q, int64_t n_cols, void *stream);

/* ------------------------------------------------------------------------- *
 * Masked (missing-data) path — RecsysDictFact, modl/decomposition/recsys.py.
 * CSR pieces are int32 as in the reference (recsys_fast.pyx:11-13).
 * ------------------------------------------------------------------------- */
/* Ridge codes on the observed columns of b CSR rows (recsys.py:168-181 and _refit :254-265):
 * code = (D_S D_S^T + alpha |S| / p I)^-1 D_S x_S.  d_row_ids[b] selects rows of the CSR matrix
 * (NULL = rows 0..b-1), d_code_rows[b] the destination rows of d_code (NULL = the CSR row id).
 * Rows without entries keep their code. */
int modl_recsys_codes_f32(const float *d_Dt, int64_t p, int k, const int32_t *d_indptr, const int32_t *d_indices,
                          const float *d_data, const int64_t *d_row_ids, const int64_t *d_code_rows, int64_t b,
                          double alpha, float *d_code, void *stream);
int modl_recsys_codes_f64(const double *d_Dt, int64_t p, int k, const int32_t *d_indptr, const int32_t *d_indices,
                          const double *d_data, const int64_t *d_row_ids, const int64_t *d_code_rows, int64_t b,
                          double alpha, double *d_code, void *stream);
/* Per-feature weighted B_ update of one minibatch (recsys.py:175,182-185).  d_subset[u]: the features
 * touched by the batch; d_fptr[u+1] / d_entry_sample / d_entry_val: for each of them its entries IN
 * BATCH ORDER (position of the row in the batch, rating); d_code_b[b][k]: the batch's codes;
 * w_times_n_iter = w * n_iter_.  Updates d_feature_n_iter (int64[p]) too. */
int modl_recsys_update_B_f32(float *d_Bt, int k, int64_t *d_feature_n_iter, const int32_t *d_subset,
                             const int32_t *d_fptr, const int32_t *d_entry_sample, const float *d_entry_val,
                             const float *d_code_b, double w_times_n_iter, int64_t u, void *stream);
int modl_recsys_update_B_f64(double *d_Bt, int k, int64_t *d_feature_n_iter, const int32_t *d_subset,
                             const int32_t *d_fptr, const int32_t *d_entry_sample, const double *d_entry_val,
                             const double *d_code_b, double w_times_n_iter, int64_t u, void *stream);
/* _predict (recsys_fast.pyx:10-38) against the feature-major dictionary: d_out[nnz] (double) */
int modl_recsys_predict_f32(double *d_out, const int32_t *d_indices, const int32_t *d_indptr, const float *d_code,
                            int64_t n_rows, int k, const float *d_Dt, void *stream);
int modl_recsys_predict_f64(double *d_out, const int32_t *d_indices, const int32_t *d_indptr, const double *d_code,
                            int64_t n_rows, int k, const double *d_Dt, void *stream);
/* One minibatch of RecsysDictFact._single_batch_fit (recsys.py:147-165, 168-213) in ONE call: the batch's ratings are
 * grouped by item on the host (batch order kept inside an item), staged through a pinned slot of the plan, and the
 * codes, the per-item B_ update, the C_ update and the dictionary update on the union of the batch's items are
 * launched on `stream`.  Asynchronous; the plan owns its staging slots and scratch.
 *   h_* : the CSR matrix in host memory (int32 / T, as d_*), n_rows its number of rows;
 *   h_rows[b] : the batch (row ids, in batch order);  h_order[k] : the atom order (numpy permutation, recsys.py:196);
 *   w : _batch_weight;  n_iter : n_iter_ after this batch (w_B = min(1, w n_iter / feature_n_iter), :182).
 * max_entries: the largest number of ratings one batch may hold. */
typedef struct modl_recsys_plan modl_recsys_plan;
int modl_recsys_plan_create(int dtype, int64_t p, int k, int64_t max_batch, int64_t max_entries, modl_recsys_plan **out);
void modl_recsys_plan_destroy(modl_recsys_plan *plan);
int modl_recsys_minibatch_f32(modl_recsys_plan *plan, const int32_t *h_indptr, const int32_t *h_indices, const float *h_data,
                              int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const float *d_data,
                              const int64_t *h_rows, int64_t b, const int64_t *h_order, double alpha, double w,
                              double n_iter, float *d_Dt, float *d_Bt, float *d_C, float *d_code, float *d_comp_norm,
                              int64_t *d_feature_n_iter, void *stream);
int modl_recsys_minibatch_f64(modl_recsys_plan *plan, const int32_t *h_indptr, const int32_t *h_indices, const double *h_data,
                              int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const double *d_data,
                              const int64_t *h_rows, int64_t b, const int64_t *h_order, double alpha, double w,
                              double n_iter, double *d_Dt, double *d_Bt, double *d_C, double *d_code,
                              double *d_comp_norm, int64_t *d_feature_n_iter, void *stream);
/* The per-minibatch host loop of RecsysDictFact.fit (recsys.py:135-139, 147-165) for a run of minibatches in ONE call:
 * h_rows[n_rows_fit] = the (permuted) row ids of the run, cut into minibatches of batch_size (the last one ragged).  Per
 * minibatch: n_iter += its rows, w = _batch_weight(n_iter, rows, learning_rate, 0) (:154), order = the legacy permutation(k)
 * of `order_rng` (:196; load it with numpy's state, modl_rk_set_mt_state), then the minibatch as modl_recsys_minibatch_*.
 * *n_iter is advanced; *n_done (optional) = minibatches enqueued - on an error the ones before the failing minibatch, with
 * n_iter and the generator left behind the last enqueued one.  Asynchronous. */
int modl_recsys_fit_batches_f32(modl_recsys_plan *plan, const int32_t *h_indptr, const int32_t *h_indices, const float *h_data,
                                int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const float *d_data,
                                const int64_t *h_rows, int64_t n_rows_fit, int64_t batch_size, modl_rk *order_rng, double alpha,
                                double learning_rate, int64_t *n_iter, float *d_Dt, float *d_Bt, float *d_C, float *d_code,
                                float *d_comp_norm, int64_t *d_feature_n_iter, void *stream, int64_t *n_done);
int modl_recsys_fit_batches_f64(modl_recsys_plan *plan, const int32_t *h_indptr, const int32_t *h_indices, const double *h_data,
                                int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices, const double *d_data,
                                const int64_t *h_rows, int64_t n_rows_fit, int64_t batch_size, modl_rk *order_rng, double alpha,
                                double learning_rate, int64_t *n_iter, double *d_Dt, double *d_Bt, double *d_C, double *d_code,
                                double *d_comp_norm, int64_t *d_feature_n_iter, void *stream, int64_t *n_done);
/* Synchronises `stream`; MODL_ETIMEOUT if a dictionary-update launch of this plan gave up in a cross-workgroup wait since the last
 * call (csrc/bcd.hip: bcd_few_kernel on up to sixteen workgroups; the update is incomplete, the flag is cleared), MODL_OK otherwise.
 * While the flag is up every further minibatch of the plan is refused with MODL_ETIMEOUT. */
int modl_recsys_plan_status(modl_recsys_plan *plan, void *stream);
/* diagnostics: minibatches of this plan that ran as ONE launch (csrc/recsys.hip: recsys_fused_kernel: at most 64 atoms, 64 rows
 * and 8192 ratings per minibatch) / as the separate launches */
int modl_recsys_plan_counts(const modl_recsys_plan *plan, int64_t *fused, int64_t *split);
/* diagnostics: on = 1 makes the last workgroup of every one-launch minibatch leave 100 MHz wall-clock stamps ([0] entry, [1] ids,
 * [2] Gram chunk, [3] records summed, [4] factor, [5] ticket, [6] last phase, [7] C_, [8] B_, [9] sweep, [10] end, [11] items,
 * [12] sweep workgroups); h_out[16] (may be NULL) receives the last launch's (synchronises the device); on = 0 frees them */
int modl_recsys_plan_stamps(modl_recsys_plan *plan, int on, unsigned long long *h_out);
/* diagnostics: host time (ms) the plan has spent waiting for the device to read a staging slot (eight minibatches of lead) */
int modl_recsys_plan_wait_ms(const modl_recsys_plan *plan, double *ms);
/* C = beta C + alpha rows^T rows, rows[b][k]  (recsys.py:159-160: beta = 1 - w, alpha = w / b) */
int modl_gram_axpby_f32(const float *d_rows, int64_t b, int k, float *d_C, float beta, float alpha, void *stream);
int modl_gram_axpby_f64(const double *d_rows, int64_t b, int k, double *d_C, double beta, double alpha, void *stream);
/* _update_dict on device-resident state (dict_fact.py:650-715, recsys.py:187-213): block-coordinate
 * update of the rows d_subset[s] (int32, NULL = all rows 0..s-1) of d_Dt[p][k].  d_order / h_order: the
 * atom order on the device (int32) and on the host (int64).  Scratch: modl_dict_update_workspace(). */
size_t modl_dict_update_workspace(int dtype, int64_t s_max, int k);
int modl_dict_update_f32(float *d_Dt, const float *d_Bt, const float *d_C, float *d_comp_norm,
                         const int32_t *d_subset, int64_t s, const int32_t *d_order, const int64_t *h_order, int k,
                         int optimizer, int comp_pos, double comp_l1_ratio, double w, double step_size, void *d_ws,
                         size_t ws_bytes, void *stream);
int modl_dict_update_f64(double *d_Dt, const double *d_Bt, const double *d_C, double *d_comp_norm,
                         const int32_t *d_subset, int64_t s, const int32_t *d_order, const int64_t *h_order, int k,
                         int optimizer, int comp_pos, double comp_l1_ratio, double w, double step_size, void *d_ws,
                         size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Fused, device-resident minibatch step = DictFact._single_batch_fit,
 * dict_fact.py:495-533 (+ _compute_code :577-648, _update_C/_update_B :559-575,
 * _update_dict :650-715).  The host keeps drawing `subset` (modl_sampler),
 * `order` (numpy legacy RandomState.permutation, dict_fact.py:672) and `w`
 * (modl_batch_weight) exactly as the reference does and passes them in.
 * ------------------------------------------------------------------------- */
typedef struct modl_somf_desc {
    int32_t dtype;          /* MODL_F32 / MODL_F64 */
    int32_t k;              /* n_components */
    int64_t p;              /* n_features */
    int64_t n_samples;      /* rows of code_ (and of the averages) */
    int32_t G_agg;          /* MODL_AGG_* (dict_fact.py:132-133) */
    int32_t Dx_agg;
    int32_t optimizer;      /* MODL_OPT_* */
    int32_t code_pos;
    int32_t comp_pos;
    int32_t max_iter;
    double code_alpha;
    double code_l1_ratio;
    double comp_l1_ratio;
    double tol;
    double step_size;
    int32_t max_batch;      /* largest minibatch this plan will see */
    int32_t flags;          /* MODL_FLAG_* (diagnostics), 0 otherwise */
} modl_somf_desc;

/* device state of one estimator (allocated by the caller; all T = dtype) */
typedef struct modl_somf_state {
    void *d_Dt;          /* [p][k]  components_.T */
    void *d_Bt;          /* [p][k]  B_.T  (several GPUs: this rank's partial sum, B_ = sum over ranks) */
    void *d_C;           /* [k][k]  C_    (several GPUs: this rank's partial sum) */
    void *d_code;        /* [n_samples][k] code_ */
    void *d_comp_norm;   /* [k] comp_norm_ */
    void *d_G;           /* [k][k] G_ (G_agg == full), else NULL */
    void *d_Dx_average;  /* [n_samples][k] (Dx_agg == average), else NULL */
    void *d_G_average;   /* [n_samples][k][k] (G_agg == average), else NULL */
} modl_somf_state;

/* one minibatch; h_* arrays are copied to the device by the call */
typedef struct modl_somf_batch {
    const void *d_X;            /* [b][ldx] minibatch rows, device */
    int64_t ldx;
    int32_t b;                  /* rows in this rank's minibatch */
    int32_t s;                  /* subset length; s == p with h_subset == NULL means all features */
    const int64_t *h_sample_idx;/* [b] rows of code_ (NULL = 0..b-1) */
    const int64_t *h_subset;    /* [s] feature subset (unsorted ok) */
    const int64_t *h_order;     /* [k] atom order of the dictionary update */
    const void *h_w_sample;     /* [b] T, w_sample (only for *_agg == average; may be NULL otherwise) */
    double w;                   /* minibatch weight (_batch_weight) */
    double reduction;           /* self.reduction (scales Dx and G, dict_fact.py:595,604) */
    int64_t b_global;           /* sum of b over all ranks (== b on one GPU) */
} modl_somf_batch;

typedef struct modl_somf_plan modl_somf_plan;
/* allocates the per-estimator scratch (device workspace, pinned staging) on the
 * current device. */
int modl_somf_plan_create(const modl_somf_desc *desc, modl_somf_plan **out);
void modl_somf_plan_destroy(modl_somf_plan *plan);
/* update solver / aggregation parameters between minibatches (set_params,
 * dict_fact.py:339-357).  k, p, dtype, n_samples, max_batch must not change. */
int modl_somf_plan_update(modl_somf_plan *plan, const modl_somf_desc *desc);

/* One GPU: the whole minibatch step (codes, C_/B_ update in the epilogues of the increment products, dictionary
 * update).  Asynchronous on `stream`. */
int modl_somf_step(modl_somf_plan *plan, const modl_somf_state *st, const modl_somf_batch *bt, void *stream);

/* The per-minibatch HOST loop of partial_fit / _single_batch_fit (dict_fact.py:331-337, 495-526) behind the boundary:
 * n_rows rows of d_X in minibatches of batch_size (the last one may be ragged), for each of them
 *   subset  = sampler.yield_subset(reduction)                      (:507)
 *   n_iter += b_global ; w = _batch_weight(n_iter, b_global, learning_rate, 0)   (:510, :515)
 *   order   = the legacy permutation(k) of `order_rng`               (:672; load it with numpy's state, see
 *                                                                     modl_rk_set_mt_state, and read the state back)
 *   modl_somf_step (comm == NULL) or modl_somf_step_dist (comm != NULL)
 * - the same draws in the same order as the Python loop, so the same bits.  h_sample_idx: n_rows rows of code_ or
 * NULL (0 .. n_rows-1); h_b_global: rows of each GLOBAL minibatch (several ranks) or NULL (== this rank's).
 * Plans with *_agg == average need the per-sample weights the caller keeps (sample_n_iter_): MODL_EINVAL - use the
 * per-minibatch calls.  Asynchronous on `stream` (blocks only on the 8-deep staging ring).
 * With a communicator the staging copy of minibatch t + 1 rides on the last launch of step t's dictionary update
 * exactly as on one GPU (ABI 3; before, a multi-GPU step spent a launch on it).
 * Errors (ABI 3): *n_done (may be NULL) receives the number of minibatches that were enqueued completely; when a
 * minibatch fails, *n_iter, `sampler` and `order_rng` are left where the LAST ENQUEUED minibatch left them (the draws
 * of the look-ahead are rewound), so that the caller can correct the input and continue the reference's streams. */
struct modl_comm;
int modl_somf_partial_fit_chunk(modl_somf_plan *plan, const modl_somf_state *st, const void *d_X, int64_t ldx,
                                int64_t n_rows, int32_t batch_size, const int64_t *h_sample_idx, modl_sampler *sampler,
                                modl_rk *order_rng, int64_t *n_iter, double learning_rate, double reduction,
                                const int64_t *h_b_global, struct modl_comm *comm, int64_t *n_done, void *stream);

/* Several GPUs (one process per GPU, the rows of a global minibatch of b_global rows split over the ranks; every
 * rank passes the same subset / order / w).  The recursions C_ <- (1 - w) C_ + (w / b_global) code^T code and
 * B_ <- (1 - w) B_ + (w / b_global) code^T X are linear in the increments, so every rank keeps its own PARTIAL
 * statistics (st->d_C, st->d_Bt: C_ = sum over ranks, B_ likewise) and only what the dictionary update reads is
 * exchanged - the HEAD  [ C_r (k*k) | the rows of B_r of the sampled features, compact (s*k) ]:
 *   1. modl_somf_code_and_partials : codes of the rank's rows, update of its partial statistics, head -> d_head;
 *   2. the caller sums d_head[0 .. modl_somf_head_elems) over the ranks (ncclAllReduce / RCCL, in place);
 *   3. modl_somf_apply_and_update_dict : the dictionary update from the summed head, identical on every rank
 *      (replicas of the dictionary stay bit-identical).
 * d_head holds modl_somf_delta_elems() = k*k + p*k elements of T (a minibatch without a subset array carries all
 * p rows of B_r).  With one rank and no reduction the two calls give the same bits as modl_somf_step. */
int64_t modl_somf_delta_elems(const modl_somf_desc *desc);
int modl_somf_code_and_partials(modl_somf_plan *plan, const modl_somf_state *st, const modl_somf_batch *bt,
                                void *d_head, void *stream);
/* elements of the head written by the LAST modl_somf_code_and_partials (MODL_ESTATE if none is pending) */
int modl_somf_head_elems(const modl_somf_plan *plan, int64_t *head_elems);
int modl_somf_apply_and_update_dict(modl_somf_plan *plan, const modl_somf_state *st,
                                    const modl_somf_batch *bt, const void *d_head, void *stream);

/* The same exchange inside the library: an RCCL communicator (one per process / GPU, as the north star's
 * "RCCL all-reduce of A_ and B_ over xGMI") and the whole multi-GPU minibatch as ONE call, everything enqueued on
 * one stream.  librccl.so is resolved at run time (dlopen), MODL_ENORCCL if it is not there.
 *   rank 0: modl_comm_unique_id(&id); ship the 128 bytes to the other ranks (MPI, a file, torch.distributed ...);
 *   every rank, with its GPU current: modl_comm_create(&id, rank, world, &comm)   (collective);
 *   per minibatch: modl_somf_step_dist(plan, &st, &bt, comm, stream);  bt.b_global = rows of the global minibatch. */
typedef struct { char bytes[128]; } modl_comm_id;          /* ncclUniqueId */
typedef struct modl_comm modl_comm;
int modl_comm_unique_id(modl_comm_id *out);
int modl_comm_create(const modl_comm_id *id, int rank, int world, modl_comm **out);
void modl_comm_destroy(modl_comm *comm);
/* in-place sum over the ranks of n elements (dtype MODL_F32 / MODL_F64) of a device buffer, on `stream` */
int modl_comm_all_reduce_sum(modl_comm *comm, void *d_buf, int64_t n, int dtype, void *stream);
/* ABI 4 - the abort path.  A rank that dies leaves the others inside ncclAllReduce in the middle of a chunk call; a plain
 * stream synchronisation would then wait for ever.  modl_comm_wait waits for `stream` to drain WHILE watching the
 * communicator (ncclCommGetAsyncError) and the clock: an asynchronous RCCL error, or more than timeout_s seconds
 * (<= 0: no limit) without the stream draining, aborts the communicator (ncclCommAbort: the kernels of this rank that wait
 * inside a collective are released) and returns MODL_ERCCL; MODL_OK when the stream is idle.  After an abort every call
 * on the communicator returns MODL_ERCCL; modl_comm_destroy still frees the handle.  modl_comm_abort: the same, at once
 * (a launcher that has seen another rank exit).  The reference has no counterpart (it is a single-process estimator). */
int modl_comm_wait(modl_comm *comm, void *stream, double timeout_s);
int modl_comm_abort(modl_comm *comm);
int modl_somf_step_dist(modl_somf_plan *plan, const modl_somf_state *st, const modl_somf_batch *bt, modl_comm *comm,
                        void *stream);

/* d_dst[r][0..cols) = d_src[d_idx[r]][0..cols): the row permutations around the path (X = X[permutation] after
 * an epoch, dict_fact.py:309-310; masked_data[permutation], fmri.py:541) on device-resident rows. */
int modl_gather_rows_f32(const float *d_src, int64_t src_ld, const int64_t *d_idx, int64_t n_rows, int64_t cols,
                         float *d_dst, int64_t dst_ld, void *stream);
int modl_gather_rows_f64(const double *d_src, int64_t src_ld, const int64_t *d_idx, int64_t n_rows, int64_t cols,
                         double *d_dst, int64_t dst_ld, void *stream);

/* G_ = D D^T (prepare / set_params(G_agg='full'), dict_fact.py:355,477) */
int modl_somf_full_gram(modl_somf_plan *plan, const void *d_Dt, void *d_G, void *stream);

/* CodingMixin.transform, dict_fact.py:47-92: d_code_out[n][k] from ones.
 * d_G may be NULL (computed from d_Dt). */
int modl_somf_transform(modl_somf_plan *plan, const void *d_Dt, const void *d_G, const void *d_X, int64_t ldx,
                        int64_t n, void *d_code_out, void *stream);

/* diagnostics: coordinate-descent sweeps per sample of the last phase-1 call of this plan
 * (0 for the ridge branch).  Synchronises `stream`. */
int modl_somf_last_sweeps(modl_somf_plan *plan, int32_t *h_out, int cap, int *n_out, void *stream);
/* diagnostics: from now on minibatch number t (counted from this call) of this plan leaves its sweep counts in
 * d_buf[(t % cap_minibatches) * max_batch ...] (int32, device memory owned by the caller, cap_minibatches * max_batch
 * entries) - a tolerance-stopped solver may legitimately do a sweep more or less on a sample when its inputs differ in
 * the last bits, and a long-horizon parity test has to know where that happened.  d_buf == NULL: off. */
int modl_somf_sweeps_history(modl_somf_plan *plan, int32_t *d_buf, int64_t cap_minibatches);

/* diagnostics: 48 shader-clock stamps of the last fused dictionary-update block launch (40 ..: the first riding tile),
 * h_out[48] (synchronises the device) */
/* The persistent dictionary-update launch (csrc/bcd_persist.hip) needs all its workgroups resident at once.  The library asks
 * the runtime whether they fit (occupancy of the kernel's own register / LDS footprint on the plan's device); what it cannot
 * ask is whether another process, or a compute-unit mask, holds units of the same GPU.  Every wait inside the launch is
 * bounded, and what a wait that gives up means is decided on the device:
 *  - BEFORE the first block of atoms is resolved nothing has been applied: the row workgroups leave, the resolver workgroup
 *    runs the reference's sweep by itself (milliseconds instead of microseconds) and the stream carries on with a correctly
 *    updated dictionary - no error.  The plan counts the event (modl_somf_persist_recoveries, a plain read of pinned memory,
 *    no synchronisation) and, from the first one on, keeps one launch per block: whatever held the units may still be there.
 *  - LATER the update is incomplete.  The plan raises a flag that the host reads before every enqueue: the next
 *    modl_somf_step / _code_and_partials / _apply_and_update_dict / _partial_fit_chunk of that plan returns MODL_ETIMEOUT
 *    without enqueuing anything, and keeps doing so until modl_somf_status has been called.
 * modl_somf_status synchronises `stream`, returns MODL_ETIMEOUT if an update of this plan was left incomplete since the last
 * call (and clears the flag; the dictionary of that plan is then not to be trusted: DictFact.prepare again), MODL_OK otherwise.
 * The Python estimator checks it whenever it synchronises and warns once per recovery.  (The reference has no counterpart: it
 * is CPU code.) */
int modl_somf_status(modl_somf_plan *plan, void *stream);
int modl_somf_persist_recoveries(modl_somf_plan *plan, int64_t *count);
int modl_somf_debug_stamps(modl_somf_plan *plan, unsigned long long *h_out);
/* diagnostics: 192 shader-clock stamps of the last PERSISTENT dictionary-update launch (csrc/bcd_persist.hip; written by the
 * diagnostics build only): [0] resolver start, [1 + 5 b ..] per block b: arrivals complete, pieces in LDS, Gram matrix
 * formed, recursion done, S published; [96] row workgroup 0 start, [97] block 0 handed over, [98 + 4 b ..] per block b >= 1:
 * S of block b - 2 fetched, applied + corrected, candidates formed, pieces handed over */
int modl_somf_debug_persist_stamps(modl_somf_plan *plan, unsigned long long *h_out);

/* diagnostics (env MODL_GEMM_STAMPS=1): h_out[8] = shader-clock stamps of one tile of the head statistics product:
 * [0] entry, [1] loads issued, [2] first K-tile in LDS, [3] K loop done, [4] epilogue done; [6], [7] = 100 MHz wall
 * clock at entry / exit (synchronises the device) */
int modl_somf_debug_gemm_stamps(modl_somf_plan *plan, unsigned long long *h_out);

/* ------------------------------------------------------------------------- *
 * Either side of the step: the image patch pipeline and the
 * objective value of CodingMixin.score.
 * ------------------------------------------------------------------------- */
/* fill, modl/input_data/image_fast.pyx:59-74: every (pp, qq, rr) of a p x q x r grid in C order, h_out[p*q*r][3]. Host. */
int modl_image_fill(int64_t p, int64_t q, int64_t r, int64_t *h_out);
/* clean_mask, image_fast.pyx:12-57: origins of the x*y*z windows of h_image[H][W][C] that hold no missing (-1)
 * value, in C order; h_out[(H-x+1)(W-y+1)(C-z+1)][3] (may be NULL: count only), *n_out = number of rows.  Host. */
int modl_image_clean_mask_f32(const float *h_image, int64_t H, int64_t W, int64_t C, int64_t x, int64_t y, int64_t z,
                              int64_t *h_out, int64_t *n_out);
int modl_image_clean_mask_f64(const double *h_image, int64_t H, int64_t W, int64_t C, int64_t x, int64_t y, int64_t z,
                              int64_t *h_out, int64_t *n_out);
/* LazyCleanPatchExtractor.partial_transform (modl/feature_extraction/image.py:54-63) + scale_patches
 * (modl/input_data/image.py:4-23, channel-wise) + flattening (modl/decomposition/image.py:190-199) in one launch:
 * d_out[r][:] = window of d_image[H][W][C] at origin d_idx3[r] = (i, j, c0), shape (x, y, z), C order, each channel
 * centred (with_mean) and divided by (its l2 norm or 1 if 0) * sqrt(z) (with_std).  z <= 1024. */
int modl_image_patches_f32(const float *d_image, int64_t H, int64_t W, int64_t C, const int64_t *d_idx3, int64_t n, int x,
                           int y, int z, int with_mean, int with_std, float *d_out, int64_t ldo, void *stream);
int modl_image_patches_f64(const double *d_image, int64_t H, int64_t W, int64_t C, const int64_t *d_idx3, int64_t n, int x,
                           int y, int z, int with_mean, int with_std, double *d_out, int64_t ldo, void *stream);
/* The three sums of CodingMixin.score (dict_fact.py:108-114) on device-resident operands:
 * d_out3 = [ sum (X - code D)^2, sum |code|, sum code^2 ] (f64 accumulation, fixed order: run-to-run reproducible).
 * d_X[n][ldx], d_Dt[p][k], d_code[n][k]; scratch of modl_objective_workspace() bytes (any smaller size that holds at
 * least one row of X works too, in more passes). */
size_t modl_objective_workspace(int dtype, int64_t n, int64_t p);
int modl_objective_f32(const float *d_X, int64_t ldx, int64_t n, int64_t p, const float *d_Dt, int k, const float *d_code,
                       void *d_ws, size_t ws_bytes, double *d_out3, void *stream);
int modl_objective_f64(const double *d_X, int64_t ldx, int64_t n, int64_t p, const double *d_Dt, int k, const double *d_code,
                       void *d_ws, size_t ws_bytes, double *d_out3, void *stream);

/* layout helpers: out[c][r] = in[r][c]  (components_ <-> Dt) */
int modl_transpose_f32(const float *d_in, float *d_out, int64_t rows, int64_t cols, void *stream);
int modl_transpose_f64(const double *d_in, double *d_out, int64_t rows, int64_t cols, void *stream);

/* per-kernel-class timing of the fused step (HIP events on `stream`).
 * enable: 0 = off, 1 = every section, otherwise a mask: bit (i + 1) times section i of the order
 * {code_gemm, code_solve, stats_gemm, stats_apply, dict_update} — each timed section costs two event
 * records (a few microseconds of stream bubble each).  get: copies up to `cap` entries; names are static. */
#define MODL_PROF_MAX 16
typedef struct modl_prof_entry {
    const char *name;
    double ms_total;     /* accumulated device time */
    int64_t launches;    /* kernel launches accumulated */
    int64_t calls;       /* timed regions accumulated */
} modl_prof_entry;
int modl_somf_prof_enable(modl_somf_plan *plan, int enable);
/* host time (ms) the calls of this plan have spent waiting for a free staging slot, i.e. for the device: the
 * host runs at most 8 minibatches ahead.  Host enqueue time minus this is what the host itself needs. */
int modl_somf_host_wait_ms(modl_somf_plan *plan, double *out, int reset);
/* record the section events on every `every`-th minibatch only (each recorded section costs two event records, a
 * stream bubble of several microseconds: sampling keeps the timed region undisturbed) */
int modl_somf_prof_stride(modl_somf_plan *plan, int every);
int modl_somf_prof_get(modl_somf_plan *plan, modl_prof_entry *out, int cap, int *n_out);
int modl_somf_prof_reset(modl_somf_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* MODL_HIP_H */
