// Device helpers shared by the translation units of the blocked dictionary update (bcd.hip: one launch per block of 32
// atoms; bcd_persist.hip: ONE persistent launch per dictionary update, round 5): constants, the packed Gram record and its
// sinks, the fixed-point Gram accumulator, the alpha recursion on one / two wavefronts, the riding tiles.
// (Moved here verbatim from bcd.hip; what each piece replaces in the reference is cited there.)
#pragma once
#include "enet_block.hpp"
#include "gemm.hpp"
#include "gemm_dense.hpp"
#include "gemm_wide.hpp"
#include "kernels.hpp"
#include <atomic>
#include <utility>
#include <type_traits>

namespace modl {

#ifndef MODL_RT1_MAX
#define MODL_RT1_MAX 2048
#endif
#ifndef MODL_ACC_SHARD_MIN
#define MODL_ACC_SHARD_MIN 64      // workgroups of the block step above which the Gram accumulator is sharded (acc_load_sharded)
#endif
constexpr int kNB = 32;            // atoms per block of the blocked path
// diagnostics (modl_debug_set(MODL_DEBUG_BCD_ACC, 0)): the per-workgroup Gram records instead of the atomic accumulator
extern std::atomic<int> g_bcd_acc;
// diagnostics (modl_debug_set(MODL_DEBUG_ATOM_STAMPS, device pointer to 64 uint64)): cycle sums of the projecting
// workgroup (atom_project_group_kernel), accumulated over the launches ([0] = launches; layout at the kernel)
extern std::atomic<unsigned long long *> g_atom_stamps;
// diagnostics (modl_debug_set(MODL_DEBUG_BCD_TINY, 0)): the separate launches of the blocked update also for small sampled sets
extern std::atomic<int> g_bcd_tiny;
extern std::atomic<int> g_bcd_persist;   // bcd.hip (modl_debug_set(MODL_DEBUG_BCD_PERSIST, ...))
constexpr int kAccWords = 3 * (2 * 136 + 256 + kNB) + 2;   // int64 words of one Gram accumulator (3 bins x packed record + the out-of-range word: kAccStride below)
constexpr int kSetupRows = 8;      // rows per workgroup of bcd_setup_kernel's gathers
constexpr int kAccShards = 4;      // accumulators side by side for large grids (acc_load_sharded; a power of two)
constexpr int kGramRows = 128;     // feature rows per Gram slab
#ifndef MODL_KGROUP
#define MODL_KGROUP 16
#endif
constexpr int kGroup = MODL_KGROUP;         // minimum workgroups per group of the two-level partial reduction
constexpr int kCounters = 64;      // arrival counters: [0] final, [1 + g] group g
constexpr int kAtomGroupMax = 8;   // atoms per launch pair of the grouped atom update (l1 / elastic-net atoms)

__device__ __forceinline__ int64_t sub_row(const int32_t *subset, int64_t f) { return subset ? (int64_t)subset[f] : f; }
// Element (sampled feature f, sweep position c) of the PACKED dictionary of the fused block kernel, stored in the order
// its matrix-core A operands are read: tiles of 32 features x 4 atoms, 512 contiguous bytes each, so the 16-byte-per-lane
// operand load of a wavefront (lane = feature, 4 consecutive atoms; the two halves of the wave take adjacent atom
// groups) is ONE contiguous kilobyte instead of 64 separate rows (k % 4 == 0).
__device__ __forceinline__ int64_t dfrag(int64_t f, int c, int k) {
    return ((f >> 5) * (int64_t)(k >> 2) + (c >> 2)) * 128 + ((f & 31) << 2) + (c & 3);
}

constexpr int kResStride = kNB * kNB + kNB;          // doubles per Gram partial / per CA record
constexpr int kTri = 136;                            // upper triangle (with diagonal) of a 16 x 16 tile
constexpr int kPackStride = 2 * kTri + 256 + kNB;    // fused path: triangle of tile (0,0), tile (0,1), triangle of tile (1,1), old norms: 560 doubles (the Gram is symmetric)
__device__ __forceinline__ int tri_index(int row, int col) { return row * 16 - row * (row - 1) / 2 + (col - row); }   // row <= col

// Sum n Gram records (kResStride doubles each) in a fixed order, all 256 threads of the workgroup; element
// e = tid + 256 q goes to sink(e, sum).  Every load of a chunk of 16 records is issued before the first
// add (a dependent load -> add loop costs one memory round trip per record).
template <int STRIDE, typename Sink>
__device__ __forceinline__ void reduce_records(const double *rec, int n, Sink sink) {
    constexpr int NQ = (STRIDE + 255) / 256;
    double tot[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) tot[q] = 0.0;
    for (int z0 = 0; z0 < n; z0 += 16) {
        double v[NQ][16];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int ec = (e < STRIDE) ? e : 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int z = (z0 + u < n) ? z0 + u : n - 1;
                v[q][u] = rec[(int64_t)z * STRIDE + ec];
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // all requests first: one memory round trip, not NQ
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v[q][u] = (z0 + u < n) ? v[q][u] : 0.0;
            const double c = (((v[q][0] + v[q][1]) + (v[q][2] + v[q][3])) + ((v[q][4] + v[q][5]) + (v[q][6] + v[q][7]))) +
                             (((v[q][8] + v[q][9]) + (v[q][10] + v[q][11])) + ((v[q][12] + v[q][13]) + (v[q][14] + v[q][15])));
            tot[q] += c;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = threadIdx.x + 256 * q;
        if (e < STRIDE) sink(e, tot[q]);
    }
}
// The same sum with 16-byte loads: this thread owns elements 2 e2, 2 e2 + 1 of every record.  A wavefront's memory
// instruction costs the address unit ~16 cycles whatever its width (measured: 4 lanes per cycle), so the 8-byte
// version above, 12 wavefront-loads per record, kept the texture unit busy for ~6 k cycles at 32 records; this one
// needs 5 (four worker wavefronts + the first lanes of the resolver wavefront for the tail of the record).
// Same summation order, same bits.
template <int STRIDE, int CH, typename Sink>
__device__ __forceinline__ void reduce_records_chunks(const double *rec, int n, int e2, bool valid, Sink sink) {
    static_assert(STRIDE % 2 == 0, "records are read as double2");
    static_assert(CH == 16 || CH == 32, "chunks of 16 or 32 records");
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v *base = reinterpret_cast<const d2v *>(rec) + (valid ? e2 : 0);
    d2v tot = {0.0, 0.0};
    for (int z0 = 0; z0 < n; z0 += CH) {
        d2v v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int z = (z0 + u < n) ? z0 + u : n - 1;
            v[u] = base[(int64_t)z * (STRIDE / 2)];
        }
        __builtin_amdgcn_sched_barrier(0);           // all requests first: one memory round trip
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            v[u].x = (z0 + u < n) ? v[u].x : 0.0;
            v[u].y = (z0 + u < n) ? v[u].y : 0.0;
        }
#pragma unroll
        for (int h = 0; h < CH; h += 16) {           // the association of 16 records per round, whatever CH
            const d2v c = (((v[h + 0] + v[h + 1]) + (v[h + 2] + v[h + 3])) + ((v[h + 4] + v[h + 5]) + (v[h + 6] + v[h + 7]))) +
                          (((v[h + 8] + v[h + 9]) + (v[h + 10] + v[h + 11])) + ((v[h + 12] + v[h + 13]) + (v[h + 14] + v[h + 15])));
            tot += c;
        }
    }
    if (valid) {
        sink(2 * e2, tot.x);
        sink(2 * e2 + 1, tot.y);
    }
}
// 17 - 32 records (up to 32 workgroups without pre-summed groups: the metric's shape at reduction 10) in ONE memory
// round trip; otherwise rounds of 16.  Same sums, same bits either way.
template <int STRIDE, typename Sink>
__device__ __forceinline__ void reduce_records_v2(const double *rec, int n, int e2, bool valid, Sink sink) {
    if (n > 16 && n <= 32) reduce_records_chunks<STRIDE, 32>(rec, n, e2, valid, sink);
    else reduce_records_chunks<STRIDE, 16>(rec, n, e2, valid, sink);
}
struct SinkLds {
    double (*M)[kNB + 1]; double *D2;
    __device__ __forceinline__ void operator()(int e, double v) const {
        if (e < kNB * kNB) M[e / kNB][e % kNB] = v;
        else D2[e - kNB * kNB] = v;
    }
};
struct SinkLdsPacked {   // packed record -> full symmetric matrix
    double (*M)[kNB + 1]; double *D2;
    static __device__ __forceinline__ void untri(int e, int &row, int &col) {
        row = 0;
#pragma unroll
        for (int r = 1; r < 16; ++r)
            if (e >= r * 16 - r * (r - 1) / 2) row = r;
        col = row + (e - (row * 16 - row * (row - 1) / 2));
    }
    __device__ __forceinline__ void operator()(int e, double v) const {
        int i, j;
        if (e < kTri) {
            untri(e, i, j);
        } else if (e < kTri + 256) {
            i = (e - kTri) >> 4; j = 16 + ((e - kTri) & 15);
        } else if (e < 2 * kTri + 256) {
            untri(e - kTri - 256, i, j);
            i += 16; j += 16;
        } else {
            D2[e - 2 * kTri - 256] = v;
            return;
        }
        M[i][j] = v;
        M[j][i] = v;
    }
};
// packed record -> the rows the recursion's helper starts from: Base[m][lane] = e_m[lane] for lanes 0-31 (written
// once per launch by the helper wave) and M[lane - 32][m] for lanes 32-63
struct SinkBasePacked {
    double *Base; double *D2;    // Base: [NB][64]
    __device__ __forceinline__ void operator()(int e, double v) const {
        int i, j;
        if (e < kTri) {
            SinkLdsPacked::untri(e, i, j);
        } else if (e < kTri + 256) {
            i = (e - kTri) >> 4; j = 16 + ((e - kTri) & 15);
        } else if (e < 2 * kTri + 256) {
            SinkLdsPacked::untri(e - kTri - 256, i, j);
            i += 16; j += 16;
        } else {
            D2[e - 2 * kTri - 256] = v;
            return;
        }
        Base[j * 64 + 32 + i] = v;
        Base[i * 64 + 32 + j] = v;
    }
};
struct SinkGlobal {
    double *dst;
    __device__ __forceinline__ void operator()(int e, double v) const { dst[e] = v; }
};

// ---- the Gram accumulator: three signed fixed-point bins per entry, units 2^-70, 2^-30 and 2^10 (40 bits each: any
// double of magnitude below 2^50 is represented to 2^-70 - the entries are Gram products of candidate atoms, O(1) in
// dictionary units whatever the scale of the data - and 2^22 contributions fit an int64 bin).
// value = b2 2^10 + b1 2^-30 + b0 2^-70.  A contribution outside that range (|v| >= 2^50, or not a number) raises the
// accumulator's out-of-range word instead, and the readers of that block then sum the per-workgroup records, which
// every workgroup still writes (560 plain stores that nobody reads otherwise): slower, any magnitude, never a wrapped
// integer.  The bins are ABSOLUTE (2^-71): what the recursion needs is precision relative to the squared norms on the
// Gram diagonal, so a NORM entry (`norm_entry`: a diagonal element, an old squared norm) that is positive but below
// 2^-40 - a user-set dictionary of tiny atoms, a tiny norm budget - raises the same word: relative to any norm the
// accumulator accepts its quantisation is then below 2^-31, under the f32 data's own rounding.
constexpr int kAccBins = 3;
constexpr int kAccStride = kAccBins * kPackStride;        // int64 words of the bins; word kAccStride: out of range
static_assert(kAccStride + 2 == kAccWords, "accumulator size");
// The split without a conversion: adding 1.5 * 2^(52 + e) to a value below 2^(51 + e) leaves round(v / 2^e) in the low bits
// of the sum's mantissa (the exponent field is the constant's), and subtracting the constant again gives the rounded value,
// so that the remainder v - round(v / 2^e) 2^e is exact: five additions and three integer subtractions per entry, no
// branch but the one around the top bin (a double -> int64 conversion is software on this part: ~45 instructions and four
// branches per entry before - 600 to 1000 cycles of a wavefront that issues an instruction every 6-7 cycles, four entries
// per lane at the end of every block launch: profiles/r04_ab_look_ahead_stamps.txt, "G detail").  `bad` collects the
// out-of-range lanes (their contribution is dropped: the readers take the records then); acc_flag raises the word once.
__device__ __forceinline__ void acc_add(long long *acc, int idx, double v, bool norm_entry, bool &bad) {
    const bool out = !(fabs(v) < 0x1p50) || (norm_entry && v != 0.0 && fabs(v) < 0x1p-40);
    bad = bad || out;
    const double w = out ? 0.0 : v;
    const double m2 = 0x1.8p62, m1 = 0x1.8p22, m0 = 0x1.8p-18;          // units 2^10, 2^-30, 2^-70
    const double x2 = w + m2;
    const long long b2 = __double_as_longlong(x2) - __double_as_longlong(m2);
    const double r1 = w - (x2 - m2);                                      // |r1| <= 2^9, exact
    const double x1 = r1 + m1;
    const long long b1 = __double_as_longlong(x1) - __double_as_longlong(m1);
    const double r0 = r1 - (x1 - m1);                                     // |r0| <= 2^-31, exact
    const double x0 = r0 + m0;
    const long long b0 = __double_as_longlong(x0) - __double_as_longlong(m0);   // (to nearest)
    unsigned long long *a = reinterpret_cast<unsigned long long *>(acc);
    if (__builtin_expect(b2 != 0, 0)) atomicAdd(a + 2 * kPackStride + idx, (unsigned long long)b2);   // (|v| >= 2^9 only)
    atomicAdd(a + 1 * kPackStride + idx, (unsigned long long)b1);        // (device scope, no return value)
    atomicAdd(a + idx, (unsigned long long)b0);
}
__device__ __forceinline__ void acc_flag(long long *acc, bool bad) {
    if (bad) atomicOr(reinterpret_cast<unsigned long long *>(acc) + kAccStride, 1ull);
}
// this thread's elements 2 e2, 2 e2 + 1 of the accumulated record -> sink; returns the out-of-range word (uniform)
template <typename Sink>
__device__ __forceinline__ bool acc_load(const long long *acc, int e2, bool valid, Sink sink) {
    typedef long long l2v __attribute__((ext_vector_type(2)));
    const l2v *base = reinterpret_cast<const l2v *>(acc) + (valid ? e2 : 0);
    const l2v b0 = base[0], b1 = base[kPackStride / 2], b2 = base[kPackStride];   // (three 16-byte loads, one round trip)
    const long long bad = acc[kAccStride];                                         // (the same trip)
    if (valid) {
        sink(2 * e2, ((double)b2.x * 0x1p10 + (double)b1.x * 0x1p-30) + (double)b0.x * 0x1p-70);
        sink(2 * e2 + 1, ((double)b2.y * 0x1p10 + (double)b1.y * 0x1p-30) + (double)b0.y * 0x1p-70);
    }
    return bad != 0;
}
// The same over kAccShards accumulators (large grids: workgroup b adds to accumulator b % kAccShards - the atomics on one
// address are served one after the other, ~25 ns each, and a launch cannot end before the last one: 157 workgroups on one
// accumulator cost 4 us at the end of every launch, scripts/micro/atomic_drain.hip).  The bins are integers: their sums do
// not depend on the order, the result is the same as with one accumulator.  One round trip.
template <typename Sink>
__device__ __forceinline__ bool acc_load_sharded(const long long *acc, int e2, bool valid, Sink sink) {
    typedef long long l2v __attribute__((ext_vector_type(2)));
    l2v b[kAccShards][3];
    long long bad = 0;
#pragma unroll
    for (int z = 0; z < kAccShards; ++z) {
        const l2v *base = reinterpret_cast<const l2v *>(acc + (size_t)z * kAccWords) + (valid ? e2 : 0);
        b[z][0] = base[0]; b[z][1] = base[kPackStride / 2]; b[z][2] = base[kPackStride];
        bad |= acc[(size_t)z * kAccWords + kAccStride];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (valid) {
        l2v b0 = b[0][0], b1 = b[0][1], b2 = b[0][2];
#pragma unroll
        for (int z = 1; z < kAccShards; ++z) { b0 += b[z][0]; b1 += b[z][1]; b2 += b[z][2]; }
        sink(2 * e2, ((double)b2.x * 0x1p10 + (double)b1.x * 0x1p-30) + (double)b0.x * 0x1p-70);
        sink(2 * e2 + 1, ((double)b2.y * 0x1p10 + (double)b1.y * 0x1p-30) + (double)b0.y * 0x1p-70);
    }
    return bad != 0;
}
__device__ __forceinline__ void reduce_partials(const double *partial, int nslab, double (*M)[kNB + 1], double *D2) {
    reduce_records<kResStride>(partial, nslab, SinkLds{M, D2});
}

// "Last arriver" hand-off (cdna guide, split-K recipe): release our stores, take a ticket, and if we are
// the last of `expected` arrivals acquire the others' stores.  Called by every thread; no spinning.
__device__ __forceinline__ bool arrive_last(unsigned int *counter, unsigned int expected, int *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (ticket == expected - 1);
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// Coefficients of the in-block recursion of one block -> LDS, Cs[j][i] = C[o_i,o_j] / C[o_j,o_j] for i < j
// (zero otherwise); called by every thread of the workgroup.
__device__ __forceinline__ void stage_coef(const double *coef_all, int k, int j0, double *Cs) {
    for (int e = threadIdx.x; e < kNB * kNB; e += blockDim.x) {
        const int i = e / kNB, j = e % kNB;
        const bool ok = j0 + j < k;
        const double v = coef_all[ok ? (int64_t)(j0 + j) * kNB + i : 0];
        Cs[j * kNB + i] = ok ? v : 0.0;
    }
}

// both halves of the wave see (lower half's value, upper half's value)
__device__ __forceinline__ void halves(double z, double &low, double &high) {
    const long long b = __double_as_longlong(z);
    const unsigned int w0 = (unsigned int)(b & 0xffffffffll), w1 = (unsigned int)(b >> 32);
    const auto r0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(w1, w1, false, false);
    low = __longlong_as_double(((long long)r1[0] << 32) | r0[0]);
    high = __longlong_as_double(((long long)r1[1] << 32) | r0[1]);
}

// The alpha recursion of one block, run by ONE wavefront entirely in registers.  With
// u_j = a_j - sum_{i<j} c_ji alpha_i u_i  (c_ji = C[o_i,o_j] / C[o_j,o_j]) write u_j = sum_m T[j][m] a_m and
// S[j] = alpha_j T[j]; with the Gram matrix M of the a_m and Y[j] = M S[j]:
//     T[j]     = e_j    - sum_{i<j} c_ji S[i]          (lanes  0-31, lane x owns component x)
//     M T[j]   = M[:,j] - sum_{i<j} c_ji Y[i]          (lanes 32-63: the SAME instruction stream)
//     |u_j|^2  = T[j] . (M T[j])
// The wave issues one instruction every ~5 cycles and nothing else runs on its critical path, so the
// recursion is bound by its INSTRUCTION COUNT (measured: a version with a shorter dependency chain but more
// instructions was slower).  Hence:
//   * the c_ji are wave-uniform LDS broadcast reads with static addresses (two per ds_read_b128);
//   * the products T[j][x] (M T[j])[x] are formed in BOTH halves (one v_permlane32_swap pair), so their 32-lane
//     sum is four DPP stages inside the rows of 16 lanes plus one v_permlane16_swap — no second half swap (a
//     variant with two f64 MFMAs doing the sum had fewer instructions but was 20 % slower: MFMA issue + hazard
//     wait states);
//   * alpha = min(sqrt(radius) / sqrt(|u|^2), [radius > 0]) needs no compare or select: v_rsq_f64 + one
//     Newton step (rel. error ~1e-14), v_min_f64 (a NaN from |u|^2 <= 0 yields the other operand);
//     radius = 0 gives alpha = 0 (enet.pyx:57), inside the ball alpha = 1 (:65);
//   * per-step uniforms (sqrt(radius_j), [radius_j > 0]) come from LDS as one broadcast read, the results
//     (alpha_j, |u_j|^2) leave through LDS and the new budgets are formed after the loop, one lane per atom.
// The j loop is fully unrolled (static register indices, no branch) and software-pipelined by hand: the
// terms i < j of step j + 1 are accumulated under the serial tail of step j; between alpha_j and the next
// tail sits ONE fma, z_{j+1} = partial_{j+1} - alpha_j (c_{j+1,j} z_j).  Output: CA[j][m] = S[j][m] (the apply
// step forms D_j = sum_m S[j][m] a_m) and the new norm budgets.
// budget_x: the norm budget of atom x of the block before the update (0 beyond nb); jj_x: its atom index;
// scr: >= 4 * NB doubles of LDS scratch private to the wave.
template <typename T>
__device__ __forceinline__ void resolve_wave(const double (*M)[kNB + 1], const double *D2, const double *Cs,
                                             int jj_x, double budget_x, int nb, T *norm_out, double *CAout,
                                             int ca_stride, double *scr, unsigned long long *stamps = nullptr) {
    const int lane = threadIdx.x & 63, x = lane & 31;
    const bool lower = lane < 32;
    double Z[kNB];
    const double rad_x = budget_x + D2[x];                                       // budget + old squared norm
    const bool live_x = (rad_x > 0.0) && (x < nb);
    if (lower) {
        scr[2 * x] = live_x ? sqrt(rad_x) : 0.0;                                 // sqrt(radius_j) ...
        scr[2 * x + 1] = live_x ? 1.0 : 0.0;                                     // ... and the cap of alpha_j
    }
    const double hmask = lower ? 0.0 : 1.0;
    if (stamps && lane == 0) stamps[8] = clock64() + (unsigned long long)(rad_x * 0);
    double part = __builtin_fma(hmask, M[x][0], (lower && x == 0) ? 1.0 : 0.0);   // e_0 | M[:,0]
    double al_prev = 0.0, q_prev = 0.0, z_prev = 0.0;
    // coefficient rows travel one step ahead of their use: row j + 2 is requested at the top of step j (its
    // broadcast reads land under the step's serial tail), row j + 1 is consumed from registers
    double crow[2][kNB];
#pragma unroll
    for (int i = 0; i < kNB; ++i) { crow[0][i] = 0.0; crow[1][i] = Cs[1 * kNB + i]; }
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        if (stamps && lane == 0 && (j % 8) == 0 && j > 0) stamps[8 + j / 8] = clock64();
        if (j + 2 < kNB) {
#pragma unroll
            for (int i = 0; i <= j + 1; ++i) crow[j & 1][i] = Cs[(j + 2) * kNB + i];
        }
        const double z = __builtin_fma(-al_prev, q_prev, part);
        double t, w;
        halves(z, t, w);
        double pr = t * w;                                 // both halves hold the same products
        pr += dpp_perm<0xB1>(pr);                          // four DPP stages inside each row of 16 lanes ...
        pr += dpp_perm<0x4E>(pr);
        pr += dpp_perm<0x141>(pr);
        pr += dpp_perm<0x140>(pr);
        double r0, r1;
        lane_swap<true>(pr, r0, r1);                       // ... and the two rows of a half
        const double nrm = r0 + r1;                        // every lane: |u_j|^2
        // independent of this step's tail: finish S[j-1], start step j + 1
        if (j > 0) Z[j - 1] = al_prev * z_prev;
        double q = 0.0;
        if (j + 1 < kNB) {
            double p0 = __builtin_fma(hmask, M[x][j + 1], (lower && x == j + 1) ? 1.0 : 0.0), p1 = 0;
#pragma unroll
            for (int i = 0; i < j; ++i) {                  // two chains: the wave is issue-bound, not latency-bound here
                const double c = crow[(j + 1) & 1][i];
                if ((i & 1) == 0) p0 -= c * Z[i];
                else p1 -= c * Z[i];
            }
            part = p0 + p1;
            q = crow[(j + 1) & 1][j] * z;
        }
        const double sr = scr[2 * j], cap = scr[2 * j + 1];
        const double y = __builtin_amdgcn_rsq(nrm);                // v_rsq_f64
        const double r = __builtin_fma(-(0.5 * y), nrm * y, 0.5);  // Newton: y <- y + y (1/2 - (y/2)(nrm y))
        const double yn = __builtin_fma(y, r, y);
        double al;
        const double sy = sr * yn;
        asm("v_min_f64 %0, %1, %2" : "=v"(al) : "v"(sy), "v"(cap));   // min(NaN, cap) = cap; no canonicalising v_max
        scr[2 * kNB + 2 * j] = al;
        scr[2 * kNB + 2 * j + 1] = nrm;
        al_prev = al;
        q_prev = q;
        z_prev = z;
    }
    Z[kNB - 1] = al_prev * z_prev;
    if (lower) {
#pragma unroll
        for (int j = 0; j < kNB; ++j) CAout[j * ca_stride + x] = Z[j];
        if (norm_out && x < nb) {
            const double al = scr[2 * kNB + 2 * x], nrm = scr[2 * kNB + 2 * x + 1];
            norm_out[jj_x] = (T)(rad_x - al * al * nrm);
        }
    }
}

// ---- The same recursion on TWO wavefronts (fused block kernel).  One wavefront is bound by its instruction count:
// of the ~430 cycles per atom only ~200 are the dependency chain z_j -> |u_j|^2 -> alpha_j (32-lane dot: 4 DPP stages
// + 2 swaps = 144 cycles, rsqrt + Newton + min = 47; scripts/micro/chain_lat.hip), the rest are the partial sums
// sum_{i<j} c_ji S[i] of the rows ahead and the coefficient reads they need.  So a CHAIN wave keeps the chain and the
// two nearest terms (i = j - 1 on the chain, i = j - 2 beside it), and a HELPER wave on another SIMD keeps a running
// P_m = (e_m | M[:,m]) - sum_{i <= m-3} c_mi S[i] for every row m, right-looking: when S[i] arrives it first completes
// and publishes P_{i+3}, then updates the rows behind it.  The two talk through LDS mailboxes (rings of 8 slots, one
// monotonic counter each: data store, then counter store - the LDS executes a wavefront's operations in order - and
// the reader requests counter, then data, in ONE round trip, one step before it needs them).  The helper has two chain
// steps (~600 cycles) to turn S[i] into P_{i+3}: one LDS round trip each way (~76 cycles) + one fma.
// Waiting is a spin on LDS; both wavefronts belong to one workgroup, hence are resident together.
typedef __attribute__((address_space(3))) volatile double lds_vf64;   // (a generic volatile pointer would turn into
typedef __attribute__((address_space(3))) volatile int lds_vi32;      //  flat accesses with a wait behind each)
constexpr int kMbox = 8;                                               // mailbox slots (steps in flight <= 3)
struct ResolveMail {
    double *Pm, *Zm;        // [kMbox][64]
    int *pcount, *zcount;   // rows published by the helper / S rows published by the chain wave
};

template <typename T>
__device__ __forceinline__ void resolve_chain(const double *D2, const double *Cs, int jj_x, double budget_x, int nb,
                                              T *norm_out, double *scr, const ResolveMail &mb,
                                              unsigned long long *stamps = nullptr) {
    const int lane = threadIdx.x & 63, x = lane & 31;
    const bool lower = lane < 32;
    lds_vf64 *Pm = (lds_vf64 *)mb.Pm;
    lds_vf64 *Zm = (lds_vf64 *)mb.Zm;
    lds_vi32 *pcount = (lds_vi32 *)mb.pcount;
    lds_vi32 *zcount = (lds_vi32 *)mb.zcount;
    const double rad_x = budget_x + D2[x];                                       // budget + old squared norm
    const bool live_x = (rad_x > 0.0) && (x < nb);
    if (lower) {
        scr[2 * x] = live_x ? sqrt(rad_x) : 0.0;                                 // sqrt(radius_j) ...
        scr[2 * x + 1] = live_x ? 1.0 : 0.0;                                     // ... and the cap of alpha_j
    }
    if (stamps && lane == 0) { stamps[8] = clock64() + (unsigned long long)(rad_x * 0); }
    double al_prev = 0.0, q_prev = 0.0, z_prev = 0.0, Z1 = 0.0, Z2 = 0.0;        // Z1 = S[j-1], Z2 = S[j-2] (| Y)
    int rn = *pcount;
    double Pn = Pm[lane];
    while (__builtin_amdgcn_readfirstlane(rn) < 1) { rn = *pcount; Pn = Pm[lane]; }
    double srn = scr[0], capn = scr[1], c2n = 0.0, c1n = Cs[1 * kNB + 0];
    // Groups of kMbox steps (static mailbox slots inside a group; everything else the steps address advances with the
    // group), all groups unrolled: as a rolled loop (3.6 KB of code instead of 15 KB that is executed once) the same
    // steps took 14.5 k cycles per block instead of 13.4 k, measured.  The boundary cases are data-driven: step 0
    // publishes a dummy row under count 0, and the last step requests row 32, which the helper "publishes" (count 33)
    // together with row 31.
    const double *csr = Cs;                                  // row j of the coefficients
    const double *scj = scr;                                 // (sqrt(radius_j), cap_j)
    double *sco = scr + 2 * kNB;                             // (alpha_j, |u_j|^2)
#pragma unroll
    for (int jb = 0; jb < kNB; jb += kMbox) {
        if (stamps && lane == 0 && jb > 0) stamps[8 + jb / 8] = clock64();
#pragma unroll
        for (int u = 0; u < kMbox; ++u) {
            const int j = jb + u;
            if (stamps && lane == 0 && jb == 0) stamps[24 + u] = clock64();
            const double P = Pn, sr = srn, cap = capn, c2 = c2n, c1 = c1n;
            {                                                // S[j-1] | Y[j-1] -> helper (and the apply step)
                const double Zp = al_prev * z_prev;
                Zm[((u + kMbox - 1) % kMbox) * 64 + lane] = Zp;
                *zcount = j;
                Z2 = Z1; Z1 = Zp;
            }
            {                                                // everything step j + 1 needs from LDS is requested now
                rn = *pcount;
                Pn = Pm[((u + 1) % kMbox) * 64 + lane];
                srn = scj[2 * (u + 1)]; capn = scj[2 * (u + 1) + 1];
                c2n = csr[(u + 1) * kNB + u - 1];            // c_{j+1,j-1}
                c1n = csr[(u + 2) * kNB + u + 1];            // c_{j+2,j+1}
            }
            const double part = __builtin_fma(-c2, Z2, P);   // row j: P_j holds the terms i <= j - 3
            const double z = __builtin_fma(-al_prev, q_prev, part);
            double t, w;
            halves(z, t, w);
            double pr = t * w;                               // both halves hold the same products
            pr += dpp_perm<0xB1>(pr);
            pr += dpp_perm<0x4E>(pr);
            pr += dpp_perm<0x141>(pr);
            pr += dpp_perm<0x140>(pr);
            double r0, r1;
            lane_swap<true>(pr, r0, r1);
            const double nrm = r0 + r1;                      // every lane: |u_j|^2
            const double q = c1 * z;                         // c_{j+1,j} z_j
            const double y = __builtin_amdgcn_rsq(nrm);
            const double r = __builtin_fma(-(0.5 * y), nrm * y, 0.5);
            const double yn = __builtin_fma(y, r, y);
            double al;
            const double sy = sr * yn;
            asm("v_min_f64 %0, %1, %2" : "=v"(al) : "v"(sy), "v"(cap));
            sco[2 * u] = al;
            sco[2 * u + 1] = nrm;
            al_prev = al; q_prev = q; z_prev = z;
            __builtin_amdgcn_sched_barrier(0);               // (the check below waits for the LDS: not inside the chain)
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(rn) < j + 2, 0)) {
                do {
                    rn = *pcount;
                    Pn = Pm[((u + 1) % kMbox) * 64 + lane];
                } while (__builtin_amdgcn_readfirstlane(rn) < j + 2);
            }
        }
        csr += kMbox * (kNB + 1);
        scj += 2 * kMbox;
        sco += 2 * kMbox;
    }
    Zm[((kNB - 1) % kMbox) * 64 + lane] = al_prev * z_prev;
    *zcount = kNB;
    if (lower && norm_out && x < nb) {
        const double al = scr[2 * kNB + 2 * x], nrm = scr[2 * kNB + 2 * x + 1];
        norm_out[jj_x] = (T)(rad_x - al * al * nrm);
    }
}

// CsT[i][m] = Cs[m][i]: the coefficients that multiply S[i], contiguous in m
// gS (optional): every row of S also goes to global memory as it is produced, write-through (the persistent launch: the row
// workgroups read it; by the end of the recursion all rows but the last have long arrived)
__device__ __forceinline__ void resolve_helper(const double *Base, const double *CsT, double *CAout,
                                               int ca_stride, const ResolveMail &mb, double *gS = nullptr) {
    const int lane = threadIdx.x & 63, x = lane & 31;
    const bool lower = lane < 32;
    lds_vf64 *Pm = (lds_vf64 *)mb.Pm;
    lds_vf64 *Zm = (lds_vf64 *)mb.Zm;
    lds_vi32 *pcount = (lds_vi32 *)mb.pcount;
    lds_vi32 *zcount = (lds_vi32 *)mb.zcount;
    double P[kNB];                                           // e_m | M[:,m]: one conflict-free read per row
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        P[m] = Base[m * 64 + lane];
        Pm[m * 64 + lane] = P[m];
    }
    *pcount = 3;
#pragma unroll
    for (int m = 3; m < kNB; ++m) P[m] = Base[m * 64 + lane];
    // the coefficients that multiply S[i] (CsT row i) are requested one iteration ahead: no LDS round trip between the
    // arrival of S[i] and the updates; as 16-byte pairs from one base register (an 8-byte-aligned start makes the
    // compiler emit ds_read2_b64 with an address register set up per request: two extra instructions each)
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v *CsT2 = reinterpret_cast<const d2v *>(CsT);   // [NB][NB / 2]
    d2v cc[kNB / 2], cn[kNB / 2];
#pragma unroll
    for (int h = 1; h < kNB / 2; ++h) cc[h] = CsT2[h];
#pragma unroll
    for (int i = 0; i < kNB; ++i) {
        if (i + 1 < kNB) {
#pragma unroll
            for (int h = (i + 4) / 2; h < kNB / 2; ++h) cn[h] = CsT2[(i + 1) * (kNB / 2) + h];
        }
        __builtin_amdgcn_sched_barrier(0);
        int ready = *zcount;
        double Zi = Zm[(i % kMbox) * 64 + lane];
        while (__builtin_amdgcn_readfirstlane(ready) < i + 1) {
            ready = *zcount;
            Zi = Zm[(i % kMbox) * 64 + lane];
        }
        // (scheduler fences and pinned results: left alone, the compiler sinks every update of a row to the row's first
        // use - the left-looking form - and keeps all coefficients and S rows live: spills)
        __builtin_amdgcn_sched_barrier(0);
        if (i + 3 < kNB) {                                   // the most urgent row first
            P[i + 3] = __builtin_fma(-cc[(i + 3) / 2][(i + 3) % 2], Zi, P[i + 3]);
            Pm[((i + 3) % kMbox) * 64 + lane] = P[i + 3];
            *pcount = (i + 3 == kNB - 1) ? kNB + 1 : i + 4;  // (+ the row the chain wave's last step asks for)
        }
        if (lower) CAout[i * ca_stride + x] = Zi;            // S[i] for the apply step
        if (lower && gS)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(gS + i * kNB + x), (unsigned long long)__double_as_longlong(Zi),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (an sc1 store)
#pragma unroll
        for (int m = i + 4; m < kNB; ++m) {
            P[m] = __builtin_fma(-cc[m / 2][m % 2], Zi, P[m]);
            asm volatile("" : "+v"(P[m]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = (i + 4) / 2; h < kNB / 2; ++h) cc[h] = cn[h];
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global
// store to be acknowledged (s_waitcnt vmcnt(0)); the phases of the block kernel only hand LDS tiles to each
// other, their global stores may stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int kCaStride = kNB + 2;      // LDS row stride (doubles) of the S matrix: 2-way instead of 16-way conflicts
constexpr int kApStride = kNB + 4;      // LDS row stride (floats) of the staged a-tile

struct BcdBlockArgs {
    float *Dt;                      // PACKED dictionary [s][k]: row = sampled feature, column = sweep position
    const float *Bt, *CP, *cdiag;   // packed B [s][k]; CP [k][k] in sweep coordinates (rows and columns)
    const int32_t *frozen, *order;
    float *a;                       // [s][NB] the a-tile: block b - 1's on entry, block b's on exit
    double *rec_out, *grec_out;     // Gram records / group sums written by this launch
    const double *rec_in, *grec_in; // ... written by the previous launch
    // Accumulator mode (acc_in / acc_out non-null; replaces the records): every workgroup ADDS its Gram contribution to ONE
    // record of fixed-point bins with integer atomics - integer addition is associative, so the sum does not depend on
    // the order of arrival: deterministic, identical on every GPU - and the next launch reads 13 KB instead of one
    // 4.5 KB record per workgroup (134 KB at the metric's shape).  Three buffers in rotation: read / add / being cleared.
    long long *acc_out;
    const long long *acc_in;
    long long *acc_zero;
    const double *coef_all;
    const float *norm_in;           // norm budgets as they were before this dictionary update, in sweep order
    float *norm_out;                // comp_norm (written by workgroup 0 only)
    float *Dt_out;                  // the real dictionary [p][k]: every applied column also goes straight back ...
    const int32_t *subset;          // ... to row subset[f] (null: identity), column order[jj]
    unsigned int *counter;
    unsigned long long *stamps;     // optional phase timestamps of workgroup 0 (diagnostics)
    int64_t s;
    int k, j0, nb, j0_prev, nb_prev, group;   // k: atoms of the PACKED arrays (a multiple of 4, dead atoms behind the real ones)
    int kout;                                 // row stride of Dt_out: the real number of atoms
    int shards;                               // 1, or kAccShards accumulators side by side (acc_load_sharded)
};

// Riding along (see StatsRider in kernels.hpp): the workgroups behind the first `nslab` ones each take one 32 x 32
// tile of the deferred statistics product.  The block kernel needs one compute unit per workgroup (registers),
// and at the metric's shape (s = 1000) 32 of the 256 are busy with it, mostly waiting for the resolver wave: the
// tiles a launch carries run on the other ones, a few microseconds each, and are done before the block step is.
struct BcdRiderArgs {
    DenseProblem<float, EpiStatsSkip<float>> P;   // 32 x 32 tiles (gemm_stats_tile) ...
    WideProblem<EpiStatsSkip<float>> W;           // ... or k-wide tiles of `wide` features (gemm_wide_tile): X fetched once
    int wide = 0;                   // 0: P; 32 / 64: W with that many features per tile
    int t0 = 0, t1 = 0;             // tiles [t0, t1) ride with this launch
    int nslab = 0;                  // workgroups of the block step proper
    unsigned long long *dbg = nullptr;   // diagnostics: stamps of the first riding tile of a launch
    StageRide stage;                // src != null: the workgroup behind the riding tiles copies the next minibatch's parameters
};

__device__ __forceinline__ void bcd_rider_tile(const BcdRiderArgs &r, char *smem) {
    const int id = (int)blockIdx.x - r.nslab + r.t0;
    if (id >= r.t1) {
        if (id == r.t1 && r.stage.src)               // (one workgroup: the next minibatch's parameter block, kernels.hpp)
            stage_copy(r.stage.src, r.stage.dst, r.stage.off16, r.stage.n16, r.stage.ack, r.stage.use, (int)threadIdx.x, 384);
        return;
    }
    if (threadIdx.x >= 256) return;                  // the product uses four waves
    if (r.wide == 32) gemm_wide_tile<32, EpiStatsSkip<float>, 128>(r.W, id, smem, (id == r.t0 && r.dbg) ? r.dbg : nullptr);
    else gemm_stats_tile<EpiStatsSkip<float>>(r.P, id, smem);     // the very tile of gemm_stats_pair_kernel: same bits
}


// ---- the persistent launch (bcd_persist.hip) ---------------------------------------------------------------------------
// per-block accumulator of the look-ahead pieces: X = <a', N'> (32 x 32), the packed Gram matrix of the previous block's
// candidates M' = <a', a'> (528), the packed <N', N'> (528), the old squared norms of the block's columns (32)
constexpr int kPX = 0, kPMp = kNB * kNB, kPNN = kPMp + 2 * kTri + 256, kPD2 = kPNN + 2 * kTri + 256;
constexpr int kPEntries = kPD2 + kNB;                       // 2112
constexpr int kPAccWords = kAccBins * kPEntries + 2;        // three fixed-point bins per entry + the out-of-range word + the top-bins-used word
constexpr int kPersistBlocksMax = 16;                       // 512 atoms
constexpr int kPersistRowsMax = 255;                        // row workgroups (+ the resolver: co-resident on 256 compute units)
constexpr int kPersistStampWords = 192;
constexpr long long kPersistSentinel = 0x7ff8c0dec0dec0dell;   // fills the S buffers before a persistent launch: a NaN no recursion produces
static_assert(kPEntries % 2 == 0 && (kPAccWords * 8) % 16 == 0, "16-byte pairs");
struct BcdPersistArgs {
    const float *DsP;               // packed sampled rows of the dictionary [s][k], fragment order (bcd_setup_kernel), read once
    const float *BsP;               // [s][k] sampled rows of B_
    const float *CPP;               // C in sweep coordinates, fragment order, the in-block part masked
    const float *cdiag;
    const int32_t *frozen, *order, *subset;
    const double *coef_all;         // [k][32] recursion coefficients (against the atoms of the same block)
    const double *qcoef;            // [k][32] Q against the atoms of the block BEFORE
    const float *norm_in;           // budgets before this update, sweep order
    float *norm_out;                // comp_norm
    float *Dt_out;                  // the real dictionary [p][kout]
    long long *acc;                 // [nblk][shards][kPAccWords], zero on entry
    double *rec;                    // [2][nrow][kPEntries] per-workgroup records (the out-of-range fallback)
    double *Sbuf;                   // [nblk][32 * 32]
    unsigned int *arrive, *sflag;   // [nblk] each, zero on entry
    unsigned int *err;              // raised by a wait that gave up (zero on entry): every workgroup of THIS launch leaves
    unsigned int *flags;            // optional, pinned host memory, never cleared by the kernels: [0] raised when a launch gave up
                                    // AFTER its first block had been resolved (the update is incomplete: modl_somf_status ->
                                    // MODL_ETIMEOUT), [1] counts the launches that gave up BEFORE it - nothing had been applied - and
                                    // were completed by the resolver workgroup alone (persist_recover)
    const float *C;                 // [kout][kout] the statistics in natural order (persist_recover only)
    unsigned long long *stamps;     // diagnostics build: [kPersistStampWords]
    int64_t s;
    int k, kout, nblk, nrow, shards;
    int expect;                     // arrivals the resolver waits for per block: nrow
    int inject = 0;                 // diagnostics build: 3 / 4 - wait for one arrival more than will come at block 0 / from block 1 on
};
size_t bcd_persist_lds(int kp, int RT);
bool bcd_persist_fits(int kp, int RT, int nrow, size_t extra_lds);   // resident together on the current device? (occupancy query, cached)
// grid: the resolver, nrow row workgroups of 32 RT rows, extra_wgs riding workgroups (rider.nslab must be nrow + 1)
int launch_bcd_persist(hipStream_t stream, const BcdPersistArgs &p, const BcdRiderArgs &rider, int extra_wgs, size_t extra_lds,
                       int RT);

}  // namespace modl
