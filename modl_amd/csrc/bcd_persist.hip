// The blocked dictionary update as ONE persistent launch (round 5).
//
// Replaces the same thing as bcd.hip's blocked path: DictFact._update_dict, variational branch with l2 atoms
// (reference: modl/decomposition/dict_fact.py:650-715, the atom sweep :672-694; enet.pyx:38-122 for the l2 projection),
// f32, up to 512 atoms.  The algebra is bcd.hip's (blocks of 32 atoms in sweep order, every candidate of a block a
// combination of the block's alpha-independent vectors, the alpha recursion on the block's 32 x 32 Gram matrix in
// double precision); what changes is WHO does WHAT, and WHEN:
//
//   * nrow ROW workgroups, each with its 32 / 64 sampled feature rows of the dictionary resident in LDS for the whole
//     update (32 KB per 32 rows at k = 256; the one-launch-per-block kernel re-reads them from L2 every block);
//   * ONE RESOLVER workgroup that owns the alpha recursions (the two-wavefront resolve_chain / resolve_helper of
//     bcd_shared.hpp, on a compute unit of its own: nothing else issues on the chain's SIMD);
//   * LOOK-AHEAD: the Gram matrix of block b does not wait for the recursion of block b - 1.  With N'_b the candidates
//     of block b with block b - 1 LEFT OUT of the product, a_{b-1} the candidates of block b - 1 and
//     Dnew_{b-1} = a_{b-1} S_{b-1}^T the atoms it ends up with,
//         a_b = N'_b - Dnew_{b-1} Q,           Q[i][c] = C[o_i, o_c] / C[o_c, o_c],
//         <a_b, a_b> = <N', N'> - P X - (P X)^T + P M' P^T,     P = Q^T S_{b-1},  X = <a_{b-1}, N'_b>,  M' = <a_{b-1}, a_{b-1}>,
//     and <N', N'>, X, M' only need S_{b-2}: the row workgroups accumulate them (fixed-point integer atomics: the sum
//     does not depend on the order of arrival) WHILE the resolver runs the recursion of block b - 1; when that
//     recursion ends the resolver turns the pieces into the Gram matrix of block b with four 32^3 products on the f64
//     matrix cores (gram_ahead) and starts the next recursion.  The per-block critical path is recursion + transform;
//     the reduction over the features, its atomics' drain and both cross-workgroup hand-offs overlap it.
//
// Hand-offs (cdna guide, Guideline 16: agent-scope release on the producer, relaxed polling by ONE lane, one
// agent-scope acquire on the consumer, every spin bounded):
//   rows -> resolver : atomics into acc[b], then one arrival per workgroup on arrive[b];
//   resolver -> rows : S_b (32 x 32 doubles) into Sbuf[b], then sflag[b].
// One accumulator, one record set parity, one S buffer and one pair of flags PER BLOCK (zeroed by bcd_setup_kernel):
// nothing is reused inside a launch, so no hand-off needs an acknowledgement.
//
// Exactness: the sweep is the reference's in exact arithmetic (same identity as bcd.hip, plus the look-ahead identity
// above); f32 roundings differ from the one-launch-per-block kernel in the summation order of the main product
// (16 x 16 x 4 matrix-core tiles, no split over the contraction) and in a = N' - (Dnew C) / diag being formed in two
// steps.  Contributions outside the accumulator's range raise its out-of-range word and the resolver sums the
// per-workgroup records instead (any magnitude), as in bcd.hip.
#include "bcd_shared.hpp"
#include <mutex>
#include <vector>

namespace modl {

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef long long l2v __attribute__((ext_vector_type(2)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

constexpr int kTS = kNB + 2;                  // LDS row stride (doubles) of the 32 x 32 f64 matrices (= kCaStride)
constexpr int kTF = kNB + 4;                  // LDS row stride (floats) of the f32 tiles (16-byte aligned rows)
static_assert(kTS == kCaStride, "S is an operand of the Gram-domain transform");
constexpr long long kSentinel = kPersistSentinel;   // what bcd_setup_kernel fills the S buffers with (BcdPersistArgs::Sbuf)
constexpr unsigned kSpinLimit = 1u << 21;     // polls (each a memory round trip + s_sleep) before a wait gives up

__device__ __forceinline__ double i2d(long long b) {          // exact for |b| < 2^51 (a double -> int64 conversion and
    return __longlong_as_double(b + 0x4338000000000000ll) - 0x1.8p52;   // its inverse are software on this part)
}

// this workgroup's contribution to entry idx of a look-ahead accumulator (bcd_shared.hpp: acc_add, other strides): three signed
// fixed-point bins, units 2^-70, 2^-30 and 2^10 - integer addition is associative, the sums do not depend on the order of
// arrival.  The top bin only when it is not zero (|v| >= 2^9: `top`, the readers skip those bins unless told).
// (Measured and not kept: ONE 64-bit bin per entry in units of 2^-48 for everything that is not a norm - half the atomics, but
//  what a run produces after a few hundred minibatches has candidates of rarely used atoms with entries far above any fixed
//  range, the wide bins then have to be read for every entry, and agreeing on that inside the resolver cost more than it saved.)
__device__ __forceinline__ void pacc_add(long long *acc, int idx, double v, bool norm_entry, bool &bad, bool &top) {
    const bool out = !(fabs(v) < 0x1p50) || (norm_entry && v != 0.0 && fabs(v) < 0x1p-40);
    bad = bad || out;
    const double w = out ? 0.0 : v;
    const double m2 = 0x1.8p62, m1 = 0x1.8p22, m0 = 0x1.8p-18;          // units 2^10, 2^-30, 2^-70
    const double x2 = w + m2;
    const long long b2 = __double_as_longlong(x2) - __double_as_longlong(m2);
    const double r1 = w - (x2 - m2);
    const double x1 = r1 + m1;
    const long long b1 = __double_as_longlong(x1) - __double_as_longlong(m1);
    const double r0 = r1 - (x1 - m1);
    const double x0 = r0 + m0;
    const long long b0 = __double_as_longlong(x0) - __double_as_longlong(m0);
    unsigned long long *a = reinterpret_cast<unsigned long long *>(acc);
    if (__builtin_expect(b2 != 0, 0)) {
        atomicAdd(a + 2 * kPEntries + idx, (unsigned long long)b2);
        top = true;
    }
    atomicAdd(a + 1 * kPEntries + idx, (unsigned long long)b1);
    atomicAdd(a + idx, (unsigned long long)b0);
}

__device__ __forceinline__ int dl_idx(int frow, int c, int KQ) {        // element (row of the workgroup, sweep position) of the LDS-resident rows
    return (((frow >> 5) * KQ + (c >> 2)) << 7) + ((frow & 31) << 2) + (c & 3);
}

// ---- hand-off payloads: write-through stores / cache-bypassing loads (sc1), so that neither side needs a fence ------------
// (guide, Guideline 16 R1: an agent-scope release is a write-back of the XCD's whole L2 - it would flush every other
//  workgroup's dirty dictionary rows as well - and an agent-scope acquire invalidates it; a relaxed agent-scope atomic
//  store / load of up to 8 bytes IS an sc1 store / load)
__device__ __forceinline__ void store_sc1(double *ptr, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ptr), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double *ptr) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ long long load_sc1(const long long *ptr) {
    return (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- bounded waits --------------------------------------------------------------------------------------------------
// ONE lane polls ONE word (relaxed, agent scope); the payload behind it is read with sc1 loads (no acquire).
// A wait that gives up raises the launch's error word: every other wait of the launch then ends as well.  What that MEANS is
// decided by the resolver alone (persist_resolver): before its first block is resolved nothing has been applied anywhere and
// it completes the update by itself (persist_recover); afterwards the update is incomplete and it raises the plan's flag.
__device__ __forceinline__ void give_up(unsigned int *err) {
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void raise_incomplete(const BcdPersistArgs &p) {
    if (p.flags) __hip_atomic_store(p.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool poll_word(unsigned int *word, unsigned int target, unsigned int *err, unsigned limit = kSpinLimit) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) break;
        if (spins > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            give_up(err);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}
// every wave's (write-through) stores and atomics drained, then ONE lane signals
__device__ __forceinline__ void signal_word(unsigned int *word, bool add) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (add) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the Gram-domain transform (resolver) -------------------------------------------------------------------------------
template <bool BT>   // C tile (t1, t2) of A B (BT: of A B^T), 16 x 16, contraction over 32; all matrices [32][kTS] in LDS
__device__ __forceinline__ d4v tile32(const double *Am, const double *Bm, int t1, int t2, int lane) {
    d4v c = {0.0, 0.0, 0.0, 0.0};
    const double *ap = Am + (t1 * 16 + (lane & 15)) * kTS + (lane >> 4);
    const double *bp = BT ? Bm + (t2 * 16 + (lane & 15)) * kTS + (lane >> 4) : Bm + (lane >> 4) * kTS + t2 * 16 + (lane & 15);
    double fa[kNB / 4], fb[kNB / 4];
#pragma unroll
    for (int kk = 0; kk < kNB / 4; ++kk) { fa[kk] = ap[4 * kk]; fb[kk] = BT ? bp[4 * kk] : bp[4 * kk * kTS]; }
    d4v c1 = {0.0, 0.0, 0.0, 0.0};                    // (two chains: a dependent f64 matrix-core product waits for its predecessor)
#pragma unroll
    for (int kk = 0; kk < kNB / 4; kk += 2) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk], fb[kk], c, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk + 1], fb[kk + 1], c1, 0, 0, 0);
    }
    return c + c1;
}
__device__ __forceinline__ void tile32_store(double *Cm, const d4v &c, int t1, int t2, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(t1 * 16 + (lane >> 4) + 4 * r) * kTS + t2 * 16 + (lane & 15)] = c[r];
}
// Called by all six wavefronts of the resolver (waves 0-3 compute one 16 x 16 tile each).  On entry: NNs = <N', N'>
// (element (i, j) at NNs[i * kTS + j], both triangles), Qt[c][i] = Q[i][c], Ss = S of block b - 1 (left intact: it is
// being published), Mp = M', Xs = X (X[m][c] = <a_{b-1,m}, N'_c>); on exit Base holds the Gram matrix of block b's
// candidates (Base[j * 64 + 32 + i] = element (i, j)); Qt is overwritten.  Four 32^3 products on the f64 matrix cores:
// v_mfma_f64_16x16x4 issues every 64 cycles on this part, 512 cycles per product and SIMD at best.
// Run by waves 0-3 ONLY (one 16 x 16 tile each); they meet at an LDS counter (sync4), so that waves 4 and 5 - one of them is
// still waiting for the S it has published to be written through - are not part of the transform's synchronisation.
__device__ __forceinline__ void sync4(int *cnt, int target, int lane) {
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // this wave's LDS stores have landed
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__builtin_amdgcn_readfirstlane(*(lds_vint *)cnt) < target) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void gram_ahead(double *Qt, const double *Ss, const double *Mp, const double *Xs, double *Ps, double *Zs,
                                           const double *NNs, double *Base, int wid, int lane, int *cnt, int base_count,
                                           int *nn_cnt, int nn_target, unsigned long long *st) {
    const int t1 = wid >> 1, t2 = wid & 1;
    tile32_store(Ps, tile32<false>(Qt, Ss, t1, t2, lane), t1, t2, lane);                           // P = Q^T S
    sync4(cnt, base_count + 4, lane);
    if (st && wid == 0 && lane == 0) st[0] = clock64();
    {
        const d4v R = tile32<false>(Ps, Xs, t1, t2, lane);                                         // R = P X
        const d4v Z = tile32<true>(Mp, Ps, t1, t2, lane);                                          // Z = M' P^T
        tile32_store(Qt, R, t1, t2, lane);
        tile32_store(Zs, Z, t1, t2, lane);
    }
    sync4(cnt, base_count + 8, lane);
    if (st && wid == 0 && lane == 0) st[1] = clock64();
    {
        const d4v V = tile32<false>(Ps, Zs, t1, t2, lane);                                         // V = P Z
        {   // <N', N'> comes from waves 4, 5 (they loaded it under the first two steps)
            typedef __attribute__((address_space(3))) volatile int lds_vint;
            while (__builtin_amdgcn_readfirstlane(*(lds_vint *)nn_cnt) < nn_target) __builtin_amdgcn_s_sleep(0);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = t1 * 16 + (lane >> 4) + 4 * r, c2 = t2 * 16 + (lane & 15);
            Base[c2 * 64 + 32 + c] = NNs[c * kTS + c2] + ((V[r] - Qt[c * kTS + c2]) - Qt[c2 * kTS + c]);
        }
    }
    if (st && wid == 0 && lane == 0) st[2] = clock64();
}

// Entry e of a block's accumulated pieces -> where the transform wants it, as an offset (doubles) from the Mp matrix: the
// symmetric matrices arrive packed (upper triangles of the diagonal 16 x 16 tiles + the off-diagonal tile) and are stored
// as their UPPER triangles only (mirror_upper fills the rest); the table is built once per launch.
constexpr int kOffMp = 0, kOffXs = kNB * kTS, kOffNN = 4 * kNB * kTS, kOffD2 = 5 * kNB * kTS;   // (Mp, Xs, Ps, Zs, NNs, D2n are contiguous)
__device__ __forceinline__ int piece_offset(int e, bool mirrored) {           // mirrored: element (j, i) of a symmetric matrix
    if (e < kPMp) return kOffXs + (e >> 5) * kTS + (e & 31);
    if (e >= kPD2) return kOffD2 + (e - kPD2);
    const int base = e < kPNN ? kOffMp : kOffNN;
    int qq = e < kPNN ? e - kPMp : e - kPNN, i, j;
    if (qq >= kTri && qq < kTri + 256) {
        i = (qq - kTri) >> 4; j = 16 + ((qq - kTri) & 15);
    } else {
        const int hi = qq >= kTri + 256;
        if (hi) qq -= kTri + 256;
        int row = 0;
        for (int r = 1; r < 16; ++r)
            if (qq >= r * 16 - r * (r - 1) / 2) row = r;
        i = row + 16 * hi; j = row + (qq - (row * 16 - row * (row - 1) / 2)) + 16 * hi;
    }
    return mirrored ? base + j * kTS + i : base + i * kTS + j;
}

// The accumulated pieces of block b -> LDS, by the NT threads of the calling waves (tv: 0 .. NT - 1): 2112 entries x two
// (rarely three) integer bins, read with 16-byte sc1 loads (the atomics executed at the memory side; nothing of them may
// come from this XCD's caches), summed over the shards; out of range: the per-workgroup records, in workgroup order.
// P0, P1: the range of PAIRS of entries this call brings in
constexpr int kPairsXM = kPNN / 2, kPairsAll = kPEntries / 2;                    // X and M' | <N', N'> and the old norms
template <int NT, int P0, int P1>
__device__ __forceinline__ void load_pieces(const BcdPersistArgs &p, int b, int tv, double *Mp, const unsigned short *dtab,
                                            const unsigned short *dtab2) {
    constexpr int NP = kPEntries / 2, NJ = (P1 - P0 + NT - 1) / NT;
    const long long *acc = p.acc + (size_t)b * p.shards * kPAccWords;
    l2v bins[NJ][2];
    long long bad = 0, top = 0;
    for (int z = 0; z < p.shards; ++z) {
        const long long *az = acc + (size_t)z * kPAccWords;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<long long *>(az), 0, kPAccWords * 8, 0x00020000);
        u4v raw[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int e2 = P0 + tv + NT * j, ec = e2 < P1 ? e2 : P0;
#pragma unroll
            for (int w = 0; w < 2; ++w) raw[j][w] = __builtin_amdgcn_raw_buffer_load_b128(rs, (w * NP + ec) * 16, 0, 16);   // (aux 16: sc1)
        }
        bad |= load_sc1(az + 3 * kPEntries);
        top |= load_sc1(az + 3 * kPEntries + 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                l2v v;
                __builtin_memcpy(&v, &raw[j][w], 16);
                if (z == 0) bins[j][w] = v;
                else bins[j][w] += v;
            }
    }
    if (__builtin_expect(bad != 0, 0)) {
        const double *rec = p.rec + (size_t)(b & 1) * p.nrow * kPEntries;
        for (int e = 2 * P0 + tv; e < 2 * P1; e += NT) {
            double t = 0.0;
            for (int z = 0; z < p.nrow; ++z) t += load_sc1(rec + (size_t)z * kPEntries + e);
            Mp[dtab[e]] = t;
            Mp[dtab2[e]] = t;
        }
        return;
    }
    // value = b2 2^10 + b1 2^-30 + b0 2^-70: an integer added to the bit pattern of 1.5 * 2^(52 + u) lands in its mantissa with unit
    // 2^u, so (as_double(bits(m) + b) - m) IS b 2^u, exactly (|b| < 2^51) - no int64 -> double conversion (software on this part)
    const double m2 = 0x1.8p62, m1 = 0x1.8p22, m0 = 0x1.8p-18;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e2 = P0 + tv + NT * j;
        if (e2 < P1) {
            const unsigned int d = *reinterpret_cast<const unsigned int *>(dtab + 2 * e2);
            const unsigned int d2 = *reinterpret_cast<const unsigned int *>(dtab2 + 2 * e2);      // (both triangles of the symmetric ones)
            double x = (__longlong_as_double(bins[j][1].x + __double_as_longlong(m1)) - m1) +
                       (__longlong_as_double(bins[j][0].x + __double_as_longlong(m0)) - m0);
            double y = (__longlong_as_double(bins[j][1].y + __double_as_longlong(m1)) - m1) +
                       (__longlong_as_double(bins[j][0].y + __double_as_longlong(m0)) - m0);
            if (__builtin_expect(top != 0, 0)) {                                 // some |contribution| >= 2^9: the top bins as well
                for (int z = 0; z < p.shards; ++z) {
                    const long long *az = acc + (size_t)z * kPAccWords + 2 * kPEntries;
                    x += __longlong_as_double(load_sc1(az + 2 * e2) + __double_as_longlong(m2)) - m2;
                    y += __longlong_as_double(load_sc1(az + 2 * e2 + 1) + __double_as_longlong(m2)) - m2;
                }
            }
            Mp[d & 0xffffu] = x;
            Mp[d >> 16] = y;
            Mp[d2 & 0xffffu] = x;
            Mp[d2 >> 16] = y;
        }
    }
}

// ---- the update by ONE workgroup (the resolver, after a launch that could not run: persist_resolver) ---------------------
// The reference's sweep as it is written (dict_fact.py:672-694 with l2 atoms, enet.pyx:59-67): atom after atom in sweep order,
// candidate = (B_j - sum_{i != j} C[i][j] D_i) / C[j][j] against the dictionary AS IT IS NOW (frozen atoms keep their values,
// :681), scaled into the ball of the atom's budget, the budget updated.  Rows are read from and written to the real
// dictionary; sums in double.  Milliseconds instead of microseconds - it runs when the persistent launch cannot, once per
// plan (the host then keeps that plan on one launch per block, bcd.hip).
__device__ __forceinline__ void persist_recover(const BcdPersistArgs &p, char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;                 // six waves
    const int k = p.kout, kp = p.k;
    float *u = reinterpret_cast<float *>(smem_raw);                               // [s] candidates of the current atom
    double *red = reinterpret_cast<double *>(smem_raw + sizeof(float) * (size_t)((p.s + 3) & ~(int64_t)3));   // [3][8]
    for (int jj = 0; jj < k; ++jj) {
        const int oj = p.order[jj];
        const float d = p.cdiag[jj];
        const int fz = p.frozen[jj];
        const float budget_in = p.norm_in[jj];
        float cv[8];                                                             // row oj of C (= its column: C is symmetric), atoms lane + 64 q
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = lane + 64 * q;
            cv[q] = (i < k && i != oj) ? p.C[(int64_t)oj * k + i] : 0.f;
        }
        double old2 = 0.0, new2 = 0.0;
        for (int64_t f = wid; f < p.s; f += 6) {                                  // a row per wave
            const int64_t row = p.subset ? (int64_t)p.subset[f] : f;
            const float *dr = p.Dt_out + row * k;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = lane + 64 * q;
                if (i < k) acc += (double)cv[q] * (double)dr[i];
            }
#pragma unroll
            for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
            const float dold = dr[oj];
            const float bv = p.BsP[f * kp + jj];                         // (bcd_setup_kernel: plain rows, sweep order)
            const float un = fz ? dold : (float)(((double)bv - acc) / (double)d);
            if (lane == 0) u[f] = un;
            old2 += (double)dold * (double)dold;
            new2 += (double)un * (double)un;
        }
        if (lane == 0) { red[wid] = old2; red[8 + wid] = new2; }
        __syncthreads();
        double o2 = 0.0, n2 = 0.0;
#pragma unroll
        for (int w = 0; w < 6; ++w) { o2 += red[w]; n2 += red[8 + w]; }
        const double budget = (double)budget_in + o2;                             // dict_fact.py:677
        const double scale = (n2 <= budget) ? 1.0 : sqrt(n2 / budget);            // enet.pyx:60-66
        double out2 = 0.0;
        for (int64_t f = tid; f < p.s; f += 384) {
            const int64_t row = p.subset ? (int64_t)p.subset[f] : f;
            const float o = (budget > 0.0) ? (float)((double)u[f] / scale) : 0.f; // (enet.pyx:55: radius 0 -> the zero atom)
            p.Dt_out[row * k + oj] = o;
            out2 += (double)o * (double)o;
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) out2 += __shfl_xor(out2, off);
        if (lane == 0) red[16 + wid] = out2;
        __syncthreads();                                                         // (also: this atom's rows before the next atom reads them)
        if (tid == 0) {
            double t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 6; ++w) t2 += red[16 + w];
            p.norm_out[oj] = (float)(budget - t2);                                // dict_fact.py:692
        }
    }
    __syncthreads();
    if (tid == 0 && p.flags) __hip_atomic_fetch_add(p.flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the resolver workgroup ------------------------------------------------------------------------------------------
// Six wavefronts: 4 the chain of the recursion (it also publishes S while the others transform), 5 its helper, 0-3 everything
// else.  Per block b:   [transform: pieces of b + S_{b-1} -> Gram matrix of b | wave 4 publishes S_{b-1}]  barrier
//                       [recursion of b on waves 4, 5 | waves 0-3 wait for the row workgroups' pieces of block b + 1 and bring
//                        them into LDS]  barrier
// so that only the transform and the recursion are on the critical path.
__device__ __forceinline__ void persist_resolver(const BcdPersistArgs &p, char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double *Ms = reinterpret_cast<double *>(smem_raw);                         // [NB][64] Base rows (identity | Gram columns)
    double *D2s = Ms + kNB * 64;                                                // [2][NB] (double-buffered with Cs / CsT: block b uses b & 1)
    double *Cs = D2s + 2 * kNB;                                                 // [2][NB][NB]
    double *CAs = Cs + 2 * kNB * kNB;                                           // [NB][kCaStride] S of the block just resolved
    double *scr = CAs + kNB * kCaStride;                                        // [8][NB] the chain wave's scratch
    double *CsT = scr + 8 * kNB;                                                // [2][NB][NB]
    ResolveMail mail;
    mail.Pm = CsT + 2 * kNB * kNB;                                              // [kMbox][64]
    mail.Zm = mail.Pm + kMbox * 64;
    double *Qt = mail.Zm + kMbox * 64;                                          // [NB][kTS] x 6
    double *Mp = Qt + kNB * kTS, *Xs = Mp + kNB * kTS, *Ps = Xs + kNB * kTS, *Zs = Ps + kNB * kTS, *NNs = Zs + kNB * kTS;
    double *D2n = NNs + kNB * kTS;                                              // [NB] old squared norms of the NEXT block
    double *resB = D2n + kNB;                                                   // [2][NB] norm budgets of a block's atoms
    int *resJ = reinterpret_cast<int *>(resB + 2 * kNB);                        // [2][NB] ... and their indices
    int *flag = resJ + 2 * kNB;                                                 // [0] verdict of a wait, [1] pcount, [2] zcount, [3] pieces ready
    unsigned short *dtab = reinterpret_cast<unsigned short *>(flag + 20);       // [kPEntries] entry -> offset from Mp (piece_offset)
    unsigned short *dtab2 = dtab + kPEntries;                                   // ... and of its mirror image
    mail.pcount = flag + 1;
    mail.zcount = flag + 2;
    typedef __attribute__((address_space(3))) volatile int lds_vint;
#ifdef MODL_DIAG
    unsigned long long *st = p.stamps;
#else
    unsigned long long *const st = nullptr;
#endif
    if (st && tid == 0) st[0] = clock64();
    for (int e = tid; e < kPEntries; e += 384) {
        dtab[e] = (unsigned short)piece_offset(e, false);
        dtab2[e] = (unsigned short)piece_offset(e, true);
    }
    if (tid < 16) flag[4 + tid] = 0;                                             // meeting counters / verdict words (sync4, load_pieces)                                  // the transform's meeting counters (sync4; <N', N'> loaded)
    if (wid == 5) {
#pragma unroll
        for (int q = 0; q < kNB * 32 / 64; ++q) {                                // the identity half of the Base rows (never overwritten)
            const int e = lane + 64 * q, mm = e >> 5, xx = e & 31;
            Ms[mm * 64 + xx] = (mm == xx) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) flag[3] = 0;
    const int kp = p.k;
    // what waves 0-3 bring for block b: coefficients (registers -> LDS once the recursion before has finished with them),
    // budgets, Q, and - once every row workgroup has arrived - the accumulated pieces
    d2v cf[2];
    auto coef_request = [&](int b, int tv) {                                     // recursion coefficients of block b -> registers
        const int j0 = b * kNB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            cf[q] = *reinterpret_cast<const d2v *>(p.coef_all + (ok ? (int64_t)j0 * kNB + e : 0));
        }
    };
    d2v cg[2];
    auto coef_request2 = [&](int b, int tv) {                                    // (the other half when 128 threads do it: tv + 128)
        const int j0 = b * kNB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 128 + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            cg[q] = *reinterpret_cast<const d2v *>(p.coef_all + (ok ? (int64_t)j0 * kNB + e : 0));
        }
    };
    auto coef_store2 = [&](int b, int tv) {
        const int j0 = b * kNB;
        double *Cs_ = Cs + (b & 1) * kNB * kNB, *CsT_ = CsT + (b & 1) * kNB * kNB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 128 + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            const int jr = e / kNB, ic = e % kNB;
            Cs_[e] = ok ? cg[q].x : 0.0;
            Cs_[e + 1] = ok ? cg[q].y : 0.0;
            CsT_[ic * kNB + jr] = ok ? cg[q].x : 0.0;
            CsT_[(ic + 1) * kNB + jr] = ok ? cg[q].y : 0.0;
        }
    };
    auto coef_store = [&](int b, int tv) {                                       // -> buffer b & 1 (the recursion before reads the other)
        const int j0 = b * kNB;
        double *Cs_ = Cs + (b & 1) * kNB * kNB, *CsT_ = CsT + (b & 1) * kNB * kNB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            const int jr = e / kNB, ic = e % kNB;
            Cs_[e] = ok ? cf[q].x : 0.0;
            Cs_[e + 1] = ok ? cf[q].y : 0.0;
            CsT_[ic * kNB + jr] = ok ? cf[q].x : 0.0;
            CsT_[(ic + 1) * kNB + jr] = ok ? cf[q].y : 0.0;
        }
    };
    auto block_inputs = [&](int b, int tv) {                                     // Q against the block before, budgets -> LDS
        const int j0 = b * kNB;
        const int nb = (p.kout - j0 < kNB) ? p.kout - j0 : kNB;
        d2v qf[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            qf[q] = *reinterpret_cast<const d2v *>(p.qcoef + (ok ? (int64_t)j0 * kNB + e : 0));
        }
        int jj_raw = 0;
        float budget_raw = 0.f;
        if (tv < kNB) {
            jj_raw = p.order[j0 + ((tv < nb) ? tv : 0)];
            budget_raw = p.norm_in[(tv < nb) ? j0 + tv : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (tv + 256 * q);
            const bool ok = j0 + e / kNB < kp;
            const int jr = e / kNB, ic = e % kNB;
            Qt[jr * kTS + ic] = ok ? qf[q].x : 0.0;                              // Qt[c][i] = Q[i][c]
            Qt[jr * kTS + ic + 1] = ok ? qf[q].y : 0.0;
        }
        if (tv < kNB) {
            resJ[(b & 1) * kNB + tv] = (tv < nb) ? jj_raw : 0;
            resB[(b & 1) * kNB + tv] = (tv < nb) ? (double)budget_raw : 0.0;
        }
    };

    // ---- block 0: nothing to overlap with ----
    // (Measured and not kept: this prologue as a dry iteration of the loop below - waves 4, 5 running the recursion on whatever
    //  the LDS holds while the row workgroups prepare block 0, to have its 15 KB of straight-line code in the instruction cache:
    //  the first recursion 17.6 k -> 14.7 k cycles, but the generic two-stage load of block 0's pieces cost 3.6 k more than
    //  this one and every recursion 1 k more with the dry-run conditions in its arguments: 0.0970 -> 0.0988 ms.)
    {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        if (wid < 4) { coef_request(0, tv); block_inputs(0, tv); }
#ifdef MODL_DIAG
        if (tid == 0) flag[0] = poll_word(p.arrive + 0, (unsigned)p.expect + (p.inject == 3 ? 1u : 0u), p.err, p.inject == 3 ? (1u << 14) : kSpinLimit) ? 1 : 0;
#else
        if (tid == 0) flag[0] = poll_word(p.arrive + 0, (unsigned)p.expect, p.err) ? 1 : 0;
#endif
        __syncthreads();
        if (!flag[0]) {
            // a wait gave up before the first block was resolved (a row workgroup that is not resident: another process or a
            // mask holding compute units): no S has left this workgroup, so no row workgroup has applied anything - the
            // dictionary, the norm budgets and every input of the update are as the set-up kernel left them.  The row
            // workgroups see the error word and leave; this workgroup runs the whole sweep by itself.
            persist_recover(p, smem_raw);
            return;
        }
        if (st && tid == 0) st[1] = clock64();
        if (wid < 4) {
            load_pieces<256, kPairsXM, kPairsAll>(p, 0, tv, Mp, dtab, dtab2);    // (<N', N'> of block 0 IS its Gram matrix)
            coef_store(0, tv);
        }
        if (tid == 0) { *mail.pcount = 0; *mail.zcount = 0; }
        __syncthreads();
        for (int e = tv; e < kNB * kNB; e += 384) {                              // upper triangle -> the Base rows, both triangles
            const int i = e >> 5, j = e & 31;
            Ms[j * 64 + 32 + i] = NNs[i * kTS + j];
        }
        if (tv < kNB) D2s[tv] = D2n[tv];                                         // (buffer 0)
        __syncthreads();
        if (st && tid == 0) st[2] = clock64();
    }
    for (int b = 0; b < p.nblk; ++b) {
        const int j0 = b * kNB;
        const int nb = (p.kout - j0 < kNB) ? p.kout - j0 : kNB;
        // (the per-thread indices of this iteration are formed from an opaque copy of the thread index: hoisted out of the
        //  loop they would have to live across the recursion, whose helper wave needs the whole register file - the
        //  compiler then spills them to scratch, and the product library carries no scratch instruction)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        // ---- the recursion of block b (waves 4, 5) | the pieces of block b + 1 (waves 0 - 3) ----
        if (wid == 4) {
            const int x = tv & 31;
            __builtin_amdgcn_s_setprio(3);
            resolve_chain<float>(D2s + (b & 1) * kNB, Cs + (b & 1) * kNB * kNB, resJ[(b & 1) * kNB + x], resB[(b & 1) * kNB + x], nb,
                                 p.norm_out, scr, mail, nullptr);
            __builtin_amdgcn_s_setprio(0);
        } else if (wid == 5) {
            // S_b leaves for the row workgroups row by row as the recursion produces it, as 8-byte write-through stores over a
            // buffer of sentinels: the data is its own flag (guide, Guideline 16 R2) - nobody waits for a store to land
            resolve_helper(Ms, CsT + (b & 1) * kNB * kNB, CAs, kCaStride, mail, p.Sbuf + (size_t)b * kNB * kNB);
            if (st && (tv & 63) == 0) st[72 + b] = clock64();
        } else if (b + 1 < p.nblk) {
            block_inputs(b + 1, tv);
            // waves 0, 1 share their SIMDs with the chain and the helper: they go to the barrier (a waiting wave issues
            // nothing); waves 2, 3 - SIMDs of their own - wait for the row workgroups and bring the pieces in
            if (wid >= 2) {
                // (the next block's recursion coefficients into the other buffer, now: not in front of the transform)
                coef_request(b + 1, tv - 128);
                coef_request2(b + 1, tv - 128);
                if (tv == 128) {                                                // one lane waits for the row workgroups
#ifdef MODL_DIAG
                    const int ok = poll_word(p.arrive + b + 1, (unsigned)p.expect + (p.inject == 4 ? 1u : 0u), p.err, p.inject == 4 ? (1u << 14) : kSpinLimit) ? 1 : 2;
#else
                    const int ok = poll_word(p.arrive + b + 1, (unsigned)p.expect, p.err) ? 1 : 2;
#endif
                    if (st) st[5 + 4 * b] = clock64();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    *(lds_vint *)(flag + 3) = ok + 4 * (b + 1);                  // (monotonic: verdict + 4 x block)
                }
                int seen = *(lds_vint *)(flag + 3);
                while (__builtin_amdgcn_readfirstlane(seen) < 4 * (b + 1)) {
                    __builtin_amdgcn_s_sleep(8);
                    seen = *(lds_vint *)(flag + 3);
                }
                coef_store(b + 1, tv - 128);
                coef_store2(b + 1, tv - 128);
                if ((__builtin_amdgcn_readfirstlane(seen) & 3) == 1) {
                    load_pieces<128, 0, kPairsXM>(p, b + 1, tv - 128, Mp, dtab, dtab2);   // X, M' (<N', N'>: waves 4, 5, below)
                    if (st && tv == 128) st[6 + 4 * b] = clock64();
                }
            }
        }
        lds_barrier();                                                           // ---- recursion done, pieces of b + 1 in LDS
        if (st && tid == 0) st[3 + 4 * b] = clock64();
        if (b + 1 < p.nblk && (flag[3] & 3) != 1) {                              // (a wait gave up: every thread leaves)
            if (tid == 0) raise_incomplete(p);                                   // S_0 .. S_b have left: the update is incomplete
            return;
        }
        // ---- transform for block b + 1 ----
        if (b + 1 < p.nblk && wid < 4) {
            if (tv == 0) { *mail.pcount = 0; *mail.zcount = 0; }
            gram_ahead(Qt, CAs, Mp, Xs, Ps, Zs, NNs, Ms, wid, tv & 63, flag + 4, 8 * b, flag + 5, 2 * (b + 1),
                       (st && b == 2) ? st + 88 : nullptr);
        } else if (b + 1 < p.nblk) {
            // waves 4, 5 - idle in the transform - bring <N', N'> and the old norms in (only its last step reads them)
            load_pieces<128, kPairsXM, kPairsAll>(p, b + 1, tv - 256, Mp, dtab, dtab2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((tv & 63) == 0) __hip_atomic_fetch_add(flag + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (b + 1 < p.nblk) {
            lds_barrier();
            if (tv < kNB) D2s[((b + 1) & 1) * kNB + tv] = D2n[tv];
        }
        __syncthreads();                                                         // ---- Gram matrix of block b + 1 ready
        if (st && tid == 0) st[4 + 4 * b] = clock64();
    }
    // every S has been published: a row workgroup can only have given up before this point (off the update's critical path -
    // the row workgroups are still applying the last block)
    if (tid == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) raise_incomplete(p);
}

// ---- a row workgroup ---------------------------------------------------------------------------------------------------
template <int RT>
__device__ __forceinline__ void persist_rows(const BcdPersistArgs &p, char *smem_raw) {
    constexpr int RB = 32 * RT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;           // four waves
    const int ch = wid & 1;                                                    // the 16-column half of a block this wave owns
    const int rt0 = (wid >> 1) * RT;                                           // ... and its RT tiles of 16 rows
    // the lane-derived indices are re-formed from an opaque copy of the lane at the top of every phase (refresh): addresses
    // hoisted out of the phase loop and kept live across it cost more registers than the kernel has (scratch otherwise)
    int q = lane >> 4, m = lane & 15, col = 16 * ch + m;
    auto refresh = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        q = l >> 4; m = l & 15; col = 16 * ch + m;
    };
    const int kp = p.k, KQ = kp >> 2, KG = (kp + 15) >> 4;
    const int nblk = p.nblk;
    const int row_id = (int)blockIdx.x - 1;
    const int64_t f0 = (int64_t)row_id * RB;
    float *Dl = reinterpret_cast<float *>(smem_raw);                          // [RT][KQ][32][4] the rows, fragment order
    float *Tt = Dl + (size_t)RT * 32 * kp;                                      // three tiles [RB][kTF]: blocks b, b - 1, b - 2
    float *Dn = Tt + 3 * RB * kTF;                                              // [RB][kTF] the atoms the last applied block ended up with
    double *Ss = reinterpret_cast<double *>(Dn + RB * kTF);                    // [NB][kTS] S of the block being applied
    double *d2red = Ss + kNB * kTS;                                             // [8][NB]
    int *flag = reinterpret_cast<int *>(d2red + 8 * kNB);
#ifdef MODL_DIAG
    unsigned long long *st = (p.stamps && row_id == 0) ? p.stamps + 96 : nullptr;
#else
    unsigned long long *const st = nullptr;
#endif
    if (st && tid == 0) st[0] = clock64();
    if (tid == 0) flag[0] = 0;                                                   // raised by a wait that gave up (fetch_S)
    // destination rows of the applied values in the real dictionary (the f64 matrix-core output layout: row = q + 4 r)
    int subr[RT][4];
    {
        const int32_t *sub_src = p.subset ? p.subset : p.order;                 // (any readable words when there is no subset)
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = f0 + 16 * (rt0 + u) + q + 4 * r;
                subr[u][r] = sub_src[(p.subset && f < p.s) ? f : 0];
            }
    }

    // ---- pieces of the phase loop ----
    // the block's epilogue operands: B_ entries of this lane's outputs (f32 matrix-core output layout: row = 4 q + r)
    auto load_epi = [&](int jb, int nb, float (&Bv)[RT][4], float &cd, int &fz) {
        const int j0 = jb * kNB;
        const bool col_ok = col < nb;
        cd = p.cdiag[j0 + (col_ok ? col : 0)];
        fz = p.frozen[j0 + (col_ok ? col : 0)];
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t f = f0 + 16 * (rt0 + u) + 4 * q + r;
                f = f < p.s ? f : p.s - 1;
                Bv[u][r] = p.BsP[f * kp + j0 + (col_ok ? col : 0)];
            }
    };
    // acc += D . C[:, block jb] over the 16-atom steps outside [skip_lo, skip_hi), the rows from LDS, the coefficients from L2:
    // prod_load requests the coefficient fragments of 16 steps (gb ..), prod_mma consumes them
    auto prod_load = [&](int jb, int gb, f4v (&bf)[16]) {
        const float *cp0 = p.CPP + ((size_t)jb * KQ << 7) + (col << 2);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int grp = 4 * (gb + g) + q;                                    // this lane's four consecutive source atoms
            bf[g] = *reinterpret_cast<const f4v *>(cp0 + ((size_t)(grp < KQ ? grp : 0) << 7));
        }
    };
    auto prod_mma = [&](int nb, int gb, int skip_lo, int skip_hi, const f4v (&bf)[16], f4v (&acc)[RT]) {
        const bool cok = col < nb;
        // The row fragments of GH steps are read from LDS up front (one wait instead of one LDS round trip per step), and
        // the four products of a step go to FOUR accumulators: as one chain on one accumulator the 64 products of a block
        // each waited for their predecessor (40 cycles dependent against 32 issued) behind an LDS round trip per step -
        // 4.6 k cycles for 2 k of matrix-core time (ISA).
        constexpr int GH = RT == 1 ? 16 : 8;
        f4v c4[RT][4];
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int x = 0; x < 4; ++x) c4[u][x] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += GH) {
            f4v af[GH][RT];
#pragma unroll
            for (int g = 0; g < GH; ++g) {
                const int grp = 4 * (gb + g0 + g) + q;
                const int gcl = grp < KQ ? grp : 0;
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    const int rt = rt0 + u;
                    af[g][u] = *reinterpret_cast<const f4v *>(Dl + ((((rt >> 1) * KQ + gcl) << 5) + ((rt & 1) << 4) + m) * 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < GH; ++g) {
                // (a skipped step contributes zeros instead of being branched around: straight-line code)
                const int gs = gb + g0 + g;
                const int grp = 4 * gs + q;
                const bool ok = cok && grp < KQ && !(gs >= skip_lo && gs < skip_hi);
                f4v bb;
                bb.x = ok ? bf[g0 + g].x : 0.f; bb.y = ok ? bf[g0 + g].y : 0.f; bb.z = ok ? bf[g0 + g].z : 0.f; bb.w = ok ? bf[g0 + g].w : 0.f;
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    c4[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][u].x, bb.x, c4[u][0], 0, 0, 0);
                    c4[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][u].y, bb.y, c4[u][1], 0, 0, 0);
                    c4[u][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][u].z, bb.z, c4[u][2], 0, 0, 0);
                    c4[u][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][u].w, bb.w, c4[u][3], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] += (c4[u][0] + c4[u][1]) + (c4[u][2] + c4[u][3]);
    };
    // the whole product; bf: the fragments of the first 16 steps, already requested
    auto product = [&](int jb, int nb, int skip_lo, int skip_hi, f4v (&bf)[16], f4v (&acc)[RT]) {
        prod_mma(nb, 0, skip_lo, skip_hi, bf, acc);
        for (int gb = 16; gb < KG; gb += 16) {
            prod_load(jb, gb, bf);
            __builtin_amdgcn_sched_barrier(0);
            prod_mma(nb, gb, skip_lo, skip_hi, bf, acc);
        }
    };
    // candidates (with whatever was left out of `acc`) -> tile; the old squared norms of the block's columns -> d2red
    auto epilogue = [&](int jb, int nb, const f4v (&acc)[RT], const float (&Bv)[RT][4], float cd, int fz, float *Tdst) {
        const int j0 = jb * kNB;
        const bool col_ok = col < nb;
        const float cdm = col_ok ? cd : 1.f;
        const int fzm = col_ok ? fz : 0;
        double d2 = 0.0;
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int frow = 16 * (rt0 + u) + 4 * q + r;
                const bool ok = col_ok && f0 + frow < p.s;
                const float dold = Dl[dl_idx(frow, j0 + (col_ok ? col : 0), KQ)];
                float val = fzm ? dold : (Bv[u][r] - acc[u][r]) / cdm;
                val = ok ? val : 0.f;
                Tdst[frow * kTF + col] = val;
                d2 += ok ? (double)dold * (double)dold : 0.0;
            }
        d2red[((wid >> 1) * 4 + q) * kNB + col] = d2;
    };
    // Dnew = T S^T on the f64 matrix cores -> the LDS rows, the real dictionary, the Dn tile
    auto apply = [&](int jb, int nb, const float *Tsrc, int oc) {
        const int j0 = jb * kNB;
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int ft = rt0 + u;
            const float *ap = Tsrc + (16 * ft + m) * kTF + q;
            const double *sp = Ss + col * kTS + q;
            float fa[kNB / 4];
            double fs[kNB / 4];
#pragma unroll
            for (int kk = 0; kk < kNB / 4; ++kk) { fa[kk] = ap[4 * kk]; fs[kk] = sp[4 * kk]; }
            d4v dn = {0.0, 0.0, 0.0, 0.0}, dn1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < kNB / 4; kk += 2) {
                dn = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk], fs[kk], dn, 0, 0, 0);
                dn1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk + 1], fs[kk + 1], dn1, 0, 0, 0);
            }
            dn += dn1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int frow = 16 * ft + q + 4 * r;
                const int64_t f = f0 + frow;
                const bool live = f < p.s && col < nb;
                const float dnew = (float)dn[r];
                if (live) {
                    Dl[dl_idx(frow, j0 + col, KQ)] = dnew;
                    p.Dt_out[(p.subset ? (int64_t)subr[u][r] : f) * p.kout + oc] = dnew;
                }
                Dn[frow * kTF + col] = live ? dnew : 0.f;
            }
        }
    };
    // T[:, block jt] -= (Dn . C[block js, block jt]) / diag: the candidates of block jt with block js = jt - 1 put back in
    auto corr_load = [&](int jt, int js, f4v (&bf)[2]) {
        const float *cp0 = p.CPP + ((size_t)jt * KQ << 7) + (col << 2);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int grp = 4 * (2 * js + g2) + q;
            bf[g2] = *reinterpret_cast<const f4v *>(cp0 + ((size_t)(grp < KQ ? grp : 0) << 7));
        }
    };
    // acc += Dn . C[block js, target block] (rank 32; bf: the two coefficient fragments of corr_load, masked here)
    auto rank32 = [&](int nbt, int js, const f4v (&bf)[2], f4v (&acc)[RT]) {
        const bool cok = col < nbt;
        f4v c1[RT];
        f4v av[2][RT];
#pragma unroll
        for (int u = 0; u < RT; ++u) c1[u] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int u = 0; u < RT; ++u) av[g2][u] = *reinterpret_cast<const f4v *>(Dn + (16 * (rt0 + u) + m) * kTF + 16 * g2 + 4 * q);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int grp = 4 * (2 * js + g2) + q;
            const bool ok = cok && grp < KQ;
            f4v bb;
            bb.x = ok ? bf[g2].x : 0.f; bb.y = ok ? bf[g2].y : 0.f; bb.z = ok ? bf[g2].z : 0.f; bb.w = ok ? bf[g2].w : 0.f;
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const f4v a = av[g2][u];                                         // (two chains: a dependent product waits 40 cycles)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[u], 0, 0, 0);
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, c1[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[u], 0, 0, 0);
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, c1[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] += c1[u];
    };
    // T[:, target block] -= (Dn . C[block js, target block]) / diag: its candidates with block js put back in
    auto correct = [&](int nbt, int js, float cd, int fz, float *Ttile, const f4v (&bf)[2]) {
        f4v acc[RT];
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] = (f4v){0.f, 0.f, 0.f, 0.f};
        rank32(nbt, js, bf, acc);
        const bool upd = col < nbt && !fz;
        const float cdm = upd ? cd : 1.f;
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int frow = 16 * (rt0 + u) + 4 * q + r;
                if (upd && f0 + frow < p.s) Ttile[frow * kTF + col] -= acc[u][r] / cdm;
            }
    };
    // 16 x 16 tile (it, jt) of A^T B over the workgroup's rows, f64 matrix cores: out row = q + 4 r (column of A), col = m
    auto gram_tile = [&](const float *A, const float *B, int it, int jt) -> d4v {
        const float *ai = A + q * kTF + 16 * it + m;
        const float *bj = B + q * kTF + 16 * jt + m;
        float fa[RB / 4], fb[RB / 4];
#pragma unroll
        for (int kk = 0; kk < RB / 4; ++kk) { fa[kk] = ai[4 * kk * kTF]; fb[kk] = bj[4 * kk * kTF]; }
        d4v g = {0.0, 0.0, 0.0, 0.0}, g1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk += 2) {
            g = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk], (double)fb[kk], g, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[kk + 1], (double)fb[kk + 1], g1, 0, 0, 0);
        }
        return g + g1;
    };
    // (Measured and not kept: every workgroup walking its entries in an order of its own - rotated accumulator registers and
    //  tile order, so that the same address is not hit by all 30 at once: the pieces took 4.5 k cycles instead of 3.85 k and the
    //  drain in front of the arrival 5.8 k instead of 3.9 k.)
    auto emit_full = [&](long long *acc, double *rec, const d4v &g, int base, int ld, int it, int jt, bool &bad, bool &top) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = base + (16 * it + q + 4 * r) * ld + 16 * jt + m;
            pacc_add(acc, e, g[r], false, bad, top);
            store_sc1(rec + e, g[r]);
        }
    };
    auto emit_tri = [&](long long *acc, double *rec, const d4v &g, int base, bool &bad, bool &top) {      // diagonal tile: row <= col only
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q + 4 * r;
            if (row <= m) {
                const int e = base + tri_index(row, m);
                pacc_add(acc, e, g[r], row == m, bad, top);
                store_sc1(rec + e, g[r]);
            }
        }
    };
    // the pieces of block b: <N', N'> (Tn), X = <a', N'> and M' = <a', a'> (Ta: the candidates of block b - 1; block 0: none)
    auto pieces = [&](int b, const float *Tn, const float *Ta) {
        long long *acc = p.acc + ((size_t)b * p.shards + (row_id & (p.shards - 1))) * kPAccWords;
        double *rec = p.rec + ((size_t)(b & 1) * p.nrow + row_id) * kPEntries;
        bool bad = false, top = false;
        if (b == 0) {
            if (wid == 0) emit_tri(acc, rec, gram_tile(Tn, Tn, 0, 0), kPNN, bad, top);
            else if (wid == 1) emit_full(acc, rec, gram_tile(Tn, Tn, 0, 1), kPNN + kTri, 16, 0, 0, bad, top);
            else if (wid == 2) emit_tri(acc, rec, gram_tile(Tn, Tn, 1, 1), kPNN + kTri + 256, bad, top);
        } else if (wid < 2) {                                                    // X: rows of a', columns of N'
            emit_full(acc, rec, gram_tile(Ta, Tn, wid, 0), kPX, kNB, wid, 0, bad, top);
            emit_full(acc, rec, gram_tile(Ta, Tn, wid, 1), kPX, kNB, wid, 1, bad, top);
        } else {
            const float *Tq = (wid == 2) ? Tn : Ta;
            const int base = (wid == 2) ? kPNN : kPMp;
            emit_tri(acc, rec, gram_tile(Tq, Tq, 0, 0), base, bad, top);
            emit_full(acc, rec, gram_tile(Tq, Tq, 0, 1), base + kTri, 16, 0, 0, bad, top);
            emit_tri(acc, rec, gram_tile(Tq, Tq, 1, 1), base + kTri + 256, bad, top);
        }
        if (wid == 3 && lane < kNB) {                                            // + the old squared norms of block b's columns
            double t = 0.0;
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) t += d2red[gq * kNB + lane];
            pacc_add(acc, kPD2 + lane, t, true, bad, top);
            store_sc1(rec + kPD2 + lane, t);
        }
        if (bad) atomicOr(reinterpret_cast<unsigned long long *>(acc) + 3 * kPEntries, 1ull);
        if (top) atomicOr(reinterpret_cast<unsigned long long *>(acc) + 3 * kPEntries + 1, 1ull);
    };
    // S of block b -> LDS.  The resolver writes every entry with ONE 8-byte write-through store over a buffer the set-up kernel
    // filled with a sentinel (a NaN no recursion produces): each thread polls its own four entries until none is the
    // sentinel - the data is its own flag, one memory round trip after it lands.  (false: the wait gave up; every thread returns)
    auto fetch_S = [&](int b) -> bool {
        const int e = 4 * tid, i = e >> 5, x = e & 31;
        const double *src = p.Sbuf + (size_t)b * kNB * kNB + e;
        double s0, s1, s2, s3;
        bool ok = true;
        for (unsigned spins = 0;; ++spins) {
            s0 = load_sc1(src); s1 = load_sc1(src + 1); s2 = load_sc1(src + 2); s3 = load_sc1(src + 3);
            const bool have = __double_as_longlong(s0) != kSentinel && __double_as_longlong(s1) != kSentinel &&
                              __double_as_longlong(s2) != kSentinel && __double_as_longlong(s3) != kSentinel;
            if (__all(have)) break;
            if (spins > kSpinLimit || ((spins & 63) == 63 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                give_up(p.err);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) flag[0] = 1;                                                    // (set to 0 once, before the phases)
        *reinterpret_cast<d2v *>(Ss + i * kTS + x) = (d2v){s0, s1};
        *reinterpret_cast<d2v *>(Ss + i * kTS + x + 2) = (d2v){s2, s3};
        lds_barrier();
        return flag[0] == 0;
    };
    auto nb_of = [&](int jb) { return (p.kout - jb * kNB < kNB) ? p.kout - jb * kNB : kNB; };
    auto tile = [&](int jb) { return Tt + (jb % 3) * RB * kTF; };

    // ---- the phases.  Phase b (0 .. nblk + 1), while the resolver works on block b - 1:
    //   the part of block b's product that does not wait for anything (every atom outside blocks b - 2 and b - 1) | wait for
    //   S_{b-2} | apply block b - 2 | a_{b-1}: block b - 2 put back into the candidates of block b - 1 | N'_b: block b - 2's new
    //   atoms into block b's product (rank 32), epilogue | the pieces of block b -> accumulator, arrival.
    // ONE copy of the code for the first block, the middle and the tail: the persistent kernel is ~100 KB of instructions and
    // what a phase executes for the first time comes from memory.
    float Bv[RT][4], cd = 1.f, cd_prev = 1.f;
    int fz = 0, fz_prev = 0;
    f4v acc[RT], bf[16], cfr[2], cfr2[2];
    int oc = 0;
    float cd_next = 1.f;
    int fz_next = 0;
    // everything phase b reads from memory: requested at the END of phase b - 1, in front of the wait for that phase's atomics
    // (the requests then land while the atomics drain), phase 0's at the start
    auto request = [&](int b) {
        const bool has_prod = b < nblk, has_apply = b >= 2, has_prev = has_apply && b - 1 < nblk;
        if (has_prod) {
            load_epi(b, nb_of(b), Bv, cd_next, fz_next);
            prod_load(b, 0, bf);
        }
        if (has_apply) {
            oc = p.order[(b - 2) * kNB + ((col < nb_of(b - 2)) ? col : 0)];
            if (has_prev) corr_load(b - 1, b - 2, cfr);
            if (has_prod) corr_load(b, b - 2, cfr2);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // Requested at the top of its phase.  Measured and not kept (same box, dictionary update per minibatch): requested at the end
    // of the phase before, behind the pieces (the drain in front of the arrival then waits for them too: 0.0976 -> 0.0996 ms) or in
    // front of the pieces (their ~30 address computations sit on the chain S -> pieces -> arrival: 0.1024 ms).
    constexpr bool kAhead = false;
    if (kAhead) request(0);
    for (int b = 0; b <= nblk + 1; ++b) {
        refresh();
        const bool has_prod = b < nblk, has_apply = b >= 2, has_prev = has_apply && b - 1 < nblk;
        const int nb = has_prod ? nb_of(b) : 0;
        if (!kAhead) request(b);
        cd = cd_next; fz = fz_next;
        if (st && tid == 0 && b == 0) st[5] = clock64();
        if (b == 0) {
            // the rows -> LDS, once (rows beyond s: copies of the last one; whatever they produce is masked)
            {
                constexpr int NV = RT * 8;                                              // float4 elements per thread and 256 atoms
                for (int base = 0; base < RT * KQ * 32; base += NV * 256) {
                    float4 v[NV];
    #pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        int e = base + tid + 256 * u;
                        e = e < RT * KQ * 32 ? e : RT * KQ * 32 - 1;
                        const int r = e & 31, tg = e >> 5;
                        const int t32 = (RT > 1 && tg >= KQ) ? 1 : 0, g = tg - t32 * KQ;                    // (RT <= 2: no division)
                        int64_t f = f0 + 32 * t32 + r;
                        f = f < p.s ? f : p.s - 1;
                        v[u] = *reinterpret_cast<const float4 *>(p.DsP + dfrag(f, 4 * g, kp));
                    }
                    if (st && tid == 0) st[6] = clock64();
    #pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const int e = base + tid + 256 * u;
                        if (e < RT * KQ * 32) *reinterpret_cast<float4 *>(Dl + (size_t)e * 4) = v[u];
                    }
                }
            }
            __syncthreads();                                                     // (the rows are in LDS)
            if (st && tid == 0) st[1] = clock64();
        }
        if (has_prod) {
#pragma unroll
            for (int u = 0; u < RT; ++u) acc[u] = (f4v){0.f, 0.f, 0.f, 0.f};
            const int lo = 2 * (b - 2) > 0 ? 2 * (b - 2) : 0;
            product(b, nb, lo, 2 * b, bf, acc);                                  // (blocks b - 2, b - 1 left out)
        }
        if (st && tid == 0 && b >= 1 && b < 16) st[7 + 5 * (b - 1)] = clock64();
        if (has_apply) {
            if (!fetch_S(b - 2)) return;
            if (st && tid == 0 && b < 16) st[8 + 5 * (b - 1)] = clock64();
            apply(b - 2, nb_of(b - 2), tile(b - 2), oc);
            if (st && tid == 0 && b == 3) st[90] = clock64();
            lds_barrier();                                                       // (LDS only: the dictionary stores stay in flight)
            if (st && tid == 0 && b == 3) st[91] = clock64();
            if (has_prev) correct(nb_of(b - 1), b - 2, cd_prev, fz_prev, tile(b - 1), cfr);   // a_{b-1}: block b - 2 put back in
            if (st && tid == 0 && b == 3) st[92] = clock64();
            if (has_prod) rank32(nb, b - 2, cfr2, acc);                                        // N'_b: block b - 2's new atoms
        }
        if (st && tid == 0 && b >= 1 && b < 16) st[9 + 5 * (b - 1)] = clock64();
        if (has_prod) {
            epilogue(b, nb, acc, Bv, cd, fz, tile(b));
            lds_barrier();
            if (st && tid == 0) st[b == 0 ? 2 : 10 + 5 * (b - 1)] = clock64();
            cd_prev = cd; fz_prev = fz;
            if (kAhead) {
                // (in FRONT of the pieces: older than their atomics, the requests have landed long before the drain that
                //  precedes the arrival ends - behind them they would lengthen it)
                refresh();
                request(b + 1);                                                  // (Bv, bf, cfr, cfr2, oc: all consumed by now)
                refresh();
            }
            pieces(b, tile(b), b > 0 ? tile(b - 1) : nullptr);
            if (st && tid == 0) st[b == 0 ? 3 : 11 + 5 * (b - 1)] = clock64();
            signal_word(p.arrive + b, true);
            if (st && tid == 0 && b == 3) st[93] = clock64();
            if (st && tid == 0 && b == 0) st[4] = clock64();
        } else if (kAhead && b + 1 <= nblk + 1) {
            refresh();
            request(b + 1);
        }
    }
    if (st && tid == 0) st[89] = clock64();
}

}  // namespace

template <int RT>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(2, 2)))
void bcd_persist_kernel(BcdPersistArgs p, BcdRiderArgs rider) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((int)blockIdx.x > p.nrow) {                                              // riding tiles / the staging copy (rider.nslab = nrow + 1)
        bcd_rider_tile(rider, smem_raw);
        return;
    }
    if (blockIdx.x == 0) {
        persist_resolver(p, smem_raw);
        return;
    }
    if (threadIdx.x >= 256) return;                                              // a row workgroup works on four waves
    persist_rows<RT>(p, smem_raw);
}

size_t bcd_persist_lds(int kp, int RT) {
    const size_t RB = 32 * (size_t)RT;
    const size_t rows = 4 * (RB * kp + 4 * RB * kTF) + 8 * ((size_t)kNB * kTS + 8 * kNB) + 64;
    const size_t res = 8 * ((size_t)kNB * 64 + 2 * kNB + 4 * kNB * kNB + (size_t)kNB * kCaStride + 8 * kNB + 2 * kMbox * 64 + 6 * (size_t)kNB * kTS +
                            kNB + 2 * kNB) + 4 * (2 * kNB) + 80 + 4 * kPEntries + 2 * 96 + 16;
    return rows > res ? rows : res;
}

// Can the resolver + nrow row workgroups of the persistent launch be resident together on the CURRENT device?  Asked of the
// runtime for the kernel's actual register / LDS footprint (hipOccupancyMaxActiveBlocksPerMultiprocessor), per device and
// variant, once.  What it cannot know - another process or a mask holding compute units - is what persist_recover is for.
bool bcd_persist_fits(int kp, int RT, int nrow, size_t extra_lds) {
    size_t lds = bcd_persist_lds(kp, RT);
    if (extra_lds > lds) lds = extra_lds;
    if (lds > 160 * 1024 || nrow > kPersistRowsMax) return false;
    struct Slot { int dev; int rt; size_t lds; int cap; };
    static std::mutex mu;
    static std::vector<Slot> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    for (const Slot &c : cache)
        if (c.dev == dev && c.rt == RT && c.lds == lds) return 1 + nrow <= c.cap;
    void (*kern)(BcdPersistArgs, BcdRiderArgs) = (RT == 1) ? bcd_persist_kernel<1> : bcd_persist_kernel<2>;
    int per_cu = 0, ncu = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 384, lds) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
    }
    // one workgroup per compute unit is what the design counts on (the resolver alone on its unit; two row workgroups on one
    // unit would share its matrix pipe and LDS bandwidth and lengthen every phase): more than one resident per unit does not
    // raise the capacity
    const int cap = per_cu >= 1 ? ncu : 0;
    cache.push_back({dev, RT, lds, cap});
    return 1 + nrow <= cap;
}

int launch_bcd_persist(hipStream_t stream, const BcdPersistArgs &p, const BcdRiderArgs &rider, int extra_wgs, size_t extra_lds,
                       int RT) {
    void (*kern)(BcdPersistArgs, BcdRiderArgs) = (RT == 1) ? bcd_persist_kernel<1> : bcd_persist_kernel<2>;
    size_t lds = bcd_persist_lds(p.k, RT);
    if (extra_wgs > 0 && extra_lds > lds) lds = extra_lds;
    if (lds > 160 * 1024) return MODL_EINVAL;                                      // (bcd.hip asks bcd_persist_fits first)
    // (per call, like every other launch of the library: plans may live on several devices of one process)
    MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)(1 + p.nrow + extra_wgs)), dim3(384), lds, stream, p, rider);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

}  // namespace modl
