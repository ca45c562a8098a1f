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

namespace modl {

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef long long l2v __attribute__((ext_vector_type(2)));

constexpr int kTS = kNB + 2;                  // LDS row stride (doubles) of the 32 x 32 f64 matrices (= kCaStride)
constexpr int kTF = kNB + 4;                  // LDS row stride (floats) of the f32 tiles (16-byte aligned rows)
static_assert(kTS == kCaStride, "S is an operand of the Gram-domain transform");
constexpr unsigned kSpinLimit = 1u << 21;     // polls (each a memory round trip + s_sleep) before a wait gives up

__device__ __forceinline__ double i2d(long long b) {          // exact for |b| < 2^51 (a double -> int64 conversion and
    return __longlong_as_double(b + 0x4338000000000000ll) - 0x1.8p52;   // its inverse are software on this part)
}

// this workgroup's contribution to entry idx of a look-ahead accumulator (bcd_shared.hpp: acc_add, other strides)
__device__ __forceinline__ void pacc_add(long long *acc, int idx, double v, bool norm_entry, bool &bad) {
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
    if (__builtin_expect(b2 != 0, 0)) atomicAdd(a + 2 * kPEntries + idx, (unsigned long long)b2);
    atomicAdd(a + 1 * kPEntries + idx, (unsigned long long)b1);
    atomicAdd(a + idx, (unsigned long long)b0);
}

__device__ __forceinline__ int dl_idx(int frow, int c, int KQ) {        // element (row of the workgroup, sweep position) of the LDS-resident rows
    return (((frow >> 5) * KQ + (c >> 2)) << 7) + ((frow & 31) << 2) + (c & 3);
}

// ---- bounded waits --------------------------------------------------------------------------------------------------
// ONE lane polls ONE word (relaxed, agent scope), then one agent-scope acquire; the caller's barrier spreads the verdict.
__device__ __forceinline__ bool poll_word(unsigned int *word, unsigned int target, unsigned int *err) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) break;
        if (spins > kSpinLimit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
// every wave's stores drained, then ONE lane releases and signals
__device__ __forceinline__ void signal_word(unsigned int *word, bool add) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
#pragma unroll
    for (int kk = 0; kk < kNB / 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk], fb[kk], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void tile32_store(double *Cm, const d4v &c, int t1, int t2, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Cm[(t1 * 16 + (lane >> 4) + 4 * r) * kTS + t2 * 16 + (lane & 15)] = c[r];
}
// Called by all six wavefronts of the resolver (waves 0-3 compute one 16 x 16 tile each).  On entry: Base holds <N', N'>
// (Base[j * 64 + 32 + i] = element (i, j)), Qt[c][i] = Q[i][c], Ss = S of block b - 1, Mp = M', Xs = X (X[m][c] =
// <a_{b-1,m}, N'_c>); on exit Base holds the Gram matrix of block b's candidates (Qt and Ss are overwritten).
__device__ __forceinline__ void gram_ahead(double *Qt, double *Ss, const double *Mp, const double *Xs, double *Ps, double *Base,
                                           int wid, int lane) {
    const int t1 = wid >> 1, t2 = wid & 1;
    if (wid < 4) tile32_store(Ps, tile32<false>(Qt, Ss, t1, t2, lane), t1, t2, lane);            // P = Q^T S
    lds_barrier();
    if (wid < 4) {
        const d4v R = tile32<false>(Ps, Xs, t1, t2, lane);                                         // R = P X
        const d4v Z = tile32<true>(Mp, Ps, t1, t2, lane);                                          // Z = M' P^T
        tile32_store(Qt, R, t1, t2, lane);
        tile32_store(Ss, Z, t1, t2, lane);
    }
    lds_barrier();
    if (wid < 4) {
        const d4v V = tile32<false>(Ps, Ss, t1, t2, lane);                                         // V = P Z
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = t1 * 16 + (lane >> 4) + 4 * r, c2 = t2 * 16 + (lane & 15);
            Base[c2 * 64 + 32 + c] += (V[r] - Qt[c * kTS + c2]) - Qt[c2 * kTS + c];
        }
    }
}

// entry e of a block's accumulated pieces -> where the resolver wants it
struct SinkPieces {
    double *Xs, *Mp, *Base, *D2;
    const unsigned char *tri;            // [136] (row << 4) | col of a packed upper-triangle index
    __device__ __forceinline__ void packed(int qq, int &i, int &j) const {
        if (qq < kTri) {
            const int rc = tri[qq];
            i = rc >> 4; j = rc & 15;
        } else if (qq < kTri + 256) {
            i = (qq - kTri) >> 4; j = 16 + ((qq - kTri) & 15);
        } else {
            const int rc = tri[qq - kTri - 256];
            i = 16 + (rc >> 4); j = 16 + (rc & 15);
        }
    }
    __device__ __forceinline__ void operator()(int e, double v) const {
        int i, j;
        if (e < kPMp) Xs[(e >> 5) * kTS + (e & 31)] = v;
        else if (e < kPNN) {
            packed(e - kPMp, i, j);
            Mp[i * kTS + j] = v;
            Mp[j * kTS + i] = v;
        } else if (e < kPD2) {
            packed(e - kPNN, i, j);
            Base[j * 64 + 32 + i] = v;
            Base[i * 64 + 32 + j] = v;
        } else D2[e - kPD2] = v;
    }
};

// ---- the resolver workgroup ------------------------------------------------------------------------------------------
__device__ __forceinline__ void persist_resolver(const BcdPersistArgs &p, char *smem_raw) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double *Ms = reinterpret_cast<double *>(smem_raw);                         // [NB][64] Base rows (identity | Gram columns)
    double *D2s = Ms + kNB * 64;                                                // [NB]
    double *Cs = D2s + kNB;                                                     // [NB][NB]
    double *CAs = Cs + kNB * kNB;                                               // [NB][kCaStride] S of the block just resolved
    double *scr = CAs + kNB * kCaStride;                                        // [8][NB] the chain wave's scratch
    double *CsT = scr + 8 * kNB;                                                // [NB][NB]
    ResolveMail mail;
    mail.Pm = CsT + kNB * kNB;                                                  // [kMbox][64]
    mail.Zm = mail.Pm + kMbox * 64;
    double *Qt = mail.Zm + kMbox * 64;                                          // [NB][kTS] x 4
    double *Mp = Qt + kNB * kTS, *Xs = Mp + kNB * kTS, *Ps = Xs + kNB * kTS;
    int *flag = reinterpret_cast<int *>(Ps + kNB * kTS);                        // [0] verdict of a wait, [1] pcount, [2] zcount
    unsigned char *tri = reinterpret_cast<unsigned char *>(flag + 4);          // [136]
    mail.pcount = flag + 1;
    mail.zcount = flag + 2;
#ifdef MODL_DIAG
    unsigned long long *st = p.stamps;
#else
    unsigned long long *const st = nullptr;
#endif
    if (st && tid == 0) st[0] = clock64();
    if (tid < kTri) {                                                            // packed index -> (row, col), row <= col
        int row = 0;
        for (int r = 1; r < 16; ++r)
            if (tid >= r * 16 - r * (r - 1) / 2) row = r;
        tri[tid] = (unsigned char)((row << 4) | (row + (tid - (row * 16 - row * (row - 1) / 2))));
    }
    if (wid == 5) {
#pragma unroll
        for (int q = 0; q < kNB * 32 / 64; ++q) {                                // the identity half of the Base rows (never overwritten)
            const int e = lane + 64 * q, mm = e >> 5, xx = e & 31;
            Ms[mm * 64 + xx] = (mm == xx) ? 1.0 : 0.0;
        }
    }
    const int kp = p.k;
    const SinkPieces sink{Xs, Mp, Ms, D2s, tri};
    for (int b = 0; b < p.nblk; ++b) {
        const int j0 = b * kNB;
        const int nb = (p.kout - j0 < kNB) ? p.kout - j0 : kNB;
        // (1) what does not depend on the row workgroups: the recursion's coefficients, Q against the block before, the budgets
        d2v cf[2], qf[2];
        int res_jj = 0;
        double res_budget = 0.0;
        if (wid < 4) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * (tid + 256 * q);
                const bool ok = j0 + e / kNB < kp;
                cf[q] = *reinterpret_cast<const d2v *>(p.coef_all + (ok ? (int64_t)j0 * kNB + e : 0));
                qf[q] = *reinterpret_cast<const d2v *>(p.qcoef + (ok ? (int64_t)j0 * kNB + e : 0));
            }
        } else if (wid == 4) {
            const int x = lane & 31;
            const int jj_raw = p.order[j0 + ((x < nb) ? x : 0)];
            const float budget_raw = p.norm_in[(x < nb) ? j0 + x : 0];
            res_jj = (x < nb) ? jj_raw : 0;
            res_budget = (x < nb) ? (double)budget_raw : 0.0;
        }
        // (2) every row workgroup has added its pieces of block b
        if (tid == 0) flag[0] = poll_word(p.arrive + b, (unsigned)p.nrow, p.err) ? 1 : 0;
        if (wid < 4) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * (tid + 256 * q);
                const bool ok = j0 + e / kNB < kp;
                const int jr = e / kNB, ic = e % kNB;                            // coefficient of target jr against in-block / previous-block atom ic
                Cs[e] = ok ? cf[q].x : 0.0;
                Cs[e + 1] = ok ? cf[q].y : 0.0;
                CsT[ic * kNB + jr] = ok ? cf[q].x : 0.0;
                CsT[(ic + 1) * kNB + jr] = ok ? cf[q].y : 0.0;
                Qt[jr * kTS + ic] = ok ? qf[q].x : 0.0;                          // Qt[c][i] = Q[i][c]
                Qt[jr * kTS + ic + 1] = ok ? qf[q].y : 0.0;
            }
        }
        __syncthreads();
        if (!flag[0]) return;
        if (st && tid == 0) st[1 + 5 * b] = clock64();
        // (3) the accumulated pieces -> LDS (1056 pairs of entries on 256 threads; integer bins summed over the shards)
        if (wid < 4) {
            constexpr int NP = kPEntries / 2, NJ = (NP + 255) / 256;
            const long long *acc = p.acc + (size_t)b * p.shards * kPAccWords;
            l2v bins[NJ][3];
            long long bad = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int e2 = tid + 256 * j;
                const l2v *base = reinterpret_cast<const l2v *>(acc) + (e2 < NP ? e2 : 0);
                bins[j][0] = base[0]; bins[j][1] = base[NP]; bins[j][2] = base[2 * NP];
            }
            bad |= acc[3 * kPEntries];
            for (int z = 1; z < p.shards; ++z) {
                const long long *az = acc + (size_t)z * kPAccWords;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int e2 = tid + 256 * j;
                    const l2v *base = reinterpret_cast<const l2v *>(az) + (e2 < NP ? e2 : 0);
                    bins[j][0] += base[0]; bins[j][1] += base[NP]; bins[j][2] += base[2 * NP];
                }
                bad |= az[3 * kPEntries];
            }
            if (__builtin_expect(bad == 0, 1)) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int e2 = tid + 256 * j;
                    if (e2 < NP) {
                        sink(2 * e2, (i2d(bins[j][2].x) * 0x1p10 + i2d(bins[j][1].x) * 0x1p-30) + i2d(bins[j][0].x) * 0x1p-70);
                        sink(2 * e2 + 1, (i2d(bins[j][2].y) * 0x1p10 + i2d(bins[j][1].y) * 0x1p-30) + i2d(bins[j][0].y) * 0x1p-70);
                    }
                }
            } else {
                // a contribution outside the accumulator's range: the per-workgroup records, summed in workgroup order
                const double *rec = p.rec + (size_t)(b & 1) * p.nrow * kPEntries;
                for (int e = tid; e < kPEntries; e += 256) {
                    double t = 0.0;
                    for (int z = 0; z < p.nrow; ++z) t += rec[(size_t)z * kPEntries + e];
                    sink(e, t);
                }
            }
        }
        __syncthreads();
        if (st && tid == 0) st[2 + 5 * b] = clock64();
        // (4) pieces -> the Gram matrix of block b's candidates (block 0: <N', N'> is it)
        if (b > 0) gram_ahead(Qt, CAs, Mp, Xs, Ps, Ms, wid, lane);
        if (tid == 0) { *mail.pcount = 0; *mail.zcount = 0; }
        __syncthreads();
        if (st && tid == 0) st[3 + 5 * b] = clock64();
        // (5) the alpha recursion of block b: S_b -> CAs, the new norm budgets -> comp_norm
        if (wid == 4) {
            __builtin_amdgcn_s_setprio(3);
            resolve_chain<float>(D2s, Cs, res_jj, res_budget, nb, p.norm_out, scr, mail, nullptr);
            __builtin_amdgcn_s_setprio(0);
        } else if (wid == 5) {
            resolve_helper(Ms, CsT, CAs, kCaStride, mail);
        }
        __syncthreads();
        if (st && tid == 0) st[4 + 5 * b] = clock64();
        // (6) publish S_b
        if (wid < 4) {
            const int e = 4 * tid, i = e >> 5, x = e & 31;
            const d2v v0 = *reinterpret_cast<const d2v *>(CAs + i * kCaStride + x);
            const d2v v1 = *reinterpret_cast<const d2v *>(CAs + i * kCaStride + x + 2);
            d2v *dst = reinterpret_cast<d2v *>(p.Sbuf + (size_t)b * kNB * kNB + e);
            dst[0] = v0; dst[1] = v1;
        }
        signal_word(p.sflag + b, false);
        if (st && tid == 0) st[5 + 5 * b] = clock64();
    }
}

// ---- a row workgroup ---------------------------------------------------------------------------------------------------
template <int RT>
__device__ __forceinline__ void persist_rows(const BcdPersistArgs &p, char *smem_raw) {
    constexpr int RB = 32 * RT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;           // four waves
    const int q = lane >> 4, m = lane & 15;
    const int ch = wid & 1;                                                    // the 16-column half of a block this wave owns
    const int rt0 = (wid >> 1) * RT;                                           // ... and its RT tiles of 16 rows
    const int col = 16 * ch + m;
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
    // the rows -> LDS, once (rows beyond s: copies of the last one; whatever they produce is masked)
    {
        constexpr int NV = RT * 8;                                              // float4 elements per thread and 256 atoms
        for (int base = 0; base < RT * KQ * 32; base += NV * 256) {
            float4 v[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                int e = base + tid + 256 * u;
                e = e < RT * KQ * 32 ? e : RT * KQ * 32 - 1;
                const int r = e & 31, g = (e >> 5) % KQ, t32 = (e >> 5) / KQ;
                int64_t f = f0 + 32 * t32 + r;
                f = f < p.s ? f : p.s - 1;
                v[u] = *reinterpret_cast<const float4 *>(p.DsP + dfrag(f, 4 * g, kp));
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = base + tid + 256 * u;
                if (e < RT * KQ * 32) *reinterpret_cast<float4 *>(Dl + (size_t)e * 4) = v[u];
            }
        }
    }
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
    // acc += D . C[:, block jb] over the 16-atom steps outside [skip_lo, skip_hi), the rows from LDS, the coefficients from L2
    auto product = [&](int jb, int nb, int skip_lo, int skip_hi, f4v (&acc)[RT]) {
        const float *cp0 = p.CPP + ((size_t)jb * KQ << 7) + (col << 2);
        const bool cok = col < nb;
        for (int gb = 0; gb < KG; gb += 16) {
            f4v bf[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int grp = 4 * (gb + g) + q;                                // this lane's four consecutive source atoms
                bf[g] = *reinterpret_cast<const f4v *>(cp0 + ((size_t)(grp < KQ ? grp : 0) << 7));
            }
            __builtin_amdgcn_sched_barrier(0);                                   // (every request before the first mask / product)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int gs = gb + g;
                if (gs >= KG || (gs >= skip_lo && gs < skip_hi)) continue;       // (wave-uniform)
                const int grp = 4 * gs + q;
                const bool ok = cok && grp < KQ;
                f4v bb;
                bb.x = ok ? bf[g].x : 0.f; bb.y = ok ? bf[g].y : 0.f; bb.z = ok ? bf[g].z : 0.f; bb.w = ok ? bf[g].w : 0.f;
                const int gcl = grp < KQ ? grp : 0;
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    const int rt = rt0 + u;
                    const f4v a = *reinterpret_cast<const f4v *>(Dl + ((((rt >> 1) * KQ + gcl) << 5) + ((rt & 1) << 4) + m) * 4);
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc[u], 0, 0, 0);
                }
            }
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
    auto apply = [&](int jb, int nb, const float *Tsrc) {
        const int j0 = jb * kNB;
        const int oc = p.order[j0 + ((col < nb) ? col : 0)];
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
    auto correct = [&](int jt, int nbt, int js, float cd, int fz, float *Ttile) {
        const float *cp0 = p.CPP + ((size_t)jt * KQ << 7) + (col << 2);
        const bool cok = col < nbt;
        f4v bf[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int grp = 4 * (2 * js + g2) + q;
            bf[g2] = *reinterpret_cast<const f4v *>(cp0 + ((size_t)(grp < KQ ? grp : 0) << 7));
        }
        __builtin_amdgcn_sched_barrier(0);
        f4v acc[RT];
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int grp = 4 * (2 * js + g2) + q;
            const bool ok = cok && grp < KQ;
            f4v bb;
            bb.x = ok ? bf[g2].x : 0.f; bb.y = ok ? bf[g2].y : 0.f; bb.z = ok ? bf[g2].z : 0.f; bb.w = ok ? bf[g2].w : 0.f;
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const f4v a = *reinterpret_cast<const f4v *>(Dn + (16 * (rt0 + u) + m) * kTF + 16 * g2 + 4 * q);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, acc[u], 0, 0, 0);
            }
        }
        const bool upd = cok && !fz;
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
    auto emit_full = [&](long long *acc, double *rec, const d4v &g, int base, int ld, int it, int jt, bool &bad) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = base + (16 * it + q + 4 * r) * ld + 16 * jt + m;
            pacc_add(acc, e, g[r], false, bad);
            rec[e] = g[r];
        }
    };
    auto emit_tri = [&](long long *acc, double *rec, const d4v &g, int base, bool &bad) {      // diagonal tile: row <= col only
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q + 4 * r;
            if (row <= m) {
                const int e = base + tri_index(row, m);
                pacc_add(acc, e, g[r], row == m, bad);
                rec[e] = g[r];
            }
        }
    };
    // the pieces of block b: <N', N'> (Tn), X = <a', N'> and M' = <a', a'> (Ta: the candidates of block b - 1; block 0: none)
    auto pieces = [&](int b, const float *Tn, const float *Ta) {
        long long *acc = p.acc + ((size_t)b * p.shards + (row_id & (p.shards - 1))) * kPAccWords;
        double *rec = p.rec + ((size_t)(b & 1) * p.nrow + row_id) * kPEntries;
        bool bad = false;
        if (b == 0) {
            if (wid == 0) emit_tri(acc, rec, gram_tile(Tn, Tn, 0, 0), kPNN, bad);
            else if (wid == 1) emit_full(acc, rec, gram_tile(Tn, Tn, 0, 1), kPNN + kTri, 16, 0, 0, bad);
            else if (wid == 2) emit_tri(acc, rec, gram_tile(Tn, Tn, 1, 1), kPNN + kTri + 256, bad);
        } else if (wid < 2) {                                                    // X: rows of a', columns of N'
            emit_full(acc, rec, gram_tile(Ta, Tn, wid, 0), kPX, kNB, wid, 0, bad);
            emit_full(acc, rec, gram_tile(Ta, Tn, wid, 1), kPX, kNB, wid, 1, bad);
        } else {
            const float *Tq = (wid == 2) ? Tn : Ta;
            const int base = (wid == 2) ? kPNN : kPMp;
            emit_tri(acc, rec, gram_tile(Tq, Tq, 0, 0), base, bad);
            emit_full(acc, rec, gram_tile(Tq, Tq, 0, 1), base + kTri, 16, 0, 0, bad);
            emit_tri(acc, rec, gram_tile(Tq, Tq, 1, 1), base + kTri + 256, bad);
        }
        if (wid == 3 && lane < kNB) {                                            // + the old squared norms of block b's columns
            double t = 0.0;
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) t += d2red[gq * kNB + lane];
            pacc_add(acc, kPD2 + lane, t, true, bad);
            rec[kPD2 + lane] = t;
        }
        if (bad) atomicOr(reinterpret_cast<unsigned long long *>(acc) + 3 * kPEntries, 1ull);
    };
    // S of block b -> LDS (false: the wait gave up; every thread returns)
    auto fetch_S = [&](int b) -> bool {
        if (tid == 0) flag[0] = poll_word(p.sflag + b, 1u, p.err) ? 1 : 0;
        __syncthreads();
        if (!flag[0]) return false;
        const int e = 4 * tid, i = e >> 5, x = e & 31;
        const d2v *src = reinterpret_cast<const d2v *>(p.Sbuf + (size_t)b * kNB * kNB + e);
        const d2v v0 = src[0], v1 = src[1];
        *reinterpret_cast<d2v *>(Ss + i * kTS + x) = v0;
        *reinterpret_cast<d2v *>(Ss + i * kTS + x + 2) = v1;
        __syncthreads();
        return true;
    };
    auto nb_of = [&](int jb) { return (p.kout - jb * kNB < kNB) ? p.kout - jb * kNB : kNB; };
    auto tile = [&](int jb) { return Tt + (jb % 3) * RB * kTF; };

    __syncthreads();                                                             // (the rows are in LDS)
    // ---- block 0: its candidates are the full product ----
    float Bv[RT][4], cd = 1.f, cd_prev = 1.f;
    int fz = 0, fz_prev = 0;
    f4v acc[RT];
    {
        load_epi(0, nb_of(0), Bv, cd, fz);
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] = (f4v){0.f, 0.f, 0.f, 0.f};
        product(0, nb_of(0), 0, 0, acc);
        epilogue(0, nb_of(0), acc, Bv, cd, fz, tile(0));
        __syncthreads();
        pieces(0, tile(0), nullptr);
        signal_word(p.arrive + 0, true);
        cd_prev = cd; fz_prev = fz;
        if (st && tid == 0) st[1] = clock64();
    }
    // ---- block b, while the resolver runs the recursion of block b - 1 ----
    for (int b = 1; b < nblk; ++b) {
        const int nb = nb_of(b);
        load_epi(b, nb, Bv, cd, fz);
        if (b >= 2) {
            if (!fetch_S(b - 2)) return;
            if (st && tid == 0) st[2 + 4 * b] = clock64();
            apply(b - 2, nb_of(b - 2), tile(b - 2));
            __syncthreads();
            correct(b - 1, nb_of(b - 1), b - 2, cd_prev, fz_prev, tile(b - 1));   // a_{b-1}: block b - 2 put back in
        }
        if (st && tid == 0) st[3 + 4 * b] = clock64();
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[u] = (f4v){0.f, 0.f, 0.f, 0.f};
        product(b, nb, 2 * (b - 1), 2 * b, acc);                               // N'_b: block b - 1 left out
        epilogue(b, nb, acc, Bv, cd, fz, tile(b));
        __syncthreads();
        if (st && tid == 0) st[4 + 4 * b] = clock64();
        pieces(b, tile(b), tile(b - 1));
        signal_word(p.arrive + b, true);
        cd_prev = cd; fz_prev = fz;
        if (st && tid == 0) st[5 + 4 * b] = clock64();
    }
    // ---- the last two blocks' atoms ----
    if (nblk >= 2) {
        if (!fetch_S(nblk - 2)) return;
        apply(nblk - 2, nb_of(nblk - 2), tile(nblk - 2));
        __syncthreads();
        correct(nblk - 1, nb_of(nblk - 1), nblk - 2, cd_prev, fz_prev, tile(nblk - 1));
    }
    if (!fetch_S(nblk - 1)) return;                                             // (its barriers order the correction before the apply)
    apply(nblk - 1, nb_of(nblk - 1), tile(nblk - 1));
    if (st && tid == 0) st[6 + 4 * nblk] = clock64();
}

}  // namespace

template <int RT>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(2, 2)))
void bcd_persist_kernel(BcdPersistArgs p, BcdRiderArgs rider) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((int)blockIdx.x > p.nrow) {                                              // riding tiles / the staging copy (rider.nslab = nrow + 1)
        if (MODL_ROLE & 4) bcd_rider_tile(rider, smem_raw);
        return;
    }
    if (blockIdx.x == 0) {
        if (MODL_ROLE & 1) persist_resolver(p, smem_raw);
        return;
    }
    if (threadIdx.x >= 256) return;                                              // a row workgroup works on four waves
    if (MODL_ROLE & 2) persist_rows<RT>(p, smem_raw);
}

size_t bcd_persist_lds(int kp, int RT) {
    const size_t RB = 32 * (size_t)RT;
    const size_t rows = 4 * (RB * kp + 4 * RB * kTF) + 8 * ((size_t)kNB * kTS + 8 * kNB) + 64;
    const size_t res = 8 * ((size_t)kNB * 64 + kNB + 2 * kNB * kNB + (size_t)kNB * kCaStride + 8 * kNB + 2 * kMbox * 64 + 4 * (size_t)kNB * kTS) + 16 + 144;
    return rows > res ? rows : res;
}

int launch_bcd_persist(hipStream_t stream, const BcdPersistArgs &p, const BcdRiderArgs &rider, int extra_wgs, size_t extra_lds,
                       int RT) {
    void (*kern)(BcdPersistArgs, BcdRiderArgs) = (RT == 1) ? bcd_persist_kernel<1> : bcd_persist_kernel<2>;
    size_t lds = bcd_persist_lds(p.k, RT);
    if (extra_wgs > 0 && extra_lds > lds) lds = extra_lds;
    if (lds > 160 * 1024) return MODL_EINVAL;
    static bool attr_set[2] = {false, false};
    if (!attr_set[RT - 1]) {
        MODL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[RT - 1] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(1 + p.nrow + extra_wgs)), dim3(384), lds, stream, p, rider);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

}  // namespace modl
