/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the integer / RNG part of MODL's hot path:
 *   - Mersenne-Twister stream, masked-rejection interval draw, 53-bit double
 *       (reference: modl/utils/randomkit/randomkit.c:138-152, 212-241, 260-297)
 *   - binomial by inversion / BTPE with cached set-up
 *       (reference: modl/utils/randomkit/distributions.c:262-470)
 *   - "draw all swaps first, then apply" Fisher-Yates shuffle, permutation,
 *     shuffle-with-trace
 *       (reference: modl/utils/randomkit/random_fast.pyx:79-144)
 *   - the feature Sampler with its 4 modes
 *       (reference: modl/utils/randomkit/sampler.pyx:10-70)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  Pinned by: tests/golden/rng_*.npz (generated from the real
 * reference by tests/golden/make_golden.py) and, when oracle/_ref/librk_ref.so
 * exists, by direct comparison with the reference's own randomkit.c build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORK_N 624
#define ORK_M 397

typedef struct ork_state {
    uint32_t mt[ORK_N];
    int pos;
    /* binomial set-up cache (distributions.c keeps it inside rk_state,
       randomkit.h:77-96) */
    int has_binomial;
    double psave;
    long nsave;
    double r, q, fm, p1, xm, xl, xr, c, laml, lamr, p2, p3, p4;
    long m;
} ork_state;

/* randomkit.c:138-152 : only the low 32 bits of the seed are used */
void ork_seed(ork_state *st, uint64_t seed)
{
    uint32_t s = (uint32_t)(seed & 0xffffffffu);
    for (int i = 0; i < ORK_N; i++) {
        st->mt[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    st->pos = ORK_N;
    st->has_binomial = 0;
}

/* randomkit.c:212-241 */
uint32_t ork_random(ork_state *st)
{
    if (st->pos == ORK_N) {
        uint32_t *mt = st->mt;
        int i;
        for (i = 0; i < ORK_N; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % ORK_N] & 0x7fffffffu);
            uint32_t v = mt[(i + ORK_M) % ORK_N] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[i] = v;
        }
        st->pos = 0;
    }
    uint32_t y = st->mt[st->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* randomkit.c:260-289 (64-bit unsigned long build) */
uint64_t ork_interval(ork_state *st, uint64_t max)
{
    if (max == 0) return 0; /* consumes nothing */
    uint64_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint64_t v;
    if (max <= 0xffffffffull) {
        do { v = (uint64_t)ork_random(st) & mask; } while (v > max);
    } else {
        do {
            uint64_t hi = ork_random(st);
            uint64_t lo = ork_random(st);
            v = ((hi << 32) | lo) & mask;
        } while (v > max);
    }
    return v;
}

/* randomkit.c:292-297 */
double ork_double(ork_state *st)
{
    long a = (long)(ork_random(st) >> 5);
    long b = (long)(ork_random(st) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* distributions.c:399-440 */
static long ork_binomial_inversion(ork_state *st, long n, double p)
{
    double q, qn, np, px, U;
    long X, bound;
    if (!st->has_binomial || st->nsave != n || st->psave != p) {
        st->nsave = n; st->psave = p; st->has_binomial = 1;
        st->q = q = 1.0 - p;
        st->r = qn = exp(n * log(q));
        st->c = np = n * p;
        double bb = np + 10.0 * sqrt(np * q + 1);
        st->m = bound = (long)((double)n < bb ? (double)n : bb);
    } else {
        q = st->q; qn = st->r; np = st->c; bound = st->m;
    }
    (void)np;
    X = 0; px = qn; U = ork_double(st);
    while (U > px) {
        X++;
        if (X > bound) {
            X = 0; px = qn; U = ork_double(st);
        } else {
            U -= px;
            px = ((n - X + 1) * p * px) / (X * q);
        }
    }
    return X;
}

/* distributions.c:262-397 (BTPE, Kachitvichyanukul & Schmeiser) */
static long ork_binomial_btpe(ork_state *st, long n, double p)
{
    double r, q, fm, p1, xm, xl, xr, c, laml, lamr, p2, p3, p4;
    double a, u, v, s, F, rho, t, A, nrq, x1, x2, f1, f2, z, z2, w, w2, x;
    long m, y, k, i;

    if (!st->has_binomial || st->nsave != n || st->psave != p) {
        st->nsave = n; st->psave = p; st->has_binomial = 1;
        st->r = r = (p < 1.0 - p) ? p : 1.0 - p;
        st->q = q = 1.0 - r;
        st->fm = fm = n * r + r;
        st->m = m = (long)floor(fm);
        st->p1 = p1 = floor(2.195 * sqrt(n * r * q) - 4.6 * q) + 0.5;
        st->xm = xm = m + 0.5;
        st->xl = xl = xm - p1;
        st->xr = xr = xm + p1;
        st->c = c = 0.134 + 20.5 / (15.3 + m);
        a = (fm - xl) / (fm - xl * r);
        st->laml = laml = a * (1.0 + a / 2.0);
        a = (xr - fm) / (xr * q);
        st->lamr = lamr = a * (1.0 + a / 2.0);
        st->p2 = p2 = p1 * (1.0 + 2.0 * c);
        st->p3 = p3 = p2 + c / laml;
        st->p4 = p4 = p3 + c / lamr;
    } else {
        r = st->r; q = st->q; fm = st->fm; m = st->m; p1 = st->p1; xm = st->xm;
        xl = st->xl; xr = st->xr; c = st->c; laml = st->laml; lamr = st->lamr;
        p2 = st->p2; p3 = st->p3; p4 = st->p4;
    }
    (void)fm;

    for (;;) { /* Step10 */
        int accept_direct = 0;
        nrq = n * r * q;
        u = ork_double(st) * p4;
        v = ork_double(st);
        if (u <= p1) {
            y = (long)floor(xm - p1 * v + u);
            accept_direct = 1;
        } else if (u <= p2) { /* Step20 */
            x = xl + (u - p1) / c;
            v = v * c + 1.0 - fabs(m - x + 0.5) / p1;
            if (v > 1.0) continue;
            y = (long)floor(x);
        } else if (u <= p3) { /* Step30 */
            y = (long)floor(xl + log(v) / laml);
            if (y < 0) continue;
            v = v * (u - p2) * laml;
        } else { /* Step40 */
            y = (long)floor(xr - log(v) / lamr);
            if (y > n) continue;
            v = v * (u - p3) * lamr;
        }
        if (!accept_direct) { /* Step50 */
            k = labs(y - m);
            if ((k > 20) && (k < ((nrq) / 2.0 - 1))) { /* Step52 */
                rho = (k / (nrq)) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
                t = -k * k / (2 * nrq);
                A = log(v);
                if (A >= (t - rho)) {
                    if (A > (t + rho)) continue;
                    x1 = y + 1; f1 = m + 1; z = n + 1 - m; w = n - y + 1;
                    x2 = x1 * x1; f2 = f1 * f1; z2 = z * z; w2 = w * w;
                    if (A > (xm * log(f1 / x1)
                             + (n - m + 0.5) * log(z / w)
                             + (y - m) * log(w * r / (x1 * q))
                             + (13680. - (462. - (132. - (99. - 140. / f2) / f2) / f2) / f2) / f1 / 166320.
                             + (13680. - (462. - (132. - (99. - 140. / z2) / z2) / z2) / z2) / z / 166320.
                             + (13680. - (462. - (132. - (99. - 140. / x2) / x2) / x2) / x2) / x1 / 166320.
                             + (13680. - (462. - (132. - (99. - 140. / w2) / w2) / w2) / w2) / w / 166320.))
                        continue;
                }
            } else {
                s = r / q;
                a = s * (n + 1);
                F = 1.0;
                if (m < y) {
                    for (i = m + 1; i <= y; i++) F *= (a / i - s);
                } else if (m > y) {
                    for (i = y + 1; i <= m; i++) F /= (a / i - s);
                }
                if (v > F) continue;
            }
        }
        break; /* Step60 */
    }
    if (p > 0.5) y = n - y;
    return y;
}

/* distributions.c:442-470 */
long ork_binomial(ork_state *st, long n, double p)
{
    if (p <= 0.5) {
        if (p * n <= 30.0) return ork_binomial_inversion(st, n, p);
        return ork_binomial_btpe(st, n, p);
    } else {
        double q = 1.0 - p;
        if (q * n <= 30.0) return n - ork_binomial_inversion(st, n, q);
        return n - ork_binomial_btpe(st, n, q);
    }
}

/* random_fast.pyx:87-111 : all swap targets are drawn before any swap is
   applied; i runs n-1 .. 1.  `swaps` (length n) receives the targets. */
void ork_draw_swaps(ork_state *st, long n, int64_t *swaps)
{
    for (long i = n - 1; i > 0; i--)
        swaps[i] = (int64_t)ork_interval(st, (uint64_t)i);
}

void ork_apply_swaps_i64(int64_t *x, long n, const int64_t *swaps)
{
    for (long i = n - 1; i > 0; i--) {
        int64_t j = swaps[i];
        int64_t t = x[i]; x[i] = x[j]; x[j] = t;
    }
}

/* row swaps of a C-contiguous (n, row_bytes) array (random_fast.pyx:112-119) */
void ork_apply_swaps_rows(void *base, long n, size_t row_bytes, const int64_t *swaps)
{
    char *b = (char *)base;
    char *tmp = (char *)malloc(row_bytes);
    for (long i = n - 1; i > 0; i--) {
        long j = (long)swaps[i];
        if (j == i) continue;
        memcpy(tmp, b + (size_t)i * row_bytes, row_bytes);
        memcpy(b + (size_t)i * row_bytes, b + (size_t)j * row_bytes, row_bytes);
        memcpy(b + (size_t)j * row_bytes, tmp, row_bytes);
    }
    free(tmp);
}

void ork_shuffle_i64(ork_state *st, int64_t *x, long n)
{
    if (n <= 1) return;
    int64_t *sw = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    ork_draw_swaps(st, n, sw);
    ork_apply_swaps_i64(x, n, sw);
    free(sw);
}

/* random_fast.pyx:79-85 */
void ork_permutation(ork_state *st, int64_t *out, long n)
{
    for (long i = 0; i < n; i++) out[i] = i;
    ork_shuffle_i64(st, out, n);
}

/* random_fast.pyx:127-144 : swaps drawn once, applied to the trace while
   being drawn; caller applies `swaps` to each array afterwards. */
void ork_shuffle_trace(ork_state *st, long n, int64_t *trace, int64_t *swaps)
{
    for (long i = 0; i < n; i++) trace[i] = i;
    for (long i = n - 1; i > 0; i--) {
        int64_t j = (int64_t)ork_interval(st, (uint64_t)i);
        swaps[i] = j;
        int64_t t = trace[i]; trace[i] = trace[j]; trace[j] = t;
    }
}

/* ------------------------------------------------------------------ */
/* Sampler (sampler.pyx:10-70)                                          */

typedef struct ork_sampler {
    ork_state rs;
    long range;
    int rand_size, replacement;
    int64_t *box, *temp;
    long lim_inf, lim_sup;
} ork_sampler;

ork_sampler *ork_sampler_new(long range, int rand_size, int replacement, uint64_t seed)
{
    ork_sampler *s = (ork_sampler *)calloc(1, sizeof(ork_sampler));
    s->range = range; s->rand_size = rand_size; s->replacement = replacement;
    ork_seed(&s->rs, seed);
    s->box = (int64_t *)malloc(sizeof(int64_t) * (size_t)(range > 0 ? range : 1));
    s->temp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(range > 0 ? range : 1));
    ork_permutation(&s->rs, s->box, range);   /* sampler.pyx:34 */
    ork_shuffle_i64(&s->rs, s->box, range);   /* sampler.pyx:39 : second full shuffle */
    s->lim_inf = s->lim_sup = 0;
    return s;
}

void ork_sampler_free(ork_sampler *s)
{
    if (!s) return;
    free(s->box); free(s->temp); free(s);
}

/* writes the subset into out (capacity >= range), returns its length */
long ork_sampler_yield(ork_sampler *s, double reduction, int64_t *out)
{
    long len;
    if (s->rand_size)
        len = (long)(int)ork_binomial(&s->rs, (long)(int)s->range, 1.0 / reduction); /* random_fast.pyx:146-147 */
    else
        len = (long)((double)s->range / reduction); /* int(range / reduction) */
    if (s->replacement) {
        ork_shuffle_i64(&s->rs, s->box, s->range);
        s->lim_inf = 0; s->lim_sup = len;
    } else {
        if (s->range != len) {
            s->lim_inf = s->lim_sup;
            long rem = s->range - s->lim_inf;
            if (rem == 0) {
                ork_shuffle_i64(&s->rs, s->box, s->range);
                s->lim_inf = 0;
            } else if (rem < len) {
                /* rotate the unseen tail to the front, reshuffle the rest
                   (sampler.pyx:60-65) */
                memcpy(s->temp, s->box, sizeof(int64_t) * (size_t)rem);
                memmove(s->box, s->box + s->lim_inf, sizeof(int64_t) * (size_t)rem);
                memcpy(s->box + s->lim_inf, s->temp, sizeof(int64_t) * (size_t)rem);
                ork_shuffle_i64(&s->rs, s->box + rem, s->range - rem);
                s->lim_inf = 0;
            }
            s->lim_sup = s->lim_inf + len;
        } else {
            s->lim_inf = 0; s->lim_sup = s->range;
        }
    }
    long n = s->lim_sup - s->lim_inf;
    memcpy(out, s->box + s->lim_inf, sizeof(int64_t) * (size_t)(n > 0 ? n : 0));
    return n;
}

/* standalone RandomState handle for ctypes */
ork_state *ork_state_new(uint64_t seed)
{
    ork_state *st = (ork_state *)calloc(1, sizeof(ork_state));
    ork_seed(st, seed);
    return st;
}
void ork_state_free(ork_state *st) { free(st); }
