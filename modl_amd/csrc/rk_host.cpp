// Host-side random streams of the SOMF path: MT19937 generator, bounded draws,
// binomial variates and the feature-subset sampler.  The streams must be
// bit-identical to the reference's RandomKit build for a given seed:
//   reference: modl/utils/randomkit/randomkit.c:138-152,212-297 (generator),
//              modl/utils/randomkit/distributions.c:262-470 (binomial),
//              modl/utils/randomkit/random_fast.pyx:79-144 (shuffles),
//              modl/utils/randomkit/sampler.pyx:10-70 (Sampler).
// Pure host C++ (no HIP calls) so it also runs where no GPU is present.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/modl_hip.h"

namespace modl {

class Mt19937 {
public:
    static constexpr int kN = 624, kM = 397;

    void seed(uint64_t s64) {
        uint32_t s = static_cast<uint32_t>(s64);   // only the low word seeds the stream
        for (int i = 0; i < kN; ++i) {
            key_[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + static_cast<uint32_t>(i) + 1u;
        }
        pos_ = kN;
    }

    uint32_t next() {
        if (pos_ >= kN) refill();
        return out_[pos_++];
    }

    // uniform integer in [0, hi] by masked rejection; hi == 0 consumes nothing
    uint64_t bounded(uint64_t hi) {
        if (hi == 0) return 0;
        uint64_t mask = hi;
        for (int sh = 1; sh < 64; sh <<= 1) mask |= mask >> sh;
        uint64_t v;
        if (hi <= 0xffffffffull) {
            do v = next() & mask; while (v > hi);
        } else {
            do {
                uint64_t top = next();
                v = ((top << 32) | next()) & mask;
            } while (v > hi);
        }
        return v;
    }

    // 53-bit uniform in [0, 1)
    double uniform() {
        const int64_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

    uint32_t key_[kN];
    int32_t pos_ = kN;              // outputs of the current block of 624 consumed so far (kN: none left)

    // after the state (key_, pos_) has been set from outside (pickling): the tempered outputs of the current block
    void restore() { temper(); }

private:
    uint32_t out_[kN];              // tempered outputs of the current block (a pure function of key_)

    static uint32_t twist(uint32_t hi, uint32_t lo, uint32_t far) {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    // The state update in three branch-free runs (no modulo in the index: the loops vectorise - the distance to
    // the element a run depends on is 227 or more) and the tempering of the whole block at once: the subset draw
    // is a shuffle of all p feature indices per minibatch, i.e. p outputs (2.7 -> 1.x ms at p = 200 000).
    void refill() {
        for (int i = 0; i < kN - kM; ++i) key_[i] = twist(key_[i], key_[i + 1], key_[i + kM]);
        for (int i = kN - kM; i < kN - 1; ++i) key_[i] = twist(key_[i], key_[i + 1], key_[i + kM - kN]);
        key_[kN - 1] = twist(key_[kN - 1], key_[0], key_[kM - 1]);
        temper();
        pos_ = 0;
    }
    void temper() {
        for (int i = 0; i < kN; ++i) {
            uint32_t y = key_[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out_[i] = y;
        }
    }
};

// Binomial sampler with the set-up cache keyed on (n, p), shared between the
// inversion and the BTPE algorithm exactly as distributions.c shares rk_state.
struct BinomialCache {
    bool valid = false;
    int64_t n = 0;
    double p = 0;
    double r = 0, q = 0, fm = 0, p1 = 0, xm = 0, xl = 0, xr = 0, c = 0, laml = 0, lamr = 0, p2 = 0, p3 = 0,
           p4 = 0;
    int64_t m = 0;
    bool hit(int64_t n_, double p_) const { return valid && n == n_ && p == p_; }
};

static int64_t binomial_inversion(Mt19937 &g, BinomialCache &bc, int64_t n, double p) {
    if (!bc.hit(n, p)) {
        bc.valid = true; bc.n = n; bc.p = p;
        bc.q = 1.0 - p;
        bc.r = std::exp(n * std::log(bc.q));
        bc.c = n * p;
        const double lim = bc.c + 10.0 * std::sqrt(bc.c * bc.q + 1);
        bc.m = static_cast<int64_t>(static_cast<double>(n) < lim ? static_cast<double>(n) : lim);
    }
    const double q = bc.q, qn = bc.r;
    const int64_t bound = bc.m;
    int64_t x = 0;
    double px = qn, u = g.uniform();
    while (u > px) {
        ++x;
        if (x > bound) { x = 0; px = qn; u = g.uniform(); }
        else { u -= px; px = ((n - x + 1) * p * px) / (x * q); }
    }
    return x;
}

static inline double stirling_tail(double v) {
    const double v2 = v * v;
    return (13680. - (462. - (132. - (99. - 140. / v2) / v2) / v2) / v2) / v / 166320.;
}

static int64_t binomial_btpe(Mt19937 &g, BinomialCache &bc, int64_t n, double p) {
    if (!bc.hit(n, p)) {
        bc.valid = true; bc.n = n; bc.p = p;
        bc.r = std::min(p, 1.0 - p);
        bc.q = 1.0 - bc.r;
        bc.fm = n * bc.r + bc.r;
        bc.m = static_cast<int64_t>(std::floor(bc.fm));
        bc.p1 = std::floor(2.195 * std::sqrt(n * bc.r * bc.q) - 4.6 * bc.q) + 0.5;
        bc.xm = bc.m + 0.5;
        bc.xl = bc.xm - bc.p1;
        bc.xr = bc.xm + bc.p1;
        bc.c = 0.134 + 20.5 / (15.3 + bc.m);
        double a = (bc.fm - bc.xl) / (bc.fm - bc.xl * bc.r);
        bc.laml = a * (1.0 + a / 2.0);
        a = (bc.xr - bc.fm) / (bc.xr * bc.q);
        bc.lamr = a * (1.0 + a / 2.0);
        bc.p2 = bc.p1 * (1.0 + 2.0 * bc.c);
        bc.p3 = bc.p2 + bc.c / bc.laml;
        bc.p4 = bc.p3 + bc.c / bc.lamr;
    }
    const double r = bc.r, q = bc.q, p1 = bc.p1, xm = bc.xm, xl = bc.xl, xr = bc.xr, c = bc.c,
                 laml = bc.laml, lamr = bc.lamr, p2 = bc.p2, p3 = bc.p3, p4 = bc.p4;
    const int64_t m = bc.m;
    const double nrq = n * r * q;
    int64_t y;
    for (;;) {
        const double u = g.uniform() * p4;
        double v = g.uniform();
        if (u <= p1) {                       // triangular centre: accept at once
            y = static_cast<int64_t>(std::floor(xm - p1 * v + u));
            break;
        }
        if (u <= p2) {                       // parallelogram
            const double x = xl + (u - p1) / c;
            v = v * c + 1.0 - std::fabs(m - x + 0.5) / p1;
            if (v > 1.0) continue;
            y = static_cast<int64_t>(std::floor(x));
        } else if (u <= p3) {                // left exponential tail
            y = static_cast<int64_t>(std::floor(xl + std::log(v) / laml));
            if (y < 0) continue;
            v = v * (u - p2) * laml;
        } else {                             // right exponential tail
            y = static_cast<int64_t>(std::floor(xr - std::log(v) / lamr));
            if (y > n) continue;
            v = v * (u - p3) * lamr;
        }
        const int64_t kd = std::llabs(y - m);
        if (kd > 20 && kd < nrq / 2.0 - 1) { // squeeze, then Stirling comparison
            const double rho = (kd / nrq) * ((kd * (kd / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
            const double t = -kd * kd / (2 * nrq);
            const double A = std::log(v);
            if (A < t - rho) break;
            if (A > t + rho) continue;
            const double x1 = y + 1, f1 = m + 1, z = n + 1 - m, w = n - y + 1;
            const double bound = xm * std::log(f1 / x1) + (n - m + 0.5) * std::log(z / w) +
                                 (y - m) * std::log(w * r / (x1 * q)) + stirling_tail(f1) + stirling_tail(z) +
                                 stirling_tail(x1) + stirling_tail(w);
            if (A > bound) continue;
            break;
        }
        const double s = r / q, a = s * (n + 1);   // explicit recursive evaluation of f(y)/f(m)
        double F = 1.0;
        if (m < y) for (int64_t i = m + 1; i <= y; ++i) F *= (a / i - s);
        else if (m > y) for (int64_t i = y + 1; i <= m; ++i) F /= (a / i - s);
        if (v > F) continue;
        break;
    }
    return (p > 0.5) ? n - y : y;
}

struct RandomStream {
    Mt19937 gen;
    BinomialCache bc;

    explicit RandomStream(uint64_t seed) { reseed(seed); }
    void reseed(uint64_t seed) { gen.seed(seed); bc.valid = false; }

    int64_t binomial(int64_t n, double p) {
        if (p <= 0.5)
            return (p * n <= 30.0) ? binomial_inversion(gen, bc, n, p) : binomial_btpe(gen, bc, n, p);
        const double q = 1.0 - p;
        return n - ((q * n <= 30.0) ? binomial_inversion(gen, bc, n, q) : binomial_btpe(gen, bc, n, q));
    }

    // the reference draws every swap target first (i = n-1 .. 1), then swaps
    void draw_swaps(int64_t n, int64_t *sw) {
        for (int64_t i = n - 1; i > 0; --i) sw[i] = static_cast<int64_t>(gen.bounded(static_cast<uint64_t>(i)));
    }
    // The reference draws every swap target first and swaps afterwards (random_fast.pyx:93-111); a target does not
    // depend on the array, so drawing and swapping in one pass gives the same permutation from the same draws.  The
    // rejection mask (the smallest all-ones mask >= i, randomkit.c:268-276) only changes when i crosses a power of
    // two: it is carried along instead of being rebuilt per element (the subset draw is ~8 ns per feature of pure
    // host time per minibatch: 1.6 ms at p = 200 000).
    template <typename E>
    void shuffle(E *x, int64_t n) {
        if (n < 2) return;
        if (static_cast<uint64_t>(n - 1) > 0xffffffffull) {          // 64-bit targets: the general path
            scratch_.resize(static_cast<size_t>(n));
            draw_swaps(n, scratch_.data());
            for (int64_t i = n - 1; i > 0; --i) std::swap(x[i], x[scratch_[i]]);
            return;
        }
        uint32_t mask = static_cast<uint32_t>(n - 1);
        for (int sh = 1; sh < 32; sh <<= 1) mask |= mask >> sh;
        // Targets are drawn in runs of up to kRun swaps, then applied (same draws, same swaps, same order: a target
        // does not depend on the array).  The draw loop has NO data-dependent branch: the rejection test of the masked
        // draw (randomkit.c:268-276) accepts with a probability between 1/2 and 1 that no predictor can follow - every
        // output is stored, and the cursor and the bound only move when it was accepted (2.7 -> 1.x ms per draw at
        // p = 200 000, where the subset draw is what the host spends per minibatch).
        constexpr int kRun = 256;
        uint32_t tgt[kRun + 1];
        int64_t i = n - 1;
        while (i > 0) {
            const int run = static_cast<int>(i < kRun ? i : kRun);
            uint32_t hi = static_cast<uint32_t>(i);
            int cnt = 0;
            while (cnt < run) {
                while ((mask >> 1) >= hi) mask >>= 1;                 // (changes log2(n) times in all)
                const uint32_t v = gen.next() & mask;
                const uint32_t ok = v <= hi ? 1u : 0u;
                tgt[cnt] = v;
                cnt += static_cast<int>(ok);
                hi -= ok;
            }
            for (int j = 0; j < run; ++j) std::swap(x[i - j], x[tgt[j]]);
            i -= run;
        }
    }
    template <typename E>
    void permutation(E *out, int64_t n) {
        for (int64_t i = 0; i < n; ++i) out[i] = static_cast<E>(i);
        shuffle(out, n);
    }

private:
    std::vector<int64_t> scratch_;
};

class FeatureSampler {
public:
    FeatureSampler(int64_t range, bool rand_size, bool replacement, uint64_t seed)
        : rs_(seed), range_(range), rand_size_(rand_size), replacement_(replacement),
          box_(static_cast<size_t>(std::max<int64_t>(range, 1))), tmp_(box_.size()) {
        rs_.permutation(box_.data(), range_);
        rs_.shuffle(box_.data(), range_);            // the constructor shuffles the fresh permutation again
    }

    int64_t draw(double reduction, int64_t *out) {
        int64_t len;
        if (rand_size_)  // n goes through a C int in the reference (random_fast.pyx:146)
            len = static_cast<int32_t>(rs_.binomial(static_cast<int32_t>(range_), 1.0 / reduction));
        else
            len = static_cast<int64_t>(static_cast<double>(range_) / reduction);
        if (replacement_) {
            rs_.shuffle(box_.data(), range_);
            lo_ = 0; hi_ = len;
        } else if (range_ == len) {
            lo_ = 0; hi_ = range_;
        } else {
            lo_ = hi_;
            const int64_t left = range_ - lo_;
            if (left == 0) {
                rs_.shuffle(box_.data(), range_);
                lo_ = 0;
            } else if (left < len) {
                // bring the unseen tail to the front, swap the displaced head in behind it and
                // reshuffle everything after the tail
                std::copy(box_.begin(), box_.begin() + left, tmp_.begin());
                std::memmove(box_.data(), box_.data() + lo_, sizeof(int32_t) * static_cast<size_t>(left));
                std::copy(tmp_.begin(), tmp_.begin() + left, box_.begin() + lo_);
                rs_.shuffle(box_.data() + left, range_ - left);
                lo_ = 0;
            }
            hi_ = lo_ + len;
        }
        const int64_t n = hi_ - lo_;
        for (int64_t i = 0; i < n; ++i) out[i] = box_[static_cast<size_t>(lo_ + i)];
        return n;
    }

    RandomStream rs_;
    int64_t range_;
    bool rand_size_, replacement_;
    std::vector<int32_t> box_, tmp_;   // feature indices (range <= 2^31 - 1): half the cache footprint of the int64 box
    int64_t lo_ = 0, hi_ = 0;
};

struct SamplerBlob {   // serialised state header (followed by box_)
    int64_t range, lo, hi;
    int32_t rand_size, replacement, pos, bc_valid;
    uint32_t key[Mt19937::kN];
    BinomialCache bc;
};

}  // namespace modl

struct modl_rk { modl::RandomStream rs; explicit modl_rk(uint64_t s) : rs(s) {} };
struct modl_sampler { modl::FeatureSampler fs; modl_sampler(int64_t r, bool a, bool b, uint64_t s) : fs(r, a, b, s) {} };

extern "C" {

int modl_abi_version(void) { return MODL_ABI_VERSION; }

const char *modl_error_string(int code) {
    switch (code) {
        case MODL_OK: return "ok";
        case MODL_EINVAL: return "invalid argument";
        case MODL_ENOMEM: return "out of memory / workspace too small";
        case MODL_ESTATE: return "invalid state";
        case MODL_ENOGPU: return "no HIP device (or one with less than 160 KiB of LDS per compute unit)";
        case MODL_ENORCCL: return "librccl.so could not be loaded";
        case MODL_ERCCL: return "RCCL call failed";
        case MODL_ETIMEOUT: return "a persistent dictionary-update launch gave up half-way (its wait gave up: workgroups not resident); the update is incomplete";
        default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
    }
}

int modl_rk_create(uint64_t seed, modl_rk **out) {
    if (!out) return MODL_EINVAL;
    *out = new (std::nothrow) modl_rk(seed);
    return *out ? MODL_OK : MODL_ENOMEM;
}
void modl_rk_destroy(modl_rk *rk) { delete rk; }
int modl_rk_seed(modl_rk *rk, uint64_t seed) { if (!rk) return MODL_EINVAL; rk->rs.reseed(seed); return MODL_OK; }
int modl_rk_random(modl_rk *rk, uint32_t *out) { if (!rk || !out) return MODL_EINVAL; *out = rk->rs.gen.next(); return MODL_OK; }
int modl_rk_randint(modl_rk *rk, uint64_t high, int64_t *out) {
    if (!rk || !out) return MODL_EINVAL;
    *out = static_cast<int64_t>(rk->rs.gen.bounded(high));
    return MODL_OK;
}
int modl_rk_double(modl_rk *rk, double *out) { if (!rk || !out) return MODL_EINVAL; *out = rk->rs.gen.uniform(); return MODL_OK; }
int modl_rk_binomial(modl_rk *rk, int64_t n, double p, int64_t *out) {
    if (!rk || !out || n < 0 || !(p >= 0.0 && p <= 1.0)) return MODL_EINVAL;
    *out = rk->rs.binomial(n, p);
    return MODL_OK;
}
/* The generator's raw MT19937 state (key[624], pos: outputs of the current block already consumed, 624 = none left) -
 * the layout of numpy's legacy RandomState.get_state()[1:3].  numpy's legacy permutation(n) is shuffle(arange(n)) with
 * the same masked-rejection draws as randomkit's (legacy random_interval == rk_interval), so a modl_rk loaded with
 * numpy's state continues numpy's stream: the atom order of dict_fact.py:672 can be drawn on this side of the ABI. */
int modl_rk_get_mt_state(const modl_rk *rk, uint32_t *h_key624, int32_t *pos) {
    if (!rk || !h_key624 || !pos) return MODL_EINVAL;
    std::memcpy(h_key624, rk->rs.gen.key_, sizeof(uint32_t) * modl::Mt19937::kN);
    *pos = rk->rs.gen.pos_;
    return MODL_OK;
}
int modl_rk_set_mt_state(modl_rk *rk, const uint32_t *h_key624, int32_t pos) {
    if (!rk || !h_key624 || pos < 0 || pos > modl::Mt19937::kN) return MODL_EINVAL;
    std::memcpy(rk->rs.gen.key_, h_key624, sizeof(uint32_t) * modl::Mt19937::kN);
    rk->rs.gen.pos_ = pos;
    rk->rs.gen.restore();
    return MODL_OK;
}
int modl_rk_permutation(modl_rk *rk, int64_t n, int64_t *h_out) {
    if (!rk || n < 0 || (n > 0 && !h_out)) return MODL_EINVAL;
    rk->rs.permutation(h_out, n);
    return MODL_OK;
}
int modl_rk_shuffle_i64(modl_rk *rk, int64_t *h_x, int64_t n) {
    if (!rk || n < 0 || (n > 0 && !h_x)) return MODL_EINVAL;
    rk->rs.shuffle(h_x, n);
    return MODL_OK;
}
int modl_rk_shuffle_trace(modl_rk *rk, int64_t n, int64_t *h_trace, int64_t *h_swaps) {
    if (!rk || n < 0 || (n > 0 && (!h_trace || !h_swaps))) return MODL_EINVAL;
    for (int64_t i = 0; i < n; ++i) { h_trace[i] = i; h_swaps[i] = i; }
    // the trace follows each swap as it is drawn (same outcome as draw-all-then-apply)
    for (int64_t i = n - 1; i > 0; --i) {
        const int64_t j = static_cast<int64_t>(rk->rs.gen.bounded(static_cast<uint64_t>(i)));
        h_swaps[i] = j;
        std::swap(h_trace[i], h_trace[j]);
    }
    return MODL_OK;
}
int modl_apply_swaps_rows(void *h_base, int64_t n, size_t row_bytes, const int64_t *h_swaps) {
    if (n < 0 || (n > 1 && (!h_base || !h_swaps))) return MODL_EINVAL;
    if (row_bytes == 0) return MODL_OK;
    std::vector<char> tmp(row_bytes);
    char *b = static_cast<char *>(h_base);
    for (int64_t i = n - 1; i > 0; --i) {
        const int64_t j = h_swaps[i];
        if (j < 0 || j > i) return MODL_EINVAL;
        if (j == i) continue;
        std::memcpy(tmp.data(), b + i * row_bytes, row_bytes);
        std::memcpy(b + i * row_bytes, b + j * row_bytes, row_bytes);
        std::memcpy(b + j * row_bytes, tmp.data(), row_bytes);
    }
    return MODL_OK;
}

int modl_sampler_create(int64_t range, int rand_size, int replacement, uint64_t seed, modl_sampler **out) {
    if (!out || range < 0 || range > 0x7fffffff) return MODL_EINVAL;   // the reference shuffles with C ints
    *out = new (std::nothrow) modl_sampler(range, rand_size != 0, replacement != 0, seed);
    return *out ? MODL_OK : MODL_ENOMEM;
}
void modl_sampler_destroy(modl_sampler *s) { delete s; }
int modl_sampler_yield_subset(modl_sampler *s, double reduction, int64_t *h_out, int64_t *n_out) {
    if (!s || !h_out || !n_out || !(reduction >= 1.0)) return MODL_EINVAL;
    *n_out = s->fs.draw(reduction, h_out);
    return MODL_OK;
}
int modl_sampler_get(modl_sampler *s, int64_t *range, int64_t *lim_inf, int64_t *lim_sup, int64_t *h_box) {
    if (!s) return MODL_EINVAL;
    if (range) *range = s->fs.range_;
    if (lim_inf) *lim_inf = s->fs.lo_;
    if (lim_sup) *lim_sup = s->fs.hi_;
    if (h_box)
        for (int64_t i = 0; i < s->fs.range_; ++i) static_cast<int64_t *>(h_box)[i] = s->fs.box_[static_cast<size_t>(i)];
    return MODL_OK;
}
size_t modl_sampler_state_bytes(const modl_sampler *s) {
    return s ? sizeof(modl::SamplerBlob) + sizeof(int64_t) * static_cast<size_t>(s->fs.range_) : 0;
}
int modl_sampler_get_state(const modl_sampler *s, void *h_buf, size_t bytes) {
    if (!s || !h_buf || bytes < modl_sampler_state_bytes(s)) return MODL_EINVAL;
    modl::SamplerBlob hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.range = s->fs.range_; hd.lo = s->fs.lo_; hd.hi = s->fs.hi_;
    hd.rand_size = s->fs.rand_size_; hd.replacement = s->fs.replacement_;
    hd.pos = s->fs.rs_.gen.pos_;
    std::memcpy(hd.key, s->fs.rs_.gen.key_, sizeof(hd.key));
    hd.bc = s->fs.rs_.bc;
    std::memcpy(h_buf, &hd, sizeof(hd));
    int64_t *box = reinterpret_cast<int64_t *>(static_cast<char *>(h_buf) + sizeof(hd));   // (the blob keeps int64 entries)
    for (int64_t i = 0; i < s->fs.range_; ++i) box[i] = s->fs.box_[static_cast<size_t>(i)];
    return MODL_OK;
}
int modl_sampler_set_state(modl_sampler *s, const void *h_buf, size_t bytes) {
    if (!s || !h_buf || bytes < sizeof(modl::SamplerBlob)) return MODL_EINVAL;
    modl::SamplerBlob hd;
    std::memcpy(&hd, h_buf, sizeof(hd));
    if (hd.range != s->fs.range_ || bytes < modl_sampler_state_bytes(s)) return MODL_EINVAL;
    s->fs.lo_ = hd.lo; s->fs.hi_ = hd.hi;
    s->fs.rand_size_ = hd.rand_size != 0; s->fs.replacement_ = hd.replacement != 0;
    s->fs.rs_.gen.pos_ = hd.pos;
    std::memcpy(s->fs.rs_.gen.key_, hd.key, sizeof(hd.key));
    s->fs.rs_.gen.restore();
    s->fs.rs_.bc = hd.bc;
    const int64_t *box = reinterpret_cast<const int64_t *>(static_cast<const char *>(h_buf) + sizeof(hd));
    for (int64_t i = 0; i < s->fs.range_; ++i) s->fs.box_[static_cast<size_t>(i)] = static_cast<int32_t>(box[i]);
    return MODL_OK;
}

int modl_batch_weight(int64_t count, int64_t batch_size, double learning_rate, double offset, double *out) {
    if (!out || batch_size < 0) return MODL_EINVAL;
    double keep = 1;
    for (int64_t i = count + 1 - batch_size; i <= count; ++i)
        keep *= (1 - std::pow((1 + offset) / (offset + i), learning_rate));
    *out = 1 - keep;
    return MODL_OK;
}

}  // extern "C"
