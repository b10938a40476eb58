// gauss_table.cpp -- host-side construction of the cumulative table of a discrete Gaussian
// (the "barriers" of FastGaussianNoise, reference include/nfl/prng/FastGaussianNoise.hpp:231-330).
//
// The reference computes the barriers with MPFR at bit_precision = ceil(k + log2(2*tail*sigma)) bits, where
// k = security + 1 + ceil(log2(samples)) and tail solves tail^2 - 2 ln(tail) - 1 - 2 k ln 2 = 0 (lines 239-262).
// No multiprecision library is assumed here: probabilities are evaluated in 64.448-bit unsigned fixed point
// (exp by range reduction + Taylor series) and exported with 64*W bits, W = ceil(bit_precision/64) <= 6 (security
// parameters up to ~ 360 bits; the reference takes its precision from MPFR and has no such cap).
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "gauss_table.h"

namespace nflhip {
namespace {

typedef unsigned __int128 u128;

// unsigned fixed point, value = sum l[i] * 2^(64 i - FB); l[FL] is the integer part
constexpr int FL = 7;            // fraction limbs (one guard limb beyond the widest export)
constexpr int NL = FL + 1;       // limbs
constexpr int FB = 64 * FL;      // fraction bits
constexpr int kMaxWords = 6;     // widest exported entry
struct Fix {
  uint64_t l[NL];
};
static Fix fix_zero() {
  Fix r;
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
  return r;
}
static Fix fix_int(uint64_t v) {
  Fix r = fix_zero();
  r.l[FL] = v;
  return r;
}
static bool fix_is_zero(const Fix &a) {
  uint64_t o = 0;
  for (int i = 0; i < NL; ++i) o |= a.l[i];
  return o == 0;
}
static Fix fix_add(const Fix &a, const Fix &b) {
  Fix r;
  u128 c = 0;
  for (int i = 0; i < NL; ++i) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return r;
}
static Fix fix_sub(const Fix &a, const Fix &b) {  // a >= b
  Fix r;
  unsigned borrow = 0;
  for (int i = 0; i < NL; ++i) {
    const u128 t = (u128)a.l[i] - b.l[i] - borrow;
    r.l[i] = (uint64_t)t;
    borrow = (unsigned)((t >> 64) & 1);
  }
  return r;
}
static int fix_cmp(const Fix &a, const Fix &b) {
  for (int i = NL - 1; i >= 0; --i)
    if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
  return 0;
}
static Fix fix_mul(const Fix &a, const Fix &b) {  // truncated product (the result must fit)
  uint64_t p[2 * NL];
  for (int i = 0; i < 2 * NL; ++i) p[i] = 0;
  for (int i = 0; i < NL; ++i) {
    u128 c = 0;
    for (int j = 0; j < NL; ++j) {
      c += (u128)a.l[i] * b.l[j] + p[i + j];
      p[i + j] = (uint64_t)c;
      c >>= 64;
    }
    p[i + NL] = (uint64_t)c;
  }
  Fix r;
  for (int i = 0; i < NL; ++i) r.l[i] = p[i + FL];  // >> FB
  return r;
}
static Fix fix_div_small(const Fix &a, uint64_t d) {
  Fix r;
  u128 rem = 0;
  for (int i = NL - 1; i >= 0; --i) {
    const u128 cur = (rem << 64) | a.l[i];
    r.l[i] = (uint64_t)(cur / d);
    rem = cur % d;
  }
  return r;
}
// a / b as fixed point (restoring division, 64*NL result bits)
static Fix fix_div(const Fix &a, const Fix &b) {
  // numerator a * 2^FB as 2*NL limbs, shifted in bit by bit
  uint64_t num[2 * NL];
  for (int i = 0; i < 2 * NL; ++i) num[i] = 0;
  for (int i = 0; i < NL; ++i) num[i + FL] = a.l[i];
  Fix rem = fix_zero(), q = fix_zero();
  uint64_t rem_hi = 0;  // the bit above the running remainder
  for (int bit = 64 * (NL + FL) - 1; bit >= 0; --bit) {
    // rem = (rem << 1) | num[bit]
    rem_hi = rem.l[NL - 1] >> 63;
    for (int i = NL - 1; i > 0; --i) rem.l[i] = (rem.l[i] << 1) | (rem.l[i - 1] >> 63);
    rem.l[0] = (rem.l[0] << 1) | ((num[bit >> 6] >> (bit & 63)) & 1);
    if (rem_hi || fix_cmp(rem, b) >= 0) {
      rem = fix_sub(rem, b);  // (wraps correctly when rem_hi is set)
      if (bit < 64 * NL) q.l[bit >> 6] |= ((uint64_t)1) << (bit & 63);
    }
  }
  return q;
}
static Fix fix_from_double(double v) {  // v >= 0, exact (a double has 53 significant bits)
  Fix r = fix_zero();
  if (!(v > 0)) return r;
  int e;
  const double m = std::frexp(v, &e);             // v = m * 2^e, m in [0.5, 1)
  const uint64_t mant = (uint64_t)std::ldexp(m, 53);  // 53-bit integer
  const int sh = e - 53 + FB;                      // value = mant * 2^(sh - FB)
  for (int i = 0; i < 53; ++i)
    if ((mant >> i) & 1) {
      const int pos = sh + i;
      if (pos >= 0 && pos < 64 * NL) r.l[pos >> 6] |= ((uint64_t)1) << (pos & 63);
    }
  return r;
}
// exp(-t), t >= 0
static Fix fix_exp_neg(const Fix &t) {
  const uint64_t m = t.l[FL];
  Fix f = t;
  f.l[FL] = 0;  // fractional part in [0,1)
  // e^-f = sum (-f)^k / k!  (alternating, terms decrease): accumulate positive and negative parts separately
  Fix pos = fix_int(1), neg = fix_zero(), term = fix_int(1);
  for (uint64_t k = 1; k < 200; ++k) {
    term = fix_div_small(fix_mul(term, f), k);
    if (fix_is_zero(term)) break;
    if (k & 1) neg = fix_add(neg, term); else pos = fix_add(pos, term);
  }
  Fix r = fix_sub(pos, neg);
  if (m) {
    // e^-1 by the same series at f = 1
    Fix p1 = fix_int(1), n1 = fix_zero(), t1 = fix_int(1);
    for (uint64_t k = 1; k < 200; ++k) {
      t1 = fix_div_small(t1, k);
      if (fix_is_zero(t1)) break;
      if (k & 1) n1 = fix_add(n1, t1); else p1 = fix_add(p1, t1);
    }
    Fix base = fix_sub(p1, n1);
    uint64_t e = m;
    while (e) {  // square and multiply
      if (e & 1) r = fix_mul(r, base);
      base = fix_mul(base, base);
      e >>= 1;
    }
  }
  return r;
}

}  // namespace

int build_gauss_table(double sigma, unsigned security, unsigned samples, double center, GaussTable *out,
                      std::string *err) {
  if (!(sigma > 0) || !std::isfinite(sigma) || !std::isfinite(center) || security == 0 || samples == 0) {
    *err = "gaussian: sigma must be positive and finite, security and samples positive";
    return 1;
  }
  // FastGaussianNoise::init (FastGaussianNoise.hpp:239-262)
  const double k = (double)security + 1 + std::ceil(std::log((double)samples) / std::log(2.0));
  double tail = std::sqrt(1 + 2 * k * std::log(2.0));
  for (int it = 0; it < 1 << 15; ++it) {  // Newton on x^2 - 2 ln x - 1 - 2 k ln 2 (lines 118-158)
    const double fv = tail * tail - 2 * std::log(tail) - 1 - 2 * k * std::log(2.0), dv = 2 * tail - 2 / tail;
    const double delta = fv / dv;
    tail -= delta;
    if (std::fabs(delta) / std::fabs(tail) < 1e-3) break;
  }
  while (0.95 * tail * 0.95 * tail - 2 * std::log(0.95 * tail) - 1 - 2 * k * std::log(2.0) >= 0) tail *= 0.95;
  const double epsi = k + std::log2(2 * tail * sigma);
  const unsigned bit_precision = (unsigned)std::ceil(epsi);
  int words = (int)((bit_precision + 63) / 64);
  if (words < 1) words = 1;
  if (words > kMaxWords) {
    *err = "gaussian: the requested security needs more than 384 bits of table precision";
    return 1;
  }
  const long long half = (long long)std::ceil(tail * sigma);
  const long long rc = (long long)std::llround(center);
  const size_t entries = (size_t)(2 * half + 1);  // _number_of_barriers (line 272)
  if (entries > (1u << 22)) {
    *err = "gaussian: table too large";
    return 1;
  }
  // 1 / (2 sigma^2)
  const Fix sig = fix_from_double(sigma);
  const Fix two_sig2 = fix_mul(fix_mul(sig, sig), fix_int(2));
  if (fix_is_zero(two_sig2)) {
    *err = "gaussian: sigma too small";
    return 1;
  }
  const Fix inv = fix_div(fix_int(1), two_sig2);
  std::vector<Fix> cum(entries);
  Fix sum = fix_zero();
  for (size_t i = 0; i < entries; ++i) {
    const long long x = rc - half + (long long)i;
    const double dist = std::fabs((double)x - center);  // exact: |x - c| with x an integer and c a double
    const Fix dd = fix_from_double(dist);
    const Fix t = fix_mul(fix_mul(dd, dd), inv);
    sum = fix_add(sum, fix_exp_neg(t));
    cum[i] = sum;
  }
  out->words = words;
  out->entries = entries;
  out->x_min = rc - half;
  out->tail = tail;
  out->bit_precision = bit_precision;
  out->cdt.assign(entries * (size_t)words, 0);
  for (size_t i = 0; i < entries; ++i) {
    uint64_t *e = &out->cdt[i * (size_t)words];
    if (i + 1 == entries) {
      for (int w = 0; w < words; ++w) e[w] = ~(uint64_t)0;  // P(X <= max) = 1
      break;
    }
    const Fix q = fix_div(cum[i], sum);  // in [0,1): fractional limbs l[FL-1] (most significant) ... l[0]
    for (int w = 0; w < words; ++w) e[w] = q.l[FL - 1 - w];
  }
  return 0;
}

}  // namespace nflhip
