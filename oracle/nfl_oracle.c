/* nfl_oracle.c -- CPU oracle for the NFLlib NTT polynomial-ring hot path.
 *
 * TEST INFRASTRUCTURE ONLY: parity checker for the HIP engine and the
 * `cpu_baseline` leg of bench.py.  See nfl_oracle.h for the parity-pinning
 * statement.  A plain-C restatement of quarkslab/NFLlib @ v1; every function
 * cites the reference file:line it follows.
 */
#include "nfl_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* counter-based splitmix64: value g of stream (seed, operand)   */
static inline uint64_t oracle_splitmix64(uint64_t seed, int operand, uint64_t g) {
  uint64_t z = (seed ^ ((uint64_t)operand << 62)) + (g + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* ---- three instantiations of the typed body ---- */
#define FN(x) u16_##x
#define T uint16_t
#define ST int16_t
#define W uint32_t
#define WB 16
#define MULLO(a, b) ((uint16_t)((uint32_t)(a) * (uint32_t)(b)))
#include "nfl_oracle_typed.inc"
#undef FN
#undef T
#undef ST
#undef W
#undef WB
#undef MULLO

#define FN(x) u32_##x
#define T uint32_t
#define ST int32_t
#define W uint64_t
#define WB 32
#define MULLO(a, b) ((uint32_t)((a) * (b)))
#include "nfl_oracle_typed.inc"
#undef FN
#undef T
#undef ST
#undef W
#undef WB
#undef MULLO

#define FN(x) u64_##x
#define T uint64_t
#define ST int64_t
#define W unsigned __int128
#define WB 64
#define MULLO(a, b) ((uint64_t)((a) * (b)))
#include "nfl_oracle_typed.inc"
#undef FN
#undef T
#undef ST
#undef W
#undef WB
#undef MULLO

/* ---- little-endian multi-limb helpers for the CRT constants (stand in for libgmp,
 * which the reference uses: gmp.hpp:113-219) ---- */
#define BIG_MAX 2048
typedef struct {
  size_t n; /* significant limbs (no leading zero limb unless value is 0 -> n==0) */
  uint64_t v[BIG_MAX];
} big;

static void big_norm(big *a) {
  while (a->n > 0 && a->v[a->n - 1] == 0) a->n--;
}
static void big_set_u64(big *a, uint64_t x) {
  memset(a, 0, sizeof(*a));
  a->v[0] = x;
  a->n = x ? 1 : 0;
}
static int big_cmp(const big *a, const big *b) {
  if (a->n != b->n) return a->n > b->n ? 1 : -1;
  for (size_t k = a->n; k-- > 0;)
    if (a->v[k] != b->v[k]) return a->v[k] > b->v[k] ? 1 : -1;
  return 0;
}
static size_t big_bits(const big *a) {
  if (a->n == 0) return 0; /* mpz_sizeinbase(0,2) is 1, never hit here */
  uint64_t top = a->v[a->n - 1];
  size_t b = 0;
  while (top) { b++; top >>= 1; }
  return (a->n - 1) * 64 + b;
}
static void big_mul_u64(big *r, const big *a, uint64_t w) {
  unsigned __int128 c = 0;
  size_t i;
  big out;
  memset(&out, 0, sizeof(out));
  for (i = 0; i < a->n; i++) {
    c += (unsigned __int128)a->v[i] * w;
    out.v[i] = (uint64_t)c;
    c >>= 64;
  }
  out.v[i] = (uint64_t)c;
  out.n = a->n + 1;
  big_norm(&out);
  *r = out;
}
static void big_addmul_u64(big *acc, const big *a, uint64_t w) { /* mpz_addmul_ui */
  unsigned __int128 c = 0;
  size_t i, top = acc->n > a->n ? acc->n : a->n;
  for (i = 0; i < a->n; i++) {
    c += (unsigned __int128)a->v[i] * w + acc->v[i];
    acc->v[i] = (uint64_t)c;
    c >>= 64;
  }
  for (; c; i++) {
    c += acc->v[i];
    acc->v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (i > top) top = i;
  acc->n = top + 1 < BIG_MAX ? top + 1 : BIG_MAX;
  big_norm(acc);
}
static void big_mul(big *r, const big *a, const big *b) {
  big out;
  memset(&out, 0, sizeof(out));
  for (size_t i = 0; i < a->n; i++) {
    unsigned __int128 c = 0;
    for (size_t j = 0; j < b->n; j++) {
      c += (unsigned __int128)a->v[i] * b->v[j] + out.v[i + j];
      out.v[i + j] = (uint64_t)c;
      c >>= 64;
    }
    out.v[i + b->n] += (uint64_t)c;
  }
  out.n = a->n + b->n;
  big_norm(&out);
  *r = out;
}
static void big_sub(big *a, const big *b) { /* a -= b, a >= b */
  unsigned __int128 br = 0;
  for (size_t i = 0; i < a->n; i++) {
    unsigned __int128 d = (unsigned __int128)a->v[i] - (i < b->n ? b->v[i] : 0) - br;
    a->v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  big_norm(a);
}
static void big_shr(big *r, const big *a, size_t bits) {
  big out;
  memset(&out, 0, sizeof(out));
  size_t ws = bits / 64, bs = bits % 64;
  for (size_t i = ws; i < a->n; i++) {
    uint64_t lo = a->v[i] >> bs;
    uint64_t hi = (bs && i + 1 < a->n) ? (a->v[i + 1] << (64 - bs)) : 0;
    out.v[i - ws] = lo | hi;
  }
  out.n = a->n > ws ? a->n - ws : 0;
  big_norm(&out);
  *r = out;
}
static uint64_t big_divrem_u64(big *q, const big *a, uint64_t d) {
  unsigned __int128 r = 0;
  big out;
  memset(&out, 0, sizeof(out));
  for (size_t k = a->n; k-- > 0;) {
    r = (r << 64) | a->v[k];
    out.v[k] = (uint64_t)(r / d);
    r %= d;
  }
  out.n = a->n;
  big_norm(&out);
  if (q) *q = out;
  return (uint64_t)r;
}
/* q = floor(2^e / d) by shift-subtract long division */
static void big_pow2_div(big *q, size_t e, const big *d) {
  big rem, out;
  memset(&rem, 0, sizeof(rem));
  memset(&out, 0, sizeof(out));
  for (size_t bit = e + 1; bit-- > 0;) {
    /* rem = rem*2 + (bit==e) */
    uint64_t carry = (bit == e) ? 1 : 0;
    for (size_t i = 0; i < rem.n || carry; i++) {
      uint64_t nc = rem.v[i] >> 63;
      rem.v[i] = (rem.v[i] << 1) | carry;
      carry = nc;
      if (i >= rem.n) rem.n = i + 1;
    }
    big_norm(&rem);
    if (big_cmp(&rem, d) >= 0) {
      big_sub(&rem, d);
      out.v[bit / 64] |= (uint64_t)1 << (bit % 64);
    }
  }
  out.n = e / 64 + 1;
  big_norm(&out);
  *q = out;
}
static uint64_t powmod_u64(uint64_t a, uint64_t e, uint64_t p) {
  unsigned __int128 r = 1, b = a % p;
  while (e) {
    if (e & 1) r = r * b % p;
    b = b * b % p;
    e >>= 1;
  }
  return (uint64_t)r;
}

struct nfl_oracle_ctx {
  int limb_bits;
  size_t n, nm;
  void *tabs;
  /* CRT constants, poly::GMP (poly.hpp:251-275, gmp.hpp:113-155) */
  big Q, mshoup;
  big *lifting;
  size_t bitsQ, shift, L;
};

static uint64_t ctx_modulus(const nfl_oracle_ctx *c, size_t cm) {
  switch (c->limb_bits) {
    case 16: return ((u16_tabs *)c->tabs)->P[cm];
    case 32: return ((u32_tabs *)c->tabs)->P[cm];
    default: return ((u64_tabs *)c->tabs)->P[cm];
  }
}

/* GMP::GMP() (gmp.hpp:113-155) */
static void crt_init(nfl_oracle_ctx *c) {
  big_set_u64(&c->Q, 1);
  for (size_t cm = 0; cm < c->nm; cm++) big_mul_u64(&c->Q, &c->Q, ctx_modulus(c, cm));
  c->bitsQ = big_bits(&c->Q);
  size_t lg = 0;
  while (((size_t)2 << lg) <= c->nm) lg++; /* static_log2<nmoduli> = floor(log2) (meta.hpp:12-30) */
  c->shift = c->bitsQ + (size_t)c->limb_bits + lg + 1;
  big_pow2_div(&c->mshoup, c->shift, &c->Q);
  c->L = (c->bitsQ + 63) / 64;
  c->lifting = (big *)calloc(c->nm, sizeof(big));
  for (size_t cm = 0; cm < c->nm; cm++) {
    const uint64_t p = ctx_modulus(c, cm);
    big quot;
    big_divrem_u64(&quot, &c->Q, p);              /* mpz_divexact */
    uint64_t qmod = big_divrem_u64(NULL, &quot, p);
    uint64_t inv = powmod_u64(qmod, p - 2, p);    /* mpz_invert (p prime) */
    big_mul_u64(&c->lifting[cm], &quot, inv);     /* mpz_mul */
  }
}

nfl_oracle_ctx *nfl_oracle_create(int limb_bits, size_t degree, size_t nmoduli, const void *P,
                                  const void *Pn, const void *roots, const void *invkmax,
                                  int kmax_log2) {
  if (degree == 0 || (degree & (degree - 1)) || nmoduli == 0) return NULL;
  if (degree > ((size_t)1 << kmax_log2)) return NULL; /* core.hpp:59-60 */
  if (((limb_bits == 64 ? 62 : limb_bits - 2) * nmoduli + 64) / 64 + 4 > BIG_MAX) return NULL;
  nfl_oracle_ctx *c = (nfl_oracle_ctx *)calloc(1, sizeof(*c));
  c->limb_bits = limb_bits;
  c->n = degree;
  c->nm = nmoduli;
  switch (limb_bits) {
    case 16: c->tabs = u16_tabs_create(degree, nmoduli, P, Pn, roots, invkmax, kmax_log2); break;
    case 32: c->tabs = u32_tabs_create(degree, nmoduli, P, Pn, roots, invkmax, kmax_log2); break;
    case 64: c->tabs = u64_tabs_create(degree, nmoduli, P, Pn, roots, invkmax, kmax_log2); break;
    default: free(c); return NULL;
  }
  crt_init(c);
  return c;
}

void nfl_oracle_destroy(nfl_oracle_ctx *c) {
  if (!c) return;
  switch (c->limb_bits) {
    case 16: u16_tabs_destroy((u16_tabs *)c->tabs); break;
    case 32: u32_tabs_destroy((u32_tabs *)c->tabs); break;
    default: u64_tabs_destroy((u64_tabs *)c->tabs); break;
  }
  free(c->lifting);
  free(c);
}

#define DISPATCH(c, call16, call32, call64) \
  do {                                      \
    switch ((c)->limb_bits) {               \
      case 16: call16; break;               \
      case 32: call32; break;               \
      default: call64; break;               \
    }                                       \
  } while (0)

#define TABLE_BODY(PFX)                                                           \
  {                                                                               \
    const PFX##_tabs *t = (const PFX##_tabs *)c->tabs;                            \
    switch (which) {                                                              \
      case NFL_ORACLE_TAB_PHIS: return t->phis + cm * t->n;                       \
      case NFL_ORACLE_TAB_SHOUPPHIS: return t->shoupphis + cm * t->n;             \
      case NFL_ORACLE_TAB_INVPOLY_INVPHIS: return t->ipip + cm * t->n;            \
      case NFL_ORACLE_TAB_SHOUPINVPOLY_INVPHIS: return t->shoupipip + cm * t->n;  \
      case NFL_ORACLE_TAB_OMEGAS: return t->omegas + cm * 2 * t->n;               \
      case NFL_ORACLE_TAB_INVOMEGAS: return t->invomegas + cm * 2 * t->n;         \
      case NFL_ORACLE_TAB_INVPOLYDEGREE: return t->invdeg + cm;                   \
      default: return NULL;                                                       \
    }                                                                             \
  }

const void *nfl_oracle_table(const nfl_oracle_ctx *c, int which, size_t cm) {
  if (cm >= c->nm) return NULL;
  switch (c->limb_bits) {
    case 16: TABLE_BODY(u16)
    case 32: TABLE_BODY(u32)
    default: TABLE_BODY(u64)
  }
}

void nfl_oracle_ntt_pow_phi(const nfl_oracle_ctx *c, void *data, size_t batch) {
  DISPATCH(c, u16_ntt_pow_phi((u16_tabs *)c->tabs, (uint16_t *)data, batch),
           u32_ntt_pow_phi((u32_tabs *)c->tabs, (uint32_t *)data, batch),
           u64_ntt_pow_phi((u64_tabs *)c->tabs, (uint64_t *)data, batch));
}

void nfl_oracle_invntt_pow_invphi(const nfl_oracle_ctx *c, void *data, size_t batch) {
  DISPATCH(c, u16_invntt_pow_invphi((u16_tabs *)c->tabs, (uint16_t *)data, batch),
           u32_invntt_pow_invphi((u32_tabs *)c->tabs, (uint32_t *)data, batch),
           u64_invntt_pow_invphi((u64_tabs *)c->tabs, (uint64_t *)data, batch));
}

#define ROW_BODY(PFX, TY)                                                               \
  {                                                                                     \
    const PFX##_tabs *t = (const PFX##_tabs *)c->tabs;                                  \
    const TY *w = (inverse_tables ? t->invomegas : t->omegas) + cm * 2 * t->n;          \
    PFX##_ntt(t, (TY *)row, w, w + t->n, t->P[cm]);                                     \
  }

void nfl_oracle_ntt_row(const nfl_oracle_ctx *c, void *row, size_t cm, int inverse_tables) {
  DISPATCH(c, ROW_BODY(u16, uint16_t), ROW_BODY(u32, uint32_t), ROW_BODY(u64, uint64_t));
}

void nfl_oracle_pointwise(const nfl_oracle_ctx *c, int op, void *out, const void *a, const void *b,
                          const void *bp, size_t batch) {
  DISPATCH(c,
           u16_pointwise((u16_tabs *)c->tabs, op, (uint16_t *)out, (const uint16_t *)a, (const uint16_t *)b,
                         (const uint16_t *)bp, batch),
           u32_pointwise((u32_tabs *)c->tabs, op, (uint32_t *)out, (const uint32_t *)a, (const uint32_t *)b,
                         (const uint32_t *)bp, batch),
           u64_pointwise((u64_tabs *)c->tabs, op, (uint64_t *)out, (const uint64_t *)a, (const uint64_t *)b,
                         (const uint64_t *)bp, batch));
}

void nfl_oracle_polymul(const nfl_oracle_ctx *c, void *out, const void *a, const void *b, size_t batch) {
  const size_t bytes = batch * c->nm * c->n * (size_t)(c->limb_bits / 8);
  void *ta = malloc(bytes), *tb = malloc(bytes);
  memcpy(ta, a, bytes);
  memcpy(tb, b, bytes);
  nfl_oracle_ntt_pow_phi(c, ta, batch);
  nfl_oracle_ntt_pow_phi(c, tb, batch);
  nfl_oracle_pointwise(c, NFL_ORACLE_MUL, out, ta, tb, NULL, batch);
  nfl_oracle_invntt_pow_invphi(c, out, batch);
  free(ta);
  free(tb);
}

/* the same polymul over a batch split across host threads (polys are independent); used only by
 * bench.py's socket-level CPU figure */
typedef struct {
  const nfl_oracle_ctx *c;
  char *out;
  const char *a, *b;
  size_t batch;
} mt_job;
static void *mt_worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  if (j->batch) nfl_oracle_polymul(j->c, j->out, j->a, j->b, j->batch);
  return NULL;
}
void nfl_oracle_polymul_mt(const nfl_oracle_ctx *c, void *out, const void *a, const void *b, size_t batch,
                           int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > batch) nthreads = (int)(batch ? batch : 1);
  const size_t pbytes = c->nm * c->n * (size_t)(c->limb_bits / 8);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)nthreads);
  size_t start = 0;
  for (int t = 0; t < nthreads; t++) {
    size_t cnt = batch / (size_t)nthreads + ((size_t)t < batch % (size_t)nthreads ? 1 : 0);
    jobs[t].c = c;
    jobs[t].out = (char *)out + start * pbytes;
    jobs[t].a = (const char *)a + start * pbytes;
    jobs[t].b = (const char *)b + start * pbytes;
    jobs[t].batch = cnt;
    start += cnt;
    pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

int nfl_oracle_any_eq(const nfl_oracle_ctx *c, const void *a, const void *b, size_t batch) {
  switch (c->limb_bits) {
    case 16: return u16_any_cmp((u16_tabs *)c->tabs, (const uint16_t *)a, (const uint16_t *)b, batch, 1);
    case 32: return u32_any_cmp((u32_tabs *)c->tabs, (const uint32_t *)a, (const uint32_t *)b, batch, 1);
    default: return u64_any_cmp((u64_tabs *)c->tabs, (const uint64_t *)a, (const uint64_t *)b, batch, 1);
  }
}
int nfl_oracle_any_neq(const nfl_oracle_ctx *c, const void *a, const void *b, size_t batch) {
  switch (c->limb_bits) {
    case 16: return u16_any_cmp((u16_tabs *)c->tabs, (const uint16_t *)a, (const uint16_t *)b, batch, 0);
    case 32: return u32_any_cmp((u32_tabs *)c->tabs, (const uint32_t *)a, (const uint32_t *)b, batch, 0);
    default: return u64_any_cmp((u64_tabs *)c->tabs, (const uint64_t *)a, (const uint64_t *)b, batch, 0);
  }
}

size_t nfl_oracle_crt_limbs(const nfl_oracle_ctx *c) { return c->L; }
size_t nfl_oracle_crt_bits(const nfl_oracle_ctx *c) { return c->bitsQ; }
size_t nfl_oracle_crt_shift(const nfl_oracle_ctx *c) { return c->shift; }

static size_t big_export(const big *a, uint64_t *out, size_t cap) {
  size_t k = a->n < cap ? a->n : cap;
  memset(out, 0, cap * sizeof(uint64_t));
  memcpy(out, a->v, k * sizeof(uint64_t));
  return a->n;
}
size_t nfl_oracle_crt_modulus(const nfl_oracle_ctx *c, uint64_t *out, size_t cap) { return big_export(&c->Q, out, cap); }
size_t nfl_oracle_crt_modulus_shoup(const nfl_oracle_ctx *c, uint64_t *out, size_t cap) {
  return big_export(&c->mshoup, out, cap);
}
size_t nfl_oracle_crt_lifting(const nfl_oracle_ctx *c, size_t cm, uint64_t *out, size_t cap) {
  return big_export(&c->lifting[cm], out, cap);
}

static uint64_t ctx_coeff(const nfl_oracle_ctx *c, const void *data, size_t idx) {
  switch (c->limb_bits) {
    case 16: return ((const uint16_t *)data)[idx];
    case 32: return ((const uint32_t *)data)[idx];
    default: return ((const uint64_t *)data)[idx];
  }
}

/* GMP::poly2mpz (gmp.hpp:183-209) */
void nfl_oracle_crt_lift(const nfl_oracle_ctx *c, uint64_t *out, const void *data, size_t batch) {
  const size_t n = c->n, nm = c->nm, L = c->L;
  big acc, tmp, tq;
  for (size_t b = 0; b < batch; b++)
    for (size_t i = 0; i < n; i++) {
      memset(&acc, 0, sizeof(acc));
      for (size_t cm = 0; cm < nm; cm++) {
        uint64_t x = ctx_coeff(c, data, (b * nm + cm) * n + i);
        if (x != 0) big_addmul_u64(&acc, &c->lifting[cm], x); /* gmp.hpp:192-196 */
      }
      big_mul(&tmp, &acc, &c->mshoup);     /* mpz_mul            gmp.hpp:199 */
      big_shr(&tmp, &tmp, c->shift);       /* mpz_tdiv_q_2exp    gmp.hpp:200 */
      big_mul(&tq, &tmp, &c->Q);           /* mpz_submul         gmp.hpp:201 */
      big_sub(&acc, &tq);
      if (big_cmp(&acc, &c->Q) >= 0) big_sub(&acc, &c->Q); /* gmp.hpp:202-204 */
      uint64_t *o = out + (b * n + i) * L;
      memset(o, 0, L * sizeof(uint64_t));
      memcpy(o, acc.v, (acc.n < L ? acc.n : L) * sizeof(uint64_t));
    }
}

/* GMP::mpz2poly (gmp.hpp:211-219): rop(cm,i) = X[i] mod p_cm (mpz_fdiv_ui, X >= 0) */
void nfl_oracle_crt_project(const nfl_oracle_ctx *c, void *data, const uint64_t *limbs, size_t L_in,
                            size_t batch) {
  const size_t n = c->n, nm = c->nm;
  for (size_t b = 0; b < batch; b++)
    for (size_t cm = 0; cm < nm; cm++)
      for (size_t i = 0; i < n; i++) {
        const uint64_t *x = limbs + (b * n + i) * L_in;
        const size_t idx = (b * nm + cm) * n + i;
        switch (c->limb_bits) {
          case 16: ((uint16_t *)data)[idx] = u16_limbs_mod(x, L_in, ((u16_tabs *)c->tabs)->P[cm]); break;
          case 32: ((uint32_t *)data)[idx] = u32_limbs_mod(x, L_in, ((u32_tabs *)c->tabs)->P[cm]); break;
          default: ((uint64_t *)data)[idx] = u64_limbs_mod(x, L_in, ((u64_tabs *)c->tabs)->P[cm]); break;
        }
      }
}

void nfl_oracle_fill_uniform(const nfl_oracle_ctx *c, void *data, size_t first_poly, size_t batch,
                             uint64_t seed, int operand) {
  DISPATCH(c, u16_fill_uniform((u16_tabs *)c->tabs, (uint16_t *)data, first_poly, batch, seed, operand),
           u32_fill_uniform((u32_tabs *)c->tabs, (uint32_t *)data, first_poly, batch, seed, operand),
           u64_fill_uniform((u64_tabs *)c->tabs, (uint64_t *)data, first_poly, batch, seed, operand));
}
