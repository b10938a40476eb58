/* nfl_oracle.h -- CPU oracle for the NFLlib NTT polynomial-ring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (quarkslab/NFLlib @ v1) used as the parity checker for the HIP
 * engine and as the `cpu_baseline` leg of bench.py.  Nothing in the product
 * path (libnflhip.so, nfllib_amd/, include/) may call into it.
 *
 * Parity status: PINNED.  The oracle is checked bit-for-bit (memcmp) against
 * the real reference compiled in the build container (oracle/_ref, built by
 * oracle/Makefile from /root/reference's own sources) by
 * tests/test_oracle_vs_ref.py, and against the golden fixtures under
 * tests/golden/ that were generated from that same real reference by
 * tools/gen_golden.py.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef NFL_ORACLE_H
#define NFL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nfl_oracle_ctx nfl_oracle_ctx;

/* element-wise operations understood by nfl_oracle_pointwise */
enum {
  NFL_ORACLE_ADD = 0,          /* ops::addmod        ops.hpp:124-135 */
  NFL_ORACLE_SUB = 1,          /* ops::submod        ops.hpp:141-151 */
  NFL_ORACLE_MUL = 2,          /* ops::mulmod        ops.hpp:183-219 */
  NFL_ORACLE_MUL_SHOUP = 3,    /* ops::mulmod_shoup  ops.hpp:225-242 */
  NFL_ORACLE_COMPUTE_SHOUP = 4 /* ops::compute_shoup ops.hpp:165-177 */
};

/* which table nfl_oracle_table returns (poly.hpp:228-237) */
enum {
  NFL_ORACLE_TAB_PHIS = 0,
  NFL_ORACLE_TAB_SHOUPPHIS = 1,
  NFL_ORACLE_TAB_INVPOLY_INVPHIS = 2,
  NFL_ORACLE_TAB_SHOUPINVPOLY_INVPHIS = 3,
  NFL_ORACLE_TAB_OMEGAS = 4,    /* 2*degree words: powers then Shoup companions */
  NFL_ORACLE_TAB_INVOMEGAS = 5, /* 2*degree words */
  NFL_ORACLE_TAB_INVPOLYDEGREE = 6 /* 1 word */
};

/* limb_bits in {16,32,64}.  P/Pn/roots/invkmax point at arrays of nmoduli
 * words of that width (params.hpp).  kmax_log2 = log2(kMaxPolyDegree).
 * Returns NULL on invalid arguments. */
nfl_oracle_ctx *nfl_oracle_create(int limb_bits, size_t degree, size_t nmoduli,
                                  const void *P, const void *Pn, const void *roots,
                                  const void *invkmax, int kmax_log2);
void nfl_oracle_destroy(nfl_oracle_ctx *ctx);

const void *nfl_oracle_table(const nfl_oracle_ctx *ctx, int which, size_t cm);

/* data: dense [batch][nmoduli][degree] words, in place. */
void nfl_oracle_ntt_pow_phi(const nfl_oracle_ctx *ctx, void *data, size_t batch);       /* core.hpp:594-600 */
void nfl_oracle_invntt_pow_invphi(const nfl_oracle_ctx *ctx, void *data, size_t batch); /* core.hpp:608-614 */
/* one row, cyclic transform only (core::ntt, core.hpp:455-532), modulus cm */
void nfl_oracle_ntt_row(const nfl_oracle_ctx *ctx, void *row, size_t cm, int inverse_tables);

void nfl_oracle_pointwise(const nfl_oracle_ctx *ctx, int op, void *out, const void *a,
                          const void *b, const void *bprime, size_t batch);
/* a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b; c.invntt_pow_invphi()  (poly.hpp:167-168,350) */
void nfl_oracle_polymul(const nfl_oracle_ctx *ctx, void *c, const void *a, const void *b, size_t batch);
/* nfl_oracle_polymul with the batch split over `nthreads` host threads (bench.py's socket-level figure) */
void nfl_oracle_polymul_mt(const nfl_oracle_ctx *ctx, void *c, const void *a, const void *b, size_t batch, int nthreads);
/* expr::operator bool over eqmod / neqmod (ops.hpp:81-117): "any lane" semantics */
int nfl_oracle_any_eq(const nfl_oracle_ctx *ctx, const void *a, const void *b, size_t batch);
int nfl_oracle_any_neq(const nfl_oracle_ctx *ctx, const void *a, const void *b, size_t batch);

/* CRT (gmp.hpp:113-219), restated on little-endian 64-bit limb vectors. */
size_t nfl_oracle_crt_limbs(const nfl_oracle_ctx *ctx);        /* L = ceil(bits(Q)/64) */
size_t nfl_oracle_crt_bits(const nfl_oracle_ctx *ctx);         /* bits_in_moduli_product */
size_t nfl_oracle_crt_shift(const nfl_oracle_ctx *ctx);        /* shift_modulus_shoup */
/* constants, little-endian limbs; return number of limbs written (<= cap) */
size_t nfl_oracle_crt_modulus(const nfl_oracle_ctx *ctx, uint64_t *out, size_t cap);
size_t nfl_oracle_crt_modulus_shoup(const nfl_oracle_ctx *ctx, uint64_t *out, size_t cap);
size_t nfl_oracle_crt_lifting(const nfl_oracle_ctx *ctx, size_t cm, uint64_t *out, size_t cap);
/* lift: out[batch][degree][L] little-endian limbs of X in [0,Q)  (gmp.hpp:183-209) */
void nfl_oracle_crt_lift(const nfl_oracle_ctx *ctx, uint64_t *out, const void *data, size_t batch);
/* project: data(cm,i) = X[i] mod p_cm, X given as L_in limbs (non-negative)  (gmp.hpp:211-219) */
void nfl_oracle_crt_project(const nfl_oracle_ctx *ctx, void *data, const uint64_t *limbs,
                            size_t L_in, size_t batch);

/* Seeded synthetic inputs; same mask-then-subtract
 * rule as nfl::uniform (core.hpp:165-176) on a counter-based splitmix64. */
void nfl_oracle_fill_uniform(const nfl_oracle_ctx *ctx, void *data, size_t first_poly,
                             size_t batch, uint64_t seed, int operand);

#ifdef __cplusplus
}
#endif
#endif
