/* nflhip.h -- C ABI of the MI355X-native NTT polynomial-ring engine.
 *
 * This is the drop-in boundary for the ONE hot path of quarkslab/NFLlib that
 * this project accelerates: whole-polynomial / whole-batch
 * operations on the dense modulus-major coefficient block of
 * nfl::poly<T, Degree, NbModuli> (reference include/nfl/poly.hpp:82-88:
 * `T _data[NbModuli*Degree]`, element (cm,i) at `_data[cm*Degree+i]`).  A batch
 * is an array of polys, i.e. a dense [batch][NbModuli][Degree] tensor (how the
 * reference's tests allocate: tests/tools.h:6-17).
 *
 * Every entry point names the reference interface it replaces (file:line under
 * /root/reference).  All results are bit-identical to the reference's on the
 * same inputs.  Plain C types only; no exceptions cross this boundary: every
 * call returns an int status (0 = NFLHIP_OK) and nflhip_last_error() gives the
 * text.  The library is implemented in hand-written HIP for gfx950 and there
 * is NO CPU fallback: without a usable GPU every compute call fails with
 * NFLHIP_ERR_NO_DEVICE.
 *
 * Environment (all of it; every variable is read ONCE, when a context / communicator is created)
 *   NFLHIP_VARIANT=hipcc   the context serves every call with the compiled (hipcc) kernels instead of the generated
 *                          gfx950 assembly kernels: the independent cross-check the tests use (bit-identical results)
 *   NFLHIP_XCD=0|1         rows of 32768 / 65536 words: never / always the one-launch plan of persistent workgroups
 *                          (default: by batch size)
 *   NFLHIP_COMM_PIECE_BYTES  upper bound of one RCCL message of nflhip_scatter_dev / nflhip_gather_dev (default 1 GiB)
 * Test hooks are entry points of their own (include/nflhip_debug.h), not environment variables.
 *
 * Pointer conventions
 *   *_dev entry points take DEVICE pointers and a hipStream_t (passed as
 *   void*; NULL = the null stream) and are asynchronous on that stream.
 *   The un-suffixed entry points take HOST pointers, stage through
 *   context-owned device buffers and return after the result is in host
 *   memory (the per-poly calls of the header-only nfl::poly surface use them).
 *   `out` may alias any input of an element-wise op (the reference's
 *   evaluation loop is strictly element-wise: core.hpp:24-37).
 */
#ifndef NFLHIP_H
#define NFLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFLHIP_ABI_VERSION 6

typedef struct nflhip_ctx nflhip_ctx;

/* status codes */
enum {
  NFLHIP_OK = 0,
  NFLHIP_ERR_INVALID = 1,    /* bad argument (std::runtime_error / static_assert in the reference) */
  NFLHIP_ERR_NO_DEVICE = 2,  /* no usable HIP device: there is no CPU fallback */
  NFLHIP_ERR_HIP = 3,        /* a HIP runtime call failed */
  NFLHIP_ERR_UNSUPPORTED = 4,/* shape outside what the engine implements */
  NFLHIP_ERR_NOMEM = 5
};

/* element-wise operations (the functors poly::operator=(expr) evaluates) */
enum {
  NFLHIP_OP_ADD = 0,           /* ops::addmod        ops.hpp:124-135; operator+  poly.hpp:347 */
  NFLHIP_OP_SUB = 1,           /* ops::submod        ops.hpp:141-151; operator-  poly.hpp:346 */
  NFLHIP_OP_MUL = 2,           /* ops::mulmod        ops.hpp:183-219; operator*  poly.hpp:350 */
  NFLHIP_OP_MUL_SHOUP = 3,     /* ops::mulmod_shoup  ops.hpp:225-242; shoup(a*b,b') ops.hpp:267-277 */
  NFLHIP_OP_COMPUTE_SHOUP = 4  /* ops::compute_shoup ops.hpp:165-177; poly.hpp:352 */
};

/* tables readable through nflhip_get_table (reference-format views, for parity tests) */
enum {
  NFLHIP_TAB_PSI = 0,       /* engine layout: n pairs (psi^bitrev(k), Shoup companion), k=0..n-1 */
  NFLHIP_TAB_MODULUS = 1,   /* 1 word: params<T>::P[cm]                       params.hpp:21,55,97 */
  NFLHIP_TAB_INVDEGREE = 2, /* 1 word: core::invpolyDegree[cm]                core.hpp:664-665 */
  /* the reference's OWN table layouts (poly.hpp:228-237), rebuilt on the host from the same roots: what a caller that
   * reads core::base through the tests::poly_tests_proxy friend sees (tests/ntt_perfs.cpp:132-133).  No device needed. */
  NFLHIP_TAB_PHIS = 3,                 /* degree words: phi^i                          core.hpp:649-656 */
  NFLHIP_TAB_SHOUPPHIS = 4,            /* degree words: their Shoup companions */
  NFLHIP_TAB_INVPOLY_INVPHIS = 5,      /* degree words: n^-1 * phi^-i                  core.hpp:664-676 */
  NFLHIP_TAB_SHOUPINVPOLY_INVPHIS = 6, /* degree words */
  NFLHIP_TAB_OMEGAS = 7,               /* 2*degree words: stage-concatenated powers of omega = phi^2 (core::prep_wtab,
                                          core.hpp:564-581), then their Shoup companions at offset degree (shoupomegas) */
  NFLHIP_TAB_INVOMEGAS = 8             /* 2*degree words: the same for omega^-1 (invomegas / shoupinvomegas) */
};

int nflhip_abi_version(void);
/* Text of the last error any nflhip call raised ON THE CALLING THREAD (errno
 * style, so host threads sharing one context never race on it; ctx may be NULL,
 * e.g. after a failed nflhip_ctx_create). Never NULL. */
const char *nflhip_last_error(const nflhip_ctx *ctx);
int nflhip_device_count(int *count);

/* ---- context: replaces the static tables of poly::core / poly::GMP ----------
 * core::initialize (core.hpp:625-686), core::prep_wtab (core.hpp:564-581) and
 * GMP::GMP (gmp.hpp:113-155).  Built lazily by the caller (never at static-init
 * time, unlike poly.hpp:247).  limb_bits in {16,32,64}; P / primitive_roots /
 * invkmax point at nmoduli words of that width taken from params<T>
 * (params.hpp:21-36, 55-76, 97-113); kmax_log2 = log2(params<T>::kMaxPolyDegree).
 * The context is immutable after creation: entry points may be called from
 * several host threads with distinct streams.  One context per device.
 * The *_dev entry points only enqueue work on the caller's stream (no host
 * synchronisation, no allocation after the first call of a given size), so a
 * sequence of them can be captured into a hipGraph and replayed; the exceptions
 * are nflhip_any_eq_dev / nflhip_any_neq_dev, which return a host value. */
int nflhip_ctx_create(nflhip_ctx **out, int device, int limb_bits, size_t degree, size_t nmoduli,
                      const void *P, const void *primitive_roots, const void *invkmax, int kmax_log2);
int nflhip_ctx_destroy(nflhip_ctx *ctx);

size_t nflhip_degree(const nflhip_ctx *ctx);
size_t nflhip_nmoduli(const nflhip_ctx *ctx);
int nflhip_limb_bits(const nflhip_ctx *ctx);
/* L = ceil(bits(prod p_cm)/64): 64-bit limbs per lifted coefficient (gmp.hpp:121-122) */
size_t nflhip_crt_limbs(const nflhip_ctx *ctx);
int nflhip_get_table(const nflhip_ctx *ctx, int which, size_t cm, void *host_out, size_t host_bytes);
/* CRT constants as little-endian 64-bit limbs; what: 0 = moduli_product, 1 = lifting_integers[cm]
 * (gmp.hpp:115-151). Returns the number of significant limbs through *nlimbs. */
int nflhip_get_crt_constant(const nflhip_ctx *ctx, int what, size_t cm, uint64_t *host_out, size_t cap,
                            size_t *nlimbs);

/* ---- transforms ------------------------------------------------------------
 * poly::ntt_pow_phi()       poly.hpp:167 -> core::ntt_pow_phi        core.hpp:594-600
 * poly::invntt_pow_invphi() poly.hpp:168 -> core::invntt_pow_invphi  core.hpp:608-614
 * In place on [batch][nmoduli][degree]; forward output is in the reference's
 * order (out[cm][bitrev(k)] = a_cm(phi^(2k+1))) and every output word is the
 * canonical representative in [0,p) (core.hpp:523-529, ops.hpp:240). */
int nflhip_ntt_fwd_dev(nflhip_ctx *ctx, void *d_data, size_t batch, void *stream);
int nflhip_ntt_inv_dev(nflhip_ctx *ctx, void *d_data, size_t batch, void *stream);
int nflhip_ntt_fwd(nflhip_ctx *ctx, void *h_data, size_t batch);
int nflhip_ntt_inv(nflhip_ctx *ctx, void *h_data, size_t batch);

/* ---- the cyclic transform of single rows ------------------------------------------
 * core::ntt(x, wtab, winvtab, p)   core.hpp:455-532 (Harvey DIF, algos.hpp:47-73): in-place CYCLIC transform of
 * `rows` contiguous rows of `degree` words, all of modulus `cm`: natural order in, bit-reversed order out,
 * out[bitrev(k)] = sum_j x[j] w^(jk), every word in [0,p).  mode bit NFLHIP_ROW_INVERSE_TABLES selects
 * w = omega^-1 (the reference's invomegas tables) instead of omega = phi^2; bit NFLHIP_ROW_BITREV_IO bit-reverses
 * the row before and after, i.e. core::inv_ntt (core.hpp:539-557) when combined with the inverse tables.
 * This is what tests/ntt_perfs.cpp:155-171 times through the poly_tests_proxy friend (BASELINE configs[0]). */
#define NFLHIP_ROW_INVERSE_TABLES 1
#define NFLHIP_ROW_BITREV_IO 2
int nflhip_ntt_row_dev(nflhip_ctx *ctx, void *d_rows, size_t cm, int mode, size_t rows, void *stream);
int nflhip_ntt_row(nflhip_ctx *ctx, void *h_rows, size_t cm, int mode, size_t rows);

/* ---- element-wise ops: poly::operator=(expr) core.hpp:24-37 ------------------
 * op in NFLHIP_OP_*; b is ignored for COMPUTE_SHOUP, bprime only used by
 * MUL_SHOUP.  Input contract as the reference's (operands < p; ops.hpp:131,148,211). */
int nflhip_pointwise_dev(nflhip_ctx *ctx, int op, void *d_out, const void *d_a, const void *d_b,
                         const void *d_bprime, size_t batch, void *stream);
int nflhip_pointwise(nflhip_ctx *ctx, int op, void *h_out, const void *h_a, const void *h_b,
                     const void *h_bprime, size_t batch);

/* ---- fused expression trees -------------------------------------------------------
 * poly::operator=(expr) evaluates a whole expression tree (e.g. `a + b*add`, tests/poly_p.cpp:62-66;
 * `resb - resa*s`, tests/nfllib_demo_main_op.cpp:51) in one pass without temporaries
 * (core.hpp:24-37, ops.hpp:52-79).  Here the tree is a postfix program executed per word on the
 * device: byte k < 8 pushes operand k; NFLHIP_EXPR_ADD/SUB/MUL pop two values and push the result
 * (`x op y` with y on top); NFLHIP_EXPR_MUL_SHOUP pops b', b, a and pushes mulmod_shoup(a,b,b');
 * NFLHIP_EXPR_COMPUTE_SHOUP replaces the top.  At most 8 operands, 24 program bytes, stack depth 4;
 * exactly one value must remain.  `out` may alias any operand.  The host-pointer variant stages at
 * most 4 distinct operands (4: `c = c + shoup(a * b, b')`, the result is formed over the first one's staging buffer). */
#define NFLHIP_EXPR_ADD 0x10
#define NFLHIP_EXPR_SUB 0x11
#define NFLHIP_EXPR_MUL 0x12
#define NFLHIP_EXPR_MUL_SHOUP 0x13
#define NFLHIP_EXPR_COMPUTE_SHOUP 0x14
#define NFLHIP_EXPR_MAX_OPERANDS 8
#define NFLHIP_EXPR_MAX_LEN 24
int nflhip_eval_dev(nflhip_ctx *ctx, void *d_out, const void *const *d_operands, size_t noperands,
                    const unsigned char *program, size_t proglen, size_t batch, void *stream);
int nflhip_eval(nflhip_ctx *ctx, void *h_out, const void *const *h_operands, size_t noperands,
                const unsigned char *program, size_t proglen, size_t batch);
/* The same over operands that advance by their own stride (in polynomials) from one batch element to the next:
 * 1 = a dense array of polynomials, 0 = ONE polynomial shared by the whole batch (a key), k = every k-th polynomial of
 * an interleaved array.  Element i reads operand j at d_operands[j] + i*strides[j] polynomials and writes
 * d_out + i*out_stride polynomials (out_stride >= 1).  What the header's deferred per-polynomial operations are
 * coalesced into (one launch for a loop of `r[i] = u[i] * key + e[i]`). */
int nflhip_eval_strided_dev(nflhip_ctx *ctx, void *d_out, size_t out_stride, const void *const *d_operands,
                            const size_t *strides, size_t noperands, const unsigned char *program, size_t proglen,
                            size_t batch, void *stream);

/* ---- the metric path ---------------------------------------------------------
 * c = INTT( NTT(a) (.) NTT(b) ): the reference sequence
 *   a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b; c.invntt_pow_invphi();
 * (poly.hpp:167-168, 350) as one fused device pass.  a, b, c in coefficient
 * form; a and b are not modified; c may alias a or b.
 * Rows of up to 16384 words are ONE launch that touches nothing but its operands: calls on distinct streams run
 * concurrently.  Longer rows (32768 words: b' = NTT(b) into a scratch, then the product kernel; 65536 and beyond: streaming
 * passes around block products) go through ONE context-owned scratch area: calls on different streams of one context are
 * ordered one after the other by events (no host synchronisation); use one context per stream to overlap them. */
int nflhip_polymul_dev(nflhip_ctx *ctx, void *d_c, const void *d_a, const void *d_b, size_t batch,
                       void *stream);
int nflhip_polymul(nflhip_ctx *ctx, void *h_c, const void *h_a, const void *h_b, size_t batch);
/* "one operand pre-transformed": c = INTT( NTT(a) (.) b_ntt ), b_ntt already in
 * NTT form (keys kept in NTT form as in tests/nfllib_demo_main_op.cpp:26-46). */
int nflhip_polymul_ntt_dev(nflhip_ctx *ctx, void *d_c, const void *d_a, const void *d_bntt, size_t batch,
                           void *stream);

/* ---- transform-fused pipelines ------------------------------------------------------
 * What the reference's callers run AROUND the transforms, as one device pass per batch.  The reference's only
 * end-to-end ring code is the LWE demo (tests/nfllib_demo_main_op.cpp:26-58):
 *   encrypt():  u.ntt_pow_phi(); e1.ntt_pow_phi(); e2.ntt_pow_phi(); resa = u * pka + e1; resb = u * pkb + e2;
 *   decrypt():  tmp = resb - resa * s; tmp.invntt_pow_invphi();
 * (poly.hpp:167-168 + the evaluation loop core.hpp:24-37).  Run operator by operator that is 8 + 2 launches and ~17 + 5
 * polynomial passes over HBM; here the transformed operands never leave the registers of the workgroup that owns the row:
 *   nflhip_fwd_fma_dev    out  = NTT(x) * k + NTT(e)
 *   nflhip_fwd_fma2_dev   out0 = NTT(x) * k0 + NTT(e0),  out1 = NTT(x) * k1 + NTT(e1)       (x is transformed once)
 *   nflhip_fma_inv_dev    out  = INTT(b + a * k)   or   INTT(b - a * k)  (subtract != 0)
 * Results are bit-identical to the operator-by-operator sequence.  Every operand names its own stride (in polynomials
 * from one batch element to the next: 1 = dense array, 0 = ONE polynomial for the whole batch -- a key) and, for the
 * inputs of the forward entries, its format: NFLHIP_FMT_WORDS = residue words [nmoduli][degree] in coefficient form, or
 * one SIGNED integer per coefficient shared by all moduli (int8 / int16 / int32; v < 0 stands for p + v; |v| must be
 * below every modulus) -- what the samplers produce before they spread a value over the moduli (core.hpp:230-277); see
 * nflhip_sample_gauss_small_dev.  k operands and the operands of nflhip_fma_inv_dev are words in NTT form (canonical,
 * ops.hpp:131,211).  Results are dense; a result may alias a DENSE (stride 1) input of the same call
 * exactly (same first byte), or lie over one in any way when it is out0 over e1 / k1 (then a plan that reads every input before it
 * stores is taken); a result that overlaps an operand shared by the batch (stride 0, batch > 1) is refused with NFLHIP_ERR_INVALID.  u64 limbs at degree 4096,
 * 8192 and 16384 run one generated gfx950 kernel per call (at degree 32768 nflhip_fma_inv_dev does for dense a / b, and the forward
 * entries run three for int8 polynomials with keys of stride 0); rows of 1024 / 2048 words -- 4096 for 32-bit limbs -- run every
 * entry as ONE pass of the wave-per-row kernels (forward: inputs of one format, strides 0 / 1); every other shape composes the same
 * result from the plain kernels through the context's scratch (calls on different streams of one context are then ordered by
 * events, as for nflhip_polymul_dev). */
#define NFLHIP_FMT_WORDS 0
#define NFLHIP_FMT_I8 1
#define NFLHIP_FMT_I16 2
#define NFLHIP_FMT_I32 3
typedef struct nflhip_operand {
  const void *ptr; /* device pointer */
  size_t stride;   /* polynomials (of the operand's own format) between consecutive batch elements */
  int format;      /* NFLHIP_FMT_* */
} nflhip_operand;
int nflhip_fwd_fma_dev(nflhip_ctx *ctx, void *d_out, const nflhip_operand *x, const nflhip_operand *k,
                       const nflhip_operand *e, size_t batch, void *stream);
int nflhip_fwd_fma2_dev(nflhip_ctx *ctx, void *d_out0, void *d_out1, const nflhip_operand *x, const nflhip_operand *k0,
                        const nflhip_operand *e0, const nflhip_operand *k1, const nflhip_operand *e1, size_t batch,
                        void *stream);
int nflhip_fma_inv_dev(nflhip_ctx *ctx, void *d_out, const nflhip_operand *a, const nflhip_operand *k,
                       const nflhip_operand *b, int subtract, size_t batch, void *stream);
/* 1 when the three entries above run as one-pass kernels on this context (the default kernel variant; u64 limbs: ONE kernel each
 * at degree 1024 / 2048 -- the wave-per-row kernels -- and 4096 / 8192 / 16384 -- generated; at degree 32768 one kernel for
 * nflhip_fma_inv_dev and, for int8 polynomials with keys shared by the batch, one per noise polynomial + one for the results;
 * u32 limbs: ONE kernel each at degree 1024 / 2048 / 4096), 0 when they compose the result from the plain kernels: what a caller
 * that can choose between its own operator sequence and these entries asks (the header's deferred queue only rewrites sequences
 * when it gains a pass) */
int nflhip_has_fused_kernels(const nflhip_ctx *ctx);
/* compact polynomial(s) -> residue words: d_data[b][cm][i] = v < 0 ? p_cm + v : v, v = element i of src's polynomial
 * b * stride (format NFLHIP_FMT_WORDS: a strided gather of word rows) */
int nflhip_expand_small_dev(nflhip_ctx *ctx, void *d_data, const nflhip_operand *src, size_t batch, void *stream);

/* ---- comparisons: expr::operator bool over eqmod / neqmod (ops.hpp:81-117) ----
 * *result = 1 iff ANY word of a equals (any_eq) / differs from (any_neq) the
 * matching word of b -- the reference's semantics for `a == b` / `a != b`.
 * These synchronise the stream. */
int nflhip_any_eq_dev(nflhip_ctx *ctx, const void *d_a, const void *d_b, size_t batch, int *result,
                      void *stream);
int nflhip_any_neq_dev(nflhip_ctx *ctx, const void *d_a, const void *d_b, size_t batch, int *result,
                       void *stream);
int nflhip_any_eq(nflhip_ctx *ctx, const void *h_a, const void *h_b, size_t batch, int *result);
int nflhip_any_neq(nflhip_ctx *ctx, const void *h_a, const void *h_b, size_t batch, int *result);

/* ---- range assertion: CHECK_STRICTMOD ---------------------------------------------
 * The reference's tests are compiled with CHECK_STRICTMOD (tests/CMakeLists.txt:10): ASSERT_STRICTMOD (debug.hpp:33-37)
 * then asserts that every operand word is the canonical representative, x < p -- at the entry of the transforms
 * (core.hpp:457-462), in addmod / submod / mulmod / mulmod_shoup (ops.hpp:131,148,190,211,235) and after the samplers
 * (core.hpp:179-185, 275-281, 319-325).  *bad = 1 iff some word of the batch is >= its row's modulus.  The _dev form is one
 * streaming compare on the device and synchronises the stream; the host form reads host words against the context's
 * copy of the moduli (no device needed).  include/nfl_hip/nfl.hpp calls them under -DCHECK_STRICTMOD (without NDEBUG) on
 * the operands of transforms and expressions and throws std::runtime_error where the reference's assert would fire. */
int nflhip_check_range_dev(nflhip_ctx *ctx, const void *d_data, size_t batch, int *bad, void *stream);
int nflhip_check_range(const nflhip_ctx *ctx, const void *h_data, size_t batch, int *bad);

/* ---- CRT ---------------------------------------------------------------------
 * lift:    GMP::poly2mpz gmp.hpp:183-209 -- limbs[b][i][0..L) = little-endian
 *          64-bit limbs (the mpz_export(order=-1) image) of
 *          X_i = sum_cm lifting[cm]*x(cm,i) mod Q, in [0,Q).
 * project: GMP::mpz2poly gmp.hpp:211-219 / poly::set_mpz gmp.hpp:73-108 --
 *          x(cm,i) = X_i mod p_cm for non-negative X_i given as L_in limbs.
 * Any number of moduli.  With 62-bit moduli the lift runs on the matrix cores (an int8 GEMM over balanced base-256 digits,
 * nfllib_amd/csrc/kernels_crt_mfma.hip) for 21..32 of them, the projection for 17..32 and inputs of 5..64 limbs, when
 * batch * degree is a multiple of 64; register-resident VALU kernels up to 32 moduli otherwise, a limb-serial kernel
 * beyond.  The results do not depend on which kernel ran. */
int nflhip_crt_lift_dev(nflhip_ctx *ctx, uint64_t *d_limbs, const void *d_data, size_t batch, void *stream);
int nflhip_crt_project_dev(nflhip_ctx *ctx, void *d_data, const uint64_t *d_limbs, size_t L_in, size_t batch,
                           void *stream);
int nflhip_crt_lift(nflhip_ctx *ctx, uint64_t *h_limbs, const void *h_data, size_t batch);
int nflhip_crt_project(nflhip_ctx *ctx, void *h_data, const uint64_t *h_limbs, size_t L_in, size_t batch);

/* ---- harness helpers ---------------------------------------------------------
 * Seeded synthetic operands generated in place on the device: word (b,cm,i) of
 * operand s = splitmix64 counter stream masked to floor(log2 p)+1 bits with one
 * conditional subtract of p -- the distribution rule of nfl::uniform
 * (core.hpp:165-176).  first_poly offsets the counter so shards of one logical
 * batch can be generated independently on several GPUs. */
int nflhip_fill_uniform_dev(nflhip_ctx *ctx, void *d_data, size_t first_poly, size_t batch, uint64_t seed,
                            int operand, void *stream);

/* ---- samplers ------------------------------------------------------------------
 * Device versions of the reference's random constructors (core.hpp:146-391):
 *   poly(uniform) / poly(non_uniform(ub[,amp])) / poly(ZO_dist(rho)) / poly(hwt_dist(h)) /
 *   poly(gaussian(&fg_prng[,amp])).
 * The reference draws from a process-global Salsa20 stream keyed from /dev/urandom
 * (lib/prng/fastrandombytes.cpp:17-37); here every call names its stream: a 32-byte key
 * and a 64-bit stream id select a ChaCha20 keystream that is read by position (word w of
 * the stream belongs to one fixed coefficient), so a call is reproducible and a batch can
 * be generated in shards (first_poly) with the same result.  Callers that want the
 * reference's behaviour draw the key from the OS once and increment stream_id per call
 * (what include/nfl_hip/nfl.hpp does).  The map from random words to coefficients is the
 * reference's (mask-and-subtract, no rejection), so the distributions are identical. */
enum {
  NFLHIP_DIST_UNIFORM = 0, /* core.hpp:152-188   one word per residue word */
  NFLHIP_DIST_BOUNDED = 1, /* core.hpp:195-277   param0 = upper_bound, param1 = amplifier; one word per coefficient */
  NFLHIP_DIST_ZO = 2,      /* core.hpp:330-340   param0 = rho (0..255) */
  NFLHIP_DIST_HWT = 3,     /* core.hpp:347-391   param0 = hamming weight (1..degree) */
  /* OR into NFLHIP_DIST_ZO / NFLHIP_DIST_HWT: store +1 exactly as the reference does, as the non-canonical word
   * p + 1 (`pm + (rnd & 2)`, core.hpp:341,387), so that raw _data images diff clean against the CPU library.
   * Default (flag absent): the canonical 1, which the engine's own operators require (ops.hpp:131,148). */
  NFLHIP_DIST_REFERENCE_WORDS = 0x100,
  /* OR into NFLHIP_DIST_UNIFORM: the NARROW draw -- residue word g reads the limb-width LANE g of the keystream (bytes
   * [g w, (g + 1) w), little endian, w = limb bytes) instead of one 64-bit word, in a keystream domain of its own.  Same map
   * from the lane to the residue (mask, one conditional subtraction), so the distribution is the reference's -- and so is the
   * consumption: the reference fills _data with fastrandombytes and reduces every limb-width word in place (core.hpp:152-188),
   * i.e. residue word g is made from bytes [g w, (g + 1) w) of ITS stream (tests/test_samplers_cpu.py pins exactly that); a u16 / u32
   * polynomial costs a quarter / half of the ChaCha20 rounds (measured: profiles/r05_sampler_rates.txt).  The values differ
   * from the wide rule's for the same (key, stream_id) -- a different domain, by design: what the wide rule produces, and
   * every digest recorded from it, keeps its meaning. */
  NFLHIP_DIST_NARROW = 0x200
};
/* KEYSTREAM DISCIPLINE.  A (key, stream_id, distribution) triple names one keystream: two calls that share all
 * three produce the same values.  Different distributions never share keystream words even for the same
 * (key, stream_id) -- the distribution's tag is part of the ChaCha20 block counter -- so a public uniform polynomial
 * never reveals the noise drawn next to it; but two draws of the SAME distribution that must be independent (the
 * secret and the error of an LWE sample) need distinct stream ids.  Callers that want the reference's behaviour draw
 * the key from the OS once and take a fresh stream id per call (what include/nfl_hip/nfl.hpp does). */
int nflhip_sample_dev(nflhip_ctx *ctx, void *d_data, size_t first_poly, size_t batch, int dist, uint64_t param0,
                      uint64_t param1, const unsigned char key[32], uint64_t stream_id, void *stream);
int nflhip_sample(nflhip_ctx *ctx, void *h_data, size_t batch, int dist, uint64_t param0, uint64_t param1,
                  const unsigned char key[32], uint64_t stream_id);
/* SEQUENCE forms: polynomial b of the dense batch at d is exactly what the one-polynomial call
 * nflhip_sample_dev(ctx, d_b, 0, 1, dist, param0, param1, key, first_stream_id + b*stream_id_stride, stream) produces
 * (one keystream per polynomial instead of one keystream per batch) -- so a loop of per-polynomial constructor calls
 * and ONE batched launch give the same polynomials.  Needs degree >= 8 (NFLHIP_ERR_UNSUPPORTED otherwise). */
int nflhip_sample_seq_dev(nflhip_ctx *ctx, void *d_data, size_t batch, int dist, uint64_t param0, uint64_t param1,
                          const unsigned char key[32], uint64_t first_stream_id, uint64_t stream_id_stride, void *stream);
/* raw keystream words [first_word, first_word + nwords) of (key, stream_id) -- what the samplers consume */
int nflhip_random_words_dev(nflhip_ctx *ctx, uint64_t *d_out, uint64_t first_word, size_t nwords,
                            const unsigned char key[32], uint64_t stream_id, void *stream);

/* Discrete Gaussian (FastGaussianNoise<in,out,lu>(sigma, security, samples, center),
 * FastGaussianNoise.hpp:163-290): cumulative table with the reference's tail bound and bit precision
 * (<= 384 bits: security parameters up to ~ 360), sampled by inversion.  The table lives on the context's device.
 * Keystream use: the top 64 bits of coefficient g's uniform number are word g of stream (key, stream_id)
 * (g = (first_poly + poly) * degree + i); its lower words -- words (W-1)*g .. of the same stream counted from
 * block 2^63 -- are only generated when the top word ties with a table entry; the sample is exactly the
 * full-precision inversion x = x_min + #{k : table[k] <= r}. */
typedef struct nflhip_gauss nflhip_gauss;
int nflhip_gauss_create(nflhip_ctx *ctx, nflhip_gauss **out, double sigma, unsigned security, unsigned samples,
                        double center);
int nflhip_gauss_destroy(nflhip_ctx *ctx, nflhip_gauss *g);
/* DRAW WIDTH of every sampler that takes this table: 64 (default, the rule above) or 32 -- the narrow draw: the top 32 bits
 * of coefficient g's uniform number are the 32-bit lane g of the stream (key, stream_id) in the Gaussian samplers' narrow
 * domain, the next 32 bits lane g of a second domain that is only generated when the first lane ties with the top half of
 * a table entry the search meets (~entries x 2^-32 per sample), the lower words as above.  The sample is still EXACTLY the
 * full-precision inversion of that number; it costs half the ChaCha20 rounds and 32-bit compares.  Values differ from the
 * 64-bit draw's for the same (key, stream_id): other keystream domains.  Sequence forms then need degree >= 16.  Set it
 * before the table is used by concurrent calls (it is a plain field). */
int nflhip_gauss_set_draw_bits(nflhip_gauss *g, int bits);
int nflhip_gauss_draw_bits(const nflhip_gauss *g);
/* the same table without a device (host arithmetic only: what nflhip_gauss_create uploads); h_table may be NULL to
 * query the sizes first, cap_words = capacity of h_table in 64-bit words */
int nflhip_gauss_table(double sigma, unsigned security, unsigned samples, double center, long long *x_min, size_t *entries,
                       int *words, unsigned *bit_precision, double *tail, uint64_t *h_table, size_t cap_words);
/* introspection: support [x_min, x_min + entries), words per entry, the reference's bit_precision and tail bound;
 * h_table (may be NULL) receives entries*words 64-bit words, most significant word of an entry first */
int nflhip_gauss_info(const nflhip_gauss *g, long long *x_min, size_t *entries, int *words, unsigned *bit_precision,
                      double *tail, uint64_t *h_table);
int nflhip_sample_gauss_dev(nflhip_ctx *ctx, void *d_data, size_t first_poly, size_t batch, const nflhip_gauss *g,
                            uint64_t amplifier, const unsigned char key[32], uint64_t stream_id, void *stream);
int nflhip_sample_gauss(nflhip_ctx *ctx, void *h_data, size_t batch, const nflhip_gauss *g, uint64_t amplifier,
                        const unsigned char key[32], uint64_t stream_id);
/* sequence form, see nflhip_sample_seq_dev */
int nflhip_sample_gauss_seq_dev(nflhip_ctx *ctx, void *d_data, size_t batch, const nflhip_gauss *g, uint64_t amplifier,
                                const unsigned char key[32], uint64_t first_stream_id, uint64_t stream_id_stride,
                                void *stream);
/* COMPACT forms (format NFLHIP_FMT_I8 / I16 / I32): d_out[b][i] = x * amplifier as ONE signed integer per coefficient,
 * x exactly the sample nflhip_sample_gauss_dev / nflhip_sample_gauss_seq_dev would spread over the moduli for the same
 * arguments -- so nflhip_expand_small_dev (or the forward entries above, which expand in their prologue) reproduce those
 * calls' words bit for bit at 1/32 (int8, 4 moduli of 64 bits) of the bytes.  NFLHIP_ERR_INVALID when a possible sample
 * does not fit the format or is not below every modulus. */
int nflhip_sample_gauss_small_dev(nflhip_ctx *ctx, void *d_out, int format, size_t first_poly, size_t batch,
                                  const nflhip_gauss *g, uint64_t amplifier, const unsigned char key[32], uint64_t stream_id,
                                  void *stream);
int nflhip_sample_gauss_small_seq_dev(nflhip_ctx *ctx, void *d_out, int format, size_t batch, const nflhip_gauss *g,
                                      uint64_t amplifier, const unsigned char key[32], uint64_t first_stream_id,
                                      uint64_t stream_id_stride, void *stream);
/* `count` (1..4) compact draws from ONE table in ONE launch: draw j writes `batch` polynomials to d_out[j] with amplifier[j] and is,
 * byte for byte, nflhip_sample_gauss_small_seq_dev(..., stream_id[j], stream_id_stride[j], ...) when stream_id_stride is given and
 * nflhip_sample_gauss_small_dev(..., first_poly 0, ..., stream_id[j], ...) when it is NULL.  What an LWE encryption draws per
 * ciphertext (FastGaussianNoise.hpp:477-595 through tests/nfllib_demo_main_op.cpp:33-35: x, e0, e1 with their own amplifiers):
 * three launches of a few hundred polynomials each leave wave slots empty in their last round and pay three ramps. */
int nflhip_sample_gauss_small_multi_dev(nflhip_ctx *ctx, void *const *d_out, size_t count, int format, size_t batch,
                                        const nflhip_gauss *g, const uint64_t *amplifier, const unsigned char key[32],
                                        const uint64_t *stream_id, const uint64_t *stream_id_stride, void *stream);
/* FastGaussianNoise::getNoise(out, rlen) (FastGaussianNoise.hpp:477-595): `count` raw signed samples; sample j is the
 * integer that coefficient first_sample + j of a polynomial batch gets from the same (key, stream_id) */
int nflhip_gauss_noise_dev(nflhip_ctx *ctx, int64_t *d_out, uint64_t first_sample, size_t count, const nflhip_gauss *g,
                           const unsigned char key[32], uint64_t stream_id, void *stream);
int nflhip_gauss_noise(nflhip_ctx *ctx, int64_t *h_out, size_t count, const nflhip_gauss *g,
                       const unsigned char key[32], uint64_t stream_id);

/* nfl::fastrandombytes(r, rlen) (nfl/prng/fastrandombytes.h:12): rlen raw keystream bytes of (key, stream_id) -- the
 * bytes of nflhip_random_words_dev's words 0.. in little-endian order -- generated on `device`, copied to the host */
int nflhip_random_bytes(int device, unsigned char *h_out, size_t nbytes, const unsigned char key[32], uint64_t stream_id);

/* plain device-memory helpers so a C caller needs no HIP headers */
int nflhip_malloc(nflhip_ctx *ctx, void **d_ptr, size_t bytes);
int nflhip_free(nflhip_ctx *ctx, void *d_ptr);
int nflhip_memcpy_h2d(nflhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
int nflhip_memcpy_d2h(nflhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
int nflhip_memcpy_d2d(nflhip_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream);
int nflhip_memset_dev(nflhip_ctx *ctx, void *d_dst, int byte, size_t bytes, void *stream);
int nflhip_stream_sync(nflhip_ctx *ctx, void *stream);
/* never blocks: *idle = 1 when everything enqueued on `stream` so far has completed, else 0 (hipStreamQuery).  What the
 * header's deferred queue asks before it starts a run early (a short loop's records would otherwise wait for the loop's
 * end with the device idle).  Not for a stream that is being captured into a hipGraph (a query ends the capture). */
int nflhip_stream_idle(nflhip_ctx *ctx, void *stream, int *idle);
/* a non-blocking stream of the context's device (what the header's resident handles enqueue on) */
int nflhip_stream_create(nflhip_ctx *ctx, void **stream);
int nflhip_stream_destroy(nflhip_ctx *ctx, void *stream);
/* d_dst[k] = *d_one for k < count: one polynomial replicated over a resident batch (one kernel, no host copies) */
int nflhip_broadcast_dev(nflhip_ctx *ctx, void *d_dst, const void *d_one, size_t count, void *stream);

/* ---- multi-GPU: the batch split --------------------------------------------------------
 * The reference has no distributed layer; its callers hold dense arrays of independent polynomials
 * (tests/tools.h:6-17: alloc_aligned<poly_t, 32>(N)) and loop over them (tests/nfl_mul_main.cpp, nfllib_demo_main_op.cpp).
 * Across the GPUs of one node that array is cut into contiguous shards: device r of n owns polynomials
 * [first, first + count) of `total` as nflhip_shard_range says, every device builds identical tables locally (one
 * context per device: nflhip_ctx_create's `device` argument) and there is NO data-path collective: operands are generated
 * in place (the `first_poly` argument of nflhip_fill_uniform_dev / nflhip_sample*_dev) or scattered once.  Two ways to
 * drive it:
 *   ONE PROCESS, n devices  -- n contexts, one stream each; nflhip_scatter_local_dev / nflhip_gather_local_dev move
 *                              shards with peer-to-peer copies (hipMemcpyPeerAsync), one per peer, each on that peer's
 *                              stream so that every xGMI link is busy at once;
 *   ONE PROCESS PER DEVICE  -- an nflhip_comm on RCCL: nflhip_scatter_dev / nflhip_gather_dev are grouped
 *                              ncclSend / ncclRecv of contiguous shards (one message per peer and group, <= 1 GiB each).
 * RCCL is bound at run time (librccl.so.1 is opened by the first nflhip_comm_* call), so single-GPU callers do not
 * depend on it. */
int nflhip_ctx_device(const nflhip_ctx *ctx);
/* contiguous, balanced split: the first (total mod nranks) ranks own one polynomial more */
int nflhip_shard_range(size_t total, int nranks, int rank, size_t *first, size_t *count);
/* 64-bit digest of `batch` resident polynomials that COMPOSES over shards: sum over words of
 * (g + 1) * mix(word) mod 2^64, g = the word's index in the whole logical batch (first_poly offsets it), so the sum of
 * the shards' digests equals the digest of the whole batch ("checksum of checksums").  Synchronises `stream`. */
int nflhip_digest_dev(nflhip_ctx *ctx, const void *d_data, size_t first_poly, size_t batch, uint64_t *h_digest,
                      void *stream);
/* copy between the devices of two contexts of this process, asynchronous on `stream` (a stream of dst_ctx's device) */
int nflhip_memcpy_peer_dev(nflhip_ctx *dst_ctx, void *d_dst, nflhip_ctx *src_ctx, const void *d_src, size_t bytes,
                           void *stream);
/* ONE PROCESS: ctxs[r] / d_shards[r] / streams[r] belong to device r of n (same shape everywhere); d_full holds `total`
 * polynomials on ctxs[root]'s device.  scatter: shard r <- full[first_r, first_r + count_r); gather: the inverse.
 * Each peer's copy runs on that peer's stream after everything enqueued so far on the root's stream (scatter) / on the
 * peer's stream (gather), and the root's stream waits for all of them: no host synchronisation. */
int nflhip_scatter_local_dev(nflhip_ctx *const *ctxs, int n, void *const *d_shards, int root, const void *d_full,
                             size_t total, void *const *streams);
int nflhip_gather_local_dev(nflhip_ctx *const *ctxs, int n, void *d_full, int root, const void *const *d_shards,
                            size_t total, void *const *streams);
/* ONE PROCESS PER DEVICE: a communicator over RCCL (ncclCommInitRank).  Rank 0 draws the id (ncclGetUniqueId) and hands
 * it to the other ranks out of band (a file, MPI, a key-value store ...).  nranks = 1 is legal (and what a
 * single-GPU box can run). */
typedef struct nflhip_comm nflhip_comm;
#define NFLHIP_COMM_ID_BYTES 128
int nflhip_comm_unique_id(unsigned char id[NFLHIP_COMM_ID_BYTES]);
int nflhip_comm_create(nflhip_comm **out, nflhip_ctx *ctx, int nranks, int rank, const unsigned char id[NFLHIP_COMM_ID_BYTES]);
int nflhip_comm_destroy(nflhip_comm *comm);
int nflhip_comm_rank(const nflhip_comm *comm);
int nflhip_comm_size(const nflhip_comm *comm);
/* scatter: rank `root` holds d_full (`total` polynomials, ignored elsewhere); every rank receives its shard_range slice
 * in d_shard.  gather: the inverse.  Asynchronous on `stream`; every rank must call with the same total and root. */
int nflhip_scatter_dev(nflhip_comm *comm, void *d_shard, const void *d_full, size_t total, int root, void *stream);
int nflhip_gather_dev(nflhip_comm *comm, void *d_full, const void *d_shard, size_t total, int root, void *stream);
/* control plane: a barrier (one-word all-reduce) and an all-gather of one 64-bit word per rank (shard digests, clocks).
 * Both synchronise `stream`. */
int nflhip_comm_barrier(nflhip_comm *comm, void *stream);
int nflhip_comm_allgather_u64(nflhip_comm *comm, uint64_t mine, uint64_t *h_all, void *stream);

/* ---- in-library timing of the metric kernel (HIP events on `stream`) -----------
 * Runs `iters` back-to-back polymul passes over the batch and returns the mean
 * milliseconds per pass measured with hipEvents recorded on the same stream the
 * kernels are launched on (bench.py's roofline leg). */
int nflhip_time_polymul_dev(nflhip_ctx *ctx, void *d_c, const void *d_a, const void *d_b, size_t batch,
                            int iters, void *stream, float *ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* NFLHIP_H */
