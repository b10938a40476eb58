// kernels.h -- internal launch interface between the C-ABI layer (api.hip) and
// the gfx950 kernels.  Not installed; the public boundary is include/nflhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nflhip {

// Per-modulus device constants (engine layout; derived from params<T>::P,
// primitive_roots, invkMaxPolyDegree -- reference params.hpp -- by tables.cpp).
template <typename T> struct ModConst {
  T p;          // modulus
  T p2;         // 2p
  T mu;         // floor(2^(2W-4)/p): Barrett constant for exact x*y mod p
  T ninv;       // n^-1 mod p                       (core.hpp:664-665)
  T ninv_sh;    // Shoup companion of ninv
  T w1ninv;     // psi_br[1] * n^-1 mod p (last inverse stage, merged scale)
  T w1ninv_sh;  // its Shoup companion
  T beta;       // 2^64 mod p (CRT project, Horner step)
  T beta_sh;    // its Shoup companion
  T yinv;       // (Q/p)^-1 mod p (CRT lift)         (gmp.hpp:141-147)
  T yinv_sh;    // its Shoup companion
  T mask;       // 2^(floor(log2 p)+1) - 1           (core.hpp:165-166)
  T delta;      // 2^(W-2) - p: the primes are 2^(W-2) - c*2*kMax + 1 (params.hpp:20,54,96)
  T mu2;        // floor(2^(2W-3)/p): Barrett constant for lazily reduced operands (< 2^(W-2) + 3*delta)
};

// Twiddle pair as stored on the device: psi^bitrev(k) and its Shoup companion.
template <typename T> struct alignas(2 * sizeof(T)) Tw {
  T w, wp;
};

struct Shape {
  int limb_bits;
  int logn;
  size_t n, nm;
  size_t crt_L;      // limbs of a lifted coefficient
  size_t crt_Lacc;   // limbs of the accumulator (L+1)
  uint64_t crt_Q0;   // the moduli product when it is below 2^64 (crt_L == 1), else its low word
  int small_delta;   // every modulus is 2^(W-2) - delta with delta < 2^32 (delta-form butterflies)
  int nm_small;      // length of the PREFIX of moduli with delta < 2^32 (== nm when small_delta): at degree 4096 these rows keep the
                     // delta-form kernels in a context whose later moduli need the general family (params.hpp:82-119: #92 on)
  // configuration, read ONCE when the context is created (include/nflhip.h "environment"):
  int compiled_only; // NFLHIP_VARIANT=hipcc: the compiled (hipcc) kernels serve every call -- the independent cross-check
                     // of the generated assembly kernels (bit-identical results, tests/test_gpu_variants.py)
  int plan;          // NFLHIP_XCD: rows of 32768 / 65536 words -- -1 by batch size (default), 0 never / 1 always the
                     // one-launch plan of persistent workgroups
};

// Device-resident tables of one context.
struct DevTables {
  void *psi;        // [nm][n] Tw<T>
  void *psi_lm;     // 64-bit limbs, n >= 4096: the same with the last four stages lane-major (what the generated kernels read)
  void *mc;         // [nm] ModConst<T>
  void *mc_inc[2];  // 64-bit limbs, n = 4096, delta-form moduli: the records the incomplete-transform products read (level 1, 2:
                    // (n / 2^level)^-1 in the n^-1 fields, floor(2^127 / p) - 2^65 in mu2; tools/asmgen/incomplete.py), or nullptr
  uint64_t *qhat;   // [nm][crt_Lacc]  Q/p_cm, little-endian limbs
  uint64_t *qsh;    // [6][crt_Lacc]   Q << k, k = 0..5
  uint32_t *qparts; // [nm][3][72]     32-bit digits of (Q/p_cm) << 21 j   (carry-free lift), or nullptr
  uint32_t *bparts; // [proj_K][2][3][nm rounded up to 4] 21-bit parts of 2^(64k+32h) mod p_cm (project), or nullptr
  int proj_K;       // input words the project table covers
  double inv_qtop;  // 2^(32 (2 crt_L - 3)) / Q in double precision (quotient estimate of the lift)
  int *flag;        // kCmpSlots ints: result slots of any_eq / any_neq (one per concurrent call, see api.hip)
  // CRT lift beyond the register-resident kernels (more than 32 moduli / 36 limbs): limb-serial kernel, any size
  uint64_t *qhat_w; // [nm][crt_Lw]   Q/p_cm, little-endian limbs, or nullptr when the fast tables cover the shape
  uint64_t *qsh_w;  // [crt_nsh][crt_Lw]  Q << k
  int crt_Lw, crt_nsh;
  // CRT lift on the matrix cores (kernels_crt_mfma.hip): 64-bit limbs, many moduli
  void *crt_bfrag;  // [8 K-steps][8 N-tiles][64 lanes][16] int8: balanced base-256 digits of Q/p_cm in B-fragment order, or nullptr
  void *crt_bproj;  // the same for the projection: digit t of 256^k mod p_cm, k = 32 s + 16 (lane >> 5) + byte, cm = lane & 31, or nullptr
  uint64_t *crt_c2048; // [32][2] 2^2048 mod p_cm and its Shoup companion: how the residue of an input's upper 32 words joins the lower words'
  uint64_t *crt_coff; // [32][2] 2^18 p_cm + 128 sum_k (256^k mod p_cm in those digits), 128 bits: what the projection adds before reducing
};

// ---- launchers (kernels_generic.hip) ----
template <typename T>
hipError_t launch_ntt_fwd(const Shape &s, const DevTables &t, const T *src, T *dst, size_t batch, hipStream_t st);
// inverse of (src (.) mul) when mul != nullptr (fused point-wise product), else of src
template <typename T>
hipError_t launch_ntt_inv(const Shape &s, const DevTables &t, const T *src, const T *mul, T *dst, size_t batch,
                          hipStream_t st);
template <typename T>
hipError_t launch_pointwise(const Shape &s, const DevTables &t, int op, T *out, const T *a, const T *b, const T *bp,
                            size_t batch, hipStream_t st);
// postfix expression program (include/nflhip.h NFLHIP_EXPR_*) over up to 8 operands, one fused pass
template <typename T>
hipError_t launch_eval_expr(const Shape &s, const DevTables &t, T *out, const void *const *operands, int noperands,
                            const unsigned char *program, int len, size_t batch, hipStream_t st,
                            const unsigned *strides = nullptr, unsigned out_stride = 1);  // strides in polynomials (0 = shared)
template <typename T>
hipError_t launch_any_cmp(const Shape &s, const DevTables &t, const T *a, const T *b, size_t batch, int want_eq,
                          int *flag, int token, hipStream_t st);   // a hit stores `token` into *flag (no clearing, no atomic)
template <typename T>
hipError_t launch_check_range(const Shape &s, const DevTables &t, const T *d, size_t batch, int *flag, int token, hipStream_t st);
template <typename T>
hipError_t launch_fill_uniform(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch, uint64_t seed,
                               int operand, hipStream_t st);
// in-place bit reversal of every row (permut.hpp:86-117), and `count` copies of one polynomial
template <typename T> hipError_t launch_bitrev_rows(const Shape &s, T *d, size_t rows, hipStream_t st);
hipError_t launch_broadcast(void *dst, const void *one, size_t bytes_per_poly, size_t count, hipStream_t st);
template <typename T>
hipError_t launch_crt_lift(const Shape &s, const DevTables &t, uint64_t *limbs, const T *d, size_t batch, hipStream_t st);
template <typename T>
hipError_t launch_crt_project(const Shape &s, const DevTables &t, T *d, const uint64_t *limbs, size_t L_in, size_t batch,
                              hipStream_t st);
// any number of moduli: `scratch` = batch * n * t.crt_Lw words (the unreduced sums, limb-major)
template <typename T>
hipError_t launch_crt_lift_wide(const Shape &s, const DevTables &t, uint64_t *limbs, const T *d, size_t batch,
                                uint64_t *scratch, hipStream_t st);

// ---- samplers (kernels_sample.hip): ChaCha20 counter streams keyed by (key32, stream_id) ----
void set_gauss_tie_shift(int shift);  // debug entry point nflhip_debug_gauss_tie_shift (include/nflhip_debug.h)
hipError_t launch_random_words(uint64_t *out, uint64_t first_word, size_t nwords, const unsigned char *key32,
                               uint64_t stream_id, hipStream_t st);
// dist: 0 uniform | 1 bounded (p0 = upper bound, p1 = amplifier) | 2 zero/one (p0 = rho) | 3 hamming weight (p0 = h);
// | 0x100 reference words (zero/one, hamming weight) | 0x200 narrow draw (uniform: keystream lanes of the limb width)
template <typename T>
hipError_t launch_sample(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch, int dist, uint64_t p0,
                         uint64_t p1, const unsigned char *key32, uint64_t stream_id, hipStream_t st, int seq_on = 0,
                         uint64_t seq_stride = 0);  // seq_on: polynomial b = a one-polynomial call with stream id + b * seq_stride
hipError_t launch_inner_fwd_fast_u32(const Shape &s, const DevTables &t, const uint32_t *src, uint32_t *dst, size_t rows,
                                     hipStream_t st);
hipError_t launch_inner_inv_fast_u32(const Shape &s, const DevTables &t, const uint32_t *src, const uint32_t *mul,
                                     uint32_t *dst, size_t rows, hipStream_t st);
// narrow (every Gaussian launcher): the table's draw width is 32 bits (nflhip_gauss_set_draw_bits) -- domains 7 / 8 of the keystream
hipError_t launch_gauss_noise(long long *out, uint64_t first_sample, size_t count, const uint64_t *cdt, int words,
                              int entries, long long x_min, const unsigned char *key32, uint64_t stream_id, hipStream_t st,
                              int narrow = 0);
template <typename T>
hipError_t launch_sample_gauss(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch,
                               const uint64_t *cdt, int words, int entries, long long x_min, uint64_t amp,
                               const unsigned char *key32, uint64_t stream_id, hipStream_t st, int seq_on = 0,
                               uint64_t seq_stride = 0, int narrow = 0, const uint16_t *lut = nullptr);
// the narrow draw's bucket table (device copy passed as `lut` above and below; empty = the table does not fit LDS)
std::vector<uint16_t> gauss_bucket_table(const uint64_t *cdt, int words, size_t entries);

// compact Gaussian polynomials (one signed integer x * amp per coefficient; format 1 int8 | 2 int16 | 3 int32) and their
// expansion into residue words (format 0 = word rows: a strided gather); stride in polynomials, 0 = shared
hipError_t launch_gauss_small(const Shape &s, void *d, int format, size_t first_poly, size_t batch, const uint64_t *cdt,
                              int words, int entries, long long x_min, uint64_t amp, const unsigned char *key32,
                              uint64_t stream_id, hipStream_t st, int seq_on = 0, uint64_t seq_stride = 0, int narrow = 0,
                              const uint16_t *lut = nullptr);
hipError_t launch_gauss_small_multi(const Shape &s, void *const *d, size_t count, int format, size_t batch, const uint64_t *cdt, int words,
                                    int entries, long long x_min, const uint64_t *amp, const unsigned char *key32, const uint64_t *stream_id,
                                    const uint64_t *seq_stride, hipStream_t st, int narrow, const uint16_t *lut);
template <typename T>
hipError_t launch_expand_small(const Shape &s, const DevTables &t, T *dst, const void *src, int format, unsigned stride,
                               size_t batch, hipStream_t st);

// transform-fused pipelines at n = 4096, 64-bit limbs (tools/gen_polymul_asm.py build_fused): kind 0 enc2 | 1 fma_fwd |
// 2 fms_inv | 3 fma_inv; x: up to three operands with their formats (forward kinds) and strides, k: key rows with strides.
// hipErrorNotSupported for other shapes / the compiled-only variant (api.hip composes the same result from the plain kernels)
// INTT(b -+ a (.) key) in one pass on the wave-per-row kernels (kernels_wave.hip): rows of 1024 / 2048 words, 4096 for 32-bit limbs
// out0 = NTT(x) k0 + NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)] on the wave-per-row kernels (kernels_wave.hip k_row_fwd_fma): rows of 1024 /
// 2048 words (4096 for 32-bit limbs), x / e0 / e1 of one format, strides 0 or 1; hipErrorNotSupported otherwise
hipError_t launch_row_fwd_fma_u32(const Shape &s, const DevTables &t, int format, uint32_t *out0, uint32_t *out1, const void *x, unsigned xs,
                                  const uint32_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint32_t *k1, unsigned k1s, const void *e1,
                                  unsigned e1s, size_t batch, hipStream_t st);
hipError_t launch_row_fwd_fma_u64(const Shape &s, const DevTables &t, int format, uint64_t *out0, uint64_t *out1, const void *x, unsigned xs,
                                  const uint64_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint64_t *k1, unsigned k1s, const void *e1,
                                  unsigned e1s, size_t batch, hipStream_t st);
hipError_t launch_row_fma_inv_u32(const Shape &s, const DevTables &t, int subtract, uint32_t *c, const uint32_t *a, const uint32_t *key,
                                  int kstride, const uint32_t *b, size_t batch, hipStream_t st);
hipError_t launch_row_fma_inv_u64(const Shape &s, const DevTables &t, int subtract, uint64_t *c, const uint64_t *a, const uint64_t *key,
                                  int kstride, const uint64_t *b, size_t batch, hipStream_t st);
// rows of 32768 words from a compact (int8) Gaussian polynomial: its forward transform, and NTT(x) k0 + e0' [, NTT(x) k1 + e1'] (e' words)
hipError_t launch_row32k_fwd_i8_u64(const Shape &s, const DevTables &t, uint64_t *dst, const void *x8, size_t batch, hipStream_t st);
hipError_t launch_row32k_fwd_fma_i8_u64(const Shape &s, const DevTables &t, uint64_t *out0, uint64_t *out1, const void *x8,
                                        const uint64_t *k0, const uint64_t *e0p, const uint64_t *k1, const uint64_t *e1p, size_t batch,
                                        hipStream_t st);
hipError_t launch_fused_asm_u64(const Shape &s, const DevTables &t, int kind, uint64_t *out0, uint64_t *out1,
                                const void *const *x, const unsigned *xstride, const int *xfmt, const void *const *k,
                                const unsigned *kstride, size_t batch, hipStream_t st);

// ---- fast paths (kernels_fast.hip); return hipErrorNotSupported when the shape has none ----
hipError_t launch_polymul_fast_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                   int b_is_ntt, size_t batch, hipStream_t st);
hipError_t launch_ntt_fwd_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st);
hipError_t launch_ntt_inv_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st);

// 4096-word inner blocks of rows with logn >= 12 on the register-tiled kernels (rows = batch * nm);
// `mul` != nullptr fuses the point-wise product into the inverse's load.
hipError_t launch_inner_fwd_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t rows,
                                     hipStream_t st);
hipError_t launch_inner_inv_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, const uint64_t *mul,
                                     uint64_t *dst, size_t rows, hipStream_t st);

// streaming outer passes alone (rows with logn > 12): forward src -> dst, inverse in place
// (`logi` = log2 of the block the following fused kernel owns: global stages [0, logn - logi) run here)
hipError_t launch_outer_fwd_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t rows,
                                hipStream_t st, int logi = 12);
hipError_t launch_outer_inv_u64(const Shape &s, const DevTables &t, uint64_t *data, size_t rows, hipStream_t st,
                                int logi = 12);
// fused NTT,NTT,(.),INTT over the 4096-word blocks of rows whose outer forward passes already ran
// (a_in, b_in) and whose outer inverse passes still have to run on c (logn > 12), or the whole
// polymul for logn == 12.  hipErrorNotSupported when the assembly code object is unavailable.
// b_is_ntt: b_in is the fully transformed operand (canonical words), only a_in went through the outer passes.
hipError_t launch_polymul_blocks_asm_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a_in,
                                         const uint64_t *b_in, size_t batch, hipStream_t st, bool b_is_ntt = false);

// 32768-word rows, ONE operand register-resident per 1024-thread workgroup: mode 1: c = INTT(NTT(a) (.) b) with b already
// transformed (streamed through the point-wise step), 2: c = NTT(a), 3: c = INTT(a); 4 / 5: modes 2 / 1 with the transformed
// operand in the layout [block][pair][thread] that only these two kernels share (the composed product's scratch: coalesced on
// both sides, no transposes); hipErrorNotSupported for other shapes
hipError_t launch_row32k_u64(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a, const uint64_t *b,
                             size_t batch, hipStream_t st);
// n = 65536: one launch of the three-role pipeline kernel (block products of chunk j-1, forward streaming pass of
// chunk j, inverse streaming pass of chunk j-2); hipErrorNotSupported for other shapes
// one launch for the whole batch, rows pinned to an XCD (n = 65536 / 32768); xcd_plan_bytes() = 0 when the shape / batch
// 16-bit limbs, n = 128, fused product: the generated gfx950 assembly kernel (hipErrorNotSupported: composed plan)
hipError_t launch_row128_u16_asm(const Shape &s, const DevTables &t, int mode, uint16_t *c, const uint16_t *a,
                                 const uint16_t *b, size_t batch, hipStream_t st);
// 32-bit limbs, n = 1024, fused product: the generated gfx950 assembly kernel (hipErrorNotSupported: use k_row)
hipError_t launch_row1024_u32_asm(const Shape &s, const DevTables &t, int mode, uint32_t *c, const uint32_t *a,
                                  const uint32_t *b, size_t batch, hipStream_t st);
// ... their transform-fused pipelines (formats words / int8, strides 0 / 1; hipErrorNotSupported: the compiled kernels)
hipError_t launch_row_fwd_fma_u32_asm(const Shape &s, const DevTables &t, int format, uint32_t *out0, uint32_t *out1, const void *x, unsigned xs,
                                      const uint32_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint32_t *k1, unsigned k1s,
                                      const void *e1, unsigned e1s, size_t batch, hipStream_t st);
hipError_t launch_row_fma_inv_u32_asm(const Shape &s, const DevTables &t, int subtract, uint32_t *c, const uint32_t *a, const uint32_t *key,
                                      int kstride, const uint32_t *b, size_t batch, hipStream_t st);
hipError_t launch_row_fwd_fma_u64_asm(const Shape &s, const DevTables &t, int format, uint64_t *out0, uint64_t *out1, const void *x, unsigned xs,
                                      const uint64_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint64_t *k1, unsigned k1s,
                                      const void *e1, unsigned e1s, size_t batch, hipStream_t st);
hipError_t launch_row_fma_inv_u64_asm(const Shape &s, const DevTables &t, int subtract, uint64_t *c, const uint64_t *a, const uint64_t *key,
                                      int kstride, const uint64_t *b, size_t batch, hipStream_t st);
// 64-bit limbs, n = 1024 / 2048: the generated twins (modes 0, 2, 3; hipErrorNotSupported: use k_row)
hipError_t launch_row1024_u64_asm(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a,
                                  const uint64_t *b, size_t batch, hipStream_t st);
// is not covered.  `work`: device memory of that many bytes, initialised by the call on `st`.
size_t xcd_plan_bytes(const Shape &s, size_t batch);
hipError_t launch_polymul_xcd_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                  size_t batch, void *work, hipStream_t st, int level = 0);
// one empty launch per translation unit (+ the generated kernels' module): api.hip warm_up_device
hipError_t warm_generic(hipStream_t st);
hipError_t warm_fast(hipStream_t st);
hipError_t warm_crt(hipStream_t st);
hipError_t warm_crt_mfma(hipStream_t st);
hipError_t warm_sample(hipStream_t st);
hipError_t warm_wave(hipStream_t st);
int polymul_level();   // 0 / 1 / 2: transforms of the coefficient-form products complete / incomplete (kernels_fast.hip, nflhip_debug_polymul_level)
hipError_t launch_polymul_pipe64k_u64(const Shape &s, const DevTables &t, uint64_t *c_v, const uint64_t *a_v,
                                      const uint64_t *b_v, int cnt_v, const uint64_t *fa_src, uint64_t *fa_dst,
                                      const uint64_t *fb_src, uint64_t *fb_dst, int cnt_f, uint64_t *inv, int cnt_i,
                                      hipStream_t st, bool b_is_ntt = false, int level = 0);

// n = 1024, 32- and 64-bit limbs: one wave per row (kernels_wave.hip).  mode 0: c = INTT(NTT(a)(.)NTT(b)); 1: b already in
// NTT form; 2: c = NTT(a); 3: c = INTT(a).  hipErrorNotSupported for other shapes.
hipError_t launch_row1024_u32(const Shape &s, const DevTables &t, int mode, uint32_t *c, const uint32_t *a,
                              const uint32_t *b, size_t batch, hipStream_t st);
hipError_t launch_row1024_u64(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a,
                              const uint64_t *b, size_t batch, hipStream_t st);

// register-resident CRT kernels for 64-bit limbs (kernels_crt.hip); hipErrorNotSupported otherwise
hipError_t launch_crt_lift_fast_u64(const Shape &s, const DevTables &t, uint64_t *limbs, const uint64_t *d, size_t batch,
                                    hipStream_t st);
// the lift as an int8 GEMM on v_mfma_i32_32x32x32_i8 (kernels_crt_mfma.hip): many moduli, batch * n a multiple of 64
hipError_t launch_crt_lift_mfma_u64(const Shape &s, const DevTables &t, uint64_t *limbs, const uint64_t *d, size_t batch,
                                    hipStream_t st);
hipError_t launch_crt_project_mfma_u64(const Shape &s, const DevTables &t, uint64_t *d, const uint64_t *limbs, size_t L_in,
                                       size_t batch, hipStream_t st);
hipError_t launch_crt_project_fast_u64(const Shape &s, const DevTables &t, uint64_t *d, const uint64_t *limbs, size_t L_in,
                                       size_t batch, hipStream_t st);

}  // namespace nflhip
