// kernels_fast.hip -- register-tiled gfx950 kernels for 64-bit limbs, degree 4096
// (BASELINE.json configs[1], the metric shape: nfl::poly<uint64_t, 4096, 4>).
//
// One 256-thread workgroup (4 wavefronts) owns one RNS row = one (polynomial,
// modulus) slab of 4096 x 8 B = 32 KiB.  Each thread keeps 16 coefficients in
// VGPRs and the 12 butterfly stages run as three radix-16 register passes:
//
//   forward (Cooley-Tukey, natural -> bit-reversed; merged psi twiddles)
//     F1 stages 0-3   thread t holds x[t + 256k]      twiddles wave-uniform (SGPR)
//     -- exchange E1 through LDS (all-to-all inside the workgroup, 1 barrier)
//     F2 stages 4-7   thread (B,r) holds x[256B + r + 16k]
//     -- exchange E2 through LDS (16-lane groups: wave-local, no barrier)
//     F3 stages 8-11  thread q holds x[16q + k]
//   inverse (Gentleman-Sande) is the mirror image I1 (no exchange needed after
//   F3: same layout), E2', I2, E1', I3 with n^-1 folded into the last stage.
//
// The fused polymul kernel therefore touches HBM exactly once per operand word:
// read a, read b, write c = 3 x 32 KiB per row (the algorithmic minimum of
// SURVEY.md 8(d)); everything else stays in VGPRs/LDS.  LDS words are stored at
// index e + (e >> 4) (one pad word per 16) which makes every ds_read_b64 /
// ds_write_b64 of both exchange patterns bank-conflict free.
//
// Reference behaviour replaced: core::ntt_pow_phi (core.hpp:594-600), the
// point-wise mulmod loop (core.hpp:24-37 with ops.hpp:201-219) and
// core::invntt_pow_invphi (core.hpp:608-614).
#include <mutex>

#include "kernels.h"
#include "modarith64.h"
#include <atomic>
#include <cstdlib>

namespace nflhip {

static constexpr int kLogN = 12;
static constexpr int kN = 1 << kLogN;
static constexpr int kThreads = 256;
static constexpr int kLdsWords = kN + (kN >> 4);  // padded slab

__device__ __forceinline__ int pad(int e) { return e + (e >> 4); }

// (the lazy butterflies ct_bfly / gs_bfly, canon and mul_lazy live in modarith64.h: kernels_wave.hip uses them too)

// radix-16 register passes; TW(s, g) yields the twiddle of sub-stage s (0..3), group g
template <int ARITH, class TW> __device__ __forceinline__ void ct16(u64 (&v)[16], TW tw, const Mod &k) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw(s, g);
#pragma unroll
      for (int h = 0; h < half; ++h) ct_bfly<ARITH>(v[g * 2 * half + h], v[g * 2 * half + h + half], w, k);
    }
  }
}
template <int ARITH, class TW> __device__ __forceinline__ void gs16(u64 (&v)[16], TW tw, const Mod &k) {
#pragma unroll
  for (int s = 3; s >= 0; --s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw(s, g);
#pragma unroll
      for (int h = 0; h < half; ++h) gs_bfly<ARITH>(v[g * 2 * half + h], v[g * 2 * half + h + half], w, k);
    }
  }
}

// Position of the 4096-word block this workgroup transforms inside a longer row of
// n = 4096 * 2^r words (r = 0: the row itself).  The streaming outer passes of
// kernels_generic.hip handle global stages [0, r); the passes below then cover global
// stages r .. r+11 with block index offsets folded into the twiddle indices.
struct Blk {
  int r;
  unsigned blk;
};

// ---- forward transform of the 16 words a thread loaded as x[t + 256k] ---------------
// On return thread q = t holds X[16q + k] (bit-reversed order positions), lazy in [0,4p).
template <int ARITH>
__device__ __forceinline__ void fwd_head(u64 (&v)[16], u64 *sm, const Tw64 *__restrict__ tw, const Mod &k, const int t,
                                         const bool war_barrier, const Blk bk) {
  // F1: stages r..r+3, block index of sub-stage s is blk*2^s + g: psi[2^(r+s) + ...] (wave-uniform)
  ct16<ARITH>(v, [&](int s, int g) { return tw[(1u << (bk.r + s)) + (bk.blk << s) + g]; }, k);
  if (war_barrier) __syncthreads();  // the slab may still be read by slower waves (previous transform)
  {
    const int base = t + (t >> 4);
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[base + 272 * k] = v[k];
  }
  __syncthreads();
  const int B = t >> 4, r = t & 15;
  {
    const int base = 272 * B + r;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sm[base + 17 * k];
  }
  // F2: stages 4-7 inside 256-word block B: psi[2^(4+s) + B*2^s + g]
  ct16<ARITH>(v, [&](int s, int g) { return tw[(16u << (bk.r + s)) + ((bk.blk * 16u + B) << s) + g]; }, k);
  // E2: 16-lane transpose through this wave's own LDS region (LDS is in-order per wave)
  {
    const int base = 272 * B + r;
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[base + 17 * k] = v[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    const int base = 17 * t;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sm[base + k];
  }
}
// F3: stages 8-11 inside 16-word block q = t: psi[2^(8+s) + q*2^s + g]
template <int ARITH>
__device__ __forceinline__ void fwd_tail(u64 (&v)[16], const Tw64 *__restrict__ tw, const Mod &k, const int t,
                                         const Blk bk) {
  ct16<ARITH>(v, [&](int s, int g) { return tw[(256u << (bk.r + s)) + ((bk.blk * 256u + t) << s) + g]; }, k);
}
template <int ARITH>
__device__ __forceinline__ void fwd_core(u64 (&v)[16], u64 *sm, const Tw64 *__restrict__ tw, const Mod &k, const int t,
                                         const bool war_barrier, const Blk bk) {
  fwd_head<ARITH>(v, sm, tw, k, t, war_barrier, bk);
  fwd_tail<ARITH>(v, tw, k, t, bk);
}

// ---- inverse transform of the 16 words a thread holds as X[16q + k], in [0,2p) -------
// On return thread t holds x[t + 256k]: canonical in [0,p) when r == 0 (n^-1 folded into
// global stage 0), lazy in [0,2p) when outer inverse passes follow (r > 0).
template <int ARITH>
__device__ __forceinline__ void inv_core(u64 (&v)[16], u64 *sm, const Tw64 *__restrict__ tw, const MC64 &c, const Mod &k,
                                         const int t, const Blk bk) {
  const u64 p = c.p, p2 = c.p2;
  // I1: stages 11..8; mirrored index 2m-1-j with m = 2^(8+s), j = q*2^s + g
  gs16<ARITH>(v, [&](int s, int g) { return tw[(512u << (bk.r + s)) - 1u - (((bk.blk * 256u + t) << s) + g)]; }, k);
  const int B = t >> 4, r = t & 15;
  {
    const int base = 17 * t;
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[base + k] = v[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    const int base = 272 * B + r;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sm[base + 17 * k];
  }
  // I2: stages 7..4; m = 2^(4+s), j = B*2^s + g
  gs16<ARITH>(v, [&](int s, int g) { return tw[(32u << (bk.r + s)) - 1u - (((bk.blk * 16u + B) << s) + g)]; }, k);
  {
    const int base = 272 * B + r;
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[base + 17 * k] = v[k];
  }
  __syncthreads();
  {
    const int base = t + (t >> 4);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sm[base + 272 * k];
  }
  // I3: stages r+3..r+1 (uniform twiddles), then stage r (with n^-1 folded in when r == 0)
#pragma unroll
  for (int s = 3; s >= 1; --s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw[(2u << (bk.r + s)) - 1u - ((bk.blk << s) + g)];
#pragma unroll
      for (int h = 0; h < half; ++h) gs_bfly<ARITH>(v[g * 2 * half + h], v[g * 2 * half + h + half], w, k);
    }
  }
  if (bk.r > 0) {  // plain last stage: the merged scale happens in the outer pass that owns global stage 0
    const Tw64 w = tw[(2u << bk.r) - 1u - bk.blk];
#pragma unroll
    for (int h = 0; h < 8; ++h) gs_bfly<ARITH>(v[h], v[h + 8], w, k);
    return;
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const u64 x = v[h], y = v[h + 8];
    if (ARITH == 0) {
      v[h] = mul_shoup<u64>(x + y, c.ninv, c.ninv_sh, p);
      v[h + 8] = mul_shoup<u64>(y - x + p2, c.w1ninv, c.w1ninv_sh, p);
    } else {
      v[h] = csub<u64>(shoup_acc(x + y, Tw64{c.ninv, c.ninv_sh}, 0, k), p);
      v[h + 8] = csub<u64>(shoup_acc(y - x + p2, Tw64{c.w1ninv, c.w1ninv_sh}, 0, k), p);
    }
  }
}

// ---- the metric kernel: c = INTT( NTT(a) (.) NTT(b) ), one row per workgroup ---------

template <bool B_IS_NTT, int ARITH>
__device__ __forceinline__ void polymul_body(u64 *sm, u64 *c, const u64 *a, const u64 *b, const Tw64 *__restrict__ psi,
                                             const MC64 *__restrict__ mc, int nm, size_t row) {
  const int t = threadIdx.x;
  const int cm = (int)(row % (size_t)nm);
  const MC64 mcc = mc[cm];
  const Mod k = make_mod(mcc);
  const Tw64 *tw = psi + ((size_t)cm << kLogN);
  const size_t off = row << kLogN;

  u64 va[16], vb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) va[i] = a[off + t + 256 * i];
  const Blk bk{0, 0u};
  fwd_head<ARITH>(va, sm, tw, k, t, false, bk);
  // b's HBM loads are issued here so their latency hides under F3(a) without
  // holding 32 more VGPRs through the first two passes
  asm volatile("" ::: "memory");
  if (!B_IS_NTT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) vb[i] = b[off + t + 256 * i];
  } else {
    // b already in NTT form: thread q needs B[16q + k]
    const ulonglong2 *b2 = reinterpret_cast<const ulonglong2 *>(b + off + 16 * t);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const ulonglong2 x = b2[i];
      vb[2 * i] = x.x;
      vb[2 * i + 1] = x.y;
    }
  }
  asm volatile("" ::: "memory");
  fwd_tail<ARITH>(va, tw, k, t, bk);
  if (!B_IS_NTT) fwd_core<ARITH>(vb, sm, tw, k, t, true, bk);
  // point-wise product on canonical representatives (operator*, ops.hpp:201-219)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (ARITH >= 2) {
      va[i] = mul_lazy(fold2(va[i], k), B_IS_NTT ? vb[i] : fold2(vb[i], k), mcc.mu2, k);
    } else {
      const u64 x = canon<ARITH>(va[i], k);
      const u64 y = B_IS_NTT ? vb[i] : canon<ARITH>(vb[i], k);
      va[i] = barrett<u64>::mul(x, y, k.p, mcc.mu);
    }
  }
  inv_core<ARITH>(va, sm, tw, mcc, k, t, bk);
#pragma unroll
  for (int i = 0; i < 16; ++i) c[off + t + 256 * i] = va[i];
}

// a launch over the moduli [cm0, cm0 + cmcnt) of every polynomial only (cmcnt > 0): logical block L -> block index in the batch.
// Contexts whose moduli beyond a prefix need the general arithmetic send the prefix to the generated kernels and the rest here.
__device__ __forceinline__ size_t block_of(unsigned L, int nm, int r, int cm0, int cmcnt) {
  if (cmcnt <= 0) return L;
  const unsigned rs = L >> r, bl = L & ((1u << r) - 1u);
  const size_t row = (size_t)(rs / (unsigned)cmcnt) * (size_t)nm + (size_t)cm0 + rs % (unsigned)cmcnt;
  return (row << r) | bl;
}

template <bool B_IS_NTT, int ARITH, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void k_polymul4096(u64 *c, const u64 *a, const u64 *b,
                                                                const Tw64 *__restrict__ psi,
                                                                const MC64 *__restrict__ mc, int nm, int cm0, int cmcnt) {
  __shared__ u64 sm[kLdsWords];
  polymul_body<B_IS_NTT, ARITH>(sm, c, a, b, psi, mc, nm, block_of(blockIdx.x, nm, 0, cm0, cmcnt));
}

// ---- stand-alone transforms (in place or out of place) --------------------------------
// One workgroup per 4096-word block of a row of n = 2^logn words (logn >= 12); for
// logn > 12 the streaming outer passes run before (forward) / after (inverse) these.
template <int ARITH>
__global__ __launch_bounds__(kThreads) void k_ntt_fwd4096(const u64 *src, u64 *dst, const Tw64 *__restrict__ psi,
                                                          const MC64 *__restrict__ mc, int nm, int logn, int cm0, int cmcnt) {
  __shared__ u64 sm[kLdsWords];
  const int t = threadIdx.x;
  const int r = logn - kLogN;
  const size_t blk = block_of(blockIdx.x, nm, r, cm0, cmcnt);
  const size_t row = blk >> r;
  const Blk bk{r, (unsigned)(blk & ((1u << r) - 1u))};
  const int cm = (int)(row % (size_t)nm);
  const Mod k = make_mod(mc[cm]);
  const Tw64 *tw = psi + ((size_t)cm << logn);
  const size_t off = blk << kLogN;
  u64 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = src[off + t + 256 * i];
  fwd_core<ARITH>(v, sm, tw, k, t, false, bk);
  // thread q holds X[16q+k]: transpose inside the wave's LDS region for coalesced stores
  {
    const int base = 17 * t;
#pragma unroll
    for (int i = 0; i < 16; ++i) sm[base + i] = canon<ARITH>(v[i], k);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int w = t >> 6, l = t & 63;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int e = 1024 * w + 64 * j + l;
    dst[off + e] = sm[pad(e)];
  }
}

// MUL: the point-wise product src (.) mul (both canonical, NTT order) is fused into the load.
template <int ARITH, bool MUL>
__global__ __launch_bounds__(kThreads) void k_ntt_inv4096(const u64 *src, const u64 *mul, u64 *dst,
                                                          const Tw64 *__restrict__ psi, const MC64 *__restrict__ mc,
                                                          int nm, int logn, int cm0, int cmcnt) {
  __shared__ u64 sm[kLdsWords];
  const int t = threadIdx.x;
  const int r = logn - kLogN;
  const size_t blk = block_of(blockIdx.x, nm, r, cm0, cmcnt);
  const size_t row = blk >> r;
  const Blk bk{r, (unsigned)(blk & ((1u << r) - 1u))};
  const int cm = (int)(row % (size_t)nm);
  const MC64 mcc = mc[cm];
  const Mod k = make_mod(mcc);
  const Tw64 *tw = psi + ((size_t)cm << logn);
  const size_t off = blk << kLogN;
  const int w = t >> 6, l = t & 63;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int e = 1024 * w + 64 * j + l;
    u64 x = src[off + e];
    if (MUL) x = ARITH >= 2 ? mul_lazy(x, mul[off + e], mcc.mu2, k) : barrett<u64>::mul(x, mul[off + e], k.p, mcc.mu);
    sm[pad(e)] = x;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  u64 v[16];
  {
    const int base = 17 * t;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = sm[base + i];
  }
  inv_core<ARITH>(v, sm, tw, mcc, k, t, bk);
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[off + t + 256 * i] = v[i];
}

// ---- launchers --------------------------------------------------------------------------
// The delta-form arithmetic needs delta < 2^32 (c < 2048 in params.hpp:94-97's c*2^21 - 1); every
// prime of the mirrored table qualifies, Shape::small_delta records it per context.
static inline bool fast_shape(const Shape &s) { return s.limb_bits == 64 && s.logn == kLogN; }

// ---- hand-scheduled assembly version (tools/gen_polymul_asm.py) ----------------------------
static const unsigned char kPolymulHsaco[] = {
#include "polymul4096_hsaco.inc"
};
enum AsmKind {
  kAsmPolymul = 0, kAsmPolymulNtt, kAsmFwd, kAsmInv, kAsmInvMul,   // 4096-word blocks, 256 threads (coefficient streams non-temporal)
  kAsmFwd2, kAsmInv2,                                                // n = 4096 stand-alone transforms, two rows per workgroup
  kAsmPolymul8k, kAsmPolymulNtt8k, kAsmFwd8k, kAsmInv8k,            // 8192-word rows, 512 threads
  kAsmPolymul16k, kAsmPolymulNtt16k, kAsmFwd16k, kAsmInv16k,        // 16384-word rows, 1024 threads
  kAsmFwd32k, kAsmInv32k, kAsmPolymulNtt32k,                         // 32768-word rows: ONE operand register-resident, 1024 threads
  kAsmFwd32kS, kAsmPolymulNtt32kS,                                   // ... the pair of the composed product: b' in the scratch layout [block][pair][thread]
  kAsmFwd16kX2, kAsmFwd8kX2,                                         // stand-alone forward transform, two rows of one modulus per workgroup
  kAsmPipe64k,                                                       // n = 65536: three-role pipeline kernel
  kAsmXcd64k, kAsmXcd32k,                                            // one launch of persistent workgroups, rows pinned to an XCD
  kAsmRow1024U32, kAsmRow2048U32, kAsmRow4096U32, kAsmRowFwd1024U32, kAsmRowFwd2048U32, kAsmRowFwd4096U32,
  kAsmRowInv1024U32, kAsmRowInv2048U32, kAsmRowInv4096U32,                                               // 32-bit limbs
  kAsmRow8U32, kAsmRowNtt8U32, kAsmRowFwd8U32, kAsmRowInv8U32,       // 32-bit limbs, n = 8: one lane per row
  kAsmRow128U16, kAsmRowNtt128U16, kAsmRowFwd128U16, kAsmRowInv128U16,   // 16-bit limbs
  kAsmFusedEnc2, kAsmFusedFmaFwd, kAsmFusedFmsInv, kAsmFusedFmaInv,      // transform-fused pipelines, n = 4096 (build_fused)
  kAsmPipe64kB,                                                      // n = 65536, operand b already transformed (build_pipe b_ntt)
  kAsmFused8kEnc2, kAsmFused8kFmaFwd, kAsmFused8kFmsInv, kAsmFused8kFmaInv,      // transform-fused pipelines, rows of 8192 words (build_fused_rows)
  kAsmFused16kEnc2, kAsmFused16kFmaFwd, kAsmFused16kFmsInv, kAsmFused16kFmaInv,  // ... of 16384 words
  kAsmFusedEnc2R, kAsmFusedFmaFwdR, kAsmFusedFmsInvR, kAsmFusedFmaInvR,  // ... of 4096 words on the ring-mode map (128 VGPRs, four workgroups per CU)
  kAsmFused32kFmsInv, kAsmFused32kFmaInv,                            // the inverse pipelines of a 32768-word row (build_row32k fms_inv / fma_inv)
  kAsmFwd32kI8, kAsmFused32kFmaFwdI8, kAsmFused32kEnc2I8,            // ... its forward transform / forward pipelines from a compact (int8) polynomial
  kAsmPolymulI1, kAsmPolymulI2,                                      // the n = 4096 product on incomplete transforms (1 / 2 stages dropped, incomplete.py)
  kAsmPipe64kI2, kAsmXcd64kI2, kAsmXcd32kI2,                         // ... the long-row plans with their block products on incomplete transforms (level 2)
  kAsmPolymul8kI2, kAsmPolymul16kI2,                                 // ... the row-resident products (rows.py build_row16k level 2)
  kAsmRow1024U64, kAsmRow2048U64, kAsmRow1024L0U64, kAsmRow2048L0U64,                                    // 64-bit limbs, one / two waves per row (rows1k.py):
  kAsmRowFwd1024U64, kAsmRowFwd2048U64, kAsmRowInv1024U64, kAsmRowInv2048U64,                            //   product (incomplete / complete transforms), transforms
  kAsmRowFmsInv1024U64, kAsmRowFmsInv2048U64, kAsmRowFmaInv1024U64, kAsmRowFmaInv2048U64,                //   INTT(b -+ a k)
  kAsmRowEnc2W1024U64, kAsmRowEnc2W2048U64, kAsmRowEnc2I81024U64, kAsmRowEnc2I82048U64,                  //   NTT(x) k + NTT(e), two results; words / int8 inputs
  kAsmRowFmaFwdW1024U64, kAsmRowFmaFwdW2048U64, kAsmRowFmaFwdI81024U64, kAsmRowFmaFwdI82048U64,          //   ... one result
  kAsmRow1024I2U32, kAsmRow2048I2U32, kAsmRow4096I2U32,                                                  // 32-bit limbs: the product on incomplete transforms
  kAsmRowFmsInv1024U32, kAsmRowFmsInv2048U32, kAsmRowFmsInv4096U32, kAsmRowFmaInv1024U32, kAsmRowFmaInv2048U32, kAsmRowFmaInv4096U32,   // 32-bit limbs: INTT(b -+ a k)
  kAsmRowEnc2W1024U32, kAsmRowEnc2W2048U32, kAsmRowEnc2W4096U32, kAsmRowEnc2I81024U32, kAsmRowEnc2I82048U32, kAsmRowEnc2I84096U32,     //   NTT(x) k + NTT(e), two results; words / int8
  kAsmRowFmaFwdW1024U32, kAsmRowFmaFwdW2048U32, kAsmRowFmaFwdW4096U32, kAsmRowFmaFwdI81024U32, kAsmRowFmaFwdI82048U32, kAsmRowFmaFwdI84096U32,   //   ... one result
  kAsmFwd32kSI2, kAsmPolymulNtt32kSI2,                               // 32768-word rows: the composed product's pair on incomplete transforms (b' stored two stages short)
  kAsmCount
};
static inline bool is8k(AsmKind k) { return (k >= kAsmPolymul8k && k <= kAsmInv8k) || k == kAsmPolymul8kI2; }
static inline bool is16k(AsmKind k) { return (k >= kAsmPolymul16k && k <= kAsmInv16k) || k == kAsmPolymul16kI2; }
static inline bool is32k(AsmKind k) { return (k >= kAsmFwd32k && k <= kAsmPolymulNtt32kS) || k == kAsmFwd32kSI2 || k == kAsmPolymulNtt32kSI2; }
static const char *const kAsmNames[kAsmCount] = {
    "nflhip_polymul4096nt_asm", "nflhip_polymul_ntt4096_asm", "nflhip_ntt_fwd4096_asm", "nflhip_ntt_inv4096_asm", "nflhip_ntt_inv_mul4096_asm",
    "nflhip_ntt_fwd4096x2nt_asm", "nflhip_ntt_inv4096x2nt_asm",
    "nflhip_polymul8192_asm", "nflhip_polymul_ntt8192_asm", "nflhip_ntt_fwd8192_asm", "nflhip_ntt_inv8192_asm",
    "nflhip_polymul16384_asm", "nflhip_polymul_ntt16384_asm", "nflhip_ntt_fwd16384_asm", "nflhip_ntt_inv16384_asm",
    "nflhip_ntt_fwd32768_asm", "nflhip_ntt_inv32768_asm", "nflhip_polymul_ntt32768_asm",
    "nflhip_ntt_fwd32768s_asm", "nflhip_polymul_ntt32768s_asm",
    "nflhip_ntt_fwd16384x2_asm", "nflhip_ntt_fwd8192x2_asm",
    "nflhip_polymul_pipe65536nt_asm",
    "nflhip_polymul_xcd65536_asm", "nflhip_polymul_xcd32768_asm",
    "nflhip_row1024_u32_asm", "nflhip_row2048_u32_asm", "nflhip_row4096_u32_asm", "nflhip_row1024_fwd_u32_asm", "nflhip_row2048_fwd_u32_asm", "nflhip_row4096_fwd_u32_asm",
    "nflhip_row1024_inv_u32_asm", "nflhip_row2048_inv_u32_asm", "nflhip_row4096_inv_u32_asm",
    "nflhip_row8_u32_asm", "nflhip_row8_ntt_u32_asm", "nflhip_row8_fwd_u32_asm", "nflhip_row8_inv_u32_asm",
    "nflhip_row128_u16_asm", "nflhip_row128_ntt_u16_asm", "nflhip_row128_fwd_u16_asm", "nflhip_row128_inv_u16_asm",
    "nflhip_fused_enc2_4096_asm", "nflhip_fused_fma_fwd4096_asm", "nflhip_fused_fms_inv4096_asm", "nflhip_fused_fma_inv4096_asm",
    "nflhip_polymul_pipe65536ntb_asm",
    "nflhip_fused_enc2_8192_asm", "nflhip_fused_fma_fwd8192_asm", "nflhip_fused_fms_inv8192_asm", "nflhip_fused_fma_inv8192_asm",
    "nflhip_fused_enc2_16384_asm", "nflhip_fused_fma_fwd16384_asm", "nflhip_fused_fms_inv16384_asm", "nflhip_fused_fma_inv16384_asm",
    "nflhip_fused_enc2_4096r_asm", "nflhip_fused_fma_fwd4096r_asm", "nflhip_fused_fms_inv4096r_asm", "nflhip_fused_fma_inv4096r_asm",
    "nflhip_fused_fms_inv32768_asm", "nflhip_fused_fma_inv32768_asm",
    "nflhip_ntt_fwd32768i8_asm", "nflhip_fused_fma_fwd32768i8_asm", "nflhip_fused_enc2_32768i8_asm",
    "nflhip_polymul4096i1_asm", "nflhip_polymul4096i2_asm",
    "nflhip_polymul_pipe65536nti2_asm", "nflhip_polymul_xcd65536i2_asm", "nflhip_polymul_xcd32768i2_asm",
    "nflhip_polymul8192i2_asm", "nflhip_polymul16384i2_asm",
    "nflhip_row1024_u64_asm", "nflhip_row2048_u64_asm", "nflhip_row1024_l0_u64_asm", "nflhip_row2048_l0_u64_asm",
    "nflhip_row1024_fwd_u64_asm", "nflhip_row2048_fwd_u64_asm", "nflhip_row1024_inv_u64_asm", "nflhip_row2048_inv_u64_asm",
    "nflhip_row1024_fmsinv_u64_asm", "nflhip_row2048_fmsinv_u64_asm", "nflhip_row1024_fmainv_u64_asm", "nflhip_row2048_fmainv_u64_asm",
    "nflhip_row1024_enc2w_u64_asm", "nflhip_row2048_enc2w_u64_asm", "nflhip_row1024_enc2i8_u64_asm", "nflhip_row2048_enc2i8_u64_asm",
    "nflhip_row1024_fmafwdw_u64_asm", "nflhip_row2048_fmafwdw_u64_asm", "nflhip_row1024_fmafwdi8_u64_asm", "nflhip_row2048_fmafwdi8_u64_asm",
    "nflhip_row1024_i2_u32_asm", "nflhip_row2048_i2_u32_asm", "nflhip_row4096_i2_u32_asm",
    "nflhip_row1024_fmsinv_u32_asm", "nflhip_row2048_fmsinv_u32_asm", "nflhip_row4096_fmsinv_u32_asm",
    "nflhip_row1024_fmainv_u32_asm", "nflhip_row2048_fmainv_u32_asm", "nflhip_row4096_fmainv_u32_asm",
    "nflhip_row1024_enc2w_u32_asm", "nflhip_row2048_enc2w_u32_asm", "nflhip_row4096_enc2w_u32_asm",
    "nflhip_row1024_enc2i8_u32_asm", "nflhip_row2048_enc2i8_u32_asm", "nflhip_row4096_enc2i8_u32_asm",
    "nflhip_row1024_fmafwdw_u32_asm", "nflhip_row2048_fmafwdw_u32_asm", "nflhip_row4096_fmafwdw_u32_asm",
    "nflhip_row1024_fmafwdi8_u32_asm", "nflhip_row2048_fmafwdi8_u32_asm", "nflhip_row4096_fmafwdi8_u32_asm",
    "nflhip_ntt_fwd32768si2_asm", "nflhip_polymul_ntt32768si2_asm",
};
struct AsmKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn[kAsmCount] = {};
  std::once_flag once;
};
static AsmKernel g_asm[16];  // per device

static hipFunction_t asm_fn(AsmKind kind) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  AsmKernel &k = g_asm[dev];
  std::call_once(k.once, [&k] {  // contexts may be used from several host threads
    if (hipModuleLoadData(&k.mod, kPolymulHsaco) != hipSuccess) {
      k.mod = nullptr;
      (void)hipGetLastError();
      return;
    }
    for (int i = 0; i < kAsmCount; ++i)
      if (hipModuleGetFunction(&k.fn[i], k.mod, kAsmNames[i]) != hipSuccess) {
        k.fn[i] = nullptr;
        (void)hipGetLastError();
      }
  });
  return k.fn[kind];
}

// The ring-mode kernels (rows of 8192 / 16384 / 32768 words) read the twiddle table with its last four stages lane-major
// (DevTables::psi_lm, tools/gen_polymul_asm.py tw_base_lm); the 4096-word kernels read the natural table.
// -DNFLHIP_NATURAL_TWIDDLES + NFL_GEN_NATURAL_TWIDDLES=1 rebuild the natural-order variant of the former for same-box
// comparisons (tools/sessions/gpu_round3_l.sh).
#ifdef NFLHIP_NATURAL_TWIDDLES
#define PSI_LM(t) ((t).psi)
#else
#define PSI_LM(t) ((t).psi_lm)
#endif

// which n = 4096 product serves coefficient-form operands: 0 = complete transforms (nflhip_polymul4096nt_asm), 1 / 2 = that many
// stages dropped each way.  Default chosen by measurement (profiles/r06_incomplete_ab.txt); the test hook switches it per process.
#ifndef NFLHIP_POLYMUL_LEVEL
#define NFLHIP_POLYMUL_LEVEL 2
#endif
static std::atomic<int> g_polymul_level{NFLHIP_POLYMUL_LEVEL};
int polymul_level() { return g_polymul_level.load(); }   // (api.hip reads it ONCE per product: the launches of a chunked plan must agree)
extern "C" int nflhip_debug_polymul_level(int level) {   // include/nflhip_debug.h; returns the previous setting; level < 0 only reads
  const int old = g_polymul_level.load();
  if (level >= 0 && level <= 2) g_polymul_level.store(level);
  return old;
}

// every generated kernel takes (dst, src_a, src_b, psi, mc, nm, logn) and one workgroup per block of its size
static hipError_t launch_asm(AsmKind kind, const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a,
                             const uint64_t *b, size_t batch, hipStream_t st, int ny = 0) {
  // ny > 0: only the moduli [0, ny) of every polynomial (grid.y; rows stay nm apart) -- the delta-form prefix of a context whose
  // later moduli take the general family
  if (s.compiled_only || (!s.small_delta && !(ny > 0 && ny <= s.nm_small)) || s.nm > 65535) return hipErrorNotSupported;
  hipFunction_t fn = asm_fn(kind);
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn;
  } args = {c, a, b, is8k(kind) || is16k(kind) || is32k(kind) ? PSI_LM(t) : t.psi, t.mc, (int)s.nm, s.logn};
  if (kind == kAsmPolymulI1 || kind == kAsmPolymulI2) {   // (their own ModConst records: scale of the shorter inverse, 2^127 Barrett constant)
    args.mc = t.mc_inc[kind - kAsmPolymulI1];
    if (!args.mc || s.logn != kLogN) return hipErrorNotSupported;
  }
  if (kind == kAsmPolymul8kI2 || kind == kAsmPolymul16kI2) {
    args.mc = t.mc_inc[1];
    if (!args.mc || s.logn != (kind == kAsmPolymul8kI2 ? kLogN + 1 : kLogN + 2)) return hipErrorNotSupported;   // whole rows only: the scale is the row's
  }
  if (kind == kAsmFwd32kSI2 || kind == kAsmPolymulNtt32kSI2) {
    args.mc = t.mc_inc[1];
    if (!args.mc || s.logn != kLogN + 3) return hipErrorNotSupported;
  }
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  // one 256-thread workgroup per 4096-word block, or one 1024-thread workgroup per 16384-word block
  const int blog = is32k(kind) ? kLogN + 3 : is16k(kind) ? kLogN + 2 : (is8k(kind) ? kLogN + 1 : kLogN);  // (kAsmPolymulNt: 4096-word blocks)
  if (s.logn < blog) return hipErrorNotSupported;
  const size_t gx = batch << (s.logn - blog);
  if (gx > 0x7fffffffull) return hipErrorInvalidValue;
  return hipModuleLaunchKernel(fn, (unsigned)gx, (unsigned)(ny > 0 ? ny : s.nm), 1, is16k(kind) || is32k(kind) ? 1024 : (is8k(kind) ? 512 : kThreads), 1, 1, 0, st,
                               nullptr, extra);
}

// n = 4096 stand-alone transforms of a batch: two polynomials (same modulus) per workgroup, like the a / b operands of the
// fused product -- twice the bytes in flight per workgroup and one set of twiddle loads for both rows.
static hipError_t launch_asm_x2(AsmKind kind, const Shape &s, const DevTables &t, uint64_t *dst, const uint64_t *src,
                                size_t batch, hipStream_t st, int ny = 0) {
  // (the same for rows of 16384 / 8192 words: the forward half of their fused products without the product)
  const bool k16 = kind == kAsmFwd16kX2, k8 = kind == kAsmFwd8kX2;
  if (s.compiled_only || (!s.small_delta && !(ny > 0 && ny <= s.nm_small)) || s.logn != (k16 ? kLogN + 2 : k8 ? kLogN + 1 : kLogN) || s.nm > 65535) return hipErrorNotSupported;
  if (batch < 2 || batch > 0x7fffffffull) return hipErrorNotSupported;
  hipFunction_t fn = asm_fn(kind);
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn, count;
  } args = {dst, src, nullptr, k16 || k8 ? PSI_LM(t) : t.psi, t.mc, (int)s.nm, s.logn, (int)batch};
  size_t size = 52;
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((batch + 1) / 2), (unsigned)(ny > 0 ? ny : s.nm), 1, k16 ? 1024 : k8 ? 512 : kThreads, 1, 1, 0, st, nullptr, extra);
}

// transform-fused pipelines (tools/gen_polymul_asm.py build_fused, kernarg ARGS_FUSED): one 256-thread workgroup per
// (batch element, modulus); the intermediate polynomials of `x.ntt_pow_phi(); r = x * k + e` / `(b - a * s).invntt_pow_invphi()`
// (tests/nfllib_demo_main_op.cpp:26-58) never reach HBM
static std::atomic<int> g_fused_grid{0};
extern "C" void nflhip_debug_fused_grid(int mode) { g_fused_grid.store(mode); }  // include/nflhip_debug.h
hipError_t launch_fused_asm_u64(const Shape &s, const DevTables &t, int kind, uint64_t *out0, uint64_t *out1,
                                const void *const *x, const unsigned *xstride, const int *xfmt, const void *const *k,
                                const unsigned *kstride, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || s.logn < kLogN || s.logn > kLogN + 3 || s.compiled_only || !s.small_delta || s.nm > 65535 || kind < 0 || kind > 3)
    return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  if (batch > 0x7fffffffull) return hipErrorInvalidValue;
  if (s.logn == kLogN + 3) {
    // rows of 32768 words: the inverse pipelines only (one operand register-resident, b and the key streamed through the idle
    // twiddle ring: build_row32k), dense a / b, the key one polynomial for the batch or one per element
    if (kind < 2 || xstride[0] != 1 || xstride[1] != 1 || kstride[0] > 1 || xfmt[0] || xfmt[1]) return hipErrorNotSupported;
    hipFunction_t fn32 = asm_fn(kind == 2 ? kAsmFused32kFmsInv : kAsmFused32kFmaInv);
    if (!fn32) return hipErrorNotSupported;
    struct {
      void *c;
      const void *a, *b, *psi, *mc;
      int nm, logn;
      const void *k;
      int kstride, pad;
    } a32 = {out0, x[0], x[1], PSI_LM(t), t.mc, (int)s.nm, s.logn, k[0], (int)kstride[0], 0};
    static_assert(sizeof(a32) == 64, "kernarg layout of nflhip_fused_*_inv32768_asm (ARGS_STD + key pointer + stride flag)");
    size_t size32 = 60;
    void *extra32[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a32, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size32, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn32, (unsigned)batch, (unsigned)s.nm, 1, 1024, 1, 1, 0, st, nullptr, extra32);
  }
  // rows of 4096 words: 256 threads on the pair-mode map; 8192 / 16384: the row-resident ring-mode map, 512 / 1024 threads,
  // lane-major twiddle copy
  const int rows_log = s.logn - kLogN;
  const int forced = g_fused_grid.load(std::memory_order_relaxed);
  // rows of 4096 words have two register maps: pair mode (168 VGPRs, two interleaved butterflies, three workgroups per CU)
  // and ring mode (128 VGPRs, one butterfly at a time, four per CU).  Measured same-box (profiles/r04_ring_vs_pair_4096.txt):
  // the inverse pipelines -- 70 % VALU, the rest exposed operand latency -- gain 4 % from the fourth workgroup; the forward
  // ones are VALU-bound and keep pair mode.  Mode 3 of the debug hook swaps the choice (A/B runs).
  const bool ring4k = rows_log == 0 && ((kind >= 2) != (forced == 3));
  hipFunction_t fn = ring4k ? asm_fn((AsmKind)(kAsmFusedEnc2R + kind))
                            : asm_fn((AsmKind)((rows_log == 0 ? kAsmFusedEnc2 : rows_log == 1 ? kAsmFused8kEnc2 : kAsmFused16kEnc2) + kind));
  if (!fn) return hipErrorNotSupported;
  const int nx = kind == 0 ? 3 : 2, nk = kind == 0 ? 2 : 1;
  struct {
    void *out0, *out1;
    const void *x[3], *k[2], *psi, *mc;
    int nm, logn, fmt;
    unsigned sx[3], sk[2], count, magic;
  } args = {};
  static_assert(sizeof(args) == 112, "kernarg layout of nflhip_fused_*_asm (ARGS_FUSED)");
  args.out0 = out0;
  args.out1 = out1;
  for (int i = 0; i < nx; ++i) {
    // (the stride multiplies the batch index in 32 bits inside the kernel)
    if ((uint64_t)xstride[i] * (batch - 1) > 0xffffffffull || xfmt[i] < 0 || xfmt[i] > 3 || (kind >= 2 && xfmt[i])) return hipErrorInvalidValue;
    args.x[i] = x[i];
    args.sx[i] = xstride[i];
    args.fmt |= xfmt[i] << (4 * i);
  }
  for (int i = 0; i < nk; ++i) {
    if ((uint64_t)kstride[i] * (batch - 1) > 0xffffffffull) return hipErrorInvalidValue;
    args.k[i] = k[i];
    args.sk[i] = kstride[i];
  }
  args.psi = rows_log || ring4k ? PSI_LM(t) : t.psi;
  args.mc = t.mc;
  args.nm = (int)s.nm;
  args.logn = s.logn;
  args.count = (unsigned)batch;
  // forward kinds with more than one modulus: the nm rows of a batch element back to back on one XCD (1-D grid, the kernel
  // deals the workgroups itself), so that compact inputs -- one copy for all moduli -- come from HBM once
  const size_t groups = (batch + 7) / 8, wgs = groups * 8 * s.nm;
  const bool fits = wgs <= 0x7fffffffull && groups * s.nm < (0xffffffffull / s.nm);
  // (rows of 8192 / 16384 words keep the 2-D grid by default: with the nm rows of an element on one XCD that L2 holds nm
  // twiddle tables and the key rows of nm moduli at once -- measured at 16384 x 8: encrypt traffic 1.24x -> 1.32x)
  const bool remap = fits && forced != 1 && (forced == 2 || (rows_log == 0 && kind < 2 && s.nm > 1 && args.fmt != 0));
  args.magic = remap ? (unsigned)(0x100000000ull / s.nm + 1) : 0u;
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  if (remap) return hipModuleLaunchKernel(fn, (unsigned)wgs, 1, 1, (unsigned)(kThreads << rows_log), 1, 1, 0, st, nullptr, extra);
  return hipModuleLaunchKernel(fn, (unsigned)batch, (unsigned)s.nm, 1, (unsigned)(kThreads << rows_log), 1, 1, 0, st, nullptr, extra);
}

// rows of 32768 words, forward side (tools/gen_polymul_asm.py build_row32k fwd_i8 / fma_fwd_i8 / enc2_i8): a compact Gaussian
// polynomial (one signed byte per coefficient) -> the NTT words of every modulus; and out0 = NTT(x) k0 + e0' [, out1 = NTT(x) k1 +
// e1'] with x compact, the keys one polynomial each (NTT form) and e' ALREADY transformed words (what the first kernel wrote)
hipError_t launch_row32k_fwd_i8_u64(const Shape &s, const DevTables &t, uint64_t *dst, const void *x8, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || s.logn != kLogN + 3 || s.compiled_only || !s.small_delta || s.nm > 65535) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  if (batch > 0x7fffffffull) return hipErrorInvalidValue;
  hipFunction_t fn = asm_fn(kAsmFwd32kI8);
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn;
  } args = {dst, x8, nullptr, PSI_LM(t), t.mc, (int)s.nm, s.logn};
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)batch, (unsigned)s.nm, 1, 1024, 1, 1, 0, st, nullptr, extra);
}
hipError_t launch_row32k_fwd_fma_i8_u64(const Shape &s, const DevTables &t, uint64_t *out0, uint64_t *out1, const void *x8,
                                        const uint64_t *k0, const uint64_t *e0p, const uint64_t *k1, const uint64_t *e1p, size_t batch,
                                        hipStream_t st) {
  if (s.limb_bits != 64 || s.logn != kLogN + 3 || s.compiled_only || !s.small_delta || s.nm > 65535) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  if (batch > 0x7fffffffull) return hipErrorInvalidValue;
  hipFunction_t fn = asm_fn(out1 ? kAsmFused32kEnc2I8 : kAsmFused32kFmaFwdI8);
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn;
    const void *k0, *k1, *e1p;
    void *out1;
  } args = {out0, x8, e0p, PSI_LM(t), t.mc, (int)s.nm, s.logn, k0, out1 ? k1 : k0, out1 ? e1p : e0p, out1 ? out1 : out0};
  static_assert(sizeof(args) == 80, "kernarg layout of nflhip_fused_{fma_fwd,enc2_}32768i8_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)batch, (unsigned)s.nm, 1, 1024, 1, 1, 0, st, nullptr, extra);
}

// n = 65536: one launch of the three-role kernel (tools/gen_polymul_asm.py build_pipe): fused block products of `cnt_v`
// polynomials whose operands already went through the forward streaming pass (a_v, b_v -> c_v), the forward streaming
// pass of `cnt_f` polynomials (fa_src -> fa_dst, fb_src -> fb_dst) and the inverse streaming pass of `cnt_i`
// polynomials in place (inv).  Counts may be zero.
hipError_t launch_polymul_pipe64k_u64(const Shape &s, const DevTables &t, uint64_t *c_v, const uint64_t *a_v,
                                      const uint64_t *b_v, int cnt_v, const uint64_t *fa_src, uint64_t *fa_dst,
                                      const uint64_t *fb_src, uint64_t *fb_dst, int cnt_f, uint64_t *inv, int cnt_i,
                                      hipStream_t st, bool b_is_ntt, int level) {
  if (s.limb_bits != 64 || s.logn != 16 || s.compiled_only || !s.small_delta || s.nm > 65535) return hipErrorNotSupported;
  // (coefficient loads / stores carry `nt`: they pass through the L2 once, the twiddle tables stay resident: measured +3 %)
  // b_is_ntt: b_v is the caller's transformed operand (canonical words), read block-wise as it lies; no forward role for it
  // level 2 (coefficient-form operands only): the block products run on incomplete transforms, and the streaming inverse role
  // folds in (n / 4)^-1 from the level-2 records -- every launch of one product takes the same level
  const bool inc = level == 2 && !b_is_ntt && t.mc_inc[1];
  hipFunction_t fn = asm_fn(b_is_ntt ? kAsmPipe64kB : inc ? kAsmPipe64kI2 : kAsmPipe64k);
  if (!fn) return hipErrorNotSupported;
  const int mx = cnt_v > cnt_f ? (cnt_v > cnt_i ? cnt_v : cnt_i) : (cnt_f > cnt_i ? cnt_f : cnt_i);
  if (mx <= 0) return hipSuccess;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn;
    int cnt_v, cnt_f, cnt_i, remap_gx;
    const void *fa_src;
    void *fa_dst;
    const void *fb_src;
    void *fb_dst, *inv;
    unsigned remap_per, remap_magic;
  } args = {c_v, a_v, b_v, t.psi, inc ? t.mc_inc[1] : t.mc, (int)s.nm, s.logn, cnt_v, cnt_f, cnt_i, 0, fa_src, fa_dst, fb_src, fb_dst, inv, 0u, 0u};
  static_assert(sizeof(args) == 112, "kernarg layout of nflhip_polymul_pipe65536_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  // per polynomial row: 16 block products + 3 x 4 streaming workgroups (2 x 4 when b needs no forward pass)
  const size_t gx = (size_t)mx * (b_is_ntt ? 24 : 28);
  if (gx > 0x7fffffffull) return hipErrorInvalidValue;
#ifndef NFLHIP_NO_PIPE_REMAP
  // modulus-major units in contiguous ranges per XCD slot (see build_pipe): every twiddle table is then fetched by ~1.3 of
  // the 8 private L2s instead of all 8.  Needs units divisible by 8 and the kernel's one-multiply division by gx exact.
  const unsigned long long units = (unsigned long long)gx * s.nm;
  if (units % 8 == 0 && units * gx < (1ull << 32)) {
    args.remap_gx = (int)gx;
    args.remap_per = (unsigned)(units / 8);
    args.remap_magic = (unsigned)((1ull << 32) / gx + 1);
  }
#endif
  return hipModuleLaunchKernel(fn, (unsigned)gx, (unsigned)s.nm, 1, kThreads, 1, 1, 0, st, nullptr, extra);
}

// n = 65536 / 32768, whole batch in ONE launch of persistent workgroups (tools/gen_polymul_asm.py fused_header): the three
// roles of a row run on one XCD and hand the intermediates over through that XCD's L2.  `work` is device memory of at
// least xcd_plan_bytes(); it is (re)initialised here, on `st`.
static std::atomic<unsigned long long> g_xcd_launches{0};
extern "C" unsigned long long nflhip_debug_xcd_launches(void) { return g_xcd_launches.load(); }  // include/nflhip_debug.h
// test / profiling hook: a device buffer (32 domains x 65536 records x 16 bytes) into which every role of the NEXT one-launch
// products writes {ticket | kind << 28, t0 = workgroup free, t1 = inputs ready, t2 = done} (low words of s_memtime); nullptr = off
static std::atomic<void *> g_xcd_trace{nullptr};
extern "C" void nflhip_debug_xcd_trace(void *device_buffer) { g_xcd_trace.store(device_buffer); }
__global__ void k_xcd_reset(uint4 *ctl) {   // block 0: the header; block d + 1: record d at byte 4096 + 69632 d (2 KiB each)
  uint4 *p = blockIdx.x == 0 ? ctl : ctl + (4096 + (size_t)(blockIdx.x - 1) * 0x11000) / 16;
  p[threadIdx.x] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && (threadIdx.x == 8 || threadIdx.x == 9)) p[threadIdx.x] = make_uint4(~0u, ~0u, ~0u, ~0u);
  // (bytes 128 .. 159: one free mask of 32 scratch slots per XCD -- the pooled plan)
}
struct XcdPlan {
  int rlog, wgs, dlog;
  unsigned magic;
  size_t ctl_bytes, slot_bytes, total;
};
static bool xcd_plan(const Shape &s, size_t batch, XcdPlan *p) {
  if (s.limb_bits != 64 || (s.logn != 16 && s.logn != 15) || s.compiled_only || !s.small_delta || s.nm > 65535) return false;
  // the kernel derives a row's XCD from the hardware XCC id: it needs the whole 8-XCD device (no compute partition)
  static int cus[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (cus[dev] != 256) return false;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (batch < 2 || rows < 8 || rows > 0x0fffffffull) return false;  // every XCD serves rows xcd, xcd + 8, ...
  const bool pow2 = (batch & (batch - 1)) == 0;
  if (!pow2 && rows * batch >= (1ull << 32)) return false;  // the kernel divides row numbers by the batch with one multiply
  p->magic = (unsigned)(pow2 ? (1ull << 32) / batch : (1ull << 32) / batch + 1);
  p->rlog = 3;   // 2^rlog rows in flight per scheduling domain
  p->dlog = 2;   // 2^dlog scheduling domains per XCD (measured: 1 domain 13.8 k, 2: 23.7 k, 4: 26.1 k products/s at n = 65536)
  p->wgs = 768;  // persistent workgroups: three per CU
  if (rows < (8ull << p->dlog)) return false;
  p->ctl_bytes = 4096 + ((size_t)8 << p->dlog) * 0x11000;  // word 0: next row; one 256 B scheduler record per domain, 68 KiB apart, from byte 4096
  p->slot_bytes = (size_t)rows * (s.n * 8);                 // per operand: the scratch mirrors the batch (every row its own scratch rows)
  p->total = p->ctl_bytes + 2 * p->slot_bytes;
  return true;
}
size_t xcd_plan_bytes(const Shape &s, size_t batch) {
  XcdPlan p;
  return xcd_plan(s, batch, &p) ? p.total : 0;
}
hipError_t launch_polymul_xcd_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                  size_t batch, void *work, hipStream_t st, int level) {
  XcdPlan p;
  if (!xcd_plan(s, batch, &p)) return hipErrorNotSupported;
  const bool inc = level == 2 && t.mc_inc[1];
  hipFunction_t fn = asm_fn(s.logn == 16 ? (inc ? kAsmXcd64kI2 : kAsmXcd64k) : (inc ? kAsmXcd32kI2 : kAsmXcd32k));
  if (!fn) return hipErrorNotSupported;
  // fresh counters: word block 0 (workgroups that joined, per XCD) and the first KiB of every domain's record
  hipLaunchKernelGGL(k_xcd_reset, dim3((8u << p.dlog) + 1), dim3(128), 0, st, (uint4 *)work);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  char *w = (char *)work;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    int nm, logn;
    int rows, batch;
    unsigned magic;
    int d, rlog, jmax, spin, inv;
    void *scr_a, *scr_b, *ctl, *trace;
  } args = {c, a, b, t.psi, inc ? t.mc_inc[1] : t.mc, (int)s.nm, s.logn, (int)(batch * s.nm), (int)batch, p.magic, p.dlog, p.rlog, 0,
            1 << 22, 0, w + p.ctl_bytes, w + p.ctl_bytes + p.slot_bytes, w, g_xcd_trace.load()};
  static_assert(sizeof(args) == 112, "kernarg layout of nflhip_polymul_xcd*_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  g_xcd_launches.fetch_add(1);
  return hipModuleLaunchKernel(fn, (unsigned)p.wgs, 1, 1, kThreads, 1, 1, 0, st, nullptr, extra);
}

// 32-bit limbs, n = 1024 / 2048 / 4096: the fused product with one / two / four waves per row (tools/gen_row1024_u32_asm.py)
hipError_t launch_row1024_u32_asm(const Shape &s, const DevTables &t, int mode, uint32_t *c, const uint32_t *a,
                                  const uint32_t *b, size_t batch, hipStream_t st) {
  // mode (as launch_row1024_u32): 0 fused product, 1 product with b already transformed (n = 8 only), 2 forward
  // (canonical NTT-form words out), 3 inverse
  if (s.limb_bits == 32 && s.logn == 3 && mode >= 0 && mode <= 3 && !s.compiled_only) {
    // n = 8 (the reference's (8, 60, uint32_t) config): one LANE per row, 256 rows per workgroup (tools/gen_row8_u32_asm.py)
    const unsigned long long rows8 = (unsigned long long)batch * s.nm;
    if (rows8 == 0) return hipSuccess;
    if (rows8 * s.nm >= (1ull << 32)) return hipErrorNotSupported;
    hipFunction_t f8 = asm_fn((AsmKind)(kAsmRow8U32 + mode));
    if (!f8) return hipErrorNotSupported;
    struct {
      void *c;
      const void *a, *b, *psi, *mc;
      unsigned nm, magic;
      unsigned long long rows;
    } a8 = {c, a, b, t.psi, t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), rows8};
    size_t sz8 = sizeof(a8);
    void *ex8[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a8, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz8, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(f8, (unsigned)((rows8 + 255) / 256), 1, 1, 256, 1, 1, 0, st, nullptr, ex8);
  }
  if (s.limb_bits != 32 || s.logn < 10 || s.logn > 12 || s.compiled_only || (mode != 0 && mode != 2 && mode != 3)) return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;  // (row mod nm is one multiply in the kernel)
  const bool inc = mode == 0 && g_polymul_level.load() == 2 && t.mc_inc[1];   // coefficient form in and out: incomplete transforms
  const int first = mode == 0 ? (inc ? kAsmRow1024I2U32 : kAsmRow1024U32) : (mode == 2 ? kAsmRowFwd1024U32 : kAsmRowInv1024U32);
  hipFunction_t fn = asm_fn((AsmKind)(first + (s.logn - 10)));
  const unsigned rpb = 4u >> (s.logn - 10);  // rows per 256-thread workgroup
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    unsigned nm, magic;
    unsigned long long rows;
  } args = {c, a, b, t.psi, inc ? t.mc_inc[1] : t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), rows};
  static_assert(sizeof(args) == 56, "kernarg layout of nflhip_row1024_u32_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}

// 64-bit limbs, n = 1024 / 2048: the fused product (on incomplete transforms unless nflhip_debug_polymul_level says 0) and the
// stand-alone transforms, one wave / two waves per row (tools/asmgen/rows1k.py); hipErrorNotSupported: the compiled k_row<Pol64, ...>
hipError_t launch_row1024_u64_asm(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a,
                                  const uint64_t *b, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || s.logn < 10 || s.logn > 11 || s.compiled_only || !s.small_delta || (mode != 0 && mode != 2 && mode != 3))
    return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;  // (row mod nm is one multiply in the kernel)
  const bool inc = mode == 0 && g_polymul_level.load() == 2 && t.mc_inc[1];
  const int first = mode == 0 ? (inc ? kAsmRow1024U64 : kAsmRow1024L0U64) : (mode == 2 ? kAsmRowFwd1024U64 : kAsmRowInv1024U64);
  hipFunction_t fn = asm_fn((AsmKind)(first + (s.logn - 10)));
  if (!fn) return hipErrorNotSupported;
  const unsigned rpb = 4u >> (s.logn - 10);  // rows per 256-thread workgroup
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    unsigned nm, magic;
    unsigned long long rows;
  } args = {c, a, b, t.psi, inc ? t.mc_inc[1] : t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), rows};
  static_assert(sizeof(args) == 56, "kernarg layout of nflhip_row1024_u64_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}

// ... and the transform-fused pipelines on those rows (rows1k.py build_row1k_fwd_fma / build_row1k_fma_inv): operands of format words
// or int8, strides 0 / 1; hipErrorNotSupported: the compiled k_row_fwd_fma / k_row_fma_inv (kernels_wave.hip)
hipError_t launch_row_fwd_fma_u64_asm(const Shape &s, const DevTables &t, int format, uint64_t *out0, uint64_t *out1, const void *x, unsigned xs,
                                      const uint64_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint64_t *k1, unsigned k1s,
                                      const void *e1, unsigned e1s, size_t batch, hipStream_t st) {
  if (g_fused_grid.load(std::memory_order_relaxed) == 4) return hipErrorNotSupported;   // (nflhip_debug_fused_grid: the compiled one-pass template instead)
  if (s.limb_bits != 64 || s.logn < 10 || s.logn > 11 || s.compiled_only || !s.small_delta || (format != 0 && format != 1)) return hipErrorNotSupported;
  if (xs > 1 || k0s > 1 || e0s > 1 || (out1 && (k1s > 1 || e1s > 1))) return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;
  const int first = out1 ? (format == 0 ? kAsmRowEnc2W1024U64 : kAsmRowEnc2I81024U64) : (format == 0 ? kAsmRowFmaFwdW1024U64 : kAsmRowFmaFwdI81024U64);
  hipFunction_t fn = asm_fn((AsmKind)(first + (s.logn - 10)));
  if (!fn) return hipErrorNotSupported;
  const unsigned rpb = 4u >> (s.logn - 10);
  struct {
    void *out0, *out1;
    const void *x, *psi, *mc;
    unsigned nm, magic;
    const void *k0, *e0, *k1, *e1;
    unsigned long long rows;
    unsigned xs, k0s, e0s, k1s, e1s, pad;
  } args = {out0, out1, x, t.psi, t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), k0, e0, out1 ? k1 : k0, out1 ? e1 : e0,
            rows, xs, k0s, e0s, out1 ? k1s : 0u, out1 ? e1s : 0u, 0u};
  static_assert(sizeof(args) == 112, "kernarg layout of nflhip_row*_enc2*_u64_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}
hipError_t launch_row_fma_inv_u64_asm(const Shape &s, const DevTables &t, int subtract, uint64_t *c, const uint64_t *a, const uint64_t *key,
                                      int kstride, const uint64_t *b, size_t batch, hipStream_t st) {
  if (g_fused_grid.load(std::memory_order_relaxed) == 4) return hipErrorNotSupported;   // (nflhip_debug_fused_grid: the compiled one-pass template instead)
  if (s.limb_bits != 64 || s.logn < 10 || s.logn > 11 || s.compiled_only || !s.small_delta || kstride < 0 || kstride > 1) return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;
  hipFunction_t fn = asm_fn((AsmKind)((subtract ? kAsmRowFmsInv1024U64 : kAsmRowFmaInv1024U64) + (s.logn - 10)));
  if (!fn) return hipErrorNotSupported;
  const unsigned rpb = 4u >> (s.logn - 10);
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    unsigned nm, magic;
    unsigned long long rows;
    const void *key;
    unsigned kstride, pad;
  } args = {c, a, b, t.psi, t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), rows, key, (unsigned)kstride, 0u};
  static_assert(sizeof(args) == 72, "kernarg layout of nflhip_row*_fm?inv_u64_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}

// 32-bit limbs, n = 1024 / 2048 / 4096: the transform-fused pipelines (tools/gen_row1024_u32_asm.py build_fwd_fma / build_fma_inv): operands
// of format words or int8, strides 0 / 1; hipErrorNotSupported: the compiled k_row_fwd_fma / k_row_fma_inv (kernels_wave.hip).  The forward
// kinds reduce x k + e (lazily reduced x, e) with the base multiplication's Barrett step: they read the level-2 records
hipError_t launch_row_fwd_fma_u32_asm(const Shape &s, const DevTables &t, int format, uint32_t *out0, uint32_t *out1, const void *x, unsigned xs,
                                      const uint32_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint32_t *k1, unsigned k1s,
                                      const void *e1, unsigned e1s, size_t batch, hipStream_t st) {
  if (g_fused_grid.load(std::memory_order_relaxed) == 4) return hipErrorNotSupported;   // (nflhip_debug_fused_grid: the compiled one-pass template instead)
  if (s.limb_bits != 32 || s.logn < 10 || s.logn > 12 || s.compiled_only || !t.mc_inc[1] || (format != 0 && format != 1)) return hipErrorNotSupported;
  if (xs > 1 || k0s > 1 || e0s > 1 || (out1 && (k1s > 1 || e1s > 1))) return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;
  const int first = out1 ? (format == 0 ? kAsmRowEnc2W1024U32 : kAsmRowEnc2I81024U32) : (format == 0 ? kAsmRowFmaFwdW1024U32 : kAsmRowFmaFwdI81024U32);
  hipFunction_t fn = asm_fn((AsmKind)(first + (s.logn - 10)));
  if (!fn) return hipErrorNotSupported;
  const unsigned rpb = 4u >> (s.logn - 10);
  struct {
    void *out0, *out1;
    const void *x, *psi, *mc;
    unsigned nm, magic;
    const void *k0, *e0, *k1, *e1;
    unsigned long long rows;
    unsigned xs, k0s, e0s, k1s, e1s, pad;
  } args = {out0, out1, x, t.psi, t.mc_inc[1], (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), k0, e0, out1 ? k1 : k0,
            out1 ? e1 : e0, rows, xs, k0s, e0s, out1 ? k1s : 0u, out1 ? e1s : 0u, 0u};
  static_assert(sizeof(args) == 112, "kernarg layout of nflhip_row*_enc2*_u32_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}
hipError_t launch_row_fma_inv_u32_asm(const Shape &s, const DevTables &t, int subtract, uint32_t *c, const uint32_t *a, const uint32_t *key,
                                      int kstride, const uint32_t *b, size_t batch, hipStream_t st) {
  if (g_fused_grid.load(std::memory_order_relaxed) == 4) return hipErrorNotSupported;   // (nflhip_debug_fused_grid: the compiled one-pass template instead)
  if (s.limb_bits != 32 || s.logn < 10 || s.logn > 12 || s.compiled_only || kstride < 0 || kstride > 1) return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows * s.nm >= (1ull << 32)) return hipErrorNotSupported;
  hipFunction_t fn = asm_fn((AsmKind)((subtract ? kAsmRowFmsInv1024U32 : kAsmRowFmaInv1024U32) + (s.logn - 10)));
  if (!fn) return hipErrorNotSupported;
  const unsigned rpb = 4u >> (s.logn - 10);
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    unsigned nm, magic;
    unsigned long long rows;
    const void *key;
    unsigned kstride, pad;
  } args = {c, a, b, t.psi, t.mc, (unsigned)s.nm, s.nm == 1 ? 0u : (unsigned)((1ull << 32) / s.nm + 1), rows, key, (unsigned)kstride, 0u};
  static_assert(sizeof(args) == 72, "kernarg layout of nflhip_row*_fm?inv_u32_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + rpb - 1) / rpb), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}

// 16-bit limbs, n = 128 (the reference's (128, 14, uint16_t) config): the fused product, eight rows per wave
// (tools/gen_row128_u16_asm.py)
hipError_t launch_row128_u16_asm(const Shape &s, const DevTables &t, int mode, uint16_t *c, const uint16_t *a,
                                 const uint16_t *b, size_t batch, hipStream_t st) {
  // mode: 0 fused product, 1 product with b already transformed, 2 forward (canonical NTT-form words out), 3 inverse
  if (s.limb_bits != 16 || s.logn != 7 || s.compiled_only || (s.nm & (s.nm - 1)) != 0 || mode < 0 || mode > 3)
    return hipErrorNotSupported;
  const unsigned long long rows = (unsigned long long)batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorNotSupported;
  hipFunction_t fn = asm_fn((AsmKind)(kAsmRow128U16 + mode));
  if (!fn) return hipErrorNotSupported;
  struct {
    void *c;
    const void *a, *b, *psi, *mc;
    unsigned nm, pad;
    unsigned long long rows;
  } args = {c, a, b, t.psi, t.mc, (unsigned)s.nm, 0u, rows};
  static_assert(sizeof(args) == 56, "kernarg layout of nflhip_row128_u16_asm");
  size_t size = sizeof(args);
  void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)((rows + 31) / 32), 1, 1, 256, 1, 1, 0, st, nullptr, extra);
}

hipError_t launch_polymul_blocks_asm_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a_in,
                                         const uint64_t *b_in, size_t batch, hipStream_t st, bool b_is_ntt) {
  if (s.limb_bits != 64 || s.logn < kLogN) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  return launch_asm(b_is_ntt ? kAsmPolymulNtt : kAsmPolymul, s, t, c, a_in, b_in, batch, st);
}

template <bool B_IS_NTT>
static hipError_t launch_polymul_v(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                   unsigned rows, hipStream_t st, int cm0 = 0, int cmcnt = 0) {
  const Tw64 *psi = (const Tw64 *)t.psi;
  const MC64 *mc = (const MC64 *)t.mc;
  const int nm = (int)s.nm;
  // delta-form arithmetic (two-bit fold, one-off quotient: the assembly kernel's formulation) needs delta < 2^32; any other
  // modulus takes the Harvey ranges with the generic Shoup product.  cmcnt > 0: `rows` counts only the moduli [cm0, cm0 + cmcnt)
  if (s.small_delta) hipLaunchKernelGGL((k_polymul4096<B_IS_NTT, 3, 2>), dim3(rows), dim3(kThreads), 0, st, c, a, b, psi, mc, nm, cm0, cmcnt);
  else hipLaunchKernelGGL((k_polymul4096<B_IS_NTT, 0, 2>), dim3(rows), dim3(kThreads), 0, st, c, a, b, psi, mc, nm, cm0, cmcnt);
  return hipGetLastError();
}

// A context whose moduli beyond a prefix have delta >= 2^32 (the reference's table from its 93rd 62-bit prime on, params.hpp:82-119):
// at degree 4096 the prefix rows keep the generated delta-form kernels (grid restricted to those moduli), the others take the
// general-modulus kernels -- two launches on the stream, disjoint rows.
static inline bool split_families(const Shape &s, size_t batch) {
  return s.limb_bits == 64 && s.logn == kLogN && !s.small_delta && s.nm_small > 0 && !s.compiled_only && batch > 0 &&
         batch * (s.nm - (size_t)s.nm_small) <= 0x7fffffffull;
}

static inline bool row16k_shape(const Shape &s) { return s.limb_bits == 64 && s.logn == kLogN + 2; }
static inline bool row8k_shape(const Shape &s) { return s.limb_bits == 64 && s.logn == kLogN + 1; }
// 32768-word rows: one operand register-resident in a 1024-thread workgroup (tools/gen_polymul_asm.py build_row32k).
static inline bool row32k_shape(const Shape &s) { return s.limb_bits == 64 && s.logn == kLogN + 3 && !s.compiled_only; }
hipError_t launch_row32k_u64(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a, const uint64_t *b,
                             size_t batch, hipStream_t st) {
  if (!row32k_shape(s)) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  // modes 4 / 5: the two halves of the composed product -- b' goes through the context's scratch in a layout only these two share;
  // 6 / 7: the same pair on incomplete transforms (b' two stages short and unreduced; BOTH launches of a product take the same pair)
  const AsmKind k = mode == 1 ? kAsmPolymulNtt32k : mode == 2 ? kAsmFwd32k : mode == 3 ? kAsmInv32k : mode == 4 ? kAsmFwd32kS :
                    mode == 6 ? kAsmFwd32kSI2 : mode == 7 ? kAsmPolymulNtt32kSI2 : kAsmPolymulNtt32kS;
  return launch_asm(k, s, t, c, a, b, batch, st);
}

hipError_t launch_polymul_fast_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                   int b_is_ntt, size_t batch, hipStream_t st) {
  const bool inc2 = !b_is_ntt && g_polymul_level.load() == 2 && t.mc_inc[1];   // coefficient form in and out: incomplete transforms
  if (row16k_shape(s))  // a 16384-word row fits one CU: the whole product is a single launch
    return batch == 0 ? hipSuccess : launch_asm(b_is_ntt ? kAsmPolymulNtt16k : inc2 ? kAsmPolymul16kI2 : kAsmPolymul16k, s, t, c, a, b, batch, st);
  if (row8k_shape(s))   // 8192-word rows: 512 threads, two rows per CU
    return batch == 0 ? hipSuccess : launch_asm(b_is_ntt ? kAsmPolymulNtt8k : inc2 ? kAsmPolymul8kI2 : kAsmPolymul8k, s, t, c, a, b, batch, st);
  if (row32k_shape(s) && b_is_ntt) return launch_row32k_u64(s, t, 1, c, a, b, batch, st);
  if (!fast_shape(s)) return hipErrorNotSupported;
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorInvalidValue;
  if (split_families(s, batch)) {
    const int ns = s.nm_small, level = b_is_ntt ? 0 : g_polymul_level.load();
    hipError_t e = hipErrorNotSupported;
    if (level == 1 || level == 2) e = launch_asm(level == 1 ? kAsmPolymulI1 : kAsmPolymulI2, s, t, c, a, b, batch, st, ns);
    if (e == hipErrorNotSupported) e = launch_asm(b_is_ntt ? kAsmPolymulNtt : kAsmPolymul, s, t, c, a, b, batch, st, ns);
    if (e == hipSuccess) {
      const unsigned rest = (unsigned)(batch * (s.nm - (size_t)ns));
      return b_is_ntt ? launch_polymul_v<true>(s, t, c, a, b, rest, st, ns, (int)s.nm - ns)
                      : launch_polymul_v<false>(s, t, c, a, b, rest, st, ns, (int)s.nm - ns);
    }
    if (e != hipErrorNotSupported) return e;
  }
  if (!b_is_ntt && s.logn == kLogN) {
    // coefficient form in AND out: the transforms may stay incomplete (tools/asmgen/incomplete.py) -- same words out
    const int level = g_polymul_level.load();
    if (level == 1 || level == 2) {
      const hipError_t e = launch_asm(level == 1 ? kAsmPolymulI1 : kAsmPolymulI2, s, t, c, a, b, batch, st);
      if (e != hipErrorNotSupported) return e;
    }
  }
  {
    const hipError_t e = launch_asm(b_is_ntt ? kAsmPolymulNtt : kAsmPolymul, s, t, c, a, b, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  return b_is_ntt ? launch_polymul_v<true>(s, t, c, a, b, (unsigned)rows, st)
                  : launch_polymul_v<false>(s, t, c, a, b, (unsigned)rows, st);
}

// inner 4096-word blocks of rows with logn >= 12 (used directly for n = 4096 and by the
// generic launch plans of kernels_generic.hip after / before their streaming outer passes)
hipError_t launch_inner_fwd_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t rows,
                                     hipStream_t st) {
  if (s.limb_bits != 64 || s.logn < kLogN) return hipErrorNotSupported;
  const size_t blocks = rows << (s.logn - kLogN);
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  if (rows % s.nm == 0) {  // the generated kernels index rows as (poly, modulus)
    hipError_t e = launch_asm_x2(kAsmFwd2, s, t, dst, src, rows / s.nm, st);
    if (e == hipErrorNotSupported) e = launch_asm(kAsmFwd, s, t, dst, src, nullptr, rows / s.nm, st);
    if (e != hipErrorNotSupported) return e;
  }
  const Tw64 *psi = (const Tw64 *)t.psi;
  const MC64 *mc = (const MC64 *)t.mc;
  if (rows % s.nm == 0 && split_families(s, rows / s.nm)) {
    const int ns = s.nm_small;
    hipError_t e = launch_asm_x2(kAsmFwd2, s, t, dst, src, rows / s.nm, st, ns);
    if (e == hipErrorNotSupported) e = launch_asm(kAsmFwd, s, t, dst, src, nullptr, rows / s.nm, st, ns);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_ntt_fwd4096<0>, dim3((unsigned)(rows / s.nm * (s.nm - (size_t)ns))), dim3(kThreads), 0, st, src, dst, psi, mc,
                         (int)s.nm, s.logn, ns, (int)s.nm - ns);
      return hipGetLastError();
    }
    if (e != hipErrorNotSupported) return e;
  }
  const dim3 g((unsigned)blocks), b(kThreads);
  if (s.small_delta) hipLaunchKernelGGL(k_ntt_fwd4096<3>, g, b, 0, st, src, dst, psi, mc, (int)s.nm, s.logn, 0, 0);
  else hipLaunchKernelGGL(k_ntt_fwd4096<0>, g, b, 0, st, src, dst, psi, mc, (int)s.nm, s.logn, 0, 0);
  return hipGetLastError();
}

hipError_t launch_inner_inv_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, const uint64_t *mul,
                                     uint64_t *dst, size_t rows, hipStream_t st) {
  if (s.limb_bits != 64 || s.logn < kLogN) return hipErrorNotSupported;
  const size_t blocks = rows << (s.logn - kLogN);
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  if (rows % s.nm == 0) {
    hipError_t e = mul ? hipErrorNotSupported : launch_asm_x2(kAsmInv2, s, t, dst, src, rows / s.nm, st);
    if (e == hipErrorNotSupported) e = launch_asm(mul ? kAsmInvMul : kAsmInv, s, t, dst, src, mul, rows / s.nm, st);
    if (e != hipErrorNotSupported) return e;
  }
  const Tw64 *psi = (const Tw64 *)t.psi;
  const MC64 *mc = (const MC64 *)t.mc;
  dim3 g((unsigned)blocks), b(kThreads);
  int cm0 = 0, cmcnt = 0;
  if (rows % s.nm == 0 && split_families(s, rows / s.nm)) {
    const int ns = s.nm_small;
    hipError_t e = mul ? hipErrorNotSupported : launch_asm_x2(kAsmInv2, s, t, dst, src, rows / s.nm, st, ns);
    if (e == hipErrorNotSupported) e = launch_asm(mul ? kAsmInvMul : kAsmInv, s, t, dst, src, mul, rows / s.nm, st, ns);
    if (e == hipSuccess) {
      cm0 = ns, cmcnt = (int)s.nm - ns;
      g = dim3((unsigned)(rows / s.nm * (size_t)cmcnt));
    } else if (e != hipErrorNotSupported) {
      return e;
    }
  }
#define NFLHIP_INV(A)                                                                                                  \
  if (mul) hipLaunchKernelGGL((k_ntt_inv4096<A, true>), g, b, 0, st, src, mul, dst, psi, mc, (int)s.nm, s.logn, cm0, cmcnt);      \
  else hipLaunchKernelGGL((k_ntt_inv4096<A, false>), g, b, 0, st, src, mul, dst, psi, mc, (int)s.nm, s.logn, cm0, cmcnt);
  if (s.small_delta) { NFLHIP_INV(2) }
  else { NFLHIP_INV(0) }
#undef NFLHIP_INV
  return hipGetLastError();
}

hipError_t launch_ntt_fwd_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st) {
  if (row16k_shape(s) || row8k_shape(s)) {
    if (batch == 0) return hipSuccess;
    const hipError_t e = launch_asm_x2(row16k_shape(s) ? kAsmFwd16kX2 : kAsmFwd8kX2, s, t, dst, src, batch, st);   // (batch >= 2)
    if (e != hipErrorNotSupported) return e;
    return launch_asm(row16k_shape(s) ? kAsmFwd16k : kAsmFwd8k, s, t, dst, src, nullptr, batch, st);
  }
  if (row32k_shape(s)) return launch_row32k_u64(s, t, 2, dst, src, nullptr, batch, st);
  if (!fast_shape(s)) return hipErrorNotSupported;
  return launch_inner_fwd_fast_u64(s, t, src, dst, batch * s.nm, st);
}

hipError_t launch_ntt_inv_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st) {
  if (row16k_shape(s)) return batch == 0 ? hipSuccess : launch_asm(kAsmInv16k, s, t, dst, src, nullptr, batch, st);
  if (row8k_shape(s)) return batch == 0 ? hipSuccess : launch_asm(kAsmInv8k, s, t, dst, src, nullptr, batch, st);
  if (row32k_shape(s)) return launch_row32k_u64(s, t, 3, dst, src, nullptr, batch, st);
  if (!fast_shape(s)) return hipErrorNotSupported;
  return launch_inner_inv_fast_u64(s, t, src, nullptr, dst, batch * s.nm, st);
}

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_fast() {}
hipError_t warm_fast(hipStream_t st) {
  (void)asm_fn(kAsmPolymul);   // ... and the module of the generated kernels (hipModuleLoadData + its function table)
  hipLaunchKernelGGL(k_warm_fast, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
