// kernels_crt.hip -- register-resident CRT lift / project for 64-bit limbs.
//
// Both directions are multiply-accumulate bound for many moduli (nm^2 word products per coefficient), so the
// default kernels (k_crt_lift64_mac / k_crt_project64_mac) are organised around v_mad_u64_u32 with NO carry
// handling in the inner loop: one factor of every product is cut into three 21-bit parts, the other is a 32-bit
// digit of a precomputed constant, so a 64-bit accumulator takes > 2000 products before it can overflow and the
// digits are only normalised once per coefficient.  The older word-by-word kernels below them remain as the
// fallback for shapes the tables do not cover.
//
// lift   (GMP::poly2mpz, gmp.hpp:183-209): X_i = sum_cm lifting[cm]*x(cm,i) mod Q.  The value in [0,Q)
//        is unique, so the device uses the small-quotient form
//            X_i = sum_cm (Q/p_cm) * ((x(cm,i) * (Q/p_cm)^-1) mod p_cm)     (< nm * Q)
//        followed by ceil(log2 nm) conditional subtractions of Q << k.  One thread per coefficient,
//        the multi-limb accumulator lives in VGPRs (compile-time limb buckets), the constants Q/p_cm and
//        Q << k are wave-uniform (scalar loads).
// project (GMP::mpz2poly, gmp.hpp:211-219): x(cm,i) = X_i mod p_cm by Horner over the 64-bit limbs,
//        r <- r*beta + limb with beta = 2^64 mod p_cm, all residues of one coefficient kept in VGPRs so each
//        limb is read once and the stores are coalesced.
#include "modarith64.h"

namespace nflhip {

static constexpr int kCrtStride = 36;  // limbs per row of the qhat / qsh tables (zero padded)

template <int LACC>
__global__ __launch_bounds__(256) void k_crt_lift64(uint64_t *out, const u64 *d, const MC64 *__restrict__ mc,
                                                    const u64 *__restrict__ qhat, const u64 *__restrict__ qsh, int logn,
                                                    int nm, int L, int rounds, size_t ncoef) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncoef) return;
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
  u64 acc[LACC];
#pragma unroll
  for (int k = 0; k < LACC; ++k) acc[k] = 0;
  for (int cm = 0; cm < nm; ++cm) {
    const MC64 c = mc[cm];
    const u64 x = d[((b * nm + cm) << logn) + i];
    const u64 y = mul_shoup<u64>(x, c.yinv, c.yinv_sh, c.p);
    const u64 *qh = qhat + (size_t)cm * kCrtStride;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < LACC; ++k) {  // acc += (Q/p_cm) * y
      const u64 q = qh[k];
      const u64 lo = q * y, hi = __umul64hi(q, y);
      u64 s = acc[k] + lo;
      u64 c1 = s < lo ? 1 : 0;
      s += carry;
      c1 += s < carry ? 1 : 0;
      acc[k] = s;
      carry = hi + c1;
    }
  }
  for (int sft = rounds - 1; sft >= 0; --sft) {  // acc < 2^rounds * Q
    const u64 *qs = qsh + (size_t)sft * kCrtStride;
    u64 tmp[LACC];
    u64 borrow = 0;
#pragma unroll
    for (int k = 0; k < LACC; ++k) {
      const u64 a = acc[k], q = qs[k];
      tmp[k] = a - q - borrow;
      borrow = (a < q || (a == q && borrow)) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < LACC; ++k) acc[k] = borrow ? acc[k] : tmp[k];
  }
  u64 *o = out + gid * (size_t)L;
#pragma unroll
  for (int k = 0; k < LACC; ++k)
    if (k < L) o[k] = acc[k];
}

// ---- lift, carry-free MAC form ------------------------------------------------------------------------
// y_cm = (x_cm * (Q/p_cm)^-1) mod p_cm < 2^62 is cut into y0 + y1*2^21 + y2*2^42; the table holds the 32-bit digits
// of (Q/p_cm) * 2^(21 j), j = 0..2 (each < Q, so 2L digits).  acc[k] (64-bit, weight 2^(32 k)) gathers
// sum_cm sum_j digit_j[cm][k] * y_j[cm] < nm * 3 * 2^53, far below 2^64 for nm <= 32.
static constexpr int kCrtStride32 = 2 * kCrtStride;  // 32-bit digits per table row

template <int L>
__global__ __launch_bounds__(256) void k_crt_lift64_mac(uint64_t *out, const u64 *d, const MC64 *__restrict__ mc,
                                                        const u32 *__restrict__ qparts, const u32 *__restrict__ qdig,
                                                        double inv_qtop, int logn, int nm, size_t ncoef) {
  __shared__ u64 sl[L <= 4 ? 1 : 256][9];
  const int t = threadIdx.x;
  const size_t coef0 = (size_t)blockIdx.x * 256, left = ncoef - coef0;
  const size_t gid = coef0 + ((size_t)t < left ? t : 0);  // surplus threads of the last block shadow thread 0 (no stores)
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
  constexpr int W = 2 * L;
  u64 acc[W];
#pragma unroll
  for (int k = 0; k < W; ++k) acc[k] = 0;
  for (int cm = 0; cm < nm; ++cm) {
    const MC64 c = mc[cm];
    const u64 x = d[((b * nm + cm) << logn) + i];
    const u64 y = mul_shoup<u64>(x, c.yinv, c.yinv_sh, c.p);
    const u32 y0 = (u32)y & 0x1fffffu, y1 = (u32)(y >> 21) & 0x1fffffu, y2 = (u32)(y >> 42);
    const u32 *q0 = qparts + (size_t)cm * 3 * kCrtStride32, *q1 = q0 + kCrtStride32, *q2 = q1 + kCrtStride32;
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = acc[k] + (u64)q0[k] * y0 + (u64)q1[k] * y1 + (u64)q2[k] * y2;
  }
  // normalise: S = sum_k acc[k] * 2^(32 k) as W + 2 32-bit digits (S < nm * Q)
  u32 dg[W + 2];
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const u64 t = acc[k] + carry;
    dg[k] = (u32)t;
    carry = t >> 32;
  }
  dg[W] = (u32)carry;
  dg[W + 1] = (u32)(carry >> 32);
  // t~ = floor(S/Q) or one less, from the top five digits in double precision (error < 2^-40, margin 2^-20):
  // S - t~*Q is in [0, 2Q), so ONE conditional subtraction finishes (instead of ceil(log2 nm) of them).
  double sd = 0.0;
#pragma unroll
  for (int j = W + 1; j >= W - 3; --j)
    if (j >= 0) sd = sd * 4294967296.0 + (double)dg[j];
  int tq = (int)floor(sd * inv_qtop - 0x1p-20);
  tq = tq < 0 ? 0 : tq;
  {
    u64 cy = 0;
    unsigned bw = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const u64 pr = (u64)qdig[k] * (u32)tq + cy;
      cy = pr >> 32;
      dg[k] = __builtin_subc(dg[k], (u32)pr, bw, &bw);
    }
    dg[W] = __builtin_subc(dg[W], (u32)cy, bw, &bw);
  }
  {
    u32 tmp[W + 1];
    unsigned bw = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) tmp[k] = __builtin_subc(dg[k], qdig[k], bw, &bw);
    tmp[W] = __builtin_subc(dg[W], 0u, bw, &bw);
#pragma unroll
    for (int k = 0; k < W; ++k) dg[k] = bw ? dg[k] : tmp[k];
  }
  if (L <= 4) {  // a short run per thread: plain (vectorisable) stores
    if ((size_t)t < left) {
      u64 *o = out + gid * (size_t)L;
#pragma unroll
      for (int k = 0; k < L; ++k) o[k] = (u64)dg[2 * k] | ((u64)dg[2 * k + 1] << 32);
    }
    return;
  }
  // the L words of one coefficient are contiguous in HBM: transpose through LDS, eight words per coefficient at a
  // time, so the stores are coalesced 64-byte runs instead of 64 scattered words per instruction
  constexpr int CH = 8;
  u64 *ob = out + coef0 * (size_t)L;
#pragma unroll
  for (int c = 0; c < (L + CH - 1) / CH; ++c) {
    if (c) __syncthreads();
#pragma unroll
    for (int lb = 0; lb < CH; ++lb)
      if (c * CH + lb < L) sl[t][lb] = (u64)dg[2 * (c * CH + lb)] | ((u64)dg[2 * (c * CH + lb) + 1] << 32);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int w = t + 256 * j, coef = w >> 3, k = c * CH + (w & 7);
      if (k < L && (size_t)coef < left) ob[(size_t)coef * L + k] = sl[coef][w & 7];
    }
  }
}

// ---- project, carry-free MAC form ---------------------------------------------------------------------
// x mod p = sum_k sum_h half_{k,h} * (2^(64k+32h) mod p); each constant is cut into three 21-bit parts so
// S_j (64-bit) gathers sum half * part_j < 2 Lin * 2^53.  One 128-bit recombination and one reduction per residue.
// One launch covers the residues [cm0, cm0 + NMB) (NMB <= 16 keeps 3*NMB 64-bit sums + temporaries near 100 VGPRs);
// the table's rows are zero padded to a multiple of 4 (stride nms), so the inner loop has no guards.  The input words
// of a block's 256 coefficients are contiguous in HBM but strided per thread (Lin words apart), so they are staged
// through LDS eight words per coefficient at a time: coalesced 64-byte runs in, conflict-free column reads out,
// double buffered behind the products of the previous chunk.
static constexpr int kCrtChunk = 8;

template <int NMB>
__global__ __launch_bounds__(256) void k_crt_project64_mac(u64 *d, const u64 *limbs, const MC64 *__restrict__ mc,
                                                           const u32 *__restrict__ bparts, int logn, int nm, int nms,
                                                           int cm0, int Lin, size_t ncoef) {
  __shared__ u64 sl[2][256][kCrtChunk + 1];
  const int t = threadIdx.x;
  const size_t coef0 = (size_t)blockIdx.x * 256, gid = coef0 + t;
  const size_t left = ncoef - coef0;  // coefficients of this block (>= 1)
  const u64 *xb = limbs + coef0 * (size_t)Lin;
  u64 S0[NMB], S1[NMB], S2[NMB];
#pragma unroll
  for (int cm = 0; cm < NMB; ++cm) S0[cm] = S1[cm] = S2[cm] = 0;
  u64 stage[kCrtChunk];
  auto fetch = [&](int c) {
#pragma unroll
    for (int j = 0; j < kCrtChunk; ++j) {
      const int w = t + 256 * j, coef = w >> 3, k = c * kCrtChunk + (w & 7);
      stage[j] = (k < Lin && (size_t)coef < left) ? xb[(size_t)coef * Lin + k] : 0;
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kCrtChunk; ++j) {
      const int w = t + 256 * j;
      sl[buf][w >> 3][w & 7] = stage[j];
    }
  };
  const int nchunk = (Lin + kCrtChunk - 1) / kCrtChunk;
  fetch(0);
  park(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) fetch(c + 1);
    const int kend = Lin - c * kCrtChunk < kCrtChunk ? Lin - c * kCrtChunk : kCrtChunk;
    for (int lb = 0; lb < kend; ++lb) {
      const u64 l = sl[c & 1][t][lb];
      const u32 lo = (u32)l, hi = (u32)(l >> 32);
      const u32 *e = bparts + (size_t)(c * kCrtChunk + lb) * 6 * nms + cm0, *f = e + 3 * nms;  // [k][half][part][nms]
#pragma unroll
      for (int cm = 0; cm < NMB; ++cm) {
        S0[cm] = S0[cm] + (u64)e[cm] * lo + (u64)f[cm] * hi;
        S1[cm] = S1[cm] + (u64)e[nms + cm] * lo + (u64)f[nms + cm] * hi;
        S2[cm] = S2[cm] + (u64)e[2 * nms + cm] * lo + (u64)f[2 * nms + cm] * hi;
      }
    }
    if (c + 1 < nchunk) park((c + 1) & 1);
    __syncthreads();
  }
  if (gid >= ncoef) return;
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
#pragma unroll
  for (int cm = 0; cm < NMB; ++cm) {
    if (cm0 + cm < nm) {
      const MC64 c = mc[cm0 + cm];
      const Mod m = make_mod(c);
      // T = S0 + S1*2^21 + S2*2^42 < 2^104 as (Thi, Tlo); x = Thi*2^64 + Tlo = Thi*beta + Tlo (mod p)
      const unsigned __int128 T =
          (unsigned __int128)S0[cm] + ((unsigned __int128)S1[cm] << 21) + ((unsigned __int128)S2[cm] << 42);
      const u64 r = shoup_acc<true>((u64)(T >> 64), Tw64{c.beta, c.beta_sh}, fold2((u64)T, m), m);
      d[((b * nm + cm0 + cm) << logn) + i] = csub<u64>(fold2(r, m), m.p);
    }
  }
}

template <int NMB>
__global__ __launch_bounds__(256) void k_crt_project64(u64 *d, const u64 *limbs, const MC64 *__restrict__ mc, int logn,
                                                       int nm, int Lin, size_t ncoef) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncoef) return;
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
  const u64 *x = limbs + gid * (size_t)Lin;
  u64 r[NMB];
#pragma unroll
  for (int cm = 0; cm < NMB; ++cm) r[cm] = 0;
  for (int k = Lin - 1; k >= 0; --k) {
    const u64 l = x[k];
#pragma unroll
    for (int cm = 0; cm < NMB; ++cm) {
      if (cm < nm) {
        const MC64 c = mc[cm];
        const Mod m = make_mod(c);
        // r*beta mod p (one-off quotient: < 3p) + fold2(limb) (< p + 4*delta) stays below 2^64
        r[cm] = shoup_acc<true>(r[cm], Tw64{c.beta, c.beta_sh}, fold2(l, m), m);
      }
    }
  }
#pragma unroll
  for (int cm = 0; cm < NMB; ++cm) {
    if (cm < nm) {
      const Mod m = make_mod(mc[cm]);
      d[((b * nm + cm) << logn) + i] = csub<u64>(fold2(r[cm], m), m.p);
    }
  }
}

hipError_t launch_crt_lift_fast_u64(const Shape &s, const DevTables &t, uint64_t *limbs, const uint64_t *d, size_t batch,
                                    hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta || s.nm > 32 || (int)s.crt_Lacc > 33) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  const size_t ncoef = batch * s.n;
  if (t.crt_bfrag && ncoef % 64 == 0) {  // many moduli: the sum is a GEMM (kernels_crt_mfma.hip)
    const hipError_t e = launch_crt_lift_mfma_u64(s, t, limbs, d, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  int rounds = 0;
  while ((1u << rounds) < s.nm) ++rounds;
  const dim3 g((unsigned)((ncoef + 255) / 256)), b(256);
  const MC64 *mc = (const MC64 *)t.mc;
#define NFLHIP_LIFT(B)                                                                                                    \
  hipLaunchKernelGGL((k_crt_lift64<B>), g, b, 0, st, limbs, d, mc, t.qhat, t.qsh, s.logn, (int)s.nm, (int)s.crt_L, rounds, \
                     ncoef)
  if (t.qparts && s.crt_L <= 32) {
#define NFLHIP_LIFT_MAC(B)                                                                                                \
  hipLaunchKernelGGL((k_crt_lift64_mac<B>), g, b, 0, st, limbs, d, mc, t.qparts, (const u32 *)t.qsh, t.inv_qtop, s.logn, \
                     (int)s.nm, ncoef)
    switch ((int)s.crt_L) {  // exact digit count: no wasted products
#define NFLHIP_CASE(B) case B: NFLHIP_LIFT_MAC(B); break;
      NFLHIP_CASE(1) NFLHIP_CASE(2) NFLHIP_CASE(3) NFLHIP_CASE(4) NFLHIP_CASE(5) NFLHIP_CASE(6) NFLHIP_CASE(7) NFLHIP_CASE(8)
      NFLHIP_CASE(9) NFLHIP_CASE(10) NFLHIP_CASE(11) NFLHIP_CASE(12) NFLHIP_CASE(13) NFLHIP_CASE(14) NFLHIP_CASE(15)
      NFLHIP_CASE(16) NFLHIP_CASE(17) NFLHIP_CASE(18) NFLHIP_CASE(19) NFLHIP_CASE(20) NFLHIP_CASE(21) NFLHIP_CASE(22)
      NFLHIP_CASE(23) NFLHIP_CASE(24) NFLHIP_CASE(25) NFLHIP_CASE(26) NFLHIP_CASE(27) NFLHIP_CASE(28) NFLHIP_CASE(29)
      NFLHIP_CASE(30) NFLHIP_CASE(31) NFLHIP_CASE(32)
#undef NFLHIP_CASE
    }
#undef NFLHIP_LIFT_MAC
    return hipGetLastError();
  }
  const int la = (int)s.crt_Lacc;
  if (la <= 3) NFLHIP_LIFT(3);
  else if (la <= 5) NFLHIP_LIFT(5);
  else if (la <= 9) NFLHIP_LIFT(9);
  else if (la <= 17) NFLHIP_LIFT(17);
  else NFLHIP_LIFT(33);
#undef NFLHIP_LIFT
  return hipGetLastError();
}

hipError_t launch_crt_project_fast_u64(const Shape &s, const DevTables &t, uint64_t *d, const uint64_t *limbs, size_t L_in,
                                       size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta || s.nm > 32) return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  const size_t ncoef = batch * s.n;
  if (t.crt_bproj && L_in > 4 && L_in <= 64 && ncoef % 64 == 0) {  // many moduli: the sum is a GEMM (kernels_crt_mfma.hip)
    const hipError_t e = launch_crt_project_mfma_u64(s, t, d, limbs, L_in, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  const dim3 g((unsigned)((ncoef + 255) / 256)), b(256);
  const MC64 *mc = (const MC64 *)t.mc;
  if (t.bparts && L_in <= (size_t)t.proj_K && L_in > 4) {  // (few words: Horner's short chain beats the recombination)
#define NFLHIP_PROJ_MAC(B)                                                                                        \
  case B:                                                                                                        \
    hipLaunchKernelGGL((k_crt_project64_mac<B>), g, b, 0, st, d, limbs, mc, t.bparts, s.logn, (int)s.nm, nms, cm0, \
                       (int)L_in, ncoef);                                                                        \
    break;
    const int nms = (int)((s.nm + 3) & ~(size_t)3);
    for (int cm0 = 0; cm0 < nms; cm0 += 16) {
      switch (nms - cm0 < 16 ? nms - cm0 : 16) { NFLHIP_PROJ_MAC(4) NFLHIP_PROJ_MAC(8) NFLHIP_PROJ_MAC(12) NFLHIP_PROJ_MAC(16) }
    }
#undef NFLHIP_PROJ_MAC
    return hipGetLastError();
  }
#define NFLHIP_PROJ(B) \
  hipLaunchKernelGGL((k_crt_project64<B>), g, b, 0, st, d, limbs, mc, s.logn, (int)s.nm, (int)L_in, ncoef)
  if (s.nm <= 4) NFLHIP_PROJ(4);
  else if (s.nm <= 8) NFLHIP_PROJ(8);
  else if (s.nm <= 16) NFLHIP_PROJ(16);
  else NFLHIP_PROJ(32);
#undef NFLHIP_PROJ
  return hipGetLastError();
}

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_crt() {}
hipError_t warm_crt(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_crt, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
