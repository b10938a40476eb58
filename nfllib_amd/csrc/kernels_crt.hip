// kernels_crt.hip -- register-resident CRT lift / project for 64-bit limbs.
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
  int rounds = 0;
  while ((1u << rounds) < s.nm) ++rounds;
  const dim3 g((unsigned)((ncoef + 255) / 256)), b(256);
  const MC64 *mc = (const MC64 *)t.mc;
#define NFLHIP_LIFT(B)                                                                                                    \
  hipLaunchKernelGGL((k_crt_lift64<B>), g, b, 0, st, limbs, d, mc, t.qhat, t.qsh, s.logn, (int)s.nm, (int)s.crt_L, rounds, \
                     ncoef)
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
  const dim3 g((unsigned)((ncoef + 255) / 256)), b(256);
  const MC64 *mc = (const MC64 *)t.mc;
#define NFLHIP_PROJ(B) \
  hipLaunchKernelGGL((k_crt_project64<B>), g, b, 0, st, d, limbs, mc, s.logn, (int)s.nm, (int)L_in, ncoef)
  if (s.nm <= 4) NFLHIP_PROJ(4);
  else if (s.nm <= 8) NFLHIP_PROJ(8);
  else if (s.nm <= 16) NFLHIP_PROJ(16);
  else NFLHIP_PROJ(32);
#undef NFLHIP_PROJ
  return hipGetLastError();
}

}  // namespace nflhip
