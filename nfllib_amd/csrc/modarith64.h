// modarith64.h -- 64-bit-limb modular arithmetic specialised for the reference's
// 62-bit primes p = 2^62 - delta (params.hpp:94-97), shared by the register-tiled NTT
// kernels (kernels_fast.hip) and the CRT kernels (kernels_crt.hip).
#pragma once
#include "kernels.h"
#include "modarith.h"

namespace nflhip {

typedef uint64_t u64;
typedef Tw<uint64_t> Tw64;
typedef ModConst<uint64_t> MC64;

// ---- modulus in the form the butterflies want it ------------------------------------
// The reference's 62-bit primes are p = 2^62 - delta with delta = c*2^21 - 1 (params.hpp:94-97);
// delta fits 32 bits for c < 2048 (every prime the reference lists up to index 63 and beyond), so
// q*p = (q << 62) - q*delta costs one 32x32 multiply-add.  Every range argument below needs only
// delta < 2^32: fold2 lands below 2^62 + 3*delta = p + 4*delta and 4p + 4*delta = 2^64 exactly.
struct Mod {
  u64 p, p2, p3;
  uint32_t d;  // delta
};
typedef uint32_t u32;

// S = a*b + c with the carry-out of the 64-bit accumulate materialised as 0/1 in `carry`.
// v_mad_u64_u32's SGPR carry has no C spelling; both instructions sit in ONE asm statement
// because gfx950 needs 2 wait states between a VALU SGPR write and a VALU read of that SGPR
// and hipcc pads nothing inside (or around the operands of) an asm string.
__device__ __forceinline__ u64 mad_carry(const u32 a, const u32 b, const u64 c, u32 &carry) {
  u64 S, cy;
  asm("v_mad_u64_u32 %0, %2, %3, %4, %5\n\ts_nop 1\n\tv_addc_co_u32 %1, %2, 0, 0, %2"
      : "=v"(S), "=v"(carry), "=&s"(cy)
      : "v"(a), "v"(b), "v"(c));
  return S;
}
// exact floor(y*wp / 2^64): y0*a1 + hi32(y0*a0) cannot overflow, the second cross product is
// accumulated with its carry, and (sum >> 32 | carry << 32) is the addend of the top product.
__device__ __forceinline__ u64 mulhi_x(const u64 y, const u64 wp) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wp, a1 = (u32)(wp >> 32);
  const u64 X = (u64)y0 * a1 + (u64)__umulhi(y0, a0);
  u32 c;
  const u64 S = mad_carry(y1, a0, X, c);
  return (u64)y1 * a1 + (((u64)c << 32) | (u32)(S >> 32));
}
// floor(y*wp / 2^64) - e with e in {0,1}: the hi32(y0*a0) term of the exact quotient is
// dropped, which removes the zero-extension register shuffles of the exact chain.
__device__ __forceinline__ u64 mulhi_a(const u64 y, const u64 wp) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), a0 = (u32)wp, a1 = (u32)(wp >> 32);
  const u64 A = (u64)y1 * a0;
  u32 c;
  const u64 S = mad_carry(y0, a1, A, c);
  return (u64)y1 * a1 + (((u64)c << 32) | (u32)(S >> 32));
}
// seed + (y*w mod p, lazily in [0,2p)) for ANY 64-bit y: Shoup quotient, then
// y*w - q*p = y*w + q*delta - (q << 62), all modulo 2^64, accumulated onto the seed.
template <bool APPROX = false>
__device__ __forceinline__ u64 shoup_acc(const u64 y, const Tw64 w, const u64 seed, const Mod &k) {
  const u64 q = APPROX ? mulhi_a(y, w.wp) : mulhi_x(y, w.wp);  // APPROX: result in [0,3p) instead of [0,2p)
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w.w, w1 = (u32)(w.w >> 32);
  const u32 q0 = (u32)q, q1 = (u32)(q >> 32);
  u64 acc = (u64)y0 * w0 + seed;
  acc = (u64)q0 * k.d + acc;
  const u32 hi = (u32)(acc >> 32) + y0 * w1 + y1 * w0 + q1 * k.d - (q0 << 30);
  return ((u64)hi << 32) | (u32)acc;
}

// z -> z mod-ish p in [0, 2^62 + 3*delta) for ANY 64-bit z: 2^62 == delta (mod p), so the
// top two bits fold down with one multiply-add (v_lshrrev, v_and, v_mad_u64_u32).
__device__ __forceinline__ u64 fold2(const u64 z, const Mod &k) {
  return (z & 0x3fffffffffffffffull) + (u64)(u32)(z >> 62) * k.d;
}


__device__ __forceinline__ Mod make_mod(const MC64 &c) {
  Mod k;
  k.p = c.p;
  k.p2 = c.p2;
  k.p3 = c.p2 + c.p;
  k.d = (u32)c.delta;
  return k;
}

// ---- one lazy butterfly each way -------------------------------------------------
// Cooley-Tukey, x' = x + w*y, y' = x - w*y.
//  ARITH 0: Harvey's ranges, x,y in [0,4p) -> [0,4p).
//  ARITH 2: x,y ANY 64-bit word -> any 64-bit word: U = fold2(x) < p + 4*delta, m < 2p exactly, so
//           U + m and U - m + 2p stay below 2^64; the x-path sum is folded into the multiply-add
//           chain and y' = (2U + 2p) - x'.
template <int ARITH>
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, const Tw64 w, const Mod &k) {
  if (ARITH == 0) {
    const u64 u = csub<u64>(x, k.p2);
    const u64 m = mul_shoup_lazy<u64>(y, w.w, w.wp, k.p);
    x = u + m;
    y = u - m + k.p2;
  } else if (ARITH == 2) {
    const u64 U = fold2(x, k);
    const u64 xn = shoup_acc(y, w, U, k);
    y = ((U << 1) + k.p2) - xn;
    x = xn;
  } else {
    //  ARITH 3: U < p + 4*delta and the product is reduced with the one-off quotient, m < 3p:
    //           x' = U + m < 4p + 4*delta = 2^64 and y' = U + 3p - m < 2^64 still fit the word.
    const u64 U = fold2(x, k);
    const u64 xn = shoup_acc<true>(y, w, U, k);
    y = ((U << 1) + k.p3) - xn;
    x = xn;
  }
}
// Gentleman-Sande with the negated mirrored twiddle: u,v in [0,2p) ->
// u' = u + v, v' = (v - u) * w, both in [0,2p)
//  ARITH 2: inputs < 2p; the sum is folded to < 2^62 + 3*delta (< 2p) with no compare.
template <int ARITH>
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, const Tw64 w, const Mod &k) {
  const u64 s = ARITH >= 2 ? fold2(x + y, k) : csub<u64>(x + y, k.p2);
  const u64 d = y - x + k.p2;
  x = s;
  y = ARITH == 0 ? mul_shoup_lazy<u64>(d, w.w, w.wp, k.p) : shoup_acc(d, w, 0, k);
}
// any 64-bit word (ARITH 1) or [0,4p) (ARITH 0) -> [0,p)
template <int ARITH> __device__ __forceinline__ u64 canon(u64 x, const Mod &k) {
  if (ARITH >= 2) return csub<u64>(fold2(x, k), k.p);  // < p + 4*delta, one subtract left
  x = csub<u64>(x, k.p2);
  return csub<u64>(x, k.p);
}
// x*y mod p for lazily reduced x, y (< 2^62 + 3*delta): T < 2^125, q = mulhi(T >> 61, mu2)
// is within 3 of floor(T/p), r = T - q*p < 4p, folded to < p + 4*delta.
__device__ __forceinline__ u64 mul_lazy(const u64 x, const u64 y, const u64 mu2, const Mod &k) {
  const u64 lo = x * y, hi = __umul64hi(x, y);
  const u64 th = (hi << 3) | (lo >> 61);
  const u64 q = __umul64hi(th, mu2);
  const u32 q0 = (u32)q, q1 = (u32)(q >> 32);
  u64 r = (u64)q0 * k.d + lo;                                   // lo - q*p = lo + q*delta - (q << 62)
  r += (u64)(q1 * k.d - (q0 << 30)) << 32;
  return fold2(r, k);
}

}  // namespace nflhip
