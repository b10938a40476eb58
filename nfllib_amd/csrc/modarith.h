// modarith.h -- word-level modular arithmetic for gfx950 kernels.
//
// All moduli are the reference's NTT primes: 2 bits below the word
// (params.hpp:27-28, 61-62, 104-105), so 4p fits the word and the lazy ranges
// [0,2p) / [0,4p) of Harvey's butterflies (algos.hpp:27-41) are representable.
// Nothing here is a translation of the reference's functors: outputs are the
// canonical representatives, so any exact arithmetic is bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nflhip {

template <typename T> struct limb;

template <> struct limb<uint64_t> {
  static constexpr int bits = 64;
  __device__ __forceinline__ static uint64_t mulhi(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
};
template <> struct limb<uint32_t> {
  static constexpr int bits = 32;
  __device__ __forceinline__ static uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
};
template <> struct limb<uint16_t> {
  static constexpr int bits = 16;
  __device__ __forceinline__ static uint16_t mulhi(uint16_t a, uint16_t b) {
    return (uint16_t)(((uint32_t)a * (uint32_t)b) >> 16);
  }
};

template <typename T> __device__ __forceinline__ T mullo(T a, T b) { return (T)(a * b); }
template <> __device__ __forceinline__ uint16_t mullo<uint16_t>(uint16_t a, uint16_t b) {
  return (uint16_t)((uint32_t)a * (uint32_t)b);
}

// x - m if x >= m (one conditional subtract)
template <typename T> __device__ __forceinline__ T csub(T x, T m) { return (T)(x >= m ? (T)(x - m) : x); }

// Shoup multiplication by a constant w with companion wp = floor(w*2^W/p):
// any word x -> x*w mod p in [0,2p)   (lazy; same identity as ops.hpp:231-241)
template <typename T> __device__ __forceinline__ T mul_shoup_lazy(T x, T w, T wp, T p) {
  const T q = limb<T>::mulhi(x, wp);
  return (T)(mullo<T>(x, w) - mullo<T>(q, p));
}
// 32-bit limbs: gfx950 has no full-word v_mad_u32, but the LOW dword of v_mad_u64_u32 (32 x 32 + 64) is a*b + lo(c)
// whatever the addend's high dword holds -- x w - q p is two such multiply-adds instead of two multiplies and a
// subtract.  The register-tiled 32-bit kernels (kernels_wave.hip) use it by carrying every word in the low half of a
// 64-bit pair whose high half is never looked at; these two helpers keep the compiler from narrowing such expressions
// back to 32-bit multiplies: an arbitrary register as high half (no instruction), and a consumer of a high half (no
// instruction either).
__device__ __forceinline__ uint64_t junk_above(uint32_t lo) {
  uint32_t junk = 0;
  junk = __builtin_nondeterministic_value(junk);  // an arbitrary (frozen) value: no instruction, any register
  return ((uint64_t)junk << 32) | lo;
}
__device__ __forceinline__ uint32_t low_of_pair(uint64_t acc) {
  asm volatile("" ::"v"((uint32_t)(acc >> 32)));
  return (uint32_t)acc;
}
template <typename T> __device__ __forceinline__ T mul_shoup(T x, T w, T wp, T p) {
  return csub<T>(mul_shoup_lazy<T>(x, w, wp, p), p);
}

// [0,4p) -> [0,p)
template <typename T> __device__ __forceinline__ T reduce4(T x, T p) {
  x = csub<T>(x, (T)(2 * p));
  return csub<T>(x, p);
}

// Exact x*y mod p for x,y < p: Barrett with mu = floor(2^(2W-4)/p).
// T = x*y < 2^(2W-4); q = mulhi(T >> (W-4), mu) in [floor(T/p)-2, floor(T/p)].
template <typename T> struct barrett;
template <> struct barrett<uint64_t> {
  __device__ __forceinline__ static uint64_t mul(uint64_t x, uint64_t y, uint64_t p, uint64_t mu) {
    const uint64_t lo = x * y, hi = __umul64hi(x, y);
    const uint64_t th = (hi << 4) | (lo >> 60);
    const uint64_t q = __umul64hi(th, mu);
    uint64_t r = lo - q * p;
    r = csub<uint64_t>(r, 2 * p);
    return csub<uint64_t>(r, p);
  }
};
template <> struct barrett<uint32_t> {
  __device__ __forceinline__ static uint32_t mul(uint32_t x, uint32_t y, uint32_t p, uint32_t mu) {
    const uint64_t t = (uint64_t)x * y;
    const uint32_t th = (uint32_t)(t >> 28);
    const uint32_t q = __umulhi(th, mu);
    uint32_t r = (uint32_t)t - q * p;
    r = csub<uint32_t>(r, 2 * p);
    return csub<uint32_t>(r, p);
  }
};
template <> struct barrett<uint16_t> {
  __device__ __forceinline__ static uint16_t mul(uint16_t x, uint16_t y, uint16_t p, uint16_t mu) {
    const uint32_t t = (uint32_t)x * y;
    const uint32_t th = t >> 12;
    const uint32_t q = (th * (uint32_t)mu) >> 16;
    uint32_t r = (t - q * p) & 0xffffu;
    r = r >= 2u * p ? r - 2u * p : r;
    r = r >= p ? r - p : r;
    return (uint16_t)r;
  }
};

// floor(x * 2^W / p) for x < p  (compute_shoup, ops.hpp:165-177, after its x mod p loop)
template <typename T> struct shoup_of;
template <> struct shoup_of<uint64_t> {
  // mu = floor(2^124/p).  q_est = (x*mu) >> 60 is in [q-5, q]; fix up with the
  // 128-bit remainder r = x*2^64 - q_est*p.
  __device__ __forceinline__ static uint64_t get(uint64_t x, uint64_t p, uint64_t mu) {
    const uint64_t lo = x * mu, hi = __umul64hi(x, mu);
    uint64_t q = (hi << 4) | (lo >> 60);
    const uint64_t qp_lo = q * p, qp_hi = __umul64hi(q, p);
    uint64_t r_lo = 0 - qp_lo;
    uint64_t r_hi = x - qp_hi - (qp_lo != 0 ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool ge = (r_hi != 0) || (r_lo >= p);
      const uint64_t nlo = r_lo - p;
      r_hi -= (ge && r_lo < p) ? 1 : 0;
      r_lo = ge ? nlo : r_lo;
      q += ge ? 1 : 0;
    }
    return q;
  }
};
template <> struct shoup_of<uint32_t> {
  __device__ __forceinline__ static uint32_t get(uint32_t x, uint32_t p, uint32_t) {
    return (uint32_t)((((uint64_t)x) << 32) / p);
  }
};
template <> struct shoup_of<uint16_t> {
  __device__ __forceinline__ static uint16_t get(uint16_t x, uint16_t p, uint16_t) {
    return (uint16_t)((((uint32_t)x) << 16) / p);
  }
};

// counter-based splitmix64 (harness inputs)
__host__ __device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, int operand, uint64_t g) {
  uint64_t z = (seed ^ ((uint64_t)operand << 62)) + (g + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace nflhip
