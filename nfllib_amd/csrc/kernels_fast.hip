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
#include "kernels.h"
#include "modarith.h"

namespace nflhip {

typedef uint64_t u64;
typedef Tw<uint64_t> Tw64;
typedef ModConst<uint64_t> MC64;

static constexpr int kLogN = 12;
static constexpr int kN = 1 << kLogN;
static constexpr int kThreads = 256;
static constexpr int kLdsWords = kN + (kN >> 4);  // padded slab

__device__ __forceinline__ int pad(int e) { return e + (e >> 4); }

// ---- one lazy butterfly each way -------------------------------------------------
// Cooley-Tukey: x,y in [0,4p) -> x' = x + w*y, y' = x - w*y, both in [0,4p)
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, const Tw64 w, const u64 p, const u64 p2) {
  const u64 u = csub<u64>(x, p2);
  const u64 m = mul_shoup_lazy<u64>(y, w.w, w.wp, p);
  x = u + m;
  y = u - m + p2;
}
// Gentleman-Sande with the negated mirrored twiddle: u,v in [0,2p) ->
// u' = u + v, v' = (v - u) * w, both in [0,2p)
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, const Tw64 w, const u64 p, const u64 p2) {
  const u64 s = csub<u64>(x + y, p2);
  const u64 d = y - x + p2;
  x = s;
  y = mul_shoup_lazy<u64>(d, w.w, w.wp, p);
}

// radix-16 register passes; TW(s, g) yields the twiddle of sub-stage s (0..3), group g
template <class TW> __device__ __forceinline__ void ct16(u64 (&v)[16], TW tw, const u64 p, const u64 p2) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw(s, g);
#pragma unroll
      for (int h = 0; h < half; ++h) ct_bfly(v[g * 2 * half + h], v[g * 2 * half + h + half], w, p, p2);
    }
  }
}
template <class TW> __device__ __forceinline__ void gs16(u64 (&v)[16], TW tw, const u64 p, const u64 p2) {
#pragma unroll
  for (int s = 3; s >= 0; --s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw(s, g);
#pragma unroll
      for (int h = 0; h < half; ++h) gs_bfly(v[g * 2 * half + h], v[g * 2 * half + h + half], w, p, p2);
    }
  }
}

// ---- forward transform of the 16 words a thread loaded as x[t + 256k] ---------------
// On return thread q = t holds X[16q + k] (bit-reversed order positions), lazy in [0,4p).
__device__ __forceinline__ void fwd_core(u64 (&v)[16], u64 *sm, const Tw64 *__restrict__ tw, const u64 p, const u64 p2,
                                         const int t, const bool war_barrier) {
  // F1: stages 0-3, block index of sub-stage s is the group g: psi[2^s + g] (wave-uniform)
  ct16(v, [&](int s, int g) { return tw[(1 << s) + g]; }, p, p2);
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
  ct16(v, [&](int s, int g) { return tw[(16 << s) + (B << s) + g]; }, p, p2);
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
  // F3: stages 8-11 inside 16-word block q = t: psi[2^(8+s) + q*2^s + g]
  ct16(v, [&](int s, int g) { return tw[(256 << s) + (t << s) + g]; }, p, p2);
}

// ---- inverse transform of the 16 words a thread holds as X[16q + k], in [0,2p) -------
// On return thread t holds x[t + 256k], canonical in [0,p).
__device__ __forceinline__ void inv_core(u64 (&v)[16], u64 *sm, const Tw64 *__restrict__ tw, const MC64 &c, const int t) {
  const u64 p = c.p, p2 = c.p2;
  // I1: stages 11..8; mirrored index 2m-1-j with m = 2^(8+s), j = q*2^s + g
  gs16(v, [&](int s, int g) { return tw[(512 << s) - 1 - ((t << s) + g)]; }, p, p2);
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
  gs16(v, [&](int s, int g) { return tw[(32 << s) - 1 - ((B << s) + g)]; }, p, p2);
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
  // I3: stages 3..1 (uniform twiddles), then stage 0 with n^-1 folded in
#pragma unroll
  for (int s = 3; s >= 1; --s) {
    const int half = 8 >> s;
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw64 w = tw[(2 << s) - 1 - g];
#pragma unroll
      for (int h = 0; h < half; ++h) gs_bfly(v[g * 2 * half + h], v[g * 2 * half + h + half], w, p, p2);
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const u64 x = v[h], y = v[h + 8];
    v[h] = mul_shoup<u64>(x + y, c.ninv, c.ninv_sh, p);
    v[h + 8] = mul_shoup<u64>(y - x + p2, c.w1ninv, c.w1ninv_sh, p);
  }
}

// ---- the metric kernel: c = INTT( NTT(a) (.) NTT(b) ), one row per workgroup ---------
template <bool B_IS_NTT>
__global__ __launch_bounds__(kThreads) void k_polymul4096(u64 *c, const u64 *a, const u64 *b,
                                                          const Tw64 *__restrict__ psi, const MC64 *__restrict__ mc,
                                                          int nm) {
  __shared__ u64 sm[kLdsWords];
  const int t = threadIdx.x;
  const size_t row = blockIdx.x;
  const int cm = (int)(row % (size_t)nm);
  const MC64 mcc = mc[cm];
  const u64 p = mcc.p, p2 = mcc.p2;
  const Tw64 *tw = psi + ((size_t)cm << kLogN);
  const size_t off = row << kLogN;

  u64 va[16], vb[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) va[k] = a[off + t + 256 * k];
  if (!B_IS_NTT) {
#pragma unroll
    for (int k = 0; k < 16; ++k) vb[k] = b[off + t + 256 * k];
  } else {
    // b already in NTT form: thread q needs B[16q + k]
    const ulonglong2 *b2 = reinterpret_cast<const ulonglong2 *>(b + off + 16 * t);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const ulonglong2 x = b2[k];
      vb[2 * k] = x.x;
      vb[2 * k + 1] = x.y;
    }
  }
  fwd_core(va, sm, tw, p, p2, t, false);
  if (!B_IS_NTT) fwd_core(vb, sm, tw, p, p2, t, true);
  // point-wise product on canonical representatives (operator*, ops.hpp:201-219)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const u64 x = reduce4<u64>(va[k], p);
    const u64 y = B_IS_NTT ? vb[k] : reduce4<u64>(vb[k], p);
    va[k] = barrett<u64>::mul(x, y, p, mcc.mu);
  }
  inv_core(va, sm, tw, mcc, t);
#pragma unroll
  for (int k = 0; k < 16; ++k) c[off + t + 256 * k] = va[k];
}

// ---- stand-alone transforms (in place or out of place) --------------------------------
__global__ __launch_bounds__(kThreads) void k_ntt_fwd4096(const u64 *src, u64 *dst, const Tw64 *__restrict__ psi,
                                                          const MC64 *__restrict__ mc, int nm) {
  __shared__ u64 sm[kLdsWords];
  const int t = threadIdx.x;
  const size_t row = blockIdx.x;
  const int cm = (int)(row % (size_t)nm);
  const u64 p = mc[cm].p, p2 = mc[cm].p2;
  const Tw64 *tw = psi + ((size_t)cm << kLogN);
  const size_t off = row << kLogN;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = src[off + t + 256 * k];
  fwd_core(v, sm, tw, p, p2, t, false);
  // thread q holds X[16q+k]: transpose inside the wave's LDS region for coalesced stores
  {
    const int base = 17 * t;
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[base + k] = reduce4<u64>(v[k], p);
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

__global__ __launch_bounds__(kThreads) void k_ntt_inv4096(const u64 *src, u64 *dst, const Tw64 *__restrict__ psi,
                                                          const MC64 *__restrict__ mc, int nm) {
  __shared__ u64 sm[kLdsWords];
  const int t = threadIdx.x;
  const size_t row = blockIdx.x;
  const int cm = (int)(row % (size_t)nm);
  const MC64 mcc = mc[cm];
  const Tw64 *tw = psi + ((size_t)cm << kLogN);
  const size_t off = row << kLogN;
  const int w = t >> 6, l = t & 63;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int e = 1024 * w + 64 * j + l;
    sm[pad(e)] = src[off + e];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  u64 v[16];
  {
    const int base = 17 * t;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sm[base + k];
  }
  inv_core(v, sm, tw, mcc, t);
#pragma unroll
  for (int k = 0; k < 16; ++k) dst[off + t + 256 * k] = v[k];
}

// ---- launchers --------------------------------------------------------------------------
static inline bool fast_shape(const Shape &s) { return s.limb_bits == 64 && s.logn == kLogN; }

hipError_t launch_polymul_fast_u64(const Shape &s, const DevTables &t, uint64_t *c, const uint64_t *a, const uint64_t *b,
                                   int b_is_ntt, size_t batch, hipStream_t st) {
  if (!fast_shape(s)) return hipErrorNotSupported;
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorInvalidValue;
  if (b_is_ntt)
    hipLaunchKernelGGL((k_polymul4096<true>), dim3((unsigned)rows), dim3(kThreads), 0, st, c, a, b, (const Tw64 *)t.psi,
                       (const MC64 *)t.mc, (int)s.nm);
  else
    hipLaunchKernelGGL((k_polymul4096<false>), dim3((unsigned)rows), dim3(kThreads), 0, st, c, a, b, (const Tw64 *)t.psi,
                       (const MC64 *)t.mc, (int)s.nm);
  return hipGetLastError();
}

hipError_t launch_ntt_fwd_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st) {
  if (!fast_shape(s)) return hipErrorNotSupported;
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ntt_fwd4096, dim3((unsigned)rows), dim3(kThreads), 0, st, src, dst, (const Tw64 *)t.psi,
                     (const MC64 *)t.mc, (int)s.nm);
  return hipGetLastError();
}

hipError_t launch_ntt_inv_fast_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t batch,
                                   hipStream_t st) {
  if (!fast_shape(s)) return hipErrorNotSupported;
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  if (rows > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ntt_inv4096, dim3((unsigned)rows), dim3(kThreads), 0, st, src, dst, (const Tw64 *)t.psi,
                     (const MC64 *)t.mc, (int)s.nm);
  return hipGetLastError();
}

}  // namespace nflhip
