// kernels_u32.hip -- register-tiled kernels for 32-bit limbs (30-bit moduli), n = 1024: ONE WAVE PER RNS ROW.
//
// A row is 1024 words = 64 lanes x 16 words, so the whole transform lives in one wavefront's registers and needs no
// workgroup barrier at all: ten stages = radix-16 (lane holds x[lane + 64k]) -> wave-local LDS exchange -> radix-16 on
// the 16 independent 64-word blocks -> wave-local exchange -> radix-4 on runs of 16 consecutive words.  The fused
// product keeps both operands resident (32 VGPRs) and touches HBM once per operand word; the reference's sequence
// a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b; c.invntt_pow_invphi() (poly.hpp:167-168,350) is 7 passes over memory.
// Arithmetic: Harvey lazy butterflies (values < 4p forward, < 2p inverse; 4p < 2^32 because p < 2^30,
// params.hpp:54-62) with Shoup constants from the same merged psi_br table as every other kernel (kernels.h).
// 32-bit multiplies are native here, which is why this limb size has the highest coefficient throughput.
#include <cstdlib>

#include "kernels.h"
#include "modarith.h"

namespace nflhip {

typedef uint32_t u32;
typedef Tw<u32> Tw32;
typedef ModConst<u32> MC32;

static constexpr int kLogN32 = 10;
static constexpr int kSlab32 = 1088;  // words of LDS per wave (1024 + padding of either exchange layout)

__device__ __forceinline__ u32 lazy2(u32 x, u32 p2) { return min(x, x - p2); }  // [0,4p) -> [0,2p)

__device__ __forceinline__ void ct32(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2) {
  const u32 X = lazy2(x, p2);
  const u32 T = mul_shoup_lazy<u32>(y, w.w, w.wp, p);  // any word -> [0,2p)
  x = X + T;
  y = X - T + p2;
}
__device__ __forceinline__ void gs32(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2) {  // inputs < 2p
  const u32 s = x + y, d = y - x + p2;
  x = lazy2(s, p2);
  y = mul_shoup_lazy<u32>(d, w.w, w.wp, p);
}

__device__ __forceinline__ int pad1(int e) { return e + ((e >> 6) << 2); }  // +4 words per 64: exchange 1
__device__ __forceinline__ int pad2(int e) { return e + (e >> 4); }         // +1 word per 16: exchange 2

// keeps the compiler from hoisting every twiddle load of a transform to its top (188 VGPRs, 2 waves per SIMD
// without it): loads stay inside the stage that uses them
__device__ __forceinline__ void stage_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// forward: r[k] = x[lane + 64k] on entry (any words), r[k] = NTT word 16*lane + k on exit (< 4p)
__device__ __forceinline__ void fwd1024(u32 (&r)[16], u32 *lds, const Tw32 *__restrict__ tw, int lane, u32 p, u32 p2) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw32 w = tw[(1 << s) + g];
#pragma unroll
      for (int h = 0; h < half; ++h) ct32(r[g * 2 * half + h], r[g * 2 * half + h + half], w, p, p2);
    }
  }
  const int B = lane >> 2, l2 = lane & 3;
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[pad1(lane + 64 * k)] = r[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = lds[pad1(64 * B + 4 * k + l2)];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw32 w = tw[(16 << s) + (B << s) + g];
#pragma unroll
      for (int h = 0; h < half; ++h) ct32(r[g * 2 * half + h], r[g * 2 * half + h + half], w, p, p2);
    }
  }
  wave_sync();  // (all reads of exchange 1 are done before its words are overwritten)
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[pad2(64 * B + 4 * k + l2)] = r[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = lds[pad2(16 * lane + k)];
  stage_fence();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const Tw32 w = tw[256 + 4 * lane + g];
    ct32(r[4 * g], r[4 * g + 2], w, p, p2);
    ct32(r[4 * g + 1], r[4 * g + 3], w, p, p2);
  }
  stage_fence();
#pragma unroll
  for (int g = 0; g < 8; ++g) ct32(r[2 * g], r[2 * g + 1], tw[512 + 8 * lane + g], p, p2);
}

// inverse: r[k] = NTT word 16*lane + k (< 2p) on entry, r[k] = x[lane + 64k] canonical on exit
__device__ __forceinline__ void inv1024(u32 (&r)[16], u32 *lds, const Tw32 *__restrict__ tw, const MC32 &c, int lane) {
  const u32 p = c.p, p2 = c.p2;
#pragma unroll
  for (int g = 0; g < 8; ++g) gs32(r[2 * g], r[2 * g + 1], tw[512 + (511 - (8 * lane + g))], p, p2);
  stage_fence();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const Tw32 w = tw[256 + (255 - (4 * lane + g))];
    gs32(r[4 * g], r[4 * g + 2], w, p, p2);
    gs32(r[4 * g + 1], r[4 * g + 3], w, p, p2);
  }
  const int B = lane >> 2, l2 = lane & 3;
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[pad2(16 * lane + k)] = r[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = lds[pad2(64 * B + 4 * k + l2)];
#pragma unroll
  for (int s = 3; s >= 0; --s) {
    const int half = 8 >> s, m = 16 << s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw32 w = tw[m + (m - 1 - ((B << s) + g))];
#pragma unroll
      for (int h = 0; h < half; ++h) gs32(r[g * 2 * half + h], r[g * 2 * half + h + half], w, p, p2);
    }
  }
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) lds[pad1(64 * B + 4 * k + l2)] = r[k];
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = lds[pad1(lane + 64 * k)];
#pragma unroll
  for (int s = 3; s >= 1; --s) {
    const int half = 8 >> s, m = 1 << s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const Tw32 w = tw[m + (m - 1 - g)];
#pragma unroll
      for (int h = 0; h < half; ++h) gs32(r[g * 2 * half + h], r[g * 2 * half + h + half], w, p, p2);
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h) {  // last stage with n^-1 folded in; canonical outputs
    const u32 u = r[h], x = r[h + 8];
    r[h] = mul_shoup<u32>(u + x, c.ninv, c.ninv_sh, p);
    r[h + 8] = mul_shoup<u32>(x - u + p2, c.w1ninv, c.w1ninv_sh, p);
  }
}

// MODE 0: c = INTT(NTT(a) (.) NTT(b));  1: the same with b already in NTT form;  2: dst = NTT(a);  3: dst = INTT(a)
template <int MODE>
__device__ __forceinline__ void row1024(u32 *c, const u32 *a, const u32 *b, size_t row, u32 *lds, const Tw32 *tw,
                                        const MC32 &k, int lane) {
  const u32 *ar = a + (row << kLogN32);
  u32 ra[16];
  if (MODE == 3) {  // NTT-form input: lane holds words 16*lane .. 16*lane+15
    const uint4 *v = reinterpret_cast<const uint4 *>(ar + 16 * lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 t = v[q];
      ra[4 * q] = t.x; ra[4 * q + 1] = t.y; ra[4 * q + 2] = t.z; ra[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) ra[j] = ar[lane + 64 * j];
    fwd1024(ra, lds, tw, lane, k.p, k.p2);
  }
  if (MODE == 2) {
    uint4 *o = reinterpret_cast<uint4 *>(c + (row << kLogN32) + 16 * lane);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = make_uint4(reduce4<u32>(ra[4 * q], k.p), reduce4<u32>(ra[4 * q + 1], k.p), reduce4<u32>(ra[4 * q + 2], k.p),
                        reduce4<u32>(ra[4 * q + 3], k.p));
    return;
  }
  if (MODE == 0 || MODE == 1) {
    const u32 *br = b + (row << kLogN32);
    u32 rb[16];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) rb[j] = br[lane + 64 * j];
      wave_sync();  // the slab is reused
      fwd1024(rb, lds, tw, lane, k.p, k.p2);
#pragma unroll
      for (int j = 0; j < 16; ++j) rb[j] = reduce4<u32>(rb[j], k.p);
    } else {
      const uint4 *v = reinterpret_cast<const uint4 *>(br + 16 * lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 t = v[q];
        rb[4 * q] = t.x; rb[4 * q + 1] = t.y; rb[4 * q + 2] = t.z; rb[4 * q + 3] = t.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) ra[j] = barrett<u32>::mul(reduce4<u32>(ra[j], k.p), rb[j], k.p, k.mu);
  }
  inv1024(ra, lds, tw, k, lane);
  u32 *cr = c + (row << kLogN32);
#pragma unroll
  for (int j = 0; j < 16; ++j) cr[lane + 64 * j] = ra[j];
}

template <int MODE>
__global__ __launch_bounds__(256, 5) void k_row1024_u32(u32 *c, const u32 *a, const u32 *b, const Tw32 *__restrict__ psi,
                                                     const MC32 *__restrict__ mc, int nm, size_t rows) {
  __shared__ u32 slab[4][kSlab32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;  // whole waves only: no workgroup barrier is used
  const int cm = (int)(row % (size_t)nm);
  row1024<MODE>(c, a, b, row, slab[wave], psi + ((size_t)cm << kLogN32), mc[cm], lane);
}

// Persistent variant for few moduli (NMT <= 4 tables of 8 KiB): the twiddle tables are copied into LDS once per
// workgroup and every wave walks over many rows, so the per-lane twiddle reads of the middle stages are LDS reads
// (~100 cycles) instead of L2 reads (~700): the transform is latency-bound at 4 waves per SIMD otherwise.
template <int MODE, int NMT>
__global__ __launch_bounds__(256) void k_row1024_u32_lds(u32 *c, const u32 *a, const u32 *b, const Tw32 *__restrict__ psi,
                                                         const MC32 *__restrict__ mc, int nm, size_t rows) {
  __shared__ u32 slab[4][kSlab32];
  __shared__ Tw32 table[NMT][1 << kLogN32];
  for (int i = threadIdx.x; i < nm << kLogN32; i += 256) table[i >> kLogN32][i & ((1 << kLogN32) - 1)] = psi[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * 4;
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += stride) {
    const int cm = (int)(row % (size_t)nm);
    row1024<MODE>(c, a, b, row, slab[wave], table[cm], mc[cm], lane);
    wave_sync();  // the slab is reused by the next row
  }
}

static inline bool shape32(const Shape &s) { return s.limb_bits == 32 && s.logn == kLogN32; }

// mode as in k_row1024_u32; hipErrorNotSupported for every other shape
hipError_t launch_row1024_u32(const Shape &s, const DevTables &t, int mode, uint32_t *c, const uint32_t *a,
                              const uint32_t *b, size_t batch, hipStream_t st) {
  if (!shape32(s)) return hipErrorNotSupported;
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  const size_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 bl(256);
  const Tw32 *psi = (const Tw32 *)t.psi;
  const MC32 *mc = (const MC32 *)t.mc;
  static const int use_lds = getenv("NFLHIP_U32_LDS") ? atoi(getenv("NFLHIP_U32_LDS")) : 1;
  // measured (u32/1024/1, batch 2^19): forward 430 -> 482 M/s, inverse 498 -> 514 M/s with the LDS tables; the fused
  // products do not gain (they are bound by VALU issue, not by twiddle latency), so they keep the plain kernel
  if (use_lds && mode >= 2 && s.nm <= 4 && blocks >= 4096) {
    const dim3 g(1024);  // 4 resident workgroups per CU (120 VGPRs): every wave walks rows at stride 4096
#define NFLHIP_U32_LDS(M, N) hipLaunchKernelGGL((k_row1024_u32_lds<M, N>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows)
#define NFLHIP_U32_LDS_M(M)                      \
  if (s.nm == 1) NFLHIP_U32_LDS(M, 1);           \
  else if (s.nm == 2) NFLHIP_U32_LDS(M, 2);      \
  else NFLHIP_U32_LDS(M, 4)
    switch (mode) {
      case 0: NFLHIP_U32_LDS_M(0); break;
      case 1: NFLHIP_U32_LDS_M(1); break;
      case 2: NFLHIP_U32_LDS_M(2); break;
      default: NFLHIP_U32_LDS_M(3); break;
    }
#undef NFLHIP_U32_LDS_M
#undef NFLHIP_U32_LDS
    return hipGetLastError();
  }
  const dim3 g((unsigned)blocks);
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_row1024_u32<0>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    case 1: hipLaunchKernelGGL((k_row1024_u32<1>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    case 2: hipLaunchKernelGGL((k_row1024_u32<2>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    default: hipLaunchKernelGGL((k_row1024_u32<3>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
  }
  return hipGetLastError();
}

}  // namespace nflhip
