// kernels_wave.hip -- register-tiled kernels for n = 1024, 32-bit limbs (30-bit moduli) and 64-bit limbs (62-bit
// moduli): ONE WAVE PER RNS ROW.
//
// A row is 1024 words = 64 lanes x 16 words, so the whole transform lives in one wavefront's registers and needs no
// workgroup barrier at all: ten stages = radix-16 (lane holds x[lane + 64k]) -> wave-local LDS exchange -> radix-16 on
// the 16 independent 64-word blocks -> wave-local exchange -> radix-4 on runs of 16 consecutive words.  The fused
// product keeps both operands resident (32 VGPRs) and touches HBM once per operand word; the reference's sequence
// a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b; c.invntt_pow_invphi() (poly.hpp:167-168,350) is 7 passes over memory.
// Arithmetic: Harvey lazy butterflies (values < 4p forward, < 2p inverse; 4p < 2^32 because p < 2^30,
// params.hpp:54-62) with Shoup constants from the same merged psi_br table as every other kernel (kernels.h).
// 32-bit multiplies are native here, which is why this limb size has the highest coefficient throughput.
// The structure is a template over an arithmetic policy; the 64-bit policy reuses the fold2 / one-off-quotient
// butterflies of the 4096-word kernels (modarith64.h), so u64 rows of 1024 words (tests/ntt_perfs.cpp's shape) get a
// fused product too.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "modarith.h"
#include "modarith64.h"

namespace nflhip {

typedef Tw<u32> Tw32;
typedef ModConst<u32> MC32;

// workgroups of 256 threads per CU the 64-bit row kernels are compiled for (register budget 512 / (2 x this) VGPRs per thread).
// Measured (round 5, same box, u64/1024/2 and u64/2048/2 products per second): 2: 38.8 M / 16.8 M; 3: 27.0 M / 16.9 M; 4: 21.4 M /
// 13.4 M -- the fused product keeps two rows of 16 words in registers and spills beyond two workgroups per CU
#ifndef NFLHIP_W64_OCC
#define NFLHIP_W64_OCC 2
#endif
static constexpr int kSlabWords = 1088;  // LDS words per 1024 row words (padding of either exchange layout included)

__device__ __forceinline__ u32 lazy2(u32 x, u32 p2) { return min(x, x - p2); }  // [0,4p) -> [0,2p)

// 32-bit limbs: v_mad_u64_u32's low result dword is a*b + lo(c) whatever the high dwords hold, so
//   x' = X + (y w - q p) = lo( y*w + ( q*(-p) + (X, *) ) )        two multiply-adds, seeded with X
// and a Cooley-Tukey butterfly is 7 instructions (sub, min, mul_hi, mad, mad, lshl_add, sub) instead of 9, a
// Gentleman-Sande one 8 (the product is seeded with a pair whose low dword is 0).  Same words as before.
// The expressions are plain 64-bit C (the compiler schedules them like any instruction); what keeps it from narrowing
// them back to 32-bit multiplies is that every result's HIGH dword is "used": it is threaded, at no instruction,
// through a chain of empty asm statements (`tok`) that ends in one volatile consumer per pass.
// NFLHIP_U32_MAD=0 compiles the multiply / subtract butterflies instead.
#ifndef NFLHIP_U32_MAD
#define NFLHIP_U32_MAD 1
#endif
__device__ __forceinline__ u32 mad_low(u64 acc, u32 &tok) {
  asm("" : "=v"(tok) : "0"(tok), "v"((u32)(acc >> 32)));  // tok "depends" on the high dword; no instruction
  return (u32)acc;
}
__device__ __forceinline__ void tok_end(u32 tok) { asm volatile("" ::"v"(tok)); }

// Cooley-Tukey, Harvey ranges (x < 4p, y any word -> both < 4p): x' = X + (y w - q p), y' = 2X + 2p - x'
__device__ __forceinline__ void ct32(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2, u32 &tok) {
  const u32 X = lazy2(x, p2);
  const u32 q = __umulhi(y, w.wp);
  x = mad_low((u64)y * w.w + ((u64)q * (0u - p) + junk_above(X)), tok);
  y = (X << 1) + p2 - x;
}
__device__ __forceinline__ void gs32(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2, u32 &tok) {  // inputs < 2p
  const u32 s = x + y, d = y - x + p2;
  x = lazy2(s, p2);
  const u32 q = __umulhi(d, w.wp);
  y = mad_low((u64)d * w.w + ((u64)q * (0u - p) + junk_above(0u)), tok);
}
__device__ __forceinline__ void ct32_plain(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2) {
  const u32 X = lazy2(x, p2);
  const u32 T = mul_shoup_lazy<u32>(y, w.w, w.wp, p);  // any word -> [0,2p)
  x = X + T;
  y = X - T + p2;
}
__device__ __forceinline__ void gs32_plain(u32 &x, u32 &y, const Tw32 w, u32 p, u32 p2) {  // inputs < 2p
  const u32 s = x + y, d = y - x + p2;
  x = lazy2(s, p2);
  y = mul_shoup_lazy<u32>(d, w.w, w.wp, p);
}

// ---- arithmetic policies ----------------------------------------------------------------------------------------
// MAD: the multiply-add butterflies (ct32 / gs32).  Measured (MI355X, round 2): rows of 4096 words (one 256-thread
// workgroup per row) 9.9 -> 11.4 M products/s at u32/4096/4 (+15 %); wave-per-row kernels (n = 1024 / 2048) 201 -> 194 and
// 90 -> 85 M/s (-4 %: there the schedule, not the instruction count, is what binds) -- so only the 4096-word kernels use them.
template <bool MAD> struct Pol32T {
  typedef u32 T;
  typedef u32 WT;
  typedef Tw32 TW;
  typedef MC32 MC;
  struct K {
    u32 p, p2, mu;
    mutable u32 tok;  // the chain that keeps the high dwords of the multiply-adds "used" (ct32); ended by pass_end()
  };
  __device__ __forceinline__ static K make(const MC &c) { return K{c.p, c.p2, c.mu, 0u}; }
  __device__ __forceinline__ static WT wrap(T x) { return x; }
  __device__ __forceinline__ static T unwrap(WT x) { return x; }
  __device__ __forceinline__ static void pass_end(const K &k) { if (MAD) tok_end(k.tok); }
  __device__ __forceinline__ static void ct(WT &x, WT &y, const TW w, const K &k) {
    if (MAD) ct32(x, y, w, k.p, k.p2, k.tok); else ct32_plain(x, y, w, k.p, k.p2);
  }
  __device__ __forceinline__ static void gs(WT &x, WT &y, const TW w, const K &k) {
    if (MAD) gs32(x, y, w, k.p, k.p2, k.tok); else gs32_plain(x, y, w, k.p, k.p2);
  }
  __device__ __forceinline__ static T canon(T x, const K &k) { return reduce4<u32>(x, k.p); }  // forward output < 4p
  __device__ __forceinline__ static T prep(T b, const K &k) { return reduce4<u32>(b, k.p); }   // operand of mul()
  __device__ __forceinline__ static T mul(T a, T b, const K &k) {  // a lazy, b prepared or canonical -> [0,p)
    return barrett<u32>::mul(reduce4<u32>(a, k.p), b, k.p, k.mu);
  }
  __device__ __forceinline__ static void last(T &u, T &x, const MC &c, const K &k) {  // stage 0 with n^-1, canonical
    const T s = u + x, d = x - u + k.p2;
    u = mul_shoup<u32>(s, c.ninv, c.ninv_sh, k.p);
    x = mul_shoup<u32>(d, c.w1ninv, c.w1ninv_sh, k.p);
  }
  // b -+ m for canonical b and a product m of mul(): what the inverse butterflies take (< 2p)
  __device__ __forceinline__ static T addsub(T b, T m, bool sub, const K &k) { return sub ? b + k.p - m : b + m; }
};
typedef Pol32T<false> Pol32;
typedef Pol32T<NFLHIP_U32_MAD != 0> Pol32M;
struct Pol64 {
  typedef u64 T;
  typedef u64 WT;
  typedef Tw64 TW;
  typedef MC64 MC;
  __device__ __forceinline__ static WT wrap(T x) { return x; }
  __device__ __forceinline__ static T unwrap(WT x) { return x; }
  struct K {
    Mod m;
    u64 mu2;
  };
  __device__ __forceinline__ static K make(const MC &c) { return K{make_mod(c), c.mu2}; }
  __device__ __forceinline__ static void pass_end(const K &) {}
  __device__ __forceinline__ static void ct(T &x, T &y, const TW w, const K &k) { ct_bfly<3>(x, y, w, k.m); }
  __device__ __forceinline__ static void gs(T &x, T &y, const TW w, const K &k) { gs_bfly<3>(x, y, w, k.m); }
  __device__ __forceinline__ static T canon(T x, const K &k) { return nflhip::canon<3>(x, k.m); }
  __device__ __forceinline__ static T prep(T b, const K &k) { return fold2(b, k.m); }
  __device__ __forceinline__ static T mul(T a, T b, const K &k) { return mul_lazy(fold2(a, k.m), b, k.mu2, k.m); }  // < 2p
  __device__ __forceinline__ static T addsub(T b, T m, bool sub, const K &k) { return fold2(sub ? b + k.m.p2 - m : b + m, k.m); }
  __device__ __forceinline__ static void last(T &u, T &x, const MC &c, const K &k) {
    const T s = u + x, d = x - u + k.m.p2;
    u = mul_shoup<u64>(s, c.ninv, c.ninv_sh, k.m.p);
    x = mul_shoup<u64>(d, c.w1ninv, c.w1ninv_sh, k.m.p);
  }
};

// LB = lanes per 16-block of a row: 4 -> 64 lanes (one wave) own a 1024-word row, 8 -> 128 lanes (two waves) own a
// 2048-word row, 16 -> a whole 256-thread workgroup owns a 4096-word row (the three passes are then 4 + 4 + 4 stages).  A row is 16 blocks of BS = 16*LB words; lane t works on block B = t / LB in the middle pass.
template <int LB> __device__ __forceinline__ int pad1(int e) { return e + (e / (16 * LB)) * LB; }  // +LB words per block
__device__ __forceinline__ int pad2(int e) { return e + (e >> 4); }                                 // +1 word per 16

// keeps the compiler from hoisting every twiddle load of a transform to its top (188 VGPRs, 2 waves per SIMD
// without it): loads stay inside the stage that uses them
__device__ __forceinline__ void stage_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exchange 1 crosses the whole row: wave-local for one-wave rows, a workgroup barrier when two waves share a row
template <int LB> __device__ __forceinline__ void row_sync() {
  if (LB == 4) wave_sync(); else __syncthreads();
}

// forward: r[q] = x[t + W q] on entry (any words), r[q] = NTT word 16 t + q on exit (lazy); W = 16 LB lanes per row
template <class P, int LB>
__device__ __forceinline__ void fwd_row(typename P::T (&r)[16], typename P::T *lds, const typename P::TW *tw, int t,
                                        const typename P::K &k, const unsigned kf = 1u) {
  // kf = 2^r + blk: the row is block `blk` of a row 2^r times longer whose first r stages already ran (kf = 1: a whole row)
  constexpr int W = 16 * LB, BS = 16 * LB, NS3 = LB == 4 ? 2 : (LB == 8 ? 3 : 4);
  typename P::WT R[16];  // working form of the words (32-bit limbs: low dword of a pair, see ct32)
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(r[q]);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const typename P::TW w = tw[(kf << s) + g];
#pragma unroll
      for (int h = 0; h < half; ++h) P::ct(R[g * 2 * half + h], R[g * 2 * half + h + half], w, k);
    }
  }
  const int B = t / LB, l = t % LB;
#pragma unroll
  for (int q = 0; q < 16; ++q) lds[pad1<LB>(t + W * q)] = P::unwrap(R[q]);
  row_sync<LB>();
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(lds[pad1<LB>(BS * B + LB * q + l)]);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const typename P::TW w = tw[((16u * kf + B) << s) + g];
#pragma unroll
      for (int h = 0; h < half; ++h) P::ct(R[g * 2 * half + h], R[g * 2 * half + h + half], w, k);
    }
  }
  row_sync<LB>();  // (all reads of exchange 1 are done before its words are overwritten)
#pragma unroll
  for (int q = 0; q < 16; ++q) lds[pad2(BS * B + LB * q + l)] = P::unwrap(R[q]);
  wave_sync();     // exchange 2 stays inside a block = LB consecutive lanes
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(lds[pad2(16 * t + q)]);
#pragma unroll
  for (int i = 0; i < NS3; ++i) {  // last NS3 stages on the thread's 16 consecutive words
    const int d = 1 << (NS3 - 1 - i), G = 8 / d;
    stage_fence();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const typename P::TW w = tw[((256u * kf) << i) + G * t + g];
#pragma unroll
      for (int h = 0; h < d; ++h) P::ct(R[2 * d * g + h], R[2 * d * g + h + d], w, k);
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) r[q] = P::unwrap(R[q]);
  P::pass_end(k);
}

// inverse: r[q] = NTT word 16 t + q (< 2p) on entry, r[q] = x[t + W q] canonical on exit
template <class P, int LB>
__device__ __forceinline__ void inv_row(typename P::T (&r)[16], typename P::T *lds, const typename P::TW *tw,
                                        const typename P::MC &c, const typename P::K &k, int t, const unsigned ki = 2u,
                                        const bool plain_last = false) {
  // ki = 2^(r+1) - blk (mirrored indices of block blk); plain_last: stages r-1 .. 0 follow elsewhere, so the block's last
  // stage is an ordinary one (twiddle psi[ki - 1]) and the words stay lazy (< 2p)
  constexpr int W = 16 * LB, BS = 16 * LB, NS3 = LB == 4 ? 2 : (LB == 8 ? 3 : 4);
  typename P::WT R[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(r[q]);
#pragma unroll
  for (int i = NS3 - 1; i >= 0; --i) {
    const int d = 1 << (NS3 - 1 - i), G = 8 / d;
    stage_fence();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const typename P::TW w = tw[((256u * ki) << i) - 1u - (unsigned)(G * t + g)];
#pragma unroll
      for (int h = 0; h < d; ++h) P::gs(R[2 * d * g + h], R[2 * d * g + h + d], w, k);
    }
  }
  const int B = t / LB, l = t % LB;
  wave_sync();
#pragma unroll
  for (int q = 0; q < 16; ++q) lds[pad2(16 * t + q)] = P::unwrap(R[q]);
  wave_sync();
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(lds[pad2(BS * B + LB * q + l)]);
#pragma unroll
  for (int s = 3; s >= 0; --s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const typename P::TW w = tw[((16u * ki) << s) - 1u - (unsigned)((B << s) + g)];
#pragma unroll
      for (int h = 0; h < half; ++h) P::gs(R[g * 2 * half + h], R[g * 2 * half + h + half], w, k);
    }
  }
  row_sync<LB>();
#pragma unroll
  for (int q = 0; q < 16; ++q) lds[pad1<LB>(BS * B + LB * q + l)] = P::unwrap(R[q]);
  row_sync<LB>();
#pragma unroll
  for (int q = 0; q < 16; ++q) R[q] = P::wrap(lds[pad1<LB>(t + W * q)]);
#pragma unroll
  for (int s = 3; s >= 1; --s) {
    const int half = 8 >> s;
    stage_fence();
#pragma unroll
    for (int g = 0; g < (1 << s); ++g) {
      const typename P::TW w = tw[(ki << s) - 1u - (unsigned)g];
#pragma unroll
      for (int h = 0; h < half; ++h) P::gs(R[g * 2 * half + h], R[g * 2 * half + h + half], w, k);
    }
  }
  if (plain_last) {
    const typename P::TW w = tw[ki - 1u];
#pragma unroll
    for (int h = 0; h < 8; ++h) P::gs(R[h], R[h + 8], w, k);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) r[q] = P::unwrap(R[q]);
  P::pass_end(k);
  if (!plain_last) {
#pragma unroll
    for (int h = 0; h < 8; ++h) P::last(r[h], r[h + 8], c, k);  // last stage with n^-1 folded in; canonical outputs
  }
}

// MODE 0: c = INTT(NTT(a) (.) NTT(b));  1: the same with b already in NTT form;  2: dst = NTT(a);  3: dst = INTT(a);
//      4: dst = INTT(a (.) b), both in NTT form (the inner inverse of the composed plan of long rows)
// `store` = false: a surplus row of the last workgroup (it walks through every barrier, writes nothing)
template <class P, int MODE, int LB>
__device__ __forceinline__ void row_body(typename P::T *c, const typename P::T *a, const typename P::T *b, size_t row,
                                         typename P::T *lds, const typename P::TW *tw, const typename P::MC &mcr, int t,
                                         bool store, const unsigned kf = 1u, const unsigned ki = 2u,
                                         const bool plain_last = false) {
  typedef typename P::T T;
  constexpr int W = 16 * LB, LOGN = LB == 4 ? 10 : (LB == 8 ? 11 : 12);
  const typename P::K k = P::make(mcr);
  const T *ar = a + (row << LOGN);
  T ra[16];
  if (MODE == 3 || MODE == 4) {  // NTT-form input: thread holds words 16 t .. 16 t + 15
#pragma unroll
    for (int q = 0; q < 16; ++q) ra[q] = ar[16 * t + q];
    if (MODE == 4) {
      const T *br = b + (row << LOGN);
#pragma unroll
      for (int q = 0; q < 16; ++q) ra[q] = P::mul(ra[q], br[16 * t + q], k);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) ra[j] = ar[t + W * j];
    fwd_row<P, LB>(ra, lds, tw, t, k, kf);
  }
  if (MODE == 2) {
    T *o = c + (row << LOGN) + 16 * t;
    if (store) {
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = P::canon(ra[q], k);
    }
    return;
  }
  if (MODE == 0 || MODE == 1) {
    const T *br = b + (row << LOGN);
    T rb[16];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) rb[j] = br[t + W * j];
      row_sync<LB>();  // the slab is reused
      fwd_row<P, LB>(rb, lds, tw, t, k);
#pragma unroll
      for (int j = 0; j < 16; ++j) rb[j] = P::prep(rb[j], k);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) rb[q] = br[16 * t + q];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) ra[j] = P::mul(ra[j], rb[j], k);
  }
  inv_row<P, LB>(ra, lds, tw, mcr, k, t, ki, plain_last);
  T *cr = c + (row << LOGN);
  if (store) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cr[t + W * j] = ra[j];
  }
}

template <class P, int MODE, int LB>
// 32-bit limbs: 4 waves per SIMD for the fused products (measured round 2, u32/1024/1: 3, 4, 5 waves 198, 201, 193 M
// products/s, 6 and 8 (spills) 169 and 152 -- the kernel is bound by VALU issue, not by latency), 5 for the transforms
__global__ __launch_bounds__(256, (sizeof(typename P::T) == 4 ? (MODE == 0 || (LB == 16 && MODE == 1) ? 4 : 5) : NFLHIP_W64_OCC)) void k_row(
    typename P::T *c, const typename P::T *a, const typename P::T *b, const typename P::TW *__restrict__ psi,
    const typename P::MC *__restrict__ mc, int nm, size_t rows) {
  constexpr int W = 16 * LB, RPB = 256 / W, LOGN = LB == 4 ? 10 : (LB == 8 ? 11 : 12);  // rows per 256-thread block: 4, 2, 1
  __shared__ typename P::T slab[RPB][kSlabWords * (W / 64)];
  const int sub = threadIdx.x / W, t = threadIdx.x % W;
  size_t row = (size_t)blockIdx.x * RPB + sub;
  const bool live = row < rows;
  if (!live) {
    if (LB == 4) return;  // one-wave rows use no workgroup barrier: whole waves may leave
    row = rows - 1;       // two-wave rows: keep walking through the barriers, store nothing
  }
  const int cm = (int)(row % (size_t)nm);
  row_body<P, MODE, LB>(c, a, b, row, slab[sub], psi + ((size_t)cm << LOGN), mc[cm], t, live);
}

// 4096-word blocks of longer rows (LB = 16): block = row * 2^r + blk runs global stages r .. r + 11 of its row (forward,
// MODE 2) or r + 11 .. r (inverse, MODES 3 / 4), the streaming passes of kernels_generic.hip do stages 0 .. r - 1.
template <class P, int MODE>
__global__ __launch_bounds__(256, (sizeof(typename P::T) == 4 ? 5 : 2)) void k_row_block(
    typename P::T *c, const typename P::T *a, const typename P::T *b, const typename P::TW *__restrict__ psi,
    const typename P::MC *__restrict__ mc, int nm, int logn) {
  __shared__ typename P::T slab[kSlabWords * 4];
  const int r = logn - 12;
  const size_t block = blockIdx.x, row = block >> r;
  const unsigned blk = (unsigned)(block & ((((size_t)1) << r) - 1));
  const int cm = (int)(row % (size_t)nm);
  row_body<P, MODE, 16>(c, a, b, block, slab, psi + ((size_t)cm << logn), mc[cm], threadIdx.x, true, (1u << r) + blk,
                        (2u << r) - blk, r > 0);
}

// Persistent variant for one-wave rows and few moduli (NMT <= 4 tables): the twiddle tables are copied into LDS once
// per workgroup and every wave walks over many rows, so the per-lane twiddle reads of the middle stages are LDS reads
// (~100 cycles) instead of L2 reads (~700).
template <class P, int MODE, int NMT>
__global__ __launch_bounds__(256) void k_row1024_lds(typename P::T *c, const typename P::T *a, const typename P::T *b,
                                                     const typename P::TW *__restrict__ psi,
                                                     const typename P::MC *__restrict__ mc, int nm, size_t rows) {
  __shared__ typename P::T slab[4][kSlabWords];
  __shared__ typename P::TW table[NMT][1024];
  for (int i = threadIdx.x; i < nm << 10; i += 256) table[i >> 10][i & 1023] = psi[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * 4;
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += stride) {
    const int cm = (int)(row % (size_t)nm);
    row_body<P, MODE, 4>(c, a, b, row, slab[wave], table[cm], mc[cm], lane, true);
    wave_sync();  // the slab is reused by the next row
  }
}

// c = INTT(b - a (.) key) (SUB) or INTT(b + a (.) key), all in NTT form: the decryption of the reference's demo
// (tests/nfllib_demo_main_op.cpp:51-57) as ONE pass -- the multiply-subtract happens in the registers the inverse transform
// starts from.  key: one polynomial for the batch (kstride 0) or one per element (1).
template <class P, int SUB, int LB>
__global__ __launch_bounds__(256, (sizeof(typename P::T) == 4 ? 5 : NFLHIP_W64_OCC)) void k_row_fma_inv(
    typename P::T *c, const typename P::T *a, const typename P::T *b, const typename P::T *key, int kstride,
    const typename P::TW *__restrict__ psi, const typename P::MC *__restrict__ mc, int nm, size_t rows) {
  typedef typename P::T T;
  constexpr int W = 16 * LB, RPB = 256 / W, LOGN = LB == 4 ? 10 : (LB == 8 ? 11 : 12);
  __shared__ T slab[RPB][kSlabWords * (W / 64)];
  const int sub = threadIdx.x / W, t = threadIdx.x % W;
  size_t row = (size_t)blockIdx.x * RPB + sub;
  const bool live = row < rows;
  if (!live) {
    if (LB == 4) return;
    row = rows - 1;
  }
  const int cm = (int)(row % (size_t)nm);
  const typename P::MC &mcr = mc[cm];
  const typename P::K k = P::make(mcr);
  const T *ar = a + (row << LOGN) + 16 * t, *br = b + (row << LOGN) + 16 * t;
  const T *kr = key + ((kstride ? row : (size_t)cm) << LOGN) + 16 * t;
  T ra[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ra[q] = P::addsub(br[q], P::mul(ar[q], kr[q], k), SUB != 0, k);
  inv_row<P, LB>(ra, slab[sub], psi + ((size_t)cm << LOGN), mcr, k, t, 2u, false);
  T *cr = c + (row << LOGN);
  if (live) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cr[t + W * j] = ra[j];
  }
}
// out0 = NTT(x) (.) k0 + NTT(e0) [, out1 = NTT(x) (.) k1 + NTT(e1)]: the encryption of the reference's demo
// (tests/nfllib_demo_main_op.cpp:26-46) as ONE pass over rows this short -- x is transformed once and stays in registers, each
// noise row is transformed in the same wave(s) and the multiply-add happens in the registers the transform leaves (NTT word
// 16 t + q), against the key row read in that layout.  S = the operands' format: int8_t / int16_t / int32_t = ONE signed
// integer per coefficient shared by the moduli (v < 0 stands for p + v), or T = residue words.
// Compact rows are NOT read lane by lane (round 4 tried that: a wave instruction then moves 64 or 128 bytes and the kernel
// lost to the composed plan): every lane fetches 16 contiguous bytes -- one instruction per wave covers a whole int8 row of
// 1024 coefficients -- into the row's LDS slab, and the lane map of the first pass (x[t + W q]) is read back from there.
template <class P, typename S, int LB>
__device__ __forceinline__ void load_row_small(typename P::T (&r)[16], const S *src, typename P::T *lds, int t, typename P::T p) {
  typedef typename P::T T;
  constexpr int W = 16 * LB;
  if constexpr (sizeof(S) >= 4) {   // 32-bit integers (or words): a wave load is already 256 contiguous bytes
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const S v = src[t + W * q];
      r[q] = std::is_signed<S>::value ? (T)((long long)v < 0 ? (long long)p + (long long)v : (long long)v) : (T)v;
    }
  } else {
    uint4 *stage = reinterpret_cast<uint4 *>(lds);
    const uint4 *g = reinterpret_cast<const uint4 *>(src);
    stage[t] = g[t];                                        // 16 W bytes per step: an int8 row is one step, an int16 row two
    if constexpr (sizeof(S) == 2) stage[t + W] = g[t + W];
    row_sync<LB>();
    const S *l = reinterpret_cast<const S *>(lds);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int v = (int)l[t + W * q];
      r[q] = (T)(v < 0 ? p + (T)(long long)v : (T)v);
    }
    row_sync<LB>();   // (the slab is the transform's exchange buffer next)
  }
}
template <class P, typename S, int TWO, int LB>
__global__ __launch_bounds__(256, (sizeof(typename P::T) == 4 ? 4 : NFLHIP_W64_OCC)) void k_row_fwd_fma(
    typename P::T *out0, typename P::T *out1, const S *x, unsigned xs, const typename P::T *k0, unsigned k0s, const S *e0, unsigned e0s,
    const typename P::T *k1, unsigned k1s, const S *e1, unsigned e1s, const typename P::TW *__restrict__ psi,
    const typename P::MC *__restrict__ mc, int nm, size_t rows) {
  typedef typename P::T T;
  constexpr int W = 16 * LB, RPB = 256 / W, LOGN = LB == 4 ? 10 : (LB == 8 ? 11 : 12);
  constexpr bool words = !std::is_signed<S>::value;
  __shared__ T slab[RPB][kSlabWords * (W / 64)];
  const int sub = threadIdx.x / W, t = threadIdx.x % W;
  size_t row = (size_t)blockIdx.x * RPB + sub;
  const bool live = row < rows;
  if (!live) {
    if (LB == 4) return;
    row = rows - 1;
  }
  const size_t el = row / (size_t)nm;
  const int cm = (int)(row - el * (size_t)nm);
  const typename P::MC &mcr = mc[cm];
  const typename P::K k = P::make(mcr);
  const typename P::TW *tw = psi + ((size_t)cm << LOGN);
  // a compact polynomial is one row of n integers per element; a word polynomial nm rows of n words
  auto in_row = [&](const S *base, unsigned stride) { return base + ((((size_t)stride * el * (words ? (size_t)nm : 1)) + (words ? (size_t)cm : 0)) << LOGN); };
  T rx[16], re[16];
  load_row_small<P, S, LB>(rx, in_row(x, xs), slab[sub], t, (T)mcr.p);
  fwd_row<P, LB>(rx, slab[sub], tw, t, k);
  for (int h = 0; h < (TWO ? 2 : 1); ++h) {   // (left to the compiler: the body holds workgroup barriers for rows of several waves)
    row_sync<LB>();   // the slab is reused
    load_row_small<P, S, LB>(re, in_row(h ? e1 : e0, h ? e1s : e0s), slab[sub], t, (T)mcr.p);
    fwd_row<P, LB>(re, slab[sub], tw, t, k);
    const unsigned ks = h ? k1s : k0s;
    const T *kr = (h ? k1 : k0) + ((((size_t)ks * el) * (size_t)nm + (size_t)cm) << LOGN) + 16 * t;
    T *o = (h ? out1 : out0) + (row << LOGN) + 16 * t;
    T res[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) res[q] = P::canon(P::mul(rx[q], kr[q], k) + P::canon(re[q], k), k);
    if (live) {
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] = res[q];
    }
  }
}
template <class P, typename S, int LB>
static hipError_t launch_fwd_fma_rows_fmt(const Shape &s, const DevTables &t, typename P::T *out0, typename P::T *out1, const void *x, unsigned xs,
                                          const typename P::T *k0, unsigned k0s, const void *e0, unsigned e0s, const typename P::T *k1, unsigned k1s,
                                          const void *e1, unsigned e1s, size_t batch, hipStream_t st) {
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  constexpr int RPB = 256 / (16 * LB);
  const size_t blocks = (rows + RPB - 1) / RPB;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const typename P::TW *psi = (const typename P::TW *)t.psi;
  const typename P::MC *mc = (const typename P::MC *)t.mc;
  if (out1)
    hipLaunchKernelGGL((k_row_fwd_fma<P, S, 1, LB>), dim3((unsigned)blocks), dim3(256), 0, st, out0, out1, (const S *)x, xs, k0, k0s, (const S *)e0, e0s,
                       k1, k1s, (const S *)e1, e1s, psi, mc, (int)s.nm, rows);
  else
    hipLaunchKernelGGL((k_row_fwd_fma<P, S, 0, LB>), dim3((unsigned)blocks), dim3(256), 0, st, out0, out1, (const S *)x, xs, k0, k0s, (const S *)e0, e0s,
                       k0, k0s, (const S *)e0, e0s, psi, mc, (int)s.nm, rows);
  return hipGetLastError();
}
template <class P, int LB>
static hipError_t launch_fwd_fma_rows(const Shape &s, const DevTables &t, int format, typename P::T *out0, typename P::T *out1, const void *x, unsigned xs,
                                      const typename P::T *k0, unsigned k0s, const void *e0, unsigned e0s, const typename P::T *k1, unsigned k1s,
                                      const void *e1, unsigned e1s, size_t batch, hipStream_t st) {
  switch (format) {
    case 0: return launch_fwd_fma_rows_fmt<P, typename P::T, LB>(s, t, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    case 1: return launch_fwd_fma_rows_fmt<P, int8_t, LB>(s, t, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    case 2: return launch_fwd_fma_rows_fmt<P, int16_t, LB>(s, t, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    case 3: return launch_fwd_fma_rows_fmt<P, int32_t, LB>(s, t, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    default: return hipErrorNotSupported;
  }
}
// rows of 1024 / 2048 words (and 4096 for 32-bit limbs); x, e0, e1 of ONE format (NFLHIP_FMT_*), every stride 0 or 1; out1 == nullptr:
// one result.  hipErrorNotSupported otherwise (api.hip composes the same result from the plain kernels)
hipError_t launch_row_fwd_fma_u32(const Shape &s, const DevTables &t, int format, uint32_t *out0, uint32_t *out1, const void *x, unsigned xs,
                                  const uint32_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint32_t *k1, unsigned k1s, const void *e1,
                                  unsigned e1s, size_t batch, hipStream_t st) {
  if (s.limb_bits != 32) return hipErrorNotSupported;
  {  // the generated kernels (tools/gen_row1024_u32_asm.py build_fwd_fma) where they cover the call
    const hipError_t e = launch_row_fwd_fma_u32_asm(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (s.logn == 10) return launch_fwd_fma_rows<Pol32, 4>(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
  if (s.logn == 11) return launch_fwd_fma_rows<Pol32, 8>(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
  if (s.logn == 12) return launch_fwd_fma_rows<Pol32M, 16>(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
  return hipErrorNotSupported;
}
hipError_t launch_row_fwd_fma_u64(const Shape &s, const DevTables &t, int format, uint64_t *out0, uint64_t *out1, const void *x, unsigned xs,
                                  const uint64_t *k0, unsigned k0s, const void *e0, unsigned e0s, const uint64_t *k1, unsigned k1s, const void *e1,
                                  unsigned e1s, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta) return hipErrorNotSupported;
  {  // the generated kernels (tools/asmgen/rows1k.py) where they cover the call
    const hipError_t e = launch_row_fwd_fma_u64_asm(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (s.logn == 10) return launch_fwd_fma_rows<Pol64, 4>(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
  if (s.logn == 11) return launch_fwd_fma_rows<Pol64, 8>(s, t, format, out0, out1, x, xs, k0, k0s, e0, e0s, k1, k1s, e1, e1s, batch, st);
  return hipErrorNotSupported;
}

template <class P, int LB>
static hipError_t launch_fma_inv_rows(const Shape &s, const DevTables &t, int subtract, typename P::T *c, const typename P::T *a,
                                      const typename P::T *key, int kstride, const typename P::T *b, size_t batch, hipStream_t st) {
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  constexpr int RPB = 256 / (16 * LB);
  const size_t blocks = (rows + RPB - 1) / RPB;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const typename P::TW *psi = (const typename P::TW *)t.psi;
  const typename P::MC *mc = (const typename P::MC *)t.mc;
  if (subtract) hipLaunchKernelGGL((k_row_fma_inv<P, 1, LB>), dim3((unsigned)blocks), dim3(256), 0, st, c, a, b, key, kstride, psi, mc, (int)s.nm, rows);
  else hipLaunchKernelGGL((k_row_fma_inv<P, 0, LB>), dim3((unsigned)blocks), dim3(256), 0, st, c, a, b, key, kstride, psi, mc, (int)s.nm, rows);
  return hipGetLastError();
}
// rows of 1024 / 2048 words (and 4096 for 32-bit limbs): dense a / b, the key with stride 0 or 1; hipErrorNotSupported otherwise
hipError_t launch_row_fma_inv_u32(const Shape &s, const DevTables &t, int subtract, uint32_t *c, const uint32_t *a, const uint32_t *key,
                                  int kstride, const uint32_t *b, size_t batch, hipStream_t st) {
  if (s.limb_bits != 32) return hipErrorNotSupported;
  {
    const hipError_t e = launch_row_fma_inv_u32_asm(s, t, subtract, c, a, key, kstride, b, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (s.logn == 10) return launch_fma_inv_rows<Pol32, 4>(s, t, subtract, c, a, key, kstride, b, batch, st);
  if (s.logn == 11) return launch_fma_inv_rows<Pol32, 8>(s, t, subtract, c, a, key, kstride, b, batch, st);
  if (s.logn == 12) return launch_fma_inv_rows<Pol32M, 16>(s, t, subtract, c, a, key, kstride, b, batch, st);
  return hipErrorNotSupported;
}
hipError_t launch_row_fma_inv_u64(const Shape &s, const DevTables &t, int subtract, uint64_t *c, const uint64_t *a, const uint64_t *key,
                                  int kstride, const uint64_t *b, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta) return hipErrorNotSupported;
  {
    const hipError_t e = launch_row_fma_inv_u64_asm(s, t, subtract, c, a, key, kstride, b, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (s.logn == 10) return launch_fma_inv_rows<Pol64, 4>(s, t, subtract, c, a, key, kstride, b, batch, st);
  if (s.logn == 11) return launch_fma_inv_rows<Pol64, 8>(s, t, subtract, c, a, key, kstride, b, batch, st);
  return hipErrorNotSupported;
}

template <class P, int LB>
static hipError_t launch_rows(const Shape &s, const DevTables &t, int mode, typename P::T *c, const typename P::T *a,
                              const typename P::T *b, size_t batch, hipStream_t st) {
  const size_t rows = batch * s.nm;
  if (rows == 0) return hipSuccess;
  constexpr int RPB = 256 / (16 * LB);
  const size_t blocks = (rows + RPB - 1) / RPB;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 bl(256);
  const typename P::TW *psi = (const typename P::TW *)t.psi;
  const typename P::MC *mc = (const typename P::MC *)t.mc;
  const int use_lds = 1;
  // measured (u32/1024/1, batch 2^19): forward 430 -> 482 M/s, inverse 498 -> 514 M/s with the LDS tables; the fused
  // products do not gain (they are bound by VALU issue, not by twiddle latency), so they keep the plain kernel
  if (LB == 4 && sizeof(typename P::T) == 4 && use_lds && mode >= 2 && s.nm <= 4 && blocks >= 4096) {
    const dim3 g(1024);  // 4 resident workgroups per CU (120 VGPRs): every wave walks rows at stride 4096
#define NFLHIP_W_LDS(M, N) hipLaunchKernelGGL((k_row1024_lds<P, M, N>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows)
#define NFLHIP_W_LDS_M(M)                    \
  if (s.nm == 1) NFLHIP_W_LDS(M, 1);         \
  else if (s.nm == 2) NFLHIP_W_LDS(M, 2);    \
  else NFLHIP_W_LDS(M, 4)
    if (mode == 2) { NFLHIP_W_LDS_M(2); } else { NFLHIP_W_LDS_M(3); }
#undef NFLHIP_W_LDS_M
#undef NFLHIP_W_LDS
    return hipGetLastError();
  }
  const dim3 g((unsigned)blocks);
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_row<P, 0, LB>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    case 1: hipLaunchKernelGGL((k_row<P, 1, LB>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    case 2: hipLaunchKernelGGL((k_row<P, 2, LB>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
    default: hipLaunchKernelGGL((k_row<P, 3, LB>), g, bl, 0, st, c, a, b, psi, mc, (int)s.nm, rows); break;
  }
  return hipGetLastError();
}

// mode as in row_body; n = 1024 (one wave per row), 2048 (two waves per row) or, for 32-bit limbs, 4096 (a workgroup per
// row; 64-bit limbs have the assembly kernels there); hipErrorNotSupported otherwise
hipError_t launch_row1024_u32(const Shape &s, const DevTables &t, int mode, uint32_t *c, const uint32_t *a,
                              const uint32_t *b, size_t batch, hipStream_t st) {
  if (s.limb_bits != 32) return hipErrorNotSupported;
  // the generated assembly kernels of the fused product and the stand-alone transforms, n = 1024 / 2048 / 4096 and n = 8
  // (Shape::compiled_only = the compiled kernels below instead).  Measured (MI355X, round 2): 201 -> 243 M products/s at
  // u32/1024/1, 78.9 -> 96.7 M at u32/2048/1, 11.2 -> 13.3 M at u32/4096/4; transforms (batch 2^17) forward / inverse
  // 439 / 408 -> 463 / 444 M at n = 1024, 189 / 207 -> 212 / 234 M at 2048, 25.7 / 23.5 -> 28.9 / 30.6 M at 4096/4
  if (!s.compiled_only && ((s.logn >= 10 && s.logn <= 12 && (mode == 0 || mode == 2 || mode == 3)) || s.logn == 3)) {
    const hipError_t e = launch_row1024_u32_asm(s, t, mode, c, a, b, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  // multiply-add butterflies (Pol32M) where they measured faster: the 4096-word rows
  if (s.logn == 10) return launch_rows<Pol32, 4>(s, t, mode, c, a, b, batch, st);
  if (s.logn == 11) return launch_rows<Pol32, 8>(s, t, mode, c, a, b, batch, st);
  if (s.logn == 12) return launch_rows<Pol32M, 16>(s, t, mode, c, a, b, batch, st);
  return hipErrorNotSupported;
}
// 4096-word blocks of rows longer than 4096 words, 32-bit limbs: the inner kernels of launch_ntt_fwd / launch_ntt_inv
hipError_t launch_inner_fwd_fast_u32(const Shape &s, const DevTables &t, const uint32_t *src, uint32_t *dst, size_t rows,
                                     hipStream_t st) {
  if (s.limb_bits != 32 || s.logn <= 12) return hipErrorNotSupported;
  const size_t blocks = rows << (s.logn - 12);
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_row_block<Pol32M, 2>), dim3((unsigned)blocks), dim3(256), 0, st, dst, src, (const uint32_t *)nullptr,
                     (const Tw32 *)t.psi, (const MC32 *)t.mc, (int)s.nm, s.logn);
  return hipGetLastError();
}
hipError_t launch_inner_inv_fast_u32(const Shape &s, const DevTables &t, const uint32_t *src, const uint32_t *mul,
                                     uint32_t *dst, size_t rows, hipStream_t st) {
  if (s.limb_bits != 32 || s.logn <= 12) return hipErrorNotSupported;
  const size_t blocks = rows << (s.logn - 12);
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  if (mul)
    hipLaunchKernelGGL((k_row_block<Pol32M, 4>), dim3((unsigned)blocks), dim3(256), 0, st, dst, src, mul, (const Tw32 *)t.psi,
                       (const MC32 *)t.mc, (int)s.nm, s.logn);
  else
    hipLaunchKernelGGL((k_row_block<Pol32M, 3>), dim3((unsigned)blocks), dim3(256), 0, st, dst, src, (const uint32_t *)nullptr,
                       (const Tw32 *)t.psi, (const MC32 *)t.mc, (int)s.nm, s.logn);
  return hipGetLastError();
}

hipError_t launch_row1024_u64(const Shape &s, const DevTables &t, int mode, uint64_t *c, const uint64_t *a,
                              const uint64_t *b, size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta) return hipErrorNotSupported;
  // the generated kernels of the fused product and the stand-alone transforms (tools/asmgen/rows1k.py; Shape::compiled_only = the
  // compiled template below instead)
  if (!s.compiled_only && (s.logn == 10 || s.logn == 11) && (mode == 0 || mode == 2 || mode == 3)) {
    const hipError_t e = launch_row1024_u64_asm(s, t, mode, c, a, b, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (s.logn == 10) return launch_rows<Pol64, 4>(s, t, mode, c, a, b, batch, st);
  if (s.logn == 11) return launch_rows<Pol64, 8>(s, t, mode, c, a, b, batch, st);
  return hipErrorNotSupported;
}

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_wave() {}
hipError_t warm_wave(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_wave, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
