// kernels_generic.hip -- shape-generic gfx950 kernels of the NTT polynomial-ring
// engine: negacyclic NTT (any power-of-two degree, any limb width), element-wise
// modular ops, comparisons, seeded input generation and CRT lift/project.
//
// These are the correctness-first paths every shape can take; the tuned
// register-tiled kernels for the headline shape live in kernels_fast.hip.
//
// Algorithm (NOT the reference's loop structure): the reference computes
//   phi-twist (core.hpp:596) -> cyclic Harvey DIF NTT (core.hpp:455-532) and, for
//   the inverse, bit-reverse -> NTT -> bit-reverse -> scale/untwist (core.hpp:539-557, 613).
// Its forward output is out[bitrev(k)] = a(phi^(2k+1)) in [0,p), which is exactly
// what a merged-twiddle Cooley-Tukey negacyclic transform over the table
// psi_br[k] = phi^bitrev(k) produces in place; the inverse is the mirrored
// Gentleman-Sande network over the SAME table, using
//   psi_br[m+j]^-1 = -psi_br[m + (m-1-j)]   and   (u-v)*(-W) = (v-u)*W,
// with n^-1 folded into the last stage.  Every public word is the canonical
// representative, so results are bit-identical to the reference's.
#include "kernels.h"
#include "modarith.h"

#include <cstdlib>
#include <type_traits>

namespace nflhip {

static constexpr int kInnerLogMax = 12;  // rows up to 4096 words are transformed inside LDS

// ---------------------------------------------------------------------------
// forward, LDS-resident stages [logn-logi, logn)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_ntt_fwd_lds(const T *src, T *dst, const Tw<T> *__restrict__ psi,
                              const ModConst<T> *__restrict__ mc, int logn, int logi, int nm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  const int s0 = logn - logi;
  const size_t row = (size_t)blockIdx.x >> s0;
  const unsigned blk = blockIdx.x & ((1u << s0) - 1u);
  const int cm = (int)(row % (size_t)nm);
  const T p = mc[cm].p, p2 = mc[cm].p2;
  const Tw<T> *tw = psi + ((size_t)cm << logn);
  const unsigned I = 1u << logi;
  const size_t base = (row << logn) + ((size_t)blk << logi);
  for (unsigned i = threadIdx.x; i < I; i += blockDim.x) sm[i] = src[base + i];
  for (int s = s0; s < logn; ++s) {
    const int lt = logn - s - 1;  // log2 of the half-block length t
    const unsigned t = 1u << lt;
    const unsigned jbase = (1u << s) + (blk << (s - s0));
    __syncthreads();
    for (unsigned q = threadIdx.x; q < (I >> 1); q += blockDim.x) {
      const unsigned jl = q >> lt, o = q & (t - 1u);
      const unsigned i = (jl << (lt + 1)) + o;
      const Tw<T> w = tw[jbase + jl];
      T x = sm[i];
      const T y = sm[i + t];
      x = csub<T>(x, p2);                                     // [0,4p) -> [0,2p)
      const T m = mul_shoup_lazy<T>(y, w.w, w.wp, p);         // [0,2p)
      sm[i] = (T)(x + m);                                     // [0,4p)
      sm[i + t] = (T)(x - m + p2);                            // [0,4p)
    }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < I; i += blockDim.x) dst[base + i] = reduce4<T>(sm[i], p);
}

// forward, streaming radix-2^R pass over global stages [done, done+R): strides >= 4096 words
template <typename T, int R>
__global__ void k_ntt_fwd_outer(const T *src, T *dst, const Tw<T> *__restrict__ psi,
                                const ModConst<T> *__restrict__ mc, int logn, int done, int nm) {
  constexpr int E = 1 << R;
  const int lstride = logn - done - R;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per E-point column
  const size_t row = gid >> (logn - R);
  const unsigned q = (unsigned)(gid & ((((size_t)1) << (logn - R)) - 1));
  const unsigned j = q >> lstride, o = q & ((1u << lstride) - 1u);
  const int cm = (int)(row % (size_t)nm);
  const T p = mc[cm].p, p2 = mc[cm].p2;
  const Tw<T> *tw = psi + ((size_t)cm << logn);
  const size_t base = (row << logn) + ((size_t)j << (lstride + R)) + o;
  T v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = src[base + ((size_t)k << lstride)];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int half = 1 << (R - 1 - r);
#pragma unroll
    for (int g = 0; g < (1 << r); ++g) {
      const Tw<T> w = tw[(1u << (done + r)) + (j << r) + (unsigned)g];
#pragma unroll
      for (int h = 0; h < half; ++h) {
        const int i0 = g * 2 * half + h, i1 = i0 + half;
        const T x = csub<T>(v[i0], p2);
        const T m = mul_shoup_lazy<T>(v[i1], w.w, w.wp, p);
        v[i0] = (T)(x + m);
        v[i1] = (T)(x - m + p2);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < E; ++k) dst[base + ((size_t)k << lstride)] = v[k];  // stays lazy in [0,4p)
}

// ---------------------------------------------------------------------------
// inverse, LDS-resident stages logn-1 down to logn-logi (Gentleman-Sande)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_ntt_inv_lds(const T *src, const T *mul, T *dst, const Tw<T> *__restrict__ psi,
                              const ModConst<T> *__restrict__ mc, int logn, int logi, int nm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  const int s0 = logn - logi;
  const size_t row = (size_t)blockIdx.x >> s0;
  const unsigned blk = blockIdx.x & ((1u << s0) - 1u);
  const int cm = (int)(row % (size_t)nm);
  const ModConst<T> c = mc[cm];
  const T p = c.p, p2 = c.p2;
  const Tw<T> *tw = psi + ((size_t)cm << logn);
  const unsigned I = 1u << logi;
  const size_t base = (row << logn) + ((size_t)blk << logi);
  if (mul != nullptr) {
    for (unsigned i = threadIdx.x; i < I; i += blockDim.x)
      sm[i] = barrett<T>::mul(src[base + i], mul[base + i], p, c.mu);  // fused point-wise product
  } else {
    for (unsigned i = threadIdx.x; i < I; i += blockDim.x) sm[i] = src[base + i];
  }
  for (int s = logn - 1; s >= s0; --s) {
    const int lt = logn - s - 1;
    const unsigned t = 1u << lt;
    const unsigned m = 1u << s;
    const unsigned jg0 = blk << (s - s0);
    __syncthreads();
    if (s > 0) {
      for (unsigned q = threadIdx.x; q < (I >> 1); q += blockDim.x) {
        const unsigned jl = q >> lt, o = q & (t - 1u);
        const unsigned i = (jl << (lt + 1)) + o;
        const Tw<T> w = tw[m + (m - 1u - (jg0 + jl))];  // -(psi_br[m+j])^-1
        const T u = sm[i], v = sm[i + t];
        sm[i] = csub<T>((T)(u + v), p2);
        sm[i + t] = mul_shoup_lazy<T>((T)(v - u + p2), w.w, w.wp, p);
      }
    } else {  // last stage: fold n^-1, emit canonical words
      for (unsigned q = threadIdx.x; q < (I >> 1); q += blockDim.x) {
        const T u = sm[q], v = sm[q + t];
        sm[q] = mul_shoup<T>((T)(u + v), c.ninv, c.ninv_sh, p);
        sm[q + t] = mul_shoup<T>((T)(v - u + p2), c.w1ninv, c.w1ninv_sh, p);
      }
    }
  }
  __syncthreads();
  // when outer passes follow, words stay lazy in [0,2p); otherwise they are canonical already
  for (unsigned i = threadIdx.x; i < I; i += blockDim.x) dst[base + i] = sm[i];
}

// inverse, streaming radix-2^R pass over global stages [sa, sa+R), processed high to low
template <typename T, int R>
__global__ void k_ntt_inv_outer(const T *src, T *dst, const Tw<T> *__restrict__ psi,
                                const ModConst<T> *__restrict__ mc, int logn, int sa, int nm) {
  constexpr int E = 1 << R;
  const int lstride = logn - sa - R;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = gid >> (logn - R);
  const unsigned q = (unsigned)(gid & ((((size_t)1) << (logn - R)) - 1));
  const unsigned j = q >> lstride, o = q & ((1u << lstride) - 1u);
  const int cm = (int)(row % (size_t)nm);
  const ModConst<T> c = mc[cm];
  const T p = c.p, p2 = c.p2;
  const Tw<T> *tw = psi + ((size_t)cm << logn);
  const size_t base = (row << logn) + ((size_t)j << (lstride + R)) + o;
  T v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = src[base + ((size_t)k << lstride)];
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    const int half = 1 << (R - 1 - r);
    const unsigned m = 1u << (sa + r);
#pragma unroll
    for (int g = 0; g < (1 << r); ++g) {
      if (sa + r > 0) {
        const Tw<T> w = tw[m + (m - 1u - ((j << r) + (unsigned)g))];
#pragma unroll
        for (int h = 0; h < half; ++h) {
          const int i0 = g * 2 * half + h, i1 = i0 + half;
          const T u = v[i0], x = v[i1];
          v[i0] = csub<T>((T)(u + x), p2);
          v[i1] = mul_shoup_lazy<T>((T)(x - u + p2), w.w, w.wp, p);
        }
      } else {
#pragma unroll
        for (int h = 0; h < half; ++h) {
          const int i0 = g * 2 * half + h, i1 = i0 + half;
          const T u = v[i0], x = v[i1];
          v[i0] = mul_shoup<T>((T)(u + x), c.ninv, c.ninv_sh, p);
          v[i1] = mul_shoup<T>((T)(x - u + p2), c.w1ninv, c.w1ninv_sh, p);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < E; ++k) dst[base + ((size_t)k << lstride)] = v[k];
}

// ---------------------------------------------------------------------------
// launch plans
// ---------------------------------------------------------------------------
static inline int inner_log(const Shape &s) { return s.logn < kInnerLogMax ? s.logn : kInnerLogMax; }
static inline unsigned lds_threads(int logi) {
  const unsigned half = 1u << (logi > 0 ? logi - 1 : 0);
  return half < 64u ? 64u : (half > 256u ? 256u : half);
}

template <typename T>
static hipError_t outer_fwd(const Shape &s, const DevTables &t, const T *src, T *dst, size_t rows, int done, int R,
                            hipStream_t st) {
  const size_t threads = rows << (s.logn - R);
  const dim3 block(256), grid((unsigned)(threads / 256));
  const Tw<T> *psi = (const Tw<T> *)t.psi;
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  switch (R) {
    case 1: hipLaunchKernelGGL((k_ntt_fwd_outer<T, 1>), grid, block, 0, st, src, dst, psi, mc, s.logn, done, (int)s.nm); break;
    case 2: hipLaunchKernelGGL((k_ntt_fwd_outer<T, 2>), grid, block, 0, st, src, dst, psi, mc, s.logn, done, (int)s.nm); break;
    case 3: hipLaunchKernelGGL((k_ntt_fwd_outer<T, 3>), grid, block, 0, st, src, dst, psi, mc, s.logn, done, (int)s.nm); break;
    default: hipLaunchKernelGGL((k_ntt_fwd_outer<T, 4>), grid, block, 0, st, src, dst, psi, mc, s.logn, done, (int)s.nm); break;
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t outer_inv(const Shape &s, const DevTables &t, const T *src, T *dst, size_t rows, int sa, int R,
                            hipStream_t st) {
  const size_t threads = rows << (s.logn - R);
  const dim3 block(256), grid((unsigned)(threads / 256));
  const Tw<T> *psi = (const Tw<T> *)t.psi;
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  switch (R) {
    case 1: hipLaunchKernelGGL((k_ntt_inv_outer<T, 1>), grid, block, 0, st, src, dst, psi, mc, s.logn, sa, (int)s.nm); break;
    case 2: hipLaunchKernelGGL((k_ntt_inv_outer<T, 2>), grid, block, 0, st, src, dst, psi, mc, s.logn, sa, (int)s.nm); break;
    case 3: hipLaunchKernelGGL((k_ntt_inv_outer<T, 3>), grid, block, 0, st, src, dst, psi, mc, s.logn, sa, (int)s.nm); break;
    default: hipLaunchKernelGGL((k_ntt_inv_outer<T, 4>), grid, block, 0, st, src, dst, psi, mc, s.logn, sa, (int)s.nm); break;
  }
  return hipGetLastError();
}

// all streaming forward passes (global stages [0, logn - logi)): src -> dst, then in place on dst
template <typename T>
static hipError_t outer_fwd_all(const Shape &s, const DevTables &t, const T *src, T *dst, size_t rows, hipStream_t st,
                                int logi_override = 0) {
  const int logi = logi_override ? logi_override : inner_log(s);
  int done = 0;
  const T *cur = src;
  while (done < s.logn - logi) {
    const int rem = s.logn - logi - done;
    const int R = rem >= 4 ? 4 : rem;
    hipError_t e = outer_fwd<T>(s, t, cur, dst, rows, done, R, st);
    if (e != hipSuccess) return e;
    cur = dst;
    done += R;
  }
  return hipSuccess;
}
// all streaming inverse passes (global stages logn - logi - 1 .. 0), in place
template <typename T>
static hipError_t outer_inv_all(const Shape &s, const DevTables &t, T *data, size_t rows, hipStream_t st,
                                int logi_override = 0) {
  int top = s.logn - (logi_override ? logi_override : inner_log(s));
  while (top > 0) {
    const int R = top >= 4 ? 4 : top;
    hipError_t e = outer_inv<T>(s, t, data, data, rows, top - R, R, st);
    if (e != hipSuccess) return e;
    top -= R;
  }
  return hipSuccess;
}
hipError_t launch_outer_fwd_u64(const Shape &s, const DevTables &t, const uint64_t *src, uint64_t *dst, size_t rows,
                                hipStream_t st, int logi) {
  return outer_fwd_all<uint64_t>(s, t, src, dst, rows, st, logi);
}
hipError_t launch_outer_inv_u64(const Shape &s, const DevTables &t, uint64_t *data, size_t rows, hipStream_t st, int logi) {
  return outer_inv_all<uint64_t>(s, t, data, rows, st, logi);
}

template <typename T>
hipError_t launch_ntt_fwd(const Shape &s, const DevTables &t, const T *src, T *dst, size_t batch, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t rows = batch * s.nm;
  const int logi = inner_log(s);
  hipError_t oe = outer_fwd_all<T>(s, t, src, dst, rows, st);
  if (oe != hipSuccess) return oe;
  const T *cur = s.logn > logi ? dst : src;
  if (std::is_same<T, uint64_t>::value && logi == kInnerLogMax) {  // register-tiled 4096-word blocks
    hipError_t e = launch_inner_fwd_fast_u64(s, t, (const uint64_t *)cur, (uint64_t *)dst, rows, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (std::is_same<T, uint32_t>::value && logi == kInnerLogMax) {  // register-tiled 4096-word blocks, 30-bit moduli
    hipError_t e = launch_inner_fwd_fast_u32(s, t, (const uint32_t *)cur, (uint32_t *)dst, rows, st);
    if (e != hipErrorNotSupported) return e;
  }
  const unsigned nblk = (unsigned)(rows << (s.logn - logi));
  hipLaunchKernelGGL((k_ntt_fwd_lds<T>), dim3(nblk), dim3(lds_threads(logi)), sizeof(T) << logi, st, cur, dst,
                     (const Tw<T> *)t.psi, (const ModConst<T> *)t.mc, s.logn, logi, (int)s.nm);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_ntt_inv(const Shape &s, const DevTables &t, const T *src, const T *mul, T *dst, size_t batch,
                          hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t rows = batch * s.nm;
  const int logi = inner_log(s);
  hipError_t e = hipErrorNotSupported;
  if (std::is_same<T, uint64_t>::value && logi == kInnerLogMax)  // register-tiled 4096-word blocks
    e = launch_inner_inv_fast_u64(s, t, (const uint64_t *)src, (const uint64_t *)mul, (uint64_t *)dst, rows, st);
  if (std::is_same<T, uint32_t>::value && logi == kInnerLogMax)
    e = launch_inner_inv_fast_u32(s, t, (const uint32_t *)src, (const uint32_t *)mul, (uint32_t *)dst, rows, st);
  if (e == hipErrorNotSupported) {
    const unsigned nblk = (unsigned)(rows << (s.logn - logi));
    hipLaunchKernelGGL((k_ntt_inv_lds<T>), dim3(nblk), dim3(lds_threads(logi)), sizeof(T) << logi, st, src, mul, dst,
                       (const Tw<T> *)t.psi, (const ModConst<T> *)t.mc, s.logn, logi, (int)s.nm);
    e = hipGetLastError();
  }
  if (e != hipSuccess) return e;
  return outer_inv_all<T>(s, t, dst, rows, st);  // global stages [0, logn - logi) remain, high to low
}

// ---------------------------------------------------------------------------
// element-wise ops (the functors of poly::operator=(expr), core.hpp:24-37)
// ---------------------------------------------------------------------------
template <typename T, int OP>
__global__ void k_pointwise(T *out, const T *a, const T *b, const T *bp, const ModConst<T> *__restrict__ mc, int logn,
                            int nm, size_t total) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) Vec { T e[V]; };
  const size_t nvec = total / V;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const Vec va = reinterpret_cast<const Vec *>(a)[v];
    Vec vb = va, vp = va, vo;
    if (OP != 4) vb = reinterpret_cast<const Vec *>(b)[v];
    if (OP == 3) vp = reinterpret_cast<const Vec *>(bp)[v];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int cm = (int)((((v * V + k) >> logn)) % (size_t)nm);
      const T p = mc[cm].p;
      T r;
      if (OP == 0) r = csub<T>((T)(va.e[k] + vb.e[k]), p);                       // addmod
      else if (OP == 1) r = csub<T>((T)(va.e[k] + (T)(p - vb.e[k])), p);           // submod = addmod(x, p-y)
      else if (OP == 2) r = barrett<T>::mul(va.e[k], vb.e[k], p, mc[cm].mu);       // mulmod
      else if (OP == 3) r = mul_shoup<T>(va.e[k], vb.e[k], vp.e[k], p);            // mulmod_shoup
      else {                                                                       // compute_shoup
        T x = va.e[k];
        x = csub<T>(x, (T)(4 * p)); x = csub<T>(x, (T)(2 * p)); x = csub<T>(x, p);
        if (sizeof(T) < 8) { while (x >= p) x -= p; }
        r = shoup_of<T>::get(x, p, mc[cm].mu);
      }
      vo.e[k] = r;
    }
    reinterpret_cast<Vec *>(out)[v] = vo;
  }
}

// Grid of the grid-stride streaming kernels: the chip's copy rate peaks with two to four 256-thread workgroups per CU
// (measured on the point-wise kernels, u64/4096/4, batch 16 384: add 5.43 / 5.23 / 5.13 / 4.74 TB/s and mul 5.25 / 5.14 /
// 5.50 / 5.00 TB/s at 512 / 768 / 1024 / 4096 workgroups; copy kernel: 5.8 TB/s at 1024 against 4.7 at 4096,
// profiles/r02_ubench_gfx950.txt) -- more resident waves only add DRAM page conflicts.
// The interpreter loop of the expression kernel is the opposite case (it needs the waves to hide its own latency:
// a*b+d 3.1 TB/s at 768 workgroups, 4.9 at 4096), comparisons and fills are indifferent: they keep their wide grids.
static inline size_t stream_blocks(size_t items_per_thread_total, size_t dflt = 768) {
  const size_t cap = dflt;
  size_t blocks = (items_per_thread_total + 255) / 256;
  if (blocks > cap) blocks = cap;
  return blocks ? blocks : 1;
}

template <typename T>
hipError_t launch_pointwise(const Shape &s, const DevTables &t, int op, T *out, const T *a, const T *b, const T *bp,
                            size_t batch, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t total = batch * s.nm * s.n;
  constexpr size_t V = 16 / sizeof(T);
  if (total % V) return hipErrorInvalidValue;
  const size_t nvec = total / V;
  const size_t blocks = stream_blocks(nvec);
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  const dim3 g((unsigned)blocks), bl(256);
  switch (op) {
    case 0: hipLaunchKernelGGL((k_pointwise<T, 0>), g, bl, 0, st, out, a, b, bp, mc, s.logn, (int)s.nm, total); break;
    case 1: hipLaunchKernelGGL((k_pointwise<T, 1>), g, bl, 0, st, out, a, b, bp, mc, s.logn, (int)s.nm, total); break;
    case 2: hipLaunchKernelGGL((k_pointwise<T, 2>), g, bl, 0, st, out, a, b, bp, mc, s.logn, (int)s.nm, total); break;
    case 3: hipLaunchKernelGGL((k_pointwise<T, 3>), g, bl, 0, st, out, a, b, bp, mc, s.logn, (int)s.nm, total); break;
    case 4: hipLaunchKernelGGL((k_pointwise<T, 4>), g, bl, 0, st, out, a, b, bp, mc, s.logn, (int)s.nm, total); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fused expression trees: poly::operator=(expr) evaluates an arbitrary tree of the
// element-wise functors in ONE pass with no temporaries (core.hpp:24-37, ops.hpp:52-79).
// The tree arrives in postfix form (include/nflhip.h: NFLHIP_EXPR_*); every lane runs
// the same tiny program over a 4-deep register stack.
// ---------------------------------------------------------------------------
struct ExprProgram {
  unsigned char code[24];
  int len;
  const void *operand[8];
};

template <typename T> struct ExprStack {
  T s0, s1, s2, s3;
  __device__ __forceinline__ void push(T v) { s3 = s2; s2 = s1; s1 = s0; s0 = v; }
  __device__ __forceinline__ void drop() { s0 = s1; s1 = s2; s2 = s3; }
};

template <typename T>
__global__ void k_eval_expr(T *out, ExprProgram prog, const ModConst<T> *__restrict__ mc, int logn, int nm, size_t total) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) Vec { T e[V]; };
  const size_t nvec = total / V;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const int cm = (int)(((v * V) >> logn) % (size_t)nm);  // n >= V words: a vector never straddles moduli
    const T p = mc[cm].p, mu = mc[cm].mu;
    ExprStack<T> st[V];
    for (int pc = 0; pc < prog.len; ++pc) {
      const unsigned c = prog.code[pc];
      if (c < 8) {
        const Vec x = reinterpret_cast<const Vec *>(prog.operand[c])[v];
#pragma unroll
        for (int k = 0; k < V; ++k) st[k].push(x.e[k]);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          ExprStack<T> &s = st[k];
          switch (c) {
            case 0x10: { const T r = csub<T>((T)(s.s1 + s.s0), p); s.drop(); s.s0 = r; } break;            // addmod
            case 0x11: { const T r = csub<T>((T)(s.s1 + (T)(p - s.s0)), p); s.drop(); s.s0 = r; } break;    // submod
            case 0x12: { const T r = barrett<T>::mul(s.s1, s.s0, p, mu); s.drop(); s.s0 = r; } break;       // mulmod
            case 0x13: { const T r = mul_shoup<T>(s.s2, s.s1, s.s0, p); s.drop(); s.drop(); s.s0 = r; } break;  // a b b' -> mulmod_shoup
            default: {                                                                                        // compute_shoup
              T x = s.s0;
              x = csub<T>(x, (T)(4 * p)); x = csub<T>(x, (T)(2 * p)); x = csub<T>(x, p);
              if (sizeof(T) < 8) { while (x >= p) x -= p; }
              s.s0 = shoup_of<T>::get(x, p, mu);
            } break;
          }
        }
      }
    }
    Vec o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.e[k] = st[k].s0;
    reinterpret_cast<Vec *>(out)[v] = o;
  }
}

// The same program over a batch whose operands (and result) advance by their own stride from one polynomial to the
// next: stride 1 = a dense array of polynomials, 0 = ONE polynomial shared by the whole batch (a key), k = every k-th
// polynomial of an interleaved array.  blockIdx.y = polynomial, so no division is needed to find it.
struct ExprStrides {
  unsigned op[8];
  unsigned out;
};
template <typename T>
__global__ void k_eval_expr_strided(T *out, ExprProgram prog, ExprStrides sd, const ModConst<T> *__restrict__ mc, int logn,
                                    int nm) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) Vec { T e[V]; };
  const size_t vpp = (((size_t)nm) << logn) / V;  // vectors per polynomial
  const size_t poly = blockIdx.y;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < vpp; w += (size_t)gridDim.x * blockDim.x) {
    const int cm = (int)((w * V) >> logn);
    const T p = mc[cm].p, mu = mc[cm].mu;
    ExprStack<T> st[V];
    for (int pc = 0; pc < prog.len; ++pc) {
      const unsigned c = prog.code[pc];
      if (c < 8) {
        const Vec x = reinterpret_cast<const Vec *>(prog.operand[c])[poly * sd.op[c] * vpp + w];
#pragma unroll
        for (int k = 0; k < V; ++k) st[k].push(x.e[k]);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          ExprStack<T> &s = st[k];
          switch (c) {
            case 0x10: { const T r = csub<T>((T)(s.s1 + s.s0), p); s.drop(); s.s0 = r; } break;
            case 0x11: { const T r = csub<T>((T)(s.s1 + (T)(p - s.s0)), p); s.drop(); s.s0 = r; } break;
            case 0x12: { const T r = barrett<T>::mul(s.s1, s.s0, p, mu); s.drop(); s.s0 = r; } break;
            case 0x13: { const T r = mul_shoup<T>(s.s2, s.s1, s.s0, p); s.drop(); s.drop(); s.s0 = r; } break;
            default: {
              T x = s.s0;
              x = csub<T>(x, (T)(4 * p)); x = csub<T>(x, (T)(2 * p)); x = csub<T>(x, p);
              if (sizeof(T) < 8) { while (x >= p) x -= p; }
              s.s0 = shoup_of<T>::get(x, p, mu);
            } break;
          }
        }
      }
    }
    Vec o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.e[k] = st[k].s0;
    reinterpret_cast<Vec *>(out)[poly * sd.out * vpp + w] = o;
  }
}

template <typename T>
hipError_t launch_eval_expr(const Shape &s, const DevTables &t, T *out, const void *const *operands, int noperands,
                            const unsigned char *program, int len, size_t batch, hipStream_t st,
                            const unsigned *strides, unsigned out_stride) {
  if (batch == 0) return hipSuccess;
  if (len <= 0 || len > 24 || noperands < 1 || noperands > 8) return hipErrorInvalidValue;
  const size_t total = batch * s.nm * s.n;
  constexpr size_t V = 16 / sizeof(T);
  if (total % V || s.n < V) return hipErrorNotSupported;  // tiny rows: the caller evaluates node by node
  ExprProgram prog;
  int depth = 0;
  for (int i = 0; i < len; ++i) {  // validate: operands in range, stack depth 1..4, exactly one result
    const unsigned c = program[i];
    prog.code[i] = (unsigned char)c;
    if (c < 8) { if ((int)c >= noperands) return hipErrorInvalidValue; ++depth; }
    else if (c == 0x10 || c == 0x11 || c == 0x12) { if (depth < 2) return hipErrorInvalidValue; --depth; }
    else if (c == 0x13) { if (depth < 3) return hipErrorInvalidValue; depth -= 2; }
    else if (c == 0x14) { if (depth < 1) return hipErrorInvalidValue; }
    else return hipErrorInvalidValue;
    if (depth > 4) return hipErrorInvalidValue;
  }
  if (depth != 1) return hipErrorInvalidValue;
  prog.len = len;
  for (int i = 0; i < 8; ++i) prog.operand[i] = i < noperands ? operands[i] : nullptr;
  if (strides) {
    if (batch > 65535 * 1024u) return hipErrorInvalidValue;
    ExprStrides sd;
    for (int i = 0; i < 8; ++i) sd.op[i] = i < noperands ? strides[i] : 0;
    sd.out = out_stride;
    const size_t vpp = s.nm * s.n / V;
    size_t bx = (vpp + 255) / 256;
    if (bx > 64) bx = 64;
    // (every polynomial is a grid row: all of them are launched, the column count is what can be trimmed)
    while (bx > 1 && bx * (batch < 65535 ? batch : 65535) > 4096) bx /= 2;
    // grid.y is limited to 65535: longer batches go in slices (the strides advance the base pointers)
    for (size_t lo = 0; lo < batch; lo += 65535) {
      const size_t cnt = batch - lo < 65535 ? batch - lo : 65535;
      ExprProgram pp = prog;
      for (int i = 0; i < noperands; ++i) pp.operand[i] = (const T *)prog.operand[i] + lo * sd.op[i] * s.nm * s.n;
      hipLaunchKernelGGL((k_eval_expr_strided<T>), dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st,
                         out + lo * sd.out * s.nm * s.n, pp, sd, (const ModConst<T> *)t.mc, s.logn, (int)s.nm);
    }
    return hipGetLastError();
  }
  const size_t nvec = total / V;
  const size_t blocks = stream_blocks(nvec, 256 * 16);
  hipLaunchKernelGGL((k_eval_expr<T>), dim3((unsigned)blocks), dim3(256), 0, st, out, prog, (const ModConst<T> *)t.mc, s.logn,
                     (int)s.nm, total);
  return hipGetLastError();
}

// expr::operator bool over eqmod / neqmod (ops.hpp:81-117): "any word" semantics
template <typename T>
__global__ void k_any_cmp(const T *a, const T *b, size_t total, int want_eq, int *flag, int token) {
  int hit = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    hit |= ((a[i] == b[i]) == (want_eq != 0)) ? 1 : 0;
  if (__any(hit)) {
    // every wave that saw a hit stores the SAME word: the call's token (no clearing pass in front of the kernel, no atomic -- the flag may
    // live in pinned host memory, where the caller reads it after the stream has drained)
    if ((threadIdx.x & 63) == 0) *flag = token;
  }
}

template <typename T>
hipError_t launch_any_cmp(const Shape &s, const DevTables &t, const T *a, const T *b, size_t batch, int want_eq,
                          int *flag, int token, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t total = batch * s.nm * s.n;
  const size_t blocks = stream_blocks(total, 2048);
  hipLaunchKernelGGL((k_any_cmp<T>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, total, want_eq, flag, token);
  return hipGetLastError();
}

// CHECK_STRICTMOD (debug.hpp:33-37; the asserts of ops.hpp:131,148,211,235 and core.hpp:457-462): any word that is not the
// canonical representative of its row's modulus
template <typename T>
__global__ void k_check_range(const T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, size_t total, int *flag, int token) {
  int hit = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    hit |= d[i] >= mc[(int)((i >> logn) % (size_t)nm)].p ? 1 : 0;
  if (__any(hit)) {
    if ((threadIdx.x & 63) == 0) *flag = token;   // (see k_any_cmp)
  }
}

template <typename T>
hipError_t launch_check_range(const Shape &s, const DevTables &t, const T *d, size_t batch, int *flag, int token, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t total = batch * s.nm * s.n;
  hipLaunchKernelGGL((k_check_range<T>), dim3((unsigned)stream_blocks(total, 2048)), dim3(256), 0, st, d, (const ModConst<T> *)t.mc,
                     s.logn, (int)s.nm, total, flag, token);
  return hipGetLastError();
}

// seeded synthetic operands: mask-then-subtract rule of nfl::uniform (core.hpp:165-176)
template <typename T>
__global__ void k_fill_uniform(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, size_t first_word,
                               size_t total, uint64_t seed, int operand) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t g = first_word + i;
    const int cm = (int)((g >> logn) % (size_t)nm);
    const uint64_t p = mc[cm].p;
    uint64_t v = splitmix64_at(seed, operand, g) & (uint64_t)mc[cm].mask;
    if (v >= p) v -= p;
    d[i] = (T)v;
  }
}

template <typename T>
hipError_t launch_fill_uniform(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch, uint64_t seed,
                               int operand, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const size_t total = batch * s.nm * s.n;
  const size_t blocks = stream_blocks(total, 256 * 32);
  hipLaunchKernelGGL((k_fill_uniform<T>), dim3((unsigned)blocks), dim3(256), 0, st, d, (const ModConst<T> *)t.mc, s.logn,
                     (int)s.nm, first_poly * s.nm * s.n, total, seed, operand);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// CRT lift (GMP::poly2mpz, gmp.hpp:183-209).  The reference accumulates
// sum_cm lifting[cm]*x(cm,i) and Barrett-reduces mod Q; the value in [0,Q) is
// unique, so we use the equivalent small-quotient form
//   X = sum_cm (Q/p_cm) * ((x(cm,i) * (Q/p_cm)^-1) mod p_cm)   (< nm*Q)
// followed by conditional subtractions of Q<<k.  One thread per coefficient.
// ---------------------------------------------------------------------------
static constexpr int kCrtMaxLimbs = 36;  // also the row stride of the qhat / qsh tables

template <typename T>
__global__ void k_crt_lift(uint64_t *out, const T *d, const ModConst<T> *__restrict__ mc,
                           const uint64_t *__restrict__ qhat, const uint64_t *__restrict__ qsh, int logn, int nm, int L,
                           int Lacc, size_t ncoef) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncoef) return;
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
  uint64_t acc[kCrtMaxLimbs];
  for (int k = 0; k < Lacc; ++k) acc[k] = 0;
  for (int cm = 0; cm < nm; ++cm) {
    const ModConst<T> c = mc[cm];
    const T x = d[((b * nm + cm) << logn) + i];
    const uint64_t y = (uint64_t)mul_shoup<T>(x, c.yinv, c.yinv_sh, c.p);
    const uint64_t *qh = qhat + (size_t)cm * kCrtMaxLimbs;
    uint64_t carry = 0;
    for (int k = 0; k < Lacc; ++k) {  // acc += qhat[cm] * y
      const uint64_t lo = qh[k] * y, hi = __umul64hi(qh[k], y);
      uint64_t s = acc[k] + lo;
      uint64_t c1 = s < lo ? 1 : 0;
      s += carry;
      c1 += s < carry ? 1 : 0;
      acc[k] = s;
      carry = hi + c1;
    }
  }
  for (int sft = 5; sft >= 0; --sft) {  // acc < 32*Q: subtract Q<<5 .. Q<<0 when possible
    const uint64_t *qs = qsh + (size_t)sft * kCrtMaxLimbs;
    bool ge = true;
    for (int k = Lacc - 1; k >= 0; --k) {
      if (acc[k] != qs[k]) { ge = acc[k] > qs[k]; break; }
    }
    if (ge) {
      uint64_t borrow = 0;
      for (int k = 0; k < Lacc; ++k) {
        const uint64_t a = acc[k], q = qs[k];
        const uint64_t dd = a - q - borrow;
        borrow = (a < q || (a == q && borrow)) ? 1 : 0;
        acc[k] = dd;
      }
    }
  }
  uint64_t *o = out + gid * (size_t)L;
  for (int k = 0; k < L; ++k) o[k] = acc[k];
}

// 16- and 32-bit limbs with a moduli product below 2^64 (one lifted limb: every configuration the reference's tests use for
// these widths, e.g. (1024, 60-bit, uint32_t) and (128, 14-bit, uint16_t), tests/CMakeLists.txt:19-48): HBM-bound streaming
// kernels.  A thread owns FOUR consecutive coefficients: one vector load per modulus row, two 16-byte stores of lifted
// words; algorithmic bytes = nm n w + 8 n per polynomial.
template <typename T, int NM> struct alignas(4 * sizeof(T)) Quad { T v[4]; };

template <typename T, int NM>
__global__ void __launch_bounds__(256) k_crt_lift_small(uint64_t *__restrict__ out, const T *__restrict__ d,
                                                        const ModConst<T> *__restrict__ mc, const uint64_t *__restrict__ qhat,
                                                        uint64_t Q, int logn, size_t nquads) {
  typedef Quad<T, NM> Q4;
  uint64_t qh[NM];
  T yinv[NM], yinv_sh[NM], p[NM];
#pragma unroll
  for (int cm = 0; cm < NM; ++cm) {
    qh[cm] = qhat[(size_t)cm * kCrtMaxLimbs];
    yinv[cm] = mc[cm].yinv;
    yinv_sh[cm] = mc[cm].yinv_sh;
    p[cm] = mc[cm].p;
  }
  const size_t qlog = (size_t)logn - 2;  // quads per row
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < nquads; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g >> qlog, i4 = g & ((((size_t)1) << qlog) - 1);
    unsigned __int128 acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int cm = 0; cm < NM; ++cm) {
      const Q4 x = reinterpret_cast<const Q4 *>(d + ((b * NM + cm) << logn))[i4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += (unsigned __int128)qh[cm] * (uint64_t)mul_shoup<T>(x.v[k], yinv[cm], yinv_sh[cm], p[cm]);
    }
    ulonglong2 o[2];
    uint64_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned __int128 a = acc[k];          // < NM * Q: at most NM - 1 subtractions
#pragma unroll
      for (int t = 1; t < NM; ++t) a = a >= (unsigned __int128)Q ? a - Q : a;
      r[k] = (uint64_t)a;
    }
    o[0] = make_ulonglong2(r[0], r[1]);
    o[1] = make_ulonglong2(r[2], r[3]);
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(out + g * 4);
    dst[0] = o[0];
    dst[1] = o[1];
  }
}

// x mod p for p < 2^30 and any 64-bit x: Barrett with mu = floor(2^64 / p) (the quotient estimate is at most 2 short)
__device__ __forceinline__ uint32_t mod_small(uint64_t x, uint32_t p, uint64_t mu) {
  uint64_t r = x - __umul64hi(x, mu) * p;
  r = r >= p ? r - p : r;
  r = r >= p ? r - p : r;
  return (uint32_t)r;
}

template <typename T, int NM>
__global__ void __launch_bounds__(256) k_crt_project_small(T *__restrict__ d, const uint64_t *__restrict__ limbs,
                                                           const ModConst<T> *__restrict__ mc, int logn, int Lin, size_t nquads) {
  typedef Quad<T, NM> Q4;
  uint32_t p[NM], beta[NM];
  uint64_t mu[NM];
#pragma unroll
  for (int cm = 0; cm < NM; ++cm) {
    p[cm] = mc[cm].p;
    beta[cm] = mc[cm].beta;            // 2^64 mod p
    mu[cm] = ~0ull / p[cm];            // floor((2^64 - 1) / p) = floor(2^64 / p) for odd p
  }
  const size_t qlog = (size_t)logn - 2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < nquads; g += (size_t)gridDim.x * blockDim.x) {
    const size_t b = g >> qlog, i4 = g & ((((size_t)1) << qlog) - 1);
    const uint64_t *x = limbs + g * 4 * (size_t)Lin;
    uint32_t r[NM][4];
#pragma unroll
    for (int cm = 0; cm < NM; ++cm)
#pragma unroll
      for (int k = 0; k < 4; ++k) r[cm][k] = 0;
    for (int l = Lin - 1; l >= 0; --l) {   // Horner over the limbs, most significant first
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint64_t w = x[(size_t)k * Lin + l];
#pragma unroll
        for (int cm = 0; cm < NM; ++cm)
          r[cm][k] = mod_small((uint64_t)r[cm][k] * beta[cm] + mod_small(w, p[cm], mu[cm]), p[cm], mu[cm]);
      }
    }
#pragma unroll
    for (int cm = 0; cm < NM; ++cm) {
      Q4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o.v[k] = (T)r[cm][k];
      reinterpret_cast<Q4 *>(d + ((b * NM + cm) << logn))[i4] = o;
    }
  }
}

template <typename T>
static hipError_t launch_crt_small(const Shape &s, const DevTables &t, uint64_t *limbs_out, const T *d_in, T *d_out,
                                   const uint64_t *limbs_in, size_t L_in, size_t batch, hipStream_t st) {
  const size_t nquads = batch * (s.n >> 2);
  size_t blocks = (nquads + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
#define NFLHIP_CRT_SMALL(NM)                                                                                              \
  if (limbs_out) hipLaunchKernelGGL((k_crt_lift_small<T, NM>), dim3((unsigned)blocks), dim3(256), 0, st, limbs_out, d_in, mc, \
                                    t.qhat, s.crt_Q0, s.logn, nquads);                                                    \
  else hipLaunchKernelGGL((k_crt_project_small<T, NM>), dim3((unsigned)blocks), dim3(256), 0, st, d_out, limbs_in, mc,     \
                          s.logn, (int)L_in, nquads);
  switch (s.nm) {
    case 1: NFLHIP_CRT_SMALL(1) break;
    case 2: NFLHIP_CRT_SMALL(2) break;
    case 3: NFLHIP_CRT_SMALL(3) break;
    case 4: NFLHIP_CRT_SMALL(4) break;
    default: return hipErrorNotSupported;
  }
#undef NFLHIP_CRT_SMALL
  return hipGetLastError();
}
template <> hipError_t launch_crt_small<uint64_t>(const Shape &, const DevTables &, uint64_t *, const uint64_t *, uint64_t *,
                                                  const uint64_t *, size_t, size_t, hipStream_t) {
  return hipErrorNotSupported;
}

template <typename T>
hipError_t launch_crt_lift(const Shape &s, const DevTables &t, uint64_t *limbs, const T *d, size_t batch, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  if (sizeof(T) < 8 && s.crt_L == 1 && s.nm <= 4 && s.logn >= 2 && (((uintptr_t)d | (uintptr_t)limbs) & 15) == 0) {
    hipError_t e = launch_crt_small<T>(s, t, limbs, d, nullptr, nullptr, 0, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (std::is_same<T, uint64_t>::value) {
    hipError_t e = launch_crt_lift_fast_u64(s, t, limbs, (const uint64_t *)d, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if ((int)s.crt_Lacc > kCrtMaxLimbs || s.nm > 32) return hipErrorNotSupported;
  const size_t ncoef = batch * s.n;
  hipLaunchKernelGGL((k_crt_lift<T>), dim3((unsigned)((ncoef + 127) / 128)), dim3(128), 0, st, limbs, d,
                     (const ModConst<T> *)t.mc, t.qhat, t.qsh, s.logn, (int)s.nm, (int)s.crt_L, (int)s.crt_Lacc, ncoef);
  return hipGetLastError();
}

// The same lift for ANY number of moduli (the reference's GMP::poly2mpz has no cap, gmp.hpp:183-209): limb-serial, one
// thread per coefficient.  Limb k of the unreduced sum S = sum_cm (Q/p_cm) * y_cm is
//   sum_cm [ lo64(qhat[cm][k] * y_cm) + hi64(qhat[cm][k-1] * y_cm) ] + carry   (a 128-bit accumulator per limb),
// so nothing but the running carry survives from one limb to the next and no per-thread array is needed; y_cm is
// recomputed per limb.  S (< nm * Q, Lw = L + 2 limbs) goes to a limb-major scratch, then Q << k is subtracted
// conditionally for k = nsh-1 .. 0 (2^nsh > nm).  Throughput is not the point here: no caller of the reference goes
// beyond a few dozen moduli; this removes the cap.
template <typename T>
__global__ void k_crt_lift_wide(uint64_t *out, uint64_t *scr, const T *d, const ModConst<T> *__restrict__ mc,
                                const uint64_t *__restrict__ qhat, const uint64_t *__restrict__ qsh, int nsh, int logn,
                                int nm, int L, int Lw, size_t ncoef) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncoef) return;
  const size_t b = gid >> logn, i = gid & ((((size_t)1) << logn) - 1);
  const T *col = d + ((b * nm) << logn) + i;
  unsigned __int128 carry = 0;
  for (int k = 0; k < Lw; ++k) {
    unsigned __int128 acc = carry;
    for (int cm = 0; cm < nm; ++cm) {
      const ModConst<T> c = mc[cm];
      const uint64_t y = (uint64_t)mul_shoup<T>(col[(size_t)cm << logn], c.yinv, c.yinv_sh, c.p);
      const uint64_t *qh = qhat + (size_t)cm * Lw;
      acc += (unsigned __int128)(qh[k] * y);
      if (k > 0) acc += (unsigned __int128)__umul64hi(qh[k - 1], y);
    }
    scr[(size_t)k * ncoef + gid] = (uint64_t)acc;
    carry = acc >> 64;
  }
  for (int sft = nsh - 1; sft >= 0; --sft) {
    const uint64_t *qs = qsh + (size_t)sft * Lw;
    bool ge = true;
    for (int k = Lw - 1; k >= 0; --k) {
      const uint64_t a = scr[(size_t)k * ncoef + gid];
      if (a != qs[k]) { ge = a > qs[k]; break; }
    }
    if (ge) {
      uint64_t borrow = 0;
      for (int k = 0; k < Lw; ++k) {
        const uint64_t a = scr[(size_t)k * ncoef + gid], q = qs[k];
        scr[(size_t)k * ncoef + gid] = a - q - borrow;
        borrow = (a < q || (a == q && borrow)) ? 1 : 0;
      }
    }
  }
  uint64_t *o = out + gid * (size_t)L;
  for (int k = 0; k < L; ++k) o[k] = scr[(size_t)k * ncoef + gid];
}

template <typename T>
hipError_t launch_crt_lift_wide(const Shape &s, const DevTables &t, uint64_t *limbs, const T *d, size_t batch,
                                uint64_t *scratch, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  if (!t.qhat_w) return hipErrorNotSupported;
  const size_t ncoef = batch * s.n;
  hipLaunchKernelGGL((k_crt_lift_wide<T>), dim3((unsigned)((ncoef + 63) / 64)), dim3(64), 0, st, limbs, scratch, d,
                     (const ModConst<T> *)t.mc, t.qhat_w, t.qsh_w, t.crt_nsh, s.logn, (int)s.nm, (int)s.crt_L, t.crt_Lw,
                     ncoef);
  return hipGetLastError();
}

// CRT project (GMP::mpz2poly gmp.hpp:211-219): x(cm,i) = X_i mod p_cm by Horner
// over the 64-bit limbs with beta = 2^64 mod p.
template <typename T>
__global__ void k_crt_project(T *d, const uint64_t *limbs, const ModConst<T> *__restrict__ mc, int logn, int nm, int Lin,
                              size_t total) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const size_t i = gid & ((((size_t)1) << logn) - 1);
  const size_t rowi = gid >> logn;
  const size_t b = rowi / (size_t)nm;
  const int cm = (int)(rowi % (size_t)nm);
  const ModConst<T> c = mc[cm];
  const uint64_t p = c.p;
  const uint64_t *x = limbs + ((b << logn) + i) * (size_t)Lin;
  uint64_t r = 0;
  for (int k = Lin - 1; k >= 0; --k) {
    uint64_t l = x[k];
    if (sizeof(T) == 8) {
      l = csub<uint64_t>(l, 4 * p); l = csub<uint64_t>(l, 2 * p); l = csub<uint64_t>(l, p);
      // r < 4p: r*beta mod p lazily in [0,2p), plus l < p  -> < 3p
      r = mul_shoup_lazy<uint64_t>(r, (uint64_t)c.beta, (uint64_t)c.beta_sh, p) + l;
    } else {
      l %= p;
      r = ((r * (uint64_t)c.beta) % p + l);  // p < 2^30: no overflow
      r = r >= p ? r - p : r;
    }
  }
  if (sizeof(T) == 8) r = reduce4<uint64_t>(r, p);
  d[gid] = (T)r;
}

template <typename T>
hipError_t launch_crt_project(const Shape &s, const DevTables &t, T *d, const uint64_t *limbs, size_t L_in, size_t batch,
                              hipStream_t st) {
  if (batch == 0) return hipSuccess;
  if (std::is_same<T, uint64_t>::value) {
    hipError_t e = launch_crt_project_fast_u64(s, t, (uint64_t *)d, limbs, L_in, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  if (sizeof(T) < 8 && s.nm <= 4 && s.logn >= 2 && L_in <= 64 && (((uintptr_t)d | (uintptr_t)limbs) & 15) == 0) {
    hipError_t e = launch_crt_small<T>(s, t, nullptr, nullptr, d, limbs, L_in, batch, st);
    if (e != hipErrorNotSupported) return e;
  }
  const size_t total = batch * s.nm * s.n;
  hipLaunchKernelGGL((k_crt_project<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, limbs,
                     (const ModConst<T> *)t.mc, s.logn, (int)s.nm, (int)L_in, total);
  return hipGetLastError();
}

// permut<degree>::compute (permut.hpp:86-117) in place: word i of every row trades places with word bitrev(i)
// (disjoint transpositions, so no temporary row is needed)
template <typename T>
__global__ void k_bitrev_rows(T *d, int logn, size_t total) {
  const size_t mask = (((size_t)1) << logn) - 1;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const unsigned i = (unsigned)(g & mask);
    const unsigned r = __brev(i) >> (32 - logn);
    if (i < r) {
      T *row = d + (g & ~mask);
      const T x = row[i], y = row[r];
      row[i] = y;
      row[r] = x;
    }
  }
}
template <typename T> hipError_t launch_bitrev_rows(const Shape &s, T *d, size_t rows, hipStream_t st) {
  if (rows == 0) return hipSuccess;
  const size_t total = rows * s.n;
  const size_t blocks = stream_blocks(total);
  hipLaunchKernelGGL((k_bitrev_rows<T>), dim3((unsigned)blocks), dim3(256), 0, st, d, s.logn, total);
  return hipGetLastError();
}

// `count` copies of one polynomial (a key shared by every ciphertext of a resident batch): 16 bytes per lane
__global__ void k_broadcast(uint4 *dst, const uint4 *__restrict__ one, size_t vec_per_poly, size_t total) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x)
    dst[g] = one[g % vec_per_poly];
}
__global__ void k_broadcast_bytes(unsigned char *dst, const unsigned char *__restrict__ one, size_t bytes_per_poly,
                                  size_t total) {
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x)
    dst[g] = one[g % bytes_per_poly];
}
hipError_t launch_broadcast(void *dst, const void *one, size_t bytes_per_poly, size_t count, hipStream_t st) {
  if (count == 0 || bytes_per_poly == 0) return hipSuccess;
  const bool vec = (bytes_per_poly % 16 == 0) && (((uintptr_t)dst | (uintptr_t)one) % 16 == 0);
  const size_t total = vec ? bytes_per_poly / 16 * count : bytes_per_poly * count;
  const size_t blocks = stream_blocks(total);
  if (vec)
    hipLaunchKernelGGL(k_broadcast, dim3((unsigned)blocks), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)one,
                       bytes_per_poly / 16, total);
  else
    hipLaunchKernelGGL(k_broadcast_bytes, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)dst,
                       (const unsigned char *)one, bytes_per_poly, total);
  return hipGetLastError();
}

// ---- explicit instantiations ----
#define NFLHIP_INST(T)                                                                                              \
  template hipError_t launch_ntt_fwd<T>(const Shape &, const DevTables &, const T *, T *, size_t, hipStream_t);      \
  template hipError_t launch_ntt_inv<T>(const Shape &, const DevTables &, const T *, const T *, T *, size_t,         \
                                        hipStream_t);                                                                \
  template hipError_t launch_pointwise<T>(const Shape &, const DevTables &, int, T *, const T *, const T *,          \
                                          const T *, size_t, hipStream_t);                                           \
  template hipError_t launch_eval_expr<T>(const Shape &, const DevTables &, T *, const void *const *, int,           \
                                          const unsigned char *, int, size_t, hipStream_t, const unsigned *,         \
                                          unsigned);                                                                 \
  template hipError_t launch_check_range<T>(const Shape &, const DevTables &, const T *, size_t, int *, int, hipStream_t); \
  template hipError_t launch_any_cmp<T>(const Shape &, const DevTables &, const T *, const T *, size_t, int, int *,  \
                                        int, hipStream_t);                                                           \
  template hipError_t launch_fill_uniform<T>(const Shape &, const DevTables &, T *, size_t, size_t, uint64_t, int,   \
                                             hipStream_t);                                                           \
  template hipError_t launch_bitrev_rows<T>(const Shape &, T *, size_t, hipStream_t);                                \
  template hipError_t launch_crt_lift<T>(const Shape &, const DevTables &, uint64_t *, const T *, size_t,            \
                                         hipStream_t);                                                               \
  template hipError_t launch_crt_project<T>(const Shape &, const DevTables &, T *, const uint64_t *, size_t, size_t, \
                                            hipStream_t);                                                            \
  template hipError_t launch_crt_lift_wide<T>(const Shape &, const DevTables &, uint64_t *, const T *, size_t,       \
                                              uint64_t *, hipStream_t);
NFLHIP_INST(uint16_t)
NFLHIP_INST(uint32_t)
NFLHIP_INST(uint64_t)

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_generic() {}
hipError_t warm_generic(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_generic, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
