// tools/ubench.hip -- gfx950 instruction-rate and HBM micro-benchmarks that the
// kernel design in DESIGN.md is priced against (hard part 1:
// "62-bit modular arithmetic has no native wide multiply on CDNA").
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
//
// Prints one line per instruction: wave-instructions/s chip-wide and the
// implied issue cycles per wave64 instruction per SIMD (at the measured clock).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "../nfllib_amd/csrc/modarith64.h"   // the product kernels' own butterflies (ct_bfly<ARITH>)

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);    \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

constexpr int ITER = 4096;
constexpr int ACC = 8;

#define DEF_KERNEL32(NAME, ASM)                                                        \
  __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b) {                        \
    uint32_t r[ACC];                                                                   \
    for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x + k;                              \
    uint32_t x = a + threadIdx.x, y = b;                                               \
    for (int it = 0; it < ITER; ++it) {                                                \
      _Pragma("unroll") for (int k = 0; k < ACC; ++k) asm volatile(ASM : "+v"(r[k]) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                  \
    uint32_t s = 0;                                                                    \
    for (int k = 0; k < ACC; ++k) s ^= r[k];                                           \
    if (s == 0x12345678u) out[0] = s;                                                  \
  }

#define DEF_KERNEL64(NAME, ASM)                                                        \
  __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b) {                        \
    uint64_t r[ACC];                                                                   \
    for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x + k;                              \
    uint32_t x = a + threadIdx.x, y = b;                                               \
    uint64_t z = ((uint64_t)a << 32) | b;                                              \
    for (int it = 0; it < ITER; ++it) {                                                \
      _Pragma("unroll") for (int k = 0; k < ACC; ++k)                                  \
          asm volatile(ASM : "+v"(r[k]) : "v"(x), "v"(y), "v"(z) : "vcc");             \
    }                                                                                  \
    uint64_t s = 0;                                                                    \
    for (int k = 0; k < ACC; ++k) s ^= r[k];                                           \
    if (s == 0x12345678u) out[0] = (uint32_t)s;                                        \
  }

DEF_KERNEL32(k_add_u32, "v_add_u32 %0, %1, %0")
DEF_KERNEL32(k_sub_u32, "v_sub_u32 %0, %1, %0")
DEF_KERNEL32(k_and_b32, "v_and_b32 %0, %1, %0")
DEF_KERNEL32(k_xor_b32, "v_xor_b32 %0, %1, %0")
DEF_KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEF_KERNEL32(k_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")
DEF_KERNEL32(k_min_u32, "v_min_u32 %0, %1, %0")
DEF_KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %1, %0")
DEF_KERNEL32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %1, %0, vcc")
DEF_KERNEL32(k_sub_co_u32, "v_sub_co_u32 %0, vcc, %1, %0")
DEF_KERNEL32(k_cndmask_s, "v_cndmask_b32 %0, %1, %0, s[20:21]")
DEF_KERNEL32(k_cndmask_e32, "v_cndmask_b32_e32 %0, %1, %0, vcc")
DEF_KERNEL32(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %1, %0")
DEF_KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %1, 2, %0")
DEF_KERNEL32(k_and_or_b32, "v_and_or_b32 %0, %1, %2, %0")
DEF_KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 20")
DEF_KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0")
DEF_KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
DEF_KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
DEF_KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %0")
DEF_KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0")
DEF_KERNEL32(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL32(k_alignbit, "v_alignbit_b32 %0, %1, %0, 7")
DEF_KERNEL32(k_add3, "v_add3_u32 %0, %1, %2, %0")
DEF_KERNEL32(k_addco_pair, "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %0, vcc, %2, %0, vcc")
DEF_KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %3, 0, %0")
DEF_KERNEL64(k_fma_f64, "v_fma_f64 %0, %3, %3, %0")
DEF_KERNEL64(k_cmp_u64_only, "v_cmp_ge_u64 vcc, %0, %3")

// composite: compiler-generated 64-bit Shoup lazy multiply and Harvey butterfly
__global__ void k_shoup64(uint64_t *out, uint64_t w, uint64_t wp, uint64_t p) {
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x * 0x9E3779B97F4A7C15ull + k;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; ++k) {
      const uint64_t q = __umul64hi(r[k], wp);
      r[k] = r[k] * w - q * p;
    }
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void k_bfly64(uint64_t *out, uint64_t w, uint64_t wp, uint64_t p) {
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = (threadIdx.x * 0x9E3779B97F4A7C15ull + k) >> 2;
  const uint64_t p2 = 2 * p;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; k += 2) {
      uint64_t x = r[k], y = r[k + 1];
      x = x >= p2 ? x - p2 : x;
      const uint64_t q = __umul64hi(y, wp);
      const uint64_t m = y * w - q * p;
      r[k] = x + m;
      r[k + 1] = x - m + p2;
    }
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}
// ---- the alternatives for one lazy Cooley-Tukey butterfly on 62-bit primes p = 2^62 - delta, as hipcc compiles them
// (the generated assembly kernel is the ARITH 3 formulation hand-scheduled: 18 instructions, profiles/ r01_v6_asm_pmc.txt).
//   ARITH 0: Harvey ranges + Shoup with q*p as a full 64-bit multiply        (what the reference's algorithm maps to)
//   ARITH 2: two-bit fold + Shoup, q*p = (q<<62) - q*delta, exact quotient
//   ARITH 3: the same with the one-off quotient                              (the shipped formulation)
//   SOLINAS: no Shoup companion at all: 128-bit product y*w, then 2^62 = delta (mod p) folded three times
template <int ARITH> __global__ void k_bfly_arith(uint64_t *out, uint64_t w, uint64_t wp, uint64_t p) {
  using namespace nflhip;
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = (threadIdx.x * 0x9E3779B97F4A7C15ull + k) >> 2;
  Mod m;
  m.p = p; m.p2 = 2 * p; m.p3 = 3 * p; m.d = (uint32_t)((1ull << 62) - p);
  Tw64 tw;
  tw.w = w; tw.wp = wp;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; k += 2) ct_bfly<ARITH>(r[k], r[k + 1], tw, m);
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void k_bfly_solinas(uint64_t *out, uint64_t w, uint64_t wp, uint64_t p) {
  typedef unsigned __int128 u128;
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = (threadIdx.x * 0x9E3779B97F4A7C15ull + k) >> 2;
  const uint64_t M62 = (1ull << 62) - 1, delta = (1ull << 62) - p, p3 = 3 * p;
  (void)wp;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; k += 2) {
      const uint64_t x = r[k], y = r[k + 1];
      const uint64_t U = (x & M62) + (x >> 62) * delta;                   // fold2(x) < p + 4 delta
      const u128 P = (u128)y * w;                                          // < 2^126
      const u128 t = (u128)(uint64_t)(P >> 62) * delta + ((uint64_t)P & M62);   // < 2^96 + 2^62
      const u128 t2 = (u128)(uint64_t)(t >> 62) * delta + ((uint64_t)t & M62);  // < 2^67 + 2^62
      const uint64_t mm = ((uint64_t)t2 & M62) + (uint64_t)(t2 >> 62) * delta;  // < 2^62 + 2^38: lazily reduced y*w
      r[k] = U + mm;
      r[k + 1] = U + p3 - mm;
    }
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void k_umul64hi(uint64_t *out, uint64_t w) {
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x * 0x9E3779B97F4A7C15ull + k;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; ++k) r[k] = __umul64hi(r[k], w) + k;
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void k_mullo64(uint64_t *out, uint64_t w) {
  uint64_t r[ACC];
  for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x * 0x9E3779B97F4A7C15ull + k;
  for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
    for (int k = 0; k < ACC; ++k) r[k] = r[k] * w + k;
  }
  uint64_t s = 0;
  for (int k = 0; k < ACC; ++k) s ^= r[k];
  if (s == 0x12345678u) out[0] = s;
}

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_copy8(const uint2 *__restrict__ src, uint2 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_clock(long long *out) {
  long long t0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - t0 < 10000000) {}
  out[0] = clock64() - c0;
  out[1] = wall_clock64() - t0;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / reps;
}

int main() {
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clockRate=%d kHz  LDS/block=%zu  L2=%d\n", prop.name, cus, prop.clockRate,
         prop.sharedMemPerBlock, prop.l2CacheSize);
  uint32_t *d32;
  uint64_t *d64;
  CK(hipMalloc(&d32, 1024));
  CK(hipMalloc(&d64, 1024));
  const int blocks = cus * 8, threads = 256;  // 8 waves/SIMD
  const double waves = (double)blocks * threads / 64.0;
  const double simds = cus * 4.0;
  const double clk = 2.4e9;
#define RUN32(K, N_PER)                                                                              \
  {                                                                                                  \
    double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, d32, 3u, 5u); }, 5); \
    double inst = waves * ITER * ACC * (N_PER);                                                      \
    printf("%-18s %8.3f ms  %7.2f Gwaveinst/s  %6.2f cyc/inst/SIMD@2.4GHz\n", #K, ms, inst / ms / 1e6, \
           ms * 1e-3 * clk * simds / inst);                                                          \
  }
  RUN32(k_add_u32, 1)
  RUN32(k_sub_u32, 1)
  RUN32(k_and_b32, 1)
  RUN32(k_xor_b32, 1)
  RUN32(k_lshlrev_b32, 1)
  RUN32(k_ashrrev_i32, 1)
  RUN32(k_min_u32, 1)
  RUN32(k_mov_b32, 1)
  RUN32(k_fma_f32, 1)
  RUN32(k_add_co_u32, 1)
  RUN32(k_addc_co_u32, 1)
  RUN32(k_sub_co_u32, 1)
  RUN32(k_cndmask_s, 1)
  RUN32(k_cndmask_e32, 1)
  RUN32(k_cmp_lt_u32, 1)
  RUN32(k_lshl_add_u32, 1)
  RUN32(k_and_or_b32, 1)
  RUN32(k_bfe_u32, 1)
  RUN32(k_mad_u32_u16, 1)
  RUN32(k_pk_add_u16, 1)
  RUN32(k_mul_lo_u32, 1)
  RUN32(k_mul_hi_u32, 1)
  RUN32(k_mul_u32_u24, 1)
  RUN32(k_mad_u32_u24, 1)
  RUN32(k_mul_hi_u32_u24, 1)
  RUN32(k_cndmask, 1)
  RUN32(k_alignbit, 1)
  RUN32(k_add3, 1)
  RUN32(k_addco_pair, 2)
  RUN32(k_mad_u64_u32, 1)
  RUN32(k_lshl_add_u64, 1)
  RUN32(k_fma_f64, 1)
  RUN32(k_cmp_u64_only, 1)
#define RUN64(K, ...)                                                                                \
  {                                                                                                  \
    double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, d64, __VA_ARGS__); }, 5); \
    double ops = waves * (ITER / 4) * ACC;                                                           \
    printf("%-18s %8.3f ms  %7.2f Gwaveop/s   %6.2f cyc/op/SIMD@2.4GHz  (%.2f T lane-ops/s)\n", #K, ms, ops / ms / 1e6, \
           ms * 1e-3 * clk * simds / ops, ops * 64 / ms / 1e9);                                      \
  }
  const uint64_t p = 4611686018326724609ull, w = 2262382610096409597ull;
  const uint64_t wp = (uint64_t)((((unsigned __int128)w) << 64) / p);
  RUN64(k_shoup64, w, wp, p)
  RUN64(k_umul64hi, w)
  RUN64(k_mullo64, w)
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_bfly64, dim3(blocks), dim3(threads), 0, 0, d64, w, wp, p); }, 5);
    double ops = waves * (ITER / 4) * (ACC / 2);
    printf("%-18s %8.3f ms  %7.2f Gwavebfly/s %6.2f cyc/bfly/SIMD@2.4GHz (%.2f T bfly/s)\n", "k_bfly64", ms,
           ops / ms / 1e6, ms * 1e-3 * clk * simds / ops, ops * 64 / ms / 1e9);
  }
  {
    auto line = [&](const char *name, double ms) {
      double ops = waves * (ITER / 4) * (ACC / 2);
      printf("%-18s %8.3f ms  %7.2f Gwavebfly/s %6.2f cyc/bfly/SIMD@2.4GHz (%.2f T bfly/s)\n", name, ms, ops / ms / 1e6,
             ms * 1e-3 * clk * simds / ops, ops * 64 / ms / 1e9);
    };
    line("ct_harvey_shoup", time_ms([&] { hipLaunchKernelGGL(k_bfly_arith<0>, dim3(blocks), dim3(threads), 0, 0, d64, w, wp, p); }, 5));
    line("ct_fold2_exactq", time_ms([&] { hipLaunchKernelGGL(k_bfly_arith<2>, dim3(blocks), dim3(threads), 0, 0, d64, w, wp, p); }, 5));
    line("ct_fold2_oneoffq", time_ms([&] { hipLaunchKernelGGL(k_bfly_arith<3>, dim3(blocks), dim3(threads), 0, 0, d64, w, wp, p); }, 5));
    line("ct_solinas_3fold", time_ms([&] { hipLaunchKernelGGL(k_bfly_solinas, dim3(blocks), dim3(threads), 0, 0, d64, w, wp, p); }, 5));
  }
  // clock
  long long *dclk;
  CK(hipMalloc(&dclk, 16));
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, dclk);
  long long hclk[2];
  CK(hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost));
  printf("clock64/wall_clock64 ratio: %.4f (wall clock is 100 MHz => shader clock %.1f MHz)\n",
         (double)hclk[0] / hclk[1], 100.0 * hclk[0] / hclk[1]);
  // HBM copy
  const size_t bytes = (size_t)2 << 30;
  void *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 2, bytes));
  for (int bl : {1024, 2048, 4096, 8192}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(bl), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); }, 5);
    printf("copy16 %5d blocks: %.3f ms  %.2f TB/s (read+write)\n", bl, ms, 2.0 * bytes / ms / 1e9);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_copy8, dim3(4096), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, bytes / 8); }, 5);
    printf("copy8   4096 blocks: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
  }
  {
    double ms = time_ms([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
    printf("hipMemcpy D2D: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
  }
  return 0;
}
