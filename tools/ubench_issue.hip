// tools/ubench_issue.hip -- the VALU issue model of gfx950 that the product kernels are priced against, measured as a
// GRID instead of one point (round-2 verdict, "light opcodes issue at 2.3-2.6 cycles isolated, ~4.2 in situ"):
//   opcode x encoding  x  waves per SIMD (1..8)  x  dependency distance (1..8 independent accumulators)
//   + light:heavy interleave ratios, + the 7-instruction 30-bit butterfly and the 18-instruction 62-bit butterfly of
//     the generated kernels as straight-line assembly, + the EFFECTIVE SHADER CLOCK of every run:
// every kernel brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz), so "cycles per instruction"
// is in cycles that actually elapsed, not at an assumed 2.4 GHz.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_issue tools/ubench_issue.hip && ./ubench_issue > profiles/r03_ubench_issue.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int ITER = 2048;

struct Clk { long long cyc, ref; };

// ACC independent accumulators: instruction k of an iteration depends on instruction k of the previous iteration, i.e.
// the dependency distance is ACC instructions.  NPER = instructions per accumulator and iteration in ASM.
#define DEF_GRID(NAME, DECL, ASM, OUTC, INC)                                                        \
  template <int ACC> __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b, Clk *clk) {       \
    DECL r[ACC];                                                                                    \
    for (int k = 0; k < ACC; ++k) r[k] = threadIdx.x + k;                                           \
    uint32_t x = a + threadIdx.x, y = b;                                                            \
    uint64_t z = ((uint64_t)a << 32) | b;                                                           \
    (void)z;                                                                                        \
    const long long c0 = clock64(), w0 = wall_clock64();                                            \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int k = 0; k < ACC; ++k) asm volatile(ASM : OUTC(r[k]) : INC : "vcc"); \
    }                                                                                               \
    const long long c1 = clock64(), w1 = wall_clock64();                                            \
    DECL s = 0;                                                                                     \
    for (int k = 0; k < ACC; ++k) s ^= r[k];                                                        \
    if (s == 0x12345678u) out[0] = (uint32_t)s;                                                     \
    if (threadIdx.x == 0) clk[blockIdx.x] = Clk{c1 - c0, w1 - w0};                                  \
  }
#define IN32 "v"(x), "v"(y), "s"(b)
#define IN64 "v"(x), "v"(y), "v"(z), "s"(b)
#define RW "+v"

// light candidates (VOP2, no carry)
DEF_GRID(g_add_e32, uint32_t, "v_add_u32_e32 %0, %1, %0", RW, IN32)
DEF_GRID(g_add_e64, uint32_t, "v_add_u32_e64 %0, %1, %0", RW, IN32)
DEF_GRID(g_add_sgpr, uint32_t, "v_add_u32_e32 %0, %3, %0", RW, IN32)
DEF_GRID(g_sub_e32, uint32_t, "v_sub_u32_e32 %0, %1, %0", RW, IN32)
DEF_GRID(g_and_e32, uint32_t, "v_and_b32_e32 %0, %1, %0", RW, IN32)
DEF_GRID(g_mov_e32, uint32_t, "v_mov_b32_e32 %0, %1", RW, IN32)
DEF_GRID(g_lshl_e32, uint32_t, "v_lshlrev_b32_e32 %0, 3, %0", RW, IN32)
DEF_GRID(g_lshr_e32, uint32_t, "v_lshrrev_b32_e32 %0, 3, %0", RW, IN32)
DEF_GRID(g_ashr_e32, uint32_t, "v_ashrrev_i32_e32 %0, 3, %0", RW, IN32)
DEF_GRID(g_or_e32, uint32_t, "v_or_b32_e32 %0, %1, %0", RW, IN32)
DEF_GRID(g_max_e32, uint32_t, "v_max_u32_e32 %0, %1, %0", RW, IN32)
// heavy
DEF_GRID(g_min_e32, uint32_t, "v_min_u32_e32 %0, %1, %0", RW, IN32)
DEF_GRID(g_mulhi, uint32_t, "v_mul_hi_u32 %0, %1, %0", RW, IN32)
DEF_GRID(g_lshl_add, uint32_t, "v_lshl_add_u32 %0, %0, 1, %1", RW, IN32)
DEF_GRID(g_addco, uint32_t, "v_add_co_u32_e32 %0, vcc, %1, %0", RW, IN32)
DEF_GRID(g_mad64, uint64_t, "v_mad_u64_u32 %0, vcc, %1, %2, %0", RW, IN64)
DEF_GRID(g_mad64_sgpr, uint64_t, "v_mad_u64_u32 %0, vcc, %1, %4, %0", RW, IN64)
DEF_GRID(g_lshl_add64, uint64_t, "v_lshl_add_u64 %0, %3, 0, %0", RW, IN64)
// interleaves: H = v_mad_u64_u32 on a 64-bit accumulator, L = v_add_u32 / v_and / v_sub on a 32-bit accumulator of its own
// (two chains per slot; AMDGPU inline assembly cannot name the halves of a compiler-allocated pair)
#define DEF_GRID2(NAME, ASM)                                                                        \
  template <int ACC> __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b, Clk *clk) {       \
    uint64_t r[ACC];                                                                                \
    uint32_t l[ACC];                                                                                \
    for (int k = 0; k < ACC; ++k) { r[k] = threadIdx.x + k; l[k] = 7 * threadIdx.x + k; }           \
    uint32_t x = a + threadIdx.x, y = b;                                                            \
    const long long c0 = clock64(), w0 = wall_clock64();                                            \
    for (int it = 0; it < ITER; ++it) {                                                             \
      _Pragma("unroll") for (int k = 0; k < ACC; ++k) asm volatile(ASM : "+v"(r[k]), "+v"(l[k]) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                               \
    const long long c1 = clock64(), w1 = wall_clock64();                                            \
    uint64_t s = 0;                                                                                 \
    for (int k = 0; k < ACC; ++k) s ^= r[k] ^ l[k];                                                 \
    if (s == 0x12345678u) out[0] = (uint32_t)s;                                                     \
    if (threadIdx.x == 0) clk[blockIdx.x] = Clk{c1 - c0, w1 - w0};                                  \
  }
DEF_GRID2(g_mix_h1l1, "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32_e32 %1, %2, %1")
DEF_GRID2(g_mix_h1l2, "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32_e32 %1, %2, %1\n v_and_b32_e32 %1, %3, %1")
DEF_GRID2(g_mix_h2l1, "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %3, %2, %0\n v_add_u32_e32 %1, %2, %1")
DEF_GRID2(g_mix_h1l3, "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32_e32 %1, %2, %1\n v_and_b32_e32 %1, %3, %1\n v_sub_u32_e32 %1, %3, %1")

// the butterflies of the generated kernels as straight-line assembly on fixed registers (tools/gen_ubench_issue.py)
#include "ubench_issue_bfly.inc"

// an IDLE probe of the two clocks (one lane): what s_memtime counts against the 100 MHz reference with nothing else running
__global__ void k_clock(Clk *out) {
  const long long t0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - t0 < 10000000) {}
  out[0] = Clk{clock64() - c0, wall_clock64() - t0};
}

static int g_cus = 256;
static uint32_t *g_d32;
static Clk *g_clk;

struct Result { double ms, cyc_per_inst_ref24, cyc_per_inst_true, mhz; };

template <typename K, typename... Args> static Result run(K kernel, int waves_per_simd, double inst_per_wave, Args... args) {
  const int blocks = g_cus * waves_per_simd;  // 256 threads = one wave on each of the CU's four SIMDs
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, g_d32, args..., g_clk);
  hipDeviceSynchronize();
  const int reps = 3;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, g_d32, args..., g_clk);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  std::vector<Clk> h(blocks);
  hipMemcpy(h.data(), g_clk, sizeof(Clk) * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, ref = 0;
  for (auto &c : h) { cyc += (double)c.cyc; ref += (double)c.ref; }
  Result r;
  r.ms = ms;
  r.mhz = 100.0 * cyc / ref;  // s_memrealtime counts at 100 MHz
  // per SIMD: waves_per_simd waves issue inst_per_wave instructions each during the loop's (mean) s_memtime span
  r.cyc_per_inst_true = (cyc / blocks) / (inst_per_wave * waves_per_simd);
  r.cyc_per_inst_ref24 = (ref / blocks) * 24.0 / (inst_per_wave * waves_per_simd);
  return r;
}

#define GRID_ROW(NAME, NPER)                                                                                  \
  {                                                                                                           \
    printf("%-14s", #NAME + 2);                                                                               \
    for (int w : {1, 2, 4, 8}) {                                                                              \
      Result r1 = run(NAME<1>, w, (double)ITER * 1 * (NPER), 3u, 5u);                                         \
      Result r2 = run(NAME<2>, w, (double)ITER * 2 * (NPER), 3u, 5u);                                         \
      Result r4 = run(NAME<4>, w, (double)ITER * 4 * (NPER), 3u, 5u);                                         \
      Result r8 = run(NAME<8>, w, (double)ITER * 8 * (NPER), 3u, 5u);                                         \
      printf(" | %5.2f %5.2f %5.2f %5.2f (%4.0f)", r1.cyc_per_inst_true, r2.cyc_per_inst_true, r4.cyc_per_inst_true, \
             r8.cyc_per_inst_true, r8.mhz);                                                                   \
    }                                                                                                         \
    printf("\n");                                                                                             \
    fflush(stdout);                                                                                           \
  }

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  g_cus = prop.multiProcessorCount;
  hipMalloc(&g_d32, 1024);
  hipMalloc(&g_clk, sizeof(Clk) * g_cus * 8);
  printf("device: %s  CUs=%d  clockRate=%d kHz\n", prop.name, g_cus, prop.clockRate);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, g_clk);
  Clk idle;
  hipMemcpy(&idle, g_clk, sizeof(Clk), hipMemcpyDeviceToHost);
  printf("idle probe: s_memtime / s_memrealtime = %.4f  (s_memrealtime = 100 MHz => s_memtime counts at %.1f MHz)\n",
         (double)idle.cyc / idle.ref, 100.0 * idle.cyc / idle.ref);
  printf("\ncycles per wave64 instruction per SIMD, in s_memtime cycles that elapsed inside the kernel (NOT at an assumed clock).\n"
         "columns: waves per SIMD 1 | 2 | 4 | 8; inside a column: dependency distance 1 2 4 8 instructions; (MHz) = s_memtime rate\n"
         "against the 100 MHz reference during the 8-accumulator run\n");
  printf("%-14s | %-30s | %-30s | %-30s | %-30s\n", "opcode", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
  GRID_ROW(g_add_e32, 1)
  GRID_ROW(g_add_e64, 1)
  GRID_ROW(g_add_sgpr, 1)
  GRID_ROW(g_sub_e32, 1)
  GRID_ROW(g_and_e32, 1)
  GRID_ROW(g_or_e32, 1)
  GRID_ROW(g_mov_e32, 1)
  GRID_ROW(g_lshl_e32, 1)
  GRID_ROW(g_lshr_e32, 1)
  GRID_ROW(g_ashr_e32, 1)
  GRID_ROW(g_max_e32, 1)
  GRID_ROW(g_min_e32, 1)
  GRID_ROW(g_mulhi, 1)
  GRID_ROW(g_lshl_add, 1)
  GRID_ROW(g_addco, 1)
  GRID_ROW(g_mad64, 1)
  GRID_ROW(g_mad64_sgpr, 1)
  GRID_ROW(g_lshl_add64, 1)
  printf("\ninterleaves (H = v_mad_u64_u32, L = add / and / sub on a half of the same pair): cycles per INSTRUCTION\n");
  GRID_ROW(g_mix_h1l1, 2)
  GRID_ROW(g_mix_h1l2, 3)
  GRID_ROW(g_mix_h2l1, 3)
  GRID_ROW(g_mix_h1l3, 4)
  printf("\nthe generated kernels' butterflies as straight-line assembly on fixed registers: cycles per BUTTERFLY\n"
         "(bfly32 = the 7-instruction 30-bit Cooley-Tukey butterfly of gen_row1024_u32_asm.py, ideal 5 x 4 + 2 x 2 = 24 cycles if the two\n"
         " plain subtractions issue at 2 cycles; bfly64 = the 18-instruction 62-bit one of gen_polymul_asm.py; 'i' = two butterflies\n"
         " interleaved instruction by instruction as the kernels do; columns: waves per SIMD 1 | 2 | 3 | 4 | 8; (MHz) of the last entry)\n");
  {
    const uint32_t p32 = 1073479681u, w32 = 123456789u, wp32 = (uint32_t)(((uint64_t)w32 << 32) / p32);
    const uint64_t p = 4611686018326724609ull, w = 2262382610096409597ull;
    const uint64_t wsh = (uint64_t)((((unsigned __int128)w) << 64) / p);
    const uint32_t delta = (uint32_t)((1ull << 62) - p);
#define BROW32(K, ACC)                                                                                           \
    {                                                                                                            \
      printf("%-14s", #K + 2);                                                                                   \
      for (int wv : {1, 2, 3, 4, 8}) {                                                                           \
        Result r = run(K, wv, (double)(ITER / 4) * (ACC), 2 * p32, 0u - p32, w32, wp32);                         \
        printf(" | %6.1f (%4.0f)", r.cyc_per_inst_true, r.mhz);                                                  \
      }                                                                                                          \
      printf("\n");                                                                                              \
    }
#define BROW64(K, ACC)                                                                                           \
    {                                                                                                            \
      printf("%-14s", #K + 2);                                                                                   \
      for (int wv : {1, 2, 3, 4, 8}) {                                                                           \
        Result r = run(K, wv, (double)(ITER / 4) * (ACC), delta, 0x3fffffffu, (uint32_t)w, (uint32_t)(w >> 32), \
                       (uint32_t)wsh, (uint32_t)(wsh >> 32), 3 * p, 0xC0000000u);                                \
        printf(" | %6.1f (%4.0f)", r.cyc_per_inst_true, r.mhz);                                                  \
      }                                                                                                          \
      printf("\n");                                                                                              \
    }
    BROW32(g_bfly32_1, 1) BROW32(g_bfly32_2, 2) BROW32(g_bfly32i_2, 2) BROW32(g_bfly32_4, 4) BROW32(g_bfly32i_4, 4) BROW32(g_bfly32_8, 8) BROW32(g_bfly32i_8, 8)
    BROW64(g_bfly64_1, 1) BROW64(g_bfly64_2, 2) BROW64(g_bfly64i_2, 2) BROW64(g_bfly64_4, 4) BROW64(g_bfly64i_4, 4)
  }
  return 0;
}
