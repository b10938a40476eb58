// tools/ubench_issue.hip -- the VALU issue model of gfx950 that the product kernels are priced against, measured as a
// GRID instead of one point (round-2 verdict: "light opcodes issue at 2.3-2.6 cycles isolated, ~4.2 in situ"):
//   stream (tools/gen_ubench_issue.py: single opcodes at two dependency distances, light:heavy mixes, the 30- and 62-bit
//   butterflies of the generated kernels, the same butterflies as long straight-line code)  x  waves per SIMD 1..8.
// Every stream is ONE inline-assembly block on fixed registers (hipcc pads separate asm statements with s_nop, which
// made round 2's isolated figures a two-instruction measurement).  Waves per SIMD are FORCED: every 256-thread block
// (one wave per SIMD) asks for 160 KiB / W of LDS, so exactly W blocks fit a CU and all 256 W blocks are resident at once.
// Two clocks bracket every loop: s_memtime (shader clock) and s_memrealtime (100 MHz), so cycles are cycles that elapsed.
//   per-SIMD cycles per unit = mean over waves of (elapsed s_memtime cycles) / (units per wave * W)
//   the same from the wall clock of the whole launch (HIP events) at the measured MHz -- they must agree
//
//   python tools/gen_ubench_issue.py && hipcc --offload-arch=gfx950 -O3 -o build/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Clk { long long cyc, ref; };

#include "ubench_issue_gen.inc"

__global__ void k_clock(Clk *out) {
  const long long t0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - t0 < 10000000) {}
  out[0] = Clk{clock64() - c0, wall_clock64() - t0};
}

static int g_cus = 256;
static uint32_t *g_d32;
static Clk *g_clk;

struct Result { double ms, per_wave, per_time, mhz; };

static Result run(const KernelRow &k, int W, int iters) {
  const int blocks = g_cus * W;
  size_t lds = (size_t)(163840 / W) & ~(size_t)127;
  if (W == 8) lds = 20480;
  (void)hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t p32 = 1073479681u, w32 = 123456789u, wp32 = (uint32_t)(((uint64_t)w32 << 32) / p32);
  const uint64_t p = 4611686018326724609ull, w = 2262382610096409597ull;
  const uint64_t wsh = (uint64_t)((((unsigned __int128)w) << 64) / p);
  const uint32_t delta = (uint32_t)((1ull << 62) - p);
  auto launch = [&] {
    if (!strcmp(k.kind, "b64"))
      hipLaunchKernelGGL((void (*)(uint32_t *, int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint64_t, uint32_t, Clk *))k.fn,
                         dim3(blocks), dim3(256), lds, 0, g_d32, iters, delta, 0x3fffffffu, (uint32_t)w, (uint32_t)(w >> 32), (uint32_t)wsh,
                         (uint32_t)(wsh >> 32), 3 * p, 0xC0000000u, g_clk);
    else if (!strcmp(k.kind, "b32"))
      hipLaunchKernelGGL((void (*)(uint32_t *, int, uint32_t, uint32_t, uint32_t, uint32_t, Clk *))k.fn, dim3(blocks), dim3(256), lds, 0, g_d32,
                         iters, 2 * p32, 0u - p32, w32, wp32, g_clk);
    else
      hipLaunchKernelGGL((void (*)(uint32_t *, int, uint32_t, Clk *))k.fn, dim3(blocks), dim3(256), lds, 0, g_d32, iters, 5u, g_clk);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  const int reps = 3;
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<Clk> h((size_t)blocks * 4);
  (void)hipMemcpy(h.data(), g_clk, sizeof(Clk) * h.size(), hipMemcpyDeviceToHost);
  double cyc = 0, ref = 0;
  for (auto &c : h) { cyc += (double)c.cyc; ref += (double)c.ref; }
  Result r;
  r.ms = ms;
  r.mhz = 100.0 * cyc / ref;
  const double units_per_wave = (double)iters * k.units;
  r.per_wave = (cyc / h.size()) / (units_per_wave * W);
  r.per_time = ms * 1e-3 * (r.mhz * 1e6) * (g_cus * 4.0) / (units_per_wave * blocks * 4.0);
  return r;
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  g_cus = prop.multiProcessorCount;
  (void)hipMalloc(&g_d32, 1024);
  (void)hipMalloc(&g_clk, sizeof(Clk) * g_cus * 8 * 4);
  printf("device: %s  CUs=%d  clockRate=%d kHz\n", prop.name, g_cus, prop.clockRate);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(1), 0, 0, g_clk);
  Clk idle;
  (void)hipMemcpy(&idle, g_clk, sizeof(Clk), hipMemcpyDeviceToHost);
  printf("idle probe: s_memtime counts at %.1f MHz against the 100 MHz s_memrealtime\n", 100.0 * idle.cyc / idle.ref);
  printf("\nper-SIMD cycles per unit (unit = one instruction for op_* / mix_*, one BUTTERFLY for bfly*); every cell:\n"
         "  from the waves' own s_memtime spans / from the launch's wall clock at the measured MHz (MHz)\n");
  const int Ws[] = {1, 2, 3, 4, 6, 8};
  printf("%-22s %5s", "stream", "instr");
  for (int W : Ws) printf(" | %d wave%s/SIMD        ", W, W > 1 ? "s" : " ");
  printf("\n");
  // --sustain STREAM W SECONDS: the same launch back to back for SECONDS, the in-kernel clock printed every half second
  // (sample rocm-smi -P -c beside it): does the stream hold its clock once the power controller has had time to react?
  if (argc > 4 && !strcmp(argv[1], "--sustain")) {
    const int W = atoi(argv[3]);
    const double seconds = atof(argv[4]);
    for (const KernelRow &k : kRows) {
      if (strcmp(k.name, argv[2])) continue;
      const int iters = std::max(4, 400000 / k.ninstr);
      double elapsed = 0, next = 0.5;
      while (elapsed < seconds) {
        Result r = run(k, W, iters);
        elapsed += 4 * r.ms * 1e-3;
        if (elapsed >= next) {
          printf("sustain %s W=%d t=%.1f s: %.2f cycles per unit per SIMD (wall clock), %.0f MHz, %.3f ms per launch\n", k.name, W, elapsed,
                 r.per_time, r.mhz, r.ms);
          fflush(stdout);
          next += 0.5;
        }
      }
      return 0;
    }
    printf("no stream named %s\n", argv[2]);
    return 1;
  }
  const std::string only = argc > 1 ? argv[1] : "";
  for (const KernelRow &k : kRows) {
    if (!only.empty() && std::string(k.name).find(only) == std::string::npos) continue;
    printf("%-22s %5d", k.name, k.ninstr);
    const int iters = std::max(4, 400000 / k.ninstr);
    for (int W : Ws) {
      Result r = run(k, W, iters);
      printf(" | %6.2f %6.2f (%4.0f)", r.per_wave, r.per_time, r.mhz);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
