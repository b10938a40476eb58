// Probe (MI355X): how fast, and how fresh, are the ways one workgroup can observe a counter that another workgroup bumps
// with a plain device-scope atomic add -- from the SAME XCD and from another one.  Decides how the one-launch kernel's
// scheduler record should be polled.   hipcc --offload-arch=gfx950 -O2 -o l2_flag_probe l2_flag_probe.hip && ./l2_flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { kVecPlain, kVecSc0, kVecSc1, kScalarPlain, kScalarGlc, kAtomicRet, kVecNt, kVecSc0Nt, kVecInvSc0, kVecInvSc1, kKinds };
__device__ __forceinline__ unsigned probe(const unsigned *p, int kind) {
  unsigned v;
  switch (kind) {
    case kVecPlain: asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kVecSc0: asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kVecSc1: asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kScalarPlain: asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); break;
    case kScalarGlc: asm volatile("s_load_dword %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory"); break;
    case kVecNt: asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kVecSc0Nt: asm volatile("global_load_dword %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kVecInvSc0: asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    case kVecInvSc1: asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); break;
    default: v = __hip_atomic_fetch_add(const_cast<unsigned *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
  }
  return v;
}

// out[wg][kind][k] = {value seen, ticks of the probe}; the producer publishes its own counter+time in prod[]
__global__ void k(unsigned *flag, unsigned *out, unsigned *xcc, int probes, int bumps, int producer_wg) {
  const int wg = blockIdx.x;
  if (threadIdx.x == 0) xcc[wg] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
  if (threadIdx.x != 0) return;
  if (wg == producer_wg) {
    for (int i = 0; i < bumps; ++i) {
      __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(40);
    }
    return;
  }
  for (int kind = 0; kind < kKinds; ++kind)
    for (int i = 0; i < probes; ++i) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      const unsigned v = probe(flag, kind);
      const unsigned long long t1 = __builtin_readcyclecounter();
      unsigned *o = out + ((size_t)(wg * kKinds + kind) * probes + i) * 2;
      o[0] = v;
      o[1] = (unsigned)(t1 - t0);
      __builtin_amdgcn_s_sleep(20);
    }
}

int main() {
  const int wgs = 16, probes = 400, bumps = 60000;
  unsigned *flag, *out, *xcc;
  CHECK(hipMalloc(&flag, 256));
  CHECK(hipMalloc(&out, (size_t)wgs * kKinds * probes * 8));
  CHECK(hipMalloc(&xcc, wgs * 4));
  CHECK(hipMemset(flag, 0, 256));
  CHECK(hipMemset(out, 0, (size_t)wgs * kKinds * probes * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(64), 0, 0, flag, out, xcc, probes, bumps, 0);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h((size_t)wgs * kKinds * probes * 2), hx(wgs);
  CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hx.data(), xcc, wgs * 4, hipMemcpyDeviceToHost));
  printf("kernel %.3f ms (producer: %d bumps)\n", ms, bumps);
  const char *names[kKinds] = {"vector plain", "vector sc0", "vector sc1", "scalar plain", "scalar glc", "atomic +0 ret", "vector nt", "vector sc0 nt", "inv sc0 + plain", "inv sc1 + plain"};
  for (int wg : {8, 1, 9}) {
    printf("consumer wg %d on XCC %u (producer wg 0 on XCC %u)\n", wg, hx[wg], hx[0]);
    for (int kind = 0; kind < kKinds; ++kind) {
      unsigned long long sum = 0;
      unsigned mn = ~0u, distinct = 0, last = ~0u, first = 0, lastv = 0;
      for (int i = 0; i < probes; ++i) {
        const unsigned v = h[((size_t)(wg * kKinds + kind) * probes + i) * 2], t = h[((size_t)(wg * kKinds + kind) * probes + i) * 2 + 1];
        sum += t;
        if (t < mn) mn = t;
        if (v != last) ++distinct, last = v;
        if (i == 0) first = v;
        lastv = v;
      }
      printf("  %-14s avg %7.1f ticks  min %5u   values %u .. %u, %u distinct of %d probes\n", names[kind], (double)sum / probes, mn,
             first, lastv, distinct, probes);
    }
  }
  return 0;
}
