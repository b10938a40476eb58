// mall_probe.hip -- does the 256 MiB Infinity Cache (MALL) serve write -> read and read -> read reuse, and at what rate?
// (round 5, VERDICT item 3: "measure what fraction of the 3.05x is HBM and what is MALL".)  rocprofv3's FETCH_SIZE / WRITE_SIZE
// count fabric requests, MALL hits included (MI355X_MICROARCH.md), so the only way to see the MALL is the clock: stream a
// footprint of S bytes repeatedly and watch the rate fall when S leaves the cache.
//   build: hipcc --offload-arch=gfx950 -O3 -o build/mall_probe tools/probes/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_write(uint4 *p, size_t n, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}
__global__ void k_read(const uint4 *p, size_t n, unsigned *sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[i];
    acc ^= v.x + v.y + v.z + v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_copy(uint4 *d, const uint4 *s, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

int main() {
  const size_t maxS = (size_t)2 << 30;
  uint4 *a, *b;
  unsigned *sink;
  CHK(hipMalloc(&a, maxS));
  CHK(hipMalloc(&b, maxS));
  CHK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int grid = 256 * 8, blk = 256, reps = 12;
  printf("# footprint S: [write S, read S] x %d | [read S] x %d | [copy S -> S'] x %d ; TB/s of bytes touched (copy: read + write)\n", reps, reps, reps);
  for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048}) {
    const size_t S = mb << 20, n = S / 16;
    float ms;
    // write then read
    k_write<<<grid, blk>>>(a, n, 1); k_read<<<grid, blk>>>(a, n, sink);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) { k_write<<<grid, blk>>>(a, n, r); k_read<<<grid, blk>>>(a, n, sink); }
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wr = 2.0 * S * reps / (ms * 1e-3) / 1e12;
    // read only
    k_read<<<grid, blk>>>(a, n, sink);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_read<<<grid, blk>>>(a, n, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double rd = 1.0 * S * reps / (ms * 1e-3) / 1e12;
    // write only
    k_write<<<grid, blk>>>(a, n, 3);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_write<<<grid, blk>>>(a, n, r);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wo = 1.0 * S * reps / (ms * 1e-3) / 1e12;
    // copy (half footprint each so that the total touched is S)
    const size_t h = n / 2;
    k_copy<<<grid, blk>>>(b, a, h);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_copy<<<grid, blk>>>(b, a, h);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double cp = 1.0 * S * reps / (ms * 1e-3) / 1e12;
    // copy a -> b of S each (footprint 2S)
    k_copy<<<grid, blk>>>(b, a, n);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_copy<<<grid, blk>>>(b, a, n);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double cp2 = 2.0 * S * reps / (ms * 1e-3) / 1e12;
    printf("S = %5zu MiB   write+read %6.2f   read %6.2f   write %6.2f   copy(S/2 -> S/2) %6.2f   copy(S -> S) %6.2f  TB/s\n", mb, wr, rd, wo, cp, cp2);
  }
  return 0;
}
