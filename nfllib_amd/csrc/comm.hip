// comm.hip -- the batch split across the GPUs of one node (include/nflhip.h, "multi-GPU"): shard arithmetic, the
// shard-composable digest, peer-to-peer scatter / gather between the contexts of one process, and the one-process-per-
// device communicator on RCCL (grouped ncclSend / ncclRecv of contiguous shards).  The path has no data-path collective
// in steady state; everything here moves WHOLE SHARDS, once.
#include "../../include/nflhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace nflhip {
int set_error(int code, const std::string &msg);  // api.hip: the calling thread's nflhip_last_error text
}
using nflhip::set_error;

static int hip_error(hipError_t e, const char *where) {
  return set_error(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? NFLHIP_ERR_NO_DEVICE : NFLHIP_ERR_HIP,
                   std::string(where) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call)                                      \
  do {                                                    \
    hipError_t _e = (call);                               \
    if (_e != hipSuccess) return hip_error(_e, #call);    \
  } while (0)

// One message per peer and group, at most this many bytes: a config-D shard is 16 GiB and no transport should see it
// whole (nfllib_amd/sharding.py uses the same bound).  NFLHIP_COMM_PIECE_BYTES overrides it (tests use small pieces).
static size_t piece_bytes() {
  static const size_t v = [] {
    const char *e = getenv("NFLHIP_COMM_PIECE_BYTES");
    const long long x = e ? atoll(e) : 0;
    return x > 0 ? (size_t)x : ((size_t)1 << 30);
  }();
  return v;
}

static size_t poly_bytes_of(const nflhip_ctx *c) {
  return nflhip_degree(c) * nflhip_nmoduli(c) * (size_t)(nflhip_limb_bits(c) / 8);
}

// ---------------------------------------------------------------------------
// digest: sum_g (g + 1) * mix(word_g) mod 2^64 (nfllib_amd/sharding.py digest_words is the numpy statement of it)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long digest_mix(unsigned long long w) {
  return (w ^ (w >> 31)) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
}

template <typename T, int V>
__global__ void __launch_bounds__(256) k_digest(const T *__restrict__ d, unsigned long long nwords, unsigned long long first_word,
                                                unsigned long long *__restrict__ out) {
  struct alignas(sizeof(T) * V) Vec { T v[V]; };
  const unsigned long long nvec = nwords / V;
  unsigned long long acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const Vec x = reinterpret_cast<const Vec *>(d)[i];
#pragma unroll
    for (int k = 0; k < V; ++k) acc += (first_word + i * V + k + 1) * digest_mix((unsigned long long)x.v[k]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (unsigned long long g = nvec * V; g < nwords; ++g) acc += (first_word + g + 1) * digest_mix((unsigned long long)d[g]);
  // wave reduction, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)acc, off), hi = __shfl_xor((unsigned)(acc >> 32), off);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

template <typename T>
static hipError_t launch_digest(const void *d, unsigned long long nwords, unsigned long long first_word, unsigned long long *out,
                                hipStream_t st) {
  const dim3 grid(768), block(256);  // three workgroups per CU: the streaming sweet spot
  if (((uintptr_t)d & 15) == 0) hipLaunchKernelGGL((k_digest<T, 16 / sizeof(T)>), grid, block, 0, st, (const T *)d, nwords, first_word, out);
  else hipLaunchKernelGGL((k_digest<T, 1>), grid, block, 0, st, (const T *)d, nwords, first_word, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------
namespace {
struct Rccl {
  void *lib = nullptr;
  std::string why;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy the process already holds (PyTorch ships one under the same soname) is reused by the loader
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names)
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.lib) {
      r.why = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "dlopen failed");
      return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
      void *p = dlsym(r.lib, name);
      if (!p) { ok = false; r.why = std::string("RCCL lacks ") + name; }
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
  });
  return r;
}
int nccl_error(ncclResult_t e, const char *where) {
  Rccl &r = rccl();
  return set_error(NFLHIP_ERR_HIP, std::string(where) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}
}  // namespace
#define NCCLCHK(call)                                       \
  do {                                                      \
    ncclResult_t _e = (call);                               \
    if (_e != ncclSuccess) return nccl_error(_e, #call);    \
  } while (0)

struct nflhip_comm {
  nflhip_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  size_t poly_bytes = 0;
  unsigned long long *d_words = nullptr;  // [nranks + 1]: control-plane staging (all-gather / barrier)
};

// direct xGMI copies need peer access switched on once per ordered pair of devices (without it the runtime stages the copy)
static void enable_peer(int dev, int peer) {
  static std::mutex mu;
  static bool on[16][16] = {};
  if (dev == peer || dev < 0 || peer < 0 || dev >= 16 || peer >= 16) return;
  std::lock_guard<std::mutex> lk(mu);
  if (on[dev][peer]) return;
  on[dev][peer] = true;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) { (void)hipGetLastError(); return; }
  if (hipSetDevice(dev) == hipSuccess && hipDeviceEnablePeerAccess(peer, 0) != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
}

extern "C" {

int nflhip_shard_range(size_t total, int nranks, int rank, size_t *first, size_t *count) {
  if (!first || !count) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  if (nranks <= 0 || rank < 0 || rank >= nranks) return set_error(NFLHIP_ERR_INVALID, "rank out of range");
  const size_t base = total / (size_t)nranks, rem = total % (size_t)nranks, r = (size_t)rank;
  *first = r * base + (r < rem ? r : rem);
  *count = base + (r < rem ? 1 : 0);
  return NFLHIP_OK;
}

int nflhip_digest_dev(nflhip_ctx *ctx, const void *d_data, size_t first_poly, size_t batch, uint64_t *h_digest, void *stream) {
  if (!ctx || !h_digest || (batch && !d_data)) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(nflhip_ctx_device(ctx)));
  *h_digest = 0;
  if (batch == 0) return NFLHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long per = (unsigned long long)nflhip_degree(ctx) * nflhip_nmoduli(ctx);
  // one result word per device, owned by the library; calls on a device take turns (a digest is a control-plane call)
  static std::mutex mu[16];
  static unsigned long long *slot[16] = {};
  const int dev = nflhip_ctx_device(ctx);
  if (dev < 0 || dev >= 16) return set_error(NFLHIP_ERR_INVALID, "device index out of range");
  std::lock_guard<std::mutex> lk(mu[dev]);
  if (!slot[dev]) HIPCHK(hipMalloc((void **)&slot[dev], sizeof(unsigned long long)));
  unsigned long long *d_out = slot[dev];
  hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long), st);
  if (e == hipSuccess) {
    const int lb = nflhip_limb_bits(ctx);
    e = lb == 64   ? launch_digest<uint64_t>(d_data, per * batch, per * first_poly, d_out, st)
        : lb == 32 ? launch_digest<uint32_t>(d_data, per * batch, per * first_poly, d_out, st)
                   : launch_digest<uint16_t>(d_data, per * batch, per * first_poly, d_out, st);
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d_out, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_error(e, "digest");
  *h_digest = h;
  return NFLHIP_OK;
}

int nflhip_memcpy_peer_dev(nflhip_ctx *dst_ctx, void *d_dst, nflhip_ctx *src_ctx, const void *d_src, size_t bytes, void *stream) {
  if (!dst_ctx || !src_ctx) return set_error(NFLHIP_ERR_INVALID, "ctx is NULL");
  if (bytes == 0) return NFLHIP_OK;
  if (!d_dst || !d_src) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  const int dd = nflhip_ctx_device(dst_ctx), sd = nflhip_ctx_device(src_ctx);
  enable_peer(dd, sd);
  enable_peer(sd, dd);
  HIPCHK(hipSetDevice(dd));
  if (dd == sd) HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  else HIPCHK(hipMemcpyPeerAsync(d_dst, dd, d_src, sd, bytes, (hipStream_t)stream));
  return NFLHIP_OK;
}

// scatter (to_shards) / gather between n contexts of this process.  An event recorded on the producer's stream gates each
// copy; the copies run on the PEERS' streams (one per link); the root's stream then waits for every copy.
static int move_local(nflhip_ctx *const *ctxs, int n, void *const *shards, int root, void *full, size_t total,
                      void *const *streams, bool to_shards) {
  if (!ctxs || !shards || !streams || n <= 0 || root < 0 || root >= n) return set_error(NFLHIP_ERR_INVALID, "bad device list");
  for (int r = 0; r < n; ++r)
    if (!ctxs[r]) return set_error(NFLHIP_ERR_INVALID, "ctx is NULL");
  const size_t pb = poly_bytes_of(ctxs[root]);
  for (int r = 0; r < n; ++r)
    if (poly_bytes_of(ctxs[r]) != pb || nflhip_degree(ctxs[r]) != nflhip_degree(ctxs[root]))
      return set_error(NFLHIP_ERR_INVALID, "the contexts of a device list must have one shape");
  if (total && !full) return set_error(NFLHIP_ERR_INVALID, "NULL batch");
  const int rdev = nflhip_ctx_device(ctxs[root]);
  hipStream_t rst = (hipStream_t)streams[root];
  hipEvent_t ready = nullptr;
  std::vector<hipEvent_t> done((size_t)n, nullptr);
  int rc = NFLHIP_OK;
  auto cleanup = [&] {
    if (ready) { (void)hipSetDevice(rdev); (void)hipEventDestroy(ready); }
    for (int r = 0; r < n; ++r)
      if (done[r]) { (void)hipSetDevice(nflhip_ctx_device(ctxs[r])); (void)hipEventDestroy(done[r]); }
  };
#define LCHK(call)                                                          \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) { rc = hip_error(_e, #call); cleanup(); return rc; } \
  } while (0)
  if (to_shards) {  // the batch is ready once the root's stream gets here
    LCHK(hipSetDevice(rdev));
    LCHK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    LCHK(hipEventRecord(ready, rst));
  }
  for (int r = 0; r < n; ++r) {
    size_t first = 0, count = 0;
    nflhip_shard_range(total, n, r, &first, &count);
    if (count == 0) continue;
    if (!shards[r]) { cleanup(); return set_error(NFLHIP_ERR_INVALID, "NULL shard"); }
    const int dev = nflhip_ctx_device(ctxs[r]);
    hipStream_t st = (hipStream_t)streams[r];
    char *slice = (char *)full + first * pb;
    LCHK(hipSetDevice(dev));
    if (r != root) {
      if (to_shards) LCHK(hipStreamWaitEvent(st, ready, 0));
      // (gather: the shard is ready once ITS stream gets here -- the copy is enqueued on that very stream)
    }
    void *dst = to_shards ? shards[r] : (void *)slice;
    const void *src = to_shards ? (const void *)slice : (const void *)shards[r];
    const int ddev = to_shards ? dev : rdev, sdev = to_shards ? rdev : dev;
    enable_peer(ddev, sdev);
    enable_peer(sdev, ddev);
    LCHK(hipSetDevice(dev));
    hipStream_t cst = r == root ? rst : st;
    for (size_t off = 0; off < count * pb; off += piece_bytes()) {
      const size_t len = count * pb - off < piece_bytes() ? count * pb - off : piece_bytes();
      if (ddev == sdev) LCHK(hipMemcpyAsync((char *)dst + off, (const char *)src + off, len, hipMemcpyDeviceToDevice, cst));
      else LCHK(hipMemcpyPeerAsync((char *)dst + off, ddev, (const char *)src + off, sdev, len, cst));
    }
    if (r != root) {
      LCHK(hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
      LCHK(hipEventRecord(done[r], st));
    }
  }
  LCHK(hipSetDevice(rdev));
  for (int r = 0; r < n; ++r)
    if (done[r]) LCHK(hipStreamWaitEvent(rst, done[r], 0));  // the root may reuse / read the batch after this
#undef LCHK
  cleanup();
  return NFLHIP_OK;
}

int nflhip_scatter_local_dev(nflhip_ctx *const *ctxs, int n, void *const *d_shards, int root, const void *d_full, size_t total,
                             void *const *streams) {
  return move_local(ctxs, n, d_shards, root, const_cast<void *>(d_full), total, streams, true);
}
int nflhip_gather_local_dev(nflhip_ctx *const *ctxs, int n, void *d_full, int root, const void *const *d_shards, size_t total,
                            void *const *streams) {
  return move_local(ctxs, n, const_cast<void *const *>(reinterpret_cast<const void *const *>(d_shards)), root, d_full, total,
                    streams, false);
}

// ---------------------------------------------------------------------------
// one process per device: RCCL
// ---------------------------------------------------------------------------
int nflhip_comm_unique_id(unsigned char id[NFLHIP_COMM_ID_BYTES]) {
  static_assert(NFLHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  Rccl &r = rccl();
  if (!r.lib) return set_error(NFLHIP_ERR_UNSUPPORTED, r.why);
  ncclUniqueId u;
  NCCLCHK(r.GetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return NFLHIP_OK;
}

int nflhip_comm_create(nflhip_comm **out, nflhip_ctx *ctx, int nranks, int rank, const unsigned char id[NFLHIP_COMM_ID_BYTES]) {
  if (!out) return set_error(NFLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!ctx || !id) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  if (nranks <= 0 || rank < 0 || rank >= nranks) return set_error(NFLHIP_ERR_INVALID, "rank out of range");
  Rccl &r = rccl();
  if (!r.lib) return set_error(NFLHIP_ERR_UNSUPPORTED, r.why);
  nflhip_comm *c = new (std::nothrow) nflhip_comm();
  if (!c) return set_error(NFLHIP_ERR_NOMEM, "out of host memory");
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->device = nflhip_ctx_device(ctx);
  c->poly_bytes = poly_bytes_of(ctx);
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_words, ((size_t)nranks + 1) * sizeof(unsigned long long));
  if (e != hipSuccess) {
    delete c;
    return hip_error(e, "comm_create");
  }
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t ne = r.CommInitRank(&c->comm, nranks, u, rank);
  if (ne != ncclSuccess) {
    (void)hipFree(c->d_words);
    delete c;
    return nccl_error(ne, "ncclCommInitRank");
  }
  *out = c;
  return NFLHIP_OK;
}

int nflhip_comm_destroy(nflhip_comm *c) {
  if (!c) return NFLHIP_OK;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->d_words) (void)hipFree(c->d_words);
  delete c;
  return NFLHIP_OK;
}
int nflhip_comm_rank(const nflhip_comm *c) { return c ? c->rank : -1; }
int nflhip_comm_size(const nflhip_comm *c) { return c ? c->nranks : 0; }

// piece k of every peer travels in group k: all links stay busy, and each message is bounded
static int move_rccl(nflhip_comm *c, void *shard, void *full, size_t total, int root, void *stream, bool to_shards) {
  if (!c) return set_error(NFLHIP_ERR_INVALID, "comm is NULL");
  if (root < 0 || root >= c->nranks) return set_error(NFLHIP_ERR_INVALID, "root out of range");
  Rccl &r = rccl();
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const size_t pb = c->poly_bytes, piece = piece_bytes();
  size_t myfirst = 0, mycount = 0;
  nflhip_shard_range(total, c->nranks, c->rank, &myfirst, &mycount);
  if (mycount && !shard) return set_error(NFLHIP_ERR_INVALID, "NULL shard");
  if (c->rank == root) {
    if (total && !full) return set_error(NFLHIP_ERR_INVALID, "NULL batch");
    if (mycount) {  // the root's own shard never leaves the device
      char *slice = (char *)full + myfirst * pb;
      HIPCHK(hipMemcpyAsync(to_shards ? shard : (void *)slice, to_shards ? (const void *)slice : (const void *)shard, mycount * pb,
                            hipMemcpyDeviceToDevice, st));
    }
    size_t depth = 0;  // pieces of the largest shard (rank 0's)
    {
      size_t f0 = 0, c0 = 0;
      nflhip_shard_range(total, c->nranks, 0, &f0, &c0);
      depth = (c0 * pb + piece - 1) / piece;
    }
    for (size_t k = 0; k < depth; ++k) {
      NCCLCHK(r.GroupStart());
      for (int p = 0; p < c->nranks; ++p) {
        if (p == root) continue;
        size_t f = 0, n = 0;
        nflhip_shard_range(total, c->nranks, p, &f, &n);
        const size_t bytes = n * pb, off = k * piece;
        if (off >= bytes) continue;
        const size_t len = bytes - off < piece ? bytes - off : piece;
        char *at = (char *)full + f * pb + off;
        ncclResult_t ne = to_shards ? r.Send(at, len, ncclUint8, p, c->comm, st) : r.Recv(at, len, ncclUint8, p, c->comm, st);
        if (ne != ncclSuccess) { (void)r.GroupEnd(); return nccl_error(ne, to_shards ? "ncclSend" : "ncclRecv"); }
      }
      NCCLCHK(r.GroupEnd());
    }
  } else {
    const size_t bytes = mycount * pb;
    for (size_t off = 0; off < bytes; off += piece) {
      const size_t len = bytes - off < piece ? bytes - off : piece;
      NCCLCHK(r.GroupStart());
      ncclResult_t ne = to_shards ? r.Recv((char *)shard + off, len, ncclUint8, root, c->comm, st)
                                  : r.Send((const char *)shard + off, len, ncclUint8, root, c->comm, st);
      if (ne != ncclSuccess) { (void)r.GroupEnd(); return nccl_error(ne, to_shards ? "ncclRecv" : "ncclSend"); }
      NCCLCHK(r.GroupEnd());
    }
  }
  return NFLHIP_OK;
}

int nflhip_scatter_dev(nflhip_comm *comm, void *d_shard, const void *d_full, size_t total, int root, void *stream) {
  return move_rccl(comm, d_shard, const_cast<void *>(d_full), total, root, stream, true);
}
int nflhip_gather_dev(nflhip_comm *comm, void *d_full, const void *d_shard, size_t total, int root, void *stream) {
  return move_rccl(comm, const_cast<void *>(d_shard), d_full, total, root, stream, false);
}

int nflhip_comm_allgather_u64(nflhip_comm *c, uint64_t mine, uint64_t *h_all, void *stream) {
  if (!c || !h_all) return set_error(NFLHIP_ERR_INVALID, "NULL argument");
  Rccl &r = rccl();
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  unsigned long long m = mine;
  HIPCHK(hipMemcpyAsync(c->d_words + c->nranks, &m, sizeof(m), hipMemcpyHostToDevice, st));
  NCCLCHK(r.AllGather(c->d_words + c->nranks, c->d_words, 1, ncclUint64, c->comm, st));
  HIPCHK(hipMemcpyAsync(h_all, c->d_words, (size_t)c->nranks * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return NFLHIP_OK;
}

int nflhip_comm_barrier(nflhip_comm *c, void *stream) {
  if (!c) return set_error(NFLHIP_ERR_INVALID, "comm is NULL");
  Rccl &r = rccl();
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(c->d_words, 0, sizeof(unsigned long long), st));
  NCCLCHK(r.AllReduce(c->d_words, c->d_words, 1, ncclUint64, ncclSum, c->comm, st));
  HIPCHK(hipStreamSynchronize(st));
  return NFLHIP_OK;
}

}  // extern "C"
