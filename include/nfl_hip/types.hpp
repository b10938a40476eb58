// nfl_hip/types.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// simd tags, params<T>, sampler tags, error checking, the recording path's lock, per-device contexts and buffer pools, sampler state.
#ifndef NFL_HIP_TYPES_HPP
#define NFL_HIP_TYPES_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {

// ---------------------------------------------------------------- simd tags (arch/common.hpp:11-26, arch.hpp:6-17)
// The reference selects its vector ISA with a tag; here the only host tag is `serial` (the device is selected by
// linking libnflhip.so, not by a tag) and it is accepted wherever the reference's spelling names one.
namespace simd {
struct serial {
  template <class T> static inline T load(T const *p) { return *p; }
  template <class T> static inline void store(T *p, T const v) { *p = v; }
  template <class T> struct elt_count { static constexpr size_t value = 1; };
  static constexpr int mode = 0;
};
}  // namespace simd
template <class... M> struct common_mode;
template <class M> struct common_mode<M> { using type = M; };
template <class M0, class... M> struct common_mode<M0, M...> { using type = typename common_mode<M0, typename common_mode<M...>::type>::type; };
template <class M> struct common_mode<M, M> { using type = M; };
#ifndef CC_SIMD
#define CC_SIMD nfl::simd::serial
#endif

// ---------------------------------------------------------------- meta.hpp:12-45
namespace impl {
template <size_t N> struct _log2 { static constexpr size_t value = 1 + _log2<N / 2>::value; };
template <> struct _log2<1> { static constexpr size_t value = 0; };
}  // namespace impl
template <size_t N> struct static_log2 { static constexpr size_t value = impl::_log2<N>::value; };
template <> struct static_log2<0> {};

// ---------------------------------------------------------------- params<T> (params.hpp:11-119)
// Same member names as the reference's.  The tables are class-template statics (defined below, in the header, ODR-safe
// in C++11) filled from the generated initialisers of include/nflhip_params.h -- every modulus the reference offers.
template <class T> struct params;
namespace detail {
template <class Dummy> struct params_tables_u16 {
  static constexpr uint16_t P[NFLHIP_U16_NMODULI] = NFLHIP_U16_P_INIT;
  static constexpr uint16_t Pn[NFLHIP_U16_NMODULI] = NFLHIP_U16_PN_INIT;
  static constexpr uint16_t primitive_roots[NFLHIP_U16_NMODULI] = NFLHIP_U16_ROOTS_INIT;
  static constexpr uint16_t invkMaxPolyDegree[NFLHIP_U16_NMODULI] = NFLHIP_U16_INVKMAX_INIT;
};
template <class D> constexpr uint16_t params_tables_u16<D>::P[NFLHIP_U16_NMODULI];
template <class D> constexpr uint16_t params_tables_u16<D>::Pn[NFLHIP_U16_NMODULI];
template <class D> constexpr uint16_t params_tables_u16<D>::primitive_roots[NFLHIP_U16_NMODULI];
template <class D> constexpr uint16_t params_tables_u16<D>::invkMaxPolyDegree[NFLHIP_U16_NMODULI];
template <class Dummy> struct params_tables_u32 {
  static constexpr uint32_t P[NFLHIP_U32_NMODULI] = NFLHIP_U32_P_INIT;
  static constexpr uint32_t Pn[NFLHIP_U32_NMODULI] = NFLHIP_U32_PN_INIT;
  static constexpr uint32_t primitive_roots[NFLHIP_U32_NMODULI] = NFLHIP_U32_ROOTS_INIT;
  static constexpr uint32_t invkMaxPolyDegree[NFLHIP_U32_NMODULI] = NFLHIP_U32_INVKMAX_INIT;
};
template <class D> constexpr uint32_t params_tables_u32<D>::P[NFLHIP_U32_NMODULI];
template <class D> constexpr uint32_t params_tables_u32<D>::Pn[NFLHIP_U32_NMODULI];
template <class D> constexpr uint32_t params_tables_u32<D>::primitive_roots[NFLHIP_U32_NMODULI];
template <class D> constexpr uint32_t params_tables_u32<D>::invkMaxPolyDegree[NFLHIP_U32_NMODULI];
template <class Dummy> struct params_tables_u64 {
  static constexpr uint64_t P[NFLHIP_U64_NMODULI] = NFLHIP_U64_P_INIT;
  static constexpr uint64_t Pn[NFLHIP_U64_NMODULI] = NFLHIP_U64_PN_INIT;
  static constexpr uint64_t primitive_roots[NFLHIP_U64_NMODULI] = NFLHIP_U64_ROOTS_INIT;
  static constexpr uint64_t invkMaxPolyDegree[NFLHIP_U64_NMODULI] = NFLHIP_U64_INVKMAX_INIT;
};
template <class D> constexpr uint64_t params_tables_u64<D>::P[NFLHIP_U64_NMODULI];
template <class D> constexpr uint64_t params_tables_u64<D>::Pn[NFLHIP_U64_NMODULI];
template <class D> constexpr uint64_t params_tables_u64<D>::primitive_roots[NFLHIP_U64_NMODULI];
template <class D> constexpr uint64_t params_tables_u64<D>::invkMaxPolyDegree[NFLHIP_U64_NMODULI];
}  // namespace detail
template <> struct params<uint16_t> : detail::params_tables_u16<void> {
  typedef uint16_t value_type;
  typedef int16_t signed_value_type;
  typedef uint32_t greater_value_type;
  typedef value_type *poly_t;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U16_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U16_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 16;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U16_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U16_KMAX_LOG2;
};
template <> struct params<uint32_t> : detail::params_tables_u32<void> {
  typedef uint32_t value_type;
  typedef int32_t signed_value_type;
  typedef uint64_t greater_value_type;
  typedef value_type *poly_t;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U32_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U32_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 32;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U32_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U32_KMAX_LOG2;
};
template <> struct params<uint64_t> : detail::params_tables_u64<void> {
  typedef uint64_t value_type;
  typedef int64_t signed_value_type;
  typedef unsigned __int128 greater_value_type;
  typedef value_type *poly_t;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U64_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U64_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 64;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U64_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U64_KMAX_LOG2;
};

// ---- sampler tags (poly.hpp:42-67).  `uniform()` and the other tags draw fresh randomness on every use, like the
// reference (process-wide key from the OS, one new keystream per call -- see detail::sampler below);
// `uniform(seed)` is this header's addition: a seeded, reproducible operand (what benches and tests use).
struct uniform {
  uint64_t seed;
  bool seeded;
  uniform() : seed(0), seeded(false) {}
  explicit uniform(uint64_t s) : seed(s), seeded(true) {}
};
struct non_uniform {
  uint64_t upper_bound;
  uint64_t amplifier;
  non_uniform(uint64_t ub) : upper_bound{ub}, amplifier{1} {}
  non_uniform(uint64_t ub, uint64_t amp) : upper_bound{ub}, amplifier{amp} {}
};
struct hwt_dist {  // hamming weight distribution
  uint32_t hwt;
  hwt_dist(uint32_t hwt_) : hwt(hwt_) {}
};
struct ZO_dist {  // P(1) = P(-1) = ((rho + 1) / 256) / 2
  uint8_t rho;
  ZO_dist(uint8_t rho_ = 0x7F) : rho(rho_) {}
};

namespace detail {
#ifdef NFL_HIP_WITH_GMP
inline mpz_srcptr as_mpz(mpz_t const &v) { return v; }
#ifdef NFL_HIP_HAVE_GMPXX
inline mpz_srcptr as_mpz(mpz_class const &v) { return v.get_mpz_t(); }
#endif
#endif

inline void check(nflhip_ctx *ctx, int rc, const char *what) {
  if (rc != NFLHIP_OK) throw std::runtime_error(std::string("nfl(hip): ") + what + ": " + nflhip_last_error(ctx));
}

// CHECK_STRICTMOD (debug.hpp:21-37): the reference's ASSERT_STRICTMOD is assert(), i.e. active when the macro is defined and
// NDEBUG is not -- how its own tests are built (tests/CMakeLists.txt:10).  It asserts x < p on what goes INTO the
// transforms (core.hpp:457-462) and into addmod / submod / mulmod / mulmod_shoup (ops.hpp:131,148,190,211,235).  Here the
// same operands are checked where an operation is issued -- host words on the host, resident values by one streaming
// compare on the device (nflhip_check_range[_dev]) -- and a violation throws std::runtime_error, this header's error
// convention, instead of aborting.  The Shoup companion b' of mulmod_shoup is a quotient, not a residue: exempt.
#if defined(CHECK_STRICTMOD) && !defined(NDEBUG)
static constexpr bool strictmod = true;
#else
static constexpr bool strictmod = false;
#endif
inline void strict_fail(const char *what) {
  throw std::runtime_error(std::string("nfl(hip): CHECK_STRICTMOD: ") + what + ": an operand word is not below its modulus");
}
inline void strict_host(nflhip_ctx *ctx, const void *words, size_t polys, const char *what) {
  int bad = 0;
  check(ctx, nflhip_check_range(ctx, words, polys, &bad), what);
  if (bad) strict_fail(what);
}
inline void strict_dev(nflhip_ctx *ctx, const void *d, size_t polys, void *stream, const char *what) {
  int bad = 0;
  check(ctx, nflhip_check_range_dev(ctx, d, polys, &bad, stream), what);
  if (bad) strict_fail(what);
}
// operands of a postfix program that are only ever consumed as the Shoup companion of a mulmod_shoup (bit k = operand k)
inline unsigned strict_exempt(const unsigned char *code, size_t len) {
  int leaf[NFLHIP_EXPR_MAX_LEN + 1];
  int sp = 0;
  unsigned as_companion = 0, as_value = 0;
  for (size_t q = 0; q < len; ++q) {
    const unsigned char b = code[q];
    if (b < NFLHIP_EXPR_MAX_OPERANDS) {
      leaf[sp++] = int(b);
    } else if (b == NFLHIP_EXPR_MUL_SHOUP && sp >= 3) {
      if (leaf[sp - 1] >= 0) as_companion |= 1u << leaf[sp - 1];
      for (int k = 2; k <= 3; ++k)
        if (leaf[sp - k] >= 0) as_value |= 1u << leaf[sp - k];
      sp -= 2;
      leaf[sp - 1] = -1;
    } else if (b == NFLHIP_EXPR_COMPUTE_SHOUP && sp >= 1) {
      if (leaf[sp - 1] >= 0) as_value |= 1u << leaf[sp - 1];
      leaf[sp - 1] = -1;
    } else if (sp >= 2) {
      for (int k = 1; k <= 2; ++k)
        if (leaf[sp - k] >= 0) as_value |= 1u << leaf[sp - k];
      --sp;
      leaf[sp - 1] = -1;
    }
  }
  if (sp == 1 && leaf[0] >= 0) as_value |= 1u << leaf[0];   // (a bare copy)
  return as_companion & ~as_value;
}

// The lock of the recording path (detail::lazy<P>::mu, the buffer pool of detail::context): recursive, ONE atomic operation per
// outermost acquisition and a plain release store -- a std::recursive_mutex costs a locked instruction each way plus two
// calls into libc, and the LWE demo's loop takes the queue's lock sixteen times per encryption: a third of its host time
// inside a process that has other threads at all (the HIP runtime's), where glibc's single-thread shortcuts are off.
// Waiters spin, then yield, then sleep (a queue run may hold the lock for hundreds of microseconds).
class light_lock {
  // ---- the plain lock: what every thread but the bias holder takes (and every thread once the bias is gone)
  std::atomic<const void *> owner_;
  unsigned depth_;
  // ---- the bias (round 6): the FIRST thread that takes this lock keeps a claim on it and from then on enters with two plain
  // stores and two plain loads -- no locked instruction, no fence: 11 acquisitions per recorded LWE encryption were a sixth of
  // the host's time.  Another thread that wants the lock takes the plain lock, raises revoke_, issues
  // membarrier(PRIVATE_EXPEDITED) -- a full barrier on every running thread of the process, so the holder's "bias_depth_ = 1;
  // load revoke_" cannot both slip past it -- and waits for bias_depth_ == 0; the holder, seeing revoke_, backs off and
  // waits for it to clear.  After kMaxRevocations of these (a program that records from several threads) the bias is withdrawn for
  // good and the lock is the plain one.  No membarrier in the kernel / sandbox (or NFL_HIP_NO_BIASED_LOCK set): never biased.
  std::atomic<const void *> bias_owner_;
  std::atomic<unsigned> bias_depth_;    // written by the bias holder only
  std::atomic<unsigned> revoke_;        // 1 while a thread that holds the plain lock keeps the bias holder out
  std::atomic<unsigned> revocations_;
  bool revoking_;                       // (plain-lock holder's note: it raised revoke_ at its outermost acquisition)
  enum { kMaxRevocations = 16 };
  static const void *me() {
    static thread_local char tag;
    return &tag;
  }
  static const void *no_bias() {        // sentinel: the bias was withdrawn (or never available)
    static char tag;
    return &tag;
  }
  static bool asymmetric_barrier_available() {
#if defined(__linux__) && defined(__NR_membarrier)
    static const bool ok = !std::getenv("NFL_HIP_NO_BIASED_LOCK") && syscall(__NR_membarrier, 16 /* REGISTER_PRIVATE_EXPEDITED */, 0, 0) == 0;
    return ok;
#else
    return false;
#endif
  }
  static void barrier_all_threads() {
#if defined(__linux__) && defined(__NR_membarrier)
    if (syscall(__NR_membarrier, 8 /* PRIVATE_EXPEDITED */, 0, 0) != 0) std::abort();   // (registered above: cannot fail)
#endif
  }
  void lock_plain(const void *self) {
    if (owner_.load(std::memory_order_relaxed) == self) {
      ++depth_;
      return;
    }
    const void *expected = nullptr;
    for (unsigned spins = 0; !owner_.compare_exchange_weak(expected, self, std::memory_order_acquire, std::memory_order_relaxed); ++spins) {
      expected = nullptr;
      if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));
      else if (spins > 64) std::this_thread::yield();
    }
    depth_ = 1;
    // outermost acquisition of the plain lock: keep a bias holder (another thread) out for as long as we hold it
    const void *b = bias_owner_.load(std::memory_order_acquire);
    revoking_ = false;
    if (b != nullptr && b != no_bias() && b != self) {
      revoke_.store(1, std::memory_order_seq_cst);
      barrier_all_threads();
      for (unsigned spins = 0; bias_depth_.load(std::memory_order_acquire) != 0; ++spins) {
        if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (spins > 64) std::this_thread::yield();
      }
      revoking_ = true;
      if (revocations_.fetch_add(1, std::memory_order_relaxed) + 1 >= kMaxRevocations)
        bias_owner_.store(no_bias(), std::memory_order_release);   // (the holder is outside and kept out: it re-reads this when it retries)
    }
  }
 public:
  light_lock() : owner_(nullptr), depth_(0), bias_owner_(nullptr), bias_depth_(0), revoke_(0), revocations_(0), revoking_(false) {}
  light_lock(const light_lock &) = delete;
  light_lock &operator=(const light_lock &) = delete;
  void lock() {
    const void *self = me();
    for (;;) {
      const void *b = bias_owner_.load(std::memory_order_relaxed);
      if (b == self) {
        const unsigned d = bias_depth_.load(std::memory_order_relaxed);
        if (d) {                                           // recursive acquisition by the holder
          bias_depth_.store(d + 1, std::memory_order_relaxed);
          return;
        }
        bias_depth_.store(1, std::memory_order_relaxed);
        std::atomic_signal_fence(std::memory_order_seq_cst);   // compiler barrier; the revoker's membarrier is the hardware one
        if (!revoke_.load(std::memory_order_acquire)) return;  // FAST PATH
        bias_depth_.store(0, std::memory_order_release);       // somebody holds the plain lock and wants us out: wait, then retry
        for (unsigned spins = 0; revoke_.load(std::memory_order_acquire) != 0; ++spins) {
          if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));
          else if (spins > 64) std::this_thread::yield();
        }
        continue;
      }
      if (b == nullptr && asymmetric_barrier_available()) {   // nobody has the bias yet: the first thread claims it
        const void *none = nullptr;
        if (bias_owner_.compare_exchange_strong(none, self, std::memory_order_acq_rel)) {
          // (a thread may be inside the plain lock right now -- it read bias_owner_ == nullptr before our claim: wait for it once)
          for (unsigned spins = 0; owner_.load(std::memory_order_acquire) != nullptr; ++spins) {
            if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));
            else if (spins > 64) std::this_thread::yield();
          }
        }
        continue;
      }
      if (b == nullptr) {                                      // no asymmetric barrier on this system: plain lock for everybody
        const void *none = nullptr;
        bias_owner_.compare_exchange_strong(none, no_bias(), std::memory_order_acq_rel);
        continue;
      }
      lock_plain(self);
      return;
    }
  }
  void unlock() {
    if (bias_owner_.load(std::memory_order_relaxed) == me()) {
      const unsigned d = bias_depth_.load(std::memory_order_relaxed);
      if (d) {
        bias_depth_.store(d - 1, std::memory_order_release);
        return;
      }
    }
    if (--depth_ == 0) {
      if (revoking_) {
        revoking_ = false;
        revoke_.store(0, std::memory_order_release);
      }
      owner_.store(nullptr, std::memory_order_release);
    }
  }
};

// Every ring type whose per-polynomial operations can be deferred (detail::lazy<P> below) registers the function that
// runs its queue.  Whoever is about to invalidate something recorded operations refer to -- a FastGaussianNoise that
// dies (its device tables), nfl::set_sampler_key (the key recorded draws will be made with) -- runs all queues first.
// Leaked on purpose: objects with static storage may call it while the program's other statics are being destroyed.
struct queue_registry {
  std::mutex mu;
  std::vector<void (*)()> runners;
  static queue_registry &get() {
    static queue_registry *r = new queue_registry;
    return *r;
  }
  void add(void (*f)()) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto g : runners) if (g == f) return;
    runners.push_back(f);
  }
  void run_all() {
    std::vector<void (*)()> fs;
    {
      std::lock_guard<std::mutex> lk(mu);
      fs = runners;
    }
    for (auto f : fs) f();
  }
};

// The process-wide sampler state: the counterpart of fastrandombytes' static key and nonce
// (lib/prng/fastrandombytes.cpp:17-37).  The key is drawn from the OS once; every sampling call takes the next
// 64-bit stream id.  nfl::set_sampler_key() pins both for reproducible runs.
struct sampler {
  unsigned char key[32];
  std::mutex key_mu;  // set_sampler_key against a queue run's copy of the key (another thread)
  std::atomic<uint64_t> next;
  void copy_key(unsigned char out[32]) {
    std::lock_guard<std::mutex> lk(key_mu);
    std::memcpy(out, key, 32);
  }
  sampler() : next(0) {
    std::random_device rd;
    for (int i = 0; i < 32; i += 4) {
      const uint32_t v = rd();
      std::memcpy(key + i, &v, 4);
    }
  }
  static sampler &get() {
    static sampler s;
    return s;
  }
};

// One device context per (T, Degree, NbModuli): the replacement of the reference's static `core base` / `GMP gmp`
// members (poly.hpp:247, 275), created on first use (function-local static => thread-safe, never before main()).
// It also owns what the resident poly_p handles share: ONE stream every resident operation is enqueued on (so
// successive operations are ordered without events) and a free list of polynomial-sized device buffers (hipMalloc /
// hipFree per temporary would cost more than the kernels).
// The device the per-polynomial surface (poly, poly_p, the static contexts) lives on: NFL_HIP_DEVICE in the environment,
// or nfl::set_device() before the first polynomial of a ring type is used; 0 otherwise.  Batches name their device
// themselves (device_batch(count, device), sharded_batch).
inline std::atomic<int> &default_device() {
  static std::atomic<int> d(getenv("NFL_HIP_DEVICE") ? atoi(getenv("NFL_HIP_DEVICE")) : 0);
  return d;
}

template <class T, size_t Degree, size_t NbModuli> struct context {
  nflhip_ctx *ctx;
  void *stream;
  int device;
  light_lock mu;
  static constexpr size_t poly_bytes = Degree * NbModuli * sizeof(T);
  static constexpr size_t chunk_bytes = (poly_bytes + 255) / 256 * 256;          // device buffers are 256-byte aligned
  // Device buffers of the resident handles: slabs (256 MiB first, doubling up to 2 GiB -- the device has 288 GB) carved
  // into polynomial-sized chunks.  A slab hands out fresh chunks in address order (consecutive acquisitions are
  // CONTIGUOUS, which is what lets deferred per-polynomial operations run as dense batches), keeps the chunks it gets
  // back on a free list for single acquisitions, and starts over once every chunk is back.
  struct slab {
    char *base;
    size_t chunks, bump, live;
    std::vector<void *> free;
  };
  std::map<char *, slab> slabs;
  const bool is_static;  // the function-local static of inst(): the one the resident poly_p handles allocate from
  slab *last_released;
  size_t next_slab_bytes;
  explicit context(int dev, bool is_static_ = false)
      : ctx(nullptr), stream(nullptr), device(dev), is_static(is_static_), last_released(nullptr), next_slab_bytes(size_t(256) << 20) {
    static_assert(NbModuli <= params<T>::kMaxNbModuli, "not enough moduli of this size (params.hpp)");
    static_assert(Degree <= params<T>::kMaxPolyDegree, "degree is not lower or equal than kMaxPolyDegree");
    int rc = nflhip_ctx_create(&ctx, dev, int(sizeof(T) * 8), Degree, NbModuli, params<T>::P, params<T>::primitive_roots,
                               params<T>::invkMaxPolyDegree, params<T>::kMaxLog2);
    if (rc != NFLHIP_OK) throw std::runtime_error(std::string("nfl(hip): context: ") + nflhip_last_error(nullptr));
    rc = nflhip_stream_create(ctx, &stream);
    if (rc != NFLHIP_OK) {
      nflhip_ctx_destroy(ctx);
      throw std::runtime_error(std::string("nfl(hip): context stream: ") + nflhip_last_error(nullptr));
    }
    if (is_static) alive() = true;
  }
  ~context() {
    if (is_static) alive() = false;
    nflhip_stream_sync(ctx, stream);
    for (auto &kv : slabs) nflhip_free(ctx, kv.first);
    nflhip_stream_destroy(ctx, stream);
    nflhip_ctx_destroy(ctx);
  }
  context(const context &) = delete;
  context &operator=(const context &) = delete;
  static bool &alive() {  // false once the static below has been destroyed (objects with static storage may outlive it)
    static bool a = false;
    return a;
  }
  static context &inst() {
    static context c(default_device().load(), true);
    return c;
  }
  // The context of this ring type on `device`: the static one for the default device, otherwise one per device created on
  // first use (what device_batch(count, device) and sharded_batch run on).  Same tables everywhere: they are a
  // deterministic function of params<T> (core.hpp:625-686), so nothing is broadcast.
  static context &on(int dev) {
    context &def = inst();
    if (dev == def.device) return def;
    static std::mutex m;
    static std::map<int, std::unique_ptr<context>> others;
    std::lock_guard<std::mutex> lk(m);
    std::unique_ptr<context> &slot = others[dev];
    if (!slot) slot.reset(new context(dev));
    return *slot;
  }
  static nflhip_ctx *get() { return inst().ctx; }
  static void *queue() { return inst().stream; }

  slab &grow(size_t min_chunks) {
    size_t bytes = next_slab_bytes;
    while (bytes / chunk_bytes < min_chunks) bytes *= 2;
    if (next_slab_bytes < (size_t(2) << 30)) next_slab_bytes *= 2;
    const size_t n = bytes / chunk_bytes ? bytes / chunk_bytes : 1;
    void *mem = nullptr;
    check(ctx, nflhip_malloc(ctx, &mem, n * chunk_bytes), "device allocation");
    slab &sl = slabs[static_cast<char *>(mem)];
    sl.base = static_cast<char *>(mem);
    sl.chunks = n;
    sl.bump = sl.live = 0;
    return sl;
  }
  // `cnt` buffers, as contiguous as the slabs allow (fresh space first; recycled chunks only when no slab has room)
  void acquire_many_locked(size_t cnt, void **out) {
    struct sorter {  // recycled chunks come back in release order: hand them out by address, neighbours together
      void **o; size_t n;
      ~sorter() {  // (fresh space already is in order; a loop's temporaries, released in order, come back reversed)
        if (std::is_sorted(o, o + n)) return;
        if (std::is_sorted(o, o + n, std::greater<void *>())) std::reverse(o, o + n);
        else std::sort(o, o + n);
      }
    } srt{out, cnt};
    size_t got = 0;
    while (got < cnt) {
      slab *best = nullptr;
      for (auto &kv : slabs)
        if (kv.second.bump < kv.second.chunks && (!best || kv.second.chunks - kv.second.bump > best->chunks - best->bump)) best = &kv.second;
      if (!best) {
        // recycle before growing without bound -- but only when the recycled chunks cover what is still missing: a few
        // scattered chunks in front of a fresh slab cut a loop's dense result arrays into as many launches
        size_t recyclable = 0;
        for (auto &kv : slabs) recyclable += kv.second.free.size();
        if (recyclable >= cnt - got) {
          for (auto &kv : slabs) {
            slab &sl = kv.second;
            while (got < cnt && !sl.free.empty()) {
              out[got++] = sl.free.back();
              sl.free.pop_back();
              ++sl.live;
            }
          }
          return;
        }
        best = &grow(cnt - got);
      }
      while (got < cnt && best->bump < best->chunks) {
        out[got++] = best->base + best->bump++ * chunk_bytes;
        ++best->live;
      }
    }
  }
  static void acquire_many(size_t cnt, void **out) {
    context &c = inst();
    std::lock_guard<light_lock> lk(c.mu);
    c.acquire_many_locked(cnt, out);
  }
  static void *acquire() {
    context &c = inst();
    std::lock_guard<light_lock> lk(c.mu);
    for (auto &kv : c.slabs)
      if (!kv.second.free.empty()) {
        void *p = kv.second.free.back();
        kv.second.free.pop_back();
        ++kv.second.live;
        return p;
      }
    void *p = nullptr;
    c.acquire_many_locked(1, &p);
    return p;
  }
  static void release(void *p) {
    if (!p || !alive()) return;  // (after teardown the runtime reclaims it)
    context &c = inst();
    std::lock_guard<light_lock> lk(c.mu);
    slab *hit = c.last_released;  // (neighbouring handles die together: the slab of the previous release, usually)
    if (!hit || static_cast<char *>(p) < hit->base || static_cast<char *>(p) >= hit->base + hit->chunks * chunk_bytes) {
      auto it = c.slabs.upper_bound(static_cast<char *>(p));
      if (it == c.slabs.begin()) return;
      hit = c.last_released = &(--it)->second;  // (map nodes do not move; slabs are only removed by the destructor)
    }
    slab &sl = *hit;
    // stream-ordered reuse: every consumer of these buffers runs on `stream`
    if (--sl.live == 0) {
      sl.bump = 0;
      sl.free.clear();
    } else {
      sl.free.push_back(p);
    }
  }
};

struct uninitialized_t {};  // poly(uninitialized_t): storage that is about to be overwritten entirely

inline uint64_t splitmix64_at(uint64_t seed, int operand, uint64_t g) {
  uint64_t z = (seed ^ (uint64_t(operand) << 62)) + (g + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace detail
}  // namespace nfl
#endif  // NFL_HIP_TYPES_HPP
