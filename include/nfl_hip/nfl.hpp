// nfl_hip/nfl.hpp -- header-only host surface of the MI355X NTT polynomial-ring engine.
//
// Keeps the template surface of the reference's nfl::poly<T, Degree, NbModuli>
// (include/nfl/poly.hpp:82-352), nfl::poly_p (include/nfl/poly_p.hpp:11-204) and their
// expression-template operators (include/nfl/ops.hpp:18-97, 249-277) so existing callers compile
// unchanged -- the reference's own test programs build against this header as they are
// (tests/reftests/Makefile) -- but every whole-polynomial operation is forwarded through the C ABI of
// include/nflhip.h to hand-written HIP kernels.  There is no CPU arithmetic on the hot path in this
// header: without libnflhip.so + a GPU every operation throws std::runtime_error (the reference's error
// convention: core.hpp:111-115).  Reached as <nfl.hpp>, <nfl/poly.hpp>, <nfl/poly_p.hpp>, ... through the
// forwarding headers of include/nfl/.
//
// What is kept (same names, argument meaning, error behaviour):
//   params<T>::P / Pn / primitive_roots / invkMaxPolyDegree / kMax* (all moduli)  params.hpp:11-119
//   storage layout T _data[NbModuli*Degree], 32-byte aligned, modulus-major      poly.hpp:87-88,156-157
//   ctors / set(): value, initializer_list, iterator range (+reduce_coeffs)       core.hpp:64-137
//   the random constructors uniform / non_uniform / ZO_dist / hwt_dist / gaussian core.hpp:146-391
//   operator()(cm,i), begin/end, data(), get_modulus, degree/nmoduli/nbits        poly.hpp:142-162
//   ntt_pow_phi(), invntt_pow_invphi()                                            poly.hpp:167-168
//   operator+ - * == !=, shoup(a*b,b'), compute_shoup(b), nested expressions      poly.hpp:346-352
//   ops::make_op<ops::NAME<T, tag>>(...), nfl::simd::serial, CC_SIMD             ops.hpp:225-260, arch/common.hpp:11-26
//   explicit operator bool on polys, implicit on == / != expressions             core.hpp:39-43, ops.hpp:81-95
//   poly::core (ntt / inv_ntt statics) + `base` tables through the
//   tests::poly_tests_proxy friend                                                poly.hpp:69-76,85,196-247
//   serialize_manually / deserialize_manually / serialize(Archive)                poly.hpp:180-191
//   the GMP-typed surface (mpz_t / mpz_class ctors, set_mpz, poly2mpz, mpz2poly,
//   moduli_product / modulus_shoup / lifting_integers), on poly and poly_p        poly.hpp:249-307, poly_p.hpp:186-200
//   nfl::add / sub / mul, poly_from_modulus, poly_p_from_modulus, operator<<      poly.hpp:314-343
//   FastGaussianNoise (ctor, getNoise), nfl::rdtsc, nfl::fastrandombytes          FastGaussianNoise.hpp, fastrandombytes.h
// What differs, on purpose:
//   * tables live in a lazily created per-(T,Degree,NbModuli) device context, never at static-init time
//     (the reference's `static core base`, poly.hpp:247);
//   * nfl::poly_p is RESIDENT: its shared payload lives in HBM next to an (optional) host image, with a
//     valid-on-host / valid-on-device pair of bits.  Operator expressions, transforms, comparisons and random
//     constructors on poly_p handles run the *_dev entry points on the context's stream and never touch the host;
//     only operator()(cm,i), poly_obj(), serialisation and the GMP surface force a device-to-host copy
//     (SURVEY.md section 8(f) rank 2).  nfl::poly keeps its words inline on the host (poly.hpp:87-88) and its
//     member operations therefore cross PCIe per call -- use poly_p or the batch forms for throughput;
//   * CRT lift/project also exchange little-endian 64-bit limb vectors (poly2limbs / limbs2poly == the
//     mpz_export/mpz_import image of poly2mpz / mpz2poly, gmp.hpp:183-219);
//   * nfl::batch::* and nfl::device_batch<P> operate on contiguous arrays of polys (dense
//     [batch][NbModuli][Degree], as tests/tools.h:6-17 allocates) in ONE device pass;
//   * nfl::uniform(seed) is an addition: a seeded counter-based operand with the reference's
//     mask-then-subtract rule (core.hpp:165-176); plain nfl::uniform() draws fresh randomness like the reference.
// GMP: define NFL_HIP_WITH_GMP before inclusion (the <nfl.hpp> forwarding header does, as the reference always
// includes <gmpxx.h>: poly.hpp:33); NFL_HIP_REFERENCE_WORDS makes ZO_dist / hwt_dist store +1 as p + 1 like the
// reference's raw words (core.hpp:341,387) instead of the canonical 1.
#ifndef NFL_HIP_NFL_HPP
#define NFL_HIP_NFL_HPP

#include <algorithm>
#include <functional>
#include <array>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cinttypes>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <initializer_list>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <strings.h>
#include <thread>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>
#if defined(__linux__)
#include <sys/syscall.h>   // membarrier(2): the asymmetric barrier of detail::light_lock
#include <unistd.h>
#endif

#include "../nflhip.h"
#include "../nflhip_params.h"

#ifdef NFL_HIP_WITH_GMP
#include <gmp.h>
#if defined(__has_include)
#if __has_include(<gmpxx.h>)
#include <gmpxx.h>
#define NFL_HIP_HAVE_GMPXX 1
#endif
#endif
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

/* deferred execution of operations on resident poly_p handles (see detail::lazy): on by default; switching it off makes
 * every operation launch when it is called (flush the ring types in use first: poly_p<...>::flush()) */
inline void set_deferred(bool on);

/* the GPU the per-polynomial surface runs on (default: NFL_HIP_DEVICE or 0); call before the first polynomial of a ring
 * type is used -- the static context of that type is created once */
inline void set_device(int device) { detail::default_device().store(device); }
inline int device_count() {
  int n = 0;
  detail::check(nullptr, nflhip_device_count(&n), "device_count");
  return n;
}

/* pin the sampler state: `key` (32 bytes) and the id of the next keystream -- reproducible runs */
inline void set_sampler_key(const unsigned char key[32], uint64_t next_stream = 0) {
  detail::sampler &s = detail::sampler::get();
  detail::queue_registry::get().run_all();  // draws recorded so far are made with the key they were recorded under
  std::lock_guard<std::mutex> lk(s.key_mu);
  std::memcpy(s.key, key, 32);
  s.next.store(next_stream);
}

/* nfl::rdtsc (FastGaussianNoise.hpp:117-122): the cycle counter callers time getNoise with (tests/prng_demo_main.cpp:17) */
inline uint64_t rdtsc(void) {
#if defined(__x86_64__) || defined(__i386__)
  uint32_t lo, hi;
  __asm__ volatile("rdtsc" : "=a"(lo), "=d"(hi));
  return (uint64_t(hi) << 32) | lo;
#else
  return 0;
#endif
}
/* nfl::fastrandombytes (nfl/prng/fastrandombytes.h:12; lib/prng/fastrandombytes.cpp:21-37): rlen bytes of the process
 * stream.  Here: the next keystream of the process-wide sampler state, generated on the device. */
inline void fastrandombytes(unsigned char *r, unsigned long long rlen) {
  detail::sampler &s = detail::sampler::get();
  detail::check(nullptr, nflhip_random_bytes(detail::default_device().load(), r, size_t(rlen), s.key, s.next++), "fastrandombytes");
}
/* nfl::randombytes (nfl/prng/randombytes.h): OS entropy, what the reference keys its stream with */
inline void randombytes(unsigned char *x, unsigned long long xlen) {
  std::ifstream f("/dev/urandom", std::ios::binary);
  if (!f.read(reinterpret_cast<char *>(x), std::streamsize(xlen))) throw std::runtime_error("nfl(hip): /dev/urandom unreadable");
}

namespace detail {
// NARROW DRAWS (nflhip.h NFLHIP_DIST_NARROW, nflhip_gauss_set_draw_bits): poly(uniform) reads keystream lanes of the limb
// width and the Gaussian constructors 32-bit lanes -- the reference's maps from random bits to coefficients, the same
// distributions, at 1/4 - 1/2 of the ChaCha20 rounds.  NFL_HIP_WIDE_DRAWS=1 (read once) keeps the one-word-per-value rules.
inline bool narrow_draws() {
  static const bool v = !std::getenv("NFL_HIP_WIDE_DRAWS");
  return v;
}
inline int uniform_rule() { return narrow_draws() ? (NFLHIP_DIST_UNIFORM | NFLHIP_DIST_NARROW) : NFLHIP_DIST_UNIFORM; }
}  // namespace detail

/* FastGaussianNoise<in_class, out_class, _lu_depth>(sigma, security, samples, center) -- same constructor as
 * FastGaussianNoise.hpp:163-204.  The reference builds byte-indexed lookup tables over MPFR barriers; here the object
 * only carries the parameters and owns one cumulative table per device context (built on first use with the
 * reference's tail bound and bit precision, sampled by inversion on the GPU).  in_class / _lu_depth only tuned the
 * reference's lookup and are accepted for source compatibility. */
template <class in_class, class out_class, unsigned _lu_depth> class FastGaussianNoise {
 public:
  FastGaussianNoise(double sigma, unsigned int security, unsigned int samples, double center_d = 0, bool /*verbose*/ = false)
      : sigma_(sigma), security_(security), samples_(samples), center_(center_d), last_(nullptr) {
    static_assert(_lu_depth == 1 || _lu_depth == 2, "_lu_depth must be 1 or 2 (FastGaussianNoise.hpp:214)");
  }
  FastGaussianNoise(FastGaussianNoise const &) = delete;
  FastGaussianNoise &operator=(FastGaussianNoise const &) = delete;
  ~FastGaussianNoise() {
    // deferred draws refer to these tables (the reference samples inside the constructor, so a generator may well die
    // before the polynomials built from it are used): run every queue before the tables go
    try { detail::queue_registry::get().run_all(); } catch (...) {}
    // (the contexts are function-local statics and may already be gone when an object with static storage dies:
    // nflhip_gauss_destroy never dereferences its context argument)
    for (auto &kv : tables_) nflhip_gauss_destroy(nullptr, kv.second);
  }
  const nflhip_gauss *table(nflhip_ctx *ctx) {
    // (every random constructor of a loop comes through here: the context used last is answered from one atomic pointer to
    //  its map node -- node addresses are stable -- instead of a mutex and a map lookup per polynomial)
    const std::pair<nflhip_ctx *const, nflhip_gauss *> *hit = last_.load(std::memory_order_acquire);
    if (hit && hit->first == ctx) return hit->second;
    std::lock_guard<std::mutex> lk(mu_);
    auto it = tables_.find(ctx);
    if (it == tables_.end()) {
      nflhip_gauss *g = nullptr;
      detail::check(ctx, nflhip_gauss_create(ctx, &g, sigma_, security_, samples_, center_), "FastGaussianNoise");
      // the narrow draw (32 keystream bits per sample, the rest read lazily: nflhip.h nflhip_gauss_set_draw_bits) wherever the
      // sequence forms accept it -- the same exact inversion at half the ChaCha20 rounds
      if (detail::narrow_draws() && nflhip_degree(ctx) >= 16) nflhip_gauss_set_draw_bits(g, 32);
      it = tables_.emplace(ctx, g).first;
    }
    last_.store(&*it, std::memory_order_release);
    return it->second;
  }
  double sigma() const { return sigma_; }
  // FastGaussianNoise.hpp:477-595: rlen raw samples, negative values wrap into out_class exactly like the reference's
  // `(out_class)output`.  Runs on the device (a small private context only selects it); one keystream per call.
  void getNoise(out_class *const rand_data2out, uint64_t rlen) {
    nflhip_ctx *ctx = detail::context<uint64_t, 64, 1>::get();
    std::vector<int64_t> tmp(rlen);
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx, nflhip_gauss_noise(ctx, tmp.data(), rlen, table(ctx), s.key, s.next++), "getNoise");
    for (uint64_t i = 0; i < rlen; i++) rand_data2out[i] = out_class(tmp[i]);
  }

 private:
  double sigma_;
  unsigned security_, samples_;
  double center_;
  std::mutex mu_;
  std::map<nflhip_ctx *, nflhip_gauss *> tables_;
  std::atomic<const std::pair<nflhip_ctx *const, nflhip_gauss *> *> last_;
};

template <class in_class, class out_class, unsigned _lu_depth> struct gaussian {
  FastGaussianNoise<in_class, out_class, _lu_depth> *fg_prng;
  uint64_t amplifier;
  gaussian(FastGaussianNoise<in_class, out_class, _lu_depth> *prng) : fg_prng{prng}, amplifier{1} {}
  gaussian(FastGaussianNoise<in_class, out_class, _lu_depth> *prng, uint64_t amp) : fg_prng{prng}, amplifier{amp} {}
};

template <class T, size_t Degree, size_t NbModuli> class poly;
template <class T, size_t Degree, size_t NbModuli> class poly_p;
namespace detail {
inline std::atomic<bool> &deferred_flag();
}
inline void set_deferred(bool on) { detail::deferred_flag().store(on); }
namespace tests {
template <class P> class poly_tests_proxy;  // (poly.hpp:69-76) defined by the caller's test code, befriended below
}

namespace detail {
#ifdef NFL_HIP_REFERENCE_WORDS
static constexpr int dist_flags = NFLHIP_DIST_REFERENCE_WORDS;
#else
static constexpr int dist_flags = 0;
#endif

template <class P> struct lazy;

// Handle payloads come and go at the rate of the caller's temporaries (three per encryption of the LWE demo loop), and a
// general-purpose malloc / free pair per payload was the largest single item of the per-polynomial host cost (tools/hostprof:
// ~115 ns of ~185 per temporary).  std::allocate_shared with this allocator takes the control block + payload from a
// per-thread free list of fixed-size blocks instead; blocks freed on another thread simply join that thread's list.
template <class U> struct block_pool_alloc {
  typedef U value_type;
  block_pool_alloc() noexcept {}
  template <class V> block_pool_alloc(const block_pool_alloc<V> &) noexcept {}
  template <class V> struct rebind { typedef block_pool_alloc<V> other; };
  struct node { node *next; };
  struct list_t {
    node *head;
    size_t count;
    list_t() : head(nullptr), count(0) {}
    ~list_t() {
      gone() = true;   // (payloads released later in this thread's teardown -- static destructors -- go straight back to the heap)
      while (head) {
        node *n = head;
        head = n->next;
        ::operator delete(static_cast<void *>(n));
      }
      count = 0;
    }
  };
  static bool &gone() {   // trivially destructible, so it outlives the list it guards
    static thread_local bool g = false;
    return g;
  }
  static list_t &list() {
    static thread_local list_t l;
    return l;
  }
  U *allocate(size_t n) {
    if (n == 1 && sizeof(U) >= sizeof(node) && !gone()) {
      list_t &l = list();
      if (l.head) {
        node *x = l.head;
        l.head = x->next;
        --l.count;
        return reinterpret_cast<U *>(x);
      }
    }
    return static_cast<U *>(::operator new(n * sizeof(U)));
  }
  void deallocate(U *p, size_t n) noexcept {
    if (n == 1 && sizeof(U) >= sizeof(node) && !gone()) {
      list_t &l = list();
      if (l.count < (size_t(1) << 16)) {   // (bounded: a burst of 65 536 dead temporaries is kept, the rest goes back)
        node *x = reinterpret_cast<node *>(p);
        x->next = l.head;
        l.head = x;
        ++l.count;
        return;
      }
    }
    ::operator delete(static_cast<void *>(p));
  }
  template <class V> bool operator==(const block_pool_alloc<V> &) const noexcept { return true; }
  template <class V> bool operator!=(const block_pool_alloc<V> &) const noexcept { return false; }
};

// The shared payload of a poly_p handle (poly_p.hpp:11-204 keeps a std::shared_ptr<poly>): one polynomial that lives
// in HBM (`dev`), on the host (`host`), or both.  host_valid / dev_valid say which image holds the current value;
// neither valid = the zero polynomial (what poly_p() is) with nothing allocated yet.  Every device-side operation is
// enqueued on the context's stream, so the only synchronisation points are the device-to-host copies below.
// `queued`: the value is the result of operations that are still in the deferred queue (lazy<P> below); every access
// other than enqueueing more work runs the queue first (pending()).
template <class P> struct payload : std::enable_shared_from_this<payload<P>> {
  typedef typename P::value_type T;
  typedef context<T, P::degree, P::nmoduli> ctx_t;
  static constexpr size_t bytes = sizeof(T) * P::degree * P::nmoduli;
  P *host;
  void *dev;
  bool host_valid, dev_valid, queued;
  bool poisoned;  // the deferred operation that was to produce this value never ran (an earlier launch of its queue run failed)
  long qrefs;  // 1 while the deferred queue holds its (single) reference to this value, else 0: copy-on-write decisions look past it
  // levelling scratch of lazy<P>::flush (valid when `epoch` is the current flush): last level that writes / reads this value
  unsigned epoch;
  int wlev, rlev;
  int fw;  // scratch of lazy<P>::fuse (valid when `epoch` is the current flush): the recorded operation that last wrote this value
  unsigned pin_at;  // where the queue's reference to this payload sits in its pin list (valid while qrefs is set / during that run)
  // recording scratch of lazy<P>::record (valid when `rec_run` is the queue's current recording run): index of the last
  // recorded operation that writes / reads this value -- what lets a transform join the operation that produced its operand
  unsigned rec_run;
  int rec_w, rec_r;

  payload() : host(nullptr), dev(nullptr), host_valid(false), dev_valid(false), queued(false), poisoned(false), qrefs(0), epoch(0), wlev(-1), rlev(-1), fw(-1), pin_at(0), rec_run(0), rec_w(-1), rec_r(-1) {}
  payload(const payload &o) : std::enable_shared_from_this<payload<P>>(), host(nullptr), dev(nullptr), host_valid(false),
                              dev_valid(false), queued(false), poisoned(false), qrefs(0), epoch(0), wlev(-1), rlev(-1), fw(-1), pin_at(0), rec_run(0), rec_w(-1), rec_r(-1) {
    pending();
    o.usable();
    if (o.dev_valid) {  // stays on the device
      check(ctx(), nflhip_memcpy_d2d(ctx(), dev_wo(), o.dev, bytes, ctx_t::queue()), "poly_p copy");
    } else if (o.host_valid) {
      alloc_host();
      std::memcpy(host->data(), o.host->cdata(), bytes);
      host_valid = true;
    }
  }
  payload &operator=(const payload &) = delete;
  ~payload() {
    if (host) {
      host->~P();
      free(host);
    }
    ctx_t::release(dev);
  }
  static nflhip_ctx *ctx() { return ctx_t::get(); }
  static void pending() { lazy<P>::inst().flush(); }  // run whatever is still deferred
  void usable() const {  // reading a value whose producing operation never ran is an error, not stale HBM
    if (poisoned) throw std::runtime_error("nfl(hip): this polynomial's deferred operation did not run (an earlier operation of its queue failed)");
  }

  void alloc_host() {
    if (host) return;
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
    host = new (mem) P(uninitialized_t());
  }
  // the host image, current
  void to_host() {
    pending();
    usable();
    alloc_host();
    if (host_valid) return;
    if (dev_valid) {
      check(ctx(), nflhip_memcpy_d2h(ctx(), host->data(), dev, bytes, ctx_t::queue()), "poly_p download");
      check(ctx(), nflhip_stream_sync(ctx(), ctx_t::queue()), "poly_p download");
    } else {
      std::memset(static_cast<void *>(host->data()), 0, bytes);
    }
    host_valid = true;
  }
  P &host_rw() {  // the caller may write through the reference: the device image goes stale
    to_host();
    dev_valid = false;
    return *host;
  }
  P const &host_ro() {
    to_host();
    return *host;
  }
  P &host_wo() {  // about to be overwritten entirely on the host
    pending();
    alloc_host();
    host_valid = true;
    dev_valid = false;
    poisoned = false;
    return *host;
  }
  // the device image, current
  const void *dev_ro() {
    pending();
    return dev_ro_nf();
  }
  const void *dev_ro_nf() {  // (the queue's own form: never runs the queue)
    if (queued) return dev;  // produced by a deferred operation; its buffer is assigned when the queue runs
    usable();
    if (!dev) dev = ctx_t::acquire();
    if (!dev_valid) {
      if (host_valid) check(ctx(), nflhip_memcpy_h2d(ctx(), dev, host->cdata(), bytes, ctx_t::queue()), "poly_p upload");
      else check(ctx(), nflhip_memset_dev(ctx(), dev, 0, bytes, ctx_t::queue()), "poly_p zero");
      dev_valid = true;
    }
    return dev;
  }
  void *dev_rw() {  // in-place device operation
    dev_ro();
    host_valid = false;
    return dev;
  }
  void *dev_wo() {  // about to be overwritten entirely on the device
    pending();
    if (!dev) dev = ctx_t::acquire();
    dev_valid = true;
    host_valid = false;
    poisoned = false;
    return dev;
  }
};

// ---------------------------------------------------------------- deferred execution of per-polynomial operations
// One polynomial is 4 workgroups of work: a kernel launched for it runs for ~15 us on an otherwise empty GPU, and code
// written against the reference issues exactly such operations in loops (tests/nfllib_demo_main_op.cpp:26-58: three
// Gaussian polynomials, three transforms and two fused multiply-adds per encryption, one polynomial at a time).  So
// operations on resident handles are not launched when they are called: they are appended to a per-ring queue, and
// when a value is needed on the host (or the queue is long) the queue runs as a few BATCHED launches:
//   * operations are levelled by their data dependencies (read-after-write, write-after-read, write-after-write on
//     the shared payloads), so everything inside a level is independent;
//   * inside a level, operations with the same signature (same expression program, same distribution and parameters,
//     same transform) form a group; operands that are the same polynomial throughout (a key) split groups;
//   * results that have no buffer yet get CONSECUTIVE buffers (context::acquire_many), so a loop's temporaries form
//     dense arrays; a group is cut into runs whose operands advance by constant strides and every run is ONE launch of
//     the batch entry points (nflhip_eval_strided_dev, nflhip_sample[_gauss]_seq_dev, nflhip_ntt_fwd/inv_dev).
// Results are bit-identical to immediate execution (the random constructors keep the stream id they took when they
// were called).  nfl::set_deferred(false) (or -DNFL_HIP_EAGER) launches every operation at once instead.
inline std::atomic<bool> &deferred_flag() {
#ifdef NFL_HIP_EAGER
  static std::atomic<bool> f(false);
#else
  static std::atomic<bool> f(true);
#endif
  return f;
}

template <class P> struct lazy {
  typedef payload<P> pay_t;
  typedef typename pay_t::ctx_t ctx_t;
  typedef typename P::value_type T;
  typedef std::shared_ptr<pay_t> ptr_t;
  // K_FWD_FMA / K_FMA_INV / K_NOP are never recorded: a queue run rewrites recorded sequences into them (fuse())
  enum kind_t { K_EVAL = 0, K_NTT_FWD, K_NTT_INV, K_SAMPLE, K_GAUSS, K_FILL, K_FWD_FMA, K_FMA_INV, K_NOP };
  // One recorded operation: 72 bytes, trivially destructible.  The payloads it names are kept alive by ONE reference per
  // payload and queue run (`pins`), not one per mention -- a loop's temporaries are mentioned three times each.
  static constexpr int max_in = 4;  // expressions with more distinct handle operands are launched at once, not recorded
  struct op {
    pay_t *out;
    union {
      struct {
        pay_t *in[max_in];
        unsigned char code[NFLHIP_EXPR_MAX_LEN];
      } e;          // K_EVAL (the transforms use out only)
      struct {
        uint64_t p0, p1, sid;
        const nflhip_gauss *tab;
        int dist;
      } s;          // K_SAMPLE, K_GAUSS, K_FILL
      struct {
        pay_t *in[max_in];     // the key operands k0 [, k1] (same place as e.in: the levelling reads them through it)
        pay_t *out2;           // second result (out1 = NTT(x) * k1 + NTT(e1)), or nullptr
        uint64_t sid[3];       // stream ids of the Gaussian polynomials x, e0, e1
        uint32_t amp[3];       // their amplifiers
        const nflhip_gauss *tab;
      } f;          // K_FWD_FMA
    };
    unsigned char kind, nin, len;
    unsigned char post;   // 0, or K_NTT_FWD / K_NTT_INV: the result is transformed in place right after (a transform recorded on
                          // a value nothing had read since this operation produced it joins the operation instead of becoming a record)
  };
  light_lock mu;
  std::vector<op> q, running;  // recorded operations; the ones a queue run is working on (two buffers that swap: no regrowth)
  std::vector<ptr_t> pins, pins_running;  // the payloads they name, one reference each
  size_t launches, coalesced;  // statistics: launches issued / operations they carried
  unsigned rec_run_;           // the recording run: bumped whenever the queue is handed to a queue run (payload::rec_run)
  // records after which the queue runs by itself: long enough for wide launches, short enough that the device works on
  // one part of a loop while the host records the next (NFL_HIP_QUEUE_LIMIT overrides, for experiments).  Measured on the
  // LWE demo loop (profiles/r02_late_queue_limit.txt): 2 048 iterations 614 k / 738 k / 1.04 M / 861 k encryptions/s with
  // 2 048 / 4 096 / 8 192 / 16 384 records, 16 384 iterations 1.09 M / 1.21 M / 1.16 M / 1.16 M with 4 096 ... 32 768.
  static size_t max_queue() {
    static const size_t v = getenv("NFL_HIP_QUEUE_LIMIT") ? size_t(atol(getenv("NFL_HIP_QUEUE_LIMIT"))) : 8192;
    return v ? v : 1;
  }
  // (Round 5 measured two run-length policies for SHORT loops and dropped both -- LWE demo loop, encryptions/s against the fixed
  //  length: a loop's first two runs at HALF length: 2 048 iterations 1.68 -> 1.85 M, but 1 024: 1.60 -> 1.39 M, 4 096: 2.12 ->
  //  2.08 M, 16 384: 2.48 -> 2.43 M; its first run at THREE QUARTERS: 2 048: 1.67 -> 1.88 M, 4 096: 2.10 -> 2.16 M, but 1 536:
  //  1.67 -> 1.60 M, 3 072: 2.01 -> 1.86 M.  Any fixed threshold moves the sawtooth, it does not remove it: what a short loop
  //  pays beyond its recording is the device work that starts when the loop ends.  The host records an encryption in 0.36 us
  //  and a run costs it ~36 us, so 2.78 M/s is the ceiling of ANY policy at 2 048 iterations: profiles/r05_short_loops.txt.)

  // NFL_HIP_EARLY_RUN=1: from 1 024 records on, every 512 records the queue asks whether the stream is idle
  // (nflhip_stream_idle: one hipStreamQuery) and runs at once if it is.  Off by default: measured on the LWE demo's
  // 2 048-iteration loop it changes nothing (profiles/r02_late_early_run.txt); results are identical either way.
  static bool early_run() {
    static const bool v = getenv("NFL_HIP_EARLY_RUN") && atoi(getenv("NFL_HIP_EARLY_RUN")) != 0;
    return v;
  }
  lazy() : launches(0), coalesced(0), rec_run_(1), small_(nullptr), small_cap_(0), fused_fwd(0), fused_inv(0) {
    ctx_t::inst();  // (the context is constructed first, so it is destroyed last)
    alive() = true;
    queue_registry::get().add(&lazy::run_if_alive);
  }
  ~lazy() {
    alive() = false;
    if (small_ && ctx_t::alive()) nflhip_free(ctx_t::get(), small_);
  }
  static bool &alive() {
    static bool a = false;
    return a;
  }
  static void run_if_alive() {  // what queue_registry calls (possibly while the program's statics are being destroyed)
    if (alive() && ctx_t::alive()) inst().flush();
  }
  static lazy &inst() {
    static lazy l;
    return l;
  }
  // whether two recorded operations may share a launch: everything a launch takes from its first member
  static bool same_signature(const op &a, const op &b) {
    if (a.kind != b.kind || a.post != b.post) return false;
    if (a.kind == K_EVAL) return a.len == b.len && a.nin == b.nin && std::memcmp(a.e.code, b.e.code, a.len) == 0;
    if (a.kind == K_SAMPLE || a.kind == K_GAUSS) return a.s.dist == b.s.dist && a.s.p0 == b.s.p0 && a.s.p1 == b.s.p1 && a.s.tab == b.s.tab;
    if (a.kind == K_FILL) return a.s.sid == b.s.sid;
    if (a.kind == K_FWD_FMA)
      return a.nin == b.nin && a.f.tab == b.f.tab && a.f.amp[0] == b.f.amp[0] && a.f.amp[1] == b.f.amp[1] && a.f.amp[2] == b.f.amp[2];
    if (a.kind == K_FMA_INV) return a.e.code[0] == b.e.code[0];
    return true;
  }
  // ---- transform fusion.  Code written against the reference transforms, combines, transforms back:
  //        u.ntt_pow_phi(); e.ntt_pow_phi(); r = u * key + e;          out = rb - ra * s; out.invntt_pow_invphi();
  // (tests/nfllib_demo_main_op.cpp:26-58).  When this context runs such a sequence as ONE kernel (nflhip_has_fused_kernels),
  // a queue run rewrites what it recorded before it levels it:
  //   * K_GAUSS x, K_NTT_FWD x, K_GAUSS e, K_NTT_FWD e, K_EVAL r = x * k + e (either operand order; a second K_EVAL on the
  //     same x with its own k, e joins) -> K_FWD_FMA, provided nothing else reads the sampled or transformed x / e and
  //     their handles are gone (the queue holds the last reference): those polynomials then never exist in HBM -- the
  //     samplers write one byte per coefficient (nflhip_sample_gauss_small_seq_dev) and the kernel transforms in registers;
  //   * K_EVAL t = c +- a * b, K_NTT_INV t with nothing reading t in between -> K_FMA_INV.
  // Results are bit-identical to the operator-by-operator run.  NFL_HIP_NO_FUSION=1 switches the rewriting off.
  static bool fusion_on() {
    static const bool v = !getenv("NFL_HIP_NO_FUSION") && nflhip_has_fused_kernels(ctx_t::get()) != 0;
    return v;
  }
  // c +- a * b as a 5-byte postfix program over three distinct operands: {a, b, c, subtract}, or false.  A record that
  // carries a joined transform (op::post, join_transform below) is NOT that expression: its result is the transformed
  // value, and a rewrite that took only the expression would drop the transform.  The one place that models a joined
  // transform -- the expression followed by its own inverse transform -- asks for it by name (`joined`).
  static bool parse_fma(const op &o, int &a, int &b, int &c, bool &sub, unsigned char joined = 0) {
    if (o.kind != K_EVAL || o.len != 5 || o.nin != 3 || o.post != joined) return false;
    const unsigned char *q = o.e.code;
    if (q[0] < 3 && q[1] < 3 && q[2] == NFLHIP_EXPR_MUL && q[3] < 3 && q[4] == NFLHIP_EXPR_ADD) {            // a b * c +
      a = q[0]; b = q[1]; c = q[3]; sub = false;
    } else if (q[0] < 3 && q[1] < 3 && q[2] < 3 && q[3] == NFLHIP_EXPR_MUL && (q[4] == NFLHIP_EXPR_ADD || q[4] == NFLHIP_EXPR_SUB)) {  // c a b * +-
      c = q[0]; a = q[1]; b = q[2]; sub = q[4] == NFLHIP_EXPR_SUB;
    } else {
      return false;
    }
    return a != b && a != c && b != c;
  }
  static bool mentions(const op &o, const pay_t *p) {
    if (o.kind == K_NOP) return false;
    if (o.out == p || (o.kind == K_FWD_FMA && o.f.out2 == p)) return true;
    if (o.kind == K_EVAL || o.kind == K_FWD_FMA || o.kind == K_FMA_INV)
      for (int j = 0; j < o.nin; ++j)
        if (o.e.in[j] == p) return true;
    return false;
  }
  // compact Gaussian polynomials of a fused run live in ONE grow-only device buffer (every consumer is on the queue's stream)
  void *small_;
  size_t small_cap_;
  void *small_buffer(size_t bytes) {
    if (bytes > small_cap_) {
      nflhip_ctx *c = ctx_t::get();
      if (small_) {
        check(c, nflhip_stream_sync(c, ctx_t::queue()), "compact sampler buffer");   // (its last readers)
        nflhip_free(c, small_);
        small_ = nullptr;
        small_cap_ = 0;
      }
      size_t cap = size_t(1) << 20;
      while (cap < bytes) cap *= 2;
      check(c, nflhip_malloc(c, &small_, cap), "compact sampler buffer");
      small_cap_ = cap;
    }
    return small_;
  }
  // the narrowest compact format that holds every sample of `tab` times `amp` (NFLHIP_FMT_I8 / I16 / I32; 99 = none)
  static int small_format(const nflhip_gauss *tab, uint32_t amp) {
    struct last_t { const nflhip_gauss *tab; uint64_t mag; };
    static thread_local last_t last = {nullptr, 0};
    if (last.tab != tab) {
      long long x_min = 0;
      size_t entries = 0;
      check(ctx_t::get(), nflhip_gauss_info(tab, &x_min, &entries, nullptr, nullptr, nullptr, nullptr), "gaussian table");
      const long long hi = x_min + (long long)entries - 1;
      last.tab = tab;
      last.mag = uint64_t(std::max(x_min < 0 ? -x_min : x_min, hi < 0 ? -hi : hi));
    }
    const uint64_t v = last.mag * uint64_t(amp);
    return v <= 127 ? NFLHIP_FMT_I8 : v <= 32767 ? NFLHIP_FMT_I16 : v <= 2147483647ull ? NFLHIP_FMT_I32 : 99;
  }
  size_t fused_fwd, fused_inv;  // statistics: sequences rewritten so far
  void fuse(std::vector<op> &ops, const std::vector<ptr_t> &held, unsigned ep) {
    const int n = int(ops.size());
    if (n < 2 || !fusion_on()) return;
    // definitions: prev[i] = the operation that wrote ops[i].out before i (what an in-place transform reads), def[i][j] =
    // the one that wrote input j of an expression; uses[d] = reads of the value operation d wrote; -1 = from before this run
    std::vector<int> prev(size_t(n), -1), uses(size_t(n), 0), def(size_t(n) * 3, -1);
    auto touch = [ep](pay_t *p) {
      if (p->epoch != ep) {
        p->epoch = ep;
        p->wlev = p->rlev = -1;
        p->fw = -1;
      }
    };
    bool any_fwd = false, any_inv = false;
    for (int i = 0; i < n; ++i) {
      op &o = ops[size_t(i)];
      if (o.kind == K_EVAL)
        for (int j = 0; j < o.nin; ++j) {
          touch(o.e.in[j]);
          const int d = o.e.in[j]->fw;
          if (j < 3) def[size_t(i) * 3 + size_t(j)] = d;
          if (d >= 0) ++uses[size_t(d)];
        }
      touch(o.out);
      prev[size_t(i)] = o.out->fw;
      if ((o.kind == K_NTT_FWD || o.kind == K_NTT_INV) && o.out->fw >= 0) ++uses[size_t(o.out->fw)];
      o.out->fw = i;
      any_fwd |= o.kind == K_NTT_FWD || o.post == K_NTT_FWD;
      any_inv |= o.kind == K_NTT_INV || o.post == K_NTT_INV;
    }
    // the value an operation wrote is still its payload's at the end of the run: only fusable away when no handle is left
    auto dead_after = [&](int d) {
      pay_t *p = ops[size_t(d)].out;
      return p->fw != d || held[p->pin_at].use_count() == 1;   // (the queue's pin is the last reference)
    };
    // a sampled-and-transformed polynomial nobody else sees: -> index of its K_GAUSS record, or -1
    auto gauss_chain = [&](int dn, int want_uses) {
      if (dn >= 0 && ops[size_t(dn)].kind == K_GAUSS && ops[size_t(dn)].post == K_NTT_FWD) {   // the transform joined its constructor's record
        const op &g = ops[size_t(dn)];
        if (uses[size_t(dn)] != want_uses || !dead_after(dn) || (g.s.p1 >> 32) != 0) return -1;
        return small_format(g.s.tab, uint32_t(g.s.p1)) > NFLHIP_FMT_I32 ? -1 : dn;
      }
      if (dn < 0 || ops[size_t(dn)].kind != K_NTT_FWD || uses[size_t(dn)] != want_uses || !dead_after(dn)) return -1;
      const int g = prev[size_t(dn)];
      if (g < 0 || ops[size_t(g)].kind != K_GAUSS || ops[size_t(g)].post != 0 || uses[size_t(g)] != 1 || (ops[size_t(g)].s.p1 >> 32) != 0)
        return -1;   // (post: the constructor's record already carries one transform; this would be the second)
      if (small_format(ops[size_t(g)].s.tab, uint32_t(ops[size_t(g)].s.p1)) > NFLHIP_FMT_I32) return -1;
      return g;
    };
    if (any_inv)
      for (int i = 0; i < n; ++i) {
        op &t = ops[size_t(i)];
        if (t.kind == K_EVAL && t.post == K_NTT_INV) {   // the transform joined the expression's record: rewrite in place
          int a, b, c;
          bool sub;
          if (!parse_fma(t, a, b, c, sub, K_NTT_INV)) continue;
          pay_t *pc = t.e.in[c], *pa = t.e.in[a], *pb = t.e.in[b];
          t.kind = K_FMA_INV;
          t.post = 0;
          t.nin = 3;
          t.len = 1;
          t.e.in[0] = pc;
          t.e.in[1] = pa;
          t.e.in[2] = pb;
          t.e.code[0] = sub ? 1 : 0;
          ++fused_inv;
          continue;
        }
        if (t.kind != K_NTT_INV) continue;
        const int d = prev[size_t(i)];
        int a, b, c;
        bool sub;
        if (d < 0 || i - d > 4 || uses[size_t(d)] != 1 || !parse_fma(ops[size_t(d)], a, b, c, sub)) continue;
        op &e = ops[size_t(d)];
        bool clean = true;   // nothing between the two rewrites an operand (the fused operation reads them at i, not at d)
        for (int k = d + 1; k < i && clean; ++k)
          clean = ops[size_t(k)].kind == K_NOP || (ops[size_t(k)].out != e.e.in[0] && ops[size_t(k)].out != e.e.in[1] && ops[size_t(k)].out != e.e.in[2]);
        if (!clean) continue;
        pay_t *pc = e.e.in[c], *pa = e.e.in[a], *pb = e.e.in[b];
        t.kind = K_FMA_INV;
        t.nin = 3;
        t.len = 1;
        t.e.in[0] = pc;
        t.e.in[1] = pa;
        t.e.in[2] = pb;
        t.e.code[0] = sub ? 1 : 0;
        e.kind = K_NOP;
        ++fused_inv;
      }
    if (!any_fwd) return;
    // forward: candidates per transformed x (an expression names it once; a second expression on the same x joins)
    struct cand { int i, xs, ks, es, gx, ge; };
    std::vector<cand> cands;
    for (int i = 0; i < n; ++i) {
      int a, b, c;
      bool sub;
      if (!parse_fma(ops[size_t(i)], a, b, c, sub) || sub) continue;
      for (int turn = 0; turn < 2; ++turn) {
        const int xs = turn ? b : a, ks = turn ? a : b;
        const int dx = def[size_t(i) * 3 + size_t(xs)], de = def[size_t(i) * 3 + size_t(c)];
        if (dx < 0 || de < 0 || dx == de) continue;
        const int ux = uses[size_t(dx)];
        if (ux != 1 && ux != 2) continue;
        const int gx = gauss_chain(dx, ux), ge = gauss_chain(de, 1);
        if (gx < 0 || ge < 0 || ops[size_t(gx)].s.tab != ops[size_t(ge)].s.tab) continue;
        cands.push_back(cand{i, xs, ks, c, gx, ge});
        break;
      }
    }
    for (size_t q = 0; q < cands.size(); ++q) {
      const cand &c0 = cands[q];
      if (c0.i < 0) continue;
      const int dx = def[size_t(c0.i) * 3 + size_t(c0.xs)];
      const cand *c1 = nullptr;
      if (uses[size_t(dx)] == 2) {   // the other reader of NTT(x) must be a candidate too, close by, and independent of this one
        for (size_t r = q + 1; r < cands.size() && !c1; ++r)
          if (cands[r].i >= 0 && def[size_t(cands[r].i) * 3 + size_t(cands[r].xs)] == dx) c1 = &cands[r];
        if (!c1 || c1->i - c0.i > 4) continue;
        const op &e0 = ops[size_t(c0.i)], &e1 = ops[size_t(c1->i)];
        bool clean = e1.e.in[c1->ks] != e0.out && e1.out != e0.out;   // (the fused operation writes both results at e1's place)
        for (int k = c0.i + 1; k < c1->i && clean; ++k)
          clean = !mentions(ops[size_t(k)], e0.out) && (ops[size_t(k)].kind == K_NOP || ops[size_t(k)].out != e0.e.in[c0.ks]);
        if (!clean) continue;
      }
      const op e0 = ops[size_t(c0.i)];
      op &t = ops[size_t(c1 ? c1->i : c0.i)];
      const op e1 = t;
      const op &gx = ops[size_t(c0.gx)], &g0 = ops[size_t(c0.ge)];
      t.kind = K_FWD_FMA;
      t.out = e0.out;
      t.nin = c1 ? 2 : 1;
      t.len = 0;
      t.f.in[0] = e0.e.in[c0.ks];
      t.f.in[1] = c1 ? e1.e.in[c1->ks] : nullptr;
      t.f.in[2] = t.f.in[3] = nullptr;
      t.f.out2 = c1 ? e1.out : nullptr;
      t.f.tab = gx.s.tab;
      t.f.sid[0] = gx.s.sid;
      t.f.amp[0] = uint32_t(gx.s.p1);
      t.f.sid[1] = g0.s.sid;
      t.f.amp[1] = uint32_t(g0.s.p1);
      t.f.sid[2] = c1 ? ops[size_t(c1->ge)].s.sid : 0;
      t.f.amp[2] = c1 ? uint32_t(ops[size_t(c1->ge)].s.p1) : 0;
      // the records the fused operation stands for
      const int gone[] = {c0.gx, dx, c0.ge, def[size_t(c0.i) * 3 + size_t(c0.es)], c1 ? c0.i : -1, c1 ? c1->ge : -1,
                          c1 ? def[size_t(c1->i) * 3 + size_t(c1->es)] : -1};
      for (int g : gone)
        if (g >= 0) ops[size_t(g)].kind = K_NOP;
      if (c1) const_cast<cand *>(c1)->i = -1;
      ++fused_fwd;
    }
  }
  // whether this ring's operations can be deferred at all: dense chunks, vectors of 16 bytes, sequence samplers
  static bool usable() {
    return deferred_flag().load(std::memory_order_relaxed) && ctx_t::chunk_bytes == ctx_t::poly_bytes && P::degree >= 8 &&
           P::degree * sizeof(T) >= 16;
  }
  // ascending order for addresses that usually are `period` interleaved ascending sequences already (a loop body that
  // transforms u, e1, e2 -- each kind a dense array of its own -- yields u0 e1_0 e2_0 u1 e1_1 e2_1 ...): merged in O(n)
  static void sort_interleaved(std::vector<char *> &v) {
    if (std::is_sorted(v.begin(), v.end())) return;
    for (size_t period = 2; period <= 8 && period * 2 <= v.size(); ++period) {
      bool ok = true;
      for (size_t i = period; i < v.size() && ok; ++i) ok = !(v[i] < v[i - period]);
      if (!ok) continue;
      std::vector<char *> out;
      out.reserve(v.size());
      for (size_t r = 0; r < period; ++r) {
        const size_t mid = out.size();
        for (size_t i = r; i < v.size(); i += period) out.push_back(v[i]);
        std::inplace_merge(out.begin(), out.begin() + ptrdiff_t(mid), out.end());
      }
      v.swap(out);
      return;
    }
    std::sort(v.begin(), v.end());
  }
  // the queue's reference to a payload (taken the first time a queue run's operations mention it)
  void pin(pay_t *p) {
    if (!p->qrefs) {
      p->pin_at = unsigned(pins.size());
      pins.push_back(p->shared_from_this());
      p->qrefs = 1;
    }
  }
  pay_t *rec_tag(pay_t *p) {
    if (p->rec_run != rec_run_) {
      p->rec_run = rec_run_;
      p->rec_w = p->rec_r = -1;
    }
    return p;
  }
  // A transform recorded on a value that a Gaussian constructor or an expression of THIS recording run produced and that
  // nothing has read since does not become a record of its own: the producing operation notes "then transform in place"
  // (op::post).  The reference's loops are written that way -- poly_p u{gaussian}; u.ntt_pow_phi();  out = rb - ra * s;
  // out.invntt_pow_invphi(); -- and every record costs the host the same whatever it stands for.  Results are those of the
  // separate records; NFL_HIP_NO_FUSION=1 switches this off together with the transform fusion.
  static bool joining_on() {
    static const bool v = !getenv("NFL_HIP_NO_FUSION");
    return v;
  }
  bool join_transform(pay_t *p, int kind) {
    if (!joining_on()) return false;
    std::lock_guard<light_lock> lk(mu);
    if (p->rec_run != rec_run_ || p->rec_w < 0 || p->rec_r > p->rec_w || p->poisoned) return false;
    op &t = q[size_t(p->rec_w)];
    if (t.post || t.out != p || !((t.kind == K_GAUSS && kind == K_NTT_FWD) || t.kind == K_EVAL)) return false;
    t.post = static_cast<unsigned char>(kind);
    return true;
  }
  // `fill(op &)` writes the record in place, in the queue; the payloads it names are pinned here
  template <class F> void record(F fill) {
    std::lock_guard<light_lock> lk(mu);
    q.emplace_back();
    op &o = q.back();
    o.nin = 0;
    o.len = 0;
    o.post = 0;
    try {  // inputs must hold a device value (or be produced by the queue) before the operation counts as recorded
      fill(o);
      for (int j = 0; j < o.nin; ++j) o.e.in[j]->dev_ro_nf();
      if (o.kind == K_NTT_FWD || o.kind == K_NTT_INV) o.out->dev_ro_nf();
      pin(o.out);
      for (int j = 0; j < o.nin; ++j) pin(o.e.in[j]);
    } catch (...) {
      q.pop_back();
      throw;
    }
    const int at = int(q.size()) - 1;
    for (int j = 0; j < o.nin; ++j) rec_tag(o.e.in[j])->rec_r = at;
    rec_tag(o.out)->rec_w = at;
    o.out->queued = true;
    o.out->dev_valid = true;
    o.out->host_valid = false;
    if (o.kind != K_NTT_FWD && o.kind != K_NTT_INV) o.out->poisoned = false;  // overwritten entirely
    if (q.size() >= max_queue()) flush();
    else if (early_run() && q.size() >= 1024 && q.size() % 512 == 0) {
      // a loop shorter than the queue: do not let the device sit idle until the loop's end
      int idle = 0;
      if (nflhip_stream_idle(ctx_t::get(), ctx_t::queue(), &idle) == NFLHIP_OK && idle) flush();
    }
  }
  void flush() {
    std::lock_guard<light_lock> lk(mu);
    if (q.empty()) return;
    std::vector<op> local;  // (a queue run started from inside another one: cannot happen today, costs nothing to allow)
    std::vector<ptr_t> local_pins;
    const bool outer = running.empty() && pins_running.empty();
    std::vector<op> &ops = outer ? running : local;
    std::vector<ptr_t> &held = outer ? pins_running : local_pins;
    ops.swap(q);
    held.swap(pins);
    ++rec_run_;   // (what is recorded from now on cannot join operations of this run)
    if (q.capacity() < ops.capacity()) q.reserve(ops.capacity());
    for (auto &p : held) p->qrefs = 0;  // (operations recorded from now on belong to the next run and pin again)
    std::vector<unsigned char> launched(ops.size(), 0);
    struct done_guard {  // whatever happens, the payloads stop claiming a queued value, and the run's references go
      std::vector<op> &o;
      std::vector<ptr_t> &h;
      std::vector<unsigned char> &launched;
      bool complete;
      ~done_guard() {
        for (auto &x : o) x.out->queued = false;
        if (!complete)  // a launch failed: what was never launched holds no value -- later accesses throw (payload::usable)
          for (size_t i = 0; i < o.size(); ++i)
            if (!launched[i]) {
              o[i].out->dev_valid = false;
              o[i].out->poisoned = true;
            }
        o.clear();
        if (ctx_t::alive()) {   // the temporaries' buffers go back to the pool one by one: its lock is taken once for all of them
          std::lock_guard<light_lock> pool(ctx_t::inst().mu);
          h.clear();
        } else {
          h.clear();
        }
      }
    } guard{ops, held, launched, false};
    // ---- 1. levels (the last writing / reading level of a value is kept in its payload, tagged with this flush's epoch)
    static unsigned epoch_counter = 0;
    const unsigned ep = ++epoch_counter;
    auto touch = [ep](pay_t *p) {
      if (p->epoch != ep) {
        p->epoch = ep;
        p->wlev = p->rlev = -1;
      }
    };
    fuse(ops, held, ep);   // (tags the payloads it sees with this flush's epoch: wlev / rlev start at -1 either way)
    std::vector<int> lvl(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); ++i) {
      op &o = ops[i];
      if (o.kind == K_NOP) {   // its work moved into a fused operation
        lvl[i] = -1;
        launched[i] = 1;
        continue;
      }
      int L = 0;
      for (int j = 0; j < o.nin; ++j) {
        touch(o.e.in[j]);
        L = std::max(L, o.e.in[j]->wlev + 1);
      }
      touch(o.out);
      L = std::max(L, std::max(o.out->wlev, o.out->rlev) + 1);
      pay_t *out2 = o.kind == K_FWD_FMA ? o.f.out2 : nullptr;
      if (out2) {
        touch(out2);
        L = std::max(L, std::max(out2->wlev, out2->rlev) + 1);
      }
      lvl[i] = L;
      o.out->wlev = L;
      if (out2) out2->wlev = L;
      for (int j = 0; j < o.nin; ++j) o.e.in[j]->rlev = std::max(o.e.in[j]->rlev, L);
      if (o.kind == K_NTT_FWD || o.kind == K_NTT_INV) o.out->rlev = std::max(o.out->rlev, L);
    }
    // ---- 2. groups: (level, signature) -> operations in program order.  A loop produces a handful of distinct
    // signatures, so a linear table of the ones seen (64-bit FNV-1a of the fields, plus the level) beats a map.
    struct gkey { int level; uint64_t hash; };
    std::vector<gkey> keys;
    std::vector<std::vector<size_t>> members;
    auto mix = [](uint64_t h, uint64_t v) {  // (one multiply-xorshift round per 64-bit field: the signatures are a few words)
      h = (h ^ v) * 0x9E3779B97F4A7C15ull;
      return h ^ (h >> 29);
    };
    static_assert(NFLHIP_EXPR_MAX_LEN <= 24, "the program is hashed as three words");
    for (size_t i = 0; i < ops.size(); ++i) {
      const op &o = ops[i];
      if (o.kind == K_NOP) continue;
      uint64_t h = mix(0xcbf29ce484222325ull, uint64_t(o.kind) | (uint64_t(o.post) << 8));
      if (o.kind == K_FWD_FMA) {
        h = mix(mix(mix(h, uint64_t(reinterpret_cast<uintptr_t>(o.f.tab))), (uint64_t(o.f.amp[0]) << 32) | o.f.amp[1]), (uint64_t(o.f.amp[2]) << 8) | o.nin);
      } else if (o.kind == K_FMA_INV) {
        h = mix(h, o.e.code[0]);
      } else if (o.kind == K_EVAL) {
        uint64_t w[3] = {0, 0, 0};
        std::memcpy(w, o.e.code, size_t(o.len));
        h = mix(mix(mix(mix(h, w[0]), w[1]), w[2]), (uint64_t(o.len) << 8) | uint64_t(o.nin));
      } else if (o.kind == K_SAMPLE || o.kind == K_GAUSS || o.kind == K_FILL) {
        h = mix(mix(mix(mix(h, uint64_t(o.s.dist)), o.s.p0), o.s.p1), uint64_t(reinterpret_cast<uintptr_t>(o.s.tab)));
        if (o.kind == K_FILL) h = mix(h, o.s.sid);
      }
      size_t g = keys.size();
      for (size_t k = keys.size(); k-- > 0;)   // (recent groups first: neighbouring operations repeat)
        if (keys[k].level == lvl[i] && keys[k].hash == h && same_signature(ops[members[k][0]], o)) { g = k; break; }
      if (g == keys.size()) {
        keys.push_back(gkey{lvl[i], h});
        members.emplace_back();
        members.back().reserve(ops.size() / 4 + 1);
      }
      members[g].push_back(i);
    }
    // groups run level by level (inside a level the order is irrelevant: they are independent)
    std::vector<size_t> order(keys.size());
    for (size_t g = 0; g < order.size(); ++g) order[g] = g;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return keys[x].level < keys[y].level; });
    nflhip_ctx *ctx = ctx_t::get();
    void *st = ctx_t::queue();
    struct { unsigned char key[32]; } smp;  // the key as it is NOW: set_sampler_key runs the queues before it changes it
    detail::sampler::get().copy_key(smp.key);
    static const bool trace = getenv("NFL_HIP_TRACE_DEFERRED") != nullptr;
    // in-place transforms of the results of `idx` (mutually independent): by address, so that neighbours become one dense batch
    auto launch_transforms = [&](const std::vector<size_t> &idx, int tkind) {
      std::vector<char *> ptr;
      ptr.reserve(idx.size());
      for (size_t i : idx) ptr.push_back(static_cast<char *>(ops[i].out->dev));
      sort_interleaved(ptr);
      for (size_t a = 0; a < ptr.size();) {
        size_t b = a + 1;
        while (b < ptr.size() && ptr[b] == ptr[b - 1] + ctx_t::chunk_bytes) ++b;
        check(ctx, tkind == K_NTT_FWD ? nflhip_ntt_fwd_dev(ctx, ptr[a], b - a, st) : nflhip_ntt_inv_dev(ctx, ptr[a], b - a, st),
              "deferred transform");
        ++launches;
        coalesced += b - a;
        a = b;
      }
    };
    // (operations that carry a joined transform count as launched only once it has been issued)
    auto finish_post = [&](const std::vector<size_t> &idx) {
      const int post = ops[idx[0]].post;
      if (!post) return;
      for (size_t i : idx) launched[i] = 0;
      launch_transforms(idx, post);
      for (size_t i : idx) launched[i] = 1;
    };
    for (size_t gi : order) {
      std::vector<size_t> &idx = members[gi];
      const int kind = ops[idx[0]].kind;
      const size_t launches_before = launches;
      struct tracer {
        bool on; int level, kind; size_t n; const size_t &now; size_t before;
        ~tracer() { if (on) std::fprintf(stderr, "nfl(hip) deferred: level %d kind %d: %zu operations -> %zu launches\n", level, kind, n, now - before); }
      } tr{trace, keys[gi].level, kind, idx.size(), launches, launches_before};
      if ((kind == K_SAMPLE || kind == K_GAUSS) && idx.size() >= 4) {
        // A loop body that draws several polynomials of one distribution (e1, e2 of an encryption) interleaves their
        // stream ids: k+1, k+2, k+4, k+5, ...  Find the period of the id differences and regroup the operations into
        // that many arithmetic progressions, each of which then gets its own dense array of buffers and is one launch.
        for (size_t period = 2; period <= 8 && period * 2 <= idx.size(); ++period) {
          bool periodic = true, constant = true;
          for (size_t i = 0; i + 1 < idx.size() && periodic; ++i) {
            const uint64_t d = ops[idx[i + 1]].s.sid - ops[idx[i]].s.sid;
            if (i + 1 + period < idx.size()) periodic = d == ops[idx[i + 1 + period]].s.sid - ops[idx[i + period]].s.sid;
            constant &= d == ops[idx[1]].s.sid - ops[idx[0]].s.sid;
          }
          if (constant) break;
          if (periodic) {
            std::vector<size_t> re;
            for (size_t r = 0; r < period; ++r)
              for (size_t i = r; i < idx.size(); i += period) re.push_back(idx[i]);
            idx.swap(re);
            break;
          }
        }
      }
      // ---- 3. buffers for results that have none yet: consecutive, in program order
      std::vector<size_t> need;
      for (size_t i : idx)
        if (!ops[i].out->dev) need.push_back(i);
      if (!need.empty()) {
        std::vector<void *> bufs(need.size());
        ctx_t::acquire_many(need.size(), bufs.data());
        for (size_t k = 0; k < need.size(); ++k) ops[need[k]].out->dev = bufs[k];
      }
      if (kind == K_FWD_FMA) {   // the second results: a dense array of their own
        need.clear();
        for (size_t i : idx)
          if (ops[i].f.out2 && !ops[i].f.out2->dev) need.push_back(i);
        if (!need.empty()) {
          std::vector<void *> bufs(need.size());
          ctx_t::acquire_many(need.size(), bufs.data());
          for (size_t k = 0; k < need.size(); ++k) ops[need[k]].f.out2->dev = bufs[k];
        }
      }
      if (kind == K_NTT_FWD || kind == K_NTT_INV) {
        launch_transforms(idx, kind);
        for (size_t i : idx) launched[i] = 1;
        continue;
      }
      if (kind == K_FILL) {
        for (size_t i : idx) {
          check(ctx, nflhip_fill_uniform_dev(ctx, ops[i].out->dev, 0, 1, ops[i].s.sid, 0, st), "deferred set(uniform)");
          launched[i] = 1;
          ++launches;
          ++coalesced;
        }
        continue;
      }
      if (kind == K_SAMPLE || kind == K_GAUSS) {
        // program order; a run = consecutive buffers + stream ids in arithmetic progression
        for (size_t a = 0; a < idx.size();) {
          const op &o0 = ops[idx[a]];
          size_t b = a + 1;
          uint64_t stride = 0;
          while (b < idx.size()) {
            const op &prev = ops[idx[b - 1]], &cur = ops[idx[b]];
            if (static_cast<char *>(cur.out->dev) != static_cast<char *>(prev.out->dev) + ctx_t::chunk_bytes) break;
            const uint64_t d = cur.s.sid - prev.s.sid;
            if (b == a + 1) stride = d;
            else if (d != stride) break;
            ++b;
          }
          const size_t cnt = b - a;
          if (kind == K_SAMPLE)
            check(ctx, cnt == 1 ? nflhip_sample_dev(ctx, o0.out->dev, 0, 1, o0.s.dist, o0.s.p0, o0.s.p1, smp.key, o0.s.sid, st)
                                : nflhip_sample_seq_dev(ctx, o0.out->dev, cnt, o0.s.dist, o0.s.p0, o0.s.p1, smp.key, o0.s.sid, stride, st),
                  "deferred random constructor");
          else
            check(ctx, cnt == 1 ? nflhip_sample_gauss_dev(ctx, o0.out->dev, 0, 1, o0.s.tab, o0.s.p1, smp.key, o0.s.sid, st)
                                : nflhip_sample_gauss_seq_dev(ctx, o0.out->dev, cnt, o0.s.tab, o0.s.p1, smp.key, o0.s.sid, stride, st),
                  "deferred set(gaussian)");
          for (size_t k = a; k < b; ++k) launched[idx[k]] = 1;
          ++launches;
          coalesced += cnt;
          a = b;
        }
        finish_post(idx);
        continue;
      }
      // ---- K_EVAL (and the fused kinds, whose operands sit in the same slots): operands that are one polynomial for
      // (almost) the whole group split it; then stride runs
      const int nin = ops[idx[0]].nin;
      // a "key" slot holds one of a few polynomials throughout the group (at most 8, and at most every eighth operation a
      // new one); each combination of keys becomes its own sub-group, whose other operands then advance by strides
      struct subgroup { const pay_t *key[NFLHIP_EXPR_MAX_OPERANDS]; std::vector<size_t> idx; };
      std::vector<subgroup> sub;
      {
        bool keyslot[NFLHIP_EXPR_MAX_OPERANDS];
        const size_t cap = std::min<size_t>(idx.size() / 8 + 1, 8);
        for (int j = 0; j < nin; ++j) {
          const pay_t *seen[8];
          size_t ns = 0;
          bool few = idx.size() >= 2;
          for (size_t i : idx) {
            if (!few) break;
            const pay_t *p = ops[i].e.in[j];
            size_t k = ns;
            while (k-- > 0 && seen[k] != p) {}
            if (k == size_t(-1)) {
              if (ns == cap) few = false;
              else seen[ns++] = p;
            }
          }
          keyslot[j] = few && ns < idx.size();
        }
        for (size_t i : idx) {
          const pay_t *key[NFLHIP_EXPR_MAX_OPERANDS];
          for (int j = 0; j < nin; ++j) key[j] = keyslot[j] ? ops[i].e.in[j] : nullptr;
          size_t g = sub.size();
          for (size_t k = sub.size(); k-- > 0;)
            if (std::equal(key, key + nin, sub[k].key)) { g = k; break; }
          if (g == sub.size()) {
            sub.emplace_back();
            std::copy(key, key + nin, sub.back().key);
          }
          sub[g].idx.push_back(i);
        }
      }
      for (auto &sv : sub) {
        std::vector<size_t> &sidx = sv.idx;
        if (kind == K_FWD_FMA) {
          // program order; a run = consecutive result buffers (both results), keys at constant strides, stream ids of
          // every Gaussian operand in arithmetic progression: the samplers' compact outputs and ONE fused launch
          const bool two = nin == 2;
          const int nx = two ? 3 : 2;
          for (size_t a = 0; a < sidx.size();) {
            const op &o0 = ops[sidx[a]];
            size_t kstride[2] = {0, 0};
            uint64_t sstride[3] = {0, 0, 0};
            size_t b = a + 1;
            while (b < sidx.size()) {
              const op &prev = ops[sidx[b - 1]], &cur = ops[sidx[b]];
              bool ok = static_cast<char *>(cur.out->dev) == static_cast<char *>(prev.out->dev) + ctx_t::chunk_bytes &&
                        (!two || static_cast<char *>(cur.f.out2->dev) == static_cast<char *>(prev.f.out2->dev) + ctx_t::chunk_bytes);
              for (int j = 0; j < nin && ok; ++j) {
                const ptrdiff_t d = static_cast<char *>(cur.f.in[j]->dev) - static_cast<char *>(prev.f.in[j]->dev);
                if (d < 0 || d % ptrdiff_t(ctx_t::chunk_bytes)) ok = false;
                else if (b == a + 1) kstride[j] = size_t(d) / ctx_t::chunk_bytes;
                else if (size_t(d) != kstride[j] * ctx_t::chunk_bytes) ok = false;
              }
              for (int j = 0; j < nx && ok; ++j) {
                const uint64_t d = cur.f.sid[j] - prev.f.sid[j];
                if (b == a + 1) sstride[j] = d;
                else if (d != sstride[j]) ok = false;
              }
              if (!ok) break;
              ++b;
            }
            const size_t cnt = b - a;
            int fmt = NFLHIP_FMT_I8;
            for (int j = 0; j < nx; ++j) fmt = std::max(fmt, small_format(o0.f.tab, o0.f.amp[j]));
            const size_t es = fmt == NFLHIP_FMT_I8 ? 1 : fmt == NFLHIP_FMT_I16 ? 2 : 4, each = (cnt * P::degree * es + 255) / 256 * 256;
            char *buf = static_cast<char *>(small_buffer(each * size_t(nx)));
            nflhip_operand x[3], k[2];
            for (int j = 0; j < nx; ++j) {
              check(ctx, nflhip_sample_gauss_small_seq_dev(ctx, buf + each * size_t(j), fmt, cnt, o0.f.tab, o0.f.amp[j], smp.key, o0.f.sid[j],
                                                           sstride[j], st), "deferred set(gaussian), compact");
              x[j].ptr = buf + each * size_t(j);
              x[j].stride = 1;
              x[j].format = fmt;
              ++launches;
            }
            for (int j = 0; j < nin; ++j) {
              k[j].ptr = o0.f.in[j]->dev;
              k[j].stride = cnt > 1 ? kstride[j] : 0;
              k[j].format = NFLHIP_FMT_WORDS;
            }
            check(ctx, two ? nflhip_fwd_fma2_dev(ctx, o0.out->dev, o0.f.out2->dev, &x[0], &k[0], &x[1], &k[1], &x[2], cnt, st)
                           : nflhip_fwd_fma_dev(ctx, o0.out->dev, &x[0], &k[0], &x[1], cnt, st),
                  "deferred transform + multiply-add");
            for (size_t q = a; q < b; ++q) launched[sidx[q]] = 1;
            ++launches;
            coalesced += cnt * (two ? 8 : 5);   // (the operations the run's members were recorded as)
            a = b;
          }
          continue;
        }
        {  // by destination address (program order among equals); a loop's results already are in that order
          bool sorted = true;
          for (size_t k = 1; k < sidx.size() && sorted; ++k) sorted = !(ops[sidx[k]].out->dev < ops[sidx[k - 1]].out->dev);
          if (!sorted) std::stable_sort(sidx.begin(), sidx.end(), [&](size_t x, size_t y) { return ops[x].out->dev < ops[y].out->dev; });
        }
        for (size_t a = 0; a < sidx.size();) {
          const op &o0 = ops[sidx[a]];
          size_t stride[NFLHIP_EXPR_MAX_OPERANDS], ostride = 1;
          size_t b = a + 1;
          while (b < sidx.size()) {
            const op &prev = ops[sidx[b - 1]], &cur = ops[sidx[b]];
            bool ok = true;
            const ptrdiff_t od = static_cast<char *>(cur.out->dev) - static_cast<char *>(prev.out->dev);
            if (od <= 0 || od % ptrdiff_t(ctx_t::chunk_bytes)) break;
            if (kind == K_FMA_INV && size_t(od) != ctx_t::chunk_bytes) break;   // (the fused entry writes dense results)
            if (b == a + 1) ostride = size_t(od) / ctx_t::chunk_bytes;
            else if (size_t(od) != ostride * ctx_t::chunk_bytes) break;
            for (int j = 0; j < nin && ok; ++j) {
              const ptrdiff_t d = static_cast<char *>(cur.e.in[j]->dev) - static_cast<char *>(prev.e.in[j]->dev);
              if (d < 0 || d % ptrdiff_t(ctx_t::chunk_bytes)) ok = false;
              else if (b == a + 1) stride[j] = size_t(d) / ctx_t::chunk_bytes;
              else if (size_t(d) != stride[j] * ctx_t::chunk_bytes) ok = false;
            }
            if (!ok) break;
            ++b;
          }
          const size_t cnt = b - a;
          const void *d[NFLHIP_EXPR_MAX_OPERANDS];
          for (int j = 0; j < nin; ++j) d[j] = o0.e.in[j]->dev;
          if (kind == K_FMA_INV) {   // in[0] +- in[1] * in[2], then the inverse transform: one launch
            nflhip_operand w[3];
            for (int j = 0; j < 3; ++j) {
              w[j].ptr = d[j];
              w[j].stride = cnt > 1 ? stride[j] : 0;
              w[j].format = NFLHIP_FMT_WORDS;
            }
            check(ctx, nflhip_fma_inv_dev(ctx, o0.out->dev, &w[1], &w[2], &w[0], o0.e.code[0], cnt, st), "deferred multiply-add + inverse transform");
            coalesced += cnt;   // (two recorded operations per member)
          } else if (cnt == 1) {
            check(ctx, nflhip_eval_dev(ctx, o0.out->dev, d, size_t(nin), o0.e.code, size_t(o0.len), 1, st), "deferred operator=(expr)");
          } else {
            check(ctx, nflhip_eval_strided_dev(ctx, o0.out->dev, ostride, d, stride, size_t(nin), o0.e.code, size_t(o0.len), cnt, st),
                  "deferred operator=(expr)");
          }
          for (size_t k = a; k < b; ++k) launched[sidx[k]] = 1;
          ++launches;
          coalesced += cnt;
          a = b;
        }
      }
      finish_post(idx);
    }
    guard.complete = true;
  }
};
}  // namespace detail

// ---------------------------------------------------------------- expression templates (ops.hpp:52-97)
namespace ops {

// the functors poly::operator=(expr) evaluates (ops.hpp:99-242): on the device they are opcodes; the template
// parameters keep the reference's spelling (ops::mulmod_shoup<T, nfl::simd::serial>, tests/nfllib_demo_main_op.cpp:79)
template <class T, class tag> struct addmod { using simd_mode = tag; static constexpr int code = NFLHIP_OP_ADD; };
template <class T, class tag> struct submod { using simd_mode = tag; static constexpr int code = NFLHIP_OP_SUB; };
template <class T, class tag> struct mulmod { using simd_mode = tag; static constexpr int code = NFLHIP_OP_MUL; };
template <class T, class tag> struct mulmod_shoup { using simd_mode = tag; static constexpr int code = NFLHIP_OP_MUL_SHOUP; };
template <class T, class tag> struct compute_shoup { using simd_mode = tag; static constexpr int code = NFLHIP_OP_COMPUTE_SHOUP; };
template <class T, class tag> struct eqmod { using simd_mode = tag; static constexpr int code = -1; };
template <class T, class tag> struct neqmod { using simd_mode = tag; static constexpr int code = -1; };
template <class T, class tag> struct shoup { using simd_mode = tag; static constexpr int code = -1; };  // marker, ops.hpp:153-163

template <class Op, class... Args> struct expr;

// leaves of an expression tree: a poly, or a poly_p handle (poly_p.hpp:11-204) standing for its polynomial
template <class A, class Poly> struct is_leaf : std::is_same<A, Poly> {};
template <class T, size_t D, size_t M> struct is_leaf<poly_p<T, D, M>, poly<T, D, M>> : std::true_type {};
template <class T, size_t D, size_t M> inline const poly<T, D, M> &leaf(const poly<T, D, M> &p) { return p; }
template <class T, size_t D, size_t M> inline const poly<T, D, M> &leaf(const poly_p<T, D, M> &p) { return p.poly_obj(); }

// postfix program of one expression tree (include/nflhip.h, NFLHIP_EXPR_*), built when the tree is assigned: distinct
// leaves become operands 0..7, every node appends its opcode.  A leaf is an inline host poly (`host` = its words) or a
// resident handle (`pay` = its payload).
struct program {
  unsigned char code[NFLHIP_EXPR_MAX_LEN];
  size_t len = 0;
  const void *id[NFLHIP_EXPR_MAX_OPERANDS];
  const void *host[NFLHIP_EXPR_MAX_OPERANDS];
  void *pay[NFLHIP_EXPR_MAX_OPERANDS];
  size_t noperands = 0, nhandles = 0;
  int depth = 0;
  bool ok = true;
  void push_leaf(const void *ident, const void *h, void *p) {
    size_t k = 0;
    while (k < noperands && id[k] != ident) ++k;
    if (k == noperands) {
      if (noperands == NFLHIP_EXPR_MAX_OPERANDS) { ok = false; return; }
      id[k] = ident;
      host[k] = h;
      pay[k] = p;
      ++noperands;
      if (p) ++nhandles;
    }
    emit((unsigned char)k, +1);
  }
  void emit(unsigned char c, int delta) {
    if (len == NFLHIP_EXPR_MAX_LEN) { ok = false; return; }
    code[len++] = c;
    depth += delta;
    if (depth > 4) ok = false;
  }
};
template <class T, size_t D, size_t M> inline void push(program &pr, const poly<T, D, M> &p) { pr.push_leaf(&p, p.cdata(), nullptr); }
template <class T, size_t D, size_t M> inline void push(program &pr, const poly_p<T, D, M> &p) {
  pr.push_leaf(p.payload_id(), nullptr, p.payload_id());
}

template <class Op> struct opcode { static constexpr int value = -1; static constexpr int delta = 0; };
template <class T, class tag> struct opcode<addmod<T, tag>> { static constexpr int value = NFLHIP_EXPR_ADD; static constexpr int delta = -1; };
template <class T, class tag> struct opcode<submod<T, tag>> { static constexpr int value = NFLHIP_EXPR_SUB; static constexpr int delta = -1; };
template <class T, class tag> struct opcode<mulmod<T, tag>> { static constexpr int value = NFLHIP_EXPR_MUL; static constexpr int delta = -1; };
template <class T, class tag> struct opcode<mulmod_shoup<T, tag>> { static constexpr int value = NFLHIP_EXPR_MUL_SHOUP; static constexpr int delta = -2; };
template <class T, class tag> struct opcode<compute_shoup<T, tag>> { static constexpr int value = NFLHIP_EXPR_COMPUTE_SHOUP; static constexpr int delta = 0; };
template <class Op> struct is_eq : std::false_type {};
template <class T, class tag> struct is_eq<eqmod<T, tag>> : std::true_type {};
template <class Op> struct is_neq : std::false_type {};
template <class T, class tag> struct is_neq<neqmod<T, tag>> : std::true_type {};

template <class Op, class... Args> struct expr {
  using simd_mode = typename Op::simd_mode;
  std::tuple<Args const &...> args;
  expr(Args const &... a) : args(a...) {}
  typedef typename std::remove_cv<typename std::remove_reference<
      decltype(std::get<0>(std::declval<std::tuple<Args const &...>>()))>::type>::type first_type;
  typedef typename first_type::value_type value_type;
  typedef typename first_type::poly_type poly_type;
  typedef detail::payload<poly_type> payload_type;
  static constexpr size_t degree = first_type::degree;
  static constexpr size_t nmoduli = first_type::nmoduli;
  static constexpr size_t nbits = first_type::nbits;
  static constexpr size_t aggregated_modulus_bit_size = first_type::aggregated_modulus_bit_size;
  using p = params<value_type>;

  // evaluate this node into the host polynomial `out`: the whole tree in ONE fused device pass when it fits the
  // host-pointer program limits (<= 3 distinct leaves, stack depth <= 4), else one pass per node
  void eval(poly_type &out) const {
    program pr;
    lower(pr);
    if (pr.ok && opcode<Op>::value >= 0 && pr.noperands <= 3) {
      const void *h[3] = {nullptr, nullptr, nullptr};
      for (size_t k = 0; k < pr.noperands; ++k)
        h[k] = pr.pay[k] ? static_cast<payload_type *>(pr.pay[k])->host_ro().cdata() : pr.host[k];
      if (out.apply_program(pr, h)) return;
    }
    eval_impl(out, std::integral_constant<size_t, sizeof...(Args)>());
  }
  // evaluate this node into a resident payload: every leaf is read on the device (handles as they are, inline polys
  // through a pooled staging buffer), the result stays in HBM.  `pr` was lowered by the caller (before it re-seated its
  // own payload).  false = the program does not fit / the engine declined (tiny rows): the caller goes through the host.
  static bool run_resident(const program &pr, payload_type &out) {
    if (!pr.ok || opcode<Op>::value < 0) return false;
    typedef typename payload_type::ctx_t ctx_t;
    if (degree * sizeof(value_type) < 16) return false;  // (rows shorter than one 16-byte vector: nflhip_eval declines)
    if (detail::strictmod) {   // (runs the queue: a debugging build trades the batching for the assertion)
      const unsigned skip = detail::strict_exempt(pr.code, pr.len);
      for (size_t k = 0; k < pr.noperands; ++k) {
        if (skip >> k & 1) continue;
        if (pr.pay[k]) detail::strict_dev(ctx_t::get(), static_cast<payload_type *>(pr.pay[k])->dev_ro(), 1, ctx_t::queue(), "operator=(expr)");
        else detail::strict_host(ctx_t::get(), pr.host[k], 1, "operator=(expr)");
      }
    }
    typedef detail::lazy<poly_type> lazy_t;
    if (lazy_t::usable() && pr.nhandles == pr.noperands && pr.noperands <= size_t(lazy_t::max_in)) {  // every leaf is a handle: record, do not launch
      lazy_t::inst().record([&](typename lazy_t::op &o) {
        o.kind = lazy_t::K_EVAL;
        o.out = &out;
        o.nin = static_cast<unsigned char>(pr.noperands);
        for (size_t k = 0; k < pr.noperands; ++k) o.e.in[k] = static_cast<payload_type *>(pr.pay[k]);
        o.len = static_cast<unsigned char>(pr.len);
        std::memcpy(o.e.code, pr.code, pr.len);
      });
      return true;
    }
    nflhip_ctx *ctx = ctx_t::get();
    const void *d[NFLHIP_EXPR_MAX_OPERANDS];
    void *staged[NFLHIP_EXPR_MAX_OPERANDS];
    size_t nstaged = 0;
    for (size_t k = 0; k < pr.noperands; ++k) {
      if (pr.pay[k]) {
        d[k] = static_cast<payload_type *>(pr.pay[k])->dev_ro();
      } else {
        void *s = ctx_t::acquire();
        staged[nstaged++] = s;
        detail::check(ctx, nflhip_memcpy_h2d(ctx, s, pr.host[k], payload_type::bytes, ctx_t::queue()), "operator=(expr)");
        d[k] = s;
      }
    }
    bool aliases = false;
    for (size_t k = 0; k < pr.noperands; ++k) aliases |= pr.pay[k] == static_cast<void *>(&out);
    void *o = aliases ? out.dev_rw() : out.dev_wo();
    const int rc = nflhip_eval_dev(ctx, o, d, pr.noperands, pr.code, pr.len, 1, ctx_t::queue());
    for (size_t k = 0; k < nstaged; ++k) ctx_t::release(staged[k]);
    detail::check(ctx, rc, "operator=(expr)");
    return true;
  }
  // append this subtree to a postfix program
  void lower(program &pr) const {
    lower_args(pr, std::integral_constant<size_t, 0>());
    if (opcode<Op>::value < 0) pr.ok = false;
    else pr.emit((unsigned char)opcode<Op>::value, opcode<Op>::delta);
  }

  // expr::operator bool (ops.hpp:81-95): true as soon as ONE lane of the value is non-zero
  operator bool() const { return truth(is_eq<Op>(), is_neq<Op>()); }  // (implicit, as in the reference: `ok &= (a == b);` compiles)

 private:
  template <class A> static void lower_one(const A &a, program &pr, std::true_type) { push(pr, a); }
  template <class A> static void lower_one(const A &a, program &pr, std::false_type) { a.lower(pr); }
  template <size_t I> void lower_args(program &pr, std::integral_constant<size_t, I>) const {
    typedef typename std::remove_cv<typename std::remove_reference<decltype(std::get<I>(args))>::type>::type A;
    lower_one(std::get<I>(args), pr, is_leaf<A, poly_type>());
    lower_args(pr, std::integral_constant<size_t, I + 1>());
  }
  void lower_args(program &, std::integral_constant<size_t, sizeof...(Args)>) const {}
  template <class A> static const poly_type &materialise(const A &a, poly_type &tmp, std::true_type) { (void)tmp; return leaf(a); }
  template <class A> static const poly_type &materialise(const A &a, poly_type &tmp, std::false_type) {
    a.eval(tmp);
    return tmp;
  }
  template <class A> static const poly_type &mat(const A &a, poly_type &tmp) {
    return materialise(a, tmp, is_leaf<A, poly_type>());
  }
  void eval_impl(poly_type &out, std::integral_constant<size_t, 1>) const {
    poly_type *t0 = poly_type::make_temp();
    const poly_type &a = mat(std::get<0>(args), *t0);
    out.apply(Op::code, a, a, a);
    poly_type::drop_temp(t0);
  }
  void eval_impl(poly_type &out, std::integral_constant<size_t, 2>) const {
    poly_type *t0 = poly_type::make_temp(), *t1 = poly_type::make_temp();
    const poly_type &a = mat(std::get<0>(args), *t0);
    const poly_type &b = mat(std::get<1>(args), *t1);
    out.apply(Op::code, a, b, b);
    poly_type::drop_temp(t0);
    poly_type::drop_temp(t1);
  }
  void eval_impl(poly_type &out, std::integral_constant<size_t, 3>) const {
    poly_type *t0 = poly_type::make_temp(), *t1 = poly_type::make_temp(), *t2 = poly_type::make_temp();
    const poly_type &a = mat(std::get<0>(args), *t0);
    const poly_type &b = mat(std::get<1>(args), *t1);
    const poly_type &c = mat(std::get<2>(args), *t2);
    out.apply(Op::code, a, b, c);
    poly_type::drop_temp(t0);
    poly_type::drop_temp(t1);
    poly_type::drop_temp(t2);
  }
  bool truth(std::false_type, std::false_type) const {  // arithmetic expression: any non-zero word
    poly_type *t = poly_type::make_temp();
    eval(*t);
    const bool r = bool(*t);
    poly_type::drop_temp(t);
    return r;
  }
  // both sides resident handles: compare in HBM; otherwise on host images through the host-pointer entry
  template <class A, class B> static bool cmp_resident(const A &, const B &, bool, bool &, std::false_type) { return false; }
  template <class A, class B> static bool cmp_resident(const A &a, const B &b, bool want_eq, bool &result, std::true_type) {
    typedef typename payload_type::ctx_t ctx_t;
    payload_type *pa = static_cast<payload_type *>(a.payload_id()), *pb = static_cast<payload_type *>(b.payload_id());
    if (!(pa->dev_valid || pb->dev_valid)) return false;  // both live on the host: no point uploading
    int r = 0;
    nflhip_ctx *ctx = ctx_t::get();
    const void *da = pa->dev_ro(), *db = pb->dev_ro();
    detail::check(ctx, want_eq ? nflhip_any_eq_dev(ctx, da, db, 1, &r, ctx_t::queue()) : nflhip_any_neq_dev(ctx, da, db, 1, &r, ctx_t::queue()),
                  "operator== / !=");
    result = r != 0;
    return true;
  }
  bool cmp(bool want_eq) const {
    typedef typename std::remove_cv<typename std::remove_reference<decltype(std::get<0>(args))>::type>::type A;
    typedef typename std::remove_cv<typename std::remove_reference<decltype(std::get<1>(args))>::type>::type B;
    bool result = false;
    if (cmp_resident(std::get<0>(args), std::get<1>(args), want_eq, result,
                     std::integral_constant<bool, std::is_same<A, poly_p<value_type, degree, nmoduli>>::value &&
                                                      std::is_same<B, poly_p<value_type, degree, nmoduli>>::value>()))
      return result;
    poly_type *t0 = poly_type::make_temp(), *t1 = poly_type::make_temp();
    const poly_type &a = mat(std::get<0>(args), *t0);
    const poly_type &b = mat(std::get<1>(args), *t1);
    const bool r = poly_type::any_cmp(a, b, want_eq);
    poly_type::drop_temp(t0);
    poly_type::drop_temp(t1);
    return r;
  }
  bool truth(std::true_type, std::false_type) const { return cmp(true); }    // "any lane equal"  (the reference's quirk)
  bool truth(std::false_type, std::true_type) const { return cmp(false); }   // "any lane differs"
};

// ops::make_op<Op>(args...) (ops.hpp:249-260) incl. the shoup(a*b, b') -> mulmod_shoup(a, b, b') rewrite (ops.hpp:267-277)
template <class Op, class... Args> struct _make_op {
  expr<Op, Args...> operator()(Args const &... args) const { return expr<Op, Args...>(args...); }
};
template <class... Args> using retag = typename common_mode<typename Args::simd_mode...>::type;
template <class tag0, class tag1, class type, class Arg0, class Arg1, class Arg2>
struct _make_op<shoup<type, tag0>, expr<mulmod<type, tag1>, Arg0, Arg1>, Arg2> {
  expr<mulmod_shoup<type, tag1>, Arg0, Arg1, Arg2> operator()(expr<mulmod<type, tag1>, Arg0, Arg1> const &from0, Arg2 const &from1) const {
    return expr<mulmod_shoup<type, tag1>, Arg0, Arg1, Arg2>(std::get<0>(from0.args), std::get<1>(from0.args), from1);
  }
};
template <class Op, class... Args> auto make_op(Args const &... args) -> decltype(_make_op<Op, Args...>{}(args...)) {
  return _make_op<Op, Args...>{}(args...);
}

}  // namespace ops

// ---------------------------------------------------------------- poly (poly.hpp:82-310)
template <class T, size_t Degree, size_t NbModuli> class poly {
  template <class P> friend class tests::poly_tests_proxy;

  static constexpr size_t N = Degree * NbModuli;
  T _data[N] __attribute__((aligned(32)));

 public:
  typedef typename params<T>::value_type value_type;
  typedef typename params<T>::greater_value_type greater_value_type;
  typedef typename params<T>::signed_value_type signed_value_type;
  typedef T *pointer_type;
  typedef T const *const_pointer_type;
  typedef pointer_type iterator;
  typedef const_pointer_type const_iterator;
  typedef poly poly_type;
  using simd_mode = CC_SIMD;
  static constexpr size_t degree = Degree;
  static constexpr size_t nmoduli = NbModuli;
  static constexpr size_t nbits = params<T>::kModulusBitsize;
  static constexpr size_t aggregated_modulus_bit_size = NbModuli * nbits;

  /* constructors (core.hpp:64-84) */
  poly() { set(value_type(0)); }
  explicit poly(detail::uninitialized_t) {}  // (engine plumbing: storage about to be overwritten entirely)
  poly(uniform const &u) { set(u); }
  poly(non_uniform const &m) { set(m); }
  poly(hwt_dist const &m) { set(m); }
  poly(ZO_dist const &m) { set(m); }
  template <class in_class, unsigned _lu_depth> poly(gaussian<in_class, T, _lu_depth> const &m) { set(m); }
  poly(value_type v, bool reduce_coeffs = true) { set(v, reduce_coeffs); }
  poly(std::initializer_list<value_type> values, bool reduce_coeffs = true) { set(values, reduce_coeffs); }
  template <class It> poly(It first, It last, bool reduce_coeffs = true) { set(first, last, reduce_coeffs); }
  template <class Op, class... Args> poly(ops::expr<Op, Args...> const &e) { *this = e; }

  void set(value_type v, bool reduce_coeffs = true) {
    if (v == 0) std::fill(begin(), end(), value_type(0));
    else set({v}, reduce_coeffs);
  }
  void set(std::initializer_list<value_type> values, bool reduce_coeffs = true) { set(values.begin(), values.end(), reduce_coeffs); }
  // contract of core.hpp:101-137: up to `degree` values are ONE row image, zero-padded to the degree and written to every
  // modulus row (reduced per row unless reduce_coeffs is off); otherwise exactly degree * nmoduli values, row by row
  template <class It> void set(It first, It last, bool reduce_coeffs = true) {
    rows_from(first, size_t(std::distance(first, last)), "core",
              [reduce_coeffs](decltype(*first) v, value_type p) { return reduce_coeffs ? value_type(v % p) : value_type(v); });
  }
 private:
  // the row filler behind set(It, It) and set_mpz(It, It): residue(value, modulus) gives the word to store
  template <class It, class F> void rows_from(It first, size_t count, const char *who, F residue) {
    const bool row_by_row = count == degree * nmoduli;
    if (count > degree && !row_by_row)
      throw std::runtime_error(std::string(who) + ": an initializer longer than the degree must hold degree * nmoduli values");
    const size_t given = row_by_row ? degree : count;
    It src = first;
    for (size_t cm = 0; cm < nmoduli; ++cm) {
      if (!row_by_row) src = first;             // (the same row image for every modulus)
      const value_type p = get_modulus(cm);
      T *row = _data + cm * degree;
      for (size_t i = 0; i < given; ++i, ++src) row[i] = residue(*src, p);
      std::fill(row + given, row + degree, value_type(0));
    }
  }
 public:
  // mask-then-subtract rule of core.hpp:165-176: on the device's keystream (fresh per call), or on a seeded
  // counter stream for `uniform(seed)`
  void set(uniform const &u) {
    if (!u.seeded) {
      sample(detail::uniform_rule(), 0, 1, "set(uniform)");
      return;
    }
    for (size_t cm = 0; cm < nmoduli; cm++) {
      const uint64_t p = get_modulus(cm);
      int bits = 0;  // floor(log2 p) + 1 (core.hpp:165-166)
      while (bits < 63 && (uint64_t(1) << bits) <= p) ++bits;
      const uint64_t mask = (uint64_t(1) << bits) - 1;
      for (size_t i = 0; i < degree; i++) {
        uint64_t v = detail::splitmix64_at(u.seed, 0, cm * degree + i) & mask;
        if (v >= p) v -= p;
        _data[cm * degree + i] = T(v);
      }
    }
  }
  // bounded / zero-one / hamming-weight / Gaussian noise, one small integer per coefficient replicated over the
  // moduli (core.hpp:195-391); misuse throws std::runtime_error like the reference (core.hpp:205-210)
  void set(non_uniform const &m) { sample(NFLHIP_DIST_BOUNDED, m.upper_bound, m.amplifier, "set(non_uniform)"); }
  void set(ZO_dist const &m) { sample(NFLHIP_DIST_ZO | detail::dist_flags, m.rho, 1, "set(ZO_dist)"); }
  void set(hwt_dist const &m) { sample(NFLHIP_DIST_HWT | detail::dist_flags, m.hwt, 1, "set(hwt_dist)"); }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, T, _lu_depth> const &m) {
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx(), nflhip_sample_gauss(ctx(), _data, 1, m.fg_prng->table(ctx()), m.amplifier, s.key, s.next++),
                  "set(gaussian)");
  }

  poly &operator=(value_type v) { set(v); return *this; }
  poly &operator=(uniform const &u) { set(u); return *this; }
  poly &operator=(non_uniform const &m) { set(m); return *this; }
  poly &operator=(hwt_dist const &m) { set(m); return *this; }
  poly &operator=(ZO_dist const &m) { set(m); return *this; }
  template <class in_class, unsigned _lu_depth> poly &operator=(gaussian<in_class, T, _lu_depth> const &m) { set(m); return *this; }
  poly &operator=(std::initializer_list<value_type> values) { set(values); return *this; }
  // THE evaluation point of an expression tree (core.hpp:24-37)
  template <class Op, class... Args> poly &operator=(ops::expr<Op, Args...> const &e) {
    e.eval(*this);
    return *this;
  }

  explicit operator bool() const {  // core.hpp:39-43
    return std::find_if(begin(), end(), [](value_type v) { return v != 0; }) != end();
  }

  iterator begin() { return _data; }
  iterator end() { return _data + N; }
  const_iterator begin() const { return _data; }
  const_iterator end() const { return _data + N; }
  const_iterator cbegin() const { return _data; }
  const_iterator cend() const { return _data + N; }
  value_type const &operator()(size_t cm, size_t i) const { return _data[cm * degree + i]; }
  value_type &operator()(size_t cm, size_t i) { return _data[cm * degree + i]; }
  pointer_type data() { return _data; }
  const_pointer_type cdata() const { return _data; }
  template <class M> auto load(size_t cm, size_t i) const -> decltype(M::load(&(this->operator()(cm, i)))) { return M::load(&(*this)(cm, i)); }
  static constexpr value_type get_modulus(size_t n) { return params<T>::P[n]; }

  /* ntt stuff - public API (poly.hpp:167-168) */
  void ntt_pow_phi() {
    if (detail::strictmod) detail::strict_host(ctx(), _data, 1, "ntt_pow_phi");
    detail::check(ctx(), nflhip_ntt_fwd(ctx(), _data, 1), "ntt_pow_phi");
  }
  void invntt_pow_invphi() {
    if (detail::strictmod) detail::strict_host(ctx(), _data, 1, "invntt_pow_invphi");
    detail::check(ctx(), nflhip_ntt_inv(ctx(), _data, 1), "invntt_pow_invphi");
  }

  /* manual serializers (poly.hpp:180-185): raw little-endian words */
  void serialize_manually(std::ostream &os) { os.write(reinterpret_cast<char *>(_data), N * sizeof(T)); }
  void deserialize_manually(std::istream &is) { is.read(reinterpret_cast<char *>(_data), N * sizeof(T)); }
  // cereal hook, identical to the reference's (poly.hpp:186-190): works with any archive type that accepts a C array
  template <class Archive> void serialize(Archive &archive) { archive(_data); }

  /* poly::core (poly.hpp:196-245) and the static `base` object, reachable through the tests::poly_tests_proxy friend
   * exactly as tests/ntt_perfs.cpp:121-134 does: core::ntt / core::inv_ntt are the CYCLIC row transforms
   * (core.hpp:455-557) and run on the device (nflhip_ntt_row); base.omegas[cm] ... are host views of the reference's
   * own table layouts (nflhip_get_table, NFLHIP_TAB_*), fetched on first access -- nothing happens at static-init
   * time.  The device engine owns its tables: core::ntt accepts the table pointers base hands out (that is how it
   * knows the modulus row and the direction) and throws std::runtime_error for foreign tables. */
 protected:
  class core {
    template <class P> friend class tests::poly_tests_proxy;
    struct tables {
      std::vector<value_type> phis, shoupphis, invpoly_times_invphis, shoupinvpoly_times_invphis, omegas, invomegas, invpolyDegree;
      tables() {
        auto fetch = [](std::vector<value_type> &v, int which, size_t words) {
          v.assign(NbModuli * words, 0);
          for (size_t cm = 0; cm < NbModuli; ++cm)
            detail::check(ctx(), nflhip_get_table(ctx(), which, cm, v.data() + cm * words, words * sizeof(value_type)), "core tables");
        };
        fetch(phis, NFLHIP_TAB_PHIS, Degree);
        fetch(shoupphis, NFLHIP_TAB_SHOUPPHIS, Degree);
        fetch(invpoly_times_invphis, NFLHIP_TAB_INVPOLY_INVPHIS, Degree);
        fetch(shoupinvpoly_times_invphis, NFLHIP_TAB_SHOUPINVPOLY_INVPHIS, Degree);
        fetch(omegas, NFLHIP_TAB_OMEGAS, 2 * Degree);
        fetch(invomegas, NFLHIP_TAB_INVOMEGAS, 2 * Degree);
        fetch(invpolyDegree, NFLHIP_TAB_INVDEGREE, 1);
      }
    };
    static tables &tabs() {
      static tables t;
      return t;
    }
    // member views with the reference's names and index shapes: view[cm][i]
    template <int Which> struct view {
      value_type *operator[](size_t cm) const {
        tables &t = tabs();
        return Which == 0   ? t.phis.data() + cm * Degree
               : Which == 1 ? t.shoupphis.data() + cm * Degree
               : Which == 2 ? t.invpoly_times_invphis.data() + cm * Degree
               : Which == 3 ? t.shoupinvpoly_times_invphis.data() + cm * Degree
               : Which == 4 ? t.omegas.data() + cm * 2 * Degree
               : Which == 5 ? t.omegas.data() + cm * 2 * Degree + Degree
               : Which == 6 ? t.invomegas.data() + cm * 2 * Degree
                            : t.invomegas.data() + cm * 2 * Degree + Degree;
      }
    };
    struct scalar_view {
      value_type &operator[](size_t cm) const { return tabs().invpolyDegree[cm]; }
    };

   public:
    core() {}
    void ntt_pow_phi(poly &op) { op.ntt_pow_phi(); }
    void invntt_pow_invphi(poly &op) { op.invntt_pow_invphi(); }
    // core.hpp:455-532: in-place cyclic transform of one row, natural in, bit-reversed out, [0,p)
    static bool ntt(value_type *x, const value_type *wtab, const value_type *winvtab, value_type const p) {
      (void)winvtab;
      return run_row(x, wtab, p, 0);
    }
    // core.hpp:539-557: permut, ntt with the inverse tables, permut (invK is unused there too)
    static bool inv_ntt(value_type *x, const value_type *const inv_wtab, const value_type *const inv_winvtab, value_type invK,
                        value_type const p) {
      (void)inv_winvtab;
      (void)invK;
      return run_row(x, inv_wtab, p, NFLHIP_ROW_BITREV_IO);
    }

   private:
    static bool run_row(value_type *x, const value_type *wtab, value_type p, int mode) {
      tables &t = tabs();
      for (size_t cm = 0; cm < NbModuli; ++cm) {
        if (get_modulus(cm) != p) continue;
        if (wtab == t.omegas.data() + cm * 2 * Degree) {
          detail::check(ctx(), nflhip_ntt_row(ctx(), x, cm, mode, 1), "core::ntt");
          return true;
        }
        if (wtab == t.invomegas.data() + cm * 2 * Degree) {
          detail::check(ctx(), nflhip_ntt_row(ctx(), x, cm, mode | NFLHIP_ROW_INVERSE_TABLES, 1), "core::ntt");
          return true;
        }
      }
      throw std::runtime_error("nfl(hip): core::ntt runs the engine's own tables (base.omegas / base.invomegas of this modulus)");
    }

   public:  // (private in the reference, reached through the friend proxy; views are stateless)
    view<0> phis;
    view<1> shoupphis;
    view<2> invpoly_times_invphis;
    view<3> shoupinvpoly_times_invphis;
    view<4> omegas;
    view<5> shoupomegas;
    view<6> invomegas;
    view<7> shoupinvomegas;
    scalar_view invpolyDegree;
  };
  static core base;

 public:
  /* CRT (gmp.hpp:183-219) on little-endian 64-bit limb vectors */
  static size_t crt_limbs() { return nflhip_crt_limbs(ctx()); }
  // out[i*L .. i*L+L) = limbs of X_i in [0, Q): the mpz_export image of poly2mpz()
  void poly2limbs(std::vector<uint64_t> &out) const {
    out.assign(degree * crt_limbs(), 0);
    detail::check(ctx(), nflhip_crt_lift(ctx(), out.data(), _data, 1), "poly2mpz");
  }
  // mpz2poly: x(cm,i) = X_i mod p_cm for non-negative X_i given as L_in limbs each
  void limbs2poly(const uint64_t *limbs, size_t L_in) {
    detail::check(ctx(), nflhip_crt_project(ctx(), _data, limbs, L_in, 1), "mpz2poly");
  }
#ifdef NFL_HIP_WITH_GMP
  /* ---- GMP-typed surface (poly.hpp:249-307, gmp.hpp) ---- */
  // constants of the nested GMP class (gmp.hpp:113-155), imported once from the context (which built them for the
  // device) -- moduli_product, lifting_integers -- plus modulus_shoup from the reference's formula
  struct GMP {
    mpz_t moduli_product, modulus_shoup;
    size_t bits_in_moduli_product, bits_in_modulus_shoup, shift_modulus_shoup;
    std::array<mpz_t, NbModuli> lifting_integers;
    GMP() {
      const size_t cap = nflhip_crt_limbs(ctx()) + 2;
      std::vector<uint64_t> buf(cap);
      size_t nl = 0;
      detail::check(ctx(), nflhip_get_crt_constant(ctx(), 0, 0, buf.data(), cap, &nl), "GMP: moduli_product");
      mpz_init(moduli_product);
      mpz_import(moduli_product, nl, -1, sizeof(uint64_t), 0, 0, buf.data());
      bits_in_moduli_product = mpz_sizeinbase(moduli_product, 2);
      size_t lg = 0;
      while ((size_t(2) << lg) <= NbModuli) ++lg;  // static_log2<nmoduli> (meta.hpp:12-30)
      shift_modulus_shoup = bits_in_moduli_product + params<T>::kModulusRepresentationBitsize + lg + 1;  // gmp.hpp:124-126
      mpz_init2(modulus_shoup, shift_modulus_shoup);
      mpz_ui_pow_ui(modulus_shoup, 2, shift_modulus_shoup);
      mpz_tdiv_q(modulus_shoup, modulus_shoup, moduli_product);
      bits_in_modulus_shoup = mpz_sizeinbase(modulus_shoup, 2);
      for (size_t cm = 0; cm < NbModuli; cm++) {
        detail::check(ctx(), nflhip_get_crt_constant(ctx(), 1, cm, buf.data(), cap, &nl), "GMP: lifting_integers");
        mpz_init(lifting_integers[cm]);
        mpz_import(lifting_integers[cm], nl, -1, sizeof(uint64_t), 0, 0, buf.data());
      }
    }
    ~GMP() {
      for (size_t cm = 0; cm < NbModuli; cm++) mpz_clear(lifting_integers[cm]);
      mpz_clears(modulus_shoup, moduli_product, nullptr);
    }
    GMP(GMP const &) = delete;
    GMP &operator=(GMP const &) = delete;
  };
  static GMP &gmp() {
    static GMP g;
    return g;
  }
  static size_t bits_in_moduli_product() { return gmp().bits_in_moduli_product; }
  static mpz_t &moduli_product() { return gmp().moduli_product; }
  static mpz_t &modulus_shoup() { return gmp().modulus_shoup; }
  static std::array<mpz_t, NbModuli> lifting_integers() { return gmp().lifting_integers; }  // shallow, like poly.hpp:307

  poly(mpz_t const &v) { set_mpz(v); }
  poly(std::array<mpz_t, Degree> const &values) { set_mpz(values); }
  poly(std::initializer_list<mpz_t> const &values) { set_mpz(values); }
  void set_mpz(mpz_t const &v) { set_mpz(&v, &v + 1); }
  void set_mpz(std::array<mpz_t, Degree> const &values) { set_mpz(values.begin(), values.end()); }
  void set_mpz(std::initializer_list<mpz_t> const &values) { set_mpz(values.begin(), values.end()); }
  poly &operator=(mpz_t const &v) { set_mpz(v); return *this; }
  poly &operator=(std::array<mpz_t, Degree> const &values) { set_mpz(values); return *this; }
  poly &operator=(std::initializer_list<mpz_t> const &values) { set_mpz(values); return *this; }
#ifdef NFL_HIP_HAVE_GMPXX
  poly(mpz_class const &v) { set_mpz(v); }
  poly(std::array<mpz_class, Degree> const &values) { set_mpz(values); }
  poly(std::initializer_list<mpz_class> const &values) { set_mpz(values); }
  void set_mpz(mpz_class const &v) { set_mpz(&v, &v + 1); }
  void set_mpz(std::array<mpz_class, Degree> const &values) { set_mpz(values.begin(), values.end()); }
  void set_mpz(std::initializer_list<mpz_class> const &values) { set_mpz(values.begin(), values.end()); }
  poly &operator=(mpz_class const &v) { set_mpz(v); return *this; }
  poly &operator=(std::array<mpz_class, Degree> const &values) { set_mpz(values); return *this; }
  poly &operator=(std::initializer_list<mpz_class> const &values) { set_mpz(values); return *this; }
#endif
  // gmp.hpp:73-108: fewer than `degree` integers are zero-padded and replicated over the moduli, exactly
  // degree*nmoduli are taken row by row; every value is reduced with floor semantics (mpz_fdiv_ui: negative
  // integers give non-negative residues).  A setter, like set(It, It): runs on the host.
  template <class It> void set_mpz(It first, It last) {
    rows_from(first, size_t(std::distance(first, last)), "gmp",
              [](decltype(*first) v, value_type p) { return value_type(mpz_fdiv_ui(detail::as_mpz(v), p)); });
  }

  // gmp.hpp:169-209 on the device (nflhip_crt_lift); the returned integers are initialised here and owned by the
  // caller (mpz_clear), as in the reference
  std::array<mpz_t, Degree> poly2mpz() const {
    std::array<mpz_t, Degree> rop;
    for (size_t i = 0; i < degree; i++) mpz_init2(rop[i], gmp().shift_modulus_shoup - 1);
    poly2mpz(rop);
    return rop;
  }
  void poly2mpz(std::array<mpz_t, Degree> &rop) const {
    std::vector<uint64_t> limbs;
    poly2limbs(limbs);
    const size_t L = crt_limbs();
    for (size_t i = 0; i < degree; i++) mpz_import(rop[i], L, -1, sizeof(uint64_t), 0, 0, limbs.data() + i * L);
  }
  // gmp.hpp:211-219 on the device (nflhip_crt_project).  mpz_fdiv_ui semantics: a negative integer is first
  // brought into [0, Q) (same residues), the device only sees magnitudes.
  void mpz2poly(std::array<mpz_t, Degree> const &v) {
    size_t L = 1;
    bool any_negative = false;
    for (size_t i = 0; i < degree; i++) {
      L = std::max(L, (mpz_sizeinbase(v[i], 2) + 63) / 64);
      any_negative |= mpz_sgn(v[i]) < 0;
    }
    if (any_negative) L = std::max(L, (bits_in_moduli_product() + 63) / 64);
    std::vector<uint64_t> limbs(degree * L, 0);
    mpz_t t;
    mpz_init(t);
    for (size_t i = 0; i < degree; i++) {
      if (mpz_sgn(v[i]) < 0) {
        mpz_fdiv_r(t, v[i], moduli_product());
        mpz_export(limbs.data() + i * L, nullptr, -1, sizeof(uint64_t), 0, 0, t);
      } else {
        mpz_export(limbs.data() + i * L, nullptr, -1, sizeof(uint64_t), 0, 0, v[i]);
      }
    }
    mpz_clear(t);
    limbs2poly(limbs.data(), L);
  }
#endif

  // ---- plumbing used by the expression templates (not part of the reference surface) ----
  static nflhip_ctx *ctx() { return detail::context<T, Degree, NbModuli>::get(); }
  static void *queue() { return detail::context<T, Degree, NbModuli>::queue(); }  // the stream resident operations run on
  static void *acquire_device() { return detail::context<T, Degree, NbModuli>::acquire(); }
  static void release_device(void *p) { detail::context<T, Degree, NbModuli>::release(p); }
  void sample(int dist, uint64_t p0, uint64_t p1, const char *what) {
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx(), nflhip_sample(ctx(), _data, 1, dist, p0, p1, s.key, s.next++), what);
  }
  void apply(int op, const poly &a, const poly &b, const poly &bp) {
    if (detail::strictmod) {
      if (op != NFLHIP_OP_COMPUTE_SHOUP) detail::strict_host(ctx(), a._data, 1, "operator=(expr)");
      detail::strict_host(ctx(), op == NFLHIP_OP_COMPUTE_SHOUP ? a._data : b._data, 1, "operator=(expr)");
    }
    detail::check(ctx(), nflhip_pointwise(ctx(), op, _data, a._data, b._data, bp._data, 1), "operator=(expr)");
  }
  // fused tree evaluation; false = the engine declined (tiny rows): the caller goes node by node
  bool apply_program(const ops::program &pr, const void *const *host_operands) {
    if (detail::strictmod) {
      const unsigned skip = detail::strict_exempt(pr.code, pr.len);
      for (size_t k = 0; k < pr.noperands; ++k)
        if (!(skip >> k & 1)) detail::strict_host(ctx(), host_operands[k], 1, "operator=(expr)");
    }
    const int rc = nflhip_eval(ctx(), _data, host_operands, pr.noperands, pr.code, pr.len, 1);
    if (rc == NFLHIP_ERR_UNSUPPORTED) return false;
    detail::check(ctx(), rc, "operator=(expr)");
    return true;
  }
  static bool any_cmp(const poly &a, const poly &b, bool want_eq) {
    int r = 0;
    detail::check(ctx(), want_eq ? nflhip_any_eq(ctx(), a._data, b._data, 1, &r) : nflhip_any_neq(ctx(), a._data, b._data, 1, &r),
                  "operator== / !=");
    return r != 0;
  }
  static poly *make_temp() {  // polys can be MBs: temporaries of nested expressions live on the heap
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(poly)) != 0) throw std::bad_alloc();
    return new (mem) poly(detail::uninitialized_t());
  }
  static void drop_temp(poly *p) {
    p->~poly();
    free(p);
  }
} __attribute__((aligned(32)));
template <class T, size_t Degree, size_t NbModuli> typename poly<T, Degree, NbModuli>::core poly<T, Degree, NbModuli>::base;

// ---------------------------------------------------------------- operators (poly.hpp:346-352, ops.hpp:18-45)
namespace ops {
template <class X> struct is_node : std::false_type {};
template <class T, size_t D, size_t M> struct is_node<poly<T, D, M>> : std::true_type {};
template <class T, size_t D, size_t M> struct is_node<poly_p<T, D, M>> : std::true_type {};
template <class Op, class... A> struct is_node<expr<Op, A...>> : std::true_type {};
// == and != on a poly_p are its own members (poly_p.hpp:112-140); everything else is generic
template <class X> struct is_cmp_node : is_node<X> {};
template <class T, size_t D, size_t M> struct is_cmp_node<poly_p<T, D, M>> : std::false_type {};
// anything else handed to the shoup marker is a compile-time error, as in the reference (ops.hpp:153-163)
template <class type, class tag, class A, class B> struct _make_op<shoup<type, tag>, A, B> {
  static_assert(sizeof(A) == 0, "shoup(expr, b') needs expr = a * b (ops.hpp:160)");
};
}  // namespace ops

#define NFL_HIP_BINARY(SYM, NAME)                                                                                       \
  template <class A, class B>                                                                                           \
  typename std::enable_if<ops::is_node<A>::value && ops::is_node<B>::value,                                             \
                          ops::expr<ops::NAME<typename A::value_type, CC_SIMD>, A, B>>::type SYM(A const &a, B const &b) { \
    static_assert(std::is_same<typename A::poly_type, typename B::poly_type>::value, "correct type combination");       \
    return ops::make_op<ops::NAME<typename A::value_type, CC_SIMD>>(a, b);                                              \
  }
NFL_HIP_BINARY(operator-, submod)
NFL_HIP_BINARY(operator+, addmod)
NFL_HIP_BINARY(operator*, mulmod)
#undef NFL_HIP_BINARY
#define NFL_HIP_COMPARE(SYM, NAME)                                                                                      \
  template <class A, class B>                                                                                           \
  typename std::enable_if<ops::is_cmp_node<A>::value && ops::is_node<B>::value,                                         \
                          ops::expr<ops::NAME<typename A::value_type, CC_SIMD>, A, B>>::type SYM(A const &a, B const &b) { \
    static_assert(std::is_same<typename A::poly_type, typename B::poly_type>::value, "correct type combination");       \
    return ops::make_op<ops::NAME<typename A::value_type, CC_SIMD>>(a, b);                                              \
  }
NFL_HIP_COMPARE(operator==, eqmod)
NFL_HIP_COMPARE(operator!=, neqmod)
#undef NFL_HIP_COMPARE

template <class A>
typename std::enable_if<ops::is_node<A>::value, ops::expr<ops::compute_shoup<typename A::value_type, CC_SIMD>, A>>::type compute_shoup(A const &a) {
  return ops::make_op<ops::compute_shoup<typename A::value_type, CC_SIMD>>(a);
}
// shoup(a*b, b') is rewritten into mulmod_shoup(a, b, b') (ops.hpp:267-277)
template <class A, class B>
auto shoup(A const &prod, B const &bprime) -> decltype(ops::make_op<ops::shoup<typename A::value_type, CC_SIMD>>(prod, bprime)) {
  return ops::make_op<ops::shoup<typename A::value_type, CC_SIMD>>(prod, bprime);
}

// ---------------------------------------------------------------- poly_p (poly_p.hpp:11-204)
// Copy-on-write handle with the reference's members; the shared payload is RESIDENT (detail::payload): operator
// expressions over handles, transforms, comparisons and the random constructors run on the device and leave the result
// in HBM; poly_obj(), operator()(cm,i), serialisation and the GMP surface bring it to the host (and a non-const access
// marks the device image stale).
template <class T, size_t Degree, size_t NbModuli> class poly_p {
 public:
  typedef poly<T, Degree, NbModuli> poly_type;
  using value_type = typename poly_type::value_type;
  using greater_value_type = typename poly_type::greater_value_type;
  using simd_mode = typename poly_type::simd_mode;
  static constexpr size_t nmoduli = poly_type::nmoduli;
  static constexpr size_t degree = poly_type::degree;
  static constexpr size_t nbits = poly_type::nbits;
  static constexpr size_t aggregated_modulus_bit_size = poly_type::aggregated_modulus_bit_size;

 private:
  typedef detail::payload<poly_type> payload_type;
  typedef typename payload_type::ctx_t ctx_t;
  typedef std::shared_ptr<payload_type> ptr_type;
  mutable ptr_type _p;

  static ptr_type fresh() { return std::allocate_shared<payload_type>(detail::block_pool_alloc<payload_type>()); }
  // constructors: the zero polynomial and the random tags never touch the host; everything else builds the host image
  // with poly's own constructor (same argument meaning, same exceptions)
  static ptr_type make_pointer() { return fresh(); }
  static ptr_type make_pointer(uniform const &m) { ptr_type p = fresh(); sample_into(*p, m); return p; }
  static ptr_type make_pointer(non_uniform const &m) { ptr_type p = fresh(); sample_into(*p, m); return p; }
  static ptr_type make_pointer(ZO_dist const &m) { ptr_type p = fresh(); sample_into(*p, m); return p; }
  static ptr_type make_pointer(hwt_dist const &m) { ptr_type p = fresh(); sample_into(*p, m); return p; }
  template <class in_class, unsigned _lu_depth> static ptr_type make_pointer(gaussian<in_class, T, _lu_depth> const &m) {
    ptr_type p = fresh();
    sample_into(*p, m);
    return p;
  }
  template <class Op, class... A> static ptr_type make_pointer(ops::expr<Op, A...> const &e) {
    ptr_type p = fresh();
    assign_expr(p, e);
    return p;
  }
  // (the overloads above must win over this forwarding template for rvalue tags and expressions)
  template <class X, class Dummy = void> struct device_init : std::false_type {};
  template <class Dummy> struct device_init<uniform, Dummy> : std::true_type {};
  template <class Dummy> struct device_init<non_uniform, Dummy> : std::true_type {};
  template <class Dummy> struct device_init<ZO_dist, Dummy> : std::true_type {};
  template <class Dummy> struct device_init<hwt_dist, Dummy> : std::true_type {};
  template <class in_class, unsigned _lu_depth, class Dummy> struct device_init<gaussian<in_class, T, _lu_depth>, Dummy> : std::true_type {};
  template <class Op, class... A, class Dummy> struct device_init<ops::expr<Op, A...>, Dummy> : std::true_type {};
  template <class A0, class... Args>
  static typename std::enable_if<!device_init<typename std::decay<A0>::type>::value || sizeof...(Args) != 0, ptr_type>::type make_pointer(
      A0 &&a0, Args &&... args) {
    ptr_type p = fresh();
    p->alloc_host();
    p->host->~poly_type();
    new (p->host) poly_type(std::forward<A0>(a0), std::forward<Args>(args)...);
    p->host_valid = true;
    return p;
  }
  // shared with another HANDLE (references held by deferred operations do not count)
  static bool shared(const ptr_type &p, long extra = 0) {
    if (p.use_count() - extra <= 1) return false;  // nobody else at all
    // The queue's reference and the flag that discounts it change together under the queue's lock -- also when a queue
    // run started by ANOTHER thread retires this handle's operations -- so they are read under it.
    std::lock_guard<detail::light_lock> lk(lazy_t::inst().mu);
    return p.use_count() - p->qrefs - extra > 1;
  }
  void detach() const {
    if (shared(_p)) _p = std::allocate_shared<payload_type>(detail::block_pool_alloc<payload_type>(), *_p);  // (device-to-device when the value lives in HBM)
  }
  void detach_for_overwrite() {
    if (shared(_p)) _p = fresh();
  }

  // ---- device-side samplers (same keystream discipline as poly::sample: a fresh stream id per call)
  typedef detail::lazy<poly_type> lazy_t;
  static void check_sample_args(int dist, uint64_t p0, uint64_t p1, const nflhip_gauss *tab) {
    // a deferred constructor must still throw where it is written (core.hpp:205-210): validate with an empty batch
    const unsigned char zero[32] = {0};
    if (tab) detail::check(ctx_t::get(), nflhip_sample_gauss_dev(ctx_t::get(), nullptr, 0, 0, tab, p1, zero, 0, ctx_t::queue()), "set(gaussian)");
    else detail::check(ctx_t::get(), nflhip_sample_dev(ctx_t::get(), nullptr, 0, 0, dist, p0, p1, zero, 0, ctx_t::queue()), "random constructor");
  }
  static bool defer_sample(payload_type &p, int kind, int dist, uint64_t p0, uint64_t p1, uint64_t sid, const nflhip_gauss *tab) {
    if (!lazy_t::usable()) return false;
    if (kind != lazy_t::K_FILL) {
      // (validated once per distinct argument tuple: loops repeat a handful of constructors -- the LWE demo's alternate between two
      //  amplifiers, so remembering only the last one made two validating C-ABI calls per encryption, more than all the recording)
      struct seen_t { int dist; uint64_t p0, p1; const nflhip_gauss *tab; };
      static thread_local seen_t seen[8];
      static thread_local unsigned nseen = 0, victim = 0;
      bool known = false;
      for (unsigned k = 0; k < nseen && !known; ++k)
        known = seen[k].dist == dist && seen[k].p0 == p0 && seen[k].p1 == p1 && seen[k].tab == tab;
      if (!known) {
        check_sample_args(dist, p0, p1, tab);
        const unsigned at = nseen < 8 ? nseen++ : victim++ % 8;
        seen[at] = seen_t{dist, p0, p1, tab};
      }
    }
    lazy_t::inst().record([&](typename lazy_t::op &o) {
      o.kind = static_cast<unsigned char>(kind);
      o.out = &p;
      o.s.dist = dist;
      o.s.p0 = p0;
      o.s.p1 = p1;
      o.s.sid = sid;
      o.s.tab = tab;
    });
    return true;
  }
  static void sample_dist(payload_type &p, int dist, uint64_t p0, uint64_t p1, const char *what) {
    detail::sampler &s = detail::sampler::get();
    const uint64_t sid = s.next++;
    if (defer_sample(p, lazy_t::K_SAMPLE, dist, p0, p1, sid, nullptr)) return;
    detail::check(ctx_t::get(), nflhip_sample_dev(ctx_t::get(), p.dev_wo(), 0, 1, dist, p0, p1, s.key, sid, ctx_t::queue()), what);
  }
  static void sample_into(payload_type &p, uniform const &u) {
    if (u.seeded) {
      if (defer_sample(p, lazy_t::K_FILL, 0, 0, 0, u.seed, nullptr)) return;
      detail::check(ctx_t::get(), nflhip_fill_uniform_dev(ctx_t::get(), p.dev_wo(), 0, 1, u.seed, 0, ctx_t::queue()), "set(uniform)");
    } else {
      sample_dist(p, detail::uniform_rule(), 0, 1, "set(uniform)");
    }
  }
  static void sample_into(payload_type &p, non_uniform const &m) { sample_dist(p, NFLHIP_DIST_BOUNDED, m.upper_bound, m.amplifier, "set(non_uniform)"); }
  static void sample_into(payload_type &p, ZO_dist const &m) { sample_dist(p, NFLHIP_DIST_ZO | detail::dist_flags, m.rho, 1, "set(ZO_dist)"); }
  static void sample_into(payload_type &p, hwt_dist const &m) { sample_dist(p, NFLHIP_DIST_HWT | detail::dist_flags, m.hwt, 1, "set(hwt_dist)"); }
  template <class in_class, unsigned _lu_depth> static void sample_into(payload_type &p, gaussian<in_class, T, _lu_depth> const &m) {
    detail::sampler &s = detail::sampler::get();
    const nflhip_gauss *tab = m.fg_prng->table(ctx_t::get());
    const uint64_t sid = s.next++;
    if (defer_sample(p, lazy_t::K_GAUSS, 0, 0, m.amplifier, sid, tab)) return;
    detail::check(ctx_t::get(), nflhip_sample_gauss_dev(ctx_t::get(), p.dev_wo(), 0, 1, tab, m.amplifier, s.key, sid, ctx_t::queue()),
                  "set(gaussian)");
  }
  // THE evaluation point of an expression tree over handles (core.hpp:24-37): one fused device pass, result resident.
  // `p` is re-seated first when it is shared (copy-on-write without the copy: the whole value is overwritten); the
  // program is lowered BEFORE that, so a tree that reads the old value still sees it.
  template <class Op, class... A> static void assign_expr(ptr_type &p, ops::expr<Op, A...> const &e) {
    ops::program pr;
    e.lower(pr);
    ptr_type keep = p;  // the old payload stays alive while the kernel reads it
    if (shared(p, 1)) p = fresh();  // shared with another handle (`keep` is the extra reference)
    if (ops::expr<Op, A...>::run_resident(pr, *p)) return;
    // through the host: node by node, or trees the fused program cannot hold
    poly_type *tmp = poly_type::make_temp();
    try {
      e.eval(*tmp);
    } catch (...) {
      poly_type::drop_temp(tmp);
      throw;
    }
    std::memcpy(p->host_wo().data(), tmp->cdata(), payload_type::bytes);
    poly_type::drop_temp(tmp);
  }

 public:
  poly_p(poly_p const &o) : _p(o._p) {}
  poly_p(poly_p &o) : _p(const_cast<poly_p const &>(o)._p) {}
  poly_p(poly_p &&o) : _p(std::move(o._p)) {}
  template <class... Args> poly_p(Args &&... args) : _p(make_pointer(std::forward<Args>(args)...)) {}
  poly_p(poly_type const &) = delete;
  poly_p(poly_type &&) = delete;

  // the polynomial as a host object (poly_p.hpp:47-53): forces the value to the host; the non-const form may be
  // written through, so it also retires the device image
  poly_type &poly_obj() {
    detach();
    return _p->host_rw();
  }
  poly_type const &poly_obj() const { return _p->host_ro(); }
  void *payload_id() const { return _p.get(); }  // (engine plumbing: identity of the shared payload)
  bool resident() const { return _p->dev_valid; }  // the current value is in HBM (no upload needed by the next device op)
  // wait for every enqueued operation of this ring type (results are otherwise only awaited when read on the host)
  static void synchronize() {
    lazy_t::inst().flush();
    detail::check(ctx_t::get(), nflhip_stream_sync(ctx_t::get(), ctx_t::queue()), "synchronize");
  }
  // run the deferred operations of this ring type now (without waiting for the device); statistics of the queue so far
  static void flush() { lazy_t::inst().flush(); }
  static size_t deferred_launches() { return lazy_t::inst().launches; }
  static size_t deferred_operations() { return lazy_t::inst().coalesced; }

  template <class Op, class... A> poly_p &operator=(ops::expr<Op, A...> const &e) {
    assign_expr(_p, e);
    return *this;
  }
  poly_p &operator=(uniform const &m) { detach_for_overwrite(); sample_into(*_p, m); return *this; }
  poly_p &operator=(non_uniform const &m) { detach_for_overwrite(); sample_into(*_p, m); return *this; }
  poly_p &operator=(ZO_dist const &m) { detach_for_overwrite(); sample_into(*_p, m); return *this; }
  poly_p &operator=(hwt_dist const &m) { detach_for_overwrite(); sample_into(*_p, m); return *this; }
  template <class in_class, unsigned _lu_depth> poly_p &operator=(gaussian<in_class, T, _lu_depth> const &m) {
    detach_for_overwrite();
    sample_into(*_p, m);
    return *this;
  }
  // (everything else goes through the host polynomial, as in the reference; the overloads above must win for rvalue
  // expressions and tags, which a plain forwarding template would otherwise capture)
  template <class O>
  typename std::enable_if<!device_init<typename std::decay<O>::type>::value && !std::is_same<typename std::decay<O>::type, poly_p>::value,
                          poly_p &>::type
  operator=(O &&o) {
    poly_obj() = std::forward<O>(o);
    return *this;
  }
  poly_p &operator=(std::initializer_list<T> values) {
    poly_obj() = values;
    return *this;
  }
  poly_p &operator=(poly_p const &o) {
    if (this != &o) _p = o._p;
    return *this;
  }
  poly_p &operator=(poly_p &o) { return *this = const_cast<poly_p const &>(o); }
  poly_p &operator=(poly_p &&o) {
    if (this != &o) _p = std::move(o._p);
    return *this;
  }

  bool operator==(poly_p const &o) const { return _p.get() == o._p.get() ? true : bool(ops::make_op<ops::eqmod<T, CC_SIMD>>(*this, o)); }
  bool operator!=(poly_p const &o) const { return _p.get() == o._p.get() ? false : bool(ops::make_op<ops::neqmod<T, CC_SIMD>>(*this, o)); }
  template <class O> bool operator==(O const &o) const { return bool(poly_obj() == o); }
  template <class O> bool operator!=(O const &o) const { return bool(poly_obj() != o); }

  value_type &operator()(size_t cm, size_t i) { return poly_obj()(cm, i); }
  value_type const &operator()(size_t cm, size_t i) const { return poly_obj()(cm, i); }
  template <class M> auto load(size_t cm, size_t i) const -> decltype(M::load(&(this->operator()(cm, i)))) { return M::load(&(*this)(cm, i)); }
  static constexpr value_type get_modulus(size_t n) { return poly_type::get_modulus(n); }

  /* ntt stuff - public API (poly_p.hpp:141-142): in place in HBM */
  void ntt_pow_phi() { transform(lazy_t::K_NTT_FWD); }
  void invntt_pow_invphi() { transform(lazy_t::K_NTT_INV); }

 private:
  void transform(int kind) {
    bool tried = false;
    if (lazy_t::usable() && !detail::strictmod && _p.use_count() > 1) {
      // queued values carry the queue's reference: the copy-on-write test and the attempt to join the producing record need
      // the queue's lock both -- taken once here instead of twice
      std::lock_guard<detail::light_lock> lk(lazy_t::inst().mu);
      if (_p.use_count() - _p->qrefs <= 1) {
        tried = true;
        if (lazy_t::inst().join_transform(_p.get(), kind)) return;
      }
    }
    detach();
    if (detail::strictmod)
      detail::strict_dev(ctx_t::get(), _p->dev_ro(), 1, ctx_t::queue(), kind == lazy_t::K_NTT_FWD ? "ntt_pow_phi" : "invntt_pow_invphi");
    if (lazy_t::usable()) {
      if (!tried && lazy_t::inst().join_transform(_p.get(), kind)) return;
      lazy_t::inst().record([&](typename lazy_t::op &o) {
        o.kind = static_cast<unsigned char>(kind);
        o.out = _p.get();
      });
      return;
    }
    detail::check(ctx_t::get(), kind == lazy_t::K_NTT_FWD ? nflhip_ntt_fwd_dev(ctx_t::get(), _p->dev_rw(), 1, ctx_t::queue())
                                                          : nflhip_ntt_inv_dev(ctx_t::get(), _p->dev_rw(), 1, ctx_t::queue()),
                  kind == lazy_t::K_NTT_FWD ? "ntt_pow_phi" : "invntt_pow_invphi");
  }

 public:
  void serialize_manually(std::ostream &os) { poly_obj().serialize_manually(os); }
  void deserialize_manually(std::istream &is) { poly_obj().deserialize_manually(is); }
  template <class Archive> void serialize(Archive &archive) { archive(poly_obj()); }

  /* set (poly_p.hpp:161-167) */
  void set(value_type v, bool reduce_coeffs = true) { poly_obj().set(v, reduce_coeffs); }
  void set(uniform const &m) { *this = m; }
  void set(non_uniform const &m) { *this = m; }
  void set(ZO_dist const &m) { *this = m; }
  void set(hwt_dist const &m) { *this = m; }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, T, _lu_depth> const &m) { *this = m; }
  void set(std::initializer_list<value_type> values, bool reduce_coeffs = true) { poly_obj().set(values, reduce_coeffs); }
  void set(std::array<value_type, Degree> values, bool reduce_coeffs = true) { poly_obj().set(values.begin(), values.end(), reduce_coeffs); }
  template <class It> void set(It first, It last, bool reduce_coeffs = true) { poly_obj().set(first, last, reduce_coeffs); }

  /* CRT on limb vectors, as on poly */
  static size_t crt_limbs() { return poly_type::crt_limbs(); }
  void poly2limbs(std::vector<uint64_t> &out) const { poly_obj().poly2limbs(out); }
  void limbs2poly(const uint64_t *limbs, size_t L_in) { poly_obj().limbs2poly(limbs, L_in); }
#ifdef NFL_HIP_WITH_GMP
  /* the GMP-typed surface (poly_p.hpp:186-200) */
  void set_mpz(mpz_t const &v) { poly_obj().set_mpz(v); }
  void set_mpz(std::array<mpz_t, Degree> const &values) { poly_obj().set_mpz(values); }
#ifdef NFL_HIP_HAVE_GMPXX
  void set_mpz(mpz_class const &v) { poly_obj().set_mpz(v); }
  void set_mpz(std::array<mpz_class, Degree> const &values) { poly_obj().set_mpz(values); }
  void set_mpz(std::initializer_list<mpz_class> const &values) { poly_obj().set_mpz(values); }
#endif
  template <class It> void set_mpz(It first, It last) { poly_obj().set_mpz(first, last); }
  std::array<mpz_t, Degree> poly2mpz() { return const_cast<poly_p const *>(this)->poly_obj().poly2mpz(); }
  void poly2mpz(std::array<mpz_t, Degree> &array) { const_cast<poly_p const *>(this)->poly_obj().poly2mpz(array); }
  void mpz2poly(std::array<mpz_t, Degree> const &array) { poly_obj().mpz2poly(array); }
  static size_t bits_in_moduli_product() { return poly_type::bits_in_moduli_product(); }
  static mpz_t &moduli_product() { return poly_type::moduli_product(); }
  static mpz_t &modulus_shoup() { return poly_type::modulus_shoup(); }
  static std::array<mpz_t, nmoduli> lifting_integers() { return poly_type::lifting_integers(); }
#endif
};

template <class T, size_t Degree, size_t AggregatedModulusBitSize>
using poly_p_from_modulus = poly_p<T, Degree, AggregatedModulusBitSize / params<T>::kModulusBitsize>;

template <class T, size_t D, size_t M> std::ostream &operator<<(std::ostream &os, poly_p<T, D, M> const &p) {
  return os << p.poly_obj();
}

/* high level wrappers (poly.hpp:314-332) */
template <class T, size_t D, size_t M> void sub(poly<T, D, M> &out, poly<T, D, M> const &a, poly<T, D, M> const &b) { out = a - b; }
template <class T, size_t D, size_t M> void add(poly<T, D, M> &out, poly<T, D, M> const &a, poly<T, D, M> const &b) { out = a + b; }
template <class T, size_t D, size_t M> void mul(poly<T, D, M> &out, poly<T, D, M> const &a, poly<T, D, M> const &b) { out = a * b; }

template <class T, size_t Degree, size_t AggregatedModulusBitSize>
using poly_from_modulus = poly<T, Degree, AggregatedModulusBitSize / params<T>::kModulusBitsize>;

// same text format as the reference's stream operator (core.hpp:398-421)
template <class T, size_t D, size_t M> std::ostream &operator<<(std::ostream &os, poly<T, D, M> const &p) {
  const char *term = sizeof(T) == 8 ? "ULL" : (sizeof(T) == 4 ? "UL" : "U");
  bool first = true;
  os << "{ ";
  for (auto v : p) {
    if (first) { first = false; os << uint64_t(v); }
    else os << term << ", " << uint64_t(v);
  }
  return os << term << " }";
}

// ---------------------------------------------------------------- batch entry points
// A contiguous array of polys is the dense [batch][NbModuli][Degree] tensor the
// device wants (sizeof(poly) == N*sizeof(T)): one H2D, one kernel pass, one D2H.
namespace batch {
template <class P> void ntt_pow_phi(P *first, size_t count) {
  static_assert(sizeof(P) == P::degree * P::nmoduli * sizeof(typename P::value_type), "dense poly array");
  detail::check(P::ctx(), nflhip_ntt_fwd(P::ctx(), first->data(), count), "batch::ntt_pow_phi");
}
template <class P> void invntt_pow_invphi(P *first, size_t count) {
  detail::check(P::ctx(), nflhip_ntt_inv(P::ctx(), first->data(), count), "batch::invntt_pow_invphi");
}
// c[k] = INTT(NTT(a[k]) (.) NTT(b[k])): the fused metric path
template <class P> void polymul(P *c, P const *a, P const *b, size_t count) {
  detail::check(P::ctx(), nflhip_polymul(P::ctx(), c->data(), a->cdata(), b->cdata(), count), "batch::polymul");
}
template <class P> void pointwise(int op, P *out, P const *a, P const *b, P const *bprime, size_t count) {
  detail::check(P::ctx(), nflhip_pointwise(P::ctx(), op, out->data(), a->cdata(), b ? b->cdata() : nullptr,
                                           bprime ? bprime->cdata() : nullptr, count), "batch::pointwise");
}
}  // namespace batch

// ---------------------------------------------------------------- device-resident batches
// The reference's poly stores its words inline on the host (poly.hpp:87-88), so every per-poly call
// above crosses PCIe twice.  device_batch<P> keeps a dense [count][NbModuli][Degree] tensor resident
// in HBM (the role poly_p's shared payload plays on the host, poly_p.hpp:11-204) and runs the same
// operations through the *_dev entry points on one stream: upload once, compute, download once.
// device_batch(count, device) puts it on a GPU of its choice (default: the per-polynomial surface's device).
template <class P> class device_batch {
 public:
  typedef typename P::value_type value_type;
  typedef detail::context<value_type, P::degree, P::nmoduli> context_type;
  explicit device_batch(size_t count) : device_batch(count, detail::default_device().load()) {}
  device_batch(size_t count, int device) : n_(count), d_(nullptr), c_(&context_type::on(device)) {
    static_assert(sizeof(P) == P::degree * P::nmoduli * sizeof(value_type), "dense poly array");
    detail::check(ctx(), nflhip_malloc(ctx(), &d_, bytes()), "device_batch");
  }
  device_batch(const P *host, size_t count) : device_batch(count) { upload(host); }
  ~device_batch() {
    if (small_) nflhip_free(ctx(), small_);
    if (d_) nflhip_free(ctx(), d_);
  }
  device_batch(const device_batch &) = delete;
  device_batch &operator=(const device_batch &) = delete;
  device_batch(device_batch &&o) noexcept : n_(o.n_), d_(o.d_), c_(o.c_), small_(o.small_), small_cap_(o.small_cap_) {
    o.d_ = nullptr;
    o.small_ = nullptr;
    o.small_cap_ = 0;
  }

  size_t size() const { return n_; }
  size_t bytes() const { return n_ * sizeof(P); }
  void *data() { return d_; }
  const void *data() const { return d_; }
  int device() const { return c_->device; }
  nflhip_ctx *ctx() const { return c_->ctx; }     // the context of this batch's device ...
  void *queue() const { return c_->stream; }      // ... and the stream its operations are enqueued on

  void upload(const P *host) {
    detail::check(ctx(), nflhip_memcpy_h2d(ctx(), d_, host->cdata(), bytes(), queue()), "upload");
    sync();
  }
  void download(P *host) const {
    detail::check(ctx(), nflhip_memcpy_d2h(ctx(), host->data(), d_, bytes(), queue()), "download");
    sync();
  }
  void sync() const { detail::check(ctx(), nflhip_stream_sync(ctx(), queue()), "sync"); }

  // same names and meaning as the poly members (poly.hpp:167-168), over the whole batch
  void strict(const char *what) const {   // CHECK_STRICTMOD's assertion over the whole resident batch
    if (detail::strictmod) detail::strict_dev(ctx(), d_, n_, queue(), what);
  }
  void ntt_pow_phi() {
    strict("ntt_pow_phi");
    detail::check(ctx(), nflhip_ntt_fwd_dev(ctx(), d_, n_, queue()), "ntt_pow_phi");
  }
  void invntt_pow_invphi() {
    strict("invntt_pow_invphi");
    detail::check(ctx(), nflhip_ntt_inv_dev(ctx(), d_, n_, queue()), "invntt_pow_invphi");
  }
  // *this = op(a, b[, b'])  (NFLHIP_OP_*); aliasing allowed
  void assign(int op, const device_batch &a, const device_batch &b) {
    same_size(a); same_size(b);
    a.strict("pointwise"); b.strict("pointwise");
    detail::check(ctx(), nflhip_pointwise_dev(ctx(), op, d_, a.d_, b.d_, nullptr, n_, queue()), "pointwise");
  }
  void assign_mul_shoup(const device_batch &a, const device_batch &b, const device_batch &bprime) {
    same_size(a); same_size(b); same_size(bprime);
    a.strict("mulmod_shoup"); b.strict("mulmod_shoup");
    detail::check(ctx(), nflhip_pointwise_dev(ctx(), NFLHIP_OP_MUL_SHOUP, d_, a.d_, b.d_, bprime.d_, n_, queue()),
                  "mulmod_shoup");
  }
  void assign_compute_shoup(const device_batch &b) {
    same_size(b);
    detail::check(ctx(), nflhip_pointwise_dev(ctx(), NFLHIP_OP_COMPUTE_SHOUP, d_, b.d_, nullptr, nullptr, n_, queue()),
                  "compute_shoup");
  }
  // *this = INTT(NTT(a) (.) NTT(b)), the fused metric path
  void assign_polymul(const device_batch &a, const device_batch &b) {
    same_size(a); same_size(b);
    detail::check(ctx(), nflhip_polymul_dev(ctx(), d_, a.d_, b.d_, n_, queue()), "polymul");
  }
  // the same with b already in NTT form (keys of the LWE demo stay transformed, tests/nfllib_demo_main_op.cpp:26-46)
  void assign_polymul_ntt(const device_batch &a, const device_batch &b_ntt) {
    same_size(a); same_size(b_ntt);
    detail::check(ctx(), nflhip_polymul_ntt_dev(ctx(), d_, a.d_, b_ntt.d_, n_, queue()), "polymul_ntt");
  }
  // CRT lift / project of the whole resident batch (gmp.hpp:183-219): out[(b*degree + i)*L .. +L) = little-endian limbs
  // of X_{b,i} in [0, Q), L = P::crt_limbs(); limbs2poly takes L_in limbs per coefficient
  void poly2limbs(std::vector<uint64_t> &out) const {
    const size_t words = n_ * P::degree * P::crt_limbs();
    out.assign(words, 0);
    void *dl = nullptr;
    detail::check(ctx(), nflhip_malloc(ctx(), &dl, words * sizeof(uint64_t)), "device allocation");
    int rc = nflhip_crt_lift_dev(ctx(), static_cast<uint64_t *>(dl), d_, n_, queue());
    if (rc == 0) rc = nflhip_memcpy_d2h(ctx(), out.data(), dl, words * sizeof(uint64_t), queue());
    if (rc == 0) rc = nflhip_stream_sync(ctx(), queue());
    nflhip_free(ctx(), dl);
    detail::check(ctx(), rc, "poly2mpz");
  }
  void limbs2poly(const uint64_t *limbs, size_t L_in) {
    const size_t words = n_ * P::degree * L_in;
    void *dl = nullptr;
    detail::check(ctx(), nflhip_malloc(ctx(), &dl, words * sizeof(uint64_t)), "device allocation");
    int rc = nflhip_memcpy_h2d(ctx(), dl, limbs, words * sizeof(uint64_t), queue());
    if (rc == 0) rc = nflhip_crt_project_dev(ctx(), d_, static_cast<const uint64_t *>(dl), L_in, n_, queue());
    if (rc == 0) rc = nflhip_stream_sync(ctx(), queue());
    nflhip_free(ctx(), dl);
    detail::check(ctx(), rc, "mpz2poly");
  }
  // fused postfix expression over up to NFLHIP_EXPR_MAX_OPERANDS resident batches
  void assign_program(const unsigned char *program, size_t len, const device_batch *const *operands, size_t count) {
    const void *ptr[NFLHIP_EXPR_MAX_OPERANDS];
    if (count > NFLHIP_EXPR_MAX_OPERANDS) throw std::runtime_error("nfl(hip): too many operands");
    for (size_t i = 0; i < count; ++i) { same_size(*operands[i]); ptr[i] = operands[i]->d_; }
    if (detail::strictmod) {
      const unsigned skip = detail::strict_exempt(program, len);
      for (size_t i = 0; i < count; ++i)
        if (!(skip >> i & 1)) operands[i]->strict("eval");
    }
    detail::check(ctx(), nflhip_eval_dev(ctx(), d_, ptr, count, program, len, n_, queue()), "eval");
  }
  // the random constructors over the whole resident batch (same tags as poly's; one keystream per call).
  // `first_poly` / `stream_id` are for shards of one logical batch (sharded_batch): polynomial k of this batch is
  // polynomial first_poly + k of the keystream, so that the shards of a batch equal the batch drawn on one device.
  void set(uniform const &u, size_t first_poly = 0) {
    if (u.seeded) detail::check(ctx(), nflhip_fill_uniform_dev(ctx(), d_, first_poly, n_, u.seed, 0, queue()), "set(uniform)");
    else sample(detail::uniform_rule(), 0, 1, "set(uniform)", first_poly, detail::sampler::get().next++);
  }
  void set(non_uniform const &m) { sample(NFLHIP_DIST_BOUNDED, m.upper_bound, m.amplifier, "set(non_uniform)", 0, detail::sampler::get().next++); }
  void set(ZO_dist const &m) { sample(NFLHIP_DIST_ZO | detail::dist_flags, m.rho, 1, "set(ZO_dist)", 0, detail::sampler::get().next++); }
  void set(hwt_dist const &m) { sample(NFLHIP_DIST_HWT | detail::dist_flags, m.hwt, 1, "set(hwt_dist)", 0, detail::sampler::get().next++); }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, value_type, _lu_depth> const &m) {
    set_at(m, 0, detail::sampler::get().next++);
  }
  void set_at(uniform const &, size_t first_poly, uint64_t stream_id) { sample(detail::uniform_rule(), 0, 1, "set(uniform)", first_poly, stream_id); }
  void set_at(non_uniform const &m, size_t first_poly, uint64_t stream_id) { sample(NFLHIP_DIST_BOUNDED, m.upper_bound, m.amplifier, "set(non_uniform)", first_poly, stream_id); }
  void set_at(ZO_dist const &m, size_t first_poly, uint64_t stream_id) { sample(NFLHIP_DIST_ZO | detail::dist_flags, m.rho, 1, "set(ZO_dist)", first_poly, stream_id); }
  void set_at(hwt_dist const &m, size_t first_poly, uint64_t stream_id) { sample(NFLHIP_DIST_HWT | detail::dist_flags, m.hwt, 1, "set(hwt_dist)", first_poly, stream_id); }
  template <class in_class, unsigned _lu_depth>
  void set_at(gaussian<in_class, value_type, _lu_depth> const &m, size_t first_poly, uint64_t stream_id) {
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx(), nflhip_sample_gauss_dev(ctx(), d_, first_poly, n_, m.fg_prng->table(ctx()), m.amplifier, s.key,
                                                 stream_id, queue()), "set(gaussian)");
  }
  // ---- the transform-fused pipelines over a resident batch (include/nflhip.h "transform-fused pipelines"; what the
  // reference's LWE demo does around its transforms, tests/nfllib_demo_main_op.cpp:26-58).  `k` operands are in NTT form
  // and hold either one polynomial (a key shared by the whole batch) or one per element.  Results are bit-identical to
  //     X.set(x); E.set(e); X.ntt_pow_phi(); E.ntt_pow_phi(); *this = X * k + E;          (stream ids taken in that order)
  //     *this = b -+ a * k; this->invntt_pow_invphi();
  // -- the Gaussian polynomials only ever exist as one signed integer per coefficient, the transformed ones not at all.
  template <class Gx, class Ge> void assign_gaussian_fma(Gx const &x, const device_batch &k, Ge const &e) {
    gaussian_fma(nullptr, x, k, e, nullptr, e);
  }
  // *this = NTT(x) * k0 + NTT(e0), out1 = NTT(x) * k1 + NTT(e1): x is drawn and transformed once
  template <class Gx, class G0, class G1>
  void assign_gaussian_fma2(device_batch &out1, Gx const &x, const device_batch &k0, G0 const &e0, const device_batch &k1, G1 const &e1) {
    same_size(out1);
    gaussian_fma(&out1, x, k0, e0, &k1, e1);
  }
  // *this = INTT(b - a * k) (subtract) or INTT(b + a * k); a, b, k in NTT form; *this may be a or b
  void assign_fma_inv(const device_batch &a, const device_batch &k, const device_batch &b, bool subtract) {
    same_size(a);
    same_size(b);
    nflhip_operand oa = {a.d_, 1, NFLHIP_FMT_WORDS}, ok = key_operand(k), ob = {b.d_, 1, NFLHIP_FMT_WORDS};
    detail::check(ctx(), nflhip_fma_inv_dev(ctx(), d_, &oa, &ok, &ob, subtract ? 1 : 0, n_, queue()), "multiply-add + inverse transform");
  }
  // replicate one polynomial over the batch (a key shared by every ciphertext, ...)
  void fill(const P &one) {  // one upload + one broadcast kernel
    void *tmp = nullptr;
    detail::check(ctx(), nflhip_malloc(ctx(), &tmp, sizeof(P)), "device allocation");
    int rc = nflhip_memcpy_h2d(ctx(), tmp, one.cdata(), sizeof(P), queue());
    if (rc == 0) rc = nflhip_broadcast_dev(ctx(), d_, tmp, n_, queue());
    if (rc == 0) rc = nflhip_stream_sync(ctx(), queue());
    nflhip_free(ctx(), tmp);
    detail::check(ctx(), rc, "fill");
  }
  // the same from a resident handle: no host copy at all (a peer-to-peer copy when the batch lives on another device)
  void fill(const poly_p<value_type, P::degree, P::nmoduli> &one) {
    typedef detail::payload<P> payload_type;
    const void *src = static_cast<payload_type *>(one.payload_id())->dev_ro();
    if (ctx() == P::ctx()) {
      detail::check(ctx(), nflhip_broadcast_dev(ctx(), d_, src, n_, queue()), "fill");
      return;
    }
    detail::check(P::ctx(), nflhip_stream_sync(P::ctx(), P::queue()), "fill");  // the handle's pending writes
    void *tmp = nullptr;
    detail::check(ctx(), nflhip_malloc(ctx(), &tmp, sizeof(P)), "device allocation");
    int rc = nflhip_memcpy_peer_dev(ctx(), tmp, P::ctx(), src, sizeof(P), queue());
    if (rc == 0) rc = nflhip_broadcast_dev(ctx(), d_, tmp, n_, queue());
    if (rc == 0) rc = nflhip_stream_sync(ctx(), queue());
    nflhip_free(ctx(), tmp);
    detail::check(ctx(), rc, "fill");
  }
  bool any_equal(const device_batch &o) const { return cmp(o, true); }    // the reference's `a == b`
  bool any_differs(const device_batch &o) const { return cmp(o, false); } // the reference's `a != b`
  // 64-bit digest that composes over shards (nflhip_digest_dev): the digests of the shards of a batch add up to the
  // digest of the batch
  uint64_t digest(size_t first_poly = 0) const {
    uint64_t h = 0;
    detail::check(ctx(), nflhip_digest_dev(ctx(), d_, first_poly, n_, &h, queue()), "digest");
    return h;
  }

 private:
  void same_size(const device_batch &o) const {
    if (o.n_ != n_) throw std::runtime_error("nfl(hip): batch size mismatch");
    if (o.c_ != c_) throw std::runtime_error("nfl(hip): the batches of one operation must live on one device");
  }
  void sample(int dist, uint64_t p0, uint64_t p1, const char *what, size_t first_poly, uint64_t stream_id) {
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx(), nflhip_sample_dev(ctx(), d_, first_poly, n_, dist, p0, p1, s.key, stream_id, queue()), what);
  }
  bool cmp(const device_batch &o, bool want_eq) const {
    same_size(o);
    int r = 0;
    detail::check(ctx(), want_eq ? nflhip_any_eq_dev(ctx(), d_, o.d_, n_, &r, queue())
                                 : nflhip_any_neq_dev(ctx(), d_, o.d_, n_, &r, queue()), "compare");
    return r != 0;
  }
  nflhip_operand key_operand(const device_batch &k) const {
    if (k.c_ != c_) throw std::runtime_error("nfl(hip): the batches of one operation must live on one device");
    if (k.n_ != 1 && k.n_ != n_) throw std::runtime_error("nfl(hip): a key operand holds one polynomial or one per element");
    nflhip_operand o = {k.d_, size_t(k.n_ == 1 ? 0 : 1), NFLHIP_FMT_WORDS};
    return o;
  }
  // compact Gaussian polynomials of one call: a grow-only buffer of this batch (its consumers are on the batch's stream)
  void *small_buffer(size_t bytes) {
    if (bytes > small_cap_) {
      if (small_) {
        sync();
        nflhip_free(ctx(), small_);
        small_ = nullptr;
        small_cap_ = 0;
      }
      detail::check(ctx(), nflhip_malloc(ctx(), &small_, bytes), "compact sampler buffer");
      small_cap_ = bytes;
    }
    return small_;
  }
  template <class Gx, class G0, class G1>
  void gaussian_fma(device_batch *out1, Gx const &x, const device_batch &k0, G0 const &e0, const device_batch *k1, G1 const &e1) {
    typedef detail::lazy<P> lazy_t;
    detail::sampler &s = detail::sampler::get();
    const nflhip_gauss *tab[3] = {x.fg_prng->table(ctx()), e0.fg_prng->table(ctx()), out1 ? e1.fg_prng->table(ctx()) : nullptr};
    const uint64_t amp[3] = {x.amplifier, e0.amplifier, out1 ? e1.amplifier : 0};
    const int nx = out1 ? 3 : 2;
    int fmt = NFLHIP_FMT_I8;
    for (int j = 0; j < nx; ++j) fmt = std::max(fmt, (amp[j] >> 32) ? 99 : lazy_t::small_format(tab[j], uint32_t(amp[j])));
    if (fmt > NFLHIP_FMT_I32) {   // samples too wide for a compact format: the operator sequence itself
      const unsigned char fma[] = {0, 1, NFLHIP_EXPR_MUL, 2, NFLHIP_EXPR_ADD};
      device_batch X(n_, c_->device), E(n_, c_->device), K(n_, c_->device);
      X.set(x);
      E.set(e0);
      X.ntt_pow_phi();
      E.ntt_pow_phi();
      const device_batch *kk[2] = {&k0, k1};
      device_batch *oo[2] = {this, out1};
      for (int r = 0; r < (out1 ? 2 : 1); ++r) {
        if (r) {
          E.set(e1);
          E.ntt_pow_phi();
        }
        const device_batch *key = kk[r];
        if (key->n_ == 1) {   // (the expression entry takes dense operands: replicate the key)
          nflhip_operand src = {key->d_, 0, NFLHIP_FMT_WORDS};
          detail::check(ctx(), nflhip_expand_small_dev(ctx(), K.d_, &src, n_, queue()), "key broadcast");
          key = &K;
        }
        const device_batch *ops[] = {&X, key, &E};
        oo[r]->assign_program(fma, sizeof(fma), ops, 3);
      }
      sync();   // (the temporaries die here)
      return;
    }
    const size_t es = fmt == NFLHIP_FMT_I8 ? 1 : fmt == NFLHIP_FMT_I16 ? 2 : 4, each = (n_ * P::degree * es + 255) / 256 * 256;
    char *buf = static_cast<char *>(small_buffer(each * size_t(nx)));
    nflhip_operand xo[3];
    for (int j = 0; j < nx; ++j) {
      detail::check(ctx(), nflhip_sample_gauss_small_dev(ctx(), buf + each * size_t(j), fmt, 0, n_, tab[j], amp[j], s.key, s.next++, queue()),
                    "set(gaussian), compact");
      xo[j].ptr = buf + each * size_t(j);
      xo[j].stride = 1;
      xo[j].format = fmt;
    }
    nflhip_operand ka = key_operand(k0);
    if (out1) {
      nflhip_operand kb = key_operand(*k1);
      detail::check(ctx(), nflhip_fwd_fma2_dev(ctx(), d_, out1->d_, &xo[0], &ka, &xo[1], &kb, &xo[2], n_, queue()), "transform + multiply-add");
    } else {
      detail::check(ctx(), nflhip_fwd_fma_dev(ctx(), d_, &xo[0], &ka, &xo[1], n_, queue()), "transform + multiply-add");
    }
  }
  size_t n_;
  void *d_;
  context_type *c_;
  void *small_ = nullptr;
  size_t small_cap_ = 0;
};

// ---------------------------------------------------------------- batches split over the GPUs of one node
// The reference's callers hold dense arrays of independent polynomials (tests/tools.h:6-17) and loop over them; nothing
// in a loop iteration depends on another (core.hpp:597-599, 610-612, 31-35).  sharded_batch<P> cuts such an array into
// CONTIGUOUS shards, one per GPU (device r of n owns polynomials [first(r), first(r) + count(r)), nflhip_shard_range),
// from ONE process: one context and one stream per device, every operation fans out as one asynchronous call per shard
// (the host thread only enqueues), and there is no data-path collective -- operands are generated in place (the random
// constructors offset their keystream by the shard's first polynomial, so the shards of a batch equal the batch drawn
// on one device), uploaded shard by shard, or scattered once from a batch that lives on one device (peer-to-peer copies,
// one per link).  digest() is the checksum of checksums: the shard digests add up to the digest of the whole batch.
template <class P> class sharded_batch {
 public:
  typedef typename P::value_type value_type;
  typedef device_batch<P> shard_type;
  // all GPUs of the node
  explicit sharded_batch(size_t count) : sharded_batch(count, all_devices()) {}
  sharded_batch(size_t count, std::vector<int> const &devices) : n_(count) {
    if (devices.empty()) throw std::runtime_error("nfl(hip): sharded_batch needs at least one device");
    const int nd = int(devices.size());
    for (int r = 0; r < nd; ++r) {
      size_t f = 0, c = 0;
      detail::check(nullptr, nflhip_shard_range(count, nd, r, &f, &c), "shard_range");
      first_.push_back(f);
      shards_.emplace_back(c, devices[size_t(r)]);
    }
  }
  static std::vector<int> all_devices() {
    std::vector<int> d;
    for (int i = 0, n = device_count(); i < n; ++i) d.push_back(i);
    return d;
  }
  size_t size() const { return n_; }
  size_t shards() const { return shards_.size(); }
  shard_type &shard(size_t r) { return shards_[r]; }
  const shard_type &shard(size_t r) const { return shards_[r]; }
  size_t first(size_t r) const { return first_[r]; }
  size_t count(size_t r) const { return shards_[r].size(); }

  // host array <-> shards: every device moves its own slice (n independent PCIe streams), then one wait for all
  void upload(const P *host) {
    for (size_t r = 0; r < shards(); ++r)
      if (count(r)) detail::check(shards_[r].ctx(), nflhip_memcpy_h2d(shards_[r].ctx(), shards_[r].data(), host[first_[r]].cdata(),
                                                                       shards_[r].bytes(), shards_[r].queue()), "upload");
    sync();
  }
  void download(P *host) const {
    for (size_t r = 0; r < shards(); ++r)
      if (count(r)) detail::check(shards_[r].ctx(), nflhip_memcpy_d2h(shards_[r].ctx(), host[first_[r]].data(), shards_[r].data(),
                                                                       shards_[r].bytes(), shards_[r].queue()), "download");
    sync();
  }
  // a batch resident on ONE device <-> shards: peer-to-peer copies, each on the receiving / sending peer's stream
  void scatter(const shard_type &full) { move(const_cast<shard_type &>(full), true); }
  void gather(shard_type &full) const { const_cast<sharded_batch *>(this)->move(full, false); }

  void sync() const { for (auto &s : shards_) s.sync(); }

  // the batch operations of device_batch, one asynchronous call per shard
  void ntt_pow_phi() { for (auto &s : shards_) if (s.size()) s.ntt_pow_phi(); }
  void invntt_pow_invphi() { for (auto &s : shards_) if (s.size()) s.invntt_pow_invphi(); }
  void assign(int op, const sharded_batch &a, const sharded_batch &b) {
    same_split(a); same_split(b);
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].assign(op, a.shards_[r], b.shards_[r]);
  }
  void assign_mul_shoup(const sharded_batch &a, const sharded_batch &b, const sharded_batch &bprime) {
    same_split(a); same_split(b); same_split(bprime);
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].assign_mul_shoup(a.shards_[r], b.shards_[r], bprime.shards_[r]);
  }
  void assign_compute_shoup(const sharded_batch &b) {
    same_split(b);
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].assign_compute_shoup(b.shards_[r]);
  }
  void assign_polymul(const sharded_batch &a, const sharded_batch &b) {
    same_split(a); same_split(b);
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].assign_polymul(a.shards_[r], b.shards_[r]);
  }
  void assign_polymul_ntt(const sharded_batch &a, const sharded_batch &b_ntt) {
    same_split(a); same_split(b_ntt);
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].assign_polymul_ntt(a.shards_[r], b_ntt.shards_[r]);
  }
  void assign_program(const unsigned char *program, size_t len, const sharded_batch *const *operands, size_t nops) {
    if (nops > NFLHIP_EXPR_MAX_OPERANDS) throw std::runtime_error("nfl(hip): too many operands");
    for (size_t i = 0; i < nops; ++i) same_split(*operands[i]);
    for (size_t r = 0; r < shards(); ++r) {
      if (!count(r)) continue;
      const shard_type *ops[NFLHIP_EXPR_MAX_OPERANDS];
      for (size_t i = 0; i < nops; ++i) ops[i] = &operands[i]->shards_[r];
      shards_[r].assign_program(program, len, ops, nops);
    }
  }
  // in-place generation: ONE keystream for the logical batch, every shard reads its own positions of it
  void set(uniform const &u) {
    const uint64_t sid = u.seeded ? 0 : detail::sampler::get().next++;
    for (size_t r = 0; r < shards(); ++r) {
      if (!count(r)) continue;
      if (u.seeded) shards_[r].set(u, first_[r]);
      else shards_[r].set_at(u, first_[r], sid);
    }
  }
  void set(non_uniform const &m) { set_shards(m); }
  void set(ZO_dist const &m) { set_shards(m); }
  void set(hwt_dist const &m) { set_shards(m); }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, value_type, _lu_depth> const &m) { set_shards(m); }
  // one polynomial replicated over every shard
  void fill(const P &one) { for (auto &s : shards_) if (s.size()) s.fill(one); }

  // the reference's `==` / `!=` over the whole array ("any lane", ops.hpp:81-117): any shard
  bool any_equal(const sharded_batch &o) const {
    same_split(o);
    bool r = false;
    for (size_t k = 0; k < shards(); ++k) if (count(k)) r = shards_[k].any_equal(o.shards_[k]) || r;
    return r;
  }
  bool any_differs(const sharded_batch &o) const {
    same_split(o);
    bool r = false;
    for (size_t k = 0; k < shards(); ++k) if (count(k)) r = shards_[k].any_differs(o.shards_[k]) || r;
    return r;
  }
  // per-shard digests (positions counted in the WHOLE batch) and their sum = the digest of the batch on one device
  std::vector<uint64_t> digests() const {
    std::vector<uint64_t> d(shards());
    for (size_t r = 0; r < shards(); ++r) d[r] = count(r) ? shards_[r].digest(first_[r]) : 0;
    return d;
  }
  uint64_t digest() const {
    uint64_t s = 0;
    for (uint64_t d : digests()) s += d;
    return s;
  }

 private:
  template <class D> void set_shards(D const &m) {
    const uint64_t sid = detail::sampler::get().next++;
    for (size_t r = 0; r < shards(); ++r) if (count(r)) shards_[r].set_at(m, first_[r], sid);
  }
  void same_split(const sharded_batch &o) const {
    if (o.n_ != n_ || o.shards() != shards()) throw std::runtime_error("nfl(hip): batch size mismatch");
    for (size_t r = 0; r < shards(); ++r)
      if (o.shards_[r].ctx() != shards_[r].ctx()) throw std::runtime_error("nfl(hip): the batches of one operation must be split over the same devices");
  }
  void move(shard_type &full, bool to_shards) {
    if (full.size() != n_) throw std::runtime_error("nfl(hip): batch size mismatch");
    std::vector<nflhip_ctx *> ctxs;
    std::vector<void *> ptrs, streams;
    int root = -1;
    for (size_t r = 0; r < shards(); ++r) {
      ctxs.push_back(shards_[r].ctx());
      ptrs.push_back(shards_[r].data());
      streams.push_back(shards_[r].queue());
      if (shards_[r].ctx() == full.ctx()) root = int(r);
    }
    if (root < 0) throw std::runtime_error("nfl(hip): the whole batch must live on one of the shards' devices");
    const int rc = to_shards ? nflhip_scatter_local_dev(ctxs.data(), int(ctxs.size()), ptrs.data(), root, full.data(), n_, streams.data())
                             : nflhip_gather_local_dev(ctxs.data(), int(ctxs.size()), full.data(), root, ptrs.data(), n_, streams.data());
    detail::check(full.ctx(), rc, to_shards ? "scatter" : "gather");
  }
  size_t n_;
  std::vector<size_t> first_;
  std::vector<shard_type> shards_;
};

}  // namespace nfl

#endif  // NFL_HIP_NFL_HPP
