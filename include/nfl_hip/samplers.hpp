// nfl_hip/samplers.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// host PRNG entry points (fastrandombytes / randombytes), FastGaussianNoise and the gaussian tag.
#ifndef NFL_HIP_SAMPLERS_HPP
#define NFL_HIP_SAMPLERS_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {

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
    // the reference computes its table HERE (FastGaussianNoise.hpp:214-300); so does this one -- on the host, once per parameter set and
    // process, kept by the library -- and the first polynomial drawn in a context only uploads it.  (A parameter error surfaces at that draw.)
    (void)nflhip_gauss_table(sigma_, security_, samples_, center_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
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

}  // namespace nfl
#endif  // NFL_HIP_SAMPLERS_HPP
