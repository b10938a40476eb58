// nfl_hip/nfl.hpp -- header-only host surface of the MI355X NTT polynomial-ring engine.
//
// Keeps the template surface of the reference's nfl::poly<T, Degree, NbModuli>
// (include/nfl/poly.hpp:82-352) and its expression-template operators
// (include/nfl/ops.hpp:18-97, 249-277) so existing callers compile unchanged,
// but every whole-polynomial operation is forwarded through the C ABI of
// include/nflhip.h to hand-written HIP kernels.  There is no CPU arithmetic in
// this header: without libnflhip.so + a GPU every operation throws
// std::runtime_error (the reference's error convention: core.hpp:111-115).
//
// What is kept (same names, argument meaning, error behaviour):
//   storage layout T _data[NbModuli*Degree], 32-byte aligned, modulus-major  poly.hpp:87-88,156-157
//   ctors / set(): value, initializer_list, iterator range (+reduce_coeffs)   core.hpp:64-137
//   operator()(cm,i), begin/end, data(), get_modulus, degree/nmoduli/nbits    poly.hpp:142-162
//   ntt_pow_phi(), invntt_pow_invphi()                                        poly.hpp:167-168
//   operator+ - * == !=, shoup(a*b,b'), compute_shoup(b), nested expressions  poly.hpp:346-352
//   explicit operator bool on polys, implicit on == / != expressions         core.hpp:39-43, ops.hpp:81-95
//   serialize_manually / deserialize_manually                                 poly.hpp:180-185
//   nfl::add / sub / mul                                                      poly.hpp:314-332
// What differs, on purpose:
//   * tables live in a lazily created per-(T,Degree,NbModuli) device context,
//     never at static-init time (the reference's `static core base`, poly.hpp:247);
//   * CRT lift/project exchange little-endian 64-bit limb vectors
//     (poly2limbs / limbs2poly == the mpz_export/mpz_import image of
//     poly2mpz / mpz2poly, gmp.hpp:183-219); with NFL_HIP_WITH_GMP defined before
//     inclusion the reference's GMP-typed surface is there as well (poly.hpp:249-307):
//     mpz_t / mpz_class constructors, set_mpz, operator=, poly2mpz / mpz2poly on
//     std::array<mpz_t, Degree>, moduli_product / modulus_shoup / lifting_integers;
//   * nfl::batch::* operate on contiguous arrays of polys (dense
//     [batch][NbModuli][Degree], as tests/tools.h:6-17 allocates) in ONE device
//     pass -- the per-poly members stay for source compatibility;
//   * nfl::uniform is a seeded counter-based generator with the reference's
//     mask-then-subtract rule (core.hpp:165-176), not a CSPRNG.
#ifndef NFL_HIP_NFL_HPP
#define NFL_HIP_NFL_HPP

#include <algorithm>
#include <array>
#include <atomic>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

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

// ---------------------------------------------------------------- params<T> (params.hpp:11-119)
template <class T> struct params;
template <> struct params<uint16_t> {
  typedef uint16_t value_type;
  typedef int16_t signed_value_type;
  typedef uint32_t greater_value_type;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U16_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U16_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 16;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U16_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U16_KMAX_LOG2;
  static const value_type *P() { return NFLHIP_U16_P; }
  static const value_type *roots() { return NFLHIP_U16_ROOTS; }
  static const value_type *invkmax() { return NFLHIP_U16_INVKMAX; }
};
template <> struct params<uint32_t> {
  typedef uint32_t value_type;
  typedef int32_t signed_value_type;
  typedef uint64_t greater_value_type;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U32_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U32_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 32;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U32_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U32_KMAX_LOG2;
  static const value_type *P() { return NFLHIP_U32_P; }
  static const value_type *roots() { return NFLHIP_U32_ROOTS; }
  static const value_type *invkmax() { return NFLHIP_U32_INVKMAX; }
};
template <> struct params<uint64_t> {
  typedef uint64_t value_type;
  typedef int64_t signed_value_type;
  typedef unsigned __int128 greater_value_type;
  static constexpr unsigned int kMaxNbModuli = NFLHIP_U64_NMODULI;
  static constexpr unsigned int kModulusBitsize = NFLHIP_U64_MODULUS_BITS;
  static constexpr unsigned int kModulusRepresentationBitsize = 64;
  static constexpr unsigned int kMaxPolyDegree = 1u << NFLHIP_U64_KMAX_LOG2;
  static constexpr int kMaxLog2 = NFLHIP_U64_KMAX_LOG2;
  static const value_type *P() { return NFLHIP_U64_P; }
  static const value_type *roots() { return NFLHIP_U64_ROOTS; }
  static const value_type *invkmax() { return NFLHIP_U64_INVKMAX; }
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

// The process-wide sampler state: the counterpart of fastrandombytes' static key and nonce
// (lib/prng/fastrandombytes.cpp:17-37).  The key is drawn from the OS once; every sampling call takes the next
// 64-bit stream id.  nfl::set_sampler_key() pins both for reproducible runs.
struct sampler {
  unsigned char key[32];
  std::atomic<uint64_t> next;
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

// One device context per (T, Degree, NbModuli): the replacement of the
// reference's static `core base` / `GMP gmp` members (poly.hpp:247, 275), created
// on first use (function-local static => thread-safe, never before main()).
template <class T, size_t Degree, size_t NbModuli> struct context {
  nflhip_ctx *ctx;
  context() : ctx(nullptr) {
    static_assert(NbModuli <= params<T>::kMaxNbModuli, "not enough moduli of this size (see nflhip_params.h)");
    static_assert(Degree <= params<T>::kMaxPolyDegree, "degree is not lower or equal than kMaxPolyDegree");
    int rc = nflhip_ctx_create(&ctx, 0, int(sizeof(T) * 8), Degree, NbModuli, params<T>::P(), params<T>::roots(),
                               params<T>::invkmax(), params<T>::kMaxLog2);
    if (rc != NFLHIP_OK) throw std::runtime_error(std::string("nfl(hip): context: ") + nflhip_last_error(nullptr));
  }
  ~context() { nflhip_ctx_destroy(ctx); }
  context(const context &) = delete;
  context &operator=(const context &) = delete;
  static nflhip_ctx *get() {
    static context c;
    return c.ctx;
  }
};

inline uint64_t splitmix64_at(uint64_t seed, int operand, uint64_t g) {
  uint64_t z = (seed ^ (uint64_t(operand) << 62)) + (g + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace detail

/* pin the sampler state: `key` (32 bytes) and the id of the next keystream -- reproducible runs */
inline void set_sampler_key(const unsigned char key[32], uint64_t next_stream = 0) {
  detail::sampler &s = detail::sampler::get();
  std::memcpy(s.key, key, 32);
  s.next.store(next_stream);
}

/* FastGaussianNoise<in_class, out_class, _lu_depth>(sigma, security, samples, center) -- same constructor as
 * FastGaussianNoise.hpp:163-204.  The reference builds byte-indexed lookup tables over MPFR barriers; here the object
 * only carries the parameters and owns one cumulative table per device context (built on first use with the
 * reference's tail bound and bit precision, sampled by inversion on the GPU).  in_class / _lu_depth only tuned the
 * reference's lookup and are accepted for source compatibility. */
template <class in_class, class out_class, unsigned _lu_depth> class FastGaussianNoise {
 public:
  FastGaussianNoise(double sigma, unsigned int security, unsigned int samples, double center_d = 0, bool /*verbose*/ = false)
      : sigma_(sigma), security_(security), samples_(samples), center_(center_d) {
    static_assert(_lu_depth == 1 || _lu_depth == 2, "_lu_depth must be 1 or 2 (FastGaussianNoise.hpp:214)");
  }
  FastGaussianNoise(FastGaussianNoise const &) = delete;
  FastGaussianNoise &operator=(FastGaussianNoise const &) = delete;
  ~FastGaussianNoise() {
    for (auto &kv : tables_) nflhip_gauss_destroy(kv.first, kv.second);
  }
  const nflhip_gauss *table(nflhip_ctx *ctx) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = tables_.find(ctx);
    if (it != tables_.end()) return it->second;
    nflhip_gauss *g = nullptr;
    detail::check(ctx, nflhip_gauss_create(ctx, &g, sigma_, security_, samples_, center_), "FastGaussianNoise");
    tables_[ctx] = g;
    return g;
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
};

template <class in_class, class out_class, unsigned _lu_depth> struct gaussian {
  FastGaussianNoise<in_class, out_class, _lu_depth> *fg_prng;
  uint64_t amplifier;
  gaussian(FastGaussianNoise<in_class, out_class, _lu_depth> *prng) : fg_prng{prng}, amplifier{1} {}
  gaussian(FastGaussianNoise<in_class, out_class, _lu_depth> *prng, uint64_t amp) : fg_prng{prng}, amplifier{amp} {}
};

template <class T, size_t Degree, size_t NbModuli> class poly;

// ---------------------------------------------------------------- expression templates (ops.hpp:52-97)
template <class T, size_t Degree, size_t NbModuli> class poly;
template <class T, size_t Degree, size_t NbModuli> class poly_p;

namespace ops {

struct addmod { static constexpr int code = NFLHIP_OP_ADD; };
struct submod { static constexpr int code = NFLHIP_OP_SUB; };
struct mulmod { static constexpr int code = NFLHIP_OP_MUL; };
struct mulmod_shoup { static constexpr int code = NFLHIP_OP_MUL_SHOUP; };
struct compute_shoup { static constexpr int code = NFLHIP_OP_COMPUTE_SHOUP; };
struct eqmod {};
struct neqmod {};
struct shoup_marker {};

template <class Op, class... Args> struct expr;

// leaves of an expression tree: a poly, or a poly_p handle (poly_p.hpp:11-204) standing for its polynomial
template <class A, class Poly> struct is_leaf : std::is_same<A, Poly> {};
template <class T, size_t D, size_t M> struct is_leaf<poly_p<T, D, M>, poly<T, D, M>> : std::true_type {};
template <class T, size_t D, size_t M> inline const poly<T, D, M> &leaf(const poly<T, D, M> &p) { return p; }
template <class T, size_t D, size_t M> inline const poly<T, D, M> &leaf(const poly_p<T, D, M> &p) { return p.poly_obj(); }

// postfix program of one expression tree (include/nflhip.h, NFLHIP_EXPR_*), built when the tree is
// assigned: distinct leaf polys become operands 0..2, every node appends its opcode.
struct program {
  unsigned char code[NFLHIP_EXPR_MAX_LEN];
  size_t len = 0;
  const void *operand[3] = {nullptr, nullptr, nullptr};
  size_t noperands = 0;
  int depth = 0;
  bool ok = true;
  void push_leaf(const void *p) {
    size_t k = 0;
    while (k < noperands && operand[k] != p) ++k;
    if (k == noperands) {
      if (noperands == 3) { ok = false; return; }
      operand[noperands++] = p;
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
template <class Op> struct opcode { static constexpr int value = -1; static constexpr int delta = 0; };
template <> struct opcode<addmod> { static constexpr int value = NFLHIP_EXPR_ADD; static constexpr int delta = -1; };
template <> struct opcode<submod> { static constexpr int value = NFLHIP_EXPR_SUB; static constexpr int delta = -1; };
template <> struct opcode<mulmod> { static constexpr int value = NFLHIP_EXPR_MUL; static constexpr int delta = -1; };
template <> struct opcode<mulmod_shoup> { static constexpr int value = NFLHIP_EXPR_MUL_SHOUP; static constexpr int delta = -2; };
template <> struct opcode<compute_shoup> { static constexpr int value = NFLHIP_EXPR_COMPUTE_SHOUP; static constexpr int delta = 0; };

template <class Op, class... Args> struct expr {
  std::tuple<Args const &...> args;
  explicit expr(Args const &... a) : args(a...) {}
  typedef typename std::remove_cv<typename std::remove_reference<
      decltype(std::get<0>(std::declval<std::tuple<Args const &...>>()))>::type>::type first_type;
  typedef typename first_type::value_type value_type;
  typedef typename first_type::poly_type poly_type;
  static constexpr size_t degree = first_type::degree;
  static constexpr size_t nmoduli = first_type::nmoduli;

  // evaluate this node into `out`: the whole tree in ONE fused device pass when it fits the
  // postfix program limits (<= 3 distinct polys, stack depth <= 4), else one pass per node
  void eval(poly_type &out) const {
    program pr;
    lower(pr);
    if (pr.ok && opcode<Op>::value >= 0 && out.apply_program(pr)) return;
    eval_impl(out, std::integral_constant<size_t, sizeof...(Args)>());
  }
  // append this subtree to a postfix program
  void lower(program &pr) const {
    lower_args(pr, std::integral_constant<size_t, 0>());
    if (opcode<Op>::value < 0) pr.ok = false;
    else pr.emit((unsigned char)opcode<Op>::value, opcode<Op>::delta);
  }

  // expr::operator bool (ops.hpp:81-95): true as soon as ONE lane of the value is non-zero
  operator bool() const { return truth(Op()); }  // (implicit, as in the reference: `ok &= (a == b);` compiles)

 private:
  template <class A> static void lower_one(const A &a, program &pr, std::true_type) { pr.push_leaf(leaf(a).cdata()); }
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
  template <class O> bool truth(O) const {  // arithmetic expression: any non-zero word
    poly_type *t = poly_type::make_temp();
    eval(*t);
    const bool r = bool(*t);
    poly_type::drop_temp(t);
    return r;
  }
  bool cmp(bool want_eq) const {
    poly_type *t0 = poly_type::make_temp(), *t1 = poly_type::make_temp();
    const poly_type &a = mat(std::get<0>(args), *t0);
    const poly_type &b = mat(std::get<1>(args), *t1);
    const bool r = poly_type::any_cmp(a, b, want_eq);
    poly_type::drop_temp(t0);
    poly_type::drop_temp(t1);
    return r;
  }
  bool truth(eqmod) const { return cmp(true); }    // "any lane equal"  (the reference's quirk)
  bool truth(neqmod) const { return cmp(false); }  // "any lane differs"
};

}  // namespace ops

// ---------------------------------------------------------------- poly (poly.hpp:82-310)
template <class T, size_t Degree, size_t NbModuli> class poly {
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
  static constexpr size_t degree = Degree;
  static constexpr size_t nmoduli = NbModuli;
  static constexpr size_t nbits = params<T>::kModulusBitsize;
  static constexpr size_t aggregated_modulus_bit_size = NbModuli * nbits;

  /* constructors (core.hpp:64-84) */
  poly() { set(value_type(0)); }
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
  // same contract as core.hpp:101-137: fewer than `degree` values are zero-padded and
  // replicated across moduli; otherwise exactly degree*nmoduli values are required
  template <class It> void set(It first, It last, bool reduce_coeffs = true) {
    const size_t size = size_t(std::distance(first, last));
    if (size > degree && size != degree * nmoduli)
      throw std::runtime_error("core: CRITICAL, initializer of size above degree but not equal to nmoduli*degree");
    T *iter = begin();
    It viter = first;
    for (size_t cm = 0; cm < nmoduli; cm++) {
      const value_type p = get_modulus(cm);
      if (size != degree * nmoduli) viter = first;
      size_t i = 0;
      for (; i < degree && viter != last; ++i, ++viter, ++iter) *iter = reduce_coeffs ? value_type((*viter) % p) : value_type(*viter);
      for (; i < degree; ++i, ++iter) *iter = 0;
    }
  }
  // mask-then-subtract rule of core.hpp:165-176: on the device's keystream (fresh per call), or on a seeded
  // counter stream for `uniform(seed)`
  void set(uniform const &u) {
    if (!u.seeded) {
      sample(NFLHIP_DIST_UNIFORM, 0, 1, "set(uniform)");
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
  void set(ZO_dist const &m) { sample(NFLHIP_DIST_ZO, m.rho, 1, "set(ZO_dist)"); }
  void set(hwt_dist const &m) { sample(NFLHIP_DIST_HWT, m.hwt, 1, "set(hwt_dist)"); }
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
  static value_type get_modulus(size_t n) { return params<T>::P()[n]; }

  /* ntt stuff - public API (poly.hpp:167-168) */
  void ntt_pow_phi() { detail::check(ctx(), nflhip_ntt_fwd(ctx(), _data, 1), "ntt_pow_phi"); }
  void invntt_pow_invphi() { detail::check(ctx(), nflhip_ntt_inv(ctx(), _data, 1), "invntt_pow_invphi"); }

  /* manual serializers (poly.hpp:180-185): raw little-endian words */
  void serialize_manually(std::ostream &os) { os.write(reinterpret_cast<char *>(_data), N * sizeof(T)); }
  void deserialize_manually(std::istream &is) { is.read(reinterpret_cast<char *>(_data), N * sizeof(T)); }
  // cereal hook, identical to the reference's (poly.hpp:186-190): works with any archive type that accepts a C array
  template <class Archive> void serialize(Archive &archive) { archive(_data); }

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
  void set_mpz(mpz_t const &v) { set_mpz(&v, &v + 1); }
  void set_mpz(std::array<mpz_t, Degree> const &values) { set_mpz(values.begin(), values.end()); }
  poly &operator=(mpz_t const &v) { set_mpz(v); return *this; }
  poly &operator=(std::array<mpz_t, Degree> const &values) { set_mpz(values); return *this; }
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
    const size_t size = size_t(std::distance(first, last));
    if (size > degree && size != degree * nmoduli)
      throw std::runtime_error("gmp: CRITICAL, initializer of size above degree but not equal to nmoduli*degree");
    T *iter = begin();
    It viter = first;
    for (size_t cm = 0; cm < nmoduli; cm++) {
      const value_type p = get_modulus(cm);
      if (size != degree * nmoduli) viter = first;
      size_t i = 0;
      for (; i < degree && viter != last; ++i, ++viter, ++iter) *iter = value_type(mpz_fdiv_ui(detail::as_mpz(*viter), p));
      for (; i < degree; ++i, ++iter) *iter = 0;
    }
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
  void sample(int dist, uint64_t p0, uint64_t p1, const char *what) {
    detail::sampler &s = detail::sampler::get();
    detail::check(ctx(), nflhip_sample(ctx(), _data, 1, dist, p0, p1, s.key, s.next++), what);
  }
  void apply(int op, const poly &a, const poly &b, const poly &bp) {
    detail::check(ctx(), nflhip_pointwise(ctx(), op, _data, a._data, b._data, bp._data, 1), "operator=(expr)");
  }
  // fused tree evaluation; false = the engine declined (tiny rows): the caller goes node by node
  bool apply_program(const ops::program &pr) {
    const int rc = nflhip_eval(ctx(), _data, pr.operand, pr.noperands, pr.code, pr.len, 1);
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
    return new (mem) poly();
  }
  static void drop_temp(poly *p) {
    p->~poly();
    free(p);
  }
} __attribute__((aligned(32)));

// ---------------------------------------------------------------- operators (poly.hpp:346-352, ops.hpp:18-45)
namespace ops {
template <class X> struct is_node : std::false_type {};
template <class T, size_t D, size_t M> struct is_node<poly<T, D, M>> : std::true_type {};
template <class T, size_t D, size_t M> struct is_node<poly_p<T, D, M>> : std::true_type {};
template <class Op, class... A> struct is_node<expr<Op, A...>> : std::true_type {};
// == and != on a poly_p are its own members (poly_p.hpp:112-140); everything else is generic
template <class X> struct is_cmp_node : is_node<X> {};
template <class T, size_t D, size_t M> struct is_cmp_node<poly_p<T, D, M>> : std::false_type {};
}  // namespace ops

#define NFL_HIP_BINARY(SYM, NAME)                                                                          \
  template <class A, class B>                                                                              \
  typename std::enable_if<ops::is_node<A>::value && ops::is_node<B>::value, ops::expr<ops::NAME, A, B>>::type SYM( \
      A const &a, B const &b) {                                                                            \
    static_assert(std::is_same<typename A::poly_type, typename B::poly_type>::value, "correct type combination"); \
    return ops::expr<ops::NAME, A, B>(a, b);                                                               \
  }
NFL_HIP_BINARY(operator-, submod)
NFL_HIP_BINARY(operator+, addmod)
NFL_HIP_BINARY(operator*, mulmod)
#undef NFL_HIP_BINARY
#define NFL_HIP_COMPARE(SYM, NAME)                                                                               \
  template <class A, class B>                                                                                    \
  typename std::enable_if<ops::is_cmp_node<A>::value && ops::is_node<B>::value, ops::expr<ops::NAME, A, B>>::type SYM( \
      A const &a, B const &b) {                                                                                  \
    static_assert(std::is_same<typename A::poly_type, typename B::poly_type>::value, "correct type combination"); \
    return ops::expr<ops::NAME, A, B>(a, b);                                                                     \
  }
NFL_HIP_COMPARE(operator==, eqmod)
NFL_HIP_COMPARE(operator!=, neqmod)
#undef NFL_HIP_COMPARE

template <class A> typename std::enable_if<ops::is_node<A>::value, ops::expr<ops::compute_shoup, A>>::type compute_shoup(A const &a) {
  return ops::expr<ops::compute_shoup, A>(a);
}
// shoup(a*b, b') is rewritten into mulmod_shoup(a, b, b') (ops.hpp:267-277); anything else
// is a compile-time error, as in the reference (ops.hpp:153-163)
template <class A0, class A1, class B>
ops::expr<ops::mulmod_shoup, A0, A1, B> shoup(ops::expr<ops::mulmod, A0, A1> const &prod, B const &bprime) {
  return ops::expr<ops::mulmod_shoup, A0, A1, B>(std::get<0>(prod.args), std::get<1>(prod.args), bprime);
}

// ---------------------------------------------------------------- poly_p (poly_p.hpp:11-204)
// Copy-on-write handle to a heap-allocated, 32-byte aligned poly: same members as the reference's class.
template <class T, size_t Degree, size_t NbModuli> class poly_p {
 public:
  typedef poly<T, Degree, NbModuli> poly_type;
  using value_type = typename poly_type::value_type;
  using greater_value_type = typename poly_type::greater_value_type;
  static constexpr size_t nmoduli = poly_type::nmoduli;
  static constexpr size_t degree = poly_type::degree;
  static constexpr size_t nbits = poly_type::nbits;
  static constexpr size_t aggregated_modulus_bit_size = poly_type::aggregated_modulus_bit_size;

 private:
  typedef std::shared_ptr<poly_type> ptr_type;
  template <class... Args> static ptr_type make_pointer(Args &&... args) {
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(poly_type)) != 0) throw std::bad_alloc();
    poly_type *p = nullptr;
    try {
      p = new (mem) poly_type(std::forward<Args>(args)...);
    } catch (...) {
      free(mem);
      throw;
    }
    return ptr_type(p, [](poly_type *q) { q->~poly_type(); free(q); });
  }
  void detach() {
    if (!_p.unique()) _p = make_pointer(*_p);
  }
  ptr_type _p;

 public:
  poly_p(poly_p const &o) : _p(o._p) {}
  poly_p(poly_p &o) : _p(const_cast<poly_p const &>(o)._p) {}
  poly_p(poly_p &&o) : _p(std::move(o._p)) {}
  template <class... Args> poly_p(Args &&... args) : _p(make_pointer(std::forward<Args>(args)...)) {}
  poly_p(poly_type const &) = delete;
  poly_p(poly_type &&) = delete;

  poly_type &poly_obj() {
    detach();
    return *_p;
  }
  poly_type const &poly_obj() const { return *_p; }

  template <class O> poly_p &operator=(O &&o) {
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

  bool operator==(poly_p const &o) const { return _p.get() == o._p.get() ? true : bool(poly_obj() == o.poly_obj()); }
  bool operator!=(poly_p const &o) const { return _p.get() == o._p.get() ? false : bool(poly_obj() != o.poly_obj()); }
  template <class O> bool operator==(O const &o) const { return bool(poly_obj() == o); }
  template <class O> bool operator!=(O const &o) const { return bool(poly_obj() != o); }

  value_type &operator()(size_t cm, size_t i) { return poly_obj()(cm, i); }
  value_type const &operator()(size_t cm, size_t i) const { return poly_obj()(cm, i); }
  static constexpr value_type get_modulus(size_t n) { return poly_type::get_modulus(n); }

  void ntt_pow_phi() { poly_obj().ntt_pow_phi(); }
  void invntt_pow_invphi() { poly_obj().invntt_pow_invphi(); }
  void serialize_manually(std::ostream &os) { poly_obj().serialize_manually(os); }
  void deserialize_manually(std::istream &is) { poly_obj().deserialize_manually(is); }
  template <class Archive> void serialize(Archive &archive) { archive(poly_obj()); }

  void set(value_type v, bool reduce_coeffs = true) { poly_obj().set(v, reduce_coeffs); }
  void set(uniform const &m) { poly_obj().set(m); }
  void set(non_uniform const &m) { poly_obj().set(m); }
  void set(ZO_dist const &m) { poly_obj().set(m); }
  void set(hwt_dist const &m) { poly_obj().set(m); }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, T, _lu_depth> const &m) { poly_obj().set(m); }
  void set(std::initializer_list<value_type> values, bool reduce_coeffs = true) { poly_obj().set(values, reduce_coeffs); }
  template <class It> void set(It first, It last, bool reduce_coeffs = true) { poly_obj().set(first, last, reduce_coeffs); }
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
template <class P> class device_batch {
 public:
  typedef typename P::value_type value_type;
  explicit device_batch(size_t count) : n_(count), d_(nullptr) {
    static_assert(sizeof(P) == P::degree * P::nmoduli * sizeof(value_type), "dense poly array");
    detail::check(P::ctx(), nflhip_malloc(P::ctx(), &d_, bytes()), "device_batch");
  }
  device_batch(const P *host, size_t count) : device_batch(count) { upload(host); }
  ~device_batch() { if (d_) nflhip_free(P::ctx(), d_); }
  device_batch(const device_batch &) = delete;
  device_batch &operator=(const device_batch &) = delete;
  device_batch(device_batch &&o) noexcept : n_(o.n_), d_(o.d_) { o.d_ = nullptr; }

  size_t size() const { return n_; }
  size_t bytes() const { return n_ * sizeof(P); }
  void *data() { return d_; }
  const void *data() const { return d_; }

  void upload(const P *host) {
    detail::check(P::ctx(), nflhip_memcpy_h2d(P::ctx(), d_, host->cdata(), bytes(), nullptr), "upload");
    sync();
  }
  void download(P *host) const {
    detail::check(P::ctx(), nflhip_memcpy_d2h(P::ctx(), host->data(), d_, bytes(), nullptr), "download");
    sync();
  }
  void sync() const { detail::check(P::ctx(), nflhip_stream_sync(P::ctx(), nullptr), "sync"); }

  // same names and meaning as the poly members (poly.hpp:167-168), over the whole batch
  void ntt_pow_phi() { detail::check(P::ctx(), nflhip_ntt_fwd_dev(P::ctx(), d_, n_, nullptr), "ntt_pow_phi"); }
  void invntt_pow_invphi() { detail::check(P::ctx(), nflhip_ntt_inv_dev(P::ctx(), d_, n_, nullptr), "invntt_pow_invphi"); }
  // *this = op(a, b[, b'])  (NFLHIP_OP_*); aliasing allowed
  void assign(int op, const device_batch &a, const device_batch &b) {
    same_size(a); same_size(b);
    detail::check(P::ctx(), nflhip_pointwise_dev(P::ctx(), op, d_, a.d_, b.d_, nullptr, n_, nullptr), "pointwise");
  }
  void assign_mul_shoup(const device_batch &a, const device_batch &b, const device_batch &bprime) {
    same_size(a); same_size(b); same_size(bprime);
    detail::check(P::ctx(), nflhip_pointwise_dev(P::ctx(), NFLHIP_OP_MUL_SHOUP, d_, a.d_, b.d_, bprime.d_, n_, nullptr),
                  "mulmod_shoup");
  }
  void assign_compute_shoup(const device_batch &b) {
    same_size(b);
    detail::check(P::ctx(), nflhip_pointwise_dev(P::ctx(), NFLHIP_OP_COMPUTE_SHOUP, d_, b.d_, nullptr, nullptr, n_, nullptr),
                  "compute_shoup");
  }
  // *this = INTT(NTT(a) (.) NTT(b)), the fused metric path
  void assign_polymul(const device_batch &a, const device_batch &b) {
    same_size(a); same_size(b);
    detail::check(P::ctx(), nflhip_polymul_dev(P::ctx(), d_, a.d_, b.d_, n_, nullptr), "polymul");
  }
  // the same with b already in NTT form (keys of the LWE demo stay transformed, tests/nfllib_demo_main_op.cpp:26-46)
  void assign_polymul_ntt(const device_batch &a, const device_batch &b_ntt) {
    same_size(a); same_size(b_ntt);
    detail::check(P::ctx(), nflhip_polymul_ntt_dev(P::ctx(), d_, a.d_, b_ntt.d_, n_, nullptr), "polymul_ntt");
  }
  // CRT lift / project of the whole resident batch (gmp.hpp:183-219): out[(b*degree + i)*L .. +L) = little-endian limbs
  // of X_{b,i} in [0, Q), L = P::crt_limbs(); limbs2poly takes L_in limbs per coefficient
  void poly2limbs(std::vector<uint64_t> &out) const {
    const size_t words = n_ * P::degree * P::crt_limbs();
    out.assign(words, 0);
    void *dl = nullptr;
    detail::check(P::ctx(), nflhip_malloc(P::ctx(), &dl, words * sizeof(uint64_t)), "device allocation");
    int rc = nflhip_crt_lift_dev(P::ctx(), static_cast<uint64_t *>(dl), d_, n_, nullptr);
    if (rc == 0) rc = nflhip_memcpy_d2h(P::ctx(), out.data(), dl, words * sizeof(uint64_t), nullptr);
    if (rc == 0) rc = nflhip_stream_sync(P::ctx(), nullptr);
    nflhip_free(P::ctx(), dl);
    detail::check(P::ctx(), rc, "poly2mpz");
  }
  void limbs2poly(const uint64_t *limbs, size_t L_in) {
    const size_t words = n_ * P::degree * L_in;
    void *dl = nullptr;
    detail::check(P::ctx(), nflhip_malloc(P::ctx(), &dl, words * sizeof(uint64_t)), "device allocation");
    int rc = nflhip_memcpy_h2d(P::ctx(), dl, limbs, words * sizeof(uint64_t), nullptr);
    if (rc == 0) rc = nflhip_crt_project_dev(P::ctx(), d_, static_cast<const uint64_t *>(dl), L_in, n_, nullptr);
    if (rc == 0) rc = nflhip_stream_sync(P::ctx(), nullptr);
    nflhip_free(P::ctx(), dl);
    detail::check(P::ctx(), rc, "mpz2poly");
  }
  // fused postfix expression over up to NFLHIP_EXPR_MAX_OPERANDS resident batches
  void assign_program(const unsigned char *program, size_t len, const device_batch *const *operands, size_t count) {
    const void *ptr[NFLHIP_EXPR_MAX_OPERANDS];
    if (count > NFLHIP_EXPR_MAX_OPERANDS) throw std::runtime_error("nfl(hip): too many operands");
    for (size_t i = 0; i < count; ++i) { same_size(*operands[i]); ptr[i] = operands[i]->d_; }
    detail::check(P::ctx(), nflhip_eval_dev(P::ctx(), d_, ptr, count, program, len, n_, nullptr), "eval");
  }
  // the random constructors over the whole resident batch (same tags as poly's; one keystream per call)
  void set(uniform const &u) {
    if (u.seeded) detail::check(P::ctx(), nflhip_fill_uniform_dev(P::ctx(), d_, 0, n_, u.seed, 0, nullptr), "set(uniform)");
    else sample(NFLHIP_DIST_UNIFORM, 0, 1, "set(uniform)");
  }
  void set(non_uniform const &m) { sample(NFLHIP_DIST_BOUNDED, m.upper_bound, m.amplifier, "set(non_uniform)"); }
  void set(ZO_dist const &m) { sample(NFLHIP_DIST_ZO, m.rho, 1, "set(ZO_dist)"); }
  void set(hwt_dist const &m) { sample(NFLHIP_DIST_HWT, m.hwt, 1, "set(hwt_dist)"); }
  template <class in_class, unsigned _lu_depth> void set(gaussian<in_class, value_type, _lu_depth> const &m) {
    detail::sampler &s = detail::sampler::get();
    detail::check(P::ctx(), nflhip_sample_gauss_dev(P::ctx(), d_, 0, n_, m.fg_prng->table(P::ctx()), m.amplifier, s.key,
                                                    s.next++, nullptr), "set(gaussian)");
  }
  // replicate one polynomial over the batch (a key shared by every ciphertext, ...)
  void fill(const P &one) {
    for (size_t k = 0; k < n_; ++k)
      detail::check(P::ctx(), nflhip_memcpy_h2d(P::ctx(), static_cast<char *>(d_) + k * sizeof(P), one.cdata(), sizeof(P), nullptr), "fill");
    sync();
  }
  bool any_equal(const device_batch &o) const { return cmp(o, true); }    // the reference's `a == b`
  bool any_differs(const device_batch &o) const { return cmp(o, false); } // the reference's `a != b`

 private:
  void same_size(const device_batch &o) const {
    if (o.n_ != n_) throw std::runtime_error("nfl(hip): batch size mismatch");
  }
  void sample(int dist, uint64_t p0, uint64_t p1, const char *what) {
    detail::sampler &s = detail::sampler::get();
    detail::check(P::ctx(), nflhip_sample_dev(P::ctx(), d_, 0, n_, dist, p0, p1, s.key, s.next++, nullptr), what);
  }
  bool cmp(const device_batch &o, bool want_eq) const {
    same_size(o);
    int r = 0;
    detail::check(P::ctx(), want_eq ? nflhip_any_eq_dev(P::ctx(), d_, o.d_, n_, &r, nullptr)
                                    : nflhip_any_neq_dev(P::ctx(), d_, o.d_, n_, &r, nullptr), "compare");
    return r != 0;
  }
  size_t n_;
  void *d_;
};

}  // namespace nfl

#endif  // NFL_HIP_NFL_HPP
