// nfl_hip/poly.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// nfl::poly (the reference's inline host array), its GMP surface and the operators.
#ifndef NFL_HIP_POLY_HPP
#define NFL_HIP_POLY_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {

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

}  // namespace nfl
#endif  // NFL_HIP_POLY_HPP
