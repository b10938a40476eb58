// nfl_hip/expr.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// expression templates (ops.hpp): trees over poly / poly_p lowered to postfix programs.
#ifndef NFL_HIP_EXPR_HPP
#define NFL_HIP_EXPR_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {

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
  // host-pointer program limits (<= 4 distinct leaves, stack depth <= 4), else one pass per node
  void eval(poly_type &out) const {
    program pr;
    lower(pr);
    if (pr.ok && opcode<Op>::value >= 0 && pr.noperands <= 4) {
      const void *h[4] = {nullptr, nullptr, nullptr, nullptr};
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
}  // namespace nfl
#endif  // NFL_HIP_EXPR_HPP
