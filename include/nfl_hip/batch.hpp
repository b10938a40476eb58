// nfl_hip/batch.hpp -- part of the drop-in header; include <nfl_hip/nfl.hpp> (or the reference's names under include/nfl*).
// batch entry points, nfl::device_batch, nfl::sharded_batch.
#ifndef NFL_HIP_BATCH_HPP
#define NFL_HIP_BATCH_HPP
#ifndef NFL_HIP_NFL_HPP
#error "include <nfl_hip/nfl.hpp>: the parts depend on each other in its order"
#endif
namespace nfl {
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
    void *dst[3];
    uint64_t sid[3];
    bool one_table = true;
    for (int j = 0; j < nx; ++j) {
      dst[j] = buf + each * size_t(j);
      sid[j] = s.next++;
      one_table &= tab[j] == tab[0];
      xo[j].ptr = dst[j];
      xo[j].stride = 1;
      xo[j].format = fmt;
    }
    if (one_table) {   // the draws of one generator: one launch
      detail::check(ctx(), nflhip_sample_gauss_small_multi_dev(ctx(), dst, size_t(nx), fmt, n_, tab[0], amp, s.key, sid, nullptr, queue()),
                    "set(gaussian), compact");
    } else {
      for (int j = 0; j < nx; ++j)
        detail::check(ctx(), nflhip_sample_gauss_small_dev(ctx(), dst[j], fmt, 0, n_, tab[j], amp[j], s.key, sid[j], queue()), "set(gaussian), compact");
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
#endif  // NFL_HIP_BATCH_HPP
