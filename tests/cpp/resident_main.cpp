// tests/cpp/resident_main.cpp -- nfl::poly_p as a RESIDENT handle: the shared payload of
// the reference's copy-on-write handle (poly_p.hpp:11-204) lives in HBM; operator expressions, transforms, comparisons
// and the random constructors on handles run on the device and only poly_obj() / operator()(cm,i) / serialisation
// bring the value to the host.  Checks, with memcmp strength against the inline-storage poly path:
//   residency bits across every kind of access, copy-on-write (incl. device-to-device detach and the re-seat on a
//   whole-value overwrite), aliasing (a = a + b), mixed poly / poly_p trees, trees beyond the fused program,
//   comparisons in HBM, the zero handle -- and the reference's LWE demo (tests/nfllib_demo_main_op.cpp:26-58, 260-332)
//   written with plain poly_p operators, timed next to the same work on a resident batch (nfl::device_batch).
// Exit code 0 = all good.  Prints one JSON line with the rates.  Needs a GPU.
#include <nfl.hpp>

#include <chrono>
#include <cstdio>
#include <cstring>

#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
      return false;                                                      \
    }                                                                    \
  } while (0)

template <class P> struct Heap {
  P *p;
  template <class... A> explicit Heap(A &&... a) {
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
    p = new (mem) P(std::forward<A>(a)...);
  }
  ~Heap() { p->~P(); free(p); }
  P &operator*() { return *p; }
  P *operator->() { return p; }
};
template <class P> static bool same(const P &a, const P &b) {
  return std::memcmp(a.cdata(), b.cdata(), sizeof(typename P::value_type) * P::degree * P::nmoduli) == 0;
}

template <class T, size_t Degree, size_t NbModuli> static bool run_residency() {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  const poly_p a{nfl::uniform(11)}, b{nfl::uniform(12)}, d{nfl::uniform(13)};
  CHECK(a.resident() && b.resident());                 // the seeded constructor fills in HBM
  Heap<poly_t> A(nfl::uniform(11)), B(nfl::uniform(12)), D(nfl::uniform(13)), R;
  poly_p c = a * b + d;                                // one fused device pass, stays resident
  CHECK(c.resident());
  *R = *A * *B + *D;
  CHECK(same(const_cast<const poly_p &>(c).poly_obj(), *R));   // const access: value on the host AND still in HBM
  CHECK(const_cast<const poly_p &>(c).resident());
  // copy-on-write: the copy shares the payload until one side is written; transforms detach device-to-device
  poly_p e = c;
  CHECK(e.payload_id() == c.payload_id());
  e.ntt_pow_phi();
  CHECK(e.payload_id() != c.payload_id() && e.resident());
  CHECK(same(const_cast<const poly_p &>(c).poly_obj(), *R));
  R->ntt_pow_phi();
  CHECK(same(const_cast<const poly_p &>(e).poly_obj(), *R));
  e.invntt_pow_invphi();
  CHECK(bool(e == c) && !bool(e != c));                // compared in HBM (the reference's "any lane" semantics)
  // a whole-value overwrite of a shared handle re-seats it without copying; the other owner keeps the old value
  poly_p f = c, g = c;
  f = a - b;
  *R = *A - *B;
  CHECK(same(const_cast<const poly_p &>(f).poly_obj(), *R) && g.payload_id() == c.payload_id());
  // ... also when the tree reads the handle being overwritten
  poly_p h = c, keep = h;
  h = h + h * a;
  *R = *A * *B + *D;
  Heap<poly_t> R2(*R + *R * *A);
  CHECK(same(const_cast<const poly_p &>(h).poly_obj(), *R2) && same(const_cast<const poly_p &>(keep).poly_obj(), *R));
  // aliasing on a unique handle: in place
  poly_p acc{nfl::uniform(11)};
  const void *id = acc.payload_id();
  acc = acc + b;
  *R = *A + *B;
  CHECK(acc.payload_id() == id && same(const_cast<const poly_p &>(acc).poly_obj(), *R));
  // host write access retires the device image; the next device op uploads the new words
  poly_p w{nfl::uniform(11)};
  w(0, 0) = 0;
  CHECK(!w.resident());
  (*A)(0, 0) = 0;
  poly_p w2 = w * b;
  *R = *A * *B;
  CHECK(w.resident() && same(const_cast<const poly_p &>(w2).poly_obj(), *R));
  (*A)(0, 0) = const_cast<const poly_p &>(a)(0, 0);
  // mixed trees: inline polys are staged, handles are read where they are
  poly_p m = a * *B + d;
  *R = *A * *B + *D;
  CHECK(m.resident() && same(const_cast<const poly_p &>(m).poly_obj(), *R));
  *R = a * b + *D;                                     // host target, handle leaves
  Heap<poly_t> R3(*A * *B + *D);
  CHECK(same(*R, *R3));
  // a tree with more than three distinct leaves still is ONE device pass for handles (8 operands)
  const poly_p x{nfl::uniform(14)}, y{nfl::uniform(15)};
  Heap<poly_t> X(nfl::uniform(14)), Y(nfl::uniform(15));
  poly_p big = (a + b) * (d - x) + y * a;
  *R = *A + *B;
  Heap<poly_t> T1(*D - *X), T2(*R * *T1), T3(*Y * *A);
  *R = *T2 + *T3;
  CHECK(big.resident() && same(const_cast<const poly_p &>(big).poly_obj(), *R));
  // shoup forms on handles
  poly_p bs = nfl::compute_shoup(b), prod = nfl::shoup(a * b, bs);
  *R = *A * *B;
  CHECK(prod.resident() && same(const_cast<const poly_p &>(prod).poly_obj(), *R));
  // the zero handle costs nothing until used, and is the zero polynomial on either side
  poly_p z;
  CHECK(!z.resident());
  poly_p zz = z + a;
  CHECK(same(const_cast<const poly_p &>(zz).poly_obj(), *A));
  CHECK(!bool(const_cast<const poly_p &>(z).poly_obj()));
  // random tags: fresh values per call, in HBM
  poly_p r1{nfl::uniform()}, r2{nfl::uniform()};
  CHECK(r1.resident() && bool(r1 != r2));
  r1 = nfl::non_uniform(5);
  for (auto v : const_cast<const poly_p &>(r1).poly_obj()) CHECK(v < 5 || v > poly_t::get_modulus(0) - 5 || NbModuli > 1);
  return true;
}

// deferred execution is an execution strategy, not a semantics: with the sampler state pinned, the same program gives the
// same polynomials whether operations are queued and coalesced or launched one by one; errors surface where they are
// written; resident batches and handles share one stream and stay ordered.
template <class T, size_t Degree, size_t NbModuli> static bool run_deferred_semantics() {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  nfl::FastGaussianNoise<uint8_t, T, 2> fg(3.19, 128, 1 << 10);
  unsigned char key[32];
  for (int i = 0; i < 32; i++) key[i] = (unsigned char)(41 * i + 7);
  const size_t K = 24;
  auto program = [&](std::vector<poly_t> &out) {
    nfl::set_sampler_key(key, 500);
    std::vector<poly_p> r(K);
    poly_p s{G(&fg)}, a{nfl::uniform()};
    s.ntt_pow_phi();
    a.ntt_pow_phi();
    for (size_t i = 0; i < K; i++) {
      poly_p e{G(&fg, 2)}, z{nfl::ZO_dist()}, h{nfl::hwt_dist(Degree / 8)}, n{nfl::non_uniform(17, 3)};
      e.ntt_pow_phi();
      z.ntt_pow_phi();
      r[i] = a * s + e + z * h - n;        // (h, n stay in coefficient form: any words do for the comparison)
      if (i % 5 == 4) r[i].invntt_pow_invphi();
      if (i % 7 == 6) r[i] = r[i] + r[i - 1];   // a dependency across iterations
    }
    out.resize(K);
    for (size_t i = 0; i < K; i++) std::memcpy(out[i].data(), const_cast<const poly_p &>(r[i]).poly_obj().cdata(), sizeof(poly_t));
  };
  std::vector<poly_t> lazy_out, eager_out;
  nfl::set_deferred(true);
  program(lazy_out);
  poly_p::synchronize();
  nfl::set_deferred(false);
  program(eager_out);
  nfl::set_deferred(true);
  for (size_t i = 0; i < K; i++) CHECK(same(lazy_out[i], eager_out[i]));
  // a constructor that must throw throws where it is written, deferred or not (core.hpp:205-210)
  for (int mode = 0; mode < 2; mode++) {
    nfl::set_deferred(mode == 0);
    bool threw = false;
    try {
      poly_p bad{nfl::non_uniform(uint64_t(poly_t::get_modulus(0)) + 1)};
    } catch (const std::runtime_error &) {
      threw = true;
    }
    CHECK(threw);
  }
  nfl::set_deferred(true);
  // handles and resident batches share the ring's stream: a key built by deferred operations, replicated into a batch
  poly_p k1{nfl::uniform(5)}, k2{nfl::uniform(6)};
  poly_p ks = k1 * k2 + k1;
  nfl::device_batch<poly_t> B(3);
  B.fill(ks);                                   // reads the handle: runs the queue first
  Heap<poly_t> K1(nfl::uniform(5)), K2(nfl::uniform(6)), KS(*K1 * *K2 + *K1);
  void *mem = nullptr;
  if (posix_memalign(&mem, 32, 3 * sizeof(poly_t)) != 0) throw std::bad_alloc();
  poly_t *host = new (mem) poly_t[3];
  B.download(host);
  bool ok = same(host[0], *KS) && same(host[1], *KS) && same(host[2], *KS);
  for (int i = 0; i < 3; i++) host[i].~poly_t();
  free(mem);
  CHECK(ok);
  return true;
}

// the reference's LWE demo with plain poly_p operators (tests/nfllib_demo_main_op.cpp:26-58, 260-332)
template <class T, size_t Degree, size_t NbModuli> static bool run_lwe(double *enc_per_s, double *dec_per_s, double *batch_enc_per_s,
                                                                       double *batch_dec_per_s, size_t *launches, size_t *operations,
                                                                       double *fused_enc_per_s, double *fused_dec_per_s) {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  nfl::FastGaussianNoise<uint8_t, T, 2> g_prng(4, 128, 1 << 10);
  const size_t REPS = getenv("NFL_LWE_REPS") ? size_t(atol(getenv("NFL_LWE_REPS"))) : 2048;
  poly_p s{G(&g_prng)};
  s.ntt_pow_phi();
  poly_p sprime = nfl::compute_shoup(s);
  poly_p pka{nfl::uniform()}, pkb{G(&g_prng, 2)};
  pkb.ntt_pow_phi();
  pkb = pkb + nfl::shoup(pka * s, sprime);
  std::vector<poly_p> resa(REPS), resb(REPS);
  auto encrypt = [&](poly_p &ra, poly_p &rb) {
    poly_p u{G(&g_prng)}, e1{G(&g_prng, 2)}, e2{G(&g_prng, 2)};
    u.ntt_pow_phi();
    e1.ntt_pow_phi();
    e2.ntt_pow_phi();
    ra = u * pka + e1;
    rb = u * pkb + e2;
  };
  auto decrypt = [&](poly_p &out, poly_p const &ra, poly_p const &rb) {
    out = rb - ra * s;
    out.invntt_pow_invphi();
  };
  // warm-up: tables, and one full round so that the buffer pool has its slabs -- the timed round is the steady state
  // (results overwrite the previous round's, temporaries recycle the pool)
  for (size_t i = 0; i < REPS; i++) encrypt(resa[i], resb[i]);
  poly_p::synchronize();
  const size_t l0 = poly_p::deferred_launches(), o0 = poly_p::deferred_operations();
  auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < REPS; i++) encrypt(resa[i], resb[i]);
  auto t_rec = std::chrono::steady_clock::now();
  poly_p::synchronize();
  auto t1 = std::chrono::steady_clock::now();
  const double enc_s = std::chrono::duration<double>(t1 - t0).count();
  if (getenv("NFL_LWE_VERBOSE"))
    std::fprintf(stderr, "lwe: %zu encryptions: recorded in %.3f ms (incl. queue runs), finished after %.3f ms\n", REPS,
                 std::chrono::duration<double>(t_rec - t0).count() * 1e3, enc_s * 1e3);
  *launches = poly_p::deferred_launches() - l0;
  *operations = poly_p::deferred_operations() - o0;
  std::vector<poly_p> dec(REPS);
  for (size_t i = 0; i < REPS; i++) decrypt(dec[i], resa[i], resb[i]);
  poly_p::synchronize();
  t1 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < REPS; i++) decrypt(dec[i], resa[i], resb[i]);
  poly_p::synchronize();
  auto t2 = std::chrono::steady_clock::now();
  *enc_per_s = REPS / enc_s;
  *dec_per_s = REPS / std::chrono::duration<double>(t2 - t1).count();
  // the demo's own check: ciphertexts of 0 decrypt to even noise, so the parities sum to 0
  const T modulus = poly_t::get_modulus(0);
  for (size_t i = 0; i < REPS; i += 97) {
    const poly_t &t = const_cast<const poly_p &>(dec[i]).poly_obj();
    for (size_t j = 0; j < Degree; j++) {
      const T v = t(0, j);
      CHECK(((v < modulus / 2) ? v % 2 : 1 - v % 2) == 0);
    }
  }
  // the same work on a resident batch (what tools/lwe_demo.py times)
  using batch_t = nfl::device_batch<poly_t>;
  const size_t B = 4096;
  batch_t S(B), PKA(B), PKB(B), U(B), E1(B), E2(B), RA(B), RB(B), DEC(B);
  S.fill(s);
  PKA.fill(pka);
  PKB.fill(pkb);
  const unsigned char fma[] = {0, 1, NFLHIP_EXPR_MUL, 2, NFLHIP_EXPR_ADD}, dcd[] = {0, 1, 2, NFLHIP_EXPR_MUL, NFLHIP_EXPR_SUB};
  auto batch_enc = [&]() {
    U.set(G(&g_prng)); E1.set(G(&g_prng, 2)); E2.set(G(&g_prng, 2));
    U.ntt_pow_phi(); E1.ntt_pow_phi(); E2.ntt_pow_phi();
    const batch_t *o1[] = {&U, &PKA, &E1}, *o2[] = {&U, &PKB, &E2};
    RA.assign_program(fma, sizeof(fma), o1, 3);
    RB.assign_program(fma, sizeof(fma), o2, 3);
  };
  auto batch_dec = [&]() {
    const batch_t *o[] = {&RB, &RA, &S};
    DEC.assign_program(dcd, sizeof(dcd), o, 3);
    DEC.invntt_pow_invphi();
  };
  batch_enc(); batch_dec(); S.sync();
  {
    // the same through the batch's fused pipelines (device_batch::assign_gaussian_fma2 / assign_fma_inv): identical words when
    // the samplers start from the same stream id; keys as ONE polynomial each (stride 0) and as one per element
    unsigned char key[32];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(7 * i + 3);
    batch_t S1(1), PKA1(1), PKB1(1), RA2(B), RB2(B), DEC2(B);
    S1.fill(s);
    PKA1.fill(pka);
    PKB1.fill(pkb);
    nfl::set_sampler_key(key, 1000);
    batch_enc();
    batch_dec();
    nfl::set_sampler_key(key, 1000);
    RA2.assign_gaussian_fma2(RB2, G(&g_prng), PKA1, G(&g_prng, 2), PKB1, G(&g_prng, 2));
    DEC2.assign_fma_inv(RA2, S1, RB2, true);
    CHECK(!RA2.any_differs(RA) && !RB2.any_differs(RB) && !DEC2.any_differs(DEC));
    nfl::set_sampler_key(key, 1000);
    RA2.assign_gaussian_fma(G(&g_prng), PKA, G(&g_prng, 2));       // one result, keys per element; e2's stream id is skipped ...
    CHECK(!RA2.any_differs(RA));
    DEC2.assign_fma_inv(RA, S, RB, true);
    CHECK(!DEC2.any_differs(DEC));
    DEC2.assign_fma_inv(RA, S1, RB, false);                        // ... and the sum: INTT(rb + ra * s)
    const batch_t *o[] = {&RB, &RA, &S};
    const unsigned char sum[] = {0, 1, 2, NFLHIP_EXPR_MUL, NFLHIP_EXPR_ADD};
    RB2.assign_program(sum, sizeof(sum), o, 3);
    RB2.invntt_pow_invphi();
    CHECK(!DEC2.any_differs(RB2));
    auto fused_enc = [&]() { RA2.assign_gaussian_fma2(RB2, G(&g_prng), PKA1, G(&g_prng, 2), PKB1, G(&g_prng, 2)); };
    auto fused_dec = [&]() { DEC2.assign_fma_inv(RA2, S1, RB2, true); };
    fused_enc(); fused_dec(); S.sync();
    auto f0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 4; k++) fused_enc();
    S.sync();
    auto f1 = std::chrono::steady_clock::now();
    for (int k = 0; k < 4; k++) fused_dec();
    S.sync();
    auto f2 = std::chrono::steady_clock::now();
    *fused_enc_per_s = 4 * B / std::chrono::duration<double>(f1 - f0).count();
    *fused_dec_per_s = 4 * B / std::chrono::duration<double>(f2 - f1).count();
  }
  t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 4; k++) batch_enc();
  S.sync();
  t1 = std::chrono::steady_clock::now();
  for (int k = 0; k < 4; k++) batch_dec();
  S.sync();
  t2 = std::chrono::steady_clock::now();
  *batch_enc_per_s = 4 * B / std::chrono::duration<double>(t1 - t0).count();
  *batch_dec_per_s = 4 * B / std::chrono::duration<double>(t2 - t1).count();
  return true;
}

int main() {
  try {
    if (!run_residency<uint64_t, 4096, 4>()) return 1;
    if (!run_residency<uint32_t, 1024, 2>()) return 1;
    if (!run_residency<uint16_t, 128, 1>()) return 1;
    if (!run_residency<uint32_t, 8, 2>()) return 1;      // rows shorter than a 16-byte vector: the host route
    if (!run_residency<uint64_t, 32768, 2>()) return 1;
    if (!run_deferred_semantics<uint64_t, 4096, 4>()) return 1;
    if (!run_deferred_semantics<uint32_t, 1024, 2>()) return 1;
    if (!run_deferred_semantics<uint64_t, 1024, 1>()) return 1;
    double e = 0, d = 0, be = 0, bd = 0, fe = 0, fd = 0, fx = 0, fy = 0;
    size_t nl = 0, no = 0;
    if (!run_lwe<uint64_t, 4096, 4>(&e, &d, &be, &bd, &nl, &no, &fe, &fd)) return 1;
    // the same loops with every operation launched when it is called (no deferral): what a per-polynomial API costs
    nfl::poly_p<uint64_t, 4096, 4>::synchronize();
    nfl::set_deferred(false);
    double ee = 0, ed = 0, x0 = 0, x1 = 0;
    size_t y0 = 0, y1 = 0;
    if (!run_lwe<uint64_t, 4096, 4>(&ee, &ed, &x0, &x1, &y0, &y1, &fx, &fy)) return 1;
    nfl::set_deferred(true);
    std::printf("{\"lwe_u64_4096_4\": {\"poly_p_encryptions_per_s\": %.1f, \"poly_p_decryptions_per_s\": %.1f, "
                "\"device_batch_encryptions_per_s\": %.1f, \"device_batch_decryptions_per_s\": %.1f, "
                "\"device_batch_fused_encryptions_per_s\": %.1f, \"device_batch_fused_decryptions_per_s\": %.1f, "
                "\"poly_p_eager_encryptions_per_s\": %.1f, \"poly_p_eager_decryptions_per_s\": %.1f, "
                "\"deferred_operations\": %zu, \"launches_they_became\": %zu}}\n", e, d, be, bd, fe, fd, ee, ed, no, nl);
    std::printf("all checks passed\n");
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
