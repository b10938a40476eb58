// tests/cpp/strictmod_main.cpp -- built with -DCHECK_STRICTMOD (and without NDEBUG), as the reference's tests are
// (tests/CMakeLists.txt:10): the header then asserts x < p on the operands of the transforms and operators, where the
// reference's ASSERT_STRICTMOD does (debug.hpp:33-37; core.hpp:457-462; ops.hpp:131,148,190,211,235) -- host words on the
// host, resident values through nflhip_check_range_dev -- and throws std::runtime_error instead of aborting.
//   * a word >= p in an operand of ntt_pow_phi / invntt_pow_invphi / operator+,-,* / shoup(a*b, b') is caught: inline polys,
//     resident poly_p handles and device batches;
//   * canonical operands pass, and the Shoup companion b' (a quotient, up to 2^64 - 1) is exempt.
// Exit code 0 = all checks passed.  Runs on the GPU (tests/test_cpp_surface.py).
#include <nfl.hpp>

#include <cstdio>
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>

#ifndef CHECK_STRICTMOD
#error "build this program with -DCHECK_STRICTMOD"
#endif

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return false; } } while (0)

template <class F> static bool trips(F f) {
  try {
    f();
  } catch (const std::runtime_error &e) {
    return std::string(e.what()).find("CHECK_STRICTMOD") != std::string::npos;
  }
  return false;
}
template <class F> static bool passes(F f) {
  try {
    f();
  } catch (const std::exception &e) {
    std::printf("unexpected exception: %s\n", e.what());
    return false;
  }
  return true;
}

template <class P> struct Heap {  // polys are large: keep them off the stack, 32-byte aligned like the reference's alloc_aligned
  P *p;
  template <class... A> explicit Heap(A &&...a) : p(nullptr) {
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
    p = new (mem) P(std::forward<A>(a)...);
  }
  ~Heap() { p->~P(); free(p); }
  P &operator*() { return *p; }
};

template <class T, size_t Degree, size_t NbModuli> static bool run() {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  const size_t last = NbModuli - 1;
  const T p_last = poly_t::get_modulus(last);
  Heap<poly_t> a(nfl::uniform(1)), b(nfl::uniform(2)), c, bad(nfl::uniform(3));
  (*bad)(last, Degree - 1) = p_last;                       // the smallest illegal word, in the last place looked at
  // inline polynomials
  CHECK(passes([&] { *c = *a + *b; *c = *a * *b - *a; (*c).ntt_pow_phi(); (*c).invntt_pow_invphi(); }));
  CHECK(trips([&] { Heap<poly_t> t(*bad); (*t).ntt_pow_phi(); }));
  CHECK(trips([&] { Heap<poly_t> t(*bad); (*t).invntt_pow_invphi(); }));
  CHECK(trips([&] { *c = *a + *bad; }));
  CHECK(trips([&] { *c = *bad * *b; }));
  CHECK(trips([&] { *c = *a - (*b * *bad); }));
  {  // mulmod_shoup: b' is a quotient -- any word -- and must not trip; a poisoned a or b must
    Heap<poly_t> bp(nfl::compute_shoup(*b));
    CHECK(passes([&] { *c = nfl::shoup(*a * *b, *bp); }));
    CHECK(trips([&] { *c = nfl::shoup(*bad * *b, *bp); }));
  }
  // resident handles (the check runs on the device)
  poly_p ha{nfl::uniform(4)}, hb{nfl::uniform(5)}, hc, hbad{nfl::uniform(6)};
  hbad(last, 0) = T(p_last + 1);                           // (a host write through the handle; uploaded when next used)
  CHECK(passes([&] { hc = ha * hb + ha; hc.ntt_pow_phi(); hc.invntt_pow_invphi(); (void)const_cast<const poly_p &>(hc)(0, 0); }));
  CHECK(trips([&] { poly_p t = hbad; t.ntt_pow_phi(); }));
  CHECK(trips([&] { hc = ha + hbad; }));
  CHECK(trips([&] { hc = hbad * hb - ha; }));
  // resident batches
  nfl::device_batch<poly_t> A(3), B(3), C(3), BAD(3);
  A.set(nfl::uniform(7));
  B.set(nfl::uniform(8));
  BAD.fill(hbad);
  CHECK(passes([&] { C.assign(NFLHIP_OP_MUL, A, B); C.ntt_pow_phi(); C.invntt_pow_invphi(); C.sync(); }));
  CHECK(trips([&] { BAD.ntt_pow_phi(); }));
  CHECK(trips([&] { C.assign(NFLHIP_OP_ADD, A, BAD); }));
  return true;
}

int main() {
  try {
    if (!run<uint64_t, 4096, 4>()) return 1;
    if (!run<uint32_t, 1024, 2>()) return 1;
    if (!run<uint16_t, 128, 1>()) return 1;
    std::printf("CHECK_STRICTMOD: poisoned operands are caught on the host and on the device, canonical ones pass\nall checks passed\n");
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
