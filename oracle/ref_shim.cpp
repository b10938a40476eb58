// ref_shim.cpp -- C entry points INTO the real reference (quarkslab/NFLlib @ v1).
//
// TEST INFRASTRUCTURE ONLY.  This translation unit contains no reference code:
// it #includes the reference's own headers where they lie under
// /root/reference/include and instantiates nfl::poly<T,Degree,NbModuli> for a
// fixed list of shapes, exposing each public operation through a flat C ABI so
// that tests/ and tools/gen_golden.py can (a) pin oracle/nfl_oracle.c against
// the real thing bit-for-bit and (b) generate the golden fixtures.  It is
// built by oracle/Makefile into oracle/_ref/libnflref.so (git-ignored) together
// with the reference's lib/params/params.cpp, lib/prng/*.cpp and the Salsa20
// assembly, against the GMP/MPFR development files that ship in this image
// under /opt/conda.  Nothing in the product path links or loads it.
#include <nfl.hpp>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <new>

namespace nfl {
namespace tests {
// The reference grants friendship to this name (poly.hpp:69-76, 85, 202) so
// that tests can reach the protected tables and core::ntt.
template <class P>
class poly_tests_proxy {
 public:
  using T = typename P::value_type;
  static const T *table(int which, size_t cm) {
    switch (which) {
      case 0: return P::base.phis[cm];
      case 1: return P::base.shoupphis[cm];
      case 2: return P::base.invpoly_times_invphis[cm];
      case 3: return P::base.shoupinvpoly_times_invphis[cm];
      case 4: return P::base.omegas[cm];
      case 5: return P::base.invomegas[cm];
      case 6: return &P::base.invpolyDegree[cm];
      default: return nullptr;
    }
  }
  static void ntt_row(T *x, size_t cm, int inv) {
    if (inv)
      P::core::ntt(x, P::base.invomegas[cm], P::base.shoupinvomegas[cm], P::get_modulus(cm));
    else
      P::core::ntt(x, P::base.omegas[cm], P::base.shoupomegas[cm], P::get_modulus(cm));
  }
  static size_t crt_bits() { return P::gmp.bits_in_moduli_product; }
  static size_t crt_shift() { return P::gmp.shift_modulus_shoup; }
  static mpz_t &crt_Q() { return P::gmp.moduli_product; }
  static mpz_t &crt_mshoup() { return P::gmp.modulus_shoup; }
  static mpz_t &crt_lifting(size_t cm) { return P::gmp.lifting_integers[cm]; }
  static void lift(std::array<mpz_t, P::degree> &rop, P const &op) { P::gmp.poly2mpz(rop, op); }
  static void project(P &rop, std::array<mpz_t, P::degree> const &v) { P::gmp.mpz2poly(rop, v); }
};
}  // namespace tests
}  // namespace nfl

namespace {

struct shape_vtbl {
  int limb_bits;
  size_t degree, nmoduli;
  void (*ntt)(void *);
  void (*intt)(void *);
  void (*pointwise)(int, void *, const void *, const void *, const void *);
  int (*any)(int, const void *, const void *);
  const void *(*table)(int, size_t);
  void (*ntt_row)(void *, size_t, int);
  size_t (*crt_info)(int);
  size_t (*crt_const)(int, size_t, uint64_t *, size_t);
  void (*lift)(const void *, uint64_t *, size_t);
  void (*project)(void *, const uint64_t *, size_t);
  long (*sample)(int, uint64_t, uint64_t, double, void *, unsigned char *, size_t);
};

template <class P>
P *make_poly(const void *src) {
  void *mem = nullptr;
  if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
  P *p = new (mem) P();
  if (src) std::memcpy(p->data(), src, sizeof(typename P::value_type) * P::degree * P::nmoduli);
  return p;
}
template <class P>
void drop_poly(P *p) {
  p->~P();
  free(p);
}
template <class P>
void store(void *dst, P *p) {
  std::memcpy(dst, p->data(), sizeof(typename P::value_type) * P::degree * P::nmoduli);
}

static size_t export_mpz(mpz_t const &z, uint64_t *out, size_t cap) {
  size_t cnt = 0;
  std::memset(out, 0, cap * sizeof(uint64_t));
  size_t need = (mpz_sizeinbase(z, 2) + 63) / 64;
  if (mpz_sgn(z) == 0) return 0;
  if (need > cap) return need;
  mpz_export(out, &cnt, -1, sizeof(uint64_t), 0, 0, z);
  return cnt;
}

template <class P>
struct ops_for {
  using T = typename P::value_type;
  using proxy = nfl::tests::poly_tests_proxy<P>;
  static void ntt(void *d) {
    P *p = make_poly<P>(d);
    p->ntt_pow_phi();
    store(d, p);
    drop_poly(p);
  }
  static void intt(void *d) {
    P *p = make_poly<P>(d);
    p->invntt_pow_invphi();
    store(d, p);
    drop_poly(p);
  }
  static void pointwise(int op, void *out, const void *a, const void *b, const void *bp) {
    P *pa = make_poly<P>(a), *pb = make_poly<P>(b), *pp = make_poly<P>(bp), *r = make_poly<P>(nullptr);
    switch (op) {
      case 0: *r = *pa + *pb; break;
      case 1: *r = *pa - *pb; break;
      case 2: *r = *pa * *pb; break;
      case 3: *r = nfl::shoup(*pa * *pb, *pp); break;
      case 4: *r = nfl::compute_shoup(*pa); break;
      default: break;
    }
    store(out, r);
    drop_poly(pa); drop_poly(pb); drop_poly(pp); drop_poly(r);
  }
  static int any(int want_eq, const void *a, const void *b) {
    P *pa = make_poly<P>(a), *pb = make_poly<P>(b);
    bool r = want_eq ? bool(*pa == *pb) : bool(*pa != *pb);
    drop_poly(pa); drop_poly(pb);
    return r ? 1 : 0;
  }
  static const void *table(int which, size_t cm) { return proxy::table(which, cm); }
  static void ntt_row(void *x, size_t cm, int inv) {
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(T) * P::degree) != 0) throw std::bad_alloc();
    std::memcpy(mem, x, sizeof(T) * P::degree);
    proxy::ntt_row(static_cast<T *>(mem), cm, inv);
    std::memcpy(x, mem, sizeof(T) * P::degree);
    free(mem);
  }
  static size_t crt_info(int what) { return what == 0 ? proxy::crt_bits() : proxy::crt_shift(); }
  static size_t crt_const(int what, size_t cm, uint64_t *out, size_t cap) {
    switch (what) {
      case 0: return export_mpz(proxy::crt_Q(), out, cap);
      case 1: return export_mpz(proxy::crt_mshoup(), out, cap);
      default: return export_mpz(proxy::crt_lifting(cm), out, cap);
    }
  }
  static void lift(const void *d, uint64_t *out, size_t L) {
    P *p = make_poly<P>(d);
    auto *arr = new std::array<mpz_t, P::degree>();
    for (size_t i = 0; i < P::degree; i++) mpz_init((*arr)[i]);
    proxy::lift(*arr, *p);
    for (size_t i = 0; i < P::degree; i++) {
      export_mpz((*arr)[i], out + i * L, L);
      mpz_clear((*arr)[i]);
    }
    delete arr;
    drop_poly(p);
  }
  static void project(void *d, const uint64_t *limbs, size_t L) {
    P *p = make_poly<P>(nullptr);
    auto *arr = new std::array<mpz_t, P::degree>();
    for (size_t i = 0; i < P::degree; i++) {
      mpz_init((*arr)[i]);
      mpz_import((*arr)[i], L, -1, sizeof(uint64_t), 0, 0, limbs + i * L);
    }
    proxy::project(*p, *arr);
    for (size_t i = 0; i < P::degree; i++) mpz_clear((*arr)[i]);
    delete arr;
    store(d, p);
    drop_poly(p);
  }
  // The random constructors (core.hpp:146-391).  For the kinds that make ONE fastrandombytes call
  // (uniform / non_uniform / ZO_dist) the bytes that call will return are captured first: the Salsa20 key and nonce
  // are process state (lib/prng/fastrandombytes.cpp:17-37), so a forked child that asks for the same number of bytes
  // sees exactly what the parent's constructor is about to consume.  Returns the number of raw bytes captured
  // (0 for hwt_dist / gaussian, whose procedures are not one replayable call), or -1 on error.
  static long sample(int kind, uint64_t p0, uint64_t p1, double sigma, void *out, unsigned char *raw, size_t raw_cap) {
    unsigned char dummy;
    nfl::fastrandombytes(&dummy, 1);  // make sure the key exists before the state is duplicated
    size_t want = 0;
    if (kind == 0) want = sizeof(T) * P::degree * P::nmoduli;
    else if (kind == 1) want = sizeof(T) * P::degree;
    else if (kind == 2) want = P::degree;
    if (want) {
      if (!raw || raw_cap < want) return -1;
      int fd[2];
      if (pipe(fd) != 0) return -1;
      const pid_t pid = fork();
      if (pid < 0) return -1;
      if (pid == 0) {
        close(fd[0]);
        unsigned char *buf = static_cast<unsigned char *>(malloc(want));
        nfl::fastrandombytes(buf, want);
        size_t off = 0;
        while (off < want) {
          const ssize_t w = write(fd[1], buf + off, want - off);
          if (w <= 0) _exit(1);
          off += size_t(w);
        }
        _exit(0);
      }
      close(fd[1]);
      size_t off = 0;
      while (off < want) {
        const ssize_t r = read(fd[0], raw + off, want - off);
        if (r <= 0) break;
        off += size_t(r);
      }
      close(fd[0]);
      int status = 0;
      waitpid(pid, &status, 0);
      if (off != want) return -1;
    }
    P *p = nullptr;
    void *mem = nullptr;
    if (posix_memalign(&mem, 32, sizeof(P)) != 0) throw std::bad_alloc();
    switch (kind) {
      case 0: p = new (mem) P(nfl::uniform()); break;
      case 1: p = new (mem) P(nfl::non_uniform(p0, p1)); break;
      case 2: p = new (mem) P(nfl::ZO_dist(uint8_t(p0))); break;
      case 3: p = new (mem) P(nfl::hwt_dist(uint32_t(p0))); break;
      case 4: {
        nfl::FastGaussianNoise<uint8_t, T, 2> fg(sigma, unsigned(p0), P::degree);
        p = new (mem) P(nfl::gaussian<uint8_t, T, 2>(&fg, p1));
        break;
      }
      default: free(mem); return -1;
    }
    store(out, p);
    drop_poly(p);
    return long(want);
  }
  static shape_vtbl vt() {
    return shape_vtbl{int(sizeof(T) * 8), P::degree, P::nmoduli, &ntt, &intt, &pointwise, &any, &table,
                      &ntt_row, &crt_info, &crt_const, &lift, &project, &sample};
  }
};

#ifndef NFLREF_SHAPES
// reference test configs (tests/CMakeLists.txt:19-48) + BASELINE.json configs
#define NFLREF_SHAPES(X)   \
  X(uint16_t, 128, 1)      \
  X(uint32_t, 8, 2)        \
  X(uint32_t, 1024, 1)     \
  X(uint32_t, 1024, 2)     \
  X(uint64_t, 8, 2)        \
  X(uint64_t, 64, 3)       \
  X(uint64_t, 1024, 2)     \
  X(uint64_t, 4096, 4)     \
  X(uint64_t, 8192, 2)     \
  X(uint64_t, 16384, 8)    \
  X(uint64_t, 32768, 2)    \
  X(uint64_t, 16, 40)      \
  X(uint32_t, 32, 64)      \
  X(uint64_t, 64, 96)      \
  X(uint64_t, 1024, 94)    \
  X(uint64_t, 65536, 30)
#endif

#define MAKE_VT(T, D, M) ops_for<nfl::poly<T, D, M>>::vt(),
const shape_vtbl g_shapes[] = {NFLREF_SHAPES(MAKE_VT)};
const int g_nshapes = int(sizeof(g_shapes) / sizeof(g_shapes[0]));

}  // namespace

extern "C" {
int nflref_shape_count() { return g_nshapes; }
int nflref_shape_info(int id, int *limb_bits, size_t *degree, size_t *nmoduli) {
  if (id < 0 || id >= g_nshapes) return -1;
  *limb_bits = g_shapes[id].limb_bits;
  *degree = g_shapes[id].degree;
  *nmoduli = g_shapes[id].nmoduli;
  return 0;
}
int nflref_find(int limb_bits, size_t degree, size_t nmoduli) {
  for (int i = 0; i < g_nshapes; i++)
    if (g_shapes[i].limb_bits == limb_bits && g_shapes[i].degree == degree && g_shapes[i].nmoduli == nmoduli)
      return i;
  return -1;
}
void nflref_ntt_pow_phi(int id, void *poly) { g_shapes[id].ntt(poly); }
void nflref_invntt_pow_invphi(int id, void *poly) { g_shapes[id].intt(poly); }
void nflref_pointwise(int id, int op, void *out, const void *a, const void *b, const void *bp) {
  g_shapes[id].pointwise(op, out, a, b, bp);
}
int nflref_any_eq(int id, const void *a, const void *b) { return g_shapes[id].any(1, a, b); }
int nflref_any_neq(int id, const void *a, const void *b) { return g_shapes[id].any(0, a, b); }
const void *nflref_table(int id, int which, size_t cm) { return g_shapes[id].table(which, cm); }
void nflref_ntt_row(int id, void *row, size_t cm, int inverse_tables) { g_shapes[id].ntt_row(row, cm, inverse_tables); }
size_t nflref_crt_bits(int id) { return g_shapes[id].crt_info(0); }
size_t nflref_crt_shift(int id) { return g_shapes[id].crt_info(1); }
size_t nflref_crt_modulus(int id, uint64_t *out, size_t cap) { return g_shapes[id].crt_const(0, 0, out, cap); }
size_t nflref_crt_modulus_shoup(int id, uint64_t *out, size_t cap) { return g_shapes[id].crt_const(1, 0, out, cap); }
size_t nflref_crt_lifting(int id, size_t cm, uint64_t *out, size_t cap) { return g_shapes[id].crt_const(2, cm, out, cap); }
void nflref_crt_lift(int id, const void *poly, uint64_t *out, size_t L) { g_shapes[id].lift(poly, out, L); }
void nflref_crt_project(int id, void *poly, const uint64_t *limbs, size_t L) { g_shapes[id].project(poly, limbs, L); }
// kind: 0 uniform | 1 non_uniform(p0 = upper bound, p1 = amplifier) | 2 ZO_dist(p0 = rho) | 3 hwt_dist(p0 = weight) |
//       4 gaussian(sigma, p0 = security, p1 = amplifier; FastGaussianNoise<uint8_t,T,2>(sigma, security, degree))
long nflref_sample_replay(int id, int kind, uint64_t p0, uint64_t p1, double sigma, void *poly, unsigned char *raw,
                          size_t raw_cap) {
  return g_shapes[id].sample(kind, p0, p1, sigma, poly, raw, raw_cap);
}
}
