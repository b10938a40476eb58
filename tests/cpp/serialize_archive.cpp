// tests/cpp/serialize_archive.cpp -- the archive hook of the drop-in header: poly::serialize(Archive &) /
// poly_p::serialize(Archive &) (reference include/nfl/poly.hpp:189-191, poly_p.hpp; exercised by the reference's
// tests/poly_serialize_cereal.cpp with cereal::BinaryOutputArchive / BinaryInputArchive, which this image does not have).
// The two archives below follow cereal's calling convention for exactly what the hook uses -- `archive(x)` calls
// x.serialize(archive) for class types and moves the raw bytes of arrays of arithmetic types -- so the hook is
// instantiated and run: a polynomial written through the archive must be byte-identical to serialize_manually's image
// (poly.hpp:180-185) and must read back equal, for inline polys and for resident poly_p handles.
// Host-only program (no device work besides what poly_p's constructor does): exit code 0 = all checks passed.
#include <nfl.hpp>

#include <cstdio>
#include <cstring>
#include <sstream>
#include <type_traits>

namespace test_archive {
class BinaryOut {
 public:
  explicit BinaryOut(std::ostream &os) : os_(os) {}
  template <class T, size_t N> typename std::enable_if<std::is_arithmetic<T>::value>::type operator()(T (&a)[N]) {
    os_.write(reinterpret_cast<const char *>(a), std::streamsize(N * sizeof(T)));
  }
  template <class C> typename std::enable_if<std::is_class<C>::value>::type operator()(C &c) { c.serialize(*this); }

 private:
  std::ostream &os_;
};
class BinaryIn {
 public:
  explicit BinaryIn(std::istream &is) : is_(is) {}
  template <class T, size_t N> typename std::enable_if<std::is_arithmetic<T>::value>::type operator()(T (&a)[N]) {
    is_.read(reinterpret_cast<char *>(a), std::streamsize(N * sizeof(T)));
  }
  template <class C> typename std::enable_if<std::is_class<C>::value>::type operator()(C &c) { c.serialize(*this); }

 private:
  std::istream &is_;
};
}  // namespace test_archive

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return false; } } while (0)

template <class T, size_t Degree, size_t NbModuli> static bool run(bool real_arithmetic) {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  const size_t bytes = sizeof(T) * Degree * NbModuli;
  void *mem = nullptr;
  if (posix_memalign(&mem, 32, 3 * sizeof(poly_t)) != 0) return false;
  poly_t *p = new (mem) poly_t[3];
  p[0] = nfl::uniform(7);
  std::stringstream manual(std::ios::in | std::ios::out | std::ios::binary), arch(std::ios::in | std::ios::out | std::ios::binary);
  p[0].serialize_manually(manual);
  test_archive::BinaryOut out(arch);
  out(p[0]);
  CHECK(manual.str().size() == bytes && arch.str() == manual.str());   // the archive image IS the manual image
  test_archive::BinaryIn in(arch);
  in(p[1]);
  CHECK(std::memcmp(p[0].data(), p[1].data(), bytes) == 0);
  p[2].deserialize_manually(manual);
  CHECK(std::memcmp(p[0].data(), p[2].data(), bytes) == 0);
  // resident handles: the hook forwards to the polynomial the handle stands for (poly_p.hpp)
  poly_p h0{nfl::uniform(9)}, h1;
  std::stringstream arch_p(std::ios::in | std::ios::out | std::ios::binary), manual_p(std::ios::in | std::ios::out | std::ios::binary);
  test_archive::BinaryOut out_p(arch_p);
  out_p(h0);
  h0.serialize_manually(manual_p);
  CHECK(arch_p.str().size() == bytes && arch_p.str() == manual_p.str());
  test_archive::BinaryIn in_p(arch_p);
  in_p(h1);
  CHECK(std::memcmp(const_cast<const poly_p &>(h0).poly_obj().cdata(), const_cast<const poly_p &>(h1).poly_obj().cdata(), bytes) == 0);
  h1.ntt_pow_phi();            // a handle that was read from an archive is a normal handle: transform on the device and back
  h1.invntt_pow_invphi();
  CHECK(!real_arithmetic || std::memcmp(const_cast<const poly_p &>(h0).poly_obj().cdata(), const_cast<const poly_p &>(h1).poly_obj().cdata(), bytes) == 0);
  for (int i = 0; i < 3; ++i) p[i].~poly_t();
  free(mem);
  return true;
}

int main(int argc, char **argv) {
  const bool real = !(argc > 1 && std::strcmp(argv[1], "toy") == 0);   // (tests/cpp/mock: the toy transforms are no inverses of each other)
  try {
    if (!run<uint64_t, 4096, 4>(real)) return 1;
    if (!run<uint32_t, 1024, 2>(real)) return 1;
    if (!run<uint16_t, 128, 1>(real)) return 1;
    std::printf("serialize(Archive &) == serialize_manually, round trips exact\nall checks passed\n");
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
