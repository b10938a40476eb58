// tests/cpp/deferred_product.cpp -- products computed the way a caller of the reference writes them, on resident poly_p
// handles with deferred execution (include/nfl_hip/nfl.hpp, detail::lazy):
//     a.ntt_pow_phi(); b.ntt_pow_phi(); c = a * b; c.invntt_pow_invphi();          (poly.hpp:167-168, 350)
// in a loop over K seeded operand pairs, so that the queue coalesces the loop into batched launches.  The words of every
// c are written to a file; tests/test_zz_gpu_deferred_loops.py recomputes them with the CPU checker from the same seeds
// (operand i: nfl::uniform(seed + i), the shared counter stream of nflhip_fill_uniform_dev).
// Usage: deferred_product <out file> <K> <seed>
#include <nfl.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <class T, size_t Degree, size_t NbModuli> static void run(std::FILE *f, size_t K, uint64_t seed) {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  nfl::set_deferred(true);
  std::vector<poly_p> c(K);
  for (size_t i = 0; i < K; ++i) {
    poly_p a{nfl::uniform(seed + 2 * i)}, b{nfl::uniform(seed + 2 * i + 1)};
    a.ntt_pow_phi();
    b.ntt_pow_phi();
    c[i] = a * b;
    c[i].invntt_pow_invphi();
  }
  for (size_t i = 0; i < K; ++i) {
    const poly_t &v = const_cast<const poly_p &>(c[i]).poly_obj();
    std::fwrite(v.cdata(), sizeof(T), Degree * NbModuli, f);
  }
}

int main(int argc, char **argv) {
  if (argc < 4) return 3;
  try {
    std::FILE *f = std::fopen(argv[1], "wb");
    if (!f) return 3;
    const size_t K = size_t(std::atol(argv[2]));
    const uint64_t seed = std::strtoull(argv[3], nullptr, 0);
    run<uint64_t, 4096, 4>(f, K, seed);
    run<uint32_t, 1024, 2>(f, K, seed);
    std::fclose(f);
    std::printf("wrote %zu + %zu products\n", K, K);
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
