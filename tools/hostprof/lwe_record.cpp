// HOST-SIDE PROFILING AID: the per-polynomial LWE loop of tests/cpp/resident_main.cpp (the reference's
// tests/nfllib_demo_main_op.cpp:26-58 with poly_p operators), timed against the null backend of make_null_backend.py --
// what is measured is the header's own cost per deferred operation (recording + the queue runs), nothing is computed.
//   make -C tools/hostprof && tools/hostprof/_build/lwe_record [reps]
#include <nfl.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef HOSTPROF_PCSAMPLE
#include "pcsample.h"
#endif

int main(int argc, char **argv) {
  using T = uint64_t;
  using poly_p = nfl::poly_p<T, 4096, 4>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  const size_t REPS = argc > 1 ? size_t(atol(argv[1])) : 16384;
  nfl::FastGaussianNoise<uint8_t, T, 2> g_prng(4, 128, 1 << 10);
  poly_p s{G(&g_prng)};
  s.ntt_pow_phi();
  poly_p pka{nfl::uniform()}, pkb{G(&g_prng, 2)};
  pkb.ntt_pow_phi();
  std::vector<poly_p> resa(REPS), resb(REPS), dec(REPS);
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
  double best_e = 1e9, best_d = 1e9;
#ifdef HOSTPROF_PCSAMPLE
  pcsample_start();
#endif
  const int ROUNDS = getenv("HOSTPROF_ROUNDS") ? atoi(getenv("HOSTPROF_ROUNDS")) : 6;
  for (int round = 0; round < ROUNDS; ++round) {
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < REPS; i++) encrypt(resa[i], resb[i]);
    auto tr = std::chrono::steady_clock::now();
    poly_p::synchronize();
    auto t1 = std::chrono::steady_clock::now();
    if (round && getenv("HOSTPROF_SPLIT"))
      std::printf("  round %d: loop %.0f ns + final queue run %.0f ns per encryption\n", round,
                  std::chrono::duration<double>(tr - t0).count() / REPS * 1e9, std::chrono::duration<double>(t1 - tr).count() / REPS * 1e9);
    for (size_t i = 0; i < REPS; i++) decrypt(dec[i], resa[i], resb[i]);
    poly_p::synchronize();
    auto t2 = std::chrono::steady_clock::now();
    const double e = std::chrono::duration<double>(t1 - t0).count(), d = std::chrono::duration<double>(t2 - t1).count();
    if (round) best_e = e < best_e ? e : best_e, best_d = d < best_d ? d : best_d;
  }
#ifdef HOSTPROF_PCSAMPLE
  pcsample_dump("/tmp/hostprof_pcs.txt");
#endif
  std::printf("host cost per encryption %.0f ns (8 deferred operations), per decryption %.0f ns (2); %zu reps, launches %zu for %zu operations\n",
              best_e / REPS * 1e9, best_d / REPS * 1e9, REPS, poly_p::deferred_launches(), poly_p::deferred_operations());
  return 0;
}
