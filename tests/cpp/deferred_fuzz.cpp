// tests/cpp/deferred_fuzz.cpp -- random programs over a small pool of resident poly_p handles, executed twice with the
// same pinned sampler state: once with deferred execution (operations queued, levelled, grouped and coalesced into
// batched launches -- include/nfl_hip/nfl.hpp, detail::lazy) and once with every operation launched when it is called.
// Deferral is an execution strategy, not a semantics: all handles must end with the same words.  The programs mix what
// the levelling has to get right: read-after-write, write-after-read and write-after-write on shared payloads,
// copy-on-write copies, in-place transforms, aliasing (a = a + b), random constructors, host reads and host writes in the
// middle of a queue, handles dying while operations on them are still queued.
// Usage: deferred_fuzz [rounds] [seed].  Exit code 0 = all rounds identical.  Needs a GPU.
#include <nfl.hpp>

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

template <class T, size_t Degree, size_t NbModuli> static bool run(unsigned rounds, unsigned seed0) {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  nfl::FastGaussianNoise<uint8_t, T, 2> fg(3.19, 100, 1 << 10);
  const size_t H = 12;
  unsigned char key[32];
  for (int i = 0; i < 32; i++) key[i] = (unsigned char)(17 * i + 3);
  for (unsigned round = 0; round < rounds; ++round) {
    std::vector<std::vector<T>> result[2];
    for (int mode = 0; mode < 2; ++mode) {
      nfl::set_deferred(mode == 0);
      nfl::set_sampler_key(key, 1000 + 100000ull * round);
      std::mt19937 rng(seed0 + round);
      auto pick = [&](size_t n) { return size_t(rng() % n); };
      std::vector<poly_p> h;
      for (size_t i = 0; i < H; ++i) h.emplace_back(nfl::uniform(uint64_t(i + 1 + round)));
      const unsigned steps = 60 + rng() % 200;
      for (unsigned s = 0; s < steps; ++s) {
        const size_t a = pick(H), b = pick(H), c = pick(H), d = pick(H);
        switch (rng() % 16) {
          case 0: h[a] = h[b] + h[c]; break;
          case 1: h[a] = h[b] - h[c]; break;
          case 2: h[a] = h[b] * h[c]; break;
          case 3: h[a] = h[b] * h[c] + h[d]; break;
          case 4: h[a] = h[a] + h[b] * h[a]; break;                    // aliasing
          case 5: h[a].ntt_pow_phi(); break;
          case 6: h[a].invntt_pow_invphi(); break;
          case 7: h[a] = G(&fg, 1 + rng() % 3); break;
          case 8: h[a] = nfl::uniform(); break;
          case 9: h[a] = (rng() & 1) ? poly_p{nfl::ZO_dist()} : poly_p{nfl::non_uniform(9)}; break;   // the old payload dies queued
          case 10: h[a] = h[b]; break;                                  // share; a later write to either detaches
          case 11: h[a](pick(NbModuli), pick(Degree)) = T(rng() % 1000); break;   // host write in the middle of a queue
          case 12: (void)const_cast<const poly_p &>(h[a])(0, 0); break; // host read in the middle of a queue
          case 13: h[a] = nfl::shoup(h[b] * h[c], nfl::compute_shoup(h[c])); break;
          case 14: {                                                    // temporaries that never reach the host
            poly_p t1 = h[b] + h[c], t2{G(&fg)};
            t2.ntt_pow_phi();
            h[a] = t1 * t2 - h[d];
          } break;
          default: h[a] = (h[b] + h[c]) * (h[d] - h[a]) + h[b] * h[d]; break;   // more than three leaves
        }
      }
      result[mode].resize(H);
      for (size_t i = 0; i < H; ++i) {
        const poly_t &p = const_cast<const poly_p &>(h[i]).poly_obj();
        result[mode][i].assign(p.begin(), p.end());
      }
      poly_p::synchronize();
    }
    for (size_t i = 0; i < H; ++i)
      if (result[0][i] != result[1][i]) {
        std::printf("FAIL round %u (seed %u): handle %zu differs between deferred and immediate execution\n", round, seed0 + round, i);
        return false;
      }
  }
  nfl::set_deferred(true);
  return true;
}

int main(int argc, char **argv) {
  const unsigned rounds = argc > 1 ? unsigned(std::atoi(argv[1])) : 40, seed = argc > 2 ? unsigned(std::atoi(argv[2])) : 12345;
  try {
    if (!run<uint64_t, 4096, 4>(rounds, seed)) return 1;
    if (!run<uint32_t, 1024, 2>(rounds, seed + 7)) return 1;
    if (!run<uint64_t, 1024, 1>(rounds, seed + 11)) return 1;
    if (!run<uint16_t, 128, 1>(rounds, seed + 13)) return 1;
    std::printf("deferred == immediate on %u random programs per ring\nall checks passed\n", rounds);
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
