// HOST-SIDE PROFILING AID (null backend, see make_null_backend.py): cost of each kind of deferred operation by itself.
#include <nfl.hpp>

#include <chrono>
#include <cstdio>
#include <vector>

template <class F> static double per_op(size_t n, F body) {
  using poly_p = nfl::poly_p<uint64_t, 4096, 4>;
  double best = 1e9;
  for (int round = 0; round < 5; ++round) {
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i++) body(i);
    poly_p::synchronize();
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (round && t < best) best = t;
  }
  return best / n * 1e9;
}

int main() {
  using T = uint64_t;
  using poly_p = nfl::poly_p<T, 4096, 4>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  const size_t N = 16384;
  nfl::FastGaussianNoise<uint8_t, T, 2> g_prng(4, 128, 1 << 10);
  std::vector<poly_p> x(N), y(N), z(N);
  poly_p key{nfl::uniform()};
  for (size_t i = 0; i < N; i++) { x[i] = poly_p{nfl::uniform()}; y[i] = poly_p{nfl::uniform()}; }
  poly_p::synchronize();
  std::printf("transform of an existing handle      %6.0f ns\n", per_op(N, [&](size_t i) { x[i].ntt_pow_phi(); }));
  std::printf("temporary{gaussian}, dropped         %6.0f ns\n", per_op(N, [&](size_t) { poly_p t{G(&g_prng, 2)}; }));
  std::printf("temporary{uniform}, dropped          %6.0f ns\n", per_op(N, [&](size_t) { poly_p t{nfl::uniform()}; }));
  std::printf("z = x * key + y (existing handles)   %6.0f ns\n", per_op(N, [&](size_t i) { z[i] = x[i] * key + y[i]; }));
  std::printf("z = x + y                            %6.0f ns\n", per_op(N, [&](size_t i) { z[i] = x[i] + y[i]; }));
  std::printf("temporary + transform + use          %6.0f ns\n", per_op(N, [&](size_t i) { poly_p t{G(&g_prng, 2)}; t.ntt_pow_phi(); z[i] = x[i] * key + t; }));
  return 0;
}
