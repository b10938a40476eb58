// tests/cpp/sharded_main.cpp -- nfl::sharded_batch<P> (include/nfl_hip/nfl.hpp): a dense array of independent polynomials
// (how the reference's callers hold them, tests/tools.h:6-17) cut into contiguous shards over several GPUs from ONE
// process.  Everything a shard computes must be word for word what the same polynomials give in ONE device_batch:
//   * in-place generation (seeded operands and the random constructors: one keystream, every shard reads its positions),
//   * the batch operations (transforms, point-wise ops, fused programs, the fused product, b pre-transformed),
//   * upload / download of host arrays, scatter from / gather into a batch that lives on one device,
//   * the checksum of checksums: the shard digests add up to the digest of the whole batch,
// for batch sizes that do not divide by the number of devices and for more devices than polynomials (empty shards).
// Usage: sharded_test <device list, e.g. 0,1,2,3 or 0,0,0> [batch] [real].  Runs against the real library (GPU; on a 1-GPU box
// with several shards on device 0) and, on the CPU, against tests/cpp/mock with NFLHIP_MOCK_DEVICES virtual devices --
// whose buffers belong to one device each, so a shard enqueued on the wrong context fails loudly
// (tests/test_sharded_batch.py).  Exit code 0 = identical.
#include <nfl.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_fail = 0;
static bool g_real = false;   // argv[3] == "real": the library under test computes (the CPU stand-in of tests/cpp/mock does not)
#define CHECK(cond, what)                                          \
  do {                                                             \
    if (!(cond)) { std::printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++g_fail; } \
  } while (0)

template <class P> static std::vector<typename P::value_type> words(const nfl::device_batch<P> &b) {
  std::vector<P> h(b.size());
  if (b.size()) b.download(h.data());
  std::vector<typename P::value_type> w;
  for (auto &p : h) w.insert(w.end(), p.begin(), p.end());
  return w;
}
template <class P> static std::vector<typename P::value_type> words(const nfl::sharded_batch<P> &b) {
  std::vector<P> h(b.size());
  if (b.size()) b.download(h.data());
  std::vector<typename P::value_type> w;
  for (auto &p : h) w.insert(w.end(), p.begin(), p.end());
  return w;
}

template <class T, size_t Degree, size_t NbModuli> static void run(const std::vector<int> &devs, size_t B, const char *name) {
  using P = nfl::poly<T, Degree, NbModuli>;
  using DB = nfl::device_batch<P>;
  using SB = nfl::sharded_batch<P>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  const int dev0 = devs[0];
  unsigned char key[32];
  for (int i = 0; i < 32; i++) key[i] = (unsigned char)(17 * i + 3);

  // the whole batch on ONE device: the reference result
  DB a1(B, dev0), b1(B, dev0), c1(B, dev0), t1(B, dev0);
  a1.set(nfl::uniform(0x1234));
  b1.set(nfl::uniform(0x9876));
  // ... and split over the devices, generated in place
  SB a(B, devs), b(B, devs), c(B, devs), t(B, devs);
  a.set(nfl::uniform(0x1234));
  b.set(nfl::uniform(0x9876));
  CHECK(a.shards() == devs.size(), "one shard per device");
  size_t covered = 0;
  for (size_t r = 0; r < a.shards(); ++r) {
    CHECK(a.first(r) == covered, "contiguous shards");
    CHECK(a.shard(r).device() == devs[r], "shard r lives on device r");
    covered += a.count(r);
  }
  CHECK(covered == B, "the shards cover the batch");
  CHECK(words(a) == words(a1) && words(b) == words(b1), "seeded operands generated in place, shard by shard");

  // the metric path, b pre-transformed, transforms, point-wise, a fused program
  c1.assign_polymul(a1, b1);
  c.assign_polymul(a, b);
  CHECK(words(c) == words(c1), "fused product");
  CHECK(c.digest() == c1.digest(), "checksum of checksums: shard digests add up to the batch digest");
  {
    uint64_t s = 0;
    for (uint64_t d : c.digests()) s += d;
    CHECK(s == c1.digest(), "sum of per-shard digests");
  }
  b1.ntt_pow_phi();
  b.ntt_pow_phi();
  CHECK(words(b) == words(b1), "forward transform");
  t1.assign_polymul_ntt(a1, b1);
  t.assign_polymul_ntt(a, b);
  CHECK(words(t) == words(t1), "product with b pre-transformed");
  if (g_real) CHECK(words(t) == words(c), "... which is the product");   // (arithmetic, not plumbing: the toy device skips it)
  b1.invntt_pow_invphi();
  b.invntt_pow_invphi();
  CHECK(words(b) == words(b1), "inverse transform");
  t1.assign(NFLHIP_OP_ADD, a1, b1);
  t.assign(NFLHIP_OP_ADD, a, b);
  CHECK(words(t) == words(t1), "point-wise add");
  t1.assign(NFLHIP_OP_MUL, t1, b1);   // aliasing
  t.assign(NFLHIP_OP_MUL, t, b);
  CHECK(words(t) == words(t1), "point-wise mul, aliased");
  {
    const unsigned char prog[] = {0, 1, NFLHIP_EXPR_MUL, 2, NFLHIP_EXPR_SUB};   // a*b - c
    const DB *o1[] = {&a1, &b1, &c1};
    const SB *os[] = {&a, &b, &c};
    t1.assign_program(prog, sizeof prog, o1, 3);
    t.assign_program(prog, sizeof prog, os, 3);
    CHECK(words(t) == words(t1), "fused expression program");
  }
  CHECK(!t.any_differs(t) && (B == 0 || t.any_equal(t)), "comparisons over shards");
  if (B) CHECK(a.any_differs(b), "a != b");

  // the random constructors: one keystream for the logical batch
  nfl::FastGaussianNoise<uint8_t, T, 2> fg(4, 128, 1 << 10);
  nfl::set_sampler_key(key, 77);
  t1.set(nfl::non_uniform(1000));
  nfl::set_sampler_key(key, 77);
  t.set(nfl::non_uniform(1000));
  CHECK(words(t) == words(t1), "non_uniform drawn shard by shard");
  nfl::set_sampler_key(key, 78);
  t1.set(G(&fg, 2));
  nfl::set_sampler_key(key, 78);
  t.set(G(&fg, 2));
  CHECK(words(t) == words(t1), "gaussian drawn shard by shard");
  nfl::set_sampler_key(key, 79);
  t1.set(nfl::uniform());
  nfl::set_sampler_key(key, 79);
  t.set(nfl::uniform());
  CHECK(words(t) == words(t1), "uniform drawn shard by shard");
  nfl::set_sampler_key(key, 80);
  t1.set(nfl::ZO_dist());
  nfl::set_sampler_key(key, 80);
  t.set(nfl::ZO_dist());
  CHECK(words(t) == words(t1), "ZO_dist drawn shard by shard");

  // host array -> shards -> host array; one-device batch -> shards (peer copies) -> one-device batch
  {
    std::vector<P> h(B), back(B);
    if (B) c1.download(h.data());
    SB u(B, devs);
    if (B) u.upload(h.data());
    CHECK(words(u) == words(c1), "upload, every device its slice");
    SB v(B, devs);
    v.scatter(c1);
    CHECK(words(v) == words(c1), "scatter from the batch on one device");
    v.ntt_pow_phi();               // work on the shards, then bring the result home
    DB home(B, dev0);
    v.gather(home);
    home.sync();
    c1.ntt_pow_phi();
    CHECK(words(home) == words(c1), "gather into the batch on one device");
    c1.invntt_pow_invphi();
    if (devs.size() > 1 && devs.back() != dev0) {   // the root need not be the first device
      DB far(B, devs.back());
      far.set(nfl::uniform(0x1234));
      SB w(B, devs);
      w.scatter(far);
      CHECK(words(w) == words(a1), "scatter from the last device");
      DB far2(B, devs.back());
      w.gather(far2);
      far2.sync();
      CHECK(words(far2) == words(a1), "gather into the last device");
    }
  }
  // one polynomial replicated over every shard
  {
    P one(nfl::uniform(0x5555));
    t1.fill(one);
    t.fill(one);
    CHECK(words(t) == words(t1), "fill");
  }
  // misuse is an exception, not a wrong answer
  if (devs.size() > 1) {
    bool threw = false;
    try {
      std::vector<int> fewer(devs.begin(), devs.end() - 1);
      SB other(B, fewer);
      t.assign_polymul(a, other);
    } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw, "batches split differently do not mix");
    if (devs[0] != devs[1]) {
      threw = false;
      try { a.shard(0).assign_polymul(a.shard(0), b.shard(1)); } catch (const std::runtime_error &) { threw = true; }
      CHECK(threw || a.count(0) != a.count(1), "batches of different devices do not mix");
    }
  }
  std::printf("%s: batch %zu over %zu shards ok\n", name, B, devs.size());
}

int main(int argc, char **argv) {
  std::vector<int> devs;
  const std::string list = argc > 1 ? argv[1] : "0";
  for (size_t i = 0; i < list.size();) {
    size_t j = list.find(',', i);
    if (j == std::string::npos) j = list.size();
    devs.push_back(std::atoi(list.substr(i, j - i).c_str()));
    i = j + 1;
  }
  const size_t B = argc > 2 ? size_t(std::atol(argv[2])) : 37;
  g_real = argc > 3 && std::string(argv[3]) == "real";
  try {
    std::printf("devices visible: %d\n", nfl::device_count());
    run<uint64_t, 4096, 4>(devs, B, "u64/4096/4");
    run<uint64_t, 4096, 4>(devs, devs.size() > 2 ? devs.size() - 2 : 1, "u64/4096/4 (fewer polynomials than devices)");
    run<uint32_t, 1024, 2>(devs, B + 4, "u32/1024/2");
    run<uint16_t, 128, 1>(devs, 3 * B, "u16/128/1");
    if (g_fail) {
      std::printf("%d checks FAILED\n", g_fail);
      return 1;
    }
    std::printf("all checks passed\n");
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
