// tests/cpp/deferred_edges.cpp -- the corners of the deferred queue (include/nfl_hip/nfl.hpp, detail::lazy) that loops and
// random programs do not reach:
//   1. a FastGaussianNoise that dies BEFORE the polynomials built from it are used (legal in the reference, which samples
//      inside the constructor: core.hpp:283-328): its device table must outlive the recorded draws;
//   2. nfl::set_sampler_key between a random constructor and the queue run: the draw is made with the key it was
//      recorded under (deferred == immediate);
//   3. a launch that fails in the middle of a queue run: what ran holds its value, what never ran THROWS on access
//      instead of handing out uninitialised HBM, and overwriting such a handle makes it usable again.  (Needs the CPU
//      stand-in's failure injection, tests/cpp/mock: skipped against the real library.)
//   4. the same inside a run that started by itself (on the queue's own thread): everything recorded since is poisoned too;
//   5. fork() while the queue's thread exists: the child executes its runs itself.  (4 and 5: CPU stand-in only.)
// Usage: deferred_edges.  Exit code 0 = all checks passed.
#include <nfl.hpp>

#include <cstdio>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

extern "C" void mock_fail_after(long) __attribute__((weak));   // only the CPU stand-in defines it

static int g_fail = 0;
#define CHECK(cond, what)                                          \
  do {                                                             \
    if (!(cond)) { std::printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++g_fail; } \
  } while (0)

template <class T, size_t Degree, size_t NbModuli> struct ring {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  using G = nfl::gaussian<uint8_t, T, 2>;

  static poly_p scoped_noise(uint64_t amp) {   // the generator is gone when the caller looks at the polynomial
    nfl::FastGaussianNoise<uint8_t, T, 2> fg(4, 128, 1 << 10);
    return poly_p(G(&fg, amp));
  }
  static std::vector<T> words(const poly_p &p) {
    const poly_t &v = p.poly_obj();
    return std::vector<T>(v.begin(), v.end());
  }

  static void run() {
    unsigned char k1[32], k2[32];
    for (int i = 0; i < 32; i++) { k1[i] = (unsigned char)(3 * i + 1); k2[i] = (unsigned char)(5 * i + 2); }
    std::vector<std::vector<T>> res[2];
    for (int mode = 0; mode < 2; ++mode) {
      nfl::set_deferred(mode == 0);
      nfl::set_sampler_key(k1, 100);
      // 1. scoped generators, several of them, interleaved with other work
      poly_p a = scoped_noise(1), b = scoped_noise(2);
      poly_p u{nfl::uniform()};
      poly_p c = a + b * u;
      poly_p d = scoped_noise(1);
      res[mode].push_back(words(c));
      res[mode].push_back(words(d));
      res[mode].push_back(words(a));
      // 2. the key changes between the constructor and the first use
      poly_p e{nfl::non_uniform(1000)};
      poly_p f{nfl::uniform()};
      nfl::set_sampler_key(k2, 7);
      poly_p g{nfl::non_uniform(1000)};
      res[mode].push_back(words(e));
      res[mode].push_back(words(f));
      res[mode].push_back(words(g));
      poly_p::synchronize();
    }
    nfl::set_deferred(true);
    CHECK(res[0].size() == res[1].size(), "same number of saved values");
    for (size_t i = 0; i < res[0].size() && i < res[1].size(); ++i) CHECK(res[0][i] == res[1][i], "deferred == immediate");
    CHECK(res[0][3] != res[0][5], "the two keys give different draws");

    // 3. a failing launch inside a queue run
    if (mock_fail_after) {
      nfl::set_sampler_key(k1, 500);
      poly_p x{nfl::uniform()}, y{nfl::uniform()};
      (void)words(x); (void)words(y);                 // the queue is empty, x and y hold values
      const std::vector<T> x0 = words(x);
      poly_p s = x + y;                               // level 0
      poly_p t = s * y;                               // level 1
      poly_p w = t - x;                               // level 2
      mock_fail_after(1);                             // the queue's SECOND launch fails
      bool threw = false;
      try { (void)words(w); } catch (const std::runtime_error &) { threw = true; }
      mock_fail_after(-1);
      CHECK(threw, "the failing launch surfaces as the reference's exception type");
      bool ok_s = true, bad_t = false, bad_w = false;
      std::vector<T> sv;
      try { sv = words(s); } catch (const std::runtime_error &) { ok_s = false; }
      try { (void)words(t); } catch (const std::runtime_error &) { bad_t = true; }
      try { (void)words(w); } catch (const std::runtime_error &) { bad_w = true; }
      CHECK(ok_s, "what was launched before the failure holds its value");
      CHECK(bad_t && bad_w, "what never ran throws on access instead of returning uninitialised memory");
      threw = false;
      try { poly_p z = t + x; (void)words(z); } catch (const std::runtime_error &) { threw = true; }
      CHECK(threw, "... also as an operand");
      CHECK(words(x) == x0, "operands are untouched");
      t = x + y;                                      // overwritten entirely: usable again
      CHECK(words(t) == sv, "a handle that is overwritten entirely is usable again");
      poly_p::synchronize();

      // 4. a launch that fails inside a run that started BY ITSELF (executed by the queue's own thread unless
      //    NFL_HIP_QUEUE_THREAD=0): the error surfaces at a later record or at the next access, what the run never launched
      //    throws on access -- and so does what was recorded on top of it while the run was in flight
      const std::vector<T> sum = words(poly_p(x + y));
      std::vector<poly_p> v(3000);
      mock_fail_after(0);                             // the next launch fails: the first launch of the loop's first run
      size_t threw_at = v.size();
      for (size_t i = 0; i < v.size(); ++i) {
        try { v[i] = x + y; } catch (const std::runtime_error &) { threw_at = i; mock_fail_after(-1); break; }
      }
      if (threw_at == v.size()) {                     // (a queue longer than the loop: the failure surfaces at the first access)
        threw = false;
        try { (void)words(v[0]); } catch (const std::runtime_error &) { threw = true; }
        mock_fail_after(-1);
        CHECK(threw, "the failure of a run surfaces at the next access");
      } else {
        CHECK(threw_at > 0, "the failure of a run that started by itself surfaces at a later record");
      }
      bad_t = false;
      try { (void)words(v[0]); } catch (const std::runtime_error &) { bad_t = true; }
      CHECK(bad_t, "what the failed run never launched throws on access");
      if (threw_at > 1 && threw_at < v.size()) {
        bad_w = false;
        try { (void)words(v[threw_at - 1]); } catch (const std::runtime_error &) { bad_w = true; }
        CHECK(bad_w, "what was recorded while the failed run was in flight throws on access");
      }
      for (size_t i = 0; i < v.size(); ++i) v[i] = x + y;   // overwritten entirely: usable again, and the queue works as before
      CHECK(words(v[0]) == sum && words(v[v.size() - 1]) == sum && words(v[v.size() / 2]) == sum, "the queue works again after a failed run");
      poly_p::synchronize();

      // 5. fork() with the queue's thread alive: the child has no such thread and executes its runs itself
      const pid_t pid = fork();
      if (pid == 0) {
        bool ok = true;
        try {
          std::vector<poly_p> c(2500);
          for (size_t i = 0; i < c.size(); ++i) c[i] = x + y;
          ok = words(c[0]) == sum && words(c[c.size() - 1]) == sum;
        } catch (...) {
          ok = false;
        }
        _exit(ok ? 0 : 3);
      }
      int status = -1;
      CHECK(pid > 0 && waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0,
            "a forked child records and runs its queue without the parent's thread");
    }
  }
};

int main() {
  try {
    ring<uint64_t, 4096, 4>::run();
    ring<uint32_t, 1024, 2>::run();
    if (g_fail) {
      std::printf("%d checks FAILED\n", g_fail);
      return 1;
    }
    std::printf("%s\nall checks passed\n", mock_fail_after ? "with failure injection" : "without failure injection (real library)");
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
