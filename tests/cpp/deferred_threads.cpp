// tests/cpp/deferred_threads.cpp -- several host threads, each working on its OWN poly_p handles of one ring type.  The
// deferred queue, the stream and the buffer pool behind the handles are shared (include/nfl_hip/nfl.hpp, detail::lazy /
// detail::context): a queue run started by one thread executes -- and retires -- the operations the others recorded.
// Every thread's results must equal those of the same program run alone.  What the programs stress: copy-on-write
// decisions (`c = t; c.ntt_pow_phi()` must leave t alone) taken while another thread's queue run drops the queue's
// references, host reads that run the queue in the middle of other threads' recordings, temporaries dying queued.
// Random constructors use explicit seeds (the implicit stream ids come from one process-wide counter).
// Usage: deferred_threads [threads] [iterations].  Exit code 0 = identical.  Runs against the real library (GPU) and,
// under ThreadSanitizer, against the toy arithmetic of tests/cpp/mock (tests/test_host_logic.py).
#include <nfl.hpp>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

template <class T, size_t D, size_t M> static void program(int id, int iters, std::vector<std::vector<T>> *out) {
  using poly_t = nfl::poly<T, D, M>;
  using poly_p = nfl::poly_p<T, D, M>;
  auto save = [&](const poly_p &p) {
    const poly_t &v = p.poly_obj();
    out->emplace_back(v.begin(), v.end());
  };
  poly_p key{nfl::uniform(uint64_t(100 + id))}, acc{nfl::uniform(uint64_t(200 + id))};
  std::vector<poly_p> ring(5);
  for (int i = 0; i < iters; ++i) {
    poly_p t{nfl::uniform(uint64_t(100000 * (id + 1) + i))};
    poly_p c = t;                      // shares t's payload
    c.ntt_pow_phi();                   // must detach: t keeps its value
    acc = acc + c * key - t;
    ring[size_t(i) % ring.size()] = t; // the old occupant dies, possibly while operations on it are queued
    if (i % 5 == 2) ring[size_t(i + 1) % ring.size()] = ring[size_t(i) % ring.size()] * acc;
    if (i % (29 + id) == 7) save(acc); // host read: runs the queue, whoever recorded into it
    if (i % 41 == 11) acc(0, size_t(i) % D) = T(i % 251);   // host write in the middle
  }
  save(acc);
  for (auto &r : ring) save(r);
}

template <class T, size_t D, size_t M> static bool run(int threads, int iters) {
  const size_t nt = size_t(threads);
  std::vector<std::vector<std::vector<T>>> par(nt), seq(nt);
  {
    std::vector<std::thread> th;
    for (int i = 0; i < threads; ++i) th.emplace_back(program<T, D, M>, i, iters, &par[size_t(i)]);
    for (auto &t : th) t.join();
  }
  for (int i = 0; i < threads; ++i) program<T, D, M>(i, iters, &seq[size_t(i)]);
  for (int i = 0; i < threads; ++i)
    if (par[size_t(i)] != seq[size_t(i)]) {
      std::printf("FAIL: thread %d's results differ from the same program run alone\n", i);
      return false;
    }
  return true;
}

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 4, iters = argc > 2 ? std::atoi(argv[2]) : 300;
  try {
    if (!run<uint64_t, 1024, 2>(threads, iters)) return 1;
    if (!run<uint32_t, 1024, 1>(threads, iters)) return 1;
    std::printf("%d threads x %d iterations: every thread's results equal the single-threaded run\nall checks passed\n", threads, iters);
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
