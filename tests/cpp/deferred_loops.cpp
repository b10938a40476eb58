// tests/cpp/deferred_loops.cpp -- the LOOP shapes the deferred queue is built for (include/nfl_hip/nfl.hpp, detail::lazy),
// each executed twice with the same pinned sampler state: deferred (queued, levelled, grouped, coalesced into batched
// strided / sequence launches) and immediate.  All results must be identical.  Complements deferred_fuzz.cpp, whose random
// programs over 12 handles never build the long dense runs that loops do:
//   1. the reference's LWE demo loop (tests/nfllib_demo_main_op.cpp:26-58): per iteration three Gaussian temporaries (two
//      of them from one amplifier: stream ids interleave with period 2), three transforms, two fused multiply-adds with a
//      key operand each; then the decryption loop over the results;
//   2. results written into every second / in reverse order of a pre-allocated array (strides other than 1, unsorted
//      destinations), operands shared between neighbours (stride 0 for a few elements, then a new key);
//   3. a queue that runs by itself in the middle of the loop (NFL_HIP_QUEUE_LIMIT small) and handles that die queued;
//   4. the sequences the transform fusion rewrites (sample, transform, multiply-add; multiply-add, inverse transform) in
//      every operand order, next to look-alikes it must leave alone: a third reader of a transformed temporary, a
//      temporary whose handle survives, a key rewritten between two results, a result feeding the next, a sum that is
//      read before it is transformed back;
//   5. transforms that join the record of the operation that produced their operand (detail::lazy::join_transform) next
//      to the ones that must stay records of their own: a reader between the two, two transforms in a row, a transform
//      after a queue run, a constructor / transform pair of the wrong kinds, a value written twice, a fused
//      multiply-add transformed back at once.
// Usage: deferred_loops [reps].  Exit code 0 = identical.  Runs against the real library (GPU) and against the toy
// arithmetic of tests/cpp/mock (CPU: tests/test_host_logic.py).
#include <nfl.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <class T, size_t Degree, size_t NbModuli> static bool run(size_t reps) {
  using poly_t = nfl::poly<T, Degree, NbModuli>;
  using poly_p = nfl::poly_p<T, Degree, NbModuli>;
  using G = nfl::gaussian<uint8_t, T, 2>;
  nfl::FastGaussianNoise<uint8_t, T, 2> fg(4, 128, 1 << 10);
  unsigned char key[32];
  for (int i = 0; i < 32; i++) key[i] = (unsigned char)(29 * i + 5);
  std::vector<std::vector<T>> result[2];
  for (int mode = 0; mode < 2; ++mode) {
    nfl::set_deferred(mode == 0);
    nfl::set_sampler_key(key, 4242);
    std::vector<poly_p> keep;
    auto save = [&](const poly_p &p) {
      const poly_t &v = p.poly_obj();
      result[mode].emplace_back(v.begin(), v.end());
    };
    {  // 1. LWE
      poly_p s{G(&fg)};
      s.ntt_pow_phi();
      poly_p pka{nfl::uniform()}, pkb{G(&fg, 2)};
      pkb.ntt_pow_phi();
      std::vector<poly_p> ra(reps), rb(reps), dec(reps);
      for (size_t i = 0; i < reps; ++i) {
        poly_p u{G(&fg)}, e1{G(&fg, 2)}, e2{G(&fg, 2)};
        u.ntt_pow_phi();
        e1.ntt_pow_phi();
        e2.ntt_pow_phi();
        ra[i] = u * pka + e1;
        rb[i] = u * pkb + e2;
      }
      for (size_t i = 0; i < reps; ++i) {
        dec[i] = rb[i] - ra[i] * s;
        dec[i].invntt_pow_invphi();
      }
      for (size_t i = 0; i < reps; i += 7) save(dec[i]);
      save(ra[reps - 1]);
      save(rb[0]);
    }
    {  // 2. strides and orders
      const size_t n = reps | 1;
      std::vector<poly_p> src(n), dst(2 * n), keys(3);
      for (auto &k : keys) k = poly_p{nfl::uniform()};
      for (size_t i = 0; i < n; ++i) src[i] = poly_p{nfl::non_uniform(17)};
      for (size_t i = 0; i < 2 * n; ++i) dst[i] = poly_p{nfl::ZO_dist()};       // buffers exist before the loops below
      (void)const_cast<const poly_p &>(dst[0])(0, 0);                            // ... (a host read runs the queue)
      for (size_t i = 0; i < n; ++i) dst[2 * i] = src[i] * keys[i * 3 / n] + dst[2 * i + 1];   // every second result; key changes twice
      for (size_t i = n; i-- > 0;) dst[2 * i + 1] = dst[2 * i] - src[n - 1 - i];                // reverse order, reversed operand
      for (size_t i = 0; i + 1 < n; ++i) src[i] = src[i] + src[i + 1];                          // chain: every step reads what the next one overwrites
      for (size_t i = 0; i < 2 * n; i += 5) save(dst[i]);
      for (size_t i = 0; i < n; i += 3) save(src[i]);
    }
    {  // 3. handles that die while queued, copies, in-place transforms of shared payloads
      poly_p acc{nfl::uniform()};
      for (size_t i = 0; i < reps; ++i) {
        poly_p t{G(&fg, 1 + i % 3)};
        poly_p c = t;                 // shares t's payload
        c.ntt_pow_phi();              // detaches
        acc = acc + c * t;
        if (i % 64 == 63) keep.push_back(acc);
      }
      save(acc);
      for (auto &k : keep) save(k);
    }
    {  // 4. shapes around the transform fusion (detail::lazy::fuse): sequences it may rewrite and sequences it must leave alone
      poly_p s{G(&fg)}, k1{nfl::uniform()}, k2{nfl::uniform()};
      s.ntt_pow_phi();
      const size_t m = reps / 4 + 3;
      std::vector<poly_p> r0(m), r1(m), r2(m), alive;
      for (size_t i = 0; i < m; ++i) {
        poly_p u{G(&fg)}, e1{G(&fg, 2)}, e2{G(&fg, 3)};
        u.ntt_pow_phi();
        e1.ntt_pow_phi();
        e2.ntt_pow_phi();
        switch (i % 6) {
          case 0: r0[i] = e1 + k1 * u; r1[i] = k2 * u + e2; break;                       // operand orders
          case 1: r0[i] = u * k1 + e1; r1[i] = u * k2 + e2; r2[i] = u + e1; break;        // a third reader of NTT(u) and NTT(e1)
          case 2: r0[i] = u * k1 + e1; alive.push_back(u); r1[i] = u * k2 + e2; break;    // the temporary outlives the run
          case 3: r0[i] = u * k1 + e1; k1 = k1 + k2; r1[i] = u * k1 + e2; break;          // the key changes between the two results
          case 4: r0[i] = u * k1 + e1; r1[i] = u * r0[i] + e2; break;                     // the second result reads the first
          default: u = u * k1 + e1; r0[i] = u; r1[i] = e2 * k2 + e2; break;               // in place; e2 twice in one expression
        }
      }
      for (size_t i = 0; i < m; ++i) {
        poly_p d, t;
        switch (i % 4) {
          case 0: r1[i] = r1[i] - r0[i] * s; r1[i].invntt_pow_invphi(); break;            // in place
          case 1: d = r0[i] * s + r1[i]; t = d; d.invntt_pow_invphi(); r1[i] = d + t; break;   // the sum is read before its transform
          case 2: d = r1[i] - r0[i] * s; r0[i] = r0[i] + s; d.invntt_pow_invphi(); r1[i] = d; break;  // an operand changes in between
          default: d = r1[i] + s * r0[i]; d.invntt_pow_invphi(); r1[i] = d; break;
        }
      }
      for (size_t i = 0; i < m; ++i) {
        save(r0[i]);
        save(r1[i]);
        if (i % 6 == 1) save(r2[i]);
      }
      for (auto &a : alive) save(a);
    }
    {  // 5. shapes around the joined transforms
      poly_p k1{nfl::uniform()}, k2{nfl::uniform()}, c0{nfl::uniform()};
      const size_t m = reps / 4 + 5;
      std::vector<poly_p> r0(m), r1(m);
      for (size_t i = 0; i < m; ++i) {
        poly_p u{G(&fg)}, e1{G(&fg, 2)};
        switch (i % 13) {
          case 0: r1[i] = u + k1; u.ntt_pow_phi(); r0[i] = u; break;                                    // u is read before its transform
          case 1: u.ntt_pow_phi(); u.invntt_pow_invphi(); r0[i] = u; r1[i] = e1; break;                 // two transforms in a row
          case 2: r0[i] = u + e1; r0[i].ntt_pow_phi(); r1[i] = r0[i] * k1; r1[i].ntt_pow_phi(); r1[i].ntt_pow_phi(); break;   // expression, then forward (twice)
          case 3: r0[i] = k1 * k2 + c0; r0[i] = r0[i] * k2 + u; r0[i].invntt_pow_invphi(); r1[i] = r0[i] + e1; break;   // written twice, in place
          case 4: u.ntt_pow_phi(); e1.ntt_pow_phi(); r0[i] = u * k1 + e1; r0[i].invntt_pow_invphi(); r1[i] = r0[i]; break;   // fused, back at once
          case 5: if (i % 2) poly_p::synchronize(); u.ntt_pow_phi(); r0[i] = u; r1[i] = e1; break;      // (sometimes) after a queue run
          case 6: u.invntt_pow_invphi(); r0[i] = u; { poly_p w{nfl::uniform()}; w.ntt_pow_phi(); r1[i] = w; } break;   // wrong kinds
          case 7: { poly_p c = u; u.ntt_pow_phi(); r0[i] = u; r1[i] = c; } break;                       // a copy shares the value
          case 8: r0[i] = c0 - k1 * k2; r1[i] = r0[i] + u; r0[i].invntt_pow_invphi(); break;            // the difference is read first
          case 9: r0[i] = k1 * k2 + c0; r0[i].ntt_pow_phi(); r0[i].invntt_pow_invphi(); r1[i] = r0[i] + e1; break;   // c + a*b, forward (joined), then inverse: NOT c + a*b -> inverse
          case 10: u.ntt_pow_phi(); u.ntt_pow_phi(); e1.ntt_pow_phi(); r0[i] = u * k1 + e1; r1[i] = k2; break;   // sampled, transformed TWICE, multiplied
          case 11: { poly_p e2{G(&fg, 2)}; u.ntt_pow_phi(); e1.ntt_pow_phi(); e2.ntt_pow_phi();       // both results, a transform joined to the FIRST
                     r0[i] = u * k1 + e1; r0[i].ntt_pow_phi(); r1[i] = u * k2 + e2; } break;
          default: { poly_p e2{G(&fg, 2)}; u.ntt_pow_phi(); e1.ntt_pow_phi(); e2.ntt_pow_phi();       // ... to the SECOND
                     r0[i] = u * k1 + e1; r1[i] = u * k2 + e2; r1[i].invntt_pow_invphi(); } break;
        }
      }
      for (size_t i = 0; i < m; ++i) {
        save(r0[i]);
        save(r1[i]);
      }
    }
    poly_p::synchronize();
  }
  nfl::set_deferred(true);
  if (result[0].size() != result[1].size()) return false;
  for (size_t i = 0; i < result[0].size(); ++i)
    if (result[0][i] != result[1][i]) {
      std::printf("FAIL: saved value %zu differs between deferred and immediate execution\n", i);
      return false;
    }
  return true;
}

int main(int argc, char **argv) {
  const size_t reps = argc > 1 ? size_t(std::atol(argv[1])) : 300;
  try {
    if (!run<uint64_t, 4096, 4>(reps)) return 1;
    if (!run<uint64_t, 8192, 2>(reps / 4 + 8)) return 1;      // (the fused entries' row-resident kernels on the GPU)
    if (!run<uint64_t, 32768, 2>(reps / 16 + 6)) return 1;    // (... and the 32768-word rows' pipelines)
    if (!run<uint32_t, 1024, 2>(reps)) return 1;
    if (!run<uint16_t, 128, 1>(reps)) return 1;
    std::printf("deferred == immediate on the loop shapes, %zu iterations per ring\nall checks passed\n", reps);
    return 0;
  } catch (const std::exception &ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
}
