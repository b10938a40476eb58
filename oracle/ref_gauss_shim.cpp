// ref_gauss_shim.cpp -- reads the cumulative table ("barriers") the REAL reference builds for its Gaussian sampler
// (FastGaussianNoise::precomputeBarrierValues, FastGaussianNoise.hpp:295-366).
//
// TEST INFRASTRUCTURE ONLY, like ref_shim.cpp: no reference code here, only an instantiation of the reference's own
// class from the headers where they lie.  The table is a private member, so this file (and only this file) is
// compiled with -fno-access-control; see oracle/Makefile.
#include <nfl.hpp>

#include <cstdint>
#include <cstring>

extern "C" long nflref_gauss_barriers(double sigma, unsigned security, unsigned samples, double center, unsigned char *out,
                                      size_t cap, unsigned *bit_precision, unsigned *word_precision, unsigned *nbarriers,
                                      int *rounded_center) {
  nfl::FastGaussianNoise<uint8_t, uint64_t, 2> fg(sigma, security, samples, center);
  *bit_precision = fg._bit_precision;
  *word_precision = fg._word_precision;
  *nbarriers = fg._number_of_barriers;
  *rounded_center = fg.rounded_center;
  const size_t need = (size_t)fg._number_of_barriers * fg._word_precision;
  if (!out || cap < need) return -(long)need;
  for (unsigned i = 0; i < fg._number_of_barriers; i++)
    std::memcpy(out + (size_t)i * fg._word_precision, fg.barriers[i], fg._word_precision);   // big-endian bytes
  return (long)need;
}

#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdlib>

// Fork-replay of FastGaussianNoise::getNoise (FastGaussianNoise.hpp:477-595): `out` receives the rlen samples the
// reference produces, `raw` the uniform bytes its fastrandombytes() calls returned -- `ncalls` buffers of `*call_bytes`
// bytes each (the reference refills its buffer with a fresh call when it runs low; the child mirrors three calls, the
// parent consumes one or two of them).  The buffer size is the reference's own expression (lines 488-497) evaluated
// on the private counters of the same object.
extern "C" long nflref_gauss_replay(double sigma, unsigned security, unsigned samples, double center, uint64_t rlen,
                                    int64_t *out, unsigned char *raw, size_t raw_cap, uint64_t *call_bytes,
                                    uint64_t *call_words) {
  typedef nfl::FastGaussianNoise<uint8_t, uint64_t, 2> FG;
  FG fg(sigma, security, samples, center);
  float innoise_multiplier = 1.05 * ((float)(fg._lu_size - fg._flag_ctr1) / (float)fg._lu_size) +
                             2.0 * ((float)fg._flag_ctr1 / (float)fg._lu_size) +
                             fg._word_precision * ((float)fg._flag_ctr2 / ((float)fg._lu_size * fg._lu_size));
  const uint64_t innoise_words = rlen * innoise_multiplier;
  const uint64_t bytes = sizeof(uint8_t) * innoise_words;
  *call_bytes = bytes;
  *call_words = innoise_words;
  const int ncalls = 3;
  if (!raw || raw_cap < ncalls * bytes) return -(long)(ncalls * bytes);
  unsigned char one;
  nfl::fastrandombytes(&one, 1);   // the key is drawn at the first call: make sure that happened before the fork
  int fd[2];
  if (pipe(fd) != 0) return -1;
  const pid_t pid = fork();
  if (pid < 0) return -1;
  if (pid == 0) {
    close(fd[0]);
    unsigned char *buf = static_cast<unsigned char *>(malloc(bytes));
    for (int k = 0; k < ncalls; k++) {
      nfl::fastrandombytes(buf, bytes);
      size_t off = 0;
      while (off < bytes) {
        const ssize_t w = write(fd[1], buf + off, bytes - off);
        if (w <= 0) _exit(1);
        off += size_t(w);
      }
    }
    _exit(0);
  }
  close(fd[1]);
  size_t off = 0;
  while (off < ncalls * bytes) {
    const ssize_t r = read(fd[0], raw + off, ncalls * bytes - off);
    if (r <= 0) break;
    off += size_t(r);
  }
  close(fd[0]);
  int status = 0;
  waitpid(pid, &status, 0);
  if (off != ncalls * bytes) return -1;
  uint64_t *tmp = static_cast<uint64_t *>(malloc(rlen * sizeof(uint64_t)));
  fg.getNoise(tmp, rlen);
  for (uint64_t i = 0; i < rlen; i++) out[i] = (int64_t)tmp[i];
  free(tmp);
  return (long)(ncalls * bytes);
}
