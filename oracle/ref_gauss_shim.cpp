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
