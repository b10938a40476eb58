// nfl/prng/FastGaussianNoise.hpp -- forwarding header: nfl::FastGaussianNoise, nfl::fastrandombytes, nfl::randombytes and nfl::rdtsc
// (FastGaussianNoise.hpp:117-122, 163-204; fastrandombytes.h:12; randombytes.h) are declared by the one header.
#ifndef NFL_HIP_FWD_PRNG_FASTGAUSSIANNOISE_HPP
#define NFL_HIP_FWD_PRNG_FASTGAUSSIANNOISE_HPP
#include "../../nfl.hpp"
#endif
