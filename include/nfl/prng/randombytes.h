// nfl/prng/randombytes.h -- forwarding header: nfl::FastGaussianNoise, nfl::fastrandombytes, nfl::randombytes and nfl::rdtsc
// (FastGaussianNoise.hpp:117-122, 163-204; fastrandombytes.h:12; randombytes.h) are declared by the one header.
#ifndef NFL_HIP_FWD_PRNG_RANDOMBYTES_H
#define NFL_HIP_FWD_PRNG_RANDOMBYTES_H
#include "../../nfl.hpp"
#endif
