// nfl/arch/common.hpp -- forwarding header (nfl::simd::serial and common_mode live in nfl_hip/nfl.hpp)
#ifndef NFL_HIP_FWD_ARCH_COMMON_HPP
#define NFL_HIP_FWD_ARCH_COMMON_HPP
#include "../../nfl.hpp"
#endif
