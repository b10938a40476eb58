// nfl.hpp -- drop-in for the reference's umbrella header (include/nfl.hpp:14-20): existing callers keep writing
// `#include <nfl.hpp>` and get the same nfl::poly / nfl::poly_p template surface, implemented over the C ABI of
// nflhip.h (hand-written HIP for gfx950) instead of the SSE/AVX host loops.  The GMP-typed members are part of that
// surface (poly.hpp:249-307), so GMP is on unless NFL_HIP_NO_GMP is defined.
#ifndef NFL_HPP
#define NFL_HPP
#if !defined(NFL_HIP_NO_GMP) && !defined(NFL_HIP_WITH_GMP)
#define NFL_HIP_WITH_GMP 1
#endif
#include "nfl_hip/nfl.hpp"
#endif
