// nfl/debug.hpp -- forwarding header: the reference splits its surface over include/nfl/*.hpp and callers include
// the pieces directly (tests/poly_p.cpp:2, tests/nfllib_demo_main.hpp:5); here every piece is the one header.
#ifndef NFL_HIP_FWD_DEBUG_HPP
#define NFL_HIP_FWD_DEBUG_HPP
#include "../nfl.hpp"
#endif
