// nfl/opt/ops.hpp -- forwarding header (the SIMD specialisations it held are what the device kernels replace)
#ifndef NFL_HIP_FWD_OPT_OPS_HPP
#define NFL_HIP_FWD_OPT_OPS_HPP
#include "../../nfl.hpp"
#endif
