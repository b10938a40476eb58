// nfl_hip/nfl.hpp -- header-only host surface of the MI355X NTT polynomial-ring engine.
//
// Keeps the template surface of the reference's nfl::poly<T, Degree, NbModuli>
// (include/nfl/poly.hpp:82-352), nfl::poly_p (include/nfl/poly_p.hpp:11-204) and their
// expression-template operators (include/nfl/ops.hpp:18-97, 249-277) so existing callers compile
// unchanged -- the reference's own test programs build against this header as they are
// (tests/reftests/Makefile) -- but every whole-polynomial operation is forwarded through the C ABI of
// include/nflhip.h to hand-written HIP kernels.  There is no CPU arithmetic on the hot path in this
// header: without libnflhip.so + a GPU every operation throws std::runtime_error (the reference's error
// convention: core.hpp:111-115).  Reached as <nfl.hpp>, <nfl/poly.hpp>, <nfl/poly_p.hpp>, ... through the
// forwarding headers of include/nfl/.
//
// What is kept (same names, argument meaning, error behaviour):
//   params<T>::P / Pn / primitive_roots / invkMaxPolyDegree / kMax* (all moduli)  params.hpp:11-119
//   storage layout T _data[NbModuli*Degree], 32-byte aligned, modulus-major      poly.hpp:87-88,156-157
//   ctors / set(): value, initializer_list, iterator range (+reduce_coeffs)       core.hpp:64-137
//   the random constructors uniform / non_uniform / ZO_dist / hwt_dist / gaussian core.hpp:146-391
//   operator()(cm,i), begin/end, data(), get_modulus, degree/nmoduli/nbits        poly.hpp:142-162
//   ntt_pow_phi(), invntt_pow_invphi()                                            poly.hpp:167-168
//   operator+ - * == !=, shoup(a*b,b'), compute_shoup(b), nested expressions      poly.hpp:346-352
//   ops::make_op<ops::NAME<T, tag>>(...), nfl::simd::serial, CC_SIMD             ops.hpp:225-260, arch/common.hpp:11-26
//   explicit operator bool on polys, implicit on == / != expressions             core.hpp:39-43, ops.hpp:81-95
//   poly::core (ntt / inv_ntt statics) + `base` tables through the
//   tests::poly_tests_proxy friend                                                poly.hpp:69-76,85,196-247
//   serialize_manually / deserialize_manually / serialize(Archive)                poly.hpp:180-191
//   the GMP-typed surface (mpz_t / mpz_class ctors, set_mpz, poly2mpz, mpz2poly,
//   moduli_product / modulus_shoup / lifting_integers), on poly and poly_p        poly.hpp:249-307, poly_p.hpp:186-200
//   nfl::add / sub / mul, poly_from_modulus, poly_p_from_modulus, operator<<      poly.hpp:314-343
//   FastGaussianNoise (ctor, getNoise), nfl::rdtsc, nfl::fastrandombytes          FastGaussianNoise.hpp, fastrandombytes.h
// What differs, on purpose:
//   * tables live in a lazily created per-(T,Degree,NbModuli) device context, never at static-init time
//     (the reference's `static core base`, poly.hpp:247);
//   * nfl::poly_p is RESIDENT: its shared payload lives in HBM next to an (optional) host image, with a
//     valid-on-host / valid-on-device pair of bits.  Operator expressions, transforms, comparisons and random
//     constructors on poly_p handles run the *_dev entry points on the context's stream and never touch the host;
//     only operator()(cm,i), poly_obj(), serialisation and the GMP surface force a device-to-host copy
//     (SURVEY.md section 8(f) rank 2).  nfl::poly keeps its words inline on the host (poly.hpp:87-88) and its
//     member operations therefore cross PCIe per call -- use poly_p or the batch forms for throughput;
//   * CRT lift/project also exchange little-endian 64-bit limb vectors (poly2limbs / limbs2poly == the
//     mpz_export/mpz_import image of poly2mpz / mpz2poly, gmp.hpp:183-219);
//   * nfl::batch::* and nfl::device_batch<P> operate on contiguous arrays of polys (dense
//     [batch][NbModuli][Degree], as tests/tools.h:6-17 allocates) in ONE device pass;
//   * nfl::uniform(seed) is an addition: a seeded counter-based operand with the reference's
//     mask-then-subtract rule (core.hpp:165-176); plain nfl::uniform() draws fresh randomness like the reference.
// GMP: define NFL_HIP_WITH_GMP before inclusion (the <nfl.hpp> forwarding header does, as the reference always
// includes <gmpxx.h>: poly.hpp:33); NFL_HIP_REFERENCE_WORDS makes ZO_dist / hwt_dist store +1 as p + 1 like the
// reference's raw words (core.hpp:341,387) instead of the canonical 1.
#ifndef NFL_HIP_NFL_HPP
#define NFL_HIP_NFL_HPP

#include <algorithm>
#include <functional>
#include <array>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cinttypes>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <initializer_list>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <strings.h>
#include <thread>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>
#if defined(__linux__)
#include <sys/syscall.h>   // membarrier(2): the asymmetric barrier of detail::light_lock
#include <unistd.h>
#include <pthread.h>
#endif

#include "../nflhip.h"
#include "../nflhip_params.h"

#ifdef NFL_HIP_WITH_GMP
#include <gmp.h>
#if defined(__has_include)
#if __has_include(<gmpxx.h>)
#include <gmpxx.h>
#define NFL_HIP_HAVE_GMPXX 1
#endif
#endif
#endif

// The header is split into parts (round 6); this file is the umbrella: it owns the include guard, the standard includes and the
// order.  types -> samplers -> queue -> expr -> poly -> poly_p -> batch.
#include "types.hpp"
#include "samplers.hpp"
#include "queue.hpp"
#include "expr.hpp"
#include "poly.hpp"
#include "poly_p.hpp"
#include "batch.hpp"

#endif  // NFL_HIP_NFL_HPP
