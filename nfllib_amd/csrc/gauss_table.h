// gauss_table.h -- cumulative table of a discrete Gaussian (host side; see gauss_table.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nflhip {

struct GaussTable {
  int words = 0;               // 64-bit words per entry, most significant first
  size_t entries = 0;          // 2*ceil(tail*sigma) + 1 support points
  long long x_min = 0;         // value of entry 0
  double tail = 0;             // tail bound (in sigmas)
  unsigned bit_precision = 0;  // the reference's bit_precision for these parameters
  std::vector<uint64_t> cdt;   // [entries][words]  floor(2^(64 words) * P(X <= x_min + k)), last entry all ones
};

// returns 0 on success, else fills *err
int build_gauss_table(double sigma, unsigned security, unsigned samples, double center, GaussTable *out, std::string *err);

}  // namespace nflhip
