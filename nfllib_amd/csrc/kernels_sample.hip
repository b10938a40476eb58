// kernels_sample.hip -- on-device samplers (the poly(uniform / non_uniform / ZO_dist / hwt_dist / gaussian)
// constructors of the reference, core.hpp:146-391).
//
// Randomness: the reference streams Salsa20 from a process-global key (lib/prng/fastrandombytes.cpp:17-37); here the
// stream is ChaCha20 (D. J. Bernstein's original layout: 64-bit block counter, 64-bit nonce) keyed per call, and it is
// COUNTER-BASED: 64-bit word w of stream (key, stream_id) is word (w mod 8) of block (w div 8).  Every coefficient reads
// a fixed word of its stream, so the output does not depend on the launch geometry or on how a batch is sharded.
// The map from random words to values is the reference's, statement for statement (cited per kernel); the tests feed
// the same words through a numpy restatement that is pinned against the real reference.
#include <atomic>
#include <vector>

#include "kernels.h"
#include <type_traits>
#include "modarith.h"

namespace nflhip {

struct ChaChaKey {
  uint32_t k[8];
  // DOMAIN SEPARATION: every distribution reads its own region of the keystream of (key, stream_id).  Bits 56..62 of
  // the 64-bit block counter carry the distribution's tag (bit 63 selects the Gaussian sampler's secondary stream), so
  // uniform / bounded / ZO / hamming-weight / Gaussian calls that share a key AND a stream id never consume the same
  // keystream word (a public uniform polynomial must not reveal the noise drawn next to it).  The low 56 bits count
  // blocks: 2^56 x 64 bytes per (key, stream_id, distribution).
  uint64_t dom;
  // SEQUENCE MODE (nflhip_sample_seq_dev): polynomial b of the batch is what a one-polynomial call with stream id
  // nonce + b * seq_stride would produce (word positions restart at 0 in every polynomial).  seq_on = 0: the batch is
  // one keystream read by global position (first_poly offsets it).
  uint32_t seq_on;
  uint64_t seq_stride;
};
// tags: NFLHIP_DIST_* + 1; 0 = the raw words of nflhip_random_words_dev, kDomGauss for both Gaussian entry points
static constexpr int kDomRaw = 0, kDomGauss = 5;
// NARROW DRAWS (round 5): the wide rules above spend one 64-bit keystream word per value whatever the limb width, so a u16 /
// u32 polynomial pays 4x / 2x the ChaCha20 rounds it needs, and a Gaussian sample -- decided by its first few bits in all
// but ~entries x 2^-32 of the cases -- as many as a uniform 62-bit residue.  Under the narrow rules a value reads a LANE of
// the keystream: residue word g of poly(uniform) the limb-width lane g (bytes [g w, (g + 1) w) of the stream, little
// endian), Gaussian coefficient g the 32-bit lane g.  They live in their OWN domains, so what the wide rules produce for a
// (key, stream id) -- and every digest recorded from them -- keeps its meaning:
//   6  poly(uniform), lanes of the limb width (NFLHIP_DIST_UNIFORM | NFLHIP_DIST_NARROW)
//   7  Gaussian, 32-bit draw: lane g of this stream is the TOP half of coefficient g's first word
//   8  ... and lane g of this one its lower half, read only when the top half ties with a table entry's
static constexpr int kDomUniformNarrow = 6, kDomGauss32 = 7, kDomGauss32Ref = 8;

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define NFLHIP_QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7)

// one 64-byte block as eight little-endian 64-bit words
__device__ __forceinline__ void chacha20_block(const ChaChaKey &key, uint64_t counter, uint64_t nonce, uint64_t out[8]) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                    key.k[4],    key.k[5],    key.k[6],    key.k[7],    (uint32_t)counter, (uint32_t)((counter | key.dom) >> 32),
                    (uint32_t)nonce, (uint32_t)(nonce >> 32)};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    NFLHIP_QR(x[0], x[4], x[8], x[12]);
    NFLHIP_QR(x[1], x[5], x[9], x[13]);
    NFLHIP_QR(x[2], x[6], x[10], x[14]);
    NFLHIP_QR(x[3], x[7], x[11], x[15]);
    NFLHIP_QR(x[0], x[5], x[10], x[15]);
    NFLHIP_QR(x[1], x[6], x[11], x[12]);
    NFLHIP_QR(x[2], x[7], x[8], x[13]);
    NFLHIP_QR(x[3], x[4], x[9], x[14]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (uint64_t)(x[2 * i] + s[2 * i]) | ((uint64_t)(x[2 * i + 1] + s[2 * i + 1]) << 32);
}
#undef NFLHIP_QR

// raw stream words [first_word, first_word + nwords)
__global__ void k_random_words(uint64_t *out, uint64_t first_word, size_t nwords, ChaChaKey key, uint64_t nonce) {
  const uint64_t fb = first_word >> 3, nb = ((first_word + nwords + 7) >> 3) - fb;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t w[8];
    chacha20_block(key, fb + b, nonce, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t g = ((fb + b) << 3) + j;
      if (g >= first_word && g < first_word + nwords) out[g - first_word] = w[j];
    }
  }
}

// v * amp mod p for a small signed value (|v| below the modulus in every sensible use; reduced anyway)
__device__ __noinline__ uint64_t residue_general(uint64_t mag, uint64_t amp, uint64_t p) {
  return (uint64_t)(((unsigned __int128)(mag % p) * (amp % p)) % p);
}
template <typename T> __device__ __forceinline__ T signed_residue(bool neg, uint64_t mag, uint64_t amp, uint64_t p) {
  const uint64_t prod = mag * amp;  // the sensible case (small noise, small amplifier): one multiply, no division
  const uint64_t m = (((mag | amp) >> 31) == 0 && prod < p) ? prod : residue_general(mag, amp, p);
  return (T)(neg ? (m ? p - m : 0) : m);
}

// ---- eight results per lane -> eight coalesced stores per wave -------------------------------------------------
// Every sampler thread turns one 64-byte keystream block into eight consecutive values; stored directly, each store
// instruction would scatter 64 words 64 bytes apart.  A wave-local LDS transpose hands lane L the values
// k*64 + L (k = 0..7) of the wave's 512, so each store instruction writes 64 consecutive words.
constexpr int kTS = 72;  // row stride: at most 2-way bank conflicts in both directions
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <typename V>
__device__ __forceinline__ void wave_transpose8(V *slab, int lane, bool active, const V (&in)[8], V (&out)[8]) {
  if (active) {
#pragma unroll
    for (int c = 0; c < 8; ++c) slab[c * kTS + lane] = in[c];
  }
  wave_sync_lds();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int t = k * 64 + lane;  // produced by lane t >> 3 as its value t & 7
    out[k] = slab[(t & 7) * kTS + (t >> 3)];
  }
  wave_sync_lds();
}

// ---- poly(uniform) (core.hpp:152-188): one stream word per residue word, mask to floor(log2 p)+1 bits, one
// conditional subtraction.  Word index = ((poly*nm + cm)*n + i).
template <typename T>
__global__ void k_sample_uniform(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, uint64_t first_word,
                                 size_t total, ChaChaKey key, uint64_t nonce) {
  const uint64_t fb = first_word >> 3, nb = ((first_word + total + 7) >> 3) - fb;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t w[8];
    chacha20_block(key, fb + b, nonce, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t g = ((fb + b) << 3) + j;
      if (g >= first_word && g < first_word + total) {
        const ModConst<T> c = mc[(int)((g >> logn) % (uint64_t)nm)];
        T v = (T)((T)w[j] & c.mask);
        if (v >= c.p) v = (T)(v - c.p);
        d[g - first_word] = v;
      }
    }
  }
}

// n >= 8: a keystream block never straddles a row, so the modulus is looked up once per block
template <typename T>
__global__ void __launch_bounds__(256) k_sample_uniform8(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm,
                                                         uint64_t first_word, size_t total, ChaChaKey key, uint64_t nonce) {
  __shared__ uint64_t xs[4][8 * kTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t fb = first_word >> 3;  // first_word and total are multiples of 8
  const size_t nblk = total >> 3, ntile = (nblk + 63) >> 6;
  for (size_t tile = (size_t)blockIdx.x * 4 + wv; tile < ntile; tile += (size_t)gridDim.x * 4) {
    const size_t b = (tile << 6) + lane;
    const bool active = b < nblk;
    uint64_t v[8], o[8];
    if (active) {
      uint64_t w[8];
      const uint64_t row = (fb + b) >> (logn - 3);
      const int cm = (row >> 32) == 0 ? (int)((uint32_t)row % (uint32_t)nm) : (int)(row % (uint64_t)nm);
      if (key.seq_on) {  // per-polynomial keystreams: block index inside the polynomial, stream id of the polynomial
        const uint64_t poly = row / (uint64_t)nm;
        chacha20_block(key, (fb + b) - ((poly * (uint64_t)nm) << (logn - 3)), nonce + poly * key.seq_stride, w);
      } else {
        chacha20_block(key, fb + b, nonce, w);
      }
      const ModConst<T> c = mc[cm];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        T x = (T)((T)w[j] & c.mask);
        if (x >= c.p) x = (T)(x - c.p);
        v[j] = x;
      }
    }
    wave_transpose8(xs[wv], lane, active, v, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t idx = (tile << 9) + (size_t)(k * 64 + lane);
      if (idx < total) d[idx] = (T)o[k];
    }
  }
}

// ---- poly(uniform), narrow draw: residue word g reads the limb-width lane g of the stream.  One keystream block per thread =
// L = 64 / sizeof(T) consecutive residue words = 64 contiguous bytes of the row, masked, reduced and stored by the thread
// itself.  Needs rows of at least L words (a block never straddles a row) and, in sequence mode, nothing else.
template <typename T> __device__ __forceinline__ uint64_t uniform_lanes(uint64_t x, T mask, T p) {
  uint64_t r = 0;
#pragma unroll
  for (int l = 0; l < (int)(8 / sizeof(T)); ++l) {
    T v = (T)((T)(x >> (8 * sizeof(T) * l)) & mask);
    if (v >= p) v = (T)(v - p);
    r |= (uint64_t)v << (8 * sizeof(T) * l);
  }
  return r;
}
template <typename T>
__global__ void __launch_bounds__(256) k_sample_uniform_narrow(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm,
                                                               uint64_t first_word, size_t total, ChaChaKey key, uint64_t nonce) {
  constexpr int LG = sizeof(T) == 2 ? 5 : sizeof(T) == 4 ? 4 : 3;   // log2 of the lanes per block
  const uint64_t fb = first_word >> LG;                             // first_word and total are multiples of the row length
  const size_t nblk = total >> LG;
  for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += (size_t)gridDim.x * blockDim.x) {
    uint64_t w[8];
    const uint64_t row = (fb + b) >> (logn - LG);
    const int cm = (row >> 32) == 0 ? (int)((uint32_t)row % (uint32_t)nm) : (int)(row % (uint64_t)nm);
    if (key.seq_on) {
      const uint64_t poly = row / (uint64_t)nm;
      chacha20_block(key, (fb + b) - ((poly * (uint64_t)nm) << (logn - LG)), nonce + poly * key.seq_stride, w);
    } else {
      chacha20_block(key, fb + b, nonce, w);
    }
    const ModConst<T> c = mc[cm];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = uniform_lanes<T>(w[j], c.mask, c.p);
    uint4 *dst = reinterpret_cast<uint4 *>(d + (b << LG));
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dst[q] = make_uint4((uint32_t)w[2 * q], (uint32_t)(w[2 * q] >> 32), (uint32_t)w[2 * q + 1], (uint32_t)(w[2 * q + 1] >> 32));
  }
}
// any row length, any alignment: the lane's modulus is looked up lane by lane (rows shorter than a keystream block)
template <typename T>
__global__ void k_sample_uniform_narrow_any(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, uint64_t first_word,
                                            size_t total, ChaChaKey key, uint64_t nonce) {
  constexpr int L = 64 / sizeof(T);
  const uint64_t per = key.seq_on ? ((uint64_t)nm << logn) : total;     // lanes of one stream (sequence mode: one polynomial)
  const uint64_t base = key.seq_on ? 0 : first_word;                    // ... and the lane its first value reads
  const uint64_t fb = base / L, bps = (base + per + L - 1) / L - fb;    // blocks per stream
  const uint64_t nstreams = key.seq_on ? total / per : 1;
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nstreams * bps; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t sidx = idx / bps, blk = fb + idx % bps;
    uint64_t w[8];
    chacha20_block(key, blk, nonce + (key.seq_on ? sidx * key.seq_stride : 0), w);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const uint64_t g = blk * L + j;
      if (g < base || g >= base + per) continue;
      const ModConst<T> c = mc[(int)((g >> logn) % (uint64_t)nm)];
      T v = (T)((T)(w[j / (L / 8)] >> (8 * sizeof(T) * (j % (L / 8)))) & c.mask);
      if (v >= c.p) v = (T)(v - c.p);
      d[sidx * per + (g - base)] = v;
    }
  }
}

// ---- one small signed integer per coefficient, replicated over the moduli.  Stream word index = poly*n + i.
//   dist 1  poly(non_uniform(ub[, amp]))  core.hpp:195-277: mask to floor(log2(2ub-1))+1 bits, one conditional
//           subtraction of 2ub-1, values >= ub are the negatives tmp - (2ub-1); times the amplifier.
//   dist 2  poly(ZO_dist(rho))            core.hpp:330-340: byte b = word & 0xff; b <= rho ? (b & 2 ? +1 : -1) : 0.
//           (The reference stores +1 as p+1; this engine stores the canonical 1 -- its own operators require < p,
//            ops.hpp:131,148.)
template <typename T>
__global__ void k_sample_small(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, uint64_t first_coef,
                               size_t ncoef, int dist, uint64_t p0, uint64_t p1, ChaChaKey key, uint64_t nonce) {
  const uint64_t fb = first_coef >> 3, nb = ((first_coef + ncoef + 7) >> 3) - fb;
  const uint64_t n = ((uint64_t)1) << logn;
  // NFLHIP_DIST_REFERENCE_WORDS: store +1 as the reference does, p + 1 (`pm + (rnd & 2)`, core.hpp:341)
  const bool refw = (dist & 0x100) != 0;
  dist &= 0xff;
  uint64_t mask = 0;
  if (dist == 1) {
    const uint64_t t = 2 * p0 - 1;  // >= 1
    int bits = 0;
    while (bits < 64 && (t >> bits) != 0) ++bits;  // floor(log2 t) + 1
    mask = bits >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << bits) - 1);
  }
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t w[8];
    chacha20_block(key, fb + b, nonce, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t g = ((fb + b) << 3) + j;
      if (g < first_coef || g >= first_coef + ncoef) continue;
      bool neg = false, zero = false;
      uint64_t mag = 0, amp = 1;
      if (dist == 1) {
        uint64_t tmp = w[j] & mask;
        if (tmp >= 2 * p0 - 1) tmp -= 2 * p0 - 1;
        neg = tmp >= p0;
        mag = neg ? (2 * p0 - 1) - tmp : tmp;
        amp = p1;
      } else {
        const unsigned byte = (unsigned)(w[j] & 0xff);
        zero = byte > (unsigned)p0;
        neg = (byte & 2u) == 0;
        mag = 1;
      }
      const uint64_t local = g - first_coef, poly = local >> logn, i = local & (n - 1);
      T *col = d + ((poly * (uint64_t)nm) << logn) + i;
      for (int cm = 0; cm < nm; ++cm)
        col[(uint64_t)cm << logn] = zero ? (T)0
                                    : (refw && dist == 2 && !neg) ? (T)(mc[cm].p + 1)
                                                                  : signed_residue<T>(neg, mag, amp, (uint64_t)mc[cm].p);
    }
  }
}

// n >= 8: one block per thread, results as signed 64-bit integers through the transpose
template <typename T>
__global__ void __launch_bounds__(256) k_sample_small8(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm,
                                                       uint64_t first_coef, size_t ncoef, int dist, uint64_t p0, uint64_t p1,
                                                       ChaChaKey key, uint64_t nonce) {
  __shared__ long long xs[4][8 * kTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t n = ((uint64_t)1) << logn, fb = first_coef >> 3;
  const bool refw = (dist & 0x100) != 0;  // see k_sample_small
  dist &= 0xff;
  uint64_t mask = 0;
  if (dist == 1) {
    const uint64_t t = 2 * p0 - 1;  // >= 1
    int bits = 0;
    while (bits < 64 && (t >> bits) != 0) ++bits;  // floor(log2 t) + 1
    mask = bits >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << bits) - 1);
  }
  const uint64_t amp = dist == 1 ? p1 : 1;
  const size_t nblk = ncoef >> 3, ntile = (nblk + 63) >> 6;
  for (size_t tile = (size_t)blockIdx.x * 4 + wv; tile < ntile; tile += (size_t)gridDim.x * 4) {
    const size_t b = (tile << 6) + lane;
    const bool active = b < nblk;
    long long v[8], o[8];
    if (active) {
      uint64_t w[8];
      if (key.seq_on) chacha20_block(key, (fb + b) & ((n >> 3) - 1), nonce + ((fb + b) >> (logn - 3)) * key.seq_stride, w);
      else chacha20_block(key, fb + b, nonce, w);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (dist == 1) {  // same statements as k_sample_small
          uint64_t tmp = w[j] & mask;
          if (tmp >= 2 * p0 - 1) tmp -= 2 * p0 - 1;
          v[j] = tmp >= p0 ? -(long long)((2 * p0 - 1) - tmp) : (long long)tmp;
        } else {
          const unsigned byte = (unsigned)(w[j] & 0xff);
          v[j] = byte > (unsigned)p0 ? 0 : ((byte & 2u) == 0 ? -1 : 1);
        }
      }
    }
    wave_transpose8(xs[wv], lane, active, v, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t idx = (tile << 9) + (size_t)(k * 64 + lane);
      if (idx < ncoef) {
        const bool neg = o[k] < 0;
        const uint64_t mag = (uint64_t)(neg ? -o[k] : o[k]);
        const uint64_t poly = idx >> logn, i = idx & (n - 1);
        T *col = d + ((poly * (uint64_t)nm) << logn) + i;
        for (int cm = 0; cm < nm; ++cm)
          col[(uint64_t)cm << logn] = mag == 0 ? (T)0
                                      : (refw && dist == 2 && !neg) ? (T)(mc[cm].p + 1)
                                                                    : signed_residue<T>(neg, mag, amp, (uint64_t)mc[cm].p);
      }
    }
  }
}

// ---- poly(gaussian(&fg, amp)) (core.hpp:284-322; FastGaussianNoise.hpp): inversion sampling from a cumulative table
// of W 64-bit words per entry (most significant first), entry k = floor(2^(64W) * P(X <= x_min + k)):
// x = x_min + #{k : cdt[k] <= r} for a uniform W-word number r.
// LAZY PRECISION: the most significant word of r is stream word g (g = global coefficient index poly*n + i); the lower
// W-1 words only matter when that word EQUALS the top word of a table entry met by the search (probability
// entries * 2^-64 per sample), and are then read from the secondary stream -- same key and nonce, block counters from
// 2^63 up: word (W-1)*g + k - 1 of it is word k of r.  The value is exactly the full-precision inversion of
// r = (word g, secondary words), at one keystream word per sample instead of W.
// `tie_shift` (0 in production; nflhip_debug_gauss_tie_shift for the tests) only widens what counts as a tie -- the first
// words are compared after dropping their low tie_shift bits -- so that the tie path, whose result is the same
// full-precision comparison, runs often enough to be tested.
static constexpr uint64_t kSecondaryCounter = ((uint64_t)1) << 63;

template <int W>
__device__ __noinline__ bool gauss_tie_less(const uint64_t r0, const uint64_t *e, uint64_t g, const ChaChaKey &key,
                                            uint64_t nonce) {
  if (r0 != e[0]) return r0 < e[0];
  uint64_t blk[8], have = ~(uint64_t)0;
  for (int k = 1; k < W; ++k) {
    const uint64_t wi = (uint64_t)(W - 1) * g + (uint64_t)(k - 1);
    if ((wi >> 3) != have) {
      have = wi >> 3;
      chacha20_block(key, kSecondaryCounter | have, nonce, blk);
    }
    uint64_t rk = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) rk = (int)(wi & 7) == j ? blk[j] : rk;  // (no dynamic register indexing)
    if (rk != e[k]) return rk < e[k];
  }
  return false;  // r == entry: not below it
}

// smallest k in [0, entries-1] with r < cdt[k] (cdt[entries-1] = all ones)
template <int W>
__device__ __forceinline__ int gauss_search(const uint64_t r0, uint64_t g, const uint64_t *__restrict__ cdt, int entries,
                                            int tie_shift, const ChaChaKey &key, uint64_t nonce) {
  int lo = 0, hi = entries - 1;
  const uint64_t rs = r0 >> tie_shift;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const uint64_t *e = cdt + (size_t)mid * W;
    const uint64_t es = e[0] >> tie_shift;
    bool less = rs < es;
    if (W > 1 || tie_shift) {
      if (rs == es) less = gauss_tie_less<W>(r0, e, g, key, nonce);
    }
    if (less) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Eight searches side by side over the table's FIRST words held in LDS (`top`, one word per entry; the launchers stage
// it when the table has at most kGaussLdsEntries entries): a fixed number of branch-free steps (iters = ceil(log2
// entries)), so the eight dependent chains interleave and no step waits for global memory.  A step whose first words tie
// (probability entries * 2^-64 per sample; under the tie_shift test hook: often) marks the sample, which is then redone by
// the exact search above -- the result is the full-precision inversion either way.
constexpr int kGaussLdsEntries = 4096;
template <int W>
__device__ __forceinline__ void gauss_search8(const uint64_t (&w)[8], uint64_t g0, const uint64_t *top,
                                              const uint64_t *__restrict__ cdt, int entries, int iters, int tie_shift,
                                              const ChaChaKey &key, uint64_t nc, int (&out)[8]) {
  int lo[8], hi[8];
  unsigned tied = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) lo[c] = 0, hi[c] = entries - 1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int mid = (lo[c] + hi[c]) >> 1;
      const uint64_t es = top[mid] >> tie_shift, rs = w[c] >> tie_shift;
      const bool open = lo[c] < hi[c], less = rs < es;
      tied |= (open && rs == es && (W > 1 || tie_shift)) ? (1u << c) : 0u;
      hi[c] = (open && less) ? mid : hi[c];
      lo[c] = (open && !less) ? mid + 1 : lo[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) out[c] = lo[c];
  if (tied) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (tied & (1u << c)) out[c] = gauss_search<W>(w[c], g0 + c, cdt, entries, tie_shift, key, nc);
  }
}
__device__ __forceinline__ void stage_gauss_top(uint64_t *top, const uint64_t *__restrict__ cdt, int entries, int W) {
  for (int k = threadIdx.x; k < entries; k += blockDim.x) top[k] = cdt[(size_t)k * W];
  __syncthreads();
}

// ---- the 32-bit draw (nflhip_gauss_set_draw_bits(g, 32)).  Coefficient g's uniform number is
//   ( lane g of stream 7 , lane g of stream 8 , secondary words as above )       (32 + 32 + 64 (W - 1) bits, most significant first)
// and the sample is its full-precision inversion, exactly as under the 64-bit draw.  The search runs on the 32-bit lane
// against the table's top halves; only when the lane EQUALS a top half it meets (probability ~entries x 2^-32 per sample)
// is the lower half fetched (one more keystream block) and the sample redone by the exact search.  Half the ChaCha20 rounds
// per sample, and 32-bit compares instead of 64-bit ones.
__device__ __forceinline__ uint32_t lane32_of(const uint64_t (&w)[8], unsigned j) {   // (no dynamic register indexing)
  uint64_t x = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) x = (int)(j >> 1) == q ? w[q] : x;
  return (uint32_t)(x >> (32 * (j & 1)));
}
__device__ __noinline__ uint64_t gauss32_first_word(uint32_t top, uint64_t g, const ChaChaKey &key, uint64_t nonce) {
  ChaChaKey kr = key;
  kr.dom = ((uint64_t)kDomGauss32Ref) << 56;
  uint64_t blk[8];
  chacha20_block(kr, g >> 4, nonce, blk);
  return ((uint64_t)top << 32) | (uint64_t)lane32_of(blk, (unsigned)(g & 15));
}
// the first word of coefficient g in full (both halves): the per-sample paths
__device__ __forceinline__ uint64_t gauss32_word(uint64_t g, const ChaChaKey &key, uint64_t nonce) {
  uint64_t blk[8];
  chacha20_block(key, g >> 4, nonce, blk);
  return gauss32_first_word(lane32_of(blk, (unsigned)(g & 15)), g, key, nonce);
}
// Sixteen searches side by side over the table's top halves in LDS (`top`, one 32-bit word per entry), each STARTED by a
// bucket table (`lut`, kGaussBuckets entries, built on the host with the table, gauss_bucket_table below): the lane's top 12
// bits name a bucket; lut[b] = (number of entries whose top half lies below the bucket) | min(entries inside it, 15) << 12.
// The answer is then known up to the entries inside the bucket -- none or one almost everywhere, so kGaussLutSteps = 2
// branch-free steps close every bucket of at most 3 entries.  A sample in a denser bucket (the far tails, where dozens of
// entries share their top 12 bits: ~2^-11 per sample) or one that ties with a top half is redone alone: the plain search over
// the whole table and, on a tie, the exact one with the lower half of its first word.  Same value either way.
constexpr int kGaussBucketBits = 12, kGaussBuckets = 1 << kGaussBucketBits, kGaussLutSteps = 2;
template <int W>
__device__ __noinline__ int gauss_search32_one(uint32_t r, uint64_t g, const uint32_t *top, const uint64_t *__restrict__ cdt, int entries,
                                               int tie_shift, const ChaChaKey &key, uint64_t nc) {
  const int ts = tie_shift > 31 ? 31 : tie_shift;
  int lo = 0, hi = entries - 1;
  bool tie = false;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const uint32_t es = top[mid] >> ts, rs = r >> ts;
    tie |= rs == es;
    if (rs < es) hi = mid; else lo = mid + 1;
  }
  return tie ? gauss_search<W>(gauss32_first_word(r, g, key, nc), g, cdt, entries, tie_shift, key, nc) : lo;
}
template <int W>
__device__ __forceinline__ void gauss_search16(const uint64_t (&w)[8], uint64_t g0, const uint32_t *top, const uint16_t *lut,
                                               const uint64_t *__restrict__ cdt, int entries, int tie_shift,
                                               const ChaChaKey &key, uint64_t nc, int (&out)[16]) {
  int lo[16], hi[16];
  uint32_t r[16];
  unsigned redo = 0;
  const int ts = tie_shift > 31 ? 31 : tie_shift;   // (the test hook widens the ties of this stage too)
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    r[c] = (uint32_t)(w[c >> 1] >> (32 * (c & 1)));
    const unsigned e = lut[r[c] >> (32 - kGaussBucketBits)];
    const int inside = (int)(e >> kGaussBucketBits);
    lo[c] = (int)(e & (kGaussBuckets - 1));
    hi[c] = min(lo[c] + inside, entries - 1);
    redo |= inside >= (1 << kGaussLutSteps) ? (1u << c) : 0u;
  }
#pragma unroll
  for (int it = 0; it < kGaussLutSteps; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int mid = (lo[c] + hi[c]) >> 1;
      const uint32_t es = top[mid] >> ts, rs = r[c] >> ts;
      const bool open = lo[c] < hi[c], less = rs < es;
      redo |= (open && rs == es) ? (1u << c) : 0u;    // the lower half always matters on a tie, one-word tables included
      hi[c] = (open && less) ? mid : hi[c];
      lo[c] = (open && !less) ? mid + 1 : lo[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) out[c] = lo[c];
  if (redo) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (redo & (1u << c)) out[c] = gauss_search32_one<W>(r[c], g0 + c, top, cdt, entries, tie_shift, key, nc);
  }
}
// LDS image of a narrow-draw kernel: top halves [entries, rounded up to a multiple of FOUR words], then the bucket table -- the
// table is copied with 16-byte LDS stores, so its offset and the dynamic-LDS base (alignas(16) below) keep it 16-byte aligned
__device__ __forceinline__ const uint16_t *stage_gauss_top32(uint32_t *top, const uint64_t *__restrict__ cdt, const uint16_t *__restrict__ lut_g,
                                                             int entries, int W) {
  uint16_t *lut = reinterpret_cast<uint16_t *>(top + ((entries + 3) & ~3));
  for (int k = threadIdx.x; k < entries; k += blockDim.x) top[k] = (uint32_t)(cdt[(size_t)k * W] >> 32);
  for (int k = threadIdx.x; k < kGaussBuckets / 8; k += blockDim.x)
    reinterpret_cast<uint4 *>(lut)[k] = reinterpret_cast<const uint4 *>(lut_g)[k];
  __syncthreads();
  return lut;
}
// sixteen consecutive coefficients of one keystream block -> their table indices (lut == nullptr: the table does not fit LDS)
template <int W>
__device__ __forceinline__ void gauss_block16(uint64_t g0, const uint32_t *top, const uint16_t *lut, const uint64_t *__restrict__ cdt,
                                              int entries, int tie_shift, const ChaChaKey &key, uint64_t nc, int (&r)[16]) {
  uint64_t w[8];
  chacha20_block(key, g0 >> 4, nc, w);
  if (lut) {
    gauss_search16<W>(w, g0, top, lut, cdt, entries, tie_shift, key, nc, r);
  } else {   // (tables beyond kGaussLdsEntries: the exact search per sample; the sixteen lower halves are ONE block of the second domain)
    ChaChaKey kr = key;
    kr.dom = ((uint64_t)kDomGauss32Ref) << 56;
    uint64_t lo[8];
    chacha20_block(kr, g0 >> 4, nc, lo);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const uint64_t r0 = ((uint64_t)(uint32_t)(w[c >> 1] >> (32 * (c & 1))) << 32) | (uint64_t)(uint32_t)(lo[c >> 1] >> (32 * (c & 1)));
      r[c] = gauss_search<W>(r0, g0 + c, cdt, entries, tie_shift, key, nc);
    }
  }
}

// compact polynomials, 32-bit draw, n >= 16: a thread's block = 16 consecutive coefficients = 16 / 32 / 64 contiguous bytes
template <typename S, int W>
__device__ __forceinline__ void gauss_small16_body(S *d, int logn, uint64_t first_coef, size_t ncoef, const uint64_t *__restrict__ cdt, int entries,
                                                   long long x_min, long long amp, const ChaChaKey &key, uint64_t nonce, int tie_shift,
                                                   const uint16_t *__restrict__ lut_g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t gtop32[];
  const uint16_t *lut = lut_g ? stage_gauss_top32(gtop32, cdt, lut_g, entries, W) : nullptr;
  const uint64_t n = ((uint64_t)1) << logn;
  const size_t ngroups = ncoef >> 4;  // first_coef and ncoef are multiples of 16 (n >= 16)
  for (size_t grp = (size_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups; grp += (size_t)gridDim.x * blockDim.x) {
    uint64_t g0 = first_coef + (grp << 4), nc = nonce;
    if (key.seq_on) {
      nc += (g0 >> logn) * key.seq_stride;
      g0 &= n - 1;
    }
    int r[16];
    gauss_block16<W>(g0, gtop32, lut, cdt, entries, tie_shift, key, nc, r);
    uint32_t v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = (uint32_t)(int32_t)(S)((x_min + r[c]) * amp);
    S *dst = d + (grp << 4);
    if (sizeof(S) == 1) {
      uint4 pk;
      pk.x = (v[0] & 0xff) | ((v[1] & 0xff) << 8) | ((v[2] & 0xff) << 16) | (v[3] << 24);
      pk.y = (v[4] & 0xff) | ((v[5] & 0xff) << 8) | ((v[6] & 0xff) << 16) | (v[7] << 24);
      pk.z = (v[8] & 0xff) | ((v[9] & 0xff) << 8) | ((v[10] & 0xff) << 16) | (v[11] << 24);
      pk.w = (v[12] & 0xff) | ((v[13] & 0xff) << 8) | ((v[14] & 0xff) << 16) | (v[15] << 24);
      *reinterpret_cast<uint4 *>(dst) = pk;
    } else if (sizeof(S) == 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        reinterpret_cast<uint4 *>(dst)[q] = make_uint4((v[8 * q] & 0xffff) | (v[8 * q + 1] << 16), (v[8 * q + 2] & 0xffff) | (v[8 * q + 3] << 16),
                                                       (v[8 * q + 4] & 0xffff) | (v[8 * q + 5] << 16), (v[8 * q + 6] & 0xffff) | (v[8 * q + 7] << 16));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) reinterpret_cast<uint4 *>(dst)[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  }
}
template <typename S, int W>
__global__ void __launch_bounds__(256) k_gauss_small16(S *d, int logn, uint64_t first_coef, size_t ncoef,
                                                       const uint64_t *__restrict__ cdt, int entries, long long x_min,
                                                       long long amp, ChaChaKey key, uint64_t nonce, int tie_shift,
                                                       const uint16_t *__restrict__ lut_g) {
  gauss_small16_body<S, W>(d, logn, first_coef, ncoef, cdt, entries, x_min, amp, key, nonce, tie_shift, lut_g);
}
// up to four compact draws of one table in ONE launch (blockIdx.y = the draw): an LWE encryption samples x, e0, e1 with their own
// amplifiers and stream ids, and three launches of a few hundred polynomials each leave a third of the wave slots empty in their
// last round (1 638 polynomials = 2.1 rounds of the 3 072 slots; three draws together = 6.4)
struct GaussMulti {
  void *d[4];
  long long amp[4];
  uint64_t nonce[4], stride[4];
};
template <typename S, int W>
__global__ void __launch_bounds__(256) k_gauss_small16_multi(GaussMulti m, int logn, size_t ncoef, const uint64_t *__restrict__ cdt,
                                                             int entries, long long x_min, ChaChaKey key, int tie_shift,
                                                             const uint16_t *__restrict__ lut_g) {
  const unsigned j = blockIdx.y;
  S *d = nullptr;
  long long amp = 0;
  uint64_t nonce = 0, stride = 0;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)   // (no dynamic indexing of the kernel argument)
    if (q == j) d = static_cast<S *>(m.d[q]), amp = m.amp[q], nonce = m.nonce[q], stride = m.stride[q];
  key.seq_stride = stride;
  gauss_small16_body<S, W>(d, logn, 0, ncoef, cdt, entries, x_min, amp, key, nonce, tie_shift, lut_g);
}

// residue words over the moduli, 32-bit draw, n >= 16: a wave's 1024 results through the wave-local LDS transpose
template <typename T, int W>
__global__ void __launch_bounds__(256) k_sample_gauss16(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm,
                                                        uint64_t first_coef, size_t ncoef, const uint64_t *__restrict__ cdt,
                                                        int entries, long long x_min, uint64_t amp, ChaChaKey key, uint64_t nonce,
                                                        int tie_shift, const uint16_t *__restrict__ lut_g) {
  constexpr int S = kTS;
  __shared__ int xs[4][16 * S];
  extern __shared__ __attribute__((aligned(16))) uint32_t gtop32[];
  const uint16_t *lut = lut_g ? stage_gauss_top32(gtop32, cdt, lut_g, entries, W) : nullptr;
  const uint64_t n = ((uint64_t)1) << logn;
  const size_t ngroups = ncoef >> 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t nwt = (ngroups + 63) >> 6;  // wave tiles of 64 groups = 1024 coefficients
  for (size_t tile = (size_t)blockIdx.x * 4 + wv; tile < nwt; tile += (size_t)gridDim.x * 4) {
    const size_t grp = (tile << 6) + lane;
    if (grp < ngroups) {
      uint64_t g0 = first_coef + (grp << 4), nc = nonce;
      if (key.seq_on) {
        nc += (g0 >> logn) * key.seq_stride;
        g0 &= n - 1;
      }
      int r[16];
      gauss_block16<W>(g0, gtop32, lut, cdt, entries, tie_shift, key, nc, r);
#pragma unroll
      for (int c = 0; c < 16; ++c) xs[wv][c * S + lane] = r[c];
    }
    wave_sync_lds();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = k * 64 + lane;               // coefficient of the tile: produced by lane t >> 4 as its result t & 15
      const size_t idx = (tile << 10) + (size_t)t;
      if (idx < ncoef) {
        const long long x = x_min + xs[wv][(t & 15) * S + (t >> 4)];
        const bool neg = x < 0;
        const uint64_t mag = (uint64_t)(neg ? -x : x);
        const uint64_t poly = idx >> logn, i = idx & (n - 1);
        T *col = d + ((poly * (uint64_t)nm) << logn) + i;
        for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = signed_residue<T>(neg, mag, amp, (uint64_t)mc[cm].p);
      }
    }
    wave_sync_lds();
  }
}

// the first word of coefficient g's uniform number, either draw (the per-sample paths)
__device__ __forceinline__ uint64_t gauss_word(uint64_t g, const ChaChaKey &key, uint64_t nonce, int narrow) {
  if (narrow) return gauss32_word(g, key, nonce);
  uint64_t blk[8];
  chacha20_block(key, g >> 3, nonce, blk);
  return blk[g & 7];
}

template <typename T, int W>
__global__ void k_sample_gauss(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, uint64_t first_coef,
                               size_t ncoef, const uint64_t *__restrict__ cdt, int entries, long long x_min, uint64_t amp,
                               ChaChaKey key, uint64_t nonce, int tie_shift, int narrow) {
  const uint64_t n = ((uint64_t)1) << logn;
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ncoef; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t g = first_coef + idx;
    const long long x = x_min + gauss_search<W>(gauss_word(g, key, nonce, narrow), g, cdt, entries, tie_shift, key, nonce);
    const bool neg = x < 0;
    const uint64_t mag = (uint64_t)(neg ? -x : x);
    const uint64_t poly = idx >> logn, i = idx & (n - 1);
    T *col = d + ((poly * (uint64_t)nm) << logn) + i;
    for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = signed_residue<T>(neg, mag, amp, (uint64_t)mc[cm].p);
  }
}

// the same map for n >= 8, one keystream block (eight consecutive coefficients) per thread; a wave's 512 results go
// through a wave-local LDS transpose so that every store instruction writes 64 consecutive words of a row
template <typename T, int W>
__global__ void __launch_bounds__(256) k_sample_gauss8(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm,
                                                       uint64_t first_coef, size_t ncoef,
                                                       const uint64_t *__restrict__ cdt, int entries, long long x_min,
                                                       uint64_t amp, ChaChaKey key, uint64_t nonce, int tie_shift, int iters) {
  constexpr int S = kTS;
  __shared__ int xs[4][8 * S];
  extern __shared__ uint64_t gtop[];   // the table's first words (iters > 0)
  if (iters) stage_gauss_top(gtop, cdt, entries, W);
  const uint64_t n = ((uint64_t)1) << logn;
  const size_t ngroups = ncoef >> 3;  // first_coef and ncoef are multiples of 8 (n >= 8)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t nwt = (ngroups + 63) >> 6;  // wave tiles of 64 groups = 512 coefficients
  for (size_t tile = (size_t)blockIdx.x * 4 + wv; tile < nwt; tile += (size_t)gridDim.x * 4) {
    const size_t grp = (tile << 6) + lane;
    if (grp < ngroups) {
      uint64_t g0 = first_coef + (grp << 3), nc = nonce;
      if (key.seq_on) {  // coefficient index inside the polynomial, stream id of the polynomial
        nc += (g0 >> logn) * key.seq_stride;
        g0 &= n - 1;
      }
      uint64_t w[8];
      chacha20_block(key, g0 >> 3, nc, w);
      if (iters) {
        int r[8];
        gauss_search8<W>(w, g0, gtop, cdt, entries, iters, tie_shift, key, nc, r);
#pragma unroll
        for (int c = 0; c < 8; ++c) xs[wv][c * S + lane] = r[c];
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          xs[wv][c * S + lane] = gauss_search<W>(w[c], g0 + c, cdt, entries, tie_shift, key, nc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int t = k * 64 + lane;               // coefficient of the tile: produced by lane t >> 3 as its result t & 7
      const size_t idx = (tile << 9) + (size_t)t;
      if (idx < ncoef) {
        const long long x = x_min + xs[wv][(t & 7) * S + (t >> 3)];
        const bool neg = x < 0;
        const uint64_t mag = (uint64_t)(neg ? -x : x);
        const uint64_t poly = idx >> logn, i = idx & (n - 1);
        T *col = d + ((poly * (uint64_t)nm) << logn) + i;
        for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = signed_residue<T>(neg, mag, amp, (uint64_t)mc[cm].p);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- compact Gaussian polynomials: ONE signed integer per coefficient (x * amplifier) instead of nm residue words.
// Coefficient g gets exactly the integer k_sample_gauss / k_sample_gauss8 spread over the moduli from the same
// (key, stream id), so expanding the compact form (v < 0 ? p + v : v, k_expand_small below or the transform-fused
// kernels' own prologue) reproduces nflhip_sample_gauss_dev's words bit for bit.  S = int8_t / int16_t / int32_t; the
// launcher checks that every possible sample fits.
template <typename S, int W>
__global__ void __launch_bounds__(256) k_gauss_small8(S *d, int logn, uint64_t first_coef, size_t ncoef,
                                                      const uint64_t *__restrict__ cdt, int entries, long long x_min,
                                                      long long amp, ChaChaKey key, uint64_t nonce, int tie_shift, int iters) {
  // a thread's keystream block = eight CONSECUTIVE coefficients = 8 / 16 / 32 contiguous bytes of the compact row: packed
  // and stored by the thread itself (a wave store covers 512 consecutive coefficients; no transpose)
  extern __shared__ uint64_t gtop[];
  if (iters) stage_gauss_top(gtop, cdt, entries, W);
  const uint64_t n = ((uint64_t)1) << logn;
  const size_t ngroups = ncoef >> 3;  // first_coef and ncoef are multiples of 8 (n >= 8)
  for (size_t grp = (size_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups; grp += (size_t)gridDim.x * blockDim.x) {
    uint64_t g0 = first_coef + (grp << 3), nc = nonce;
    if (key.seq_on) {
      nc += (g0 >> logn) * key.seq_stride;
      g0 &= n - 1;
    }
    uint64_t w[8];
    chacha20_block(key, g0 >> 3, nc, w);
    int r[8];
    if (iters) {
      gauss_search8<W>(w, g0, gtop, cdt, entries, iters, tie_shift, key, nc, r);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) r[c] = gauss_search<W>(w[c], g0 + c, cdt, entries, tie_shift, key, nc);
    }
    S v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (S)((x_min + r[c]) * amp);
    S *dst = d + (grp << 3);
    if (sizeof(S) == 1) {
      uint64_t pk = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) pk |= (uint64_t)(uint8_t)v[c] << (8 * c);
      *reinterpret_cast<uint64_t *>(dst) = pk;
    } else if (sizeof(S) == 2) {
      uint4 pk;
      pk.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
      pk.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
      pk.z = (uint32_t)(uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16);
      pk.w = (uint32_t)(uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
      *reinterpret_cast<uint4 *>(dst) = pk;
    } else {
      uint4 p0, p1;
      p0.x = (uint32_t)v[0]; p0.y = (uint32_t)v[1]; p0.z = (uint32_t)v[2]; p0.w = (uint32_t)v[3];
      p1.x = (uint32_t)v[4]; p1.y = (uint32_t)v[5]; p1.z = (uint32_t)v[6]; p1.w = (uint32_t)v[7];
      reinterpret_cast<uint4 *>(dst)[0] = p0;
      reinterpret_cast<uint4 *>(dst)[1] = p1;
    }
  }
}

template <typename S, int W>
__global__ void k_gauss_small(S *d, uint64_t first_coef, size_t ncoef, const uint64_t *__restrict__ cdt, int entries,
                              long long x_min, long long amp, ChaChaKey key, uint64_t nonce, int tie_shift, int narrow) {
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ncoef; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t g = first_coef + idx;
    d[idx] = (S)((x_min + gauss_search<W>(gauss_word(g, key, nonce, narrow), g, cdt, entries, tie_shift, key, nonce)) * amp);
  }
}

// compact -> residue words: dst[b][cm][i] = v < 0 ? p_cm + v : v with v = src[b * stride][i] (|v| < every modulus:
// checked by the producer); stride 0 = one compact polynomial for the whole batch.  S = T selects plain word rows
// (dst[b] = src[b * stride], a strided gather).
template <typename T, typename S>
__global__ void k_expand_small(T *dst, const S *src, const ModConst<T> *__restrict__ mc, int logn, int nm, size_t ncoef,
                               unsigned stride) {
  const uint64_t n = ((uint64_t)1) << logn;
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ncoef; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t poly = idx >> logn, i = idx & (n - 1);
    T *col = dst + ((poly * (uint64_t)nm) << logn) + i;
    if (sizeof(S) == sizeof(T) && !std::is_signed<S>::value) {
      const S *scol = src + ((poly * stride * (uint64_t)nm) << logn) + i;
      for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = (T)scol[(uint64_t)cm << logn];
    } else {
      const long long v = (long long)src[((poly * stride) << logn) + i];
      for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = (T)(v < 0 ? (long long)mc[cm].p + v : v);
    }
  }
}

// FastGaussianNoise::getNoise (FastGaussianNoise.hpp:477-595): raw signed samples; sample j of the call is the value
// coefficient first_sample + j of a polynomial batch would get from the same (key, stream_id)
template <int W>
__global__ void k_gauss_noise(long long *out, uint64_t first_sample, size_t count, const uint64_t *__restrict__ cdt,
                              int entries, long long x_min, ChaChaKey key, uint64_t nonce, int tie_shift, int narrow) {
  if (narrow) {   // 32-bit draw: one sample per thread (this entry point is a test / inspection path, not a hot one)
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
      const uint64_t g = first_sample + j;
      out[j] = x_min + gauss_search<W>(gauss32_word(g, key, nonce), g, cdt, entries, tie_shift, key, nonce);
    }
    return;
  }
  const uint64_t fb = first_sample >> 3, nb = ((first_sample + count + 7) >> 3) - fb;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t w[8];
    chacha20_block(key, fb + b, nonce, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t g = ((fb + b) << 3) + j;
      if (g >= first_sample && g < first_sample + count)
        out[g - first_sample] = x_min + gauss_search<W>(w[j], g, cdt, entries, tie_shift, key, nonce);
    }
  }
}

// ---- poly(hwt_dist(h)) (core.hpp:347-391): exactly h coefficients are +-1, uniformly among the C(n,h) supports.
// The reference draws them by reservoir sampling with rejection-sampled indices; here Floyd's algorithm (the same
// distribution, h draws instead of n) runs one thread per polynomial over the zero-initialised row 0 as the
// membership set (marks 1 = +1, 2 = -1), then a spread pass writes every modulus row.
// Sub-stream of polynomial P: words [P*4n, P*4n + 2n) for the index draws (rejections are rarer than 2^-40),
// word P*4n + 2n + t for the sign of draw t (bit 1, as the reference's `& 2`).
template <typename T>
__global__ void k_hwt_select(T *d, int logn, int nm, uint64_t first_poly, size_t batch, uint64_t h, ChaChaKey key,
                             uint64_t nonce) {
  const size_t P = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (P >= batch) return;
  const uint64_t n = ((uint64_t)1) << logn, base = key.seq_on ? 0 : (first_poly + P) * 4 * n;
  if (key.seq_on) nonce += P * key.seq_stride;
  T *row0 = d + ((P * (uint64_t)nm) << logn);
  uint64_t blk[8], have = ~(uint64_t)0, ctr = 0;
  auto word = [&](uint64_t wi) {
    if ((wi >> 3) != have) {
      have = wi >> 3;
      chacha20_block(key, have, nonce, blk);
    }
    return blk[wi & 7];
  };
  for (uint64_t t = 0; t < h; ++t) {
    const uint64_t j = n - h + t, bound = j + 1;  // draw uniformly from [0, j]
    const uint64_t lim = (~(uint64_t)0 / bound) * bound;
    uint64_t pos;
    for (;;) {
      pos = word(base + (ctr < 2 * n ? ctr : 2 * n - 1));
      ++ctr;
      if (pos <= lim - 1 || ctr >= 2 * n) break;  // accept pos < lim
    }
    pos %= bound;
    if (row0[pos] != 0) pos = j;
    const uint64_t sgn = word(base + 2 * n + t);
    row0[pos] = (T)((sgn & 2) ? 1 : 2);
  }
}
template <typename T>
__global__ void k_hwt_spread(T *d, const ModConst<T> *__restrict__ mc, int logn, int nm, size_t ncoef, int refw) {
  const uint64_t n = ((uint64_t)1) << logn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ncoef; idx += (size_t)gridDim.x * blockDim.x) {
    const uint64_t poly = idx >> logn, i = idx & (n - 1);
    T *col = d + ((poly * (uint64_t)nm) << logn) + i;
    const T m = col[0];
    if (m == 0) continue;
    // (reference words: `pm + (rnd & 2)`, core.hpp:387 -- +1 is stored as p + 1)
    for (int cm = 0; cm < nm; ++cm) col[(uint64_t)cm << logn] = m == 1 ? (refw ? (T)(mc[cm].p + 1) : (T)1) : (T)(mc[cm].p - 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
static inline ChaChaKey load_key(const unsigned char *key32, int domain, int seq_on = 0, uint64_t seq_stride = 0) {
  ChaChaKey k;
  k.dom = ((uint64_t)domain) << 56;
  k.seq_on = (uint32_t)seq_on;
  k.seq_stride = seq_stride;
  for (int i = 0; i < 8; ++i)
    k.k[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) |
             ((uint32_t)key32[4 * i + 3] << 24);
  return k;
}
static inline unsigned grid_for(size_t items) {
  size_t b = (items + 255) / 256;
  return (unsigned)(b > 256 * 64 ? 256 * 64 : (b ? b : 1));
}

hipError_t launch_random_words(uint64_t *out, uint64_t first_word, size_t nwords, const unsigned char *key32,
                               uint64_t stream_id, hipStream_t st) {
  if (nwords == 0) return hipSuccess;
  hipLaunchKernelGGL(k_random_words, dim3(grid_for(nwords / 8 + 2)), dim3(256), 0, st, out, first_word, nwords,
                     load_key(key32, kDomRaw), stream_id);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_sample(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch, int dist, uint64_t p0,
                         uint64_t p1, const unsigned char *key32, uint64_t stream_id, hipStream_t st, int seq_on,
                         uint64_t seq_stride) {
  if (batch == 0) return hipSuccess;
  if (seq_on && (s.n < 8 || first_poly != 0)) return hipErrorNotSupported;  // (keystream blocks must not straddle polynomials)
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  const int refw = dist & 0x100;   // NFLHIP_DIST_REFERENCE_WORDS
  const int narrow = dist & 0x200; // NFLHIP_DIST_NARROW
  dist &= 0xff;
  if (narrow && dist != 0) return hipErrorInvalidValue;
  const ChaChaKey key = load_key(key32, narrow ? kDomUniformNarrow : dist + 1, seq_on, seq_stride);
  const size_t ncoef = batch * s.n, total = ncoef * s.nm;
  if (narrow && sizeof(T) < 8) {   // (64-bit limbs: a lane IS a stream word -- the kernels below, in the narrow domain)
    if (s.n >= 64 / sizeof(T))
      hipLaunchKernelGGL((k_sample_uniform_narrow<T>), dim3(grid_for(total * sizeof(T) / 64)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm,
                         (uint64_t)first_poly * s.nm * s.n, total, key, stream_id);
    else
      hipLaunchKernelGGL((k_sample_uniform_narrow_any<T>), dim3(grid_for(total * sizeof(T) / 64 + batch + 2)), dim3(256), 0, st, d, mc, s.logn,
                         (int)s.nm, (uint64_t)first_poly * s.nm * s.n, total, key, stream_id);
    return hipGetLastError();
  }
  switch (dist) {
    case 0:
      if (s.n >= 8)
        hipLaunchKernelGGL((k_sample_uniform8<T>), dim3(grid_for(total / 8)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm,
                           (uint64_t)first_poly * s.nm * s.n, total, key, stream_id);
      else
        hipLaunchKernelGGL((k_sample_uniform<T>), dim3(grid_for(total / 8 + 2)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm,
                           (uint64_t)first_poly * s.nm * s.n, total, key, stream_id);
      break;
    case 1:
    case 2:
      if (s.n >= 8)
        hipLaunchKernelGGL((k_sample_small8<T>), dim3(grid_for(ncoef / 8)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm,
                           (uint64_t)first_poly * s.n, ncoef, dist | refw, p0, p1, key, stream_id);
      else
        hipLaunchKernelGGL((k_sample_small<T>), dim3(grid_for(ncoef / 8 + 2)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm,
                           (uint64_t)first_poly * s.n, ncoef, dist | refw, p0, p1, key, stream_id);
      break;
    case 3: {
      hipError_t e = hipMemsetAsync(d, 0, total * sizeof(T), st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((k_hwt_select<T>), dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, st, d, s.logn, (int)s.nm,
                         (uint64_t)first_poly, batch, p0, key, stream_id);
      hipLaunchKernelGGL((k_hwt_spread<T>), dim3(grid_for(ncoef)), dim3(256), 0, st, d, mc, s.logn, (int)s.nm, ncoef,
                         refw);
      break;
    }
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// test hook of gauss_search (include/nflhip_debug.h nflhip_debug_gauss_tie_shift): never set in production
static std::atomic<int> g_gauss_tie_shift{0};
void set_gauss_tie_shift(int shift) { g_gauss_tie_shift.store(shift < 0 ? 0 : (shift > 63 ? 63 : shift)); }
static int gauss_tie_shift() { return g_gauss_tie_shift.load(std::memory_order_relaxed); }
// the bucket table of the narrow draw's search (gauss_search16), from the table's first words; empty when the table does not fit LDS
std::vector<uint16_t> gauss_bucket_table(const uint64_t *cdt, int words, size_t entries) {
  std::vector<uint16_t> lut;
  if (entries < 2 || entries > (size_t)kGaussLdsEntries) return lut;
  lut.resize(kGaussBuckets);
  size_t k = 0;
  for (unsigned b = 0; b < (unsigned)kGaussBuckets; ++b) {
    while (k < entries && (cdt[k * words] >> (64 - kGaussBucketBits)) < b) ++k;       // entries whose top half lies below the bucket
    size_t in = 0;
    while (k + in < entries && (cdt[(k + in) * words] >> (64 - kGaussBucketBits)) == b) ++in;
    lut[b] = (uint16_t)((k > (size_t)kGaussBuckets - 1 ? (size_t)kGaussBuckets - 1 : k) | ((in > 15 ? 15 : in) << kGaussBucketBits));
  }
  return lut;
}
static size_t gauss_narrow_lds(int entries) { return (size_t)((entries + 3) & ~3) * 4 + (size_t)kGaussBuckets * 2; }
// steps of the LDS-resident search (gauss_search8), 0 = the table is too long for LDS: the per-sample search over global memory
static int gauss_lds_iters(int entries) {
  if (entries < 2 || entries > kGaussLdsEntries) return 0;
  int it = 0;
  while ((1 << it) < entries) ++it;
  return it;
}

hipError_t launch_gauss_noise(long long *out, uint64_t first_sample, size_t count, const uint64_t *cdt, int words,
                              int entries, long long x_min, const unsigned char *key32, uint64_t stream_id, hipStream_t st,
                              int narrow) {
  if (count == 0) return hipSuccess;
  const ChaChaKey key = load_key(key32, narrow ? kDomGauss32 : kDomGauss);
  const dim3 g(grid_for(narrow ? count : count / 8 + 2)), b(256);
  const int ts = gauss_tie_shift();
  switch (words) {
    case 1: hipLaunchKernelGGL((k_gauss_noise<1>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    case 2: hipLaunchKernelGGL((k_gauss_noise<2>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    case 3: hipLaunchKernelGGL((k_gauss_noise<3>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    case 4: hipLaunchKernelGGL((k_gauss_noise<4>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    case 5: hipLaunchKernelGGL((k_gauss_noise<5>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    case 6: hipLaunchKernelGGL((k_gauss_noise<6>), g, b, 0, st, out, first_sample, count, cdt, entries, x_min, key, stream_id, ts, narrow); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_sample_gauss(const Shape &s, const DevTables &t, T *d, size_t first_poly, size_t batch,
                               const uint64_t *cdt, int words, int entries, long long x_min, uint64_t amp,
                               const unsigned char *key32, uint64_t stream_id, hipStream_t st, int seq_on,
                               uint64_t seq_stride, int narrow, const uint16_t *lut) {
  if (batch == 0) return hipSuccess;
  if (seq_on && (s.n < (narrow ? 16u : 8u) || first_poly != 0)) return hipErrorNotSupported;
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  const ChaChaKey key = load_key(key32, narrow ? kDomGauss32 : kDomGauss, seq_on, seq_stride);
  const size_t ncoef = batch * s.n;
  const uint64_t fc = (uint64_t)first_poly * s.n;
  const int tie_shift = gauss_tie_shift();
  if (narrow && s.n >= 16) {  // sixteen coefficients per thread
    const dim3 g(grid_for(ncoef / 16)), b(256);
    const size_t lds = lut ? gauss_narrow_lds(entries) : 0;
    switch (words) {
      case 1: hipLaunchKernelGGL((k_sample_gauss16<T, 1>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      case 2: hipLaunchKernelGGL((k_sample_gauss16<T, 2>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      case 3: hipLaunchKernelGGL((k_sample_gauss16<T, 3>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      case 4: hipLaunchKernelGGL((k_sample_gauss16<T, 4>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      case 5: hipLaunchKernelGGL((k_sample_gauss16<T, 5>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      case 6: hipLaunchKernelGGL((k_sample_gauss16<T, 6>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, lut); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (!narrow && s.n >= 8) {  // eight coefficients per thread
    const dim3 g(grid_for(ncoef / 8)), b(256);
    const int iters = gauss_lds_iters(entries);
    const size_t lds = iters ? (size_t)entries * 8 : 0;
    switch (words) {
      case 1: hipLaunchKernelGGL((k_sample_gauss8<T, 1>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      case 2: hipLaunchKernelGGL((k_sample_gauss8<T, 2>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      case 3: hipLaunchKernelGGL((k_sample_gauss8<T, 3>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      case 4: hipLaunchKernelGGL((k_sample_gauss8<T, 4>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      case 5: hipLaunchKernelGGL((k_sample_gauss8<T, 5>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      case 6: hipLaunchKernelGGL((k_sample_gauss8<T, 6>), g, b, lds, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, iters); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const dim3 g(grid_for(ncoef)), b(256);
  switch (words) {
    case 1: hipLaunchKernelGGL((k_sample_gauss<T, 1>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    case 2: hipLaunchKernelGGL((k_sample_gauss<T, 2>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    case 3: hipLaunchKernelGGL((k_sample_gauss<T, 3>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    case 4: hipLaunchKernelGGL((k_sample_gauss<T, 4>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    case 5: hipLaunchKernelGGL((k_sample_gauss<T, 5>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    case 6: hipLaunchKernelGGL((k_sample_gauss<T, 6>), g, b, 0, st, d, mc, s.logn, (int)s.nm, fc, ncoef, cdt, entries, x_min, amp, key, stream_id, tie_shift, narrow); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// format: 1 int8 | 2 int16 | 3 int32 (NFLHIP_FMT_*)
hipError_t launch_gauss_small(const Shape &s, void *d, int format, size_t first_poly, size_t batch, const uint64_t *cdt,
                              int words, int entries, long long x_min, uint64_t amp, const unsigned char *key32,
                              uint64_t stream_id, hipStream_t st, int seq_on, uint64_t seq_stride, int narrow, const uint16_t *lut) {
  if (batch == 0) return hipSuccess;
  if (seq_on && (s.n < (narrow ? 16u : 8u) || first_poly != 0)) return hipErrorNotSupported;
  if (words < 1 || words > 6 || format < 1 || format > 3) return hipErrorInvalidValue;
  const ChaChaKey key = load_key(key32, narrow ? kDomGauss32 : kDomGauss, seq_on, seq_stride);
  const size_t ncoef = batch * s.n;
  const uint64_t fc = (uint64_t)first_poly * s.n;
  const int tie_shift = gauss_tie_shift();
  const long long a = (long long)amp;
  const int iters = gauss_lds_iters(entries);
  const size_t lds = narrow ? (lut ? gauss_narrow_lds(entries) : 0) : (iters ? (size_t)entries * 8 : 0);
#define NFLHIP_GS(S, W)                                                                                                              \
  do {                                                                                                                               \
    if (narrow && s.n >= 16)                                                                                                         \
      hipLaunchKernelGGL((k_gauss_small16<S, W>), dim3(grid_for(ncoef / 16)), dim3(256), lds, st, (S *)d, s.logn, fc, ncoef, cdt,  \
                         entries, x_min, a, key, stream_id, tie_shift, lut);                                                         \
    else if (!narrow && s.n >= 8)                                                                                                    \
      hipLaunchKernelGGL((k_gauss_small8<S, W>), dim3(grid_for(ncoef / 8)), dim3(256), lds, st, (S *)d, s.logn, fc, ncoef, cdt,    \
                         entries, x_min, a, key, stream_id, tie_shift, iters);                                                       \
    else                                                                                                                             \
      hipLaunchKernelGGL((k_gauss_small<S, W>), dim3(grid_for(ncoef)), dim3(256), 0, st, (S *)d, fc, ncoef, cdt, entries, x_min, a, \
                         key, stream_id, tie_shift, narrow);                                                                         \
  } while (0)
#define NFLHIP_GSW(S)                    \
  switch (words) {                       \
    case 1: NFLHIP_GS(S, 1); break;      \
    case 2: NFLHIP_GS(S, 2); break;      \
    case 3: NFLHIP_GS(S, 3); break;      \
    case 4: NFLHIP_GS(S, 4); break;      \
    case 5: NFLHIP_GS(S, 5); break;      \
    default: NFLHIP_GS(S, 6); break;     \
  }
  if (format == 1) { NFLHIP_GSW(int8_t) } else if (format == 2) { NFLHIP_GSW(int16_t) } else { NFLHIP_GSW(int32_t) }
#undef NFLHIP_GSW
#undef NFLHIP_GS
  return hipGetLastError();
}

// `count` compact draws of one table: one launch where the sixteen-per-thread kernel applies (32-bit draw, n >= 16), else one per draw
hipError_t launch_gauss_small_multi(const Shape &s, void *const *d, size_t count, int format, size_t batch, const uint64_t *cdt, int words,
                                    int entries, long long x_min, const uint64_t *amp, const unsigned char *key32, const uint64_t *stream_id,
                                    const uint64_t *seq_stride, hipStream_t st, int narrow, const uint16_t *lut) {
  if (batch == 0 || count == 0) return hipSuccess;
  if (count > 4 || words < 1 || words > 6 || format < 1 || format > 3) return hipErrorInvalidValue;
  const int seq_on = seq_stride != nullptr;
  if (!(narrow && s.n >= 16) || count == 1) {
    for (size_t j = 0; j < count; ++j) {
      hipError_t e = launch_gauss_small(s, d[j], format, 0, batch, cdt, words, entries, x_min, amp[j], key32, stream_id[j], st, seq_on,
                                        seq_on ? seq_stride[j] : 0, narrow, lut);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const ChaChaKey key = load_key(key32, kDomGauss32, seq_on, 0);
  const size_t ncoef = batch * s.n;
  GaussMulti m = {};
  for (size_t j = 0; j < count; ++j) {
    m.d[j] = d[j];
    m.amp[j] = (long long)amp[j];
    m.nonce[j] = stream_id[j];
    m.stride[j] = seq_on ? seq_stride[j] : 0;
  }
  const int tie_shift = gauss_tie_shift();
  const size_t lds = lut ? gauss_narrow_lds(entries) : 0;
  const dim3 g(grid_for(ncoef / 16), (unsigned)count), b(256);
#define NFLHIP_GM(S, W) hipLaunchKernelGGL((k_gauss_small16_multi<S, W>), g, b, lds, st, m, s.logn, ncoef, cdt, entries, x_min, key, tie_shift, lut)
#define NFLHIP_GMW(S)                \
  switch (words) {                   \
    case 1: NFLHIP_GM(S, 1); break;  \
    case 2: NFLHIP_GM(S, 2); break;  \
    case 3: NFLHIP_GM(S, 3); break;  \
    case 4: NFLHIP_GM(S, 4); break;  \
    case 5: NFLHIP_GM(S, 5); break;  \
    default: NFLHIP_GM(S, 6); break; \
  }
  if (format == 1) { NFLHIP_GMW(int8_t) } else if (format == 2) { NFLHIP_GMW(int16_t) } else { NFLHIP_GMW(int32_t) }
#undef NFLHIP_GMW
#undef NFLHIP_GM
  return hipGetLastError();
}

template <typename T>
hipError_t launch_expand_small(const Shape &s, const DevTables &t, T *dst, const void *src, int format, unsigned stride,
                               size_t batch, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  const ModConst<T> *mc = (const ModConst<T> *)t.mc;
  const size_t ncoef = batch * s.n;
  const dim3 g(grid_for(ncoef)), b(256);
  switch (format) {
    case 0: hipLaunchKernelGGL((k_expand_small<T, T>), g, b, 0, st, dst, (const T *)src, mc, s.logn, (int)s.nm, ncoef, stride); break;
    case 1: hipLaunchKernelGGL((k_expand_small<T, int8_t>), g, b, 0, st, dst, (const int8_t *)src, mc, s.logn, (int)s.nm, ncoef, stride); break;
    case 2: hipLaunchKernelGGL((k_expand_small<T, int16_t>), g, b, 0, st, dst, (const int16_t *)src, mc, s.logn, (int)s.nm, ncoef, stride); break;
    case 3: hipLaunchKernelGGL((k_expand_small<T, int32_t>), g, b, 0, st, dst, (const int32_t *)src, mc, s.logn, (int)s.nm, ncoef, stride); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#define NFLHIP_INST(T)                                                                                                   \
  template hipError_t launch_sample<T>(const Shape &, const DevTables &, T *, size_t, size_t, int, uint64_t, uint64_t,   \
                                       const unsigned char *, uint64_t, hipStream_t, int, uint64_t);                     \
  template hipError_t launch_sample_gauss<T>(const Shape &, const DevTables &, T *, size_t, size_t, const uint64_t *, int, \
                                             int, long long, uint64_t, const unsigned char *, uint64_t, hipStream_t, int,  \
                                             uint64_t, int, const uint16_t *);                                             \
  template hipError_t launch_expand_small<T>(const Shape &, const DevTables &, T *, const void *, int, unsigned, size_t, hipStream_t);
NFLHIP_INST(uint16_t)
NFLHIP_INST(uint32_t)
NFLHIP_INST(uint64_t)
#undef NFLHIP_INST

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_sample() {}
hipError_t warm_sample(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_sample, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
