// api.hip -- the C ABI of include/nflhip.h: context + device-table provisioning
// and the host/device entry points.  No CPU compute path exists here: every
// operation is a launch of a gfx950 kernel (kernels_generic.hip / kernels_fast.hip).
#include "../../include/nflhip.h"
#include "../../include/nflhip_debug.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gauss_table.h"
#include "kernels.h"

using namespace nflhip;

typedef unsigned __int128 u128;

// errno-style: one message per calling thread, so concurrent callers of one context never race on it
static thread_local std::string g_last_error = "";

struct nflhip_ctx {
  int device = 0;
  Shape shape{};
  DevTables tabs{};
  size_t word = 8;  // bytes per limb
  // host-pointer path: staging buffers + private stream, serialised by a mutex
  std::mutex mu;
  hipStream_t hstream = nullptr;
  void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes[4] = {0, 0, 0, 0};
  // a staging buffer of up to kStageHostMax bytes is PINNED HOST memory the kernels read and write directly (one polynomial per call, the
  // nfl::poly surface: a 128 KiB operand crosses PCIe inside the kernel in less time than a copy engine needs to start); larger ones are device
  // memory filled by copies
  bool stage_host[4] = {false, false, false, false};
  bool flag_host = false;          // tabs.flag is pinned host memory
  // large host-pointer calls: a three-slot pipeline of pinned staging chunks (HostPipe below), created on first use
  struct HostPipe *pipe = nullptr;
  // scratch for the composed (non-fused) polymul path, per stream use is serialised by the caller
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  std::mutex scratch_mu;
  // large-row polymul pipeline: two helper streams so that the HBM-bound streaming passes of one
  // chunk overlap the VALU-bound fused kernel of another; ev_prev orders successive calls on the scratch
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_start = nullptr, ev_done[2] = {nullptr, nullptr};
  bool ev_prev_valid = false;
  hipEvent_t ev_scratch = nullptr;  // end of the last single-stream pipeline that used the scratch
  bool ev_scratch_valid = false;
  // any_eq / any_neq: every call owns one result slot (device int) for its memset + kernel + readback, so host
  // threads comparing on distinct streams never share a flag
  static constexpr int kCmpSlots = 32;
  std::mutex cmp_mu[kCmpSlots];
  int cmp_token[kCmpSlots] = {};   // the token of the slot's latest comparison (under its mutex)
  std::atomic<unsigned> cmp_next{0};
  // host copies for introspection
  std::vector<uint64_t> h_Q;                     // moduli_product limbs
  std::vector<std::vector<uint64_t>> h_lifting;  // lifting_integers[cm]
  std::vector<uint64_t> h_P;
  std::vector<uint64_t> h_roots, h_invk, h_phi;  // params<T>::primitive_roots / invkMaxPolyDegree, phi = 2n-th root per modulus
  int kmax_log2 = 0;
  // core::ntt(x, wtab, winvtab, p) (core.hpp:455-532) on the device: one single-modulus child context per
  // (modulus, table set) whose twiddle table is the CYCLIC one, created on first use -- see nflhip_ntt_row_dev
  int cyclic = 0;  // 0: negacyclic tables (the normal context); 1 / 2: cyclic over omega / omega^-1 (child contexts)
  std::mutex row_mu;
  std::vector<nflhip_ctx *> row_ctx;  // [2 * cm + inverse_tables]
};

namespace nflhip {
int set_error(int code, const std::string &msg) {  // for the library's other translation units (comm.hip)
  g_last_error = msg;
  return code;
}
}  // namespace nflhip

static void pipe_destroy(nflhip_ctx *ctx);  // (HostPipe is defined with the host-pointer entry points)

static int fail(const nflhip_ctx *ctx, int code, const std::string &msg) {
  (void)ctx;
  g_last_error = msg;
  return code;
}
static int hipfail(const nflhip_ctx *ctx, hipError_t e, const char *where) {
  return fail(ctx, e == hipErrorNoDevice || e == hipErrorInvalidDevice ? NFLHIP_ERR_NO_DEVICE : NFLHIP_ERR_HIP,
              std::string(where) + ": " + hipGetErrorString(e));
}
#define HIPCHK(ctx, call)                                   \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return hipfail(ctx, _e, #call);   \
  } while (0)

// ---------------------------------------------------------------------------
// host-side modular helpers used only to BUILD tables (once per context)
// ---------------------------------------------------------------------------
static inline uint64_t mulmod_h(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((u128)a * b % p); }
static uint64_t powmod_h(uint64_t a, uint64_t e, uint64_t p) {
  uint64_t r = 1 % p;
  a %= p;
  while (e) {
    if (e & 1) r = mulmod_h(r, a, p);
    a = mulmod_h(a, a, p);
    e >>= 1;
  }
  return r;
}
static inline uint64_t shoup_h(uint64_t w, uint64_t p, int wb) { return (uint64_t)((((u128)w) << wb) / p); }
static unsigned bitrev_h(unsigned k, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) {
    r = (r << 1) | (k & 1u);
    k >>= 1;
  }
  return r;
}

// little-endian multi-limb helpers for the CRT constants (gmp.hpp:113-155)
typedef std::vector<uint64_t> Big;
static void big_trim(Big &a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
static Big big_mul_u64(const Big &a, uint64_t w) {
  Big r(a.size() + 1, 0);
  u128 c = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    c += (u128)a[i] * w;
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  r[a.size()] = (uint64_t)c;
  big_trim(r);
  return r;
}
static uint64_t big_divrem_u64(const Big &a, uint64_t d, Big *q) {
  Big out(a.size(), 0);
  u128 r = 0;
  for (size_t k = a.size(); k-- > 0;) {
    r = (r << 64) | a[k];
    out[k] = (uint64_t)(r / d);
    r %= d;
  }
  big_trim(out);
  if (q) *q = out;
  return (uint64_t)r;
}
static size_t big_bits(const Big &a) {
  if (a.empty()) return 0;
  size_t b = 0;
  uint64_t t = a.back();
  while (t) { ++b; t >>= 1; }
  return (a.size() - 1) * 64 + b;
}
static Big big_shl(const Big &a, int k, size_t limbs) {
  Big r(limbs, 0);
  for (size_t i = 0; i < a.size() && i < limbs; ++i) {
    r[i] |= a[i] << k;
    if (k && i + 1 < limbs) r[i + 1] |= a[i] >> (64 - k);
  }
  return r;
}

// fewest moduli for which the lift runs as an int8 GEMM on the matrix cores (kernels_crt_mfma.hip).  Its cost hardly depends on
// the modulus count (the tile is always 32 modulus slots x 256 columns: 0.48 ms at 12 moduli, 0.75 ms at 30 for 4 Mi
// coefficients), the VALU kernels of kernels_crt.hip grow with its square (0.24 ms at 12, 0.56 at 20, 0.76 at 24, 1.19 at 30):
// they cross between 20 and 21.  The projection's VALU kernels take a second launch beyond 16 residues: 0.40 ms at 16, 0.61 at 18
// against the GEMM's 0.40 / 0.43 -- it takes over at 17 (same-box sweeps in profiles/r04_crt_mfma.txt)
#ifndef NFLHIP_CRT_MFMA_MIN_NM
#define NFLHIP_CRT_MFMA_MIN_NM 21
#endif
#ifndef NFLHIP_CRT_MFMA_PROJ_MIN_NM
#define NFLHIP_CRT_MFMA_PROJ_MIN_NM 17
#endif

template <typename T>
static int build_tables(nflhip_ctx *c, const void *Pv, const void *rootsv, const void *invkv, int kmax_log2) {
  const T *P = (const T *)Pv, *roots = (const T *)rootsv, *invk = (const T *)invkv;
  const Shape &s = c->shape;
  const int wb = s.limb_bits, logn = s.logn;
  const size_t n = s.n, nm = s.nm;
  // moduli sanity: the engine relies on p being 2 bits below the word (params.hpp:27-28,61-62,104-105)
  for (size_t cm = 0; cm < nm; ++cm) {
    const uint64_t p = P[cm];
    if (p < 3 || (p >> (wb - 2)) != 0 || (p >> (wb - 3)) == 0)
      return fail(nullptr, NFLHIP_ERR_INVALID, "modulus is not (word-2) bits long");
    c->h_P.push_back(p);
    c->h_roots.push_back(roots[cm]);
    c->h_invk.push_back(invk[cm]);
    if (((((uint64_t)1) << (wb - 2)) - p) >> 32) c->shape.small_delta = 0;
    if (c->shape.small_delta) c->shape.nm_small = (int)cm + 1;
  }
  c->kmax_log2 = kmax_log2;
  // CRT constants
  Big Q(1, 1);
  for (size_t cm = 0; cm < nm; ++cm) Q = big_mul_u64(Q, P[cm]);
  const size_t bitsQ = big_bits(Q);
  c->shape.crt_L = (bitsQ + 63) / 64;
  c->shape.crt_Lacc = c->shape.crt_L + 1;
  const size_t Lacc = c->shape.crt_Lacc;
  c->h_Q = Q;
  c->h_Q.resize(c->shape.crt_L, 0);
  c->shape.crt_Q0 = Q.empty() ? 0 : Q[0];
  const size_t kStride = 36;  // fixed, zero-padded row stride of the device CRT tables
  const bool crt_ok = Lacc <= kStride;  // beyond that crt_lift reports NFLHIP_ERR_UNSUPPORTED, the transforms still work
  std::vector<uint64_t> qhat(nm * kStride, 0), qsh(6 * kStride, 0);
  std::vector<uint64_t> yinv(nm, 0);
  c->h_lifting.resize(nm);
  for (size_t cm = 0; cm < nm; ++cm) {
    Big quot;
    big_divrem_u64(Q, P[cm], &quot);                       // Q / p_cm            (mpz_divexact)
    const uint64_t qmod = big_divrem_u64(quot, P[cm], nullptr);
    yinv[cm] = powmod_h(qmod, P[cm] - 2, P[cm]);           // (Q/p_cm)^-1 mod p_cm (mpz_invert)
    for (size_t k = 0; crt_ok && k < quot.size(); ++k) qhat[cm * kStride + k] = quot[k];
    c->h_lifting[cm] = big_mul_u64(quot, yinv[cm]);        // lifting_integers[cm] (gmp.hpp:149-150)
  }
  for (int k = 0; crt_ok && k < 6; ++k) {
    Big sh = big_shl(Q, k, Lacc);
    for (size_t i = 0; i < Lacc; ++i) qsh[k * kStride + i] = sh[i];
  }
  // beyond the register-resident lift kernels (nm > 32 or more limbs than their tables hold): limb-serial tables
  std::vector<uint64_t> qhat_w, qsh_w;
  int Lw = 0, nsh = 0;
  if (!crt_ok || nm > 32) {
    Lw = (int)c->shape.crt_L + 2;
    while ((((size_t)1) << nsh) <= nm) ++nsh;  // 2^nsh > nm >= S / Q
    qhat_w.assign(nm * (size_t)Lw, 0);
    qsh_w.assign((size_t)nsh * Lw, 0);
    for (size_t cm = 0; cm < nm; ++cm) {
      Big quot;
      big_divrem_u64(Q, P[cm], &quot);
      for (size_t k = 0; k < quot.size(); ++k) qhat_w[cm * Lw + k] = quot[k];
    }
    for (int k = 0; k < nsh; ++k) {
      Big sh = big_shl(Q, k, (size_t)Lw);
      for (int i = 0; i < Lw; ++i) qsh_w[(size_t)k * Lw + i] = sh[i];
    }
  }
  // carry-free multiply-accumulate tables for 64-bit limbs (kernels_crt.hip)
  std::vector<uint32_t> qparts, bparts;
  int proj_K = 0;
  if (wb == 64 && crt_ok && nm <= 32) {
    const size_t L = c->shape.crt_L, S32 = 2 * kStride;
    qparts.assign(nm * 3 * S32, 0);
    for (size_t cm = 0; cm < nm; ++cm) {
      Big quot;
      big_divrem_u64(Q, P[cm], &quot);
      for (int j = 0; j < 3; ++j) {
        const Big sh = big_shl(quot, 21 * j, L);  // (Q/p) << 42 < Q: fits L words
        for (size_t i = 0; i < L; ++i) {
          qparts[(cm * 3 + j) * S32 + 2 * i] = (uint32_t)sh[i];
          qparts[(cm * 3 + j) * S32 + 2 * i + 1] = (uint32_t)(sh[i] >> 32);
        }
      }
    }
    proj_K = (int)(2 * L + 2);
    const size_t nmS = (nm + 3) & ~(size_t)3;  // row stride: zero padded, the kernel runs without guards
    bparts.assign((size_t)proj_K * 2 * 3 * nmS, 0);
    for (size_t cm = 0; cm < nm; ++cm) {
      const uint64_t p = P[cm], two32 = (((uint64_t)1) << 32) % p;
      uint64_t cur = 1 % p;  // 2^(32 t) mod p, t = 2k + half
      for (size_t t = 0; t < (size_t)proj_K * 2; ++t) {
        uint32_t *e = &bparts[t * 3 * nmS + cm];
        e[0] = (uint32_t)(cur & 0x1fffff);
        e[nmS] = (uint32_t)((cur >> 21) & 0x1fffff);
        e[2 * nmS] = (uint32_t)(cur >> 42);
        cur = mulmod_h(cur, two32, p);
      }
    }
  }

  // the lift as an int8 GEMM (kernels_crt_mfma.hip): balanced base-256 digits of Q/p_cm, laid out as the B operand of
  // v_mfma_i32_32x32x32_i8 -- K-step s, N-tile t, lane (column j = lane & 31, half h = lane >> 5), byte e:
  // modulus slot cm = 4 s + 2 h + (e >> 3), digit of y a = e & 7, column k = 8 j + t  ->  digit k - a of Q/p_cm.
  // Slot 31 is the quotient row: the digits of Q itself against the single digit "-floor(S / Q)" (a = 0).
  std::vector<int8_t> bfrag;
  if (wb == 64 && crt_ok && nm >= NFLHIP_CRT_MFMA_MIN_NM && nm <= 32 && c->shape.crt_L >= 4 && c->shape.crt_L <= 31) {
    const size_t ND = 264;
    std::vector<int8_t> dig(32 * ND, 0);
    for (size_t cm = 0; cm < 32; ++cm) {
      Big quot;
      if (cm < nm) big_divrem_u64(Q, P[cm], &quot);
      else if (cm == 31) quot = Q;   // (with 32 moduli the slot is the 32nd modulus's and the kernel subtracts the quotient itself)
      else continue;
      int carry = 0;
      for (size_t k = 0; k < ND; ++k) {
        const size_t w = k / 8;
        int v = (w < quot.size() ? (int)((quot[w] >> (8 * (k % 8))) & 0xff) : 0) + carry;
        carry = 0;
        if (v >= 128) { v -= 256; carry = 1; }
        dig[cm * ND + k] = (int8_t)v;
      }
    }
    bfrag.assign((size_t)8 * 8 * 64 * 16, 0);
    for (int st = 0; st < 8; ++st)
      for (int t = 0; t < 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 16; ++e) {
            const int cm = 4 * st + 2 * (lane >> 5) + (e >> 3), a = e & 7, k = 8 * (lane & 31) + t;
            if (k >= a && ((size_t)cm < nm || a == 0)) bfrag[(((size_t)st * 8 + t) * 64 + lane) * 16 + e] = dig[cm * ND + (k - a)];
          }
  }

  // the projection as an int8 GEMM (kernels_crt_mfma.hip): balanced base-256 digits of 256^k mod p_cm, k < 256, as the B
  // operand -- K-step s, N-tile t = digit, lane (cm = lane & 31, half h = lane >> 5), byte e: k = 32 s + 16 h + e -- and the
  // per-residue constant 2^18 p + 128 sum_k (256^k mod p) (the input bytes enter as a - 128; the sum is made non-negative)
  std::vector<int8_t> bproj;
  std::vector<uint64_t> coff, c2048;
  if (wb == 64 && c->shape.small_delta && nm >= NFLHIP_CRT_MFMA_PROJ_MIN_NM && nm <= 32) {
    std::vector<int8_t> dig((size_t)32 * 256 * 8, 0);   // [cm][k][digit]
    coff.assign(32 * 2, 0);
    c2048.assign(32 * 2, 0);
    for (size_t cm = 0; cm < nm; ++cm) {
      const uint64_t p = P[cm];
      uint64_t cur = 1 % p;
      __int128 colsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < 256; ++k) {
        int carry = 0;
        for (int b = 0; b < 8; ++b) {
          int v = (int)((cur >> (8 * b)) & 0xff) + carry;
          carry = 0;
          if (v >= 128 && b < 7) { v -= 256; carry = 1; }   // (the top digit of a 62-bit value is below 65: no carry out)
          dig[(cm * 256 + k) * 8 + b] = (int8_t)v;
          colsum[b] += 128 * v;
        }
        cur = mulmod_h(cur, 256 % p, p);
      }
      __int128 off = (__int128)p << 18;
      for (int b = 0; b < 8; ++b) off += colsum[b] * ((__int128)1 << (8 * b));
      coff[2 * cm] = (uint64_t)off;
      coff[2 * cm + 1] = (uint64_t)((unsigned __int128)off >> 64);
      c2048[2 * cm] = powmod_h(2 % p, 2048, p);
      c2048[2 * cm + 1] = shoup_h(c2048[2 * cm], p, 64);
    }
    bproj.assign((size_t)8 * 8 * 64 * 16, 0);
    for (int st = 0; st < 8; ++st)
      for (int t = 0; t < 8; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 16; ++e) {
            const size_t cm = lane & 31;
            const int k = 32 * st + 16 * (lane >> 5) + e;
            if (cm < nm) bproj[(((size_t)st * 8 + t) * 64 + lane) * 16 + e] = dig[(cm * 256 + k) * 8 + t];
          }
  }

  // twiddles + per-modulus constants
  std::vector<Tw<T>> psi(nm * n);
  std::vector<ModConst<T>> mc(nm);
  for (size_t cm = 0; cm < nm; ++cm) {
    const uint64_t p = P[cm];
    // phi: primitive 2n-th root from the primitive 2*kMax-th root (core.hpp:640-645)
    uint64_t phi = roots[cm];
    for (int i = 0; i < kmax_log2 - logn; ++i) phi = mulmod_h(phi, phi, p);
    if (powmod_h(phi, n, p) != p - 1) return fail(nullptr, NFLHIP_ERR_INVALID, "primitive root has the wrong order");
    c->h_phi.push_back(phi);
    const uint64_t base = c->cyclic == 2 ? powmod_h(phi, 2 * n - 1, p) : phi;  // phi^-1 for the inverse tables
    std::vector<uint64_t> pw(n);
    uint64_t cur = 1;
    for (size_t i = 0; i < n; ++i) {
      pw[i] = cur;
      cur = mulmod_h(cur, base, p);
    }
    Tw<T> *tw = psi.data() + cm * n;
    for (size_t k = 0; k < n; ++k) {
      // negacyclic: psi_br[k] = phi^bitrev(k).  With k = m + j (m = the power of two <= k: stage with m blocks, block j)
      // that exponent is (n/2m)(2 brev_m(j) + 1); the cyclic transform core::ntt computes needs omega^((n/2m) brev_m(j))
      // = phi^(bitrev(k) - n/2m) in the same slot, so every forward kernel runs it unchanged on this table.
      size_t e = bitrev_h((unsigned)k, logn);
      if (c->cyclic && k > 0) {
        size_t m = 1;
        while (2 * m <= k) m *= 2;
        e -= n / (2 * m);
      }
      const uint64_t w = pw[e];
      tw[k].w = (T)w;
      tw[k].wp = (T)shoup_h(w, p, wb);
    }
    ModConst<T> &m = mc[cm];
    m.p = (T)p;
    m.p2 = (T)(2 * p);
    m.mu = (T)((((u128)1) << (2 * wb - 4)) / p);
    // n^-1 = kMax^-1 * (kMax/n) (core.hpp:664-665)
    const uint64_t ninv = mulmod_h(invk[cm], (((uint64_t)1) << kmax_log2) / n, p);
    if (mulmod_h(ninv, n % p, p) != 1 % p) return fail(nullptr, NFLHIP_ERR_INVALID, "invkMaxPolyDegree is not the inverse");
    m.ninv = (T)ninv;
    m.ninv_sh = (T)shoup_h(ninv, p, wb);
    const uint64_t w1 = n >= 2 ? (uint64_t)tw[1].w : 1;
    const uint64_t w1n = mulmod_h(w1, ninv, p);
    m.w1ninv = (T)w1n;
    m.w1ninv_sh = (T)shoup_h(w1n, p, wb);
    const uint64_t beta = (uint64_t)((((u128)1) << 64) % p);
    m.beta = (T)beta;
    m.beta_sh = (T)shoup_h(beta, p, wb);
    m.yinv = (T)yinv[cm];
    m.yinv_sh = (T)shoup_h(yinv[cm], p, wb);
    int bits = 0;
    while (bits < wb && (((u128)1) << bits) <= (u128)p) ++bits;
    m.mask = (T)(bits >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << bits) - 1));
    m.delta = (T)((((uint64_t)1) << (wb - 2)) - p);
    m.mu2 = (T)((((u128)1) << (2 * wb - 3)) / p);
  }

  HIPCHK(nullptr, hipMalloc(&c->tabs.psi, psi.size() * sizeof(Tw<T>)));
  HIPCHK(nullptr, hipMemcpy(c->tabs.psi, psi.data(), psi.size() * sizeof(Tw<T>), hipMemcpyHostToDevice));
  c->tabs.psi_lm = nullptr;
  if (sizeof(T) == 8 && n >= 4096) {
    // the generated 64-bit kernels read the last four stages (indices n/16 .. n-1) LANE-MAJOR: stage logn-4+s transposed
    // from [(u << s) + g] to [g (n/16) + u], so that the 64 lanes of a wave fetch consecutive records
    // (tools/gen_polymul_asm.py tw_base_lm); indices below n/16 are shared with the natural table
    std::vector<Tw<T>> lm(psi);
    const size_t m = n >> 4;
    for (size_t cm = 0; cm < nm; ++cm)
      for (int s = 0; s < 4; ++s) {
        const Tw<T> *src = psi.data() + cm * n + (m << s);
        Tw<T> *dst = lm.data() + cm * n + (m << s);
        for (size_t u = 0; u < m; ++u)
          for (size_t g = 0; g < ((size_t)1 << s); ++g) dst[g * m + u] = src[(u << s) + g];
      }
    HIPCHK(nullptr, hipMalloc(&c->tabs.psi_lm, lm.size() * sizeof(Tw<T>)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.psi_lm, lm.data(), lm.size() * sizeof(Tw<T>), hipMemcpyHostToDevice));
  }
  HIPCHK(nullptr, hipMalloc(&c->tabs.mc, mc.size() * sizeof(ModConst<T>)));
  HIPCHK(nullptr, hipMemcpy(c->tabs.mc, mc.data(), mc.size() * sizeof(ModConst<T>), hipMemcpyHostToDevice));
  c->tabs.mc_inc[0] = c->tabs.mc_inc[1] = nullptr;
  if (sizeof(T) == 4 && n >= 1024 && n <= 4096 && !c->cyclic) {
    // 32-bit limbs, rows of 1024 / 2048 / 4096 words: the product on incomplete transforms (tools/gen_row1024_u32_asm.py base_mul,
    // level 2 only) reads (n / 4)^-1 in the n^-1 fields and floor(2^62 / p) - 2^32 in the mu field
    std::vector<ModConst<T>> mi(mc);
    for (size_t cm = 0; cm < nm; ++cm) {
      const uint64_t p = P[cm];
      const uint64_t ng = mulmod_h((uint64_t)mc[cm].ninv, 4, p), wg = mulmod_h((uint64_t)mc[cm].w1ninv, 4, p);
      mi[cm].ninv = (T)ng;
      mi[cm].ninv_sh = (T)shoup_h(ng, p, wb);
      mi[cm].w1ninv = (T)wg;
      mi[cm].w1ninv_sh = (T)shoup_h(wg, p, wb);
      mi[cm].mu = (T)(uint64_t)(((((u128)1) << 62) / p) - (((u128)1) << 32));
    }
    HIPCHK(nullptr, hipMalloc(&c->tabs.mc_inc[1], mi.size() * sizeof(ModConst<T>)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.mc_inc[1], mi.data(), mi.size() * sizeof(ModConst<T>), hipMemcpyHostToDevice));
  }
  if (sizeof(T) == 8 && n >= 1024 && !c->cyclic && (c->shape.small_delta || (n == 4096 && c->shape.nm_small > 0))) {
    // the metric product on incomplete transforms (nflhip_polymul4096i{1,2}_asm): the inverse undoes 12 - level stages, so the
    // scale folded into its last stage is (n / 2^level)^-1; the base multiplication reduces sums below 2^127 with
    // floor(2^127 / p) = 2^65 + m, m < 2^35 (delta < 2^32), handed over in the mu2 field
    for (int level = 1; level <= 2; ++level) {
      std::vector<ModConst<T>> mi(mc);
      for (size_t cm = 0; cm < nm; ++cm) {
        const uint64_t p = P[cm];
        const uint64_t ng = mulmod_h((uint64_t)mc[cm].ninv, (uint64_t)1 << level, p);
        const uint64_t wg = mulmod_h((uint64_t)mc[cm].w1ninv, (uint64_t)1 << level, p);
        mi[cm].ninv = (T)ng;
        mi[cm].ninv_sh = (T)shoup_h(ng, p, wb);
        mi[cm].w1ninv = (T)wg;
        mi[cm].w1ninv_sh = (T)shoup_h(wg, p, wb);
        mi[cm].mu2 = (T)(uint64_t)(((((u128)1) << 127) / p) - (((u128)1) << 65));
      }
      HIPCHK(nullptr, hipMalloc(&c->tabs.mc_inc[level - 1], mi.size() * sizeof(ModConst<T>)));
      HIPCHK(nullptr, hipMemcpy(c->tabs.mc_inc[level - 1], mi.data(), mi.size() * sizeof(ModConst<T>), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(nullptr, hipMalloc((void **)&c->tabs.qhat, qhat.size() * sizeof(uint64_t)));
  HIPCHK(nullptr, hipMemcpy(c->tabs.qhat, qhat.data(), qhat.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIPCHK(nullptr, hipMalloc((void **)&c->tabs.qsh, qsh.size() * sizeof(uint64_t)));
  HIPCHK(nullptr, hipMemcpy(c->tabs.qsh, qsh.data(), qsh.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  c->tabs.qparts = nullptr;
  c->tabs.bparts = nullptr;
  c->tabs.proj_K = 0;
  if (!qparts.empty()) {
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.qparts, qparts.size() * sizeof(uint32_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.qparts, qparts.data(), qparts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.bparts, bparts.size() * sizeof(uint32_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.bparts, bparts.data(), bparts.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->tabs.proj_K = proj_K;
    // Q / 2^(32 max(2L - 3, 0)) from its top words (the kernel divides the top five 32-bit digits of the sum by it)
    const size_t L = c->shape.crt_L;
    long double qt = 0.0L;
    for (size_t k = L; k-- > 0;) qt = qt * 18446744073709551616.0L + (long double)Q[k];
    for (long w = 0; w < 2 * (long)L - 3; ++w) qt /= 4294967296.0L;
    c->tabs.inv_qtop = (double)(1.0L / qt);
  }
  c->tabs.crt_bfrag = nullptr;
  c->tabs.crt_bproj = nullptr;
  c->tabs.crt_coff = nullptr;
  c->tabs.crt_c2048 = nullptr;
  if (!bproj.empty()) {
    HIPCHK(nullptr, hipMalloc(&c->tabs.crt_bproj, bproj.size()));
    HIPCHK(nullptr, hipMemcpy(c->tabs.crt_bproj, bproj.data(), bproj.size(), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.crt_coff, coff.size() * sizeof(uint64_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.crt_coff, coff.data(), coff.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.crt_c2048, c2048.size() * sizeof(uint64_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.crt_c2048, c2048.data(), c2048.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  if (!bfrag.empty()) {
    HIPCHK(nullptr, hipMalloc(&c->tabs.crt_bfrag, bfrag.size()));
    HIPCHK(nullptr, hipMemcpy(c->tabs.crt_bfrag, bfrag.data(), bfrag.size(), hipMemcpyHostToDevice));
  }
  c->tabs.qhat_w = nullptr;
  c->tabs.qsh_w = nullptr;
  c->tabs.crt_Lw = Lw;
  c->tabs.crt_nsh = nsh;
  if (!qhat_w.empty()) {
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.qhat_w, qhat_w.size() * sizeof(uint64_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.qhat_w, qhat_w.data(), qhat_w.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.qsh_w, qsh_w.size() * sizeof(uint64_t)));
    HIPCHK(nullptr, hipMemcpy(c->tabs.qsh_w, qsh_w.data(), qsh_w.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  // the comparison flags: pinned host memory the kernels store into and the caller reads once the stream has drained (device memory +
  // a copy when the pinned allocation is refused); a hit stores the call's TOKEN, so nothing has to be cleared in front of a launch
  if (hipHostMalloc((void **)&c->tabs.flag, nflhip_ctx::kCmpSlots * sizeof(int), hipHostMallocDefault) == hipSuccess) {
    c->flag_host = true;
    std::memset(c->tabs.flag, 0, nflhip_ctx::kCmpSlots * sizeof(int));
  } else {
    (void)hipGetLastError();
    c->tabs.flag = nullptr;
    HIPCHK(nullptr, hipMalloc((void **)&c->tabs.flag, nflhip_ctx::kCmpSlots * sizeof(int)));
    HIPCHK(nullptr, hipMemset(c->tabs.flag, 0, nflhip_ctx::kCmpSlots * sizeof(int)));
  }
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// dispatch on the limb type
// ---------------------------------------------------------------------------
#define DISPATCH_T(ctx, EXPR16, EXPR32, EXPR64) \
  ((ctx)->shape.limb_bits == 16 ? (EXPR16) : (ctx)->shape.limb_bits == 32 ? (EXPR32) : (EXPR64))

static int set_device(const nflhip_ctx *ctx) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return NFLHIP_OK;
}
static size_t poly_bytes(const nflhip_ctx *ctx, size_t batch) { return batch * ctx->shape.nm * ctx->shape.n * ctx->word; }

static constexpr size_t kStageHostMax = (size_t)1 << 20;
static void free_stage(nflhip_ctx *ctx, int slot) {
  if (ctx->stage[slot]) (void)(ctx->stage_host[slot] ? hipHostFree(ctx->stage[slot]) : hipFree(ctx->stage[slot]));
  ctx->stage[slot] = nullptr;
  ctx->stage_bytes[slot] = 0;
  ctx->stage_host[slot] = false;
}
static int ensure_stage(nflhip_ctx *ctx, int slot, size_t bytes) {
  if (ctx->stage_bytes[slot] >= bytes) return NFLHIP_OK;
  free_stage(ctx, slot);
  if (bytes <= kStageHostMax && hipHostMalloc(&ctx->stage[slot], bytes, hipHostMallocDefault) == hipSuccess) {
    ctx->stage_host[slot] = true;
  } else {   // (also when the pinned allocation is refused -- a locked-memory limit: the copies take over)
    (void)hipGetLastError();
    ctx->stage[slot] = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->stage[slot], bytes));
  }
  ctx->stage_bytes[slot] = bytes;
  return NFLHIP_OK;
}
static int ensure_scratch(nflhip_ctx *ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return NFLHIP_OK;
  if (ctx->scratch) HIPCHK(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  HIPCHK(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return NFLHIP_OK;
}

// chunks of a batch in the n = 65536 pipeline (fill + drain cost ~0.7 chunk; small grids lose efficiency: 4 measured best)
#ifndef NFLHIP_PIPE_CHUNKS
#define NFLHIP_PIPE_CHUNKS 4
#endif
static constexpr int kPipeChunks = NFLHIP_PIPE_CHUNKS;


// Rows of 65536 / 32768 words in ONE launch of persistent workgroups (kernels_fast.hip launch_polymul_xcd_u64) instead
// of the chunked pipeline / the register-resident row kernels: by default for SMALL batches, where the other plans'
// fill and drain launches (n = 65536) or the one-workgroup-per-row grid (n = 32768) leave CUs idle.  Measured (MI355X):
// n = 65536 / 30 moduli +5 % at batch 4, +11 % at 8, +-0 at 16, -2 % at 64; n = 32768 / 2 moduli against the row
// kernels 68 vs 58 k products/s at batch 8, 302 vs 223 k at 32, 622 vs 716 k at 128.  Shape::plan (NFLHIP_XCD at context
// creation) forces either.
static bool xcd_on(const nflhip_ctx *ctx, size_t batch) {
  if (ctx->shape.plan >= 0) return ctx->shape.plan != 0;
  return batch * ctx->shape.nm <= (ctx->shape.logn == 15 ? 255u : 256u);
}

// hipGraph capture: the entry points only enqueue work on the caller's stream, so they can be captured.  The
// multi-launch plans additionally order successive calls on the shared scratch with events recorded OUTSIDE any
// capture; inside a capture those waits are illegal (and meaningless: a graph orders its own nodes), so they are
// skipped -- a graph that replays a multi-launch plan must not run concurrently with other work on the same context.
static bool is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs == hipStreamCaptureStatusActive;
}

// composed path: NTT(a)->c, NTT(b)->scratch, inverse with the product fused into its load
template <typename T>
static int polymul_composed(nflhip_ctx *ctx, T *c, const T *a, const T *b, int b_is_ntt, size_t batch, hipStream_t st) {
  hipError_t e;
  const T *bn = b;
  std::unique_lock<std::mutex> lk(ctx->scratch_mu, std::defer_lock);
  T *acopy = nullptr;
  // c may alias a or b: transform a into c first only when that does not clobber b
  const size_t bytes = poly_bytes(ctx, batch);
  lk.lock();
  const bool cap = is_capturing(st);
  // (from 256 rows on; below that the one-launch plan further down spreads a row over more CUs -- measured, same box:
  // batch 8 / 32 / 128 / 512 of two moduli 58 / 223 / 716 / 785 k products/s here against 68 / 302 / 622 / 779 k)
  if (sizeof(T) == 8 && !b_is_ntt && ctx->shape.logn == 15 && !(xcd_on(ctx, batch) && xcd_plan_bytes(ctx->shape, batch) != 0)) {
    // rows of 32768 words: b' = NTT(b) into the scratch (one read, one write), then c = INTT(NTT(a) (.) b') with the row
    // of a register-resident and b' streamed through the point-wise step (two reads, one write): 5 operand passes
    int rcs = ensure_scratch(ctx, bytes);
    if (rcs) return rcs;
    if (!cap && ctx->ev_scratch_valid) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_scratch, 0));
    if (!cap && ctx->ev_prev_valid)
      for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[k], 0));
    // (the level is read ONCE per product: both launches must agree on what b' is)
    const int pair = polymul_level() == 2 && ctx->tabs.mc_inc[1] ? 6 : 4;
    e = launch_row32k_u64(ctx->shape, ctx->tabs, pair, (uint64_t *)ctx->scratch, (const uint64_t *)b, nullptr, batch, st);
    if (e == hipSuccess)
      e = launch_row32k_u64(ctx->shape, ctx->tabs, pair + 1, (uint64_t *)c, (const uint64_t *)a, (const uint64_t *)ctx->scratch, batch, st);
    if (e == hipSuccess) {
      if (!cap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_scratch, st));
        ctx->ev_scratch_valid = true;
        for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->ev_scratch, 0));
      }
      return NFLHIP_OK;
    }
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "polymul: 32768-word row kernels");
  }
  if (sizeof(T) == 8 && !b_is_ntt && xcd_on(ctx, batch)) {
    // rows of 65536 / 32768 words: ONE launch of persistent workgroups, every row's three roles on one XCD and the
    // intermediates through that XCD's L2 (kernels_fast.hip launch_polymul_xcd_u64); needs only a ring of row slots
    const size_t need = xcd_plan_bytes(ctx->shape, batch);
    if (need) {
      int rcx = ensure_scratch(ctx, need);
      if (rcx) return rcx;
      if (!cap && ctx->ev_scratch_valid) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_scratch, 0));
      if (!cap && ctx->ev_prev_valid)
        for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[k], 0));
      e = launch_polymul_xcd_u64(ctx->shape, ctx->tabs, (uint64_t *)c, (const uint64_t *)a, (const uint64_t *)b, batch,
                                 ctx->scratch, st, polymul_level());
      if (e == hipSuccess) {
        if (!cap) {
          HIPCHK(ctx, hipEventRecord(ctx->ev_scratch, st));
          ctx->ev_scratch_valid = true;
          for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->ev_scratch, 0));
        }
        return NFLHIP_OK;
      }
      if (e != hipErrorNotSupported) return hipfail(ctx, e, "polymul: one-launch kernel");
    }
  }
  int rc = ensure_scratch(ctx, 2 * bytes);
  if (rc) return rc;
  T *s0 = (T *)ctx->scratch, *s1 = (T *)((char *)ctx->scratch + bytes);
  (void)acopy;
  if (sizeof(T) == 8 && ctx->shape.logn == 16) {
    // n = 65536: the streaming passes (HBM-bound) and the fused block kernel (VALU-bound) of neighbouring chunks share
    // every CU inside ONE kernel whose workgroups take three roles; consecutive launches on the caller's stream form the
    // pipeline: launch L = forward pass of chunk L, block products of chunk L-1, inverse pass of chunk L-2.
    const size_t pw = ctx->shape.nm * ctx->shape.n;
    size_t nchunk = (size_t)kPipeChunks;
    size_t edge = 0;   // polynomials in the first and in the last chunk when they are shorter than the others (experiment knob only)
#ifdef NFLHIP_ABLATION_KNOBS   // experiment builds only (tools/sessions/gpu_round5_a.sh, gpu_round6_g.sh): chunk count / edge chunks at run time, chunk aliasing
#include "ablation_knobs.inc"
#endif
    if (nchunk * 2 > batch) nchunk = batch >= 2 ? batch / 2 : 1;
    // chunk boundaries: uniform.  (Round 6 tried SHORT first / last chunks -- the first launch runs the forward role alone and the last
    // the inverse role alone, the pipeline's fill and drain -- through the experiment knob below: nothing beyond noise at batch 128,
    // +0.8 % for eight chunks at batch 256: profiles/r06_E_edge_chunks.txt.)
    if (nchunk < 3 || 2 * edge + (nchunk - 2) > batch) edge = 0;
    auto lo_of = [&](size_t ch) -> size_t {
      if (!edge) return batch * ch / nchunk;
      if (ch == 0) return 0;
      if (ch >= nchunk) return batch;
      return edge + (batch - 2 * edge) * (ch - 1) / (nchunk - 2);
    };
#ifdef NFLHIP_ABLATION_KNOBS   // ... and every chunk laid over chunk 0's memory (WRONG results by construction: the roles of
    // neighbouring launches then share one window of a, b, c and the scratch that fits the 256 MiB Infinity Cache -- what
    // the plan would run at if none of its passes reached HBM)
    auto at_of = [&](size_t ch) { return alias_chunks ? (size_t)0 : lo_of(ch); };
#else
    auto at_of = lo_of;
#endif
    bool supported = true;
    const int level = b_is_ntt ? 0 : polymul_level();   // read once: the three roles of a chunk run in different launches
    if (!cap && ctx->ev_scratch_valid) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_scratch, 0));  // a previous call on another stream
    if (!cap && ctx->ev_prev_valid)  // ... or a helper-stream plan (polymul_ntt_dev at this shape) still reading s0
      for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[k], 0));
    for (size_t L = 0; L < nchunk + 2 && supported; ++L) {
      const bool hf = L < nchunk, hv = L >= 1 && L - 1 < nchunk, hi = L >= 2 && L - 2 < nchunk;
      const size_t f0 = hf ? at_of(L) : 0, v0 = hv ? at_of(L - 1) : 0, i0 = hi ? at_of(L - 2) : 0;
      const int cf = hf ? (int)(lo_of(L + 1) - lo_of(L)) : 0, cv = hv ? (int)(lo_of(L) - lo_of(L - 1)) : 0, ci = hi ? (int)(lo_of(L - 1) - lo_of(L - 2)) : 0;
      // (b already transformed: its blocks are read from the caller's array by the block products, no forward pass, no scratch)
      e = launch_polymul_pipe64k_u64(ctx->shape, ctx->tabs, (uint64_t *)c + v0 * pw, (const uint64_t *)s0 + v0 * pw,
                                     (b_is_ntt ? (const uint64_t *)b : (const uint64_t *)s1) + v0 * pw, cv, (const uint64_t *)a + f0 * pw,
                                     (uint64_t *)s0 + f0 * pw, b_is_ntt ? nullptr : (const uint64_t *)b + f0 * pw,
                                     b_is_ntt ? nullptr : (uint64_t *)s1 + f0 * pw, cf, (uint64_t *)c + i0 * pw, ci, st, b_is_ntt != 0, level);
      if (e == hipErrorNotSupported && L == 0) { supported = false; break; }
      if (e != hipSuccess) return hipfail(ctx, e, "polymul: pipeline kernel");
    }
    if (supported) {
      if (!cap) {  // the scratch is reused by the next call on any stream: order it after this one
        HIPCHK(ctx, hipEventRecord(ctx->ev_scratch, st));
        ctx->ev_scratch_valid = true;
        for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->ev_scratch, 0));
      }
      return NFLHIP_OK;
    }
  }
  if (sizeof(T) == 8 && ctx->shape.logn > 12 && ctx->aux[0]) {
    // large rows: streaming outer passes, then the fused assembly kernel over the 4096-word blocks,
    // then the outer inverse passes (9 operand streams of HBM traffic instead of 13; 7 when b is already transformed:
    // its 4096-word blocks are exactly what the fused kernel would have computed for it).  The batch is cut
    // into chunks that alternate between two helper streams: one chunk's HBM-bound streaming passes
    // overlap another chunk's VALU-bound fused kernel.  Fully asynchronous w.r.t. the host.
    const size_t nm = ctx->shape.nm, pw = nm * ctx->shape.n;  // words per poly
    const size_t nchunk = batch >= 8 ? 8 : batch;
    HIPCHK(ctx, hipEventRecord(ctx->ev_start, st));
    for (int k = 0; k < 2; ++k) {
      HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->ev_start, 0));
      if (!cap && ctx->ev_prev_valid) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->ev_done[1 - k], 0));  // previous call's scratch use
    }
    bool unsupported = false;
    const int logi = 12;  // words (log2) per block of the fused kernel
    for (size_t ch = 0; ch < nchunk && !unsupported; ++ch) {
      const size_t lo = batch * ch / nchunk, hi = batch * (ch + 1) / nchunk, cnt = hi - lo;
      if (cnt == 0) continue;
      hipStream_t s = ctx->aux[ch & 1];
      const uint64_t *ak = (const uint64_t *)a + lo * pw, *bk = (const uint64_t *)b + lo * pw;
      uint64_t *ck = (uint64_t *)c + lo * pw, *s0k = (uint64_t *)s0 + lo * pw, *s1k = (uint64_t *)s1 + lo * pw;
      e = launch_outer_fwd_u64(ctx->shape, ctx->tabs, ak, s0k, cnt * nm, s, logi);
      if (e == hipSuccess && !b_is_ntt) e = launch_outer_fwd_u64(ctx->shape, ctx->tabs, bk, s1k, cnt * nm, s, logi);
      if (e != hipSuccess) return hipfail(ctx, e, "polymul: outer forward");
      const uint64_t *bblk = b_is_ntt ? bk : s1k;
      e = launch_polymul_blocks_asm_u64(ctx->shape, ctx->tabs, ck, s0k, bblk, cnt, s, b_is_ntt != 0);
      if (e == hipErrorNotSupported) { unsupported = true; break; }
      if (e != hipSuccess) return hipfail(ctx, e, "polymul: fused blocks");
      e = launch_outer_inv_u64(ctx->shape, ctx->tabs, ck, cnt * nm, s, logi);
      if (e != hipSuccess) return hipfail(ctx, e, "polymul: outer inverse");
    }
    for (int k = 0; k < 2; ++k) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_done[k], ctx->aux[k]));
      HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[k], 0));
    }
    if (!cap) ctx->ev_prev_valid = true;  // (inside a capture the helper streams forked from and joined back into st)
    if (!unsupported) return NFLHIP_OK;
    // (assembly kernel unavailable: fall through to the composed plan, ordered after the helper streams)
  }
  // an earlier asynchronous plan (issued on any stream) may still be using the scratch
  if (!cap && ctx->ev_scratch_valid) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_scratch, 0));
  if (!cap && ctx->ev_prev_valid)
    for (int k = 0; k < 2; ++k) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[k], 0));
  e = launch_ntt_fwd<T>(ctx->shape, ctx->tabs, a, s0, batch, st);
  if (e != hipSuccess) return hipfail(ctx, e, "polymul: ntt(a)");
  if (!b_is_ntt) {
    e = launch_ntt_fwd<T>(ctx->shape, ctx->tabs, b, s1, batch, st);
    if (e != hipSuccess) return hipfail(ctx, e, "polymul: ntt(b)");
    bn = s1;
  }
  e = launch_ntt_inv<T>(ctx->shape, ctx->tabs, s0, bn, c, batch, st);
  if (e != hipSuccess) return hipfail(ctx, e, "polymul: intt");
  // the scratch is reused by the next call on any stream: make that safe
  if (!cap) {
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hipfail(ctx, e, "polymul: sync");
  }
  return NFLHIP_OK;
}

extern "C" {

int nflhip_abi_version(void) { return NFLHIP_ABI_VERSION; }

// include/nflhip_debug.h: test hooks, not part of the product boundary
void nflhip_debug_gauss_tie_shift(int shift) { set_gauss_tie_shift(shift); }
static void pipe_stats(const nflhip_ctx *ctx, double out[4]);
void nflhip_debug_host_pipe_seconds(const nflhip_ctx *ctx, double out[4]) { pipe_stats(ctx, out); }

const char *nflhip_last_error(const nflhip_ctx *ctx) {
  (void)ctx;
  return g_last_error.c_str();
}

int nflhip_device_count(int *count) {
  if (!count) return fail(nullptr, NFLHIP_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return hipfail(nullptr, e, "hipGetDeviceCount");
  }
  *count = n;
  return NFLHIP_OK;
}

// First-use costs paid ONCE per device when its first context is created instead of inside whichever call comes first (measured,
// profiles/r06_first_use.txt: the first element-wise call 3.2 ms, the first transform 5.4 - 6.6 ms, the first CRT call 1.3 ms against
// 35 - 60 us afterwards -- the runtime loads a translation unit's code object at the first launch of any of its kernels, the generated
// kernels' module at its first use): one empty launch per translation unit, the module, and -- per context -- the three device staging
// buffers of the host-pointer entry points at one polynomial's size: + 20 ms on the first context of a process, nothing afterwards.
static int warm_up_device(nflhip_ctx *c) {
  static std::once_flag once[16];
  if (c->device >= 0 && c->device < 16) {
    hipError_t e = hipSuccess;
    std::call_once(once[c->device], [&] {
      hipStream_t st = c->hstream;
      hipError_t (*const tus[])(hipStream_t) = {nflhip::warm_generic, nflhip::warm_fast, nflhip::warm_crt, nflhip::warm_crt_mfma,
                                                nflhip::warm_sample, nflhip::warm_wave};
      for (auto f : tus)
        if (e == hipSuccess) e = f(st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    });
    if (e != hipSuccess) return hipfail(nullptr, e, "first-use warm-up");
  }
  for (int slot = 0; slot < 3; ++slot) {
    int rc = ensure_stage(c, slot, c->shape.n * c->shape.nm * c->word);
    if (rc) return rc;
  }
  return NFLHIP_OK;
}

static int ctx_create_mode(nflhip_ctx **out, int device, int limb_bits, size_t degree, size_t nmoduli, const void *P,
                           const void *primitive_roots, const void *invkmax, int kmax_log2, int cyclic) {
  if (!out) return fail(nullptr, NFLHIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (limb_bits != 16 && limb_bits != 32 && limb_bits != 64)
    return fail(nullptr, NFLHIP_ERR_INVALID, "limb_bits must be 16, 32 or 64");
  if (!P || !primitive_roots || !invkmax) return fail(nullptr, NFLHIP_ERR_INVALID, "NULL parameter table");
  if (degree == 0 || (degree & (degree - 1)) != 0) return fail(nullptr, NFLHIP_ERR_INVALID, "degree must be a power of two");
  if (kmax_log2 < 1 || kmax_log2 > 30 || degree > (((size_t)1) << kmax_log2))
    return fail(nullptr, NFLHIP_ERR_INVALID, "degree exceeds kMaxPolyDegree (core.hpp:59-60)");
  if (degree < 4) return fail(nullptr, NFLHIP_ERR_UNSUPPORTED, "degree < 4 is not supported by the device engine");
  if (nmoduli == 0 || nmoduli > 1024) return fail(nullptr, NFLHIP_ERR_INVALID, "nmoduli out of range");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, NFLHIP_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, NFLHIP_ERR_INVALID, "device index out of range");
  HIPCHK(nullptr, hipSetDevice(device));
  nflhip_ctx *c = new (std::nothrow) nflhip_ctx();
  if (!c) return fail(nullptr, NFLHIP_ERR_NOMEM, "out of host memory");
  c->device = device;
  c->cyclic = cyclic;
  c->word = (size_t)limb_bits / 8;
  c->shape.limb_bits = limb_bits;
  c->shape.n = degree;
  c->shape.nm = nmoduli;
  c->shape.small_delta = 1;
  c->shape.nm_small = 0;
  // the environment is read HERE, once per context (include/nflhip.h "environment")
  {
    const char *v = getenv("NFLHIP_VARIANT");
    c->shape.compiled_only = v && (!strcmp(v, "hipcc") || !strcmp(v, "compiled")) ? 1 : 0;
    const char *x = getenv("NFLHIP_XCD");
    c->shape.plan = x && *x ? (atoi(x) != 0 ? 1 : 0) : -1;
  }
  c->shape.logn = 0;
  while ((((size_t)1) << c->shape.logn) < degree) c->shape.logn++;
  int rc;
  try {  // (host containers: no exception may cross the C boundary)
    rc = limb_bits == 16   ? build_tables<uint16_t>(c, P, primitive_roots, invkmax, kmax_log2)
         : limb_bits == 32 ? build_tables<uint32_t>(c, P, primitive_roots, invkmax, kmax_log2)
                           : build_tables<uint64_t>(c, P, primitive_roots, invkmax, kmax_log2);
  } catch (const std::bad_alloc &) {
    rc = fail(nullptr, NFLHIP_ERR_NOMEM, "out of host memory while building the tables");
  } catch (const std::exception &ex) {
    rc = fail(nullptr, NFLHIP_ERR_INVALID, std::string("table construction failed: ") + ex.what());
  }
  if (rc == NFLHIP_OK) {
    hipError_t se = hipStreamCreateWithFlags(&c->hstream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && se == hipSuccess; ++k) {
      se = hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking);
      if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming);
    }
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_scratch, hipEventDisableTiming);
    if (se != hipSuccess) rc = hipfail(nullptr, se, "hipStreamCreate");
  }
  if (rc == NFLHIP_OK) rc = warm_up_device(c);
  if (rc != NFLHIP_OK) {
    nflhip_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return NFLHIP_OK;
}

int nflhip_ctx_create(nflhip_ctx **out, int device, int limb_bits, size_t degree, size_t nmoduli, const void *P,
                      const void *primitive_roots, const void *invkmax, int kmax_log2) {
  return ctx_create_mode(out, device, limb_bits, degree, nmoduli, P, primitive_roots, invkmax, kmax_log2, 0);
}

int nflhip_ctx_destroy(nflhip_ctx *ctx) {
  if (!ctx) return NFLHIP_OK;
  for (nflhip_ctx *child : ctx->row_ctx) nflhip_ctx_destroy(child);
  ctx->row_ctx.clear();
  (void)hipSetDevice(ctx->device);
  if (ctx->hstream) (void)hipStreamDestroy(ctx->hstream);
  for (int k = 0; k < 2; ++k) {
    if (ctx->aux[k]) { (void)hipStreamSynchronize(ctx->aux[k]); (void)hipStreamDestroy(ctx->aux[k]); }
    if (ctx->ev_done[k]) (void)hipEventDestroy(ctx->ev_done[k]);
  }
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_scratch) (void)hipEventDestroy(ctx->ev_scratch);
  pipe_destroy(ctx);
  for (int i = 0; i < 4; ++i) free_stage(ctx, i);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->tabs.psi) (void)hipFree(ctx->tabs.psi);
  if (ctx->tabs.psi_lm) (void)hipFree(ctx->tabs.psi_lm);
  if (ctx->tabs.mc) (void)hipFree(ctx->tabs.mc);
  for (int i = 0; i < 2; ++i)
    if (ctx->tabs.mc_inc[i]) (void)hipFree(ctx->tabs.mc_inc[i]);
  if (ctx->tabs.qhat) (void)hipFree(ctx->tabs.qhat);
  if (ctx->tabs.qsh) (void)hipFree(ctx->tabs.qsh);
  if (ctx->tabs.qparts) (void)hipFree(ctx->tabs.qparts);
  if (ctx->tabs.crt_bfrag) (void)hipFree(ctx->tabs.crt_bfrag);
  if (ctx->tabs.crt_bproj) (void)hipFree(ctx->tabs.crt_bproj);
  if (ctx->tabs.crt_coff) (void)hipFree(ctx->tabs.crt_coff);
  if (ctx->tabs.crt_c2048) (void)hipFree(ctx->tabs.crt_c2048);
  if (ctx->tabs.bparts) (void)hipFree(ctx->tabs.bparts);
  if (ctx->tabs.flag) (void)(ctx->flag_host ? hipHostFree(ctx->tabs.flag) : hipFree(ctx->tabs.flag));
  if (ctx->tabs.qhat_w) (void)hipFree(ctx->tabs.qhat_w);
  if (ctx->tabs.qsh_w) (void)hipFree(ctx->tabs.qsh_w);
  delete ctx;
  return NFLHIP_OK;
}

int nflhip_ctx_device(const nflhip_ctx *ctx) { return ctx ? ctx->device : -1; }
size_t nflhip_degree(const nflhip_ctx *ctx) { return ctx ? ctx->shape.n : 0; }
size_t nflhip_nmoduli(const nflhip_ctx *ctx) { return ctx ? ctx->shape.nm : 0; }
int nflhip_limb_bits(const nflhip_ctx *ctx) { return ctx ? ctx->shape.limb_bits : 0; }
size_t nflhip_crt_limbs(const nflhip_ctx *ctx) { return ctx ? ctx->shape.crt_L : 0; }

int nflhip_get_table(const nflhip_ctx *ctx, int which, size_t cm, void *host_out, size_t host_bytes) {
  if (!ctx || !host_out) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (cm >= ctx->shape.nm) return fail(ctx, NFLHIP_ERR_INVALID, "modulus index out of range");
  const size_t w = ctx->word;
  int rc = which >= NFLHIP_TAB_PHIS ? NFLHIP_OK : set_device(ctx);
  if (rc) return rc;
  const size_t mcsz = sizeof(ModConst<uint64_t>) / 8 * w;  // sizeof(ModConst<T>)
  switch (which) {
    case NFLHIP_TAB_PSI: {
      const size_t bytes = ctx->shape.n * 2 * w;
      if (host_bytes < bytes) return fail(ctx, NFLHIP_ERR_INVALID, "output buffer too small");
      HIPCHK(ctx, hipMemcpy(host_out, (const char *)ctx->tabs.psi + cm * bytes, bytes, hipMemcpyDeviceToHost));
      return NFLHIP_OK;
    }
    case NFLHIP_TAB_MODULUS:
    case NFLHIP_TAB_INVDEGREE: {
      if (host_bytes < w) return fail(ctx, NFLHIP_ERR_INVALID, "output buffer too small");
      const size_t off = cm * mcsz + (which == NFLHIP_TAB_MODULUS ? 0 : 3 * w);
      HIPCHK(ctx, hipMemcpy(host_out, (const char *)ctx->tabs.mc + off, w, hipMemcpyDeviceToHost));
      return NFLHIP_OK;
    }
    case NFLHIP_TAB_PHIS:
    case NFLHIP_TAB_SHOUPPHIS:
    case NFLHIP_TAB_INVPOLY_INVPHIS:
    case NFLHIP_TAB_SHOUPINVPOLY_INVPHIS:
    case NFLHIP_TAB_OMEGAS:
    case NFLHIP_TAB_INVOMEGAS: {
      // the reference's own table layouts (poly.hpp:228-237), rebuilt on the host from phi: what a caller holding
      // core::base sees.  Host arithmetic, once per request; the device never reads these.
      const size_t n = ctx->shape.n, words = (which == NFLHIP_TAB_OMEGAS || which == NFLHIP_TAB_INVOMEGAS) ? 2 * n : n;
      if (host_bytes < words * w) return fail(ctx, NFLHIP_ERR_INVALID, "output buffer too small");
      const uint64_t p = ctx->h_P[cm], phi = ctx->h_phi[cm];
      const int wb = ctx->shape.limb_bits;
      std::vector<uint64_t> v(words, 0);
      const uint64_t invphi = powmod_h(phi, 2 * n - 1, p);
      if (which == NFLHIP_TAB_PHIS || which == NFLHIP_TAB_SHOUPPHIS) {  // core.hpp:649-656
        uint64_t t = 1;
        for (size_t i = 0; i < n; ++i) {
          v[i] = which == NFLHIP_TAB_PHIS ? t : shoup_h(t, p, wb);
          t = mulmod_h(t, phi, p);
        }
      } else if (which == NFLHIP_TAB_INVPOLY_INVPHIS || which == NFLHIP_TAB_SHOUPINVPOLY_INVPHIS) {  // core.hpp:664-676
        uint64_t t = mulmod_h(ctx->h_invk[cm], (((uint64_t)1) << ctx->kmax_log2) / n, p);
        for (size_t i = 0; i < n; ++i) {
          v[i] = which == NFLHIP_TAB_INVPOLY_INVPHIS ? t : shoup_h(t, p, wb);
          t = mulmod_h(t, invphi, p);
        }
      } else {  // core::prep_wtab, core.hpp:564-581: stage-concatenated powers, Shoup companions at offset n
        uint64_t wcur = which == NFLHIP_TAB_OMEGAS ? mulmod_h(phi, phi, p) : mulmod_h(invphi, invphi, p);
        size_t pos = 0;
        for (size_t K = n; K >= 2; K /= 2) {
          uint64_t wi = 1;
          for (size_t i = 0; i < K / 2; ++i, ++pos) {
            v[pos] = wi;
            v[n + pos] = shoup_h(wi, p, wb);
            wi = mulmod_h(wi, wcur, p);
          }
          wcur = mulmod_h(wcur, wcur, p);
        }
      }
      for (size_t i = 0; i < words; ++i) {
        if (w == 8) ((uint64_t *)host_out)[i] = v[i];
        else if (w == 4) ((uint32_t *)host_out)[i] = (uint32_t)v[i];
        else ((uint16_t *)host_out)[i] = (uint16_t)v[i];
      }
      return NFLHIP_OK;
    }
    default: return fail(ctx, NFLHIP_ERR_INVALID, "unknown table id");
  }
}

int nflhip_get_crt_constant(const nflhip_ctx *ctx, int what, size_t cm, uint64_t *host_out, size_t cap, size_t *nlimbs) {
  if (!ctx || !host_out || !nlimbs) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  const std::vector<uint64_t> *src = nullptr;
  if (what == 0) src = &ctx->h_Q;
  else if (what == 1 && cm < ctx->shape.nm) src = &ctx->h_lifting[cm];
  else return fail(ctx, NFLHIP_ERR_INVALID, "unknown CRT constant");
  size_t n = src->size();
  while (n > 0 && (*src)[n - 1] == 0) --n;
  *nlimbs = n;
  if (cap < n) return fail(ctx, NFLHIP_ERR_INVALID, "output buffer too small");
  memset(host_out, 0, cap * sizeof(uint64_t));
  memcpy(host_out, src->data(), n * sizeof(uint64_t));
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------
#define CHECK_CTX(ctx)                                                \
  do {                                                                \
    if (!(ctx)) return fail(nullptr, NFLHIP_ERR_INVALID, "ctx is NULL"); \
    int _rc = set_device(ctx);                                        \
    if (_rc) return _rc;                                              \
  } while (0)

int nflhip_ntt_fwd_dev(nflhip_ctx *ctx, void *d, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (!d && batch) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (ctx->shape.limb_bits == 64) {
    e = launch_ntt_fwd_fast_u64(ctx->shape, ctx->tabs, (const uint64_t *)d, (uint64_t *)d, batch, st);
    if (e == hipErrorNotSupported)
      e = launch_row1024_u64(ctx->shape, ctx->tabs, 2, (uint64_t *)d, (const uint64_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_fwd(fast)");
  }
  if (ctx->shape.limb_bits == 32) {
    e = launch_row1024_u32(ctx->shape, ctx->tabs, 2, (uint32_t *)d, (const uint32_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_fwd(u32)");
  }
  if (ctx->shape.limb_bits == 16) {
    e = launch_row128_u16_asm(ctx->shape, ctx->tabs, 2, (uint16_t *)d, (const uint16_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_fwd(u16)");
  }
  e = DISPATCH_T(ctx, launch_ntt_fwd<uint16_t>(ctx->shape, ctx->tabs, (const uint16_t *)d, (uint16_t *)d, batch, st),
                 launch_ntt_fwd<uint32_t>(ctx->shape, ctx->tabs, (const uint32_t *)d, (uint32_t *)d, batch, st),
                 launch_ntt_fwd<uint64_t>(ctx->shape, ctx->tabs, (const uint64_t *)d, (uint64_t *)d, batch, st));
  if (e != hipSuccess) return hipfail(ctx, e, "ntt_fwd");
  return NFLHIP_OK;
}

int nflhip_ntt_inv_dev(nflhip_ctx *ctx, void *d, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (!d && batch) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (ctx->shape.limb_bits == 64) {
    e = launch_ntt_inv_fast_u64(ctx->shape, ctx->tabs, (const uint64_t *)d, (uint64_t *)d, batch, st);
    if (e == hipErrorNotSupported)
      e = launch_row1024_u64(ctx->shape, ctx->tabs, 3, (uint64_t *)d, (const uint64_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_inv(fast)");
  }
  if (ctx->shape.limb_bits == 32) {
    e = launch_row1024_u32(ctx->shape, ctx->tabs, 3, (uint32_t *)d, (const uint32_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_inv(u32)");
  }
  if (ctx->shape.limb_bits == 16) {
    e = launch_row128_u16_asm(ctx->shape, ctx->tabs, 3, (uint16_t *)d, (const uint16_t *)d, nullptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "ntt_inv(u16)");
  }
  e = DISPATCH_T(ctx,
                 launch_ntt_inv<uint16_t>(ctx->shape, ctx->tabs, (const uint16_t *)d, nullptr, (uint16_t *)d, batch, st),
                 launch_ntt_inv<uint32_t>(ctx->shape, ctx->tabs, (const uint32_t *)d, nullptr, (uint32_t *)d, batch, st),
                 launch_ntt_inv<uint64_t>(ctx->shape, ctx->tabs, (const uint64_t *)d, nullptr, (uint64_t *)d, batch, st));
  if (e != hipSuccess) return hipfail(ctx, e, "ntt_inv");
  return NFLHIP_OK;
}

int nflhip_pointwise_dev(nflhip_ctx *ctx, int op, void *o, const void *a, const void *b, const void *bp, size_t batch,
                         void *stream) {
  CHECK_CTX(ctx);
  if (op < 0 || op > 4) return fail(ctx, NFLHIP_ERR_INVALID, "unknown element-wise op");
  if (batch && (!o || !a || (op != NFLHIP_OP_COMPUTE_SHOUP && !b) || (op == NFLHIP_OP_MUL_SHOUP && !bp)))
    return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(
      ctx,
      launch_pointwise<uint16_t>(ctx->shape, ctx->tabs, op, (uint16_t *)o, (const uint16_t *)a, (const uint16_t *)b,
                                 (const uint16_t *)bp, batch, st),
      launch_pointwise<uint32_t>(ctx->shape, ctx->tabs, op, (uint32_t *)o, (const uint32_t *)a, (const uint32_t *)b,
                                 (const uint32_t *)bp, batch, st),
      launch_pointwise<uint64_t>(ctx->shape, ctx->tabs, op, (uint64_t *)o, (const uint64_t *)a, (const uint64_t *)b,
                                 (const uint64_t *)bp, batch, st));
  if (e != hipSuccess) return hipfail(ctx, e, "pointwise");
  return NFLHIP_OK;
}

static int eval_dev(nflhip_ctx *ctx, void *out, const void *const *ops, size_t nops, const unsigned char *prog, size_t len,
                    size_t batch, void *stream, const unsigned *strides = nullptr, unsigned out_stride = 1) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(
      ctx, launch_eval_expr<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)out, ops, (int)nops, prog, (int)len, batch, st, strides, out_stride),
      launch_eval_expr<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)out, ops, (int)nops, prog, (int)len, batch, st, strides, out_stride),
      launch_eval_expr<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)out, ops, (int)nops, prog, (int)len, batch, st, strides, out_stride));
  if (e == hipErrorInvalidValue) return fail(ctx, NFLHIP_ERR_INVALID, "malformed expression program");
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "row shorter than one 16-byte vector");
  if (e != hipSuccess) return hipfail(ctx, e, "eval");
  return NFLHIP_OK;
}

static int polymul_any(nflhip_ctx *ctx, void *c, const void *a, const void *b, int b_is_ntt, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (batch && (!c || !a || !b)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  if (batch == 0) return NFLHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (ctx->shape.limb_bits == 64) {
    hipError_t e = launch_polymul_fast_u64(ctx->shape, ctx->tabs, (uint64_t *)c, (const uint64_t *)a, (const uint64_t *)b,
                                           b_is_ntt, batch, st);
    if (e == hipErrorNotSupported)
      e = launch_row1024_u64(ctx->shape, ctx->tabs, b_is_ntt ? 1 : 0, (uint64_t *)c, (const uint64_t *)a, (const uint64_t *)b,
                             batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "polymul(fast)");
  }
  if (ctx->shape.limb_bits == 32) {
    hipError_t e = launch_row1024_u32(ctx->shape, ctx->tabs, b_is_ntt ? 1 : 0, (uint32_t *)c, (const uint32_t *)a,
                                      (const uint32_t *)b, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "polymul(u32)");
  }
  if (ctx->shape.limb_bits == 16) {
    hipError_t e = launch_row128_u16_asm(ctx->shape, ctx->tabs, b_is_ntt ? 1 : 0, (uint16_t *)c, (const uint16_t *)a, (const uint16_t *)b, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "polymul(u16)");
  }
  return DISPATCH_T(ctx, polymul_composed<uint16_t>(ctx, (uint16_t *)c, (const uint16_t *)a, (const uint16_t *)b, b_is_ntt, batch, st),
                    polymul_composed<uint32_t>(ctx, (uint32_t *)c, (const uint32_t *)a, (const uint32_t *)b, b_is_ntt, batch, st),
                    polymul_composed<uint64_t>(ctx, (uint64_t *)c, (const uint64_t *)a, (const uint64_t *)b, b_is_ntt, batch, st));
}

int nflhip_eval_dev(nflhip_ctx *ctx, void *d_out, const void *const *d_operands, size_t noperands,
                    const unsigned char *program, size_t proglen, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (!program || !d_operands || (batch && !d_out)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (noperands == 0 || noperands > NFLHIP_EXPR_MAX_OPERANDS || proglen == 0 || proglen > NFLHIP_EXPR_MAX_LEN)
    return fail(ctx, NFLHIP_ERR_INVALID, "expression program too large");
  for (size_t i = 0; i < noperands; ++i)
    if (batch && !d_operands[i]) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  return eval_dev(ctx, d_out, d_operands, noperands, program, proglen, batch, stream);
}

int nflhip_eval_strided_dev(nflhip_ctx *ctx, void *d_out, size_t out_stride, const void *const *d_operands,
                            const size_t *strides, size_t noperands, const unsigned char *program, size_t proglen,
                            size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (!program || !d_operands || !strides || (batch && !d_out)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (noperands == 0 || noperands > NFLHIP_EXPR_MAX_OPERANDS || proglen == 0 || proglen > NFLHIP_EXPR_MAX_LEN)
    return fail(ctx, NFLHIP_ERR_INVALID, "expression program too large");
  unsigned sd[NFLHIP_EXPR_MAX_OPERANDS];
  for (size_t i = 0; i < noperands; ++i) {
    if (batch && !d_operands[i]) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
    if (strides[i] > 0xffffffffu) return fail(ctx, NFLHIP_ERR_INVALID, "operand stride out of range");
    sd[i] = (unsigned)strides[i];
  }
  if (out_stride == 0 || out_stride > 0xffffffffu) return fail(ctx, NFLHIP_ERR_INVALID, "the result stride must be positive");
  return eval_dev(ctx, d_out, d_operands, noperands, program, proglen, batch, stream, sd, (unsigned)out_stride);
}

int nflhip_polymul_dev(nflhip_ctx *ctx, void *c, const void *a, const void *b, size_t batch, void *stream) {
  return polymul_any(ctx, c, a, b, 0, batch, stream);
}
int nflhip_polymul_ntt_dev(nflhip_ctx *ctx, void *c, const void *a, const void *bntt, size_t batch, void *stream) {
  return polymul_any(ctx, c, a, bntt, 1, batch, stream);
}

// ---------------------------------------------------------------------------
// transform-fused pipelines
// ---------------------------------------------------------------------------
static int check_operand(const nflhip_ctx *ctx, const nflhip_operand *o, size_t batch, bool words_only, const char *what) {
  if (!o || (batch && !o->ptr)) return fail(ctx, NFLHIP_ERR_INVALID, std::string("NULL operand: ") + what);
  if (o->format < NFLHIP_FMT_WORDS || o->format > NFLHIP_FMT_I32 || (words_only && o->format != NFLHIP_FMT_WORDS))
    return fail(ctx, NFLHIP_ERR_INVALID, std::string("operand format: ") + what);
  if (o->stride > 0xffffffffu || (batch > 1 && (uint64_t)o->stride * (batch - 1) > 0xffffffffull))
    return fail(ctx, NFLHIP_ERR_INVALID, std::string("operand stride out of range: ") + what);
  return NFLHIP_OK;
}

static hipError_t expand_any(nflhip_ctx *ctx, void *dst, const nflhip_operand *src, size_t batch, hipStream_t st) {
  return DISPATCH_T(ctx, launch_expand_small<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)dst, src->ptr, src->format, (unsigned)src->stride, batch, st),
                    launch_expand_small<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)dst, src->ptr, src->format, (unsigned)src->stride, batch, st),
                    launch_expand_small<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)dst, src->ptr, src->format, (unsigned)src->stride, batch, st));
}

// out0 of a two-result call lies over (part of) an input of the SECOND result (k1 or e1): the one aliasing in which a plan that
// stores the first result before it has read the second one's inputs would change them under its own feet
static bool first_result_overlaps(nflhip_ctx *ctx, const void *out0, const nflhip_operand *in, size_t batch) {
  const size_t per = in->format == NFLHIP_FMT_WORDS ? poly_bytes(ctx, 1) : ctx->shape.n << (in->format - NFLHIP_FMT_I8);
  const char *o = (const char *)out0, *k = (const char *)in->ptr;
  return o < k + ((batch - 1) * in->stride + 1) * per && k < o + poly_bytes(ctx, batch);
}

static int fused_fwd_composed(nflhip_ctx *ctx, void *out0, void *out1, const nflhip_operand *x, const nflhip_operand *k0,
                              const nflhip_operand *e0, const nflhip_operand *k1, const nflhip_operand *e1, size_t batch,
                              hipStream_t st) {
  // the same result from the plain kernels: expand / gather into the context's scratch, transform there, one fused
  // multiply-add pass per result (what serves every shape without a generated kernel, and NFLHIP_VARIANT=hipcc)
  const size_t bytes = poly_bytes(ctx, batch);
  std::unique_lock<std::mutex> lk(ctx->scratch_mu);
  const bool cap = is_capturing(st);
  // three polynomials of scratch per batch element only when the first result lies over an input of the second one (then every
  // input is read before anything is stored); otherwise two: e1 is transformed into e0's place after the first result is out
  const bool overlap = out1 && (first_result_overlaps(ctx, out0, e1, batch) || first_result_overlaps(ctx, out0, k1, batch));
  int rc = ensure_scratch(ctx, (overlap ? 3 : 2) * bytes);
  if (rc) return rc;
  if (!cap && ctx->ev_scratch_valid) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_scratch, 0));
  if (!cap && ctx->ev_prev_valid)
    for (int j = 0; j < 2; ++j) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_done[j], 0));
  void *s0 = ctx->scratch, *s1 = (char *)ctx->scratch + bytes, *s2 = (char *)ctx->scratch + 2 * bytes;
  static const unsigned char prog[5] = {0, 1, NFLHIP_EXPR_MUL, 2, NFLHIP_EXPR_ADD};
  // rows of 32768 words with int8 polynomials and keys shared by the batch (the LWE demo at the reference's largest configuration):
  // the noise polynomials go from their bytes straight to NTT words in the scratch, then ONE kernel transforms x in registers and
  // writes X k + e' for both results -- 6 polynomial passes over HBM instead of 17
  if (ctx->shape.limb_bits == 64 && ctx->shape.logn == 15 && x->format == NFLHIP_FMT_I8 && e0->format == NFLHIP_FMT_I8 &&
      (!out1 || e1->format == NFLHIP_FMT_I8) && x->stride == 1 && e0->stride == 1 && (!out1 || e1->stride == 1) && k0->stride == 0 &&
      (!out1 || k1->stride == 0)) {
    hipError_t e5 = launch_row32k_fwd_i8_u64(ctx->shape, ctx->tabs, (uint64_t *)s0, e0->ptr, batch, st);
    if (e5 == hipSuccess && out1) e5 = launch_row32k_fwd_i8_u64(ctx->shape, ctx->tabs, (uint64_t *)s1, e1->ptr, batch, st);
    if (e5 == hipSuccess)
      e5 = launch_row32k_fwd_fma_i8_u64(ctx->shape, ctx->tabs, (uint64_t *)out0, (uint64_t *)out1, x->ptr, (const uint64_t *)k0->ptr,
                                        (const uint64_t *)s0, out1 ? (const uint64_t *)k1->ptr : nullptr, (const uint64_t *)s1, batch, st);
    if (e5 != hipErrorNotSupported) {
      if (e5 != hipSuccess) return hipfail(ctx, e5, "fwd_fma: 32768-word row kernels");
      if (!cap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_scratch, st));
        ctx->ev_scratch_valid = true;
        for (int j = 0; j < 2; ++j)
          if (ctx->aux[j]) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[j], ctx->ev_scratch, 0));
      }
      return NFLHIP_OK;
    }
  }
  hipError_t e = expand_any(ctx, s0, x, batch, st);
  if (e != hipSuccess) return hipfail(ctx, e, "fwd_fma: expand");
  rc = nflhip_ntt_fwd_dev(ctx, s0, batch, st);
  if (rc) return rc;
  // a result may alias a dense input of the same call (nflhip.h): every input polynomial is read into the scratch BEFORE the
  // first result is stored (e1 may be out0), and when out0 overlaps a per-element k1 the first result waits in the scratch
  // until the second has been computed -- the generated kernels read a whole row of every operand before they store, the
  // composed plan gives the same guarantee
  if (!overlap) {
    for (int h = 0; h < (out1 ? 2 : 1); ++h) {
      e = expand_any(ctx, s1, h ? e1 : e0, batch, st);
      if (e != hipSuccess) return hipfail(ctx, e, "fwd_fma: expand");
      rc = nflhip_ntt_fwd_dev(ctx, s1, batch, st);
      if (rc) return rc;
      const nflhip_operand *kk = h ? k1 : k0;
      const void *ops[3] = {s0, kk->ptr, s1};
      const unsigned sd[3] = {1, (unsigned)kk->stride, 1};
      rc = eval_dev(ctx, h ? out1 : out0, ops, 3, prog, sizeof(prog), batch, st, sd, 1);
      if (rc) return rc;
    }
  } else {
    for (int h = 0; h < 2; ++h) {
      void *sh = h ? s2 : s1;
      e = expand_any(ctx, sh, h ? e1 : e0, batch, st);
      if (e != hipSuccess) return hipfail(ctx, e, "fwd_fma: expand");
      rc = nflhip_ntt_fwd_dev(ctx, sh, batch, st);
      if (rc) return rc;
    }
    const bool hold0 = first_result_overlaps(ctx, out0, k1, batch);
    for (int h = 0; h < 2; ++h) {
      const nflhip_operand *kk = h ? k1 : k0;
      void *sh = h ? s2 : s1;
      const void *ops[3] = {s0, kk->ptr, sh};
      const unsigned sd[3] = {1, (unsigned)kk->stride, 1};
      rc = eval_dev(ctx, h ? out1 : (hold0 ? s1 : out0), ops, 3, prog, sizeof(prog), batch, st, sd, 1);
      if (rc) return rc;
    }
    if (hold0) HIPCHK(ctx, hipMemcpyAsync(out0, s1, bytes, hipMemcpyDeviceToDevice, st));
  }
  if (!cap) {  // the scratch is reused by the next call on any stream: order it after this one
    HIPCHK(ctx, hipEventRecord(ctx->ev_scratch, st));
    ctx->ev_scratch_valid = true;
    for (int j = 0; j < 2; ++j)
      if (ctx->aux[j]) HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[j], ctx->ev_scratch, 0));
  }
  return NFLHIP_OK;
}

static int fused_fwd_any(nflhip_ctx *ctx, void *out0, void *out1, const nflhip_operand *x, const nflhip_operand *k0,
                         const nflhip_operand *e0, const nflhip_operand *k1, const nflhip_operand *e1, size_t batch,
                         void *stream) {
  CHECK_CTX(ctx);
  const bool two = k1 != nullptr;
  if (batch && (!out0 || (two && !out1))) return fail(ctx, NFLHIP_ERR_INVALID, "NULL result pointer");
  int rc = check_operand(ctx, x, batch, false, "x");
  if (!rc) rc = check_operand(ctx, k0, batch, true, "k0");
  if (!rc) rc = check_operand(ctx, e0, batch, false, "e0");
  if (!rc && two) rc = check_operand(ctx, k1, batch, true, "k1");
  if (!rc && two) rc = check_operand(ctx, e1, batch, false, "e1");
  if (rc) return rc;
  if (batch == 0) return NFLHIP_OK;
  // a result may lie over a DENSE input (nflhip.h); over an operand the whole batch shares (stride 0, batch > 1) it would be
  // written by one batch element while the others still read it: refused, not raced
  if (batch > 1) {
    const nflhip_operand *ins[5] = {x, k0, e0, k1, e1};
    void *outs[2] = {out0, two ? out1 : nullptr};
    for (const nflhip_operand *in : ins)
      for (void *o : outs)
        if (in && o && in->stride == 0 && first_result_overlaps(ctx, o, in, batch))
          return fail(ctx, NFLHIP_ERR_INVALID, "a result overlaps an operand shared by the batch (stride 0)");
  }
  hipStream_t st = (hipStream_t)stream;
  // (the generated two-result kernels store a row of out0 before they load that row of k1 and, the row-resident ones, of e1: a
  // first result laid over an input of the second -- legal, nflhip.h -- takes the composed plan, which reads every polynomial
  // before it stores and holds out0 back when it must)
  if (ctx->shape.limb_bits == 64 && !(two && (first_result_overlaps(ctx, out0, k1, batch) || first_result_overlaps(ctx, out0, e1, batch)))) {
    const void *xs[3] = {x->ptr, e0->ptr, two ? e1->ptr : nullptr};
    const unsigned xstr[3] = {(unsigned)x->stride, (unsigned)e0->stride, two ? (unsigned)e1->stride : 0u};
    const int xf[3] = {x->format, e0->format, two ? e1->format : 0};
    const void *ks[2] = {k0->ptr, two ? k1->ptr : nullptr};
    const unsigned kstr[2] = {(unsigned)k0->stride, two ? (unsigned)k1->stride : 0u};
    hipError_t e = launch_fused_asm_u64(ctx->shape, ctx->tabs, two ? 0 : 1, (uint64_t *)out0, (uint64_t *)out1, xs, xstr, xf, ks, kstr,
                                        batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "fwd_fma(fused)");
  }
  // short rows (1024 / 2048 words; 4096 for 32-bit limbs): the wave-per-row kernels transform x once, keep it in registers and
  // multiply-add each transformed noise row in place (kernels_wave.hip k_row_fwd_fma) -- operands of one format, strides 0 / 1
  const bool aliased = two && (first_result_overlaps(ctx, out0, k1, batch) || first_result_overlaps(ctx, out0, e1, batch));
  // (compact rows are fetched 16 bytes per lane: their arrays must be 16-byte aligned -- device allocations are; a caller's odd offset
  //  into one takes the composed plan)
  const bool aligned16 = ((((uintptr_t)x->ptr) | ((uintptr_t)e0->ptr) | (two ? (uintptr_t)e1->ptr : 0)) & 15) == 0;
  if (!aliased && aligned16 && !ctx->cyclic && !ctx->shape.compiled_only && x->format == e0->format && (!two || e1->format == x->format) && x->stride <= 1 &&
      e0->stride <= 1 && k0->stride <= 1 && (!two || (e1->stride <= 1 && k1->stride <= 1))) {
    hipError_t e = hipErrorNotSupported;
    if (ctx->shape.limb_bits == 32)
      e = launch_row_fwd_fma_u32(ctx->shape, ctx->tabs, x->format, (uint32_t *)out0, (uint32_t *)out1, x->ptr, (unsigned)x->stride,
                                 (const uint32_t *)k0->ptr, (unsigned)k0->stride, e0->ptr, (unsigned)e0->stride, two ? (const uint32_t *)k1->ptr : nullptr,
                                 two ? (unsigned)k1->stride : 0u, two ? e1->ptr : nullptr, two ? (unsigned)e1->stride : 0u, batch, st);
    else if (ctx->shape.limb_bits == 64)
      e = launch_row_fwd_fma_u64(ctx->shape, ctx->tabs, x->format, (uint64_t *)out0, (uint64_t *)out1, x->ptr, (unsigned)x->stride,
                                 (const uint64_t *)k0->ptr, (unsigned)k0->stride, e0->ptr, (unsigned)e0->stride, two ? (const uint64_t *)k1->ptr : nullptr,
                                 two ? (unsigned)k1->stride : 0u, two ? e1->ptr : nullptr, two ? (unsigned)e1->stride : 0u, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "fwd_fma(rows)");
  }
  return fused_fwd_composed(ctx, out0, out1, x, k0, e0, k1, e1, batch, st);
}

int nflhip_fwd_fma_dev(nflhip_ctx *ctx, void *d_out, const nflhip_operand *x, const nflhip_operand *k, const nflhip_operand *e,
                       size_t batch, void *stream) {
  return fused_fwd_any(ctx, d_out, nullptr, x, k, e, nullptr, nullptr, batch, stream);
}

int nflhip_fwd_fma2_dev(nflhip_ctx *ctx, void *d_out0, void *d_out1, const nflhip_operand *x, const nflhip_operand *k0,
                        const nflhip_operand *e0, const nflhip_operand *k1, const nflhip_operand *e1, size_t batch, void *stream) {
  if (!k1 || !e1) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  return fused_fwd_any(ctx, d_out0, d_out1, x, k0, e0, k1, e1, batch, stream);
}

int nflhip_fma_inv_dev(nflhip_ctx *ctx, void *d_out, const nflhip_operand *a, const nflhip_operand *k, const nflhip_operand *b,
                       int subtract, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (batch && !d_out) return fail(ctx, NFLHIP_ERR_INVALID, "NULL result pointer");
  int rc = check_operand(ctx, a, batch, true, "a");
  if (!rc) rc = check_operand(ctx, k, batch, true, "k");
  if (!rc) rc = check_operand(ctx, b, batch, true, "b");
  if (rc) return rc;
  if (batch == 0) return NFLHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (ctx->shape.limb_bits == 64) {
    const void *xs[3] = {a->ptr, b->ptr, nullptr};
    const unsigned xstr[3] = {(unsigned)a->stride, (unsigned)b->stride, 0u};
    const int xf[3] = {0, 0, 0};
    const void *ks[2] = {k->ptr, nullptr};
    const unsigned kstr[2] = {(unsigned)k->stride, 0u};
    hipError_t e = launch_fused_asm_u64(ctx->shape, ctx->tabs, subtract ? 2 : 3, (uint64_t *)d_out, nullptr, xs, xstr, xf, ks, kstr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "fma_inv(fused)");
  }
  // short rows (1024 / 2048 words; 4096 for 32-bit limbs): the wave-per-row kernels multiply-subtract in the registers their
  // inverse transform starts from (kernels_wave.hip k_row_fma_inv)
  if (a->stride == 1 && b->stride == 1 && k->stride <= 1 && !ctx->cyclic) {
    hipError_t e = hipErrorNotSupported;
    if (ctx->shape.limb_bits == 32)
      e = launch_row_fma_inv_u32(ctx->shape, ctx->tabs, subtract, (uint32_t *)d_out, (const uint32_t *)a->ptr, (const uint32_t *)k->ptr,
                                 (int)k->stride, (const uint32_t *)b->ptr, batch, st);
    else if (ctx->shape.limb_bits == 64)
      e = launch_row_fma_inv_u64(ctx->shape, ctx->tabs, subtract, (uint64_t *)d_out, (const uint64_t *)a->ptr, (const uint64_t *)k->ptr,
                                 (int)k->stride, (const uint64_t *)b->ptr, batch, st);
    if (e == hipSuccess) return NFLHIP_OK;
    if (e != hipErrorNotSupported) return hipfail(ctx, e, "fma_inv(rows)");
  }
  // composed: one fused multiply-add / -subtract pass into the result, inverse transform in place
  const unsigned char prog[5] = {0, 1, 2, NFLHIP_EXPR_MUL, (unsigned char)(subtract ? NFLHIP_EXPR_SUB : NFLHIP_EXPR_ADD)};
  const void *ops[3] = {b->ptr, a->ptr, k->ptr};
  const unsigned sd[3] = {(unsigned)b->stride, (unsigned)a->stride, (unsigned)k->stride};
  rc = eval_dev(ctx, d_out, ops, 3, prog, sizeof(prog), batch, stream, sd, 1);
  if (rc) return rc;
  return nflhip_ntt_inv_dev(ctx, d_out, batch, stream);
}

int nflhip_has_fused_kernels(const nflhip_ctx *ctx) {
  if (!ctx || ctx->shape.compiled_only || ctx->cyclic || ctx->shape.nm > 65535) return 0;
  const Shape &s = ctx->shape;
  if (s.limb_bits == 64) return s.small_delta && s.logn >= 10 && s.logn <= 15;    // 1024 / 2048: the wave-per-row kernels; 4096 ... 32768: generated
  return s.limb_bits == 32 && s.logn >= 10 && s.logn <= 12;                       // the wave-per-row kernels
}

int nflhip_expand_small_dev(nflhip_ctx *ctx, void *d_data, const nflhip_operand *src, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (batch && !d_data) return fail(ctx, NFLHIP_ERR_INVALID, "NULL result pointer");
  int rc = check_operand(ctx, src, batch, false, "src");
  if (rc) return rc;
  hipError_t e = expand_any(ctx, d_data, src, batch, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(ctx, e, "expand_small");
  return NFLHIP_OK;
}

// the token of the next comparison on a slot (its mutex held): 1, 2, ... -- never the value the flag holds from an earlier call
static int next_cmp_token(nflhip_ctx *ctx, unsigned slot, hipStream_t st) {
  int &tok = ctx->cmp_token[slot];
  if (tok == 0x7fffffff) {   // wrap: clear the flag once, start over
    tok = 0;
    if (ctx->flag_host) ctx->tabs.flag[slot] = 0;
    else (void)hipMemsetAsync(ctx->tabs.flag + slot, 0, sizeof(int), st);
  }
  return ++tok;
}
static int read_cmp_flag(nflhip_ctx *ctx, unsigned slot, int token, hipStream_t st, int *hit) {
  int flag = 0;
  if (!ctx->flag_host) HIPCHK(ctx, hipMemcpyAsync(&flag, ctx->tabs.flag + slot, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  if (ctx->flag_host) flag = *(volatile int *)(ctx->tabs.flag + slot);
  *hit = flag == token ? 1 : 0;
  return NFLHIP_OK;
}

static int any_cmp_dev(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int want_eq, int *result, void *stream) {
  CHECK_CTX(ctx);
  if (!result || (batch && (!a || !b))) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned slot = ctx->cmp_next.fetch_add(1, std::memory_order_relaxed) % nflhip_ctx::kCmpSlots;
  std::lock_guard<std::mutex> lk(ctx->cmp_mu[slot]);  // held until the readback below has completed
  int *dflag = ctx->tabs.flag + slot;
  const int token = next_cmp_token(ctx, slot, st);
  hipError_t e = DISPATCH_T(
      ctx, launch_any_cmp<uint16_t>(ctx->shape, ctx->tabs, (const uint16_t *)a, (const uint16_t *)b, batch, want_eq, dflag, token, st),
      launch_any_cmp<uint32_t>(ctx->shape, ctx->tabs, (const uint32_t *)a, (const uint32_t *)b, batch, want_eq, dflag, token, st),
      launch_any_cmp<uint64_t>(ctx->shape, ctx->tabs, (const uint64_t *)a, (const uint64_t *)b, batch, want_eq, dflag, token, st));
  if (e != hipSuccess) return hipfail(ctx, e, "any_cmp");
  return read_cmp_flag(ctx, slot, token, st, result);
}
int nflhip_check_range_dev(nflhip_ctx *ctx, const void *d_data, size_t batch, int *bad, void *stream) {
  CHECK_CTX(ctx);
  if (!bad || (batch && !d_data)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned slot = ctx->cmp_next.fetch_add(1, std::memory_order_relaxed) % nflhip_ctx::kCmpSlots;
  std::lock_guard<std::mutex> lk(ctx->cmp_mu[slot]);  // held until the readback below has completed
  int *dflag = ctx->tabs.flag + slot;
  const int token = next_cmp_token(ctx, slot, st);
  hipError_t e = DISPATCH_T(ctx, launch_check_range<uint16_t>(ctx->shape, ctx->tabs, (const uint16_t *)d_data, batch, dflag, token, st),
                            launch_check_range<uint32_t>(ctx->shape, ctx->tabs, (const uint32_t *)d_data, batch, dflag, token, st),
                            launch_check_range<uint64_t>(ctx->shape, ctx->tabs, (const uint64_t *)d_data, batch, dflag, token, st));
  if (e != hipSuccess) return hipfail(ctx, e, "check_range");
  return read_cmp_flag(ctx, slot, token, st, bad);
}

int nflhip_check_range(const nflhip_ctx *ctx, const void *h_data, size_t batch, int *bad) {
  // host words against the host copy of the moduli: an assertion about the CALLER's data, nothing is computed
  if (!ctx) return fail(nullptr, NFLHIP_ERR_INVALID, "ctx is NULL");
  if (!bad || (batch && !h_data)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  const size_t n = ctx->shape.n, nm = ctx->shape.nm;
  int hit = 0;
  for (size_t r = 0; r < batch * nm && !hit; ++r) {
    const uint64_t p = ctx->h_P[r % nm];
    if (ctx->word == 8) { const uint64_t *w = (const uint64_t *)h_data + r * n; for (size_t i = 0; i < n; ++i) hit |= w[i] >= p; }
    else if (ctx->word == 4) { const uint32_t *w = (const uint32_t *)h_data + r * n; for (size_t i = 0; i < n; ++i) hit |= w[i] >= p; }
    else { const uint16_t *w = (const uint16_t *)h_data + r * n; for (size_t i = 0; i < n; ++i) hit |= w[i] >= p; }
  }
  *bad = hit ? 1 : 0;
  return NFLHIP_OK;
}

int nflhip_any_eq_dev(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int *result, void *stream) {
  return any_cmp_dev(ctx, a, b, batch, 1, result, stream);
}
int nflhip_any_neq_dev(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int *result, void *stream) {
  return any_cmp_dev(ctx, a, b, batch, 0, result, stream);
}

int nflhip_crt_lift_dev(nflhip_ctx *ctx, uint64_t *limbs, const void *d, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (batch && (!limbs || !d)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  if (ctx->tabs.qhat_w && batch) {
    // any number of moduli: the limb-serial kernel over a stream-ordered scratch (the unreduced sums)
    uint64_t *scr = nullptr;
    const size_t words = batch * ctx->shape.n * (size_t)ctx->tabs.crt_Lw;
    HIPCHK(ctx, hipMallocAsync((void **)&scr, words * sizeof(uint64_t), st));
    hipError_t we = DISPATCH_T(ctx, launch_crt_lift_wide<uint16_t>(ctx->shape, ctx->tabs, limbs, (const uint16_t *)d, batch, scr, st),
                               launch_crt_lift_wide<uint32_t>(ctx->shape, ctx->tabs, limbs, (const uint32_t *)d, batch, scr, st),
                               launch_crt_lift_wide<uint64_t>(ctx->shape, ctx->tabs, limbs, (const uint64_t *)d, batch, scr, st));
    (void)hipFreeAsync(scr, st);
    if (we != hipSuccess) return hipfail(ctx, we, "crt_lift (wide)");
    return NFLHIP_OK;
  }
  hipError_t e = DISPATCH_T(ctx, launch_crt_lift<uint16_t>(ctx->shape, ctx->tabs, limbs, (const uint16_t *)d, batch, st),
                            launch_crt_lift<uint32_t>(ctx->shape, ctx->tabs, limbs, (const uint32_t *)d, batch, st),
                            launch_crt_lift<uint64_t>(ctx->shape, ctx->tabs, limbs, (const uint64_t *)d, batch, st));
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "crt_lift: more than 32 moduli");
  if (e != hipSuccess) return hipfail(ctx, e, "crt_lift");
  return NFLHIP_OK;
}

int nflhip_crt_project_dev(nflhip_ctx *ctx, void *d, const uint64_t *limbs, size_t L_in, size_t batch, void *stream) {
  CHECK_CTX(ctx);
  if (batch && (!limbs || !d)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (L_in == 0 || L_in > (1u << 20)) return fail(ctx, NFLHIP_ERR_INVALID, "L_in out of range");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(ctx, launch_crt_project<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, limbs, L_in, batch, st),
                            launch_crt_project<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, limbs, L_in, batch, st),
                            launch_crt_project<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, limbs, L_in, batch, st));
  if (e != hipSuccess) return hipfail(ctx, e, "crt_project");
  return NFLHIP_OK;
}

int nflhip_fill_uniform_dev(nflhip_ctx *ctx, void *d, size_t first_poly, size_t batch, uint64_t seed, int operand,
                            void *stream) {
  CHECK_CTX(ctx);
  if (batch && !d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(ctx, launch_fill_uniform<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, first_poly, batch, seed, operand, st),
                            launch_fill_uniform<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, first_poly, batch, seed, operand, st),
                            launch_fill_uniform<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, first_poly, batch, seed, operand, st));
  if (e != hipSuccess) return hipfail(ctx, e, "fill_uniform");
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// samplers
// ---------------------------------------------------------------------------
struct nflhip_gauss {
  GaussTable tab;
  uint64_t *d_cdt = nullptr;
  int device = 0;
  int draw_bits = 64;   // keystream bits a sample normally consumes (nflhip_gauss_set_draw_bits): 64, or 32 = the narrow draw
  uint16_t *d_lut = nullptr;   // the narrow draw's bucket table (kernels_sample.hip gauss_bucket_table); NULL: table too long for LDS
};
static inline int gauss_narrow(const nflhip_gauss *g) { return g->draw_bits == 32 ? 1 : 0; }

int nflhip_random_words_dev(nflhip_ctx *ctx, uint64_t *d_out, uint64_t first_word, size_t nwords, const unsigned char *key,
                            uint64_t stream_id, void *stream) {
  CHECK_CTX(ctx);
  if (!key || (nwords && !d_out)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipError_t e = launch_random_words(d_out, first_word, nwords, key, stream_id, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(ctx, e, "random_words");
  return NFLHIP_OK;
}

int nflhip_sample_dev(nflhip_ctx *ctx, void *d, size_t first_poly, size_t batch, int dist, uint64_t p0, uint64_t p1,
                      const unsigned char *key, uint64_t stream_id, void *stream) {
  CHECK_CTX(ctx);
  if (!key || (batch && !d)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  const int dist_in = dist;
  dist &= ~(NFLHIP_DIST_REFERENCE_WORDS | NFLHIP_DIST_NARROW);
  if (dist < NFLHIP_DIST_UNIFORM || dist > NFLHIP_DIST_HWT) return fail(ctx, NFLHIP_ERR_INVALID, "unknown distribution");
  if ((dist_in & NFLHIP_DIST_NARROW) && dist != NFLHIP_DIST_UNIFORM)
    return fail(ctx, NFLHIP_ERR_INVALID, "the narrow draw applies to the uniform rule only (the others read one word per coefficient)");
  if (dist == NFLHIP_DIST_BOUNDED) {
    if (p0 == 0 || p0 >= (((uint64_t)1) << 62)) return fail(ctx, NFLHIP_ERR_INVALID, "upper_bound out of range");
    for (uint64_t p : ctx->h_P)  // core.hpp:205-210
      if (p0 >= p) return fail(ctx, NFLHIP_ERR_INVALID, "core: upper_bound is larger than the modulus");
    if (p1 == 0) return fail(ctx, NFLHIP_ERR_INVALID, "amplifier must be positive");
  }
  if (dist == NFLHIP_DIST_ZO && p0 > 255) return fail(ctx, NFLHIP_ERR_INVALID, "rho is a byte");
  if (dist == NFLHIP_DIST_HWT && (p0 == 0 || p0 > ctx->shape.n))  // assert at core.hpp:349
    return fail(ctx, NFLHIP_ERR_INVALID, "hamming weight must be in [1, degree]");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(
      ctx, launch_sample<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, first_poly, batch, dist_in, p0, p1, key, stream_id, st),
      launch_sample<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, first_poly, batch, dist_in, p0, p1, key, stream_id, st),
      launch_sample<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, first_poly, batch, dist_in, p0, p1, key, stream_id, st));
  if (e != hipSuccess) return hipfail(ctx, e, "sample");
  return NFLHIP_OK;
}

int nflhip_sample_seq_dev(nflhip_ctx *ctx, void *d, size_t batch, int dist, uint64_t p0, uint64_t p1, const unsigned char *key,
                          uint64_t first_stream_id, uint64_t stream_id_stride, void *stream) {
  // argument checks are those of nflhip_sample_dev (same messages): validate through it with an empty batch
  int rc = nflhip_sample_dev(ctx, d, 0, 0, dist, p0, p1, key, first_stream_id, stream);
  if (rc) return rc;
  if (batch && !d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = DISPATCH_T(
      ctx, launch_sample<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, 0, batch, dist, p0, p1, key, first_stream_id, st, 1, stream_id_stride),
      launch_sample<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, 0, batch, dist, p0, p1, key, first_stream_id, st, 1, stream_id_stride),
      launch_sample<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, 0, batch, dist, p0, p1, key, first_stream_id, st, 1, stream_id_stride));
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "sequence mode needs degree >= 8");
  if (e != hipSuccess) return hipfail(ctx, e, "sample_seq");
  return NFLHIP_OK;
}

int nflhip_sample_gauss_seq_dev(nflhip_ctx *ctx, void *d, size_t batch, const nflhip_gauss *g, uint64_t amplifier,
                                const unsigned char *key, uint64_t first_stream_id, uint64_t stream_id_stride, void *stream) {
  CHECK_CTX(ctx);
  if (!key || !g || (batch && !d)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (g->device != ctx->device) return fail(ctx, NFLHIP_ERR_INVALID, "gaussian table lives on another device");
  if (amplifier == 0) return fail(ctx, NFLHIP_ERR_INVALID, "amplifier must be positive");
  hipStream_t st = (hipStream_t)stream;
  const int w = g->tab.words, en = (int)g->tab.entries;
  const long long x0 = g->tab.x_min;
  hipError_t e = DISPATCH_T(
      ctx,
      launch_sample_gauss<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, 0, batch, g->d_cdt, w, en, x0, amplifier, key, first_stream_id, st, 1, stream_id_stride, gauss_narrow(g), g->d_lut),
      launch_sample_gauss<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, 0, batch, g->d_cdt, w, en, x0, amplifier, key, first_stream_id, st, 1, stream_id_stride, gauss_narrow(g), g->d_lut),
      launch_sample_gauss<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, 0, batch, g->d_cdt, w, en, x0, amplifier, key, first_stream_id, st, 1, stream_id_stride, gauss_narrow(g), g->d_lut));
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "sequence mode needs degree >= 8 (>= 16 under the 32-bit draw)");
  if (e != hipSuccess) return hipfail(ctx, e, "sample_gauss_seq");
  return NFLHIP_OK;
}

// The table of a parameter set is computed once per process (448-bit fixed point on the host: milliseconds) and shared by every context's
// generator: the header's FastGaussianNoise asks for it when it is CONSTRUCTED (nflhip_gauss_table with no output buffer) -- where the
// reference builds its MPFR table -- so that the first polynomial drawn from it does not carry the construction
static int cached_gauss_table(double sigma, unsigned security, unsigned samples, double center, GaussTable *out, std::string *err) {
  typedef std::tuple<double, unsigned, unsigned, double> Key;
  static std::mutex mu;
  static std::map<Key, std::shared_ptr<const GaussTable>> cache;
  const Key key(sigma, security, samples, center);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::shared_ptr<GaussTable> t = std::make_shared<GaussTable>();
    if (build_gauss_table(sigma, security, samples, center, t.get(), err)) return 1;
    if (cache.size() >= 64) cache.clear();
    it = cache.emplace(key, t).first;
  }
  *out = *it->second;
  return 0;
}

int nflhip_gauss_create(nflhip_ctx *ctx, nflhip_gauss **out, double sigma, unsigned security, unsigned samples,
                        double center) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  std::unique_ptr<nflhip_gauss> g(new (std::nothrow) nflhip_gauss());
  if (!g) return fail(ctx, NFLHIP_ERR_NOMEM, "out of host memory");
  std::string err;
  try {
    if (cached_gauss_table(sigma, security, samples, center, &g->tab, &err)) return fail(ctx, NFLHIP_ERR_INVALID, err);
  } catch (const std::bad_alloc &) {
    return fail(ctx, NFLHIP_ERR_NOMEM, "out of host memory while building the gaussian table");
  }
  g->device = ctx->device;
  const size_t bytes = g->tab.cdt.size() * sizeof(uint64_t);
  HIPCHK(ctx, hipMalloc((void **)&g->d_cdt, bytes));
  hipError_t e = hipMemcpy(g->d_cdt, g->tab.cdt.data(), bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(g->d_cdt);
    return hipfail(ctx, e, "gauss table upload");
  }
  const std::vector<uint16_t> lut = gauss_bucket_table(g->tab.cdt.data(), g->tab.words, g->tab.entries);
  if (!lut.empty()) {
    e = hipMalloc((void **)&g->d_lut, lut.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(g->d_lut, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(g->d_cdt);
      if (g->d_lut) (void)hipFree(g->d_lut);
      return hipfail(ctx, e, "gauss bucket table upload");
    }
  }
  *out = g.release();
  return NFLHIP_OK;
}

int nflhip_gauss_table(double sigma, unsigned security, unsigned samples, double center, long long *x_min, size_t *entries,
                       int *words, unsigned *bit_precision, double *tail, uint64_t *h_table, size_t cap_words) {
  GaussTable tab;
  std::string err;
  try {
    if (cached_gauss_table(sigma, security, samples, center, &tab, &err)) return fail(nullptr, NFLHIP_ERR_INVALID, err);
  } catch (const std::bad_alloc &) {
    return fail(nullptr, NFLHIP_ERR_NOMEM, "out of host memory while building the gaussian table");
  }
  if (x_min) *x_min = tab.x_min;
  if (entries) *entries = tab.entries;
  if (words) *words = tab.words;
  if (bit_precision) *bit_precision = tab.bit_precision;
  if (tail) *tail = tab.tail;
  if (h_table) {
    if (cap_words < tab.cdt.size()) return fail(nullptr, NFLHIP_ERR_INVALID, "output buffer too small");
    std::memcpy(h_table, tab.cdt.data(), tab.cdt.size() * sizeof(uint64_t));
  }
  return NFLHIP_OK;
}

int nflhip_gauss_set_draw_bits(nflhip_gauss *g, int bits) {
  if (!g || (bits != 64 && bits != 32)) return NFLHIP_ERR_INVALID;
  g->draw_bits = bits;
  return NFLHIP_OK;
}
int nflhip_gauss_draw_bits(const nflhip_gauss *g) { return g ? g->draw_bits : 0; }

int nflhip_gauss_destroy(nflhip_ctx *ctx, nflhip_gauss *g) {
  if (!g) return NFLHIP_OK;
  (void)ctx;  // never dereferenced: a FastGaussianNoise with static storage may outlive the context that built its table
  (void)hipSetDevice(g->device);
  if (g->d_cdt) (void)hipFree(g->d_cdt);
  if (g->d_lut) (void)hipFree(g->d_lut);
  delete g;
  return NFLHIP_OK;
}

int nflhip_gauss_info(const nflhip_gauss *g, long long *x_min, size_t *entries, int *words, unsigned *bit_precision,
                      double *tail, uint64_t *h_table) {
  if (!g) return NFLHIP_ERR_INVALID;
  if (x_min) *x_min = g->tab.x_min;
  if (entries) *entries = g->tab.entries;
  if (words) *words = g->tab.words;
  if (bit_precision) *bit_precision = g->tab.bit_precision;
  if (tail) *tail = g->tab.tail;
  if (h_table) std::memcpy(h_table, g->tab.cdt.data(), g->tab.cdt.size() * sizeof(uint64_t));
  return NFLHIP_OK;
}

int nflhip_sample_gauss_dev(nflhip_ctx *ctx, void *d, size_t first_poly, size_t batch, const nflhip_gauss *g,
                            uint64_t amplifier, const unsigned char *key, uint64_t stream_id, void *stream) {
  CHECK_CTX(ctx);
  if (!key || !g || (batch && !d)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (g->device != ctx->device) return fail(ctx, NFLHIP_ERR_INVALID, "gaussian table lives on another device");
  if (amplifier == 0) return fail(ctx, NFLHIP_ERR_INVALID, "amplifier must be positive");
  hipStream_t st = (hipStream_t)stream;
  const int w = g->tab.words, en = (int)g->tab.entries;
  const long long x0 = g->tab.x_min;
  hipError_t e = DISPATCH_T(
      ctx,
      launch_sample_gauss<uint16_t>(ctx->shape, ctx->tabs, (uint16_t *)d, first_poly, batch, g->d_cdt, w, en, x0, amplifier, key, stream_id, st, 0, 0, gauss_narrow(g), g->d_lut),
      launch_sample_gauss<uint32_t>(ctx->shape, ctx->tabs, (uint32_t *)d, first_poly, batch, g->d_cdt, w, en, x0, amplifier, key, stream_id, st, 0, 0, gauss_narrow(g), g->d_lut),
      launch_sample_gauss<uint64_t>(ctx->shape, ctx->tabs, (uint64_t *)d, first_poly, batch, g->d_cdt, w, en, x0, amplifier, key, stream_id, st, 0, 0, gauss_narrow(g), g->d_lut));
  if (e != hipSuccess) return hipfail(ctx, e, "sample_gauss");
  return NFLHIP_OK;
}

static int gauss_small_any(nflhip_ctx *ctx, void *d_out, int format, size_t first_poly, size_t batch, const nflhip_gauss *g,
                           uint64_t amplifier, const unsigned char *key, uint64_t stream_id, void *stream, int seq_on,
                           uint64_t seq_stride) {
  CHECK_CTX(ctx);
  if (!key || !g || (batch && !d_out)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (g->device != ctx->device) return fail(ctx, NFLHIP_ERR_INVALID, "gaussian table lives on another device");
  if (amplifier == 0) return fail(ctx, NFLHIP_ERR_INVALID, "amplifier must be positive");
  if (format < NFLHIP_FMT_I8 || format > NFLHIP_FMT_I32) return fail(ctx, NFLHIP_ERR_INVALID, "the compact format is int8, int16 or int32");
  // every sample x * amplifier must fit the format and stay below every modulus (so that p + v is its residue)
  const long long lo = g->tab.x_min, hi = g->tab.x_min + (long long)g->tab.entries - 1;
  const uint64_t mag = (uint64_t)std::max(lo < 0 ? -lo : lo, hi < 0 ? -hi : hi);
  const uint64_t cap = format == NFLHIP_FMT_I8 ? 127u : format == NFLHIP_FMT_I16 ? 32767u : 2147483647u;
  if (amplifier > cap || mag > cap / amplifier) return fail(ctx, NFLHIP_ERR_INVALID, "the samples do not fit the compact format");
  for (uint64_t p : ctx->h_P)
    if (mag * amplifier >= p) return fail(ctx, NFLHIP_ERR_INVALID, "the samples are not below the modulus");
  hipError_t e = launch_gauss_small(ctx->shape, d_out, format, first_poly, batch, g->d_cdt, g->tab.words, (int)g->tab.entries,
                                    g->tab.x_min, amplifier, key, stream_id, (hipStream_t)stream, seq_on, seq_stride, gauss_narrow(g), g->d_lut);
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "sequence mode needs degree >= 8 (>= 16 under the 32-bit draw)");
  if (e != hipSuccess) return hipfail(ctx, e, "sample_gauss_small");
  return NFLHIP_OK;
}

int nflhip_sample_gauss_small_dev(nflhip_ctx *ctx, void *d_out, int format, size_t first_poly, size_t batch,
                                  const nflhip_gauss *g, uint64_t amplifier, const unsigned char *key, uint64_t stream_id,
                                  void *stream) {
  return gauss_small_any(ctx, d_out, format, first_poly, batch, g, amplifier, key, stream_id, stream, 0, 0);
}

int nflhip_sample_gauss_small_seq_dev(nflhip_ctx *ctx, void *d_out, int format, size_t batch, const nflhip_gauss *g,
                                      uint64_t amplifier, const unsigned char *key, uint64_t first_stream_id,
                                      uint64_t stream_id_stride, void *stream) {
  return gauss_small_any(ctx, d_out, format, 0, batch, g, amplifier, key, first_stream_id, stream, 1, stream_id_stride);
}

int nflhip_sample_gauss_small_multi_dev(nflhip_ctx *ctx, void *const *d_out, size_t count, int format, size_t batch, const nflhip_gauss *g,
                                        const uint64_t *amplifier, const unsigned char *key, const uint64_t *stream_id,
                                        const uint64_t *stream_id_stride, void *stream) {
  CHECK_CTX(ctx);
  if (!key || !g || !d_out || !amplifier || !stream_id) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (count < 1 || count > 4) return fail(ctx, NFLHIP_ERR_INVALID, "one to four draws per call");
  if (g->device != ctx->device) return fail(ctx, NFLHIP_ERR_INVALID, "gaussian table lives on another device");
  if (format < NFLHIP_FMT_I8 || format > NFLHIP_FMT_I32) return fail(ctx, NFLHIP_ERR_INVALID, "the compact format is int8, int16 or int32");
  const long long lo = g->tab.x_min, hi = g->tab.x_min + (long long)g->tab.entries - 1;
  const uint64_t mag = (uint64_t)std::max(lo < 0 ? -lo : lo, hi < 0 ? -hi : hi);
  const uint64_t cap = format == NFLHIP_FMT_I8 ? 127u : format == NFLHIP_FMT_I16 ? 32767u : 2147483647u;
  for (size_t j = 0; j < count; ++j) {   // (the checks of gauss_small_any, per draw)
    if (batch && !d_out[j]) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
    if (amplifier[j] == 0) return fail(ctx, NFLHIP_ERR_INVALID, "amplifier must be positive");
    if (amplifier[j] > cap || mag > cap / amplifier[j]) return fail(ctx, NFLHIP_ERR_INVALID, "the samples do not fit the compact format");
    for (uint64_t p : ctx->h_P)
      if (mag * amplifier[j] >= p) return fail(ctx, NFLHIP_ERR_INVALID, "the samples are not below the modulus");
  }
  hipError_t e = launch_gauss_small_multi(ctx->shape, d_out, count, format, batch, g->d_cdt, g->tab.words, (int)g->tab.entries, g->tab.x_min,
                                          amplifier, key, stream_id, stream_id_stride, (hipStream_t)stream, gauss_narrow(g), g->d_lut);
  if (e == hipErrorNotSupported) return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "sequence mode needs degree >= 8 (>= 16 under the 32-bit draw)");
  if (e != hipSuccess) return hipfail(ctx, e, "sample_gauss_small_multi");
  return NFLHIP_OK;
}

int nflhip_gauss_noise_dev(nflhip_ctx *ctx, int64_t *d_out, uint64_t first_sample, size_t count, const nflhip_gauss *g,
                           const unsigned char *key, uint64_t stream_id, void *stream) {
  CHECK_CTX(ctx);
  if (!key || !g || (count && !d_out)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (g->device != ctx->device) return fail(ctx, NFLHIP_ERR_INVALID, "gaussian table lives on another device");
  hipError_t e = launch_gauss_noise((long long *)d_out, first_sample, count, g->d_cdt, g->tab.words, (int)g->tab.entries,
                                    g->tab.x_min, key, stream_id, (hipStream_t)stream, gauss_narrow(g));
  if (e != hipSuccess) return hipfail(ctx, e, "gauss_noise");
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// core::ntt / core::inv_ntt: the cyclic transform of single rows
// ---------------------------------------------------------------------------
static int row_child(nflhip_ctx *ctx, size_t cm, int inverse_tables, nflhip_ctx **out) {
  std::lock_guard<std::mutex> lk(ctx->row_mu);
  if (ctx->row_ctx.empty()) ctx->row_ctx.assign(2 * ctx->shape.nm, nullptr);
  nflhip_ctx *&slot = ctx->row_ctx[2 * cm + (inverse_tables ? 1 : 0)];
  if (!slot) {
    // one word of each parameter table, in the limb type's own width
    uint64_t P8 = ctx->h_P[cm], R8 = ctx->h_roots[cm], I8 = ctx->h_invk[cm];
    uint32_t P4 = (uint32_t)P8, R4 = (uint32_t)R8, I4 = (uint32_t)I8;
    uint16_t P2 = (uint16_t)P8, R2 = (uint16_t)R8, I2 = (uint16_t)I8;
    const void *pp = ctx->word == 8 ? (const void *)&P8 : ctx->word == 4 ? (const void *)&P4 : (const void *)&P2;
    const void *rr = ctx->word == 8 ? (const void *)&R8 : ctx->word == 4 ? (const void *)&R4 : (const void *)&R2;
    const void *ii = ctx->word == 8 ? (const void *)&I8 : ctx->word == 4 ? (const void *)&I4 : (const void *)&I2;
    int rc = ctx_create_mode(&slot, ctx->device, ctx->shape.limb_bits, ctx->shape.n, 1, pp, rr, ii, ctx->kmax_log2,
                             inverse_tables ? 2 : 1);
    if (rc) return rc;
  }
  *out = slot;
  return NFLHIP_OK;
}

int nflhip_ntt_row_dev(nflhip_ctx *ctx, void *d_rows, size_t cm, int mode, size_t rows, void *stream) {
  CHECK_CTX(ctx);
  if (cm >= ctx->shape.nm) return fail(ctx, NFLHIP_ERR_INVALID, "modulus index out of range");
  if (mode < 0 || mode > 3) return fail(ctx, NFLHIP_ERR_INVALID, "unknown row-transform mode");
  if (rows == 0) return NFLHIP_OK;
  if (!d_rows) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  nflhip_ctx *child = nullptr;
  int rc = row_child(ctx, cm, mode & NFLHIP_ROW_INVERSE_TABLES, &child);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (mode & NFLHIP_ROW_BITREV_IO) {  // core::inv_ntt: permut, ntt, permut (core.hpp:549-554)
    hipError_t e = DISPATCH_T(ctx, launch_bitrev_rows<uint16_t>(ctx->shape, (uint16_t *)d_rows, rows, st),
                              launch_bitrev_rows<uint32_t>(ctx->shape, (uint32_t *)d_rows, rows, st),
                              launch_bitrev_rows<uint64_t>(ctx->shape, (uint64_t *)d_rows, rows, st));
    if (e != hipSuccess) return hipfail(ctx, e, "ntt_row: bit reversal");
  }
  rc = nflhip_ntt_fwd_dev(child, d_rows, rows, stream);
  if (rc) return rc;
  if (mode & NFLHIP_ROW_BITREV_IO) {
    hipError_t e = DISPATCH_T(ctx, launch_bitrev_rows<uint16_t>(ctx->shape, (uint16_t *)d_rows, rows, st),
                              launch_bitrev_rows<uint32_t>(ctx->shape, (uint32_t *)d_rows, rows, st),
                              launch_bitrev_rows<uint64_t>(ctx->shape, (uint64_t *)d_rows, rows, st));
    if (e != hipSuccess) return hipfail(ctx, e, "ntt_row: bit reversal");
  }
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------
int nflhip_stream_create(nflhip_ctx *ctx, void **stream) {
  CHECK_CTX(ctx);
  if (!stream) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipStream_t s = nullptr;
  HIPCHK(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return NFLHIP_OK;
}
int nflhip_stream_destroy(nflhip_ctx *ctx, void *stream) {
  CHECK_CTX(ctx);
  if (stream) HIPCHK(ctx, hipStreamDestroy((hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_memcpy_d2d(nflhip_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream) {
  CHECK_CTX(ctx);
  HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_memset_dev(nflhip_ctx *ctx, void *d_dst, int byte, size_t bytes, void *stream) {
  CHECK_CTX(ctx);
  HIPCHK(ctx, hipMemsetAsync(d_dst, byte, bytes, (hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_broadcast_dev(nflhip_ctx *ctx, void *d_dst, const void *d_one, size_t count, void *stream) {
  CHECK_CTX(ctx);
  if (count && (!d_dst || !d_one)) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipError_t e = launch_broadcast(d_dst, d_one, poly_bytes(ctx, 1), count, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(ctx, e, "broadcast");
  return NFLHIP_OK;
}
int nflhip_random_bytes(int device, unsigned char *h_out, size_t nbytes, const unsigned char *key, uint64_t stream_id) {
  if (nbytes == 0) return NFLHIP_OK;
  if (!h_out || !key) return fail(nullptr, NFLHIP_ERR_INVALID, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, NFLHIP_ERR_NO_DEVICE, "no HIP device available (this engine has no CPU fallback)");
  HIPCHK(nullptr, hipSetDevice(device));
  const size_t nwords = (nbytes + 7) / 8;
  uint64_t *d = nullptr;
  HIPCHK(nullptr, hipMalloc((void **)&d, nwords * 8));
  hipError_t e = launch_random_words(d, 0, nwords, key, stream_id, nullptr);
  std::vector<uint64_t> tmp;
  if (e == hipSuccess && (nbytes & 7)) {
    tmp.resize(nwords);
    e = hipMemcpy(tmp.data(), d, nwords * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) memcpy(h_out, tmp.data(), nbytes);
  } else if (e == hipSuccess) {
    e = hipMemcpy(h_out, d, nbytes, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return hipfail(nullptr, e, "random_bytes");
  return NFLHIP_OK;
}

int nflhip_malloc(nflhip_ctx *ctx, void **p, size_t bytes) {
  CHECK_CTX(ctx);
  if (!p) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  hipError_t e = hipMalloc(p, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return fail(ctx, NFLHIP_ERR_NOMEM, "hipMalloc: out of device memory");
  if (e != hipSuccess) return hipfail(ctx, e, "hipMalloc");
  return NFLHIP_OK;
}
int nflhip_free(nflhip_ctx *ctx, void *p) {
  CHECK_CTX(ctx);
  if (p) HIPCHK(ctx, hipFree(p));
  return NFLHIP_OK;
}
int nflhip_memcpy_h2d(nflhip_ctx *ctx, void *d, const void *h, size_t bytes, void *stream) {
  CHECK_CTX(ctx);
  HIPCHK(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_memcpy_d2h(nflhip_ctx *ctx, void *h, const void *d, size_t bytes, void *stream) {
  CHECK_CTX(ctx);
  HIPCHK(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_stream_sync(nflhip_ctx *ctx, void *stream) {
  CHECK_CTX(ctx);
  HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
  return NFLHIP_OK;
}
int nflhip_stream_idle(nflhip_ctx *ctx, void *stream, int *idle) {
  CHECK_CTX(ctx);
  if (!idle) return fail(ctx, NFLHIP_ERR_INVALID, "idle is NULL");
  const hipError_t e = hipStreamQuery((hipStream_t)stream);
  if (e == hipErrorNotReady) (void)hipGetLastError();  // "not ready" is an answer, not an error: do not leave it for the next launch's check
  else if (e != hipSuccess) HIPCHK(ctx, e);
  *idle = e == hipSuccess ? 1 : 0;
  return NFLHIP_OK;
}

// ---------------------------------------------------------------------------
// host-pointer entry points: stage through context-owned device buffers
// ---------------------------------------------------------------------------
}  // extern "C"

// Large batches through the host-pointer entry points (what an unchanged caller holding arrays of inline-storage
// nfl::poly gets: poly.hpp:87-88, tests/tools.h:6-17).  hipMemcpy from pageable memory tops out at ~12 GB/s on this
// platform (the runtime's single staging thread), 5x below PCIe.  Here the batch is cut into chunks that flow through
// three slots of PINNED staging buffers: several host threads copy chunk k + 1 into its slot while chunk k crosses PCIe
// (H2D stream), chunk k - 1 is computed (compute stream) and chunk k - 2 returns (D2H stream) and is copied out.
// Results are what one call over the whole batch gives (every operation here is per-polynomial).
namespace {
class CopyPool {  // a few host threads that memcpy slices; process-wide, started on first use
 public:
  static CopyPool &get() {
    static CopyPool *p = new CopyPool();  // (leaked on purpose: worker threads must not be joined from a static destructor)
    return *p;
  }
  void copy(void *dst, const void *src, size_t bytes) {
    const size_t slice = 512 << 10;
    const size_t parts = (bytes + slice - 1) / slice;
    if (parts <= 1 || workers_.empty()) {
      std::memcpy(dst, src, bytes);
      return;
    }
    // ONE job at a time: the pool is process-wide and keeps a single job's state, while callers on different contexts
    // (one host thread per GPU, two ring types) hold only their own context's lock
    std::lock_guard<std::mutex> call(call_mu_);
    std::unique_lock<std::mutex> lk(mu_);
    dst_ = (char *)dst;
    src_ = (const char *)src;
    bytes_ = bytes;
    slice_ = slice;
    next_ = 0;
    parts_ = parts;
    done_ = 0;
    ++generation_;
    gen_hint_.store(generation_, std::memory_order_release);
    cv_.notify_all();
    lk.unlock();
    work();  // the calling thread copies too
    lk.lock();
    cv_done_.wait(lk, [&] { return done_ == parts_; });
  }

 private:
  CopyPool() {
    unsigned n = std::thread::hardware_concurrency();
    n = n >= 64 ? 15 : (n > 16 ? 7 : (n > 2 ? n / 2 - 1 : 0));  // + the caller: 16 copying threads on a server host
    for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); }), workers_.back().detach();
  }
  void work() {
    for (;;) {
      size_t k;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (next_ >= parts_) return;
        k = next_++;
      }
      const size_t off = k * slice_, len = bytes_ - off < slice_ ? bytes_ - off : slice_;
      std::memcpy(dst_ + off, src_ + off, len);
      std::lock_guard<std::mutex> lk(mu_);
      if (++done_ == parts_) cv_done_.notify_all();
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      // a chunk is ~0.3 ms of copying for one thread: a sleeping worker wakes too late to help, so workers spin for a
      // while after every job (the next chunk follows within microseconds while a call is in flight) and only then sleep
      bool got = false;
      for (int spin = 0; spin < 20000 && !got; ++spin) {
        if (gen_hint_.load(std::memory_order_acquire) != seen) got = true;
        else __builtin_ia32_pause();
      }
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (!got) cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
    }
  }
  std::atomic<unsigned long long> gen_hint_{0};
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, cv_done_;
  std::vector<std::thread> workers_;
  char *dst_ = nullptr;
  const char *src_ = nullptr;
  size_t bytes_ = 0, slice_ = 0, next_ = 0, parts_ = 0, done_ = 0;
  unsigned long long generation_ = 0;
};
}  // namespace

struct HostPipe {
  static constexpr int kSlots = 3, kBufs = 4;            // per slot: up to 3 inputs + 1 output
  static constexpr size_t kChunkBytes = size_t(8) << 20;  // per operand and slot
  void *pinned[kSlots][kBufs] = {};
  void *dev[kSlots][kBufs] = {};
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  hipEvent_t ev_h2d[kSlots] = {}, ev_k[kSlots] = {}, ev_d2h[kSlots] = {};
  double t_in = 0, t_out = 0, t_wait = 0, t_total = 0;   // seconds spent copying in / out, waiting for the device, in calls
  ~HostPipe() {
    for (int s = 0; s < kSlots; ++s) {
      for (int b = 0; b < kBufs; ++b) {
        if (pinned[s][b]) (void)hipHostFree(pinned[s][b]);
        if (dev[s][b]) (void)hipFree(dev[s][b]);
      }
      if (ev_h2d[s]) (void)hipEventDestroy(ev_h2d[s]);
      if (ev_k[s]) (void)hipEventDestroy(ev_k[s]);
      if (ev_d2h[s]) (void)hipEventDestroy(ev_d2h[s]);
    }
    if (s_h2d) (void)hipStreamDestroy(s_h2d);
    if (s_d2h) (void)hipStreamDestroy(s_d2h);
  }
};

static void pipe_stats(const nflhip_ctx *ctx, double out[4]) {
  const HostPipe *p = ctx ? ctx->pipe : nullptr;
  out[0] = p ? p->t_in : 0;
  out[1] = p ? p->t_out : 0;
  out[2] = p ? p->t_wait : 0;
  out[3] = p ? p->t_total : 0;
}
static void pipe_destroy(nflhip_ctx *ctx) {
  delete ctx->pipe;
  ctx->pipe = nullptr;
}

static int pipe_get(nflhip_ctx *ctx, HostPipe **out) {
  if (!ctx->pipe) {
    std::unique_ptr<HostPipe> p(new (std::nothrow) HostPipe());
    if (!p) return fail(ctx, NFLHIP_ERR_NOMEM, "out of host memory");
    HIPCHK(ctx, hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
    for (int s = 0; s < HostPipe::kSlots; ++s) {
      for (int b = 0; b < HostPipe::kBufs; ++b) {
        HIPCHK(ctx, hipHostMalloc(&p->pinned[s][b], HostPipe::kChunkBytes, hipHostMallocDefault));
        HIPCHK(ctx, hipMalloc(&p->dev[s][b], HostPipe::kChunkBytes));
      }
      HIPCHK(ctx, hipEventCreateWithFlags(&p->ev_h2d[s], hipEventDisableTiming));
      HIPCHK(ctx, hipEventCreateWithFlags(&p->ev_k[s], hipEventDisableTiming));
      HIPCHK(ctx, hipEventCreateWithFlags(&p->ev_d2h[s], hipEventDisableTiming));
    }
    ctx->pipe = p.release();
  }
  *out = ctx->pipe;
  return NFLHIP_OK;
}

// in[j] (nin <= 3 host arrays of `batch` polynomials) -> out (host array); launch(d_in[], d_out, count, stream) enqueues
// the per-polynomial operation on a chunk.  Caller holds ctx->mu.  Returns NFLHIP_ERR_UNSUPPORTED when the batch is too
// small to pipeline (the simple staged path then serves it).
template <typename F>
static int run_pipelined(nflhip_ctx *ctx, size_t batch, const void *const *in, int nin, void *out, F launch) {
  const size_t pb = poly_bytes(ctx, 1);
  const size_t per = HostPipe::kChunkBytes / pb;  // polynomials per chunk
  if (per == 0 || batch < 2 * per) return NFLHIP_ERR_UNSUPPORTED;
  HostPipe *p = nullptr;
  int rc = pipe_get(ctx, &p);
  if (rc) return rc;
  CopyPool &pool = CopyPool::get();
  const size_t nchunks = (batch + per - 1) / per;
  auto count_of = [&](size_t k) { return k + 1 < nchunks ? per : batch - k * per; };
  typedef std::chrono::steady_clock clk;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const clk::time_point t_begin = clk::now();
  // chunk k is back in its pinned slot: hand it to the caller.  Done by a SECOND host thread, so that results leave while
  // the calling thread (and the pool) copies the next chunks in: the two directions overlap on the host as they do on PCIe
  std::atomic<size_t> issued{0}, drained{0};
  std::atomic<int> drain_rc{NFLHIP_OK};
  std::atomic<bool> stop{false};
  std::string drain_err;
  auto drain_one = [&](size_t k) -> int {
    const int s = int(k % HostPipe::kSlots);
    const clk::time_point t0 = clk::now();
    hipError_t he = hipEventSynchronize(p->ev_d2h[s]);
    if (he != hipSuccess) {
      drain_err = std::string("hipEventSynchronize: ") + hipGetErrorString(he);
      return NFLHIP_ERR_HIP;
    }
    const clk::time_point t1 = clk::now();
    std::memcpy((char *)out + k * per * pb, p->pinned[s][3], count_of(k) * pb);
    p->t_wait += secs(t0, t1);
    p->t_out += secs(t1, clk::now());
    return NFLHIP_OK;
  };
  std::thread drainer;
  try {
    drainer = std::thread([&] {
      (void)hipSetDevice(ctx->device);
      for (size_t k = 0; k < nchunks; ++k) {
        while (issued.load(std::memory_order_acquire) <= k) {
          if (stop.load(std::memory_order_acquire)) return;
          __builtin_ia32_pause();
        }
        const int r = drain_one(k);
        if (r) { drain_rc.store(r); return; }
        drained.store(k + 1, std::memory_order_release);
      }
    });
  } catch (...) {  // (no exception crosses the C boundary)
    return fail(ctx, NFLHIP_ERR_NOMEM, "cannot start the host thread that copies results out");
  }
  // Whatever way this function is left: the drainer is joined, and -- on an error path, where copies and kernels may still
  // be in flight on the three streams against the pinned and device slots -- the streams are drained before the slots
  // can be reused by the next call on this context
  bool completed = false;
  struct joiner {
    std::thread &t; std::atomic<bool> &stop; bool &completed; HostPipe *p; nflhip_ctx *ctx;
    ~joiner() {
      stop.store(true);
      if (t.joinable()) t.join();
      if (!completed) {
        (void)hipStreamSynchronize(p->s_h2d);
        (void)hipStreamSynchronize(ctx->hstream);
        (void)hipStreamSynchronize(p->s_d2h);
        (void)hipGetLastError();
      }
    }
  } join_guard{drainer, stop, completed, p, ctx};
  for (size_t k = 0; k < nchunks; ++k) {
    const int s = int(k % HostPipe::kSlots);
    while (k >= size_t(HostPipe::kSlots) && drained.load(std::memory_order_acquire) + HostPipe::kSlots <= k) {  // the slot's previous tenant
      if (drain_rc.load()) return fail(ctx, drain_rc.load(), drain_err);
      __builtin_ia32_pause();
    }
    const size_t cnt = count_of(k), bytes = cnt * pb;
    const void *d_in[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < nin; ++j) {
      // (aliased operands -- polymul(a, a) -- are staged once)
      int same = -1;
      for (int i = 0; i < j; ++i)
        if (in[i] == in[j]) same = i;
      if (same >= 0) { d_in[j] = d_in[same]; continue; }
      const clk::time_point t0 = clk::now();
      pool.copy(p->pinned[s][j], (const char *)in[j] + k * per * pb, bytes);
      p->t_in += secs(t0, clk::now());
      HIPCHK(ctx, hipMemcpyAsync(p->dev[s][j], p->pinned[s][j], bytes, hipMemcpyHostToDevice, p->s_h2d));
      d_in[j] = p->dev[s][j];
    }
    HIPCHK(ctx, hipEventRecord(p->ev_h2d[s], p->s_h2d));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->hstream, p->ev_h2d[s], 0));
    rc = launch(d_in, p->dev[s][3], cnt, (void *)ctx->hstream);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(p->ev_k[s], ctx->hstream));
    HIPCHK(ctx, hipStreamWaitEvent(p->s_d2h, p->ev_k[s], 0));
    HIPCHK(ctx, hipMemcpyAsync(p->pinned[s][3], p->dev[s][3], bytes, hipMemcpyDeviceToHost, p->s_d2h));
    HIPCHK(ctx, hipEventRecord(p->ev_d2h[s], p->s_d2h));
    // (the next H2D into this slot's device inputs cannot overtake this chunk's kernel: the host reuses a slot only after
    // its result has been drained.  No wait on the in-order H2D stream here -- it would hold chunk k + 1's copy, which
    // goes to ANOTHER slot, behind kernel k, and copies would never overlap compute)
    issued.store(k + 1, std::memory_order_release);
  }
  while (drained.load(std::memory_order_acquire) < nchunks) {
    if (drain_rc.load()) return fail(ctx, drain_rc.load(), drain_err);
    __builtin_ia32_pause();
  }
  p->t_total += secs(t_begin, clk::now());
  completed = true;
  return NFLHIP_OK;
}

extern "C" {

struct Staged {
  nflhip_ctx *ctx;
  std::unique_lock<std::mutex> lk;
  explicit Staged(nflhip_ctx *c) : ctx(c), lk(c->mu) {}
  int in(int slot, const void *h, size_t bytes) {
    int rc = ensure_stage(ctx, slot, bytes);
    if (rc) return rc;
    if (h && ctx->stage_host[slot]) std::memcpy(ctx->stage[slot], h, bytes);   // (the stream is idle: every host-pointer call ends synchronised)
    else if (h) HIPCHK(ctx, hipMemcpyAsync(ctx->stage[slot], h, bytes, hipMemcpyHostToDevice, ctx->hstream));
    return NFLHIP_OK;
  }
  int out(void *h, int slot, size_t bytes) {
    if (!ctx->stage_host[slot]) HIPCHK(ctx, hipMemcpyAsync(h, ctx->stage[slot], bytes, hipMemcpyDeviceToHost, ctx->hstream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->hstream));
    if (ctx->stage_host[slot]) std::memcpy(h, ctx->stage[slot], bytes);
    return NFLHIP_OK;
  }
};

int nflhip_ntt_fwd(nflhip_ctx *ctx, void *h, size_t batch) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!h) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const void *ins[1] = {h};
  int rc = run_pipelined(ctx, batch, ins, 1, h, [&](const void *const *d, void *o, size_t cnt, void *st) {
    HIPCHK(ctx, hipMemcpyAsync(o, d[0], poly_bytes(ctx, cnt), hipMemcpyDeviceToDevice, (hipStream_t)st));
    return nflhip_ntt_fwd_dev(ctx, o, cnt, st);
  });
  if (rc != NFLHIP_ERR_UNSUPPORTED) return rc;
  rc = s.in(0, h, bytes);
  if (rc) return rc;
  rc = nflhip_ntt_fwd_dev(ctx, ctx->stage[0], batch, ctx->hstream);
  if (rc) return rc;
  return s.out(h, 0, bytes);
}
int nflhip_ntt_inv(nflhip_ctx *ctx, void *h, size_t batch) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!h) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const void *ins[1] = {h};
  int rc = run_pipelined(ctx, batch, ins, 1, h, [&](const void *const *d, void *o, size_t cnt, void *st) {
    HIPCHK(ctx, hipMemcpyAsync(o, d[0], poly_bytes(ctx, cnt), hipMemcpyDeviceToDevice, (hipStream_t)st));
    return nflhip_ntt_inv_dev(ctx, o, cnt, st);
  });
  if (rc != NFLHIP_ERR_UNSUPPORTED) return rc;
  rc = s.in(0, h, bytes);
  if (rc) return rc;
  rc = nflhip_ntt_inv_dev(ctx, ctx->stage[0], batch, ctx->hstream);
  if (rc) return rc;
  return s.out(h, 0, bytes);
}
int nflhip_ntt_row(nflhip_ctx *ctx, void *h_rows, size_t cm, int mode, size_t rows) {
  CHECK_CTX(ctx);
  if (rows == 0) return NFLHIP_OK;
  if (!h_rows) return fail(ctx, NFLHIP_ERR_INVALID, "NULL data pointer");
  Staged s(ctx);
  const size_t bytes = rows * ctx->shape.n * ctx->word;
  int rc = s.in(0, h_rows, bytes);
  if (rc) return rc;
  rc = nflhip_ntt_row_dev(ctx, ctx->stage[0], cm, mode, rows, ctx->hstream);
  if (rc) return rc;
  return s.out(h_rows, 0, bytes);
}
int nflhip_pointwise(nflhip_ctx *ctx, int op, void *o, const void *a, const void *b, const void *bp, size_t batch) {
  CHECK_CTX(ctx);
  if (op < 0 || op > 4) return fail(ctx, NFLHIP_ERR_INVALID, "unknown element-wise op");
  if (batch == 0) return NFLHIP_OK;
  if (!o || !a || (op != NFLHIP_OP_COMPUTE_SHOUP && !b) || (op == NFLHIP_OP_MUL_SHOUP && !bp))
    return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const void *ins[3] = {a, op != NFLHIP_OP_COMPUTE_SHOUP ? b : a, op == NFLHIP_OP_MUL_SHOUP ? bp : a};
  int rc = run_pipelined(ctx, batch, ins, 3, o, [&](const void *const *d, void *out, size_t cnt, void *st) {
    return nflhip_pointwise_dev(ctx, op, out, d[0], d[1], d[2], cnt, st);
  });
  if (rc != NFLHIP_ERR_UNSUPPORTED) return rc;
  rc = s.in(0, a, bytes);
  if (rc) return rc;
  if (op != NFLHIP_OP_COMPUTE_SHOUP && (rc = s.in(1, b, bytes))) return rc;
  if (op == NFLHIP_OP_MUL_SHOUP && (rc = s.in(2, bp, bytes))) return rc;
  rc = nflhip_pointwise_dev(ctx, op, ctx->stage[0], ctx->stage[0], ctx->stage[1], ctx->stage[2], batch, ctx->hstream);
  if (rc) return rc;
  return s.out(o, 0, bytes);
}
int nflhip_eval(nflhip_ctx *ctx, void *h_out, const void *const *h_operands, size_t noperands, const unsigned char *program,
                size_t proglen, size_t batch) {
  CHECK_CTX(ctx);
  if (!program || !h_operands) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (noperands == 0 || noperands > 4 || proglen == 0 || proglen > NFLHIP_EXPR_MAX_LEN)
    return fail(ctx, NFLHIP_ERR_UNSUPPORTED, "host-pointer eval takes at most 4 distinct operands");
  if (batch == 0) return NFLHIP_OK;
  if (!h_out) return fail(ctx, NFLHIP_ERR_INVALID, "NULL output");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  for (size_t i = 0; i < noperands; ++i)
    if (!h_operands[i]) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  if (noperands <= 3) {   // (the pipeline's slots hold three inputs and the result)
    int prc = run_pipelined(ctx, batch, h_operands, (int)noperands, h_out, [&](const void *const *d, void *out, size_t cnt, void *st) {
      return eval_dev(ctx, out, d, noperands, program, proglen, cnt, st);
    });
    if (prc != NFLHIP_ERR_UNSUPPORTED) return prc;
  }
  const void *dops[4] = {nullptr, nullptr, nullptr, nullptr};
  for (size_t i = 0; i < noperands; ++i) {
    if (!h_operands[i]) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
    int rc = s.in((int)i, h_operands[i], bytes);
    if (rc) return rc;
    dops[i] = ctx->stage[i];
  }
  // four operands (c = c + shoup(a * b, b'): the reference's FMA with a precomputed companion) fill the four staging buffers: the result
  // is written over the first one -- the evaluation is element-wise, `out` may alias an input
  const int oslot = noperands == 4 ? 0 : 3;
  int rc = s.in(oslot, nullptr, bytes);
  if (rc) return rc;
  rc = eval_dev(ctx, ctx->stage[oslot], dops, noperands, program, proglen, batch, ctx->hstream);
  if (rc) return rc;
  return s.out(h_out, oslot, bytes);
}

int nflhip_polymul(nflhip_ctx *ctx, void *c, const void *a, const void *b, size_t batch) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!c || !a || !b) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const void *ins[2] = {a, b};
  int rc = run_pipelined(ctx, batch, ins, 2, c, [&](const void *const *d, void *out, size_t cnt, void *st) {
    return nflhip_polymul_dev(ctx, out, d[0], d[1], cnt, st);
  });
  if (rc != NFLHIP_ERR_UNSUPPORTED) return rc;
  rc = s.in(0, a, bytes);
  if (rc) return rc;
  if ((rc = s.in(1, b, bytes))) return rc;
  rc = nflhip_polymul_dev(ctx, ctx->stage[0], ctx->stage[0], ctx->stage[1], batch, ctx->hstream);
  if (rc) return rc;
  return s.out(c, 0, bytes);
}
static int any_cmp_host(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int want_eq, int *result) {
  CHECK_CTX(ctx);
  if (!result) return fail(ctx, NFLHIP_ERR_INVALID, "NULL result");
  if (batch == 0) { *result = 0; return NFLHIP_OK; }
  if (!a || !b) return fail(ctx, NFLHIP_ERR_INVALID, "NULL operand");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  int rc = s.in(0, a, bytes);
  if (rc) return rc;
  if ((rc = s.in(1, b, bytes))) return rc;
  return any_cmp_dev(ctx, ctx->stage[0], ctx->stage[1], batch, want_eq, result, ctx->hstream);
}
int nflhip_any_eq(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int *result) {
  return any_cmp_host(ctx, a, b, batch, 1, result);
}
int nflhip_any_neq(nflhip_ctx *ctx, const void *a, const void *b, size_t batch, int *result) {
  return any_cmp_host(ctx, a, b, batch, 0, result);
}
int nflhip_crt_lift(nflhip_ctx *ctx, uint64_t *limbs, const void *d, size_t batch) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!limbs || !d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const size_t lbytes = batch * ctx->shape.n * ctx->shape.crt_L * sizeof(uint64_t);
  int rc = s.in(0, d, bytes);
  if (rc) return rc;
  if ((rc = s.in(1, nullptr, lbytes))) return rc;
  rc = nflhip_crt_lift_dev(ctx, (uint64_t *)ctx->stage[1], ctx->stage[0], batch, ctx->hstream);
  if (rc) return rc;
  return s.out(limbs, 1, lbytes);
}
int nflhip_crt_project(nflhip_ctx *ctx, void *d, const uint64_t *limbs, size_t L_in, size_t batch) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!limbs || !d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  if (L_in == 0) return fail(ctx, NFLHIP_ERR_INVALID, "L_in must be positive");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  const size_t lbytes = batch * ctx->shape.n * L_in * sizeof(uint64_t);
  int rc = s.in(1, limbs, lbytes);
  if (rc) return rc;
  if ((rc = s.in(0, nullptr, bytes))) return rc;
  rc = nflhip_crt_project_dev(ctx, ctx->stage[0], (const uint64_t *)ctx->stage[1], L_in, batch, ctx->hstream);
  if (rc) return rc;
  return s.out(d, 0, bytes);
}

int nflhip_sample(nflhip_ctx *ctx, void *d, size_t batch, int dist, uint64_t p0, uint64_t p1, const unsigned char *key,
                  uint64_t stream_id) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  int rc = s.in(0, nullptr, bytes);
  if (rc) return rc;
  rc = nflhip_sample_dev(ctx, ctx->stage[0], 0, batch, dist, p0, p1, key, stream_id, ctx->hstream);
  if (rc) return rc;
  return s.out(d, 0, bytes);
}

int nflhip_sample_gauss(nflhip_ctx *ctx, void *d, size_t batch, const nflhip_gauss *g, uint64_t amplifier,
                        const unsigned char *key, uint64_t stream_id) {
  CHECK_CTX(ctx);
  if (batch == 0) return NFLHIP_OK;
  if (!d) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  Staged s(ctx);
  const size_t bytes = poly_bytes(ctx, batch);
  int rc = s.in(0, nullptr, bytes);
  if (rc) return rc;
  rc = nflhip_sample_gauss_dev(ctx, ctx->stage[0], 0, batch, g, amplifier, key, stream_id, ctx->hstream);
  if (rc) return rc;
  return s.out(d, 0, bytes);
}

int nflhip_gauss_noise(nflhip_ctx *ctx, int64_t *h_out, size_t count, const nflhip_gauss *g, const unsigned char *key,
                       uint64_t stream_id) {
  CHECK_CTX(ctx);
  if (count == 0) return NFLHIP_OK;
  if (!h_out) return fail(ctx, NFLHIP_ERR_INVALID, "NULL argument");
  Staged s(ctx);
  const size_t bytes = count * sizeof(int64_t);
  int rc = s.in(0, nullptr, bytes);
  if (rc) return rc;
  rc = nflhip_gauss_noise_dev(ctx, (int64_t *)ctx->stage[0], 0, count, g, key, stream_id, ctx->hstream);
  if (rc) return rc;
  return s.out(h_out, 0, bytes);
}

// ---------------------------------------------------------------------------
// in-library timing of the metric kernel: HIP events on the launch stream
// ---------------------------------------------------------------------------
int nflhip_time_polymul_dev(nflhip_ctx *ctx, void *c, const void *a, const void *b, size_t batch, int iters, void *stream,
                            float *ms_per_pass) {
  CHECK_CTX(ctx);
  if (!ms_per_pass || iters <= 0) return fail(ctx, NFLHIP_ERR_INVALID, "bad timing arguments");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  HIPCHK(ctx, hipEventCreate(&e0));
  hipError_t he = hipEventCreate(&e1);
  if (he == hipSuccess) he = hipEventRecord(e0, st);
  if (he != hipSuccess) {
    (void)hipEventDestroy(e0);
    return hipfail(ctx, he, "timing events");
  }
  for (int i = 0; i < iters; ++i) {
    int rc = nflhip_polymul_dev(ctx, c, a, b, batch, stream);
    if (rc) {
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return rc;
    }
  }
  float ms = 0.f;
  he = hipEventRecord(e1, st);
  if (he == hipSuccess) he = hipEventSynchronize(e1);
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (he != hipSuccess) return hipfail(ctx, he, "timing events");
  *ms_per_pass = ms / (float)iters;
  return NFLHIP_OK;
}

}  // extern "C"
