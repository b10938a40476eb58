// kernels_crt_mfma.hip -- CRT lift (GMP::poly2mpz, gmp.hpp:183-209) for MANY 62-bit moduli on the matrix cores.
//
// The transforms and point-wise operators of this engine are scans of independent words: no contraction, no MFMA
// (BASELINE.json north_star).  The CRT lift is the one place where the reference's own formula IS a dense contraction:
//     X_i = sum_cm (Q/p_cm) * y_(cm,i),     y_(cm,i) = x_(cm,i) * (Q/p_cm)^-1 mod p_cm,     then mod Q
// -- for every coefficient i a vector of nm residues against a fixed nm x (L words) matrix of constants.  With 30 moduli
// (BASELINE configs[4], the XPIR-scale ring) that is 900 word products per coefficient; kernels_crt.hip runs them as 5 400
// 21-bit x 32-bit multiply-adds on the vector ALU and is bound by them (profiles/r03_crt_kernels.txt).  Here the same sum is an
// int8 GEMM on v_mfma_i32_32x32x32_i8:
//   * y (62 bits) is written in BALANCED base-256 digits, y = sum_a z_a 256^a with z_a in [-128, 127]: the bytes of
//     (y + 0x8080808080808080) ^ 0x8080808080808080 read as int8 (two VALU instructions per residue);
//   * Q/p_cm is written in balanced base-256 digits d_(cm,b) on the host, once per context;
//   * column k of the product, c_k = sum_cm sum_a z_(cm,a) d_(cm,k-a), is a dot product of length 8 nm <= 256 of int8
//     values: |c_k| <= 256 * 2^14 = 2^22, exact in the int32 accumulators; S = sum_k c_k 256^k;
//   * the reduction mod Q rides along: S / Q = sum_cm y_cm / p_cm EXACTLY, so t = floor(sum_cm y_cm / p_cm - 2^-20) (double
//     precision, error far below the margin) is floor(S / Q) or one less, and row 31 of the GEMM is "-t times the digits of
//     Q": the product comes out as S - t Q, in [0, 2Q), and one conditional subtraction of Q is all that is left.
// GEMM shape per coefficient: K = 8 * 32 (up to 31 moduli + the quotient row, zero padded; 32 moduli: P3 subtracts t Q) = 256, N = 256 columns
// (k <= 8 L for L <= 31 words): 65 536 int8 multiply-adds on the matrix cores instead of 5 400 32-bit ones on the VALU.
//
// Mapping (one 256-thread workgroup works on 64 consecutive coefficients per iteration; persistent, grid-stride; the next
// tile's residues are in flight while this one is worked on):
//   P1  thread (c = lane, g = wave) computes y for coefficient c and the moduli g, g+4, g+8, ... (the modulus is
//       wave-uniform: its constants are scalar loads), stores z into LDS in the A-fragment order and its share of
//       sum y / p next to it (from the top 32 bits of y: 2^-32 per term, far inside the 2^-20 margin);
//   P2  wave (u = w & 1, m = w >> 1): the 32 x 128 block "coefficients of half m, columns of half u" = 4 N-tiles x 8
//       K-steps = 32 MFMAs.  The wave's 32 B fragments (128 VGPRs) are loaded ONCE per kernel; A comes from LDS (one
//       ds_read_b128 per 4 MFMAs; the quotient row is patched into the last K-step's registers).  Column map: N-tile t,
//       lane column j <-> k = 8 j + t, so the four tiles of a wave are the four bytes of ONE 32-bit digit position
//       2 j + u: X = c_0 + c_1 2^8 + c_2 2^16 + c_3 2^24 (signed, < 2^47) is formed in the lane and parked in LDS;
//   P3  from here on wave w owns coefficients 16 w .. 16 w + 15 and no barrier is needed any more: lane (c = lane & 15,
//       part r = lane >> 4) takes the 16 positions 16 r .. 16 r + 15: local carry chain, then the carries between the four
//       parts by lane permutes (a part passes a carry on only when its upper 15 digits are all zeros / all ones: two flag
//       bits travel with its carry out), then the same scheme for the borrow of "- Q";
//   P4  the wave's 16 x L result words are contiguous in HBM: coalesced stores from its rows of the LDS stage.
// Two workgroup barriers per tile (A fragments ready, digit positions ready).
// The K index of an MFMA operand is only ever used symmetrically (the same (lane half, byte) slot of A and B carries
// the same (modulus, digit) pair), so the kernel depends on the instruction's row / column maps only: A row = lane & 31,
// B column = lane & 31, C (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
//
// Phase costs measured by removing one phase's arithmetic at a time: profiles/r04_crt_mfma.txt.
#include "modarith64.h"

namespace nflhip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static constexpr int kMfmaCoef = 64;              // coefficients per workgroup iteration
static constexpr int kXStride = 65;               // u64 slots per coefficient in the X array: 64 digits, odd stride (bank spread)
static constexpr int kStStride = 33;              // u64 slots per coefficient in the result stage: 32 limbs, odd stride
static constexpr u64 kBias = 0x8080808080808080ull;

// QROW: the quotient row rides in modulus slot 31 (up to 31 moduli); with 32 moduli every slot is taken and P3 subtracts t Q from
// its 16 positions instead (one multiply-add per position more)
template <bool QROW>
__global__ __launch_bounds__(256, 2) void k_crt_lift_mfma(u64 *out, const u64 *d, const MC64 *__restrict__ mc,
                                                          const uint4 *__restrict__ bfrag, const u32 *__restrict__ qdig,
                                                          int logn, int nm, int L, size_t ncoef) {
  __shared__ uint4 a_frag[2 * 8 * 64];                     // 16 KiB: [M-tile][K-step][lane] x 16 bytes
  __shared__ long long xs[kMfmaCoef * kXStride];           // 33 KiB: [coefficient][32-bit digit position] signed partial sums
  __shared__ u64 stage[kMfmaCoef * kStStride];             // 17 KiB: [coefficient][limb] results
  __shared__ double fpart[4 * 64];                         // [wave][coefficient] shares of sum y / p
  __shared__ double invp[32];                              // 2^30 / p
  __shared__ u32 qs[64];                                   // Q as 32-bit digits
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the constants of "its" moduli are scalar loads
  const int u = w & 1, m = w >> 1;
  if (tid < 32) invp[tid] = tid < nm ? 0x1p30 / (double)mc[tid].p : 0.0;
  if (tid >= 64 && tid < 128) qs[tid - 64] = qdig[tid - 64];
  // the wave's B fragments: K-step s, N-tile 4u + tt
  v4i breg[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const uint4 v = bfrag[(s * 8 + 4 * u + tt) * 64 + lane];
      breg[s][tt] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  u64 *a_words = reinterpret_cast<u64 *>(a_frag);
  const size_t ngroups = ncoef / kMfmaCoef;
  const size_t nmask = (((size_t)1) << logn) - 1;
  u64 xv[8];
  auto fetch = [&](size_t grp) {
    const size_t gid = grp * kMfmaCoef + (size_t)lane;
    const size_t b = gid >> logn, i = gid & nmask;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cm = w + 4 * q;
      xv[q] = d[((b * (size_t)nm + (size_t)(cm < nm ? cm : 0)) << logn) + i];
    }
  };
  if (blockIdx.x < ngroups) fetch(blockIdx.x);
  __syncthreads();
#ifdef NFLHIP_CRT_MFMA_STAMP
  unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime(), st_n = 0;
#define NFLHIP_STAMP(i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); st_acc[i] += now - st_t; st_t = now; }
#else
#define NFLHIP_STAMP(i)
#endif
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    // ---- P1: residues -> y -> balanced digits in A-fragment order; this thread's share of sum y / p
    {
      const int mt = lane >> 5, row = lane & 31;
      double f = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cm = w + 4 * q;
        u64 z = 0;
        if (cm < nm) {
          const u64 p = mc[cm].p, yi = mc[cm].yinv, yis = mc[cm].yinv_sh;
          const u64 y = mul_shoup<u64>(xv[q], yi, yis, p);
          f += (double)(u32)(y >> 30) * invp[cm];
          z = (y + kBias) ^ kBias;
        }
        const int s = cm >> 2, h = (cm >> 1) & 1, e = cm & 1;
        a_words[(((mt * 8 + s) * 64) + row + 32 * h) * 2 + e] = z;
      }
      fpart[w * 64 + lane] = f;
      if (grp + gridDim.x < ngroups) fetch(grp + gridDim.x);   // the next tile's residues: in flight until the next P1
    }
    NFLHIP_STAMP(0)
    __syncthreads();
    NFLHIP_STAMP(1)
    // ---- P2: 32 MFMAs per wave
    v16i acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0;
    {
      // the quotient row: modulus slot 31 = (K-step 7, lane half 1, second word), digit 0 only
      const int c = 32 * m + (lane & 31);
      const double f = ((fpart[c] + fpart[64 + c]) + fpart[128 + c]) + fpart[192 + c];
      int tq = (int)floor(f - 0x1p-20);
      tq = tq < 0 ? 0 : tq;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 av = a_frag[(m * 8 + s) * 64 + lane];
        v4i a = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
        if (QROW && s == 7 && lane >= 32) a[2] = (-tq) & 0xff;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, breg[s][tt], acc[tt], 0, 0, 0);
      }
    }
    // four neighbouring byte columns of one 32-bit digit position -> one signed sum (|c| <= 2^22: the pairs fit 32 bits)
    {
      const int j = lane & 31, hh = lane >> 5;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int t01 = acc[0][r] + (acc[1][r] << 8), t23 = acc[2][r] + (acc[3][r] << 8);
        xs[(m * 32 + row) * kXStride + 2 * j + u] = (long long)t01 + ((long long)t23 << 16);
      }
    }
    NFLHIP_STAMP(2)
    __syncthreads();
    NFLHIP_STAMP(3)
    // ---- P3: wave w owns coefficients 16 w .. 16 w + 15 from here on -- lane (c = lane & 15, part r = lane >> 4) has the
    //      16 digit positions 16 r .. 16 r + 15; everything between the parts goes through lane permutes: no barrier
    {
      const int cl = lane & 15, r = lane >> 4, c = 16 * w + cl;
      const long long *xc = xs + c * kXStride + 16 * r;
      long long tqq = 0;   // 32 moduli: the quotient estimate again (the same sum in the same order as P2's), to be subtracted here
      if (!QROW) {
        const double f = ((fpart[c] + fpart[64 + c]) + fpart[128 + c]) + fpart[192 + c];
        const int t = (int)floor(f - 0x1p-20);
        tqq = t < 0 ? 0 : t;
      }
      u32 dg[16];
      long long carry = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const long long t = xc[k] - (QROW ? 0 : tqq * (long long)qs[16 * r + k]) + carry;
        dg[k] = (u32)t;
        carry = t >> 32;
      }
      u32 orv = 0, andv = ~0u;
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        orv |= dg[k];
        andv &= dg[k];
      }
      {
        // carry into this part: a lower part passes its own carry out, plus or minus one when the carry INTO it runs
        // through all of its digits (upper 15 digits all ones / all zeros)
        const int info = (int)carry * 4 + ((orv == 0 ? 1 : 0) | (andv == ~0u ? 2 : 0));   // |carry| < 2^17
        int cin = 0, mine = 0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const int oi = __shfl(info, cl + 16 * rr, 64);
          const u32 od = (u32)__shfl((int)dg[0], cl + 16 * rr, 64);
          const long long t = (long long)od + (long long)cin;
          int ext = 0;
          if ((oi & 2) && t >= (1LL << 32)) ext = 1;
          if ((oi & 1) && t < 0) ext = -1;
          cin = (oi >> 2) + ext;
          if (r == rr + 1) mine = cin;
        }
        const long long t0 = (long long)dg[0] + (long long)mine;
        dg[0] = (u32)t0;
        long long cy = t0 >> 32;
        if (__builtin_amdgcn_ballot_w64(cy != 0) != 0) {   // rare (|cin| < 2^17 against a 32-bit digit)
#pragma unroll
          for (int k = 1; k < 16; ++k) {
            const long long t = (long long)dg[k] + cy;
            dg[k] = (u32)t;
            cy = t >> 32;
          }
        }
      }
      // S - Q on this part, then the borrows between the parts the same way
      u32 D[16];
      unsigned bw = 0;
      u32 orD = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        D[k] = __builtin_subc(dg[k], qs[16 * r + k], bw, &bw);
        orD |= D[k];
      }
      {
        const int info = (int)bw | (orD == 0 ? 2 : 0);   // borrow out, "the part's difference is 0" (a borrow in runs through)
        unsigned b = 0, bin = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const unsigned v = (unsigned)__shfl(info, cl + 16 * rr, 64);
          if (rr == r) bin = b;
          b = (v & 1u) | ((v >> 1) & b);
        }
        const bool ge = b == 0;   // S >= Q: take the difference
        unsigned brw = D[0] < bin ? 1u : 0u;
        D[0] -= bin;
        if (__builtin_amdgcn_ballot_w64(brw != 0 && ge) != 0) {   // rare: the part's lowest digit of S - Q is 0
#pragma unroll
          for (int k = 1; k < 16; ++k) {
            const unsigned nb = D[k] < brw ? 1u : 0u;
            D[k] -= brw;
            brw = nb;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const u32 lo = ge ? D[2 * k] : dg[2 * k], hi = ge ? D[2 * k + 1] : dg[2 * k + 1];
          stage[c * kStStride + 8 * r + k] = (u64)lo | ((u64)hi << 32);
        }
      }
    }
    NFLHIP_STAMP(4)
    // ---- P4: the wave's 16 x L result words are contiguous in HBM: 32 lanes per coefficient, 2 coefficients per pass.
    //      The stage rows were written by this wave: LDS operations of one wave complete in order, no barrier.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      // (the lane's offsets are recomputed per tile on purpose: hoisted out of the loop they are ten more live registers
      //  next to the 128 of the B fragments, and their spill reloads wait on EVERY outstanding load and store)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int k = ln & 31;
      u64 *o = out + (grp * (size_t)kMfmaCoef + 16 * w + (ln >> 5)) * (size_t)L + k;
      const u64 *sp = stage + (16 * w + (ln >> 5)) * kStStride + k;
      if (k < L) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) o[(size_t)(2 * rr) * (size_t)L] = sp[2 * rr * kStStride];
      }
    }
    NFLHIP_STAMP(5)
#ifdef NFLHIP_CRT_MFMA_STAMP
    ++st_n;
#endif
    // (reuse across iterations: a_frag / fpart are rewritten after this wave's P4 and were last read before barrier 2 by
    //  every wave; xs is rewritten after the next barrier 1, which every wave reaches after its P3; the stage rows are
    //  private to the wave)
  }
#ifdef NFLHIP_CRT_MFMA_STAMP
  if (blockIdx.x == 3 && lane == 0)
    printf("wave %d tiles %llu cycles/tile: P1 %llu  barrier1 %llu  P2 %llu  barrier2 %llu  P3 %llu  P4 %llu\n", w, st_n, st_acc[0] / st_n,
           st_acc[1] / st_n, st_acc[2] / st_n, st_acc[3] / st_n, st_acc[4] / st_n, st_acc[5] / st_n);
#endif
}

// `bfrag` / `qdig`: DevTables::crt_bfrag (api.hip build_tables; its modulus slot 31 holds the digits of Q itself), row 0 of
// DevTables::qsh (Q as 32-bit digits, zero padded to 72).  hipErrorNotSupported when the shape has no table (few moduli: the
// VALU kernels are on the memory system there) or the batch is not a multiple
// of the tile.
hipError_t launch_crt_lift_mfma_u64(const Shape &s, const DevTables &t, uint64_t *limbs, const uint64_t *d, size_t batch,
                                    hipStream_t st) {
  if (s.limb_bits != 64 || !t.crt_bfrag || s.nm > 32 || (batch * s.n) % kMfmaCoef != 0 || s.crt_L < 4 || s.crt_L > 31)
    return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  const size_t ncoef = batch * s.n;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  size_t blocks = (size_t)cus * 2;
  if (blocks > ncoef / kMfmaCoef) blocks = ncoef / kMfmaCoef;
  if (s.nm < 32)
    hipLaunchKernelGGL(k_crt_lift_mfma<true>, dim3((unsigned)blocks), dim3(256), 0, st, limbs, d, (const MC64 *)t.mc,
                       (const uint4 *)t.crt_bfrag, (const u32 *)t.qsh, s.logn, (int)s.nm, (int)s.crt_L, ncoef);
  else
    hipLaunchKernelGGL(k_crt_lift_mfma<false>, dim3((unsigned)blocks), dim3(256), 0, st, limbs, d, (const MC64 *)t.mc,
                       (const uint4 *)t.crt_bfrag, (const u32 *)t.qsh, s.logn, (int)s.nm, (int)s.crt_L, ncoef);
  return hipGetLastError();
}

// ---- project (GMP::mpz2poly gmp.hpp:211-219 / poly::set_mpz gmp.hpp:73-108): x_(cm,i) = X_i mod p_cm -------------------------
// The same GEMM the other way round: X_i (L_in <= 32 words = up to 256 bytes a_k) against the constants 256^k mod p_cm,
//     X mod p = sum_k a_k (256^k mod p)   (mod p).
//   * A: the bytes of X as they lie in memory, each XOR 0x80 (a' = a - 128 in [-128, 127]; bytes beyond L_in words count as
//     a = 0) -- no arithmetic, no carries; the "+128" is a constant per residue, 128 sum_k (256^k mod p), added at the end;
//   * B: 256^k mod p_cm in balanced base-256 digits b = 0..7; column (cm, b) <-> N-tile b, lane column cm: the eight byte
//     columns of a residue sit in ONE lane across the eight tiles, and a wave's four tiles are the four bytes of one 32-bit half:
//     the halves X_0, X_1 (signed, < 2^48) are formed in the lane, R = X_0 + X_1 2^32 + (2^18 p + that constant) is in (0, 2^81);
//   * R mod p: one Shoup product of the top 17 bits by 2^64 mod p plus the folded low word -- the tail of k_crt_project64_mac.
// |c| <= 256 * 255 * 128 < 2^23: exact in int32.
// Mapping (256-thread workgroup, 64 coefficients per tile, persistent, the next tile's words prefetched):
//   P1  32 lanes per coefficient read its L_in words (contiguous in HBM, two coefficients per wave instruction), XOR the bias,
//       and store them in A-fragment order (the two halves of a K-step 528 bytes apart: conflict-free for this writer);
//   P2  as the lift: 32 MFMAs per wave, halves to LDS;
//   P3  thread (c = lane, cm = wave + 4 q): the two halves from LDS, the reduction with wave-uniform constants, a coalesced store.
static constexpr int kPjHalf = 528 / 16;          // uint4 slots per (M-tile, K-step, lane half): 32 rows + one slot of padding

// NP = 1: inputs of up to 32 words.  NP = 2: up to 64 (what poly::set_mpz meets when it reduces a product of two lifted values):
// the upper 32 words go through the same GEMM first, their residues wait in LDS, and the lower words' residues take them along
// as r_lo + r_hi * (2^2048 mod p) -- one more Shoup product per residue.
template <int NP>
__global__ __launch_bounds__(256, 2) void k_crt_project_mfma(u64 *d, const u64 *limbs, const MC64 *__restrict__ mc,
                                                             const uint4 *__restrict__ bproj, const u64 *__restrict__ coff,
                                                             const u64 *__restrict__ c2048, int logn, int nm, int Lin, size_t ncoef) {
  __shared__ uint4 a_frag[2 * 8 * 2 * kPjHalf];            // 16.5 KiB: [M-tile][K-step][lane half][32 rows + pad] x 16 bytes
  __shared__ long long xs[kMfmaCoef * kXStride];           // 33 KiB: [coefficient][2 cm + half] signed partial sums
  __shared__ u64 rhi[NP > 1 ? 8 * 256 : 1];                // 16 KiB (NP = 2): the upper words' residues, [q][thread]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = w & 1, m = w >> 1;
  v4i breg[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const uint4 v = bproj[(s * 8 + 4 * u + tt) * 64 + lane];
      breg[s][tt] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  u64 *a_words = reinterpret_cast<u64 *>(a_frag);
  const size_t ngroups = ncoef / kMfmaCoef;
  const size_t nmask = (((size_t)1) << logn) - 1;
  u64 xv[8];
  auto fetch = [&](size_t grp, int ps) {   // words 32 ps .. 32 ps + 31 of the tile's coefficients
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (recomputed per tile: see the lift's P4)
    const int k = (ln & 31) + 32 * ps;
    const u64 *src = limbs + (grp * (size_t)kMfmaCoef + 16 * w + (ln >> 5)) * (size_t)Lin + k;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) xv[rr] = k < Lin ? src[(size_t)(2 * rr) * (size_t)Lin] : 0;
  };
  if (blockIdx.x < ngroups) fetch(blockIdx.x, NP - 1);
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
#pragma unroll
    for (int ps = NP - 1; ps >= 0; --ps) {
      // (every lane-dependent offset below is recomputed per tile from an opaque copy of the lane number: hoisted out of the
      //  loop they are some forty live registers next to the 128 of the B fragments -- see the lift's P4)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      // ---- P1: word k of coefficient c -> K bytes 8 k .. 8 k + 7 of row c
      {
        const int k = ln & 31, s = k >> 2, h = (k >> 1) & 1, e = k & 1;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int c = 16 * w + 2 * rr + (ln >> 5), mt = c >> 5, row = c & 31;
          a_words[((((mt * 8 + s) * 2 + h) * kPjHalf) + row) * 2 + e] = xv[rr] ^ kBias;
        }
        // the next words: this tile's lower half, or the next tile's first pass -- in flight during everything below
        if (ps > 0) fetch(grp, ps - 1);
        else if (grp + gridDim.x < ngroups) fetch(grp + gridDim.x, NP - 1);
      }
      __syncthreads();
      // ---- P2: 32 MFMAs per wave
      v16i acc[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 av = a_frag[((m * 8 + s) * 2 + (ln >> 5)) * kPjHalf + (ln & 31)];
        const v4i a = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, breg[s][tt], acc[tt], 0, 0, 0);
      }
      {
        const int j = ln & 31, hh = ln >> 5;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const int t01 = acc[0][r] + (acc[1][r] << 8), t23 = acc[2][r] + (acc[3][r] << 8);
          xs[(m * 32 + row) * kXStride + 2 * j + u] = (long long)t01 + ((long long)t23 << 16);
        }
      }
      __syncthreads();
      // ---- P3: R = X_0 + X_1 2^32 + 2^18 p, then R mod p
      {
        const size_t gid = grp * kMfmaCoef + (size_t)ln;
        const size_t b = gid >> logn, i = gid & nmask;
        const long long *xc = xs + ln * kXStride;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int cm = w + 4 * q;
          if (cm < nm) {
            const MC64 c = mc[cm];
            const Mod md = make_mod(c);
            const unsigned __int128 off = (unsigned __int128)coff[2 * cm] | ((unsigned __int128)coff[2 * cm + 1] << 64);
            const unsigned __int128 T = (unsigned __int128)((__int128)xc[2 * cm] + ((__int128)xc[2 * cm + 1] << 32)) + off;
            u64 r = fold2(shoup_acc<true>((u64)(T >> 64), Tw64{c.beta, c.beta_sh}, fold2((u64)T, md), md), md);   // < p + 4 delta
            if (NP > 1 && ps > 0) {
              rhi[q * 256 + tid] = r;   // (this thread reads it back: no barrier)
            } else {
              if (NP > 1) r = fold2(shoup_acc<true>(rhi[q * 256 + tid], Tw64{c2048[2 * cm], c2048[2 * cm + 1]}, r, md), md);
              d[((b * (size_t)nm + (size_t)cm) << logn) + i] = csub<u64>(r, md.p);
            }
          }
        }
      }
      // (a_frag is rewritten after this wave's P3 and was last read before barrier 2; xs is rewritten after the next barrier 1)
    }
  }
}

// `bproj` / `coff` / `c2048`: DevTables::crt_bproj / crt_coff / crt_c2048 (api.hip build_tables).  hipErrorNotSupported when the
// shape has no table (few moduli), the input is wider than 64 words or the batch is not a multiple of the tile.
hipError_t launch_crt_project_mfma_u64(const Shape &s, const DevTables &t, uint64_t *d, const uint64_t *limbs, size_t L_in,
                                       size_t batch, hipStream_t st) {
  if (s.limb_bits != 64 || !s.small_delta || !t.crt_bproj || s.nm > 32 || L_in == 0 || L_in > 64 || (batch * s.n) % kMfmaCoef != 0)
    return hipErrorNotSupported;
  if (batch == 0) return hipSuccess;
  const size_t ncoef = batch * s.n;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  size_t blocks = (size_t)cus * 2;
  if (blocks > ncoef / kMfmaCoef) blocks = ncoef / kMfmaCoef;
  if (L_in <= 32)
    hipLaunchKernelGGL(k_crt_project_mfma<1>, dim3((unsigned)blocks), dim3(256), 0, st, d, limbs, (const MC64 *)t.mc,
                       (const uint4 *)t.crt_bproj, (const u64 *)t.crt_coff, (const u64 *)t.crt_c2048, s.logn, (int)s.nm, (int)L_in, ncoef);
  else
    hipLaunchKernelGGL(k_crt_project_mfma<2>, dim3((unsigned)blocks), dim3(256), 0, st, d, limbs, (const MC64 *)t.mc,
                       (const uint4 *)t.crt_bproj, (const u64 *)t.crt_coff, (const u64 *)t.crt_c2048, s.logn, (int)s.nm, (int)L_in, ncoef);
  return hipGetLastError();
}

// first-use warm-up (api.hip warm_up_device): the runtime loads a translation unit's code object at the first launch of ANY of its kernels
__global__ void k_warm_crt_mfma() {}
hipError_t warm_crt_mfma(hipStream_t st) {
  hipLaunchKernelGGL(k_warm_crt_mfma, dim3(1), dim3(64), 0, st);
  return hipGetLastError();
}

}  // namespace nflhip
