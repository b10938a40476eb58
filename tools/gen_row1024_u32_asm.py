#!/usr/bin/env python3
"""Generate nfllib_amd/csrc/row1024_u32_gfx950.s -- the hand-scheduled gfx950 assembly version of the fused product
for 32-bit limbs, n = 1024 (BASELINE configs[0]'s shape): c = INTT(NTT(a) (.) NTT(b)), ONE WAVE PER RNS ROW.

Same algorithm, lane mapping, LDS layouts and device tables as k_row<Pol32, 0, 4> in kernels_wave.hip (read fwd_row /
inv_row there first); what the generator adds over hipcc:

  * every coefficient lives in the LOW half of an even-aligned VGPR pair whose high half is scratch, so the
    multiply-add butterflies need no moves at all: a Cooley-Tukey butterfly is 7 instructions
        t = x - 2p ; X = min(x, t) ; q = mulhi(y, w') ; u = 2X + 2p ; (X,*) += q*(-p) ; (X,*) += y*w ; y = u - X'
    (9 in the compiled kernel) and a Gentleman-Sande one 8 (9);
  * a's and b's forward transforms run together and share every twiddle register and twiddle load;
  * the 15 twiddles of the first forward / last inverse pass are wave-uniform: they sit in SGPRs for the whole kernel;
  * the point-wise product leaves its result in [0, 2p) (the inverse butterflies take that), 14 instead of 16.

Run by nfllib_amd/csrc/Makefile (after tools/gen_polymul_asm.py, whose emitter and file templates it reuses).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_polymul_asm as G   # noqa: E402  (Emitter, interleave, HEADER / FOOTER, args_yaml)

# LB = lanes per 16-block of a row: 4 -> one wave per 1024-word row, 8 -> two waves per 2048-word row, 16 -> a whole
# 256-thread workgroup per 4096-word row (kernels_wave.hip's LB); the last pass then has 2, 3, 4 stages
SHAPES = {4: 1024, 8: 2048, 16: 4096}

# ---- registers
S_P, S_2P, S_NEGP, S_MU, S_NINV, S_NINVSH, S_W1N, S_W1NSH = 30, 31, 32, 33, 34, 35, 36, 37
S_TW = 40            # s[40:69]: tw[1..15] = {w, w'} pairs: record k at s[40 + 2(k-1)]
S_DUMMY = "s[72:73]"
V_TID, V_LANE, V_GOFF, V_A1, V_A2, V_A3, V_TWO, V_TMP = 0, 1, 2, 3, 4, 5, 6, 7
V_A, V_B = 8, 40     # 16 even-aligned pairs each: coefficient q of a in v[8 + 2q] (low half), scratch in v[9 + 2q]
V_TWA, V_TWB = 72, 88   # two buffers of 8 per-lane twiddle records {w, w'}
V_S = [104, 108]     # per-stream temporaries: T0, Q, T2, S
V_PW = 112           # point-wise temp pair
V_P, V_2P = 116, 117  # p and 2p once more, in VGPRs: a plain VOP2 add / sub issues in ~2.5 cycles per wave64 with VGPR operands
                      # only and in ~4.4 with an SGPR operand (profiles/r03_ubench_issue.txt)
NEXT_VGPR = 118
# p / 2p of the plain subtractions and additions stay SGPR operands: the all-VGPR form (which isolated streams price at
# 2.5 instead of 4.4 cycles, tools/ubench_issue.hip) changes nothing in the kernel -- 4.454 M vs 4.443 M cycles per launch and
# XCD, 236.6 vs 237.4 M products/s on the same box (profiles/r03_operand_ab.txt).  NFL_GEN_VGPR_OPERANDS=1 rebuilds it.
SGPR_OPERANDS = not os.environ.get("NFL_GEN_VGPR_OPERANDS")


def sub_const(dst, src, which):
    """dst = src - {p, 2p}"""
    if SGPR_OPERANDS:
        return "v_subrev_u32_e32 v%d, s%d, v%d" % (dst, which, src)
    return "v_sub_u32_e32 v%d, v%d, v%d" % (dst, src, {S_P: V_P, S_2P: V_2P}[which])


def add_2p(reg):
    if SGPR_OPERANDS:
        return "v_add_u32_e32 v%d, s%d, v%d" % (reg, S_2P, reg)
    return "v_add_u32_e32 v%d, v%d, v%d" % (reg, V_2P, reg)
NEXT_SGPR = 80
SLAB = 1088 * 4      # bytes of LDS per 1024 row words (the padding of either exchange layout included)


def sreg(k):
    """twiddle record k (1..15) of the uniform passes as operand strings (w, w')"""
    return "s%d" % (S_TW + 2 * (k - 1)), "s%d" % (S_TW + 2 * (k - 1) + 1)


def vrec(buf, i):
    return "v%d" % (buf + 2 * i), "v%d" % (buf + 2 * i + 1)


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def ct(x, y, tw):
    """x' = X + (y w - q p), y' = 2X + 2p - x' (Harvey ranges: x < 4p in, both < 4p out)"""
    w, wp = tw

    def gen(s):
        T0, Q, T2 = V_S[s], V_S[s] + 1, V_S[s] + 2
        yield sub_const(T0, x, S_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, x, T0), None, None
        yield "v_mul_hi_u32 v%d, v%d, %s" % (Q, y, wp), None, None
        yield "v_lshl_add_u32 v%d, v%d, 1, s%d" % (T2, x, S_2P), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(x), S_DUMMY, Q, S_NEGP, pair(x)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (pair(x), S_DUMMY, y, w, pair(x)), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (y, T2, x), None, None
    return gen


def gs(x, y, tw, xsrc=None, ysrc=None):
    """x' = lazy(x + y), y' = (y - x + 2p) w - q p (inputs < 2p, outputs < 2p); xsrc / ysrc: read the inputs from other
    registers (the inverse-only kernel's first stage takes them from its load block)"""
    w, wp = tw
    xs, ys = (x if xsrc is None else xsrc), (y if ysrc is None else ysrc)

    def gen(s):
        T0, Q, D, S = V_S[s], V_S[s] + 1, V_S[s] + 2, V_S[s] + 3
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, xs, ys), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, ys, xs), None, None
        yield add_2p(D), None, None
        yield sub_const(T0, S, S_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, S, T0), None, None
        yield "v_mul_hi_u32 v%d, v%d, %s" % (Q, D, wp), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, s%d, 0" % (pair(y), S_DUMMY, Q, S_NEGP), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (pair(y), S_DUMMY, D, w, pair(y)), None, None
    return gen


def csub(reg, dst, bound_sgpr, s):
    T0 = V_S[s]
    yield sub_const(T0, reg, bound_sgpr), None, None
    yield "v_min_u32_e32 v%d, v%d, v%d" % (dst, reg, T0), None, None


def pointwise(a, b):
    """a = a*b mod p in [0, 2p): exact Barrett on canonical operands (barrett<uint32_t>::mul without its last subtract)"""
    def gen(s):
        Q, TH = V_S[s] + 1, V_S[s] + 2
        P0 = V_PW + 2 * s
        for r in (a, b):
            yield from csub(r, r, S_2P, s)
            yield from csub(r, r, S_P, s)
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (pair(P0), S_DUMMY, a, b), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 28" % (TH, P0 + 1, P0), None, None
        yield "v_mul_hi_u32 v%d, v%d, s%d" % (Q, TH, S_MU), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(P0), S_DUMMY, Q, S_NEGP, pair(P0)), None, None
        yield from csub(P0, a, S_2P, s)
    return gen


def mul_shoup_exact(y, dst, w_s, wp_s, s):
    """dst = y * w mod p, canonical (dst is the low half of its pair)"""
    Q = V_S[s] + 1
    yield "v_mul_hi_u32 v%d, v%d, s%d" % (Q, y, wp_s), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, s%d, 0" % (pair(dst), S_DUMMY, Q, S_NEGP), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(dst), S_DUMMY, y, w_s, pair(dst)), None, None
    yield from csub(dst, dst, S_P, s)


def last(u, x):
    """stage 0 of the inverse with n^-1 folded in: canonical outputs (Pol32T::last)"""
    def gen(s):
        D, S = V_S[s] + 2, V_S[s] + 3
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, u, x), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, x, u), None, None
        yield add_2p(D), None, None
        yield from mul_shoup_exact(S, u, S_NINV, S_NINVSH, s)
        yield from mul_shoup_exact(D, x, S_W1N, S_W1NSH, s)
    return gen


def base_mul(g4, rec, negate):
    """Round 6, products on INCOMPLETE transforms (see tools/asmgen/incomplete.py for the 64-bit twin): the four words 4 g4 .. 4 g4 + 3
    of either operand are a residue modulo X^4 - zeta (zeta = +w, or -w when `negate`; rec = the VGPR record (w, w') of the last
    retained stage).  c_k = sum_{i+j=k} a_i b_j + zeta sum_{i+j=k+4} a_i b_j: canonical operands (< p < 2^30) make every four-term
    sum < 2^62 -- it accumulates in ONE 64-bit chain of v_mad_u64_u32 without carries -- and one Barrett step reduces it:
    th = T >> 30, mu = floor(2^62 / p) = 2^32 + m, q^ = th + mulhi(th, m) (q - q^ <= 3, r < 4p < 2^32), then one conditional
    subtraction (< 2p, what the inverse butterflies take).  zeta b_j replaces b_j once its raw value is dead.  The results wait in
    the HIGH halves of a's register pairs (scratch in this kernel): the first inverse stage / exchange reads them from there."""
    w, wp = rec

    def gen(s):
        T0, Q, TH, NW = V_S[s], V_S[s] + 1, V_S[s] + 2, V_S[s] + 3
        P0 = V_PW + 2 * s
        a = [V_A + 2 * (4 * g4 + i) for i in range(4)]
        b = [V_B + 2 * (4 * g4 + i) for i in range(4)]
        zw, zwp = w, wp
        if negate:   # -w = p - w, its companion floor((p - w) 2^32 / p) = ~w'; the high half of b0's pair is free scratch
            yield "v_sub_u32_e32 v%d, s%d, %s" % (NW, S_P, w), None, None
            yield "v_not_b32_e32 v%d, %s" % (b[0] + 1, wp), None, None
            zw, zwp = "v%d" % NW, "v%d" % (b[0] + 1)
        for r in a + b:
            yield from csub(r, r, S_2P, s)
            yield from csub(r, r, S_P, s)
        for k in (3, 2, 1, 0):
            ys = [b[k - i] if k - i >= 0 else b[k - i + 4] for i in range(4)]
            for i in range(4):
                yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(P0), S_DUMMY, a[i], ys[i], "0" if i == 0 else pair(P0)), None, None
            yield "v_alignbit_b32 v%d, v%d, v%d, 30" % (TH, P0 + 1, P0), None, None
            yield "v_mul_hi_u32 v%d, v%d, s%d" % (Q, TH, S_MU), None, None
            yield "v_add_u32_e32 v%d, v%d, v%d" % (Q, Q, TH), None, None
            yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(P0), S_DUMMY, Q, S_NEGP, pair(P0)), None, None
            yield from csub(P0, a[k] + 1, S_2P, s)
            if k:   # b_k <- zeta b_k (canonical)
                yield "v_mul_hi_u32 v%d, v%d, %s" % (Q, b[k], zwp), None, None
                yield "v_mad_u64_u32 %s, %s, v%d, s%d, 0" % (pair(P0), S_DUMMY, Q, S_NEGP), None, None
                yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (pair(P0), S_DUMMY, b[k], zw, pair(P0)), None, None
                yield from csub(P0, b[k], S_P, s)
    return gen


def run(em, jobs):
    for i in range(0, len(jobs), 2):
        gens = [jobs[i](0)]
        if i + 1 < len(jobs):
            gens.append(jobs[i + 1](1))
        G.interleave(em, gens)


class RowGen:
    """what every kernel of this file shares: the lane / row bookkeeping, the row streams, the per-lane twiddle loads (counted, so
    that waits can name the load they need), the 16-point register stages and the LDS exchanges of a row of W = 16 LB lanes"""

    def __init__(self, LB):
        self.LB = LB
        self.W = 16 * LB                       # lanes per row
        self.LG = LB.bit_length() - 1          # log2 LB
        self.LOGN = 8 + self.LG
        self.NS3 = self.LG                     # stages of the last pass: 2, 3, 4
        self.WAVES = self.W // 64              # waves per row
        self.em = G.Emitter()
        self.issued = 0                        # vector-memory LOADS issued so far (in-order return)
        self.bufs = [V_TWA, V_TWB]
        self.e1_row = lambda q: 4 * (self.W + LB) * q                  # x[t + W q]
        self.e1_blk = lambda q: 4 * LB * q                             # word LB q + l of block B
        self.e2_blk = lambda q: 4 * LB * q + 4 * (q >> (4 - self.LG))
        self.e2_thr = lambda q: 4 * q                                  # word 16 t + q

    # ---- prologue pieces
    def lane_and_row(self):
        """V_LANE, V_GOFF; s18 = row of the workgroup, s19 = row, s21 = store flag (a surplus row of the last workgroup walks
        through every barrier on the last row and stores nothing; a surplus WAVE of a one-wave-per-row kernel just ends)"""
        R, V, L = self.em.raw, self.em.valu, self.em.lines.append
        V("v_and_b32_e32 v%d, %d, v%d" % (V_LANE, self.W - 1, V_TID))             # t: lane of the row
        V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_GOFF, V_LANE))
        V("v_readfirstlane_b32 s18, v%d" % V_TID)
        R("s_lshr_b32 s18, s18, %d" % (6 + (self.WAVES.bit_length() - 1)))        # row of the workgroup
        R("s_waitcnt lgkmcnt(0)")
        R("s_lshl_b32 s19, s2, %d" % (2 - (self.WAVES.bit_length() - 1)))
        R("s_add_u32 s19, s19, s18")                         # row
        R("s_mov_b32 s21, 1")                                # store the result
        R("s_cmp_lt_u32 s19, s16")
        R("s_cbranch_scc1 .Llive")
        if self.LB == 4:
            R("s_endpgm")                                    # a surplus wave of the last workgroup (no workgroup barrier anywhere)
        else:
            R("s_sub_u32 s19, s16, 1")                       # a surplus row: walk through every barrier on the last row, store nothing
            R("s_mov_b32 s21, 0")
        L(".Llive:")

    def modulus_of_row(self, keep_element=None):
        """s20 = cm = row mod nm, s[22:23] = its twiddles, s[56:63] <- its ModConst record (requested); keep_element: an SGPR that
        receives row / nm (the batch element)"""
        R = self.em.raw
        R("s_mul_hi_u32 s20, s19, s15")
        if keep_element is not None:
            R("s_mov_b32 s%d, s20" % keep_element)
        R("s_mul_i32 s20, s20, s14")
        R("s_sub_u32 s20, s19, s20")                         # cm = row mod nm
        R("s_cmp_eq_u32 s14, 1")
        R("s_cselect_b32 s20, 0, s20")
        if keep_element is not None:
            R("s_cselect_b32 s%d, s19, s%d" % (keep_element, keep_element))   # (magic is 0 when nm = 1)
        R("s_lshl_b32 s74, s20, %d" % (self.LOGN + 3))       # twiddles of the modulus: psi + cm * n * 8
        R("s_add_u32 s22, s10, s74")
        R("s_addc_u32 s23, s11, 0")
        R("s_mul_i32 s74, s20, 56")                          # its ModConst<u32> record
        R("s_add_u32 s74, s12, s74")
        R("s_addc_u32 s75, s13, 0")
        R("s_load_dwordx8 s[56:63], s[74:75], 0x0")          # p 2p mu ninv ninv_sh w1ninv w1ninv_sh beta

    def advance(self, dst, base, index_sgpr, shift):
        """s[dst:dst+1] = s[base:base+1] + (index << shift) bytes (index < 2^32, the product up to 2^(32 + shift))"""
        R = self.em.raw
        R("s_lshr_b32 s75, s%d, %d" % (index_sgpr, 32 - shift))
        R("s_lshl_b32 s74, s%d, %d" % (index_sgpr, shift))
        R("s_add_u32 s%d, s%d, s74" % (dst, base))
        R("s_addc_u32 s%d, s%d, s75" % (dst + 1, base + 1))

    def lds_addresses(self):
        """LDS addresses of the row's slab: A1 = 4 t (+ 4 (W + LB) q), A2 = 4 ((W + LB) B + l) (+ 4 LB q [+ 4 (q >> (4 - lg LB))]),
        A3 = 68 t (+ 4 q); s78 = the slab"""
        R, V = self.em.raw, self.em.valu
        R("s_mul_i32 s78, s18, %d" % (SLAB * self.WAVES))
        V("v_add_u32_e32 v%d, s78, v%d" % (V_A1, V_GOFF))
        V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TMP, self.LG, V_LANE))             # B
        V("v_and_b32_e32 v%d, %d, v%d" % (V_A2, self.LB - 1, V_LANE))              # l
        V("v_mov_b32_e32 v%d, %d" % (V_A3, self.W + self.LB))
        V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_A2, V_TMP, V_A3, V_A2))     # (W + LB) B + l
        V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_A2, V_A2))
        V("v_add_u32_e32 v%d, s78, v%d" % (V_A2, V_A2))
        V("v_mov_b32_e32 v%d, 68" % V_A3)
        V("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_A3, V_LANE, V_A3))           # 68 t
        V("v_add_u32_e32 v%d, s78, v%d" % (V_A3, V_A3))

    def constants(self):
        R, V = self.em.raw, self.em.valu
        R("s_waitcnt lgkmcnt(0)")
        R("s_mov_b32 s%d, s56" % S_P)
        R("s_mov_b32 s%d, s57" % S_2P)
        if not SGPR_OPERANDS:
            V("v_mov_b32_e32 v%d, s56" % V_P)
            V("v_mov_b32_e32 v%d, s57" % V_2P)
        R("s_sub_u32 s%d, 0, s56" % S_NEGP)
        R("s_mov_b32 s%d, s58" % S_MU)
        R("s_mov_b32 s%d, s59" % S_NINV)
        R("s_mov_b32 s%d, s60" % S_NINVSH)
        R("s_mov_b32 s%d, s61" % S_W1N)
        R("s_mov_b32 s%d, s62" % S_W1NSH)
        R("s_load_dwordx16 s[40:55], s[22:23], 0x8")         # tw[1..8]
        R("s_load_dwordx16 s[56:71], s[22:23], 0x48")        # tw[9..16) (+ one record that is not used)

    # ---- row streams
    def row_io(self, base, ptr, store=False):
        """lane t <-> x[t + W q] in pair q (the immediate offset reaches 4095 bytes: the pointer steps every 4096)"""
        R = self.em.raw
        per = 4096 // (4 * self.W)
        R("s_mov_b64 s[76:77], s[%d:%d]" % (ptr, ptr + 1))
        for q in range(16):
            if q and q % per == 0:
                R("s_add_u32 s76, s76, 0x1000")
                R("s_addc_u32 s77, s77, 0")
            off = 4 * self.W * (q % per)
            if store:
                R("global_store_dword v%d, v%d, s[76:77] offset:%d" % (V_GOFF, base + 2 * q, off))
            else:
                R("global_load_dword v%d, v%d, s[76:77] offset:%d" % (base + 2 * q, V_GOFF, off))
                self.issued += 1
        return self.issued

    def lane16_io(self, block, ptr, store=False):
        """lane t <-> words 16 t .. 16 t + 15 (NTT-form order) in the 16 consecutive registers from `block`; V_TMP = 64 t"""
        R = self.em.raw
        for i in range(4):
            if store:
                R("global_store_dwordx4 v%d, v[%d:%d], s[%d:%d] offset:%d" % (V_TMP, block + 4 * i, block + 4 * i + 3, ptr, ptr + 1, 16 * i))
            else:
                R("global_load_dwordx4 v[%d:%d], v%d, s[%d:%d] offset:%d" % (block + 4 * i, block + 4 * i + 3, V_TMP, ptr, ptr + 1, 16 * i))
                self.issued += 1
        return self.issued

    def lane_tw(self, buf, index_expr, nrec):
        """per-lane records tw[first .. first + nrec) -> buf (ascending addresses; a descending pass indexes them from the
        top).  index_expr() leaves the lane's first record index in V_TWO.  Returns the number of the last load."""
        R, V = self.em.raw, self.em.valu
        index_expr()
        V("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_TWO, V_TWO))
        if nrec == 1:
            R("global_load_dwordx2 v[%d:%d], v%d, s[22:23]" % (buf, buf + 1, V_TWO))
            self.issued += 1
        else:
            for i in range(nrec // 2):
                R("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (buf + 4 * i, buf + 4 * i + 3, V_TWO, 16 * i))
                self.issued += 1
        return self.issued

    def wait(self, seq):
        self.em.raw("s_waitcnt vmcnt(%d)" % (self.issued - seq))

    # ---- twiddle index expressions (each leaves the lane's first record index in V_TWO)
    def idx_pass2(self, s):   # tw[((16 + B) << s) + g]
        V = self.em.valu
        def f():
            V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TWO, self.LG, V_LANE))
            V("v_add_u32_e32 v%d, 16, v%d" % (V_TWO, V_TWO))
            if s:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s, V_TWO))
        return f

    def idx_pass3(self, i):   # tw[(256 << i) + G t + g], G = 8 >> (NS3 - 1 - i)
        V = self.em.valu
        lgG = 3 - (self.NS3 - 1 - i)
        def f():
            if lgG:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, lgG, V_LANE))
                V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 256 << i, V_TWO))
            else:
                V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 256 << i, V_LANE))
        return f

    def idx_zeta(self):     # incomplete transforms, no stage of the last pass left: zeta = -+ tw[2^(LOGN - 3) + 2 t + g], g < 2
        V = self.em.valu
        V("v_lshlrev_b32_e32 v%d, 1, v%d" % (V_TWO, V_LANE))
        V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 1 << (self.LOGN - 3), V_TWO))

    def idx_inv1(self, i):    # tw[(512 << i) - 1 - (G t + g)], g < G: the block [(512 << i) - G (t + 1), +G)
        V = self.em.valu
        lgG = 3 - (self.NS3 - 1 - i)
        def f():
            V("v_add_u32_e32 v%d, 1, v%d" % (V_TWO, V_LANE))
            if lgG:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, lgG, V_TWO))
            V("v_sub_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 512 << i, V_TWO))
        return f

    def idx_inv2(self, s):    # tw[(32 << s) - 1 - ((B << s) + g)], g < 2^s: the block [(32 - B - 1) << s, +2^s)
        V = self.em.valu
        def f():
            V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TWO, self.LG, V_LANE))
            V("v_sub_u32_e32 v%d, 31, v%d" % (V_TWO, V_TWO))
            if s:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s, V_TWO))
        return f

    # ---- register stages and exchanges
    def stage16(self, bases, s, twf, op):
        """radix-2 stage s of a 16-point block: groups g < 2^s, butterflies (g 2 half + h, + half)"""
        half = 8 >> s
        jobs = []
        for g in range(1 << s):
            for h in range(half):
                for b in bases:
                    i0 = g * 2 * half + h
                    jobs.append(op(b + 2 * i0, b + 2 * (i0 + half), twf(g)))
        run(self.em, jobs)

    def row_sync(self):
        if self.LB > 4:
            self.em.raw("s_waitcnt lgkmcnt(0)")
            self.em.raw("s_barrier")

    def exchange(self, bases, waddr, woff, raddr, roff, sync_between=False, sync_before=False, src_off=0):
        """LDS exchange of the listed operands, one after the other.  sync_between: writers and readers are different
        waves of the row (workgroup barrier when the row has more than one wave); sync_before: the slab's previous readers
        were other waves too"""
        R = self.em.raw
        for n, b in enumerate(bases):
            if sync_before or (n and sync_between):
                self.row_sync()
            for q in range(16):
                R("ds_write_b32 v%d, v%d offset:%d" % (waddr, b + 2 * q + src_off, woff(q)))
            if sync_between:
                self.row_sync()
            for q in range(16):
                R("ds_read_b32 v%d, v%d offset:%d" % (b + 2 * q, raddr, roff(q)))
            R("s_waitcnt lgkmcnt(0)")

    def forward(self, bases, seq, keep3=None, after=None, first_sync=False):
        """the forward transform of the operands in `bases` together (shared twiddle registers and loads), x[t + W q] in pair q
        -> NTT word 16 t + q in pair q, lazily reduced (< 4p).  seq: the load of tw[16 + B] -> V_TWA, already requested; keep3:
        stages of the last pass that are run (incomplete transforms stop early); after(buffer index): what is requested into
        the free twiddle buffer in front of the last stage that runs, returns its load number; first_sync: the slab's previous
        readers may have been other waves of the row.  Returns (load number of `after`'s request or of the last prefetch, the
        buffer the last stage read its records from)."""
        NS3, bufs = self.NS3, self.bufs
        keep3 = NS3 if keep3 is None else keep3
        for s in range(4):
            self.stage16(bases, s, lambda g, s=s: sreg((1 << s) + g), ct)
        self.exchange(bases, V_A1, self.e1_row, V_A2, self.e1_blk, sync_between=True, sync_before=first_sync)
        for s in range(4):
            cur, cur_seq = bufs[s & 1], seq
            if s < 3:
                seq = self.lane_tw(bufs[(s + 1) & 1], self.idx_pass2(s + 1), 2 << s)
            elif keep3 > 0:
                seq = self.lane_tw(bufs[(s + 1) & 1], self.idx_pass3(0), 8 >> (NS3 - 1))
            else:
                seq = self.lane_tw(bufs[(s + 1) & 1], self.idx_zeta, 2)
            self.wait(cur_seq)
            self.stage16(bases, s, lambda g, cur=cur: vrec(cur, g), ct)
        self.exchange(bases, V_A2, self.e2_blk, V_A3, self.e2_thr, sync_before=True)
        # the last NS3 stages on the lane's 16 consecutive words: stage i has d = 1 << (NS3 - 1 - i), G = 8 / d groups
        for i in range(keep3):
            d = 1 << (NS3 - 1 - i)
            Gn = 8 // d
            cur, cur_seq = bufs[i & 1], seq
            if i + 1 < keep3:
                seq = self.lane_tw(bufs[(i + 1) & 1], self.idx_pass3(i + 1), 8 // (d // 2))
            elif after is not None:
                r = after((i + 1) & 1)
                seq = seq if r is None else r
            self.wait(cur_seq)
            jobs = []
            for g in range(Gn):
                for h in range(d):
                    for b in bases:
                        jobs.append(ct(b + 2 * (2 * d * g + h), b + 2 * (2 * d * g + h + d), vrec(cur, g)))
            run(self.em, jobs)
        return seq

    def inverse(self, seq, first_buf, inv1, first_src=None, from_high_exchange=False):
        """the inverse transform of the operand in V_A (inputs < 2p; NTT word 16 t + q in pair q -> x[t + W q], canonical, n^-1
        folded into the last stage) and its store.  seq / first_buf: the first stage's records (requested by the caller); inv1: the
        stages of the first pass that are run; first_src(x, y) -> (xsrc, ysrc): other registers the first stage reads from"""
        em, R, L = self.em, self.em.raw, self.em.lines.append
        NS3, bufs, one = self.NS3, self.bufs, [V_A]
        for k, i in enumerate(inv1):
            d = 1 << (NS3 - 1 - i)
            Gn = 8 // d
            cur, cur_seq = bufs[(first_buf + k) & 1], seq
            if i > 0:
                seq = self.lane_tw(bufs[(first_buf + k + 1) & 1], self.idx_inv1(i - 1), Gn // 2)
            else:
                seq = self.lane_tw(bufs[(first_buf + k + 1) & 1], self.idx_inv2(3), 8)
            self.wait(cur_seq)
            jobs = []
            for g in range(Gn):
                for h in range(d):
                    x, y = 2 * d * g + h, 2 * d * g + h + d
                    if first_src is not None and k == 0:
                        xs, ys = first_src(x, y)
                        jobs.append(gs(V_A + 2 * x, V_A + 2 * y, vrec(cur, Gn - 1 - g), xsrc=xs, ysrc=ys))
                    else:
                        jobs.append(gs(V_A + 2 * x, V_A + 2 * y, vrec(cur, Gn - 1 - g)))
            run(em, jobs)
        self.exchange(one, V_A3, self.e2_thr, V_A2, self.e2_blk, src_off=1 if from_high_exchange else 0)
        base_k = first_buf + len(inv1)
        for k, s in enumerate((3, 2, 1, 0)):
            cur, cur_seq = bufs[(base_k + k) & 1], seq
            n = 1 << s
            if s > 0:
                seq = self.lane_tw(bufs[(base_k + k + 1) & 1], self.idx_inv2(s - 1), n // 2)
            self.wait(cur_seq)
            self.stage16(one, s, lambda g, cur=cur, n=n: vrec(cur, n - 1 - g), gs)
        self.exchange(one, V_A2, self.e1_blk, V_A1, self.e1_row, sync_between=True, sync_before=True)
        for s in (3, 2, 1):                                                                              # uniform: tw[(2 << s) - 1 - g]
            self.stage16(one, s, lambda g, s=s: sreg((2 << s) - 1 - g), gs)
        run(em, [last(V_A + 2 * h, V_A + 2 * (h + 8)) for h in range(8)])
        if self.LB > 4:
            R("s_cmp_eq_u32 s21, 0")
            R("s_cbranch_scc1 .Ldone")
        self.row_io(V_A, 28, store=True)
        L(".Ldone:")
        R("s_endpgm")


def build(LB=4, mode="polymul", level=0):
    """mode: polymul (c = INTT(NTT(a) (.) NTT(b))) | fwd (c = NTT(a), canonical) | inv (c = INTT(a));
    level 2 (polymul): both forward transforms stop two stages early, base multiplication mod X^4 -+ zeta, the inverse starts two
    stages late (the host passes ModConst records with (n / 4)^-1 and floor(2^62 / p) - 2^32 in the mu field)"""
    assert level in (0, 2) and (not level or mode == "polymul")
    K = RowGen(LB)
    NS3, LOGN, bufs = K.NS3, K.LOGN, K.bufs
    KEEP3 = NS3 - level               # ... that remain when the transforms are incomplete
    em = K.em
    R = em.raw
    L = em.lines.append
    V = em.valu
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic = ceil(2^32 / nm) (0 when nm = 1)
    R("s_load_dwordx2 s[16:17], s[0:1], 0x30")           # rows
    K.lane_and_row()
    K.modulus_of_row()
    R("s_lshr_b32 s75, s19, %d" % (32 - (LOGN + 2)))
    R("s_lshl_b32 s74, s19, %d" % (LOGN + 2))            # row * n * 4 bytes
    for base, dst in ((6, 24), (8, 26), (4, 28)):
        R("s_add_u32 s%d, s%d, s74" % (dst, base))
        R("s_addc_u32 s%d, s%d, s75" % (dst + 1, base + 1))

    if mode == "inv":     # NTT-form input: lane t holds words 16 t .. 16 t + 15 (the b register block serves as load block)
        V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
        K.lane16_io(V_B, 24)
    else:
        K.row_io(V_A, 24)
        if mode == "polymul":
            K.row_io(V_B, 26)
    K.lds_addresses()
    K.constants()
    both = [V_A, V_B] if mode == "polymul" else [V_A]
    if mode != "inv":
        # ---------------------------------------------------------------- forward (both operands of a product together)
        seq = K.lane_tw(V_TWA, K.idx_pass2(0), 1)
        R("s_waitcnt vmcnt(1) lgkmcnt(0)")               # operands landed (the twiddle prefetch may still fly)
        after = None
        if mode == "polymul" and not level:
            after = lambda b: K.lane_tw(bufs[b], K.idx_inv1(NS3 - 1), 8)   # (first inverse stage, descending)
        seq = K.forward(both, seq, KEEP3, after)
    if mode == "fwd":
        # canonical words 16 t .. 16 t + 15 into a block of consecutive registers, four 16-byte stores per lane
        def canon(q):
            def gen(s):
                yield from csub(V_A + 2 * q, V_A + 2 * q, S_2P, s)
                yield from csub(V_A + 2 * q, V_B + q, S_P, s)
            return gen
        run(em, [canon(q) for q in range(16)])
        V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
        if LB > 4:
            R("s_cmp_eq_u32 s21, 0")
            R("s_cbranch_scc1 .Ldone")
        K.lane16_io(V_B, 28, store=True)
        L(".Ldone:")
        R("s_endpgm")
        return em, 4 * SLAB
    inv1 = list(range(NS3 - 1, -1, -1))           # stages of the inverse's first pass
    from_high = False                             # the inverse's first reader takes its inputs from the high halves of a's pairs
    first_src = None
    if mode == "polymul" and level:
        # ------------------------------------------------------------ base multiplication mod X^4 -+ zeta (see base_mul)
        zb = (KEEP3 - 1) & 1 if KEEP3 else 0      # the buffer that holds zeta: the last retained stage's records
        zeta_seq = seq
        inv1 = list(range(KEEP3 - 1, -1, -1))
        first_buf = zb ^ 1
        if inv1:
            seq = K.lane_tw(bufs[first_buf], K.idx_inv1(KEEP3 - 1), 8 // (1 << (NS3 - KEEP3)))
        else:
            seq = K.lane_tw(bufs[first_buf], K.idx_inv2(3), 8)
        K.wait(zeta_seq)
        run(em, [base_mul(g4, vrec(bufs[zb], g4 // 2), bool(g4 & 1)) for g4 in range(4)])
        from_high = True
        first_src = lambda x, y: (V_A + 2 * x + 1, V_A + 2 * y + 1)     # the base multiplication's results
    elif mode == "polymul":
        # ------------------------------------------------------------ point-wise product -> a, in [0, 2p)
        run(em, [pointwise(V_A + 2 * q, V_B + 2 * q) for q in range(16)])
        first_buf = NS3 & 1
    else:
        seq = K.lane_tw(bufs[0], K.idx_inv1(NS3 - 1), 8)
        R("s_waitcnt lgkmcnt(0)")
        first_buf = 0
        first_src = lambda x, y: (V_B + x, V_B + y)                     # the loaded words sit in the consecutive block
    # ---------------------------------------------------------------- inverse (one operand)
    K.inverse(seq, first_buf, inv1, first_src, from_high_exchange=from_high and not inv1)
    return em, 4 * SLAB


# ---------------------------------------------------------------- transform-fused pipelines (the bodies of the reference's LWE demo,
# tests/nfllib_demo_main_op.cpp:26-58, as ONE launch on rows this short; the 32-bit twins of tools/asmgen/rows1k.py build_row1k_fwd_fma /
# build_row1k_fma_inv, replacing the compiled k_row_fwd_fma / k_row_fma_inv of kernels_wave.hip where they cover the call)
S_EL, S_3P = 79, 79        # the batch element (prologue only) / 3p (fma_inv, after the prologue)
FUSED_SGPR = 96


def fma_fwd_job(x, e, k):
    """k <- (x * k + e) mod p, canonical: x, e lazily reduced (< 4p), k canonical, so T = x k + e < 4 p^2 + 4 p < 2^62 and the base
    multiplication's Barrett step applies (th = T >> 30, floor(2^62 / p) = 2^32 + m in the mu field of the level-2 records,
    q^ = th + mulhi(th, m), q - q^ <= 3, r < 4p); the accumulator is e's own register pair"""
    def gen(s):
        Q, TH = V_S[s] + 1, V_S[s] + 2
        yield "v_mov_b32_e32 v%d, 0" % (e + 1), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(e), S_DUMMY, x, k, pair(e)), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 30" % (TH, e + 1, e), None, None
        yield "v_mul_hi_u32 v%d, v%d, s%d" % (Q, TH, S_MU), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (Q, Q, TH), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(e), S_DUMMY, Q, S_NEGP, pair(e)), None, None
        yield from csub(e, e, S_2P, s)
        yield from csub(e, k, S_P, s)
    return gen


def fma_inv_job(dst, a, b, k, subtract):
    """dst (low half of its pair) <- b -+ a k in [0, 2p): canonical operands, exact Barrett (mu = floor(2^60 / p), r < 3p)"""
    def gen(s):
        Q, TH = V_S[s] + 1, V_S[s] + 2
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (pair(dst), S_DUMMY, a, k), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 28" % (TH, dst + 1, dst), None, None
        yield "v_mul_hi_u32 v%d, v%d, s%d" % (Q, TH, S_MU), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, s%d, %s" % (pair(dst), S_DUMMY, Q, S_NEGP, pair(dst)), None, None
        if subtract:
            yield "v_sub_u32_e32 v%d, v%d, v%d" % (dst, b, dst), None, None
            yield "v_add_u32_e32 v%d, s%d, v%d" % (dst, S_3P, dst), None, None
        else:
            yield "v_add_u32_e32 v%d, v%d, v%d" % (dst, dst, b), None, None
        yield from csub(dst, dst, S_2P, s)
    return gen


def build_fwd_fma(LB=4, two=True, fmt="i8"):
    """out0 = NTT(x) k0 + NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)], canonical NTT-form words: x is transformed once and stays in a's
    registers; each noise row is transformed in b's and multiply-added against the key row (words 16 t .. 16 t + 15, fetched into the
    twiddle buffer the last stage leaves free), the result leaves from the key's registers.  fmt "w": x, e are residue words
    ([nm][n] per polynomial); "i8": ONE signed byte per coefficient shared by the moduli (v < 0 stands for p + v) -- a lane fetches 16
    contiguous bytes (one instruction per wave covers the row's slice), the row passes through the slab into the first pass's lane map.
    kernarg: out0 out1 x psi mc | nm magic | k0 e0 k1 e1 | rows (64 bit) | strides x k0 e0 k1 e1 (in polynomials: 0 or 1); mc = the
    level-2 records (DevTables::mc_inc[1])"""
    assert fmt in ("w", "i8")
    K = RowGen(LB)
    em, R, V, L = K.em, K.em.raw, K.em.valu, K.em.lines.append
    LOGN, W, bufs = K.LOGN, K.W, K.bufs
    R("s_load_dwordx4 s[4:7], s[0:1], 0x0")              # out0, out1
    R("s_load_dwordx2 s[8:9], s[0:1], 0x10")             # x
    R("s_load_dwordx2 s[10:11], s[0:1], 0x18")           # psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic
    R("s_load_dwordx2 s[16:17], s[0:1], 0x50")           # rows
    R("s_load_dwordx8 s[80:87], s[0:1], 0x30")           # k0, e0, k1, e1
    R("s_load_dwordx4 s[88:91], s[0:1], 0x58")           # strides: x, k0, e0, k1
    R("s_load_dword s92, s[0:1], 0x68")                  # ... e1
    K.lane_and_row()
    K.modulus_of_row(keep_element=S_EL)

    def operand_row(ptr, stride, words):
        """s[ptr:ptr+1] += ((stride * el) * (words ? nm : 1) + (words ? cm : 0)) << log2(bytes per row)"""
        R("s_mul_i32 s77, s%d, s%d" % (stride, S_EL))
        if words:
            R("s_mul_i32 s77, s77, s14")
            R("s_add_u32 s77, s77, s20")
        K.advance(ptr, ptr, 77, LOGN + (2 if words else 0))
    operand_row(8, 88, fmt == "w")
    operand_row(82, 90, fmt == "w")
    operand_row(80, 89, True)
    if two:
        operand_row(86, 92, fmt == "w")
        operand_row(84, 91, True)
        K.advance(6, 6, 19, LOGN + 2)
    K.advance(4, 4, 19, LOGN + 2)

    def request(base, ptr):
        """start fetching an operand row: words go straight to their registers (x[t + W q] -> pair q); a compact row's 16 bytes per
        lane wait in the first four registers of the (idle) file"""
        if fmt == "w":
            return K.row_io(base, ptr)
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (V_TMP, V_LANE))
        R("global_load_dwordx4 v[%d:%d], v%d, s[%d:%d]" % (base, base + 3, V_TMP, ptr, ptr + 1))
        K.issued += 1
        return K.issued

    def land(base, seq, first):
        """the row in the first pass's lane map, as words any butterfly takes"""
        K.wait(seq)
        if fmt == "w":
            return
        if not first:
            K.row_sync()                                                          # (the other waves of the row may still read the slab)
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (V_TMP, V_LANE))
        V("v_add_u32_e32 v%d, s78, v%d" % (V_TMP, V_TMP))                         # slab + 16 t
        R("ds_write_b128 v%d, v[%d:%d]" % (V_TMP, base, base + 3))
        V("v_add_u32_e32 v%d, s78, v%d" % (V_TMP, V_LANE))                        # slab + t
        R("s_waitcnt lgkmcnt(0)")
        if K.WAVES > 1:
            R("s_barrier")
        for q in range(16):
            R("ds_read_i8 v%d, v%d offset:%d" % (base + 2 * q, V_TMP, W * q))     # byte t + W q, sign-extended
        R("s_waitcnt lgkmcnt(0)")
        def expand(r):
            def gen(s):   # v >= 0 stays (v + p > v), v < 0 becomes p + v (the sum wraps below v)
                yield "v_add_u32_e32 v%d, s%d, v%d" % (V_S[s], S_P, r), None, None
                yield "v_min_u32_e32 v%d, v%d, v%d" % (r, r, V_S[s]), None, None
            return gen
        run(em, [expand(base + 2 * q) for q in range(16)])
        K.row_sync()                                                              # (the slab is the transform's exchange buffer next)

    seq_x = request(V_A, 8)
    seq_e = request(V_B, 82)
    K.lds_addresses()
    K.constants()
    seq = K.lane_tw(V_TWA, K.idx_pass2(0), 1)
    land(V_A, seq_x, True)
    R("s_waitcnt lgkmcnt(0)")                            # (the wave-uniform twiddles)
    K.forward([V_A], seq)
    for h in range(2 if two else 1):
        em.comment("noise row %d" % h)
        seq = K.lane_tw(V_TWA, K.idx_pass2(0), 1)
        land(V_B, seq_e, False)
        key = {}
        def fetch_key(b, h=h):
            key["block"] = bufs[b]
            V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
            return K.lane16_io(bufs[b], 80 if h == 0 else 84)
        seq = K.forward([V_B], seq, after=fetch_key, first_sync=True)
        em.comment("out%d = X k%d + E%d against the key row (words 16 t .. 16 t + 15), canonical, stored from the key's registers" % (h, h, h))
        K.wait(seq)
        kb = key["block"]
        run(em, [fma_fwd_job(V_A + 2 * q, V_B + 2 * q, kb + q) for q in range(16)])
        if two and h == 0:
            seq_e = request(V_B, 86)                     # the second noise row is on its way while the first result leaves
        V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
        if LB > 4:
            R("s_cmp_eq_u32 s21, 0")
            R("s_cbranch_scc1 .Lskip%d" % h)
        K.lane16_io(kb, 4 if h == 0 else 6, store=True)
        L(".Lskip%d:" % h)
    R("s_endpgm")
    return em, 4 * SLAB


def build_fma_inv(LB=4, subtract=True):
    """c = INTT(b -+ a k): a, b dense NTT-form words, the key one polynomial for the batch (kstride 0) or one per element (1); the
    multiply-add happens in the registers the inverse transform starts from.
    kernarg: c a b psi mc | nm magic | rows (64 bit) | key | kstride"""
    K = RowGen(LB)
    em, R, V = K.em, K.em.raw, K.em.valu
    LOGN, NS3, bufs = K.LOGN, K.NS3, K.bufs
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic
    R("s_load_dwordx2 s[16:17], s[0:1], 0x30")           # rows
    R("s_load_dwordx2 s[80:81], s[0:1], 0x38")           # key
    R("s_load_dword s82, s[0:1], 0x40")                  # kstride
    K.lane_and_row()
    K.modulus_of_row()
    for base, dst in ((6, 24), (8, 26), (4, 28)):
        K.advance(dst, base, 19, LOGN + 2)
    R("s_cmp_eq_u32 s82, 0")
    R("s_cselect_b32 s77, s20, s19")                     # the key's row: modulus cm of ONE polynomial, or the row itself
    K.advance(80, 80, 77, LOGN + 2)
    V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
    K.lane16_io(V_B, 24)                                 # a: words 16 t .. 16 t + 15
    K.lane16_io(V_B + 16, 26)                            # b
    seq_k = K.lane16_io(V_TWB, 80)                       # key (the second twiddle buffer: free until the first inverse stage has run)
    K.lds_addresses()
    K.constants()
    if subtract:
        R("s_add_u32 s%d, s%d, s%d" % (S_3P, S_P, S_2P))
    seq = K.lane_tw(bufs[0], K.idx_inv1(NS3 - 1), 8)
    R("s_waitcnt lgkmcnt(0)")
    K.wait(seq_k)
    run(em, [fma_inv_job(V_A + 2 * q, V_B + q, V_B + 16 + q, V_TWB + q, subtract) for q in range(16)])
    K.inverse(seq, 0, list(range(NS3 - 1, -1, -1)))
    return em, 4 * SLAB


ARGS_FWD_FMA = [("ptr", 8 * i) for i in range(5)] + [("i32", 40), ("i32", 44)] + [("ptr", 48 + 8 * i) for i in range(4)] + [("i64", 80)] + \
               [("i32", 88 + 4 * i) for i in range(6)]
ARGS_FMA_INV = [("ptr", 8 * i) for i in range(5)] + [("i32", 40), ("i32", 44), ("i64", 48), ("ptr", 56), ("i32", 64), ("i32", 68)]


def args_yaml(spec):
    out = []
    for kind, off in spec:
        if kind == "ptr":
            out.append("      - {.address_space: global, .offset: %d, .size: 8, .value_kind: global_buffer}" % off)
        else:
            out.append("      - {.offset: %d, .size: %d, .value_kind: by_value}" % (off, 8 if kind == "i64" else 4))
    return "\n".join(out) + "\n"


def write_kernel(kname, out, em, lds, karg, args, sgpr):
    accum = (NEXT_VGPR + 3) // 4 * 4
    params = dict(k=kname, lds=lds, vgpr=NEXT_VGPR, sgpr=sgpr, accum=accum, sgprc=sgpr + 6, wg=256, karg=karg, args=args)
    with open(out, "w") as f:
        f.write("; GENERATED by tools/gen_row1024_u32_asm.py -- do not edit.\n")
        f.write(G.HEADER % params)
        f.write("\n".join(em.lines) + "\n")
        f.write(G.FOOTER % params)
    print("wrote %s: %d VALU instructions (static), %d lines" % (out, em.n_valu, len(em.lines)))


ARGS = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("ptr", 48)]


def main():
    for LB, n, mode in [(LB, n, mode) for LB, n in sorted(SHAPES.items()) for mode in ("polymul", "polymul_i2", "fwd", "inv")]:
        # "polymul_i2": the product on incomplete transforms (round 6); the complete product stays the cross-check and the A/B partner
        em, lds = build(LB, "polymul" if mode == "polymul_i2" else mode, 2 if mode == "polymul_i2" else 0)
        sfx = {"polymul": "", "polymul_i2": "_i2"}.get(mode, "_" + mode)
        kname = "nflhip_row%d%s_u32_asm" % (n, sfx)
        out = os.path.join(G.ROOT, "nfllib_amd", "csrc", "row%d%s_u32_gfx950.s" % (n, sfx))
        write_kernel(kname, out, em, lds, 56, G.args_yaml(ARGS).replace("{.address_space: global, .offset: 48, .size: 8, .value_kind: global_buffer}",
                                                                        "{.offset: 48, .size: 8, .value_kind: by_value}"), NEXT_SGPR)
    for LB, n in sorted(SHAPES.items()):
        for stem, two, fmt in (("enc2w", True, "w"), ("enc2i8", True, "i8"), ("fmafwdw", False, "w"), ("fmafwdi8", False, "i8")):
            em, lds = build_fwd_fma(LB, two, fmt)
            write_kernel("nflhip_row%d_%s_u32_asm" % (n, stem), os.path.join(G.ROOT, "nfllib_amd", "csrc", "row%d_%s_u32_gfx950.s" % (n, stem)),
                         em, lds, 112, args_yaml(ARGS_FWD_FMA), FUSED_SGPR)
        for stem, subtract in (("fmsinv", True), ("fmainv", False)):
            em, lds = build_fma_inv(LB, subtract)
            write_kernel("nflhip_row%d_%s_u32_asm" % (n, stem), os.path.join(G.ROOT, "nfllib_amd", "csrc", "row%d_%s_u32_gfx950.s" % (n, stem)),
                         em, lds, 72, args_yaml(ARGS_FMA_INV), FUSED_SGPR)


if __name__ == "__main__":
    main()
