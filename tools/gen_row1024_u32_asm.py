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


def build(LB=4, mode="polymul", level=0):
    """mode: polymul (c = INTT(NTT(a) (.) NTT(b))) | fwd (c = NTT(a), canonical) | inv (c = INTT(a));
    level 2 (polymul): both forward transforms stop two stages early, base multiplication mod X^4 -+ zeta, the inverse starts two
    stages late (the host passes ModConst records with (n / 4)^-1 and floor(2^62 / p) - 2^32 in the mu field)"""
    assert level in (0, 2) and (not level or mode == "polymul")
    W = 16 * LB                       # lanes per row
    LG = LB.bit_length() - 1          # log2 LB
    LOGN = 8 + LG
    NS3 = LG                          # stages of the last pass: 2, 3, 4
    KEEP3 = NS3 - level               # ... that remain when the transforms are incomplete
    WAVES = W // 64                   # waves per row
    em = G.Emitter()
    R = em.raw
    L = em.lines.append
    V = em.valu
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic = ceil(2^32 / nm) (0 when nm = 1)
    R("s_load_dwordx2 s[16:17], s[0:1], 0x30")           # rows
    V("v_and_b32_e32 v%d, %d, v%d" % (V_LANE, W - 1, V_TID))             # t: lane of the row
    V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_GOFF, V_LANE))
    V("v_readfirstlane_b32 s18, v%d" % V_TID)
    R("s_lshr_b32 s18, s18, %d" % (6 + (WAVES.bit_length() - 1)))        # row of the workgroup
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s19, s2, %d" % (2 - (WAVES.bit_length() - 1)))
    R("s_add_u32 s19, s19, s18")                         # row
    R("s_mov_b32 s21, 1")                                # store the result
    R("s_cmp_lt_u32 s19, s16")
    R("s_cbranch_scc1 .Llive")
    if LB == 4:
        R("s_endpgm")                                    # a surplus wave of the last workgroup (no workgroup barrier anywhere)
    else:
        R("s_sub_u32 s19, s16, 1")                       # a surplus row: walk through every barrier on the last row, store nothing
        R("s_mov_b32 s21, 0")
    L(".Llive:")
    R("s_mul_hi_u32 s20, s19, s15")
    R("s_mul_i32 s20, s20, s14")
    R("s_sub_u32 s20, s19, s20")                         # cm = row mod nm
    R("s_cmp_eq_u32 s14, 1")
    R("s_cselect_b32 s20, 0, s20")
    R("s_lshl_b32 s74, s20, %d" % (LOGN + 3))            # twiddles of the modulus: psi + cm * n * 8
    R("s_add_u32 s22, s10, s74")
    R("s_addc_u32 s23, s11, 0")
    R("s_mul_i32 s74, s20, 56")                          # its ModConst<u32> record
    R("s_add_u32 s74, s12, s74")
    R("s_addc_u32 s75, s13, 0")
    R("s_load_dwordx8 s[56:63], s[74:75], 0x0")          # p 2p mu ninv ninv_sh w1ninv w1ninv_sh beta
    R("s_lshr_b32 s75, s19, %d" % (32 - (LOGN + 2)))
    R("s_lshl_b32 s74, s19, %d" % (LOGN + 2))            # row * n * 4 bytes
    for base, dst in ((6, 24), (8, 26), (4, 28)):
        R("s_add_u32 s%d, s%d, s74" % (dst, base))
        R("s_addc_u32 s%d, s%d, s75" % (dst + 1, base + 1))

    def row_io(base, ptr, store=False):
        """lane t <-> x[t + W q] in pair q (the immediate offset reaches 4095 bytes: the pointer steps every 4096)"""
        per = 4096 // (4 * W)
        R("s_mov_b64 s[76:77], s[%d:%d]" % (ptr, ptr + 1))
        for q in range(16):
            if q and q % per == 0:
                R("s_add_u32 s76, s76, 0x1000")
                R("s_addc_u32 s77, s77, 0")
            off = 4 * W * (q % per)
            if store:
                R("global_store_dword v%d, v%d, s[76:77] offset:%d" % (V_GOFF, base + 2 * q, off))
            else:
                R("global_load_dword v%d, v%d, s[76:77] offset:%d" % (base + 2 * q, V_GOFF, off))

    if mode == "inv":     # NTT-form input: lane t holds words 16 t .. 16 t + 15 (the b register block serves as load block)
        V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TMP, V_LANE))
        for i in range(4):
            R("global_load_dwordx4 v[%d:%d], v%d, s[24:25] offset:%d" % (V_B + 4 * i, V_B + 4 * i + 3, V_TMP, 16 * i))
    else:
        row_io(V_A, 24)
        if mode == "polymul":
            row_io(V_B, 26)
    # LDS addresses of the row's slab: A1 = 4 t (+ 4 (W + LB) q), A2 = 4 ((W + LB) B + l) (+ 4 LB q [+ 4 (q >> (4 - lg LB))]),
    # A3 = 68 t (+ 4 q)
    R("s_mul_i32 s78, s18, %d" % (SLAB * WAVES))
    V("v_add_u32_e32 v%d, s78, v%d" % (V_A1, V_GOFF))
    V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TMP, LG, V_LANE))             # B
    V("v_and_b32_e32 v%d, %d, v%d" % (V_A2, LB - 1, V_LANE))              # l
    V("v_mov_b32_e32 v%d, %d" % (V_A3, W + LB))
    V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_A2, V_TMP, V_A3, V_A2))     # (W + LB) B + l
    V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_A2, V_A2))
    V("v_add_u32_e32 v%d, s78, v%d" % (V_A2, V_A2))
    V("v_mov_b32_e32 v%d, 68" % V_A3)
    V("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_A3, V_LANE, V_A3))           # 68 t
    V("v_add_u32_e32 v%d, s78, v%d" % (V_A3, V_A3))
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

    class VM:
        """counts vector-memory operations so that waits can name the one they need (in-order return)"""
        issued = 0
    vm = VM()

    def lane_tw(buf, index_expr, nrec):
        """per-lane records tw[first .. first + nrec) -> buf (ascending addresses; a descending pass indexes them from the
        top).  index_expr() leaves the lane's first record index in V_TWO.  Returns the number of the last load."""
        index_expr()
        V("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_TWO, V_TWO))
        if nrec == 1:
            R("global_load_dwordx2 v[%d:%d], v%d, s[22:23]" % (buf, buf + 1, V_TWO))
            vm.issued += 1
        else:
            for i in range(nrec // 2):
                R("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (buf + 4 * i, buf + 4 * i + 3, V_TWO, 16 * i))
                vm.issued += 1
        return vm.issued

    def wait(seq):
        R("s_waitcnt vmcnt(%d)" % (vm.issued - seq))

    def idx_pass2(s):   # tw[((16 + B) << s) + g]
        def f():
            V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TWO, LG, V_LANE))
            V("v_add_u32_e32 v%d, 16, v%d" % (V_TWO, V_TWO))
            if s:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s, V_TWO))
        return f

    def idx_pass3(i):   # tw[(256 << i) + G t + g], G = 8 >> (NS3 - 1 - i)
        lgG = 3 - (NS3 - 1 - i)
        def f():
            if lgG:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, lgG, V_LANE))
                V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 256 << i, V_TWO))
            else:
                V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 256 << i, V_LANE))
        return f

    def idx_zeta():     # incomplete transforms, no stage of the last pass left: zeta = -+ tw[2^(LOGN - 3) + 2 t + g], g < 2
        V("v_lshlrev_b32_e32 v%d, 1, v%d" % (V_TWO, V_LANE))
        V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 1 << (LOGN - 3), V_TWO))

    def idx_inv1(i):    # tw[(512 << i) - 1 - (G t + g)], g < G: the block [(512 << i) - G (t + 1), +G)
        lgG = 3 - (NS3 - 1 - i)
        def f():
            V("v_add_u32_e32 v%d, 1, v%d" % (V_TWO, V_LANE))
            if lgG:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, lgG, V_TWO))
            V("v_sub_u32_e32 v%d, 0x%x, v%d" % (V_TWO, 512 << i, V_TWO))
        return f

    def idx_inv2(s):    # tw[(32 << s) - 1 - ((B << s) + g)], g < 2^s: the block [(32 - B - 1) << s, +2^s)
        def f():
            V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_TWO, LG, V_LANE))
            V("v_sub_u32_e32 v%d, 31, v%d" % (V_TWO, V_TWO))
            if s:
                V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s, V_TWO))
        return f

    def stage16(bases, s, twf, op):
        """radix-2 stage s of a 16-point block: groups g < 2^s, butterflies (g 2 half + h, + half)"""
        half = 8 >> s
        jobs = []
        for g in range(1 << s):
            for h in range(half):
                for b in bases:
                    i0 = g * 2 * half + h
                    jobs.append(op(b + 2 * i0, b + 2 * (i0 + half), twf(g)))
        run(em, jobs)

    def row_sync():
        if LB > 4:
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")

    def exchange(bases, waddr, woff, raddr, roff, sync_between=False, sync_before=False, src_off=0):
        """LDS exchange of the listed operands, one after the other.  sync_between: writers and readers are different
        waves of the row (workgroup barrier when the row has more than one wave); sync_before: the slab's previous readers
        were other waves too"""
        for n, b in enumerate(bases):
            if sync_before or (n and sync_between):
                row_sync()
            for q in range(16):
                R("ds_write_b32 v%d, v%d offset:%d" % (waddr, b + 2 * q + src_off, woff(q)))
            if sync_between:
                row_sync()
            for q in range(16):
                R("ds_read_b32 v%d, v%d offset:%d" % (b + 2 * q, raddr, roff(q)))
            R("s_waitcnt lgkmcnt(0)")

    e1_row = lambda q: 4 * (W + LB) * q                  # x[t + W q]
    e1_blk = lambda q: 4 * LB * q                        # word LB q + l of block B
    e2_blk = lambda q: 4 * LB * q + 4 * (q >> (4 - LG))
    e2_thr = lambda q: 4 * q                             # word 16 t + q
    both = [V_A, V_B] if mode == "polymul" else [V_A]
    bufs = [V_TWA, V_TWB]
    one = [V_A]
    if mode != "inv":
        # ---------------------------------------------------------------- forward (both operands of a product together)
        seq = lane_tw(V_TWA, idx_pass2(0), 1)
        R("s_waitcnt vmcnt(1) lgkmcnt(0)")               # operands landed (the twiddle prefetch may still fly)
        for s in range(4):
            stage16(both, s, lambda g, s=s: sreg((1 << s) + g), ct)
        exchange(both, V_A1, e1_row, V_A2, e1_blk, sync_between=True)
        for s in range(4):
            cur, cur_seq = bufs[s & 1], seq
            if s < 3:
                seq = lane_tw(bufs[(s + 1) & 1], idx_pass2(s + 1), 2 << s)
            elif KEEP3 > 0:
                seq = lane_tw(bufs[(s + 1) & 1], idx_pass3(0), 8 >> (NS3 - 1))
            else:
                seq = lane_tw(bufs[(s + 1) & 1], idx_zeta, 2)
            wait(cur_seq)
            stage16(both, s, lambda g, cur=cur: vrec(cur, g), ct)
        exchange(both, V_A2, e2_blk, V_A3, e2_thr, sync_before=True)
        # the last NS3 stages on the lane's 16 consecutive words: stage i has d = 1 << (NS3 - 1 - i), G = 8 / d groups
        for i in range(KEEP3):
            d = 1 << (NS3 - 1 - i)
            Gn = 8 // d
            cur, cur_seq = bufs[i & 1], seq
            if i + 1 < KEEP3:
                seq = lane_tw(bufs[(i + 1) & 1], idx_pass3(i + 1), 8 // (d // 2))
            elif mode == "polymul" and not level:
                seq = lane_tw(bufs[(i + 1) & 1], idx_inv1(NS3 - 1), 8)   # (first inverse stage, descending)
            wait(cur_seq)
            jobs = []
            for g in range(Gn):
                for h in range(d):
                    for b in both:
                        jobs.append(ct(b + 2 * (2 * d * g + h), b + 2 * (2 * d * g + h + d), vrec(cur, g)))
            run(em, jobs)
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
        for i in range(4):
            R("global_store_dwordx4 v%d, v[%d:%d], s[28:29] offset:%d" % (V_TMP, V_B + 4 * i, V_B + 4 * i + 3, 16 * i))
        L(".Ldone:")
        R("s_endpgm")
        return em, 4 * SLAB
    inv1 = list(range(NS3 - 1, -1, -1))           # stages of the inverse's first pass
    from_high = False                             # the inverse's first reader takes its inputs from the high halves of a's pairs
    if mode == "polymul" and level:
        # ------------------------------------------------------------ base multiplication mod X^4 -+ zeta (see base_mul)
        zb = (KEEP3 - 1) & 1 if KEEP3 else 0      # the buffer that holds zeta: the last retained stage's records
        zeta_seq = seq
        inv1 = list(range(KEEP3 - 1, -1, -1))
        first_buf = zb ^ 1
        if inv1:
            seq = lane_tw(bufs[first_buf], idx_inv1(KEEP3 - 1), 8 // (1 << (NS3 - KEEP3)))
        else:
            seq = lane_tw(bufs[first_buf], idx_inv2(3), 8)
        wait(zeta_seq)
        run(em, [base_mul(g4, vrec(bufs[zb], g4 // 2), bool(g4 & 1)) for g4 in range(4)])
        from_high = True
    elif mode == "polymul":
        # ------------------------------------------------------------ point-wise product -> a, in [0, 2p)
        run(em, [pointwise(V_A + 2 * q, V_B + 2 * q) for q in range(16)])
        first_buf = NS3 & 1
    else:
        seq = lane_tw(bufs[0], idx_inv1(NS3 - 1), 8)
        R("s_waitcnt lgkmcnt(0)")
        first_buf = 0
    # ---------------------------------------------------------------- inverse (one operand)
    for k, i in enumerate(inv1):
        d = 1 << (NS3 - 1 - i)
        Gn = 8 // d
        cur, cur_seq = bufs[(first_buf + k) & 1], seq
        if i > 0:
            seq = lane_tw(bufs[(first_buf + k + 1) & 1], idx_inv1(i - 1), Gn // 2)
        else:
            seq = lane_tw(bufs[(first_buf + k + 1) & 1], idx_inv2(3), 8)
        wait(cur_seq)
        jobs = []
        for g in range(Gn):
            for h in range(d):
                x, y = 2 * d * g + h, 2 * d * g + h + d
                if mode == "inv" and k == 0:   # the loaded words sit in the consecutive block
                    jobs.append(gs(V_A + 2 * x, V_A + 2 * y, vrec(cur, Gn - 1 - g), xsrc=V_B + x, ysrc=V_B + y))
                elif from_high and k == 0:     # the base multiplication's results
                    jobs.append(gs(V_A + 2 * x, V_A + 2 * y, vrec(cur, Gn - 1 - g), xsrc=V_A + 2 * x + 1, ysrc=V_A + 2 * y + 1))
                else:
                    jobs.append(gs(V_A + 2 * x, V_A + 2 * y, vrec(cur, Gn - 1 - g)))
        run(em, jobs)
    exchange(one, V_A3, e2_thr, V_A2, e2_blk, src_off=1 if (from_high and not inv1) else 0)
    base_k = first_buf + len(inv1)
    for k, s in enumerate((3, 2, 1, 0)):
        cur, cur_seq = bufs[(base_k + k) & 1], seq
        n = 1 << s
        if s > 0:
            seq = lane_tw(bufs[(base_k + k + 1) & 1], idx_inv2(s - 1), n // 2)
        wait(cur_seq)
        stage16(one, s, lambda g, cur=cur, n=n: vrec(cur, n - 1 - g), gs)
    exchange(one, V_A2, e1_blk, V_A1, e1_row, sync_between=True, sync_before=True)
    for s in (3, 2, 1):                                                                              # uniform: tw[(2 << s) - 1 - g]
        stage16(one, s, lambda g, s=s: sreg((2 << s) - 1 - g), gs)
    run(em, [last(V_A + 2 * h, V_A + 2 * (h + 8)) for h in range(8)])
    if LB > 4:
        R("s_cmp_eq_u32 s21, 0")
        R("s_cbranch_scc1 .Ldone")
    row_io(V_A, 28, store=True)
    L(".Ldone:")
    R("s_endpgm")
    return em, 4 * SLAB


ARGS = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("ptr", 48)]


def main():
    for LB, n, mode in [(LB, n, mode) for LB, n in sorted(SHAPES.items()) for mode in ("polymul", "polymul_i2", "fwd", "inv")]:
        # "polymul_i2": the product on incomplete transforms (round 6); the complete product stays the cross-check and the A/B partner
        em, lds = build(LB, "polymul" if mode == "polymul_i2" else mode, 2 if mode == "polymul_i2" else 0)
        sfx = {"polymul": "", "polymul_i2": "_i2"}.get(mode, "_" + mode)
        kname = "nflhip_row%d%s_u32_asm" % (n, sfx)
        out = os.path.join(G.ROOT, "nfllib_amd", "csrc", "row%d%s_u32_gfx950.s" % (n, sfx))
        karg = 56
        accum = (NEXT_VGPR + 3) // 4 * 4
        params = dict(k=kname, lds=lds, vgpr=NEXT_VGPR, sgpr=NEXT_SGPR, accum=accum, sgprc=NEXT_SGPR + 6, wg=256,
                      karg=karg, args=G.args_yaml(ARGS).replace("{.address_space: global, .offset: 48, .size: 8, .value_kind: global_buffer}",
                                                                  "{.offset: 48, .size: 8, .value_kind: by_value}"))
        with open(out, "w") as f:
            f.write("; GENERATED by tools/gen_row1024_u32_asm.py -- do not edit.\n")
            f.write(G.HEADER % params)
            f.write("\n".join(em.lines) + "\n")
            f.write(G.FOOTER % params)
        print("wrote %s: %d VALU instructions (static), %d lines" % (out, em.n_valu, len(em.lines)))


if __name__ == "__main__":
    main()
