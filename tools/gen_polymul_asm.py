#!/usr/bin/env python3
"""Generate nfllib_amd/csrc/polymul4096_gfx950.s -- the hand-scheduled gfx950 assembly
version of the metric kernel (fused c = INTT(NTT(a) (.) NTT(b)), u64, n = 4096).

Same algorithm, thread mapping, LDS layout and device tables as k_polymul4096 in
kernels_fast.hip (read that file's header first); what the generator adds over hipcc:

  * every v_mad_u64_u32 addend pair is placed by construction (even-aligned VGPR pairs,
    a persistent zero register behind the one zero-extended operand), so the ~1000
    v_mov copies per wave that hipcc needs to build those pairs disappear;
  * the 64-bit accumulate chains write straight into the coefficient registers
    (no result moves), the Shoup low word and the "- (q << 62)" term are one
    four-instruction v_mad_u64_u32 chain + one v_add_u32;
  * two independent butterflies are interleaved instruction by instruction, which
    covers the 2 wait states gfx950 needs between a VALU SGPR write (carry / borrow)
    and the VALU that consumes it; a hazard tracker inserts s_nop where it does not;
  * a's and b's forward passes share each pass's twiddle registers.

Run:  python tools/gen_polymul_asm.py   (writes the .s; nfllib_amd/csrc/Makefile assembles it
with clang -x assembler -mcpu=gfx950, links it with ld.lld and embeds the code object).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nfllib_amd", "csrc", "polymul4096_gfx950.s")
KNAME = "nflhip_polymul4096_asm"

# ------------------------------------------------------------------ register map
# SGPRs
S_KARG = "s[0:1]"
S_WGX, S_WGY = "s2", "s3"           # workgroup ids: x = poly index, y = modulus index
S_C, S_A, S_B, S_PSI, S_MC = "s[4:5]", "s[6:7]", "s[8:9]", "s[10:11]", "s[12:13]"
S_NM = "s14"
S_AROW, S_BROW, S_CROW, S_TW = "s[16:17]", "s[18:19]", "s[20:21]", "s[22:23]"
S_P, S_P2, S_P3 = "s[24:25]", "s[26:27]", "s[28:29]"
S_DELTA, S_MASK, S_C0 = "s30", "s31", "s15"      # delta, 0x3fffffff, 0xC0000000
S_MU2 = (32, 33)
S_NINV, S_NINVSH, S_W1N, S_W1NSH = (34, 35), (36, 37), (38, 39), (40, 41)
S_TMP = "s[42:43]"                   # scalar address scratch
S_CARRY = ["s[44:45]", "s[46:47]"]   # v_mad_u64_u32 carry-out, per stream
S_DUMMY = "s[48:49]"                 # dead carry-outs
S_BORROW = ["s[50:51]", "s[52:53]"]  # v_sub_co borrow, per stream
S_MCBUF = 56                         # s[56:83]: the ModConst record (28 dwords)
S_BASE2 = "s[84:85]"                 # scalar base of the current twiddle loads
S_R, S_BLK = "s88", "s89"            # r = logn - 12, blk = index of this 4096-word block inside its row
# K of each pass: twiddle index = (K << s) + (lane << s) + g (forward) / (K << s) - 1 - (lane << s) - g (inverse)
S_K = {"F1": "s90", "F2": "s91", "F3": "s92", "I1": "s93", "I2": "s94", "I3": "s95"}
NEXT_SGPR = 96

# VGPRs
V_TID = 0
V_OFF8 = 1        # tid*8 (global row offset)
V_L1W = 2         # LDS byte address, E1 write / E1' read : (t + (t>>4))*8
V_L1R = 3         # LDS byte address, E1 read / E2 write / E2' read / E1' write : (272*B + r)*8
V_L2R = 4         # LDS byte address, E2 read / E2' write : 17*t*8
V_BIDX = 5        # B = t >> 4
V_PHI = 6         # high dword of p (v_subb needs it in a VGPR)
V_A = 8           # v[8:39]    : a  (16 even-aligned pairs)
V_B = 40          # v[40:71]   : b
V_TW = 72         # v[72:131]  : 15 twiddle records (w lo, w hi, w' lo, w' hi)
V_T = [132, 150]  # per-stream temporaries (18 regs each)
NEXT_VGPR = 168   # 3 waves per SIMD
# address scratch lives in stream 1's temporaries (idle between butterflies)
V_TWO = V_T[1] + 1      # 32-bit per-lane twiddle offset
V_TWA = V_T[1] + 4      # 64-bit per-lane twiddle address
V_ZERO = V_T[0] + 15    # a persistent zero (the high half of stream 0's ZP pair)

LDS_BYTES = (4096 + 256) * 8


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(pair):
    return "s[%d:%d]" % pair


# Power / time ablations of the product kernels (tools/sessions/gpu_round3_g.sh; the results are WRONG by construction, the
# instruction stream is otherwise the shipped one): NFL_GEN_ABLATE = comma list of
#   tw0    every lane fetches the twiddle record of lane 0 (one cache line per wave instead of up to 64)
#   nolds  the exchanges through LDS are dropped (barriers stay)
#   row0   every workgroup works on one of the first 16 rows (operands and result stay in the L2)
#   nobar  the workgroup barriers are dropped as well
#   nobfly the butterflies of the register passes are dropped (memory, LDS and the point-wise step remain)
ABLATE = set(filter(None, os.environ.get("NFL_GEN_ABLATE", "").split(",")))
# scratchN (N a power of two): the n = 65536 pipeline's scratch rows a', b' of the WHOLE batch aliased onto N rows, i.e. the
# forward pass's writes and the block products' reads served by the on-die caches instead of HBM (round 5: what is the
# prize of a plan whose scratch never leaves the chip?)
SCRATCH_ALIAS = next((int(x[7:]) for x in ABLATE if x.startswith("scratch")), 0)
# bprimeN: the same question for rows of 32768 words (workload F): b' = NTT(b) makes a round trip through the context's
# scratch between the two launches of the composed product -- here over N row blocks instead of one per row
BPRIME_ALIAS = next((int(x[6:]) for x in ABLATE if x.startswith("bprime")), 0)
ALIAS_ROWS = ()   # set by build_row32k for its "_s" kinds: which of the row pointers s16 / s18 / s20 prologue16k aliases


class Emitter:
    """Collects instructions, counts VALU work and pads the gfx950
    'VALU writes SGPR -> VALU reads that SGPR: 2 wait states' hazard."""

    def __init__(self):
        self.lines = []
        self.pos = 0
        self.last_swrite = {}
        self.n_valu = 0
        self.n_nop = 0

    def raw(self, text):
        if ("nolds" in ABLATE and text.startswith("ds_")) or ("nobar" in ABLATE and text.startswith("s_barrier")):
            return
        self.lines.append("\t" + text)
        self.pos += 1

    def comment(self, text):
        self.lines.append("\t; " + text)

    def valu(self, text, wr=None, rd=None):
        if SWAP_MULHI and text.startswith("v_mul_hi_u32 "):   # experiment: the same question for v_mul_hi_u32
            ops = [o.strip() for o in text[len("v_mul_hi_u32 "):].split(",")]
            if len(ops) == 3:
                text = "v_mul_hi_u32 %s, %s, %s" % (ops[0], ops[2], ops[1])
        if SWAP_MAD != "0" and text.startswith("v_mad_u64_u32 "):
            # The butterfly code is written "data x constant" (v_mad_u64_u32 D, carry, data, twiddle-or-constant, addend); what is
            # EMITTED is "constant x data": same result, same issue cost, and 1.1 - 2.3 % more products/s on the metric kernel --
            # the kernels run at the package power limit and the multiplier draws less with the sparse constants (delta < 2^27,
            # the 2^62 term) in its first operand (same-box A/B, equal checksums: profiles/r03_mad_operand_order.txt; exchanging
            # only the SGPR-constant ones +1.8 %, only the twiddle ones +0.4 %, all +2.3 %).  NFL_GEN_SWAP_MAD=0 / sgpr / vgpr.
            ops = [o.strip() for o in text[len("v_mad_u64_u32 "):].split(",")]
            # operands: vdst ("v[a:b]"), sdst ("s[a:b]"), src0, src1, src2
            sgpr = ops[3].startswith("s") if len(ops) == 5 else False
            if len(ops) == 5 and (SWAP_MAD in ("", "1") or (SWAP_MAD == "sgpr" and sgpr) or (SWAP_MAD == "vgpr" and not sgpr)):
                ops[2], ops[3] = ops[3], ops[2]
                text = "v_mad_u64_u32 " + ", ".join(ops)
        if rd is not None and rd in self.last_swrite:
            gap = self.pos - self.last_swrite[rd]
            if gap < 3:
                need = 3 - gap
                self.lines.append("\ts_nop %d" % (need - 1))
                self.pos += need
                self.n_nop += 1
        self.lines.append("\t" + text)
        if wr is not None:
            self.last_swrite[wr] = self.pos
        self.pos += 1
        self.n_valu += 1


def interleave(em, gens):
    """Round-robin the instruction streams of independent butterflies."""
    gens = list(gens)
    while gens:
        for g in list(gens):
            try:
                text, wr, rd = next(g)
                em.valu(text, wr, rd)
            except StopIteration:
                gens.remove(g)


SINGLE_STREAM = False   # ring mode: one butterfly at a time (18 temporaries instead of 36)
RING_RECOMPUTE_TWA = False


def run_pairs(em, jobs):
    """jobs: list of callables(stream) -> generator; executed two at a time, interleaved."""
    if SINGLE_STREAM:
        for j in jobs:
            interleave(em, [j(0)])
        return
    for i in range(0, len(jobs), 2):
        gens = [jobs[i](0)]
        if i + 1 < len(jobs):
            gens.append(jobs[i + 1](1))
        interleave(em, gens)


# ------------------------------------------------------------------ arithmetic building blocks
# temporaries of stream s (base T = V_T[s], all pairs even-aligned):
#   T+0       t      scratch dword
#   T+2,+3    P      (sum >> 32 | carry << 32) addend pair
#   T+4,+5    U / D  folded x, 2U+3p (CT)  /  difference (GS, final, point-wise)
#   T+6,+7    A      cross-product accumulator
#   T+8,+9    Q      quotient
#   T+10,+11  H      high-word accumulator (low dword used)
#   T+12,+13  E      sum / 2p+y
#   T+14,+15  ZP     [mul_hi result, 0]  (T+15 is zeroed once and never written again)
#   T+16,+17  L      point-wise low product

def T(s, k):
    return V_T[s] + k


def quotient(s, y, tw, exact):
    """Q = floor(y*w'/2^64) (exact) or that minus e, e in {0,1} (not exact). y = VGPR pair base of
    the multiplicand; tw = (w0, w1, a0, a1) operand strings (VGPR or SGPR)."""
    w0, w1, a0, a1 = tw
    A, P, Q, ZP = T(s, 6), T(s, 2), T(s, 8), T(s, 14)
    if exact:
        yield "v_mul_hi_u32 v%d, v%d, %s" % (ZP, y, a0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), S_DUMMY, y, a1, vp(ZP)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), S_CARRY[s], y + 1, a0, vp(A)), S_CARRY[s], None
    else:
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(A), S_DUMMY, y + 1, a0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), S_CARRY[s], y, a1, vp(A)), S_CARRY[s], None
    yield "v_mov_b32_e32 v%d, v%d" % (P, A + 1), None, None
    yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, S_DUMMY, S_CARRY[s]), None, S_CARRY[s]
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), S_DUMMY, y + 1, a1, vp(P)), None, None


def lowchain(s, y, tw, acc, seed, after_low=None):
    """acc = seed + y*w - Q*p (mod 2^64) using p = 2^62 - delta: y*w + Q*delta - (Q << 62).
    The high-dword terms are accumulated first; after_low is an instruction that needs only
    acc's LOW dword and may overwrite y's low dword (slotted in once both are settled)."""
    w0, w1, a0, a1 = tw
    Q, H = T(s, 8), T(s, 10)
    yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), S_DUMMY, y, w1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), S_DUMMY, y + 1, w0, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), S_DUMMY, Q + 1, S_DELTA, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), S_DUMMY, Q, S_C0, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(acc), S_DUMMY, y, w0, seed), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(acc), S_DUMMY, Q, S_DELTA, vp(acc)), None, None
    if after_low is not None:
        yield after_low
    yield "v_add_u32_e32 v%d, v%d, v%d" % (acc + 1, acc + 1, H), None, None


def v_mask():
    """VGPR holding 0x3fffffff: the one temporary slot (T + 1 of stream 0) no butterfly uses"""
    return V_T[0] + 1


def fold2(s, dst, src):
    """dst = (src & (2^62-1)) + (src >> 62)*delta  (< 2^62 + 3*delta); clobbers src's high dword."""
    t = T(s, 0)
    yield "v_lshrrev_b32_e32 v%d, 30, v%d" % (t, src + 1), None, None
    # (the mask stays in an SGPR.  The isolated streams of tools/ubench_issue.hip price a plain VOP2 op with an SGPR operand
    # at 4.4 cycles and an all-VGPR one at 2.5, but IN the metric kernel the two forms are the same to 0.2 % (3.10 ms per
    # launch either way, same box, same checksums: profiles/r03_operand_ab.txt); NFL_GEN_VGPR_OPERANDS=1 rebuilds the other)
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        yield "v_and_b32_e32 v%d, v%d, v%d" % (src + 1, v_mask(), src + 1), None, None
    else:
        yield "v_and_b32_e32 v%d, %s, v%d" % (src + 1, S_MASK, src + 1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(dst), S_DUMMY, t, S_DELTA, vp(src)), None, None


def ct_bfly(x, y, tw):
    """Cooley-Tukey: x' = x + w*y, y' = x - w*y (any 64-bit words in, any 64-bit words out)."""
    def gen(s):
        if "nobfly" in ABLATE:
            return
        U, Y2 = T(s, 4), T(s, 12)
        yield from fold2(s, U, x)
        yield from quotient(s, y, tw, exact=False)
        yield "v_lshl_add_u64 %s, %s, 1, %s" % (vp(Y2), vp(U), S_P3), None, None
        # x' = U + m (m < 3p) lands in x; y' = (2U + 3p) - x'.  The low-dword subtract is issued as soon as
        # x' low is final, so its borrow is old enough when v_subb consumes it (no hazard nop).
        sub_lo = ("v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (y, S_BORROW[s], Y2, x), S_BORROW[s], None)
        yield from lowchain(s, y, tw, x, vp(U), after_low=sub_lo)
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (y + 1, S_DUMMY, Y2 + 1, x + 1, S_BORROW[s]), None, S_BORROW[s]
    return gen


def gs_bfly(x, y, tw):
    """Gentleman-Sande with the negated mirrored twiddle: x' = fold(x + y), y' = (y - x)*w; inputs < 2p."""
    def gen(s):
        if "nobfly" in ABLATE:
            return
        E, D, SUM = T(s, 12), T(s, 4), T(s, 16)
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(y), S_P2), None, None
        yield "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (D, S_BORROW[s], E, x), S_BORROW[s], None
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(SUM), vp(x), vp(y)), None, None
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (D + 1, S_DUMMY, E + 1, x + 1, S_BORROW[s]), None, S_BORROW[s]
        yield from fold2(s, x, SUM)
        yield from quotient(s, D, tw, exact=True)
        yield from lowchain(s, D, tw, y, "0")
    return gen


def csub_p(s, reg):
    """reg = reg >= p ? reg - p : reg  (borrow trick)."""
    E = T(s, 12)
    yield "v_sub_co_u32_e64 v%d, %s, v%d, %s" % (E, S_BORROW[s], reg, "s24"), S_BORROW[s], None
    # subb with an SGPR subtrahend needs it in src0 of the *rev* form: use a VGPR copy of p's high dword
    yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (E + 1, S_BORROW[s], reg + 1, V_PHI, S_BORROW[s]), S_BORROW[s], S_BORROW[s]
    yield "v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (reg, E, reg, S_BORROW[s]), None, S_BORROW[s]
    yield "v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (reg + 1, E + 1, reg + 1, S_BORROW[s]), None, S_BORROW[s]




def canon(reg):
    """any 64-bit word -> canonical [0,p): two-bit fold (< p + 4*delta) then one conditional subtract."""
    def gen(s):
        yield from fold2(s, reg, reg)
        yield from csub_p(s, reg)
    return gen


def final_bfly(x, y):
    """Last inverse stage with n^-1 folded in; canonical outputs."""
    tw_n = ("s%d" % S_NINV[0], "s%d" % S_NINV[1], "s%d" % S_NINVSH[0], "s%d" % S_NINVSH[1])
    tw_w = ("s%d" % S_W1N[0], "s%d" % S_W1N[1], "s%d" % S_W1NSH[0], "s%d" % S_W1NSH[1])

    def gen(s):
        E, D = T(s, 12), T(s, 4)
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(y), S_P2), None, None
        yield "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (D, S_BORROW[s], E, x), S_BORROW[s], None
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (D + 1, S_DUMMY, E + 1, x + 1, S_BORROW[s]), None, S_BORROW[s]
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(x), vp(y)), None, None
        yield from quotient(s, E, tw_n, exact=True)
        yield from lowchain(s, E, tw_n, x, "0")
        yield from csub_p(s, x)
        yield from quotient(s, D, tw_w, exact=True)
        yield from lowchain(s, D, tw_w, y, "0")
        yield from csub_p(s, y)
    return gen


def pointwise(xa, xb, fold_a=True, fold_b=True):
    """xa = fold2(xa*xb mod p) with lazily reduced operands (mul_lazy of kernels_fast.hip);
    an operand known to be canonical (< p) skips its fold."""
    mu0, mu1 = "s%d" % S_MU2[0], "s%d" % S_MU2[1]

    def gen(s):
        L, A, P, Q, H, E, ZP = T(s, 16), T(s, 6), T(s, 2), T(s, 8), T(s, 10), T(s, 12), T(s, 14)
        if fold_a:
            yield from fold2(s, xa, xa)
        if fold_b:
            yield from fold2(s, xb, xb)
        # T = xa*xb as four dwords: T0 = L.lo, T1 = A.lo, T2 = E.lo, T3 = E.hi
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(L), S_DUMMY, xa, xb), None, None
        yield "v_mov_b32_e32 v%d, v%d" % (ZP, L + 1), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), S_DUMMY, xa, xb + 1, vp(ZP)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), S_CARRY[s], xa + 1, xb, vp(A)), S_CARRY[s], None
        yield "v_mov_b32_e32 v%d, v%d" % (P, A + 1), None, None
        yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, S_DUMMY, S_CARRY[s]), None, S_CARRY[s]
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(E), S_DUMMY, xa + 1, xb + 1, vp(P)), None, None
        # th = T >> 61 -> D pair
        D = T(s, 4)
        yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D, E, A), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D + 1, E + 1, E), None, None
        # q ~ floor(th*mu2/2^64), one-off allowed (r < 4p, folded below)
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), S_DUMMY, D + 1, mu0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), S_CARRY[s], D, mu1, vp(H)), S_CARRY[s], None
        yield "v_mov_b32_e32 v%d, v%d" % (P, H + 1), None, None
        yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, S_DUMMY, S_CARRY[s]), None, S_CARRY[s]
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), S_DUMMY, D + 1, mu1, vp(P)), None, None
        # r = lo64(T) + q*delta - (q << 62)
        yield "v_mov_b32_e32 v%d, v%d" % (L + 1, A), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(L), S_DUMMY, Q, S_DELTA, vp(L)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), S_DUMMY, Q + 1, S_DELTA), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), S_DUMMY, Q, S_C0, vp(H)), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (L + 1, L + 1, H), None, None
        yield from fold2(s, xa, L)
    return gen


# ------------------------------------------------------------------ passes
def twreg(i):
    b = V_TW + 4 * i
    return ("v%d" % b, "v%d" % (b + 1), "v%d" % (b + 2), "v%d" % (b + 3))


def tw_slot(s, g):
    return (1 << s) - 1 + g          # 15 records: sub-stage s (0..3), group g (0..2^s-1)


class VmCounter:
    """In-order VMEM load bookkeeping for counted s_waitcnt vmcnt(N)."""

    def __init__(self, em):
        self.em = em
        self.issued = 0

    def load(self, text):
        self.em.raw(text)
        self.issued += 1
        return self.issued

    def wait(self, seq):
        """Block until load number `seq` (and every earlier one) has landed."""
        n = self.issued - seq
        assert 0 <= n
        self.em.raw("s_waitcnt vmcnt(%d)" % min(n, 63))


def ct_stage(em, bases, s):
    if "nobfly" in ABLATE:
        return
    half = 8 >> s
    jobs = []
    for g in range(1 << s):
        tw = twreg(tw_slot(s, g))
        for h in range(half):
            i0 = g * 2 * half + h
            for base in bases:
                jobs.append(ct_bfly(base + 2 * i0, base + 2 * (i0 + half), tw))
    run_pairs(em, jobs)


def gs_stage(em, base, s):
    if "nobfly" in ABLATE:
        return
    half = 8 >> s
    jobs = []
    for g in range(1 << s):
        tw = twreg(tw_slot(s, g))
        for h in range(half):
            i0 = g * 2 * half + h
            jobs.append(gs_bfly(base + 2 * i0, base + 2 * (i0 + half), tw))
    run_pairs(em, jobs)


def tw_base(em, kreg, s, descending, koff=0):
    """s[84:85] = tw + 16 * (((K + koff) << s) [- 1])"""
    if koff:
        em.raw("s_%s_u32 s86, %s, 0x%x" % ("add" if koff > 0 else "sub", kreg, abs(koff)))
        kreg = "s86"
    em.raw("s_lshl_b32 s86, %s, %d" % (kreg, s))
    if descending:
        em.raw("s_sub_u32 s86, s86, 1")
    em.raw("s_lshl_b32 s86, s86, 4")
    em.raw("s_add_u32 s84, s22, s86")
    em.raw("s_addc_u32 s85, s23, 0")


# In the ring-mode kernels (rows of 8192 / 16384 / 32768 words: one butterfly at a time, twiddle records streamed through
# the ring) the passes whose twiddle index depends on the THREAD (F3 / I1: global stages logn-4 .. logn-1, 15/16 of the
# table) read a lane-major copy of those stages: stage S = logn-4+s holds M 2^s records (M = n/16 = 256 << r), natural position
# (u << s) + g for thread-index u = 256 blk + t and group g, lane-major position g M + u.  A wave's 64 lanes then fetch 64
# CONSECUTIVE records per load (8 cache lines, all bytes used) instead of 64 records 16 << s bytes apart (up to 64 lines,
# 16 bytes used of each: 5.7 x the L2 -> L1 traffic over a pass).  The host lays the copy out (api.hip build_tables,
# DevTables::psi_lm); every other pass reads indices below n/16, which both layouts share.
#   ascending  (F3): index = K + ((256 c) << r) + t,          c  = 2^s - 1 + g            (K = 256 (2^r + blk))
#   descending (I1): index = K + ((256 c') << r) - 1 - t,     c' = 2^(s+1) - 2 - g        (K = (512 << r) - 256 blk)
# The lane part is the same for every stage: V_TWO = 16 t (ascending) or 16 (255 - t) (descending, base lowered by 256).
# Same-box A/B against the natural order (profiles/r03_lane_major_twiddles.txt): products +1 % (16384) / +3 % (8192) / +5 %
# (32768), inverse transforms +6 ... +16 %, forward +2 ... +8 %.  The 4096-word kernels (three workgroups per CU, all 15
# records of a pass resident) gain nothing from it (product +-0, pre-transformed product -2 %) and keep the natural table.
LANE_MAJOR = not os.environ.get("NFL_GEN_NATURAL_TWIDDLES")
SWAP_MAD = os.environ.get("NFL_GEN_SWAP_MAD", "")   # "" / "1" all (shipped), "0" none, "sgpr" / "vgpr": only the multiply-adds whose second factor is an SGPR / a VGPR
SWAP_MULHI = bool(os.environ.get("NFL_GEN_SWAP_MULHI"))
SPLIT32K = not os.environ.get("NFL_GEN_SERIAL_EXCHANGE")   # build_row32k: exchanges of one file under the arithmetic of the other


def tw_base_lm(em, kreg, s, g, descending, koff=0):
    c = (2 << s) - 2 - g if descending else (1 << s) - 1 + g
    if c:
        em.raw("s_lshl_b32 s86, 0x%x, %s" % (256 * c, S_R))
        em.raw("s_add_u32 s86, s86, %s" % kreg)
    else:
        em.raw("s_mov_b32 s86, %s" % kreg)
    k = koff - (256 if descending else 0)
    if k:
        em.raw("s_%s_u32 s86, s86, 0x%x" % ("add" if k > 0 else "sub", abs(k)))
    em.raw("s_lshl_b32 s86, s86, 4")
    em.raw("s_add_u32 s84, s22, s86")
    em.raw("s_addc_u32 s85, s23, 0")


def tw_lane_offset_lm(em, descending):
    em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (V_TWO, V_TID))
    if descending:
        em.valu("v_sub_u32_e32 v%d, 0xff0, v%d" % (V_TWO, V_TWO))
    if "tw0" in ABLATE:
        em.valu("v_mov_b32_e32 v%d, 0" % (V_TWO,))


def tw_uniform_stage(em, vm, s, kreg, descending):
    """Twiddle records of sub-stage s at wave-uniform indices (K << s) + g  /  (K << s) - 1 - g."""
    tw_base(em, kreg, s, descending)
    seq = 0
    for g in range(1 << s):
        r = V_TW + 4 * tw_slot(s, g)
        off = -g * 16 if descending else g * 16
        seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_ZERO, S_BASE2, off))
    return seq


def tw_lane_stage(em, vm, s, vidx, kreg, descending, groups=None):
    """Per-lane twiddle records of sub-stage s.  Ascending (forward): index = (K << s) + (vidx << s) + g.
    Descending (inverse, mirrored): index = (K << s) - 1 - (vidx << s) - g.  vidx: VGPR with B or t.
    groups: only these g (default: all 2^s)"""
    seq = 0
    groups = range(1 << s) if groups is None else groups
    tw_base(em, kreg, s, descending)
    em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s + 4, vidx))
    if "tw0" in ABLATE:
        em.valu("v_mov_b32_e32 v%d, 0" % (V_TWO,))
    if not descending:
        for g in groups:
            r = V_TW + 4 * tw_slot(s, g)
            seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_TWO, S_BASE2, g * 16))
    else:
        em.valu("v_mov_b32_e32 v%d, s84" % (V_TWA,))
        em.valu("v_mov_b32_e32 v%d, s85" % (V_TWA + 1,))
        em.valu("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (V_TWA, V_TWA, V_TWO), "vcc", None)
        em.valu("v_subbrev_co_u32_e32 v%d, vcc, 0, v%d, vcc" % (V_TWA + 1, V_TWA + 1), "vcc", "vcc")
        for g in groups:
            r = V_TW + 4 * tw_slot(s, g)
            seq = vm.load("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (r, r + 3, vp(V_TWA), -g * 16))
    return seq


def lds_write(em, addr, base, stride):
    if "nolds" in ABLATE:
        return
    for k in range(16):
        em.raw("ds_write_b64 v%d, %s offset:%d" % (addr, vp(base + 2 * k), stride * k))


def lds_read(em, addr, base, stride):
    if "nolds" in ABLATE:
        return
    for k in range(16):
        em.raw("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), addr, stride * k))


# twiddle index of (pass, sub-stage s, group g); see fwd_head / fwd_tail / inv_core of kernels_fast.hip:
# with Kf = 2^r + blk the forward passes use K = Kf, 16*Kf, 256*Kf; the mirrored inverse passes use
# K = (512<<r) - 256*blk, (32<<r) - 16*blk, (2<<r) - blk.
PASS_TW = {
    "F1": lambda em, vm, s: tw_uniform_stage(em, vm, s, S_K["F1"], False),
    "F2": lambda em, vm, s: tw_lane_stage(em, vm, s, V_BIDX, S_K["F2"], False),
    "F3": lambda em, vm, s: tw_lane_stage(em, vm, s, V_TID, S_K["F3"], False),
    "I1": lambda em, vm, s: tw_lane_stage(em, vm, s, V_TID, S_K["I1"], True),
    "I2": lambda em, vm, s: tw_lane_stage(em, vm, s, V_BIDX, S_K["I2"], True),
    "I3": lambda em, vm, s: tw_uniform_stage(em, vm, s, S_K["I3"], True),
}


def lane_contig_setup(em):
    """T(1,0) = byte offset of element 1024*w + l inside a 4096-word block (w = t>>6, l = t&63);
    T(1,1) = padded LDS byte address of the same element.  Per j the element 1024w + 64j + l sits at
    +512*j bytes in global memory and +544*j bytes in the padded slab."""
    g, l = T(1, 0), T(1, 6)       # (T + 1 of stream 0 holds the fold mask: in single-stream mode both streams share the temporaries)
    em.valu("v_lshrrev_b32_e32 v%d, 6, v%d" % (g, V_TID))                 # w
    em.valu("v_and_b32_e32 v%d, 63, v%d" % (l, V_TID))                    # l
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (T(1, 2), l))               # l >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (T(1, 2), T(1, 2), l))        # l + (l>>4)
    em.valu("v_mov_b32_e32 v%d, 0x440" % (T(1, 3),))                      # 1088 = 1024 + 64
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (T(1, 2), g, T(1, 3), T(1, 2)))
    em.valu("v_lshlrev_b32_e32 v%d, 10, v%d" % (g, g))
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (g, g, l))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (g, g))                      # global byte offset
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (l, T(1, 2)))                # LDS byte address
    return g, l


def prologue(em, vm, kind="polymul"):
    R = em.raw
    # ---------------- prologue
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x20")           # mc
    R("s_load_dword s14, s[0:1], 0x28")                  # nm
    R("s_load_dword s88, s[0:1], 0x2c")                  # logn
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_OFF8, V_TID))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (V_BIDX, V_TID))                     # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (V_L1W, V_TID, V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1W, V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (V_L1R, V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_L1R, V_BIDX, V_L2R, V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1R, V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_L2R, V_TID, V_L2R))              # 17*t*8
    for s in sorted(set(V_T)):
        em.valu("v_mov_b32_e32 v%d, 0" % (s + 15,))                                 # the persistent zero of ZP
    R("s_waitcnt lgkmcnt(0)")
    if kind in ("fwd2", "inv2"):
        # two rows per workgroup (n = 4096 only): polynomials 2 wgx and 2 wgx + 1 of this modulus; the odd one out at
        # the end of the batch is done twice (same words stored twice)
        R("s_load_dword s86, s[0:1], 0x30")              # count
        R("s_lshl_b32 s2, s2, 1")
        R("s_waitcnt lgkmcnt(0)")
        R("s_add_u32 s87, s2, 1")
        R("s_cmp_lt_u32 s87, s86")
        R("s_cselect_b32 s86, s14, 0")                   # rows to the second polynomial: nm or 0
        R("s_lshr_b32 s87, s86, 17")
        R("s_lshl_b32 s86, s86, 15")                     # ... in bytes -> s[86:87] (consumed below)
    # r = logn - 12; wgx = poly * 2^r + blk; block = ((poly*nm + cm) << r) + blk; byte offset = block << 15
    R("s_sub_u32 s88, s88, 12")
    R("s_lshr_b32 s42, s2, s88")                         # poly
    R("s_lshl_b32 s43, s42, s88")
    R("s_sub_u32 s89, s2, s43")                          # blk
    R("s_mul_i32 s42, s42, s14")
    R("s_add_u32 s42, s42, s3")                          # row
    R("s_lshl_b32 s42, s42, s88")
    R("s_add_u32 s42, s42, s89")                         # block index
    if "row0" in ABLATE:
        R("s_and_b32 s42, s42, 15")
    R("s_lshr_b32 s43, s42, 17")
    R("s_lshl_b32 s42, s42, 15")
    for base, row in ((6, 16), (8, 18), (4, 20)):
        R("s_add_u32 s%d, s%d, s42" % (row, base))
        R("s_addc_u32 s%d, s%d, s43" % (row + 1, base + 1))
    if kind in ("fwd2", "inv2"):
        R("s_add_u32 s18, s16, s86")                     # second source row
        R("s_addc_u32 s19, s17, s87")
        R("s_add_u32 s54, s20, s86")                     # second destination row (s[54:55] is free in the 4096-word map)
        R("s_addc_u32 s55, s21, s87")
    # tw = psi + (cm << (logn + 4)) ; mc record = mc + cm*112
    R("s_add_u32 s43, s88, 16")
    R("s_lshl_b32 s42, s3, s43")
    R("s_add_u32 s22, s10, s42")
    R("s_addc_u32 s23, s11, 0")
    # pass constants K
    R("s_lshl_b32 s90, 1, s88")
    R("s_add_u32 s90, s90, s89")                         # Kf = 2^r + blk
    R("s_lshl_b32 s91, s90, 4")
    R("s_lshl_b32 s92, s90, 8")
    R("s_lshl_b32 s93, 0x200, s88")
    R("s_lshl_b32 s42, s89, 8")
    R("s_sub_u32 s93, s93, s42")                         # (512<<r) - 256*blk
    R("s_lshl_b32 s94, 32, s88")
    R("s_lshl_b32 s42, s89, 4")
    R("s_sub_u32 s94, s94, s42")                         # (32<<r) - 16*blk
    R("s_lshl_b32 s95, 2, s88")
    R("s_sub_u32 s95, s95, s89")                         # (2<<r) - blk
    R("s_mul_i32 s42, s3, 0x70")
    R("s_add_u32 s42, s12, s42")
    R("s_addc_u32 s43, s13, 0")
    R("s_load_dwordx16 s[56:71], s[42:43], 0x0")          # p p2 mu ninv ninv_sh w1ninv w1ninv_sh beta
    R("s_load_dwordx8 s[72:79], s[42:43], 0x40")          # beta_sh yinv yinv_sh mask
    R("s_load_dwordx4 s[80:83], s[42:43], 0x60")          # delta mu2

    def row_loads(dst_base, srow):
        seq = 0
        R("s_mov_b64 s[86:87], %s" % (srow,))
        for k in range(16):
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(dst_base + 2 * k), V_OFF8, (k & 1) * 2048))
            if k & 1:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
        return seq
    def lane_loads(dst_base, srow):
        """element 1024w + 64j + l -> register pair j (fully coalesced 512 B per wave instruction)"""
        g, _ = lane_contig_setup(em)
        R("s_mov_b64 s[86:87], %s" % (srow,))
        for j in range(16):
            vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(dst_base + 2 * j), g, (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")

    def thread16_loads(dst_base, srow):
        """words 16t .. 16t+15 (the layout NTT-form data has after F3) as 8 x 16-byte loads"""
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_TID))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (dst_base + 4 * i, dst_base + 4 * i + 3, T(1, 0), srow, 16 * i))

    if kind == "polymul":
        row_loads(V_A, S_AROW)
        row_loads(V_B, S_BROW)
    elif kind == "polymul_ntt":
        row_loads(V_A, S_AROW)
        thread16_loads(V_B, S_BROW)
    elif kind == "fwd":
        row_loads(V_A, S_AROW)
    elif kind == "fwd2":
        row_loads(V_A, S_AROW)
        row_loads(V_B, S_BROW)
    elif kind == "inv":
        lane_loads(V_A, S_AROW)
    elif kind == "inv2":
        lane_loads(V_A, S_AROW)
        lane_loads(V_B, S_BROW)
    elif kind == "inv_mul":
        lane_loads(V_A, S_AROW)
        lane_loads(V_B, S_BROW)
    first = "I1" if kind in ("inv", "inv_mul", "inv2") else "F1"
    tw_seq = {}
    for s in ((3, 2, 1, 0) if first == "I1" else (0, 1, 2, 3)):
        tw_seq[(first, s)] = PASS_TW[first](em, vm, s)
    R("s_waitcnt lgkmcnt(0)")
    # constants from the ModConst record
    R("s_mov_b64 s[24:25], s[56:57]")                    # p
    R("s_mov_b64 s[26:27], s[58:59]")                    # 2p
    R("s_add_u32 s28, s58, s56")                         # 3p
    R("s_addc_u32 s29, s59, s57")
    R("s_mov_b32 s30, s80")                              # delta
    R("s_mov_b32 s31, 0x3fffffff")
    R("s_mov_b32 s15, 0xc0000000")
    R("s_mov_b64 s[32:33], s[82:83]")                    # mu2
    R("s_mov_b64 s[34:35], s[62:63]")                    # ninv
    R("s_mov_b64 s[36:37], s[64:65]")                    # ninv_sh
    R("s_mov_b64 s[38:39], s[66:67]")                    # w1ninv
    R("s_mov_b64 s[40:41], s[68:69]")                    # w1ninv_sh
    em.valu("v_mov_b32_e32 v%d, s25" % (V_PHI,))
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        em.valu("v_mov_b32_e32 v%d, 0x3fffffff" % (v_mask(),))

    return tw_seq


def strided_rows(em, vm, base, srow, stride, store=False, offset=0, nwords=16):
    """16 words x[t + k*stride/8] of the row at srow (+ offset bytes) <-> register pairs base+2k; returns the number
    of the last memory instruction issued (stores are counted too when a VmCounter is given)"""
    R = em.raw
    seq = 0
    R("s_mov_b64 s[86:87], %s" % (srow,))
    if offset:
        R("s_add_u32 s86, s86, 0x%x" % offset)
        R("s_addc_u32 s87, s87, 0")
    for k in range(nwords):
        if stride == 2048:
            off = (k & 1) * 2048
        else:
            off = 0
        if store:
            text = "global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (V_OFF8, vp(base + 2 * k), off)
            if vm is None:
                R(text)
            else:
                seq = vm.load(text)
        else:
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(base + 2 * k), V_OFF8, off))
        if stride == 2048:
            if k & 1:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
        elif k < nwords - 1:
            R("s_add_u32 s86, s86, 0x%x" % stride)
            R("s_addc_u32 s87, s87, 0")
    return seq


def epilogue_inverse(em, vm, last_plain_stage, suffix="", stride=2048):
    """stride: bytes between a thread's consecutive words x[t + 256k] (2048 inside a 4096-word block; n/16 words for
    the streaming passes of long rows)"""
    R = em.raw
    R("s_cmp_eq_u32 s88, 0")
    R("s_cbranch_scc1 .Lmerged_last_stage%s" % suffix)
    em.comment("r > 0: plain stage r (uniform twiddle psi[(2<<r) - 1 - blk]); lazy output for the outer passes")
    last_plain_stage()
    R("s_branch .Lstore%s" % suffix)
    em.lines.append(".Lmerged_last_stage%s:" % suffix)
    em.comment("r == 0: stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(V_A + 2 * h, V_A + 2 * (h + 8)) for h in range(8)])
    em.lines.append(".Lstore%s:" % suffix)
    # ---------------- store c (x[t + 256k])
    strided_rows(em, None, V_A, S_CROW, stride, store=True)
    R("s_endpgm")


def epilogue_forward(em, vm, end=True, base=None, dst=None):
    """canonical words, then a wave-local LDS transpose so the stores are fully coalesced"""
    R = em.raw
    base = V_A if base is None else base
    run_pairs(em, [canon(base + 2 * i) for i in range(16)])
    lds_write(em, V_L2R, base, 8)
    g, l = lane_contig_setup(em)
    for j in range(16):
        R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l, 544 * j))
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[86:87], %s" % (S_CROW if dst is None else dst,))
    for j in range(16):
        R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (g, vp(base + 2 * j), (j & 7) * 512))
        if j == 7:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    if end:
        R("s_endpgm")


def build(kind="polymul"):
    """kind: polymul | polymul_ntt (b already in NTT form) | fwd | inv | inv_mul (inverse of src (.) mul)"""
    em = Emitter()
    vm = VmCounter(em)
    tw_seq = prologue(em, vm, kind)
    return build_body(em, vm, kind, tw_seq)


def build_body(em, vm, kind, tw_seq, suffix=""):
    R = em.raw
    has_fwd = kind in ("polymul", "polymul_ntt", "fwd", "fwd2")
    has_inv = kind not in ("fwd", "fwd2")
    fwd_bases = [V_A, V_B] if kind in ("polymul", "fwd2") else [V_A]
    passes = (["F1", "F2", "F3"] if has_fwd else []) + (["I1", "I2", "I3"] if has_inv else [])
    inv_bases = [V_A, V_B] if kind == "inv2" else [V_A]

    def nxt_of(name):
        i = passes.index(name)
        return passes[i + 1] if i + 1 < len(passes) else None

    def fwd_pass(name):
        nxt = nxt_of(name)
        em.comment("%s; prefetching %s" % (name, nxt))
        for s in range(4):
            vm.wait(tw_seq[(name, s)])
            ct_stage(em, fwd_bases, s)
            if nxt is not None:
                tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)

    def inv_pass(name, stages=(3, 2, 1, 0)):
        nxt = nxt_of(name)
        em.comment("%s; prefetching %s" % (name, nxt))
        for s in stages:
            vm.wait(tw_seq[(name, s)])
            for base in inv_bases:
                gs_stage(em, base, s)
            if nxt is not None:
                tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)

    if has_fwd:
        fwd_pass("F1")
        for i, base in enumerate(fwd_bases):
            em.comment("E1")
            if i:
                R("s_barrier")       # WAR: the slab is still being read for the previous operand
            lds_write(em, V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F2")
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        for base in fwd_bases:
            lds_write(em, V_L1R, base, 136)
            lds_read(em, V_L2R, base, 8)
        R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F3")
    if kind == "fwd":
        epilogue_forward(em, vm)
        return em
    if kind == "fwd2":
        epilogue_forward(em, vm, end=False)
        epilogue_forward(em, vm, base=V_B, dst="s[54:55]")
        return em

    if kind in ("polymul", "polymul_ntt"):
        em.comment("point-wise product (thread q holds words 16q..16q+15 of both operands)")
        run_pairs(em, [pointwise(V_A + 2 * i, V_B + 2 * i, True, kind == "polymul") for i in range(16)])
    else:
        R("s_waitcnt vmcnt(%d)" % (vm.issued - (32 if kind in ("inv_mul", "inv2") else 16)))   # the row loads have landed
        if kind == "inv_mul":
            em.comment("point-wise product of canonical NTT-form operands (any common layout works)")
            run_pairs(em, [pointwise(V_A + 2 * i, V_B + 2 * i, False, False) for i in range(16)])
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        for base in inv_bases:
            for j in range(16):
                R("ds_write_b64 v%d, %s offset:%d" % (l, vp(base + 2 * j), 544 * j))
            lds_read(em, V_L2R, base, 8)
            R("s_waitcnt lgkmcnt(0)")

    inv_pass("I1")
    em.comment("E2'")
    for base in inv_bases:
        lds_write(em, V_L2R, base, 8)
        lds_read(em, V_L1R, base, 136)
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2")
    em.comment("E1'")
    for i, base in enumerate(inv_bases):
        if i:
            R("s_barrier")       # WAR: the slab is still being read for the previous row
        lds_write(em, V_L1R, base, 136)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, V_L1W, base, 2176)
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3", stages=(3, 2, 1))
    if kind == "inv2":
        em.comment("stage 0 with n^-1 folded in, both rows (n = 4096 only)")
        for base, dst in ((V_A, S_CROW), (V_B, "s[54:55]")):
            run_pairs(em, [final_bfly(base + 2 * h, base + 2 * (h + 8)) for h in range(8)])
            strided_rows(em, None, base, dst, 2048, store=True)
        R("s_endpgm")
        return em

    def last_plain():
        vm.wait(tw_seq[("I3", 0)])
        gs_stage(em, V_A, 0)
    # I3's sub-stage-0 record is only used by the r > 0 tail; make sure it was requested
    if ("I3", 0) not in tw_seq:
        tw_seq[("I3", 0)] = PASS_TW["I3"](em, vm, 0)
    epilogue_inverse(em, vm, last_plain)
    return em


# ------------------------------------------------------------------ 16384-word rows: one 1024-thread workgroup
# A row of 16384 words (or a 16384-word block of a longer row) stays on one CU for the whole product:
# 16 waves x 16 words per thread, 128 VGPRs (4 waves per SIMD).  Sub-group q = tid >> 8 (4 waves) runs the
# 4096-word passes F1..F3 / I1..I3 above on block q in its own LDS slab; one extra radix-4 pass F0 / I0
# (global stages r-2, r-1) in front / behind couples the four blocks through a workgroup-wide exchange X0.
# Register budget: one butterfly at a time (18 temporaries) and the twiddle records stream through a
# 9-slot ring in the static order the kernel consumes them.
class Ring:
    """Twiddle records stream through a small ring of register slots: the order in which the
    whole kernel consumes its records is static, so each slot is refilled with the record
    that is NSLOTS uses ahead as soon as its last butterfly has been issued."""

    def __init__(self, em, vm, nslots, uses, passes, side=None):
        self.em, self.vm, self.uses, self.passes = em, vm, uses, passes
        self.side = side or {}    # issue index -> callables: other loads woven into the twiddle stream (row prefetches)
        self.free = list(range(nslots))
        self.slot_of, self.seq_of = {}, {}
        self.next = 0
        self.cur = None   # (pass, s) whose scalar base / lane offset registers are currently set up

    def regs(self, use):
        b = V_TW + 4 * self.slot_of[use]
        return ("v%d" % b, "v%d" % (b + 1), "v%d" % (b + 2), "v%d" % (b + 3))

    def _issue(self):
        self._issue_record()
        for f in self.side.get(self.next - 1, ()):
            f()

    def _issue_record(self):
        use = self.uses[self.next]
        self.next += 1
        slot = self.free.pop(0)
        self.slot_of[use] = slot
        name, s, g = use
        em, r = self.em, V_TW + 4 * slot
        if callable(self.passes[name]):   # not a twiddle record: the pass supplies the load (row32k streams b' this way)
            text = self.passes[name](em, r, s, g, self.cur != (name, s))
            self.cur = (name, s)
            self.seq_of[use] = self.vm.load(text)
            return
        kreg, vidx, desc = self.passes[name][:3]
        koff = self.passes[name][3] if len(self.passes[name]) > 3 else 0
        fresh = self.cur != (name, s)
        if LANE_MAJOR and vidx is not None and vidx == V_TID:
            if self.cur is None or self.cur[0] != name:
                tw_lane_offset_lm(em, desc)
            self.cur = (name, s)
            tw_base_lm(em, kreg, s, g, desc, koff)
            self.seq_of[use] = self.vm.load("global_load_dwordx4 v[%d:%d], v%d, %s" % (r, r + 3, V_TWO, S_BASE2))
            return
        if fresh:
            tw_base(em, kreg, s, desc, koff)
            if vidx is not None:
                em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, s + 4, vidx))
                if "tw0" in ABLATE:
                    em.valu("v_mov_b32_e32 v%d, 0" % (V_TWO,))
            self.cur = (name, s)
        if vidx is not None and desc and (fresh or RING_RECOMPUTE_TWA):   # (the address pair is butterfly scratch in ringpair mode)
            em.valu("v_mov_b32_e32 v%d, s84" % (V_TWA,))
            em.valu("v_mov_b32_e32 v%d, s85" % (V_TWA + 1,))
            em.valu("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (V_TWA, V_TWA, V_TWO), "vcc", None)
            em.valu("v_subbrev_co_u32_e32 v%d, vcc, 0, v%d, vcc" % (V_TWA + 1, V_TWA + 1), "vcc", "vcc")
        off = -g * 16 if desc else g * 16
        if vidx is None:
            text = "global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_ZERO, S_BASE2, off)
        elif not desc:
            text = "global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, V_TWO, S_BASE2, off)
        else:
            text = "global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (r, r + 3, vp(V_TWA), off)
        self.seq_of[use] = self.vm.load(text)

    def prime(self):
        while self.free and self.next < len(self.uses):
            self._issue()

    def get(self, use):
        self.vm.wait(self.seq_of[use])
        return self.regs(use)

    def done(self, use):
        self.free.append(self.slot_of[use])
        if self.next < len(self.uses):
            self._issue()


# (SGPRs of the second butterfly stream, idle in single-stream mode; s54/s55 are unused by the 4096-word map)
S_Q, S_SLAB = "s54", "s55"                  # sub-group index, byte offset of its LDS slab
S_K0 = {"F0": "s46", "I0": "s47"}
SLAB_BYTES = (4096 + 256) * 8


def configure(mode, groups=4):
    """Select the register map: "pair" = two interleaved butterflies, 15 twiddle records resident,
    168 VGPRs (3 waves/SIMD); "ring" = one butterfly at a time, 9-slot twiddle ring, 128 VGPRs (4 waves/SIMD)."""
    g = globals()
    if mode == "pair":
        g.update(SINGLE_STREAM=False, V_BIDX=5, V_PHI=6, V_A=8, V_B=40, V_TW=72, V_T=[132, 150], NEXT_VGPR=168,
                 NEXT_SGPR=96, LDS_BYTES=SLAB_BYTES, WG_SIZE=256)
        g.update(V_TWO=g["V_T"][1] + 1, V_TWA=g["V_T"][1] + 4, V_ZERO=g["V_T"][0] + 15)
    elif mode == "ringpair":
        # experiment (NFL_GEN_RINGPAIR=1): the 128-VGPR row kernels with TWO interleaved butterflies and a 5-slot ring instead
        # of one butterfly at a time and 9 slots; the twiddle address scratch lives in stream 1's temporaries
        g.update(SINGLE_STREAM=False, V_BIDX=5, V_PHI=6, V_A=8, V_B=40, V_TW=72, V_T=[92, 110], NEXT_VGPR=128, NEXT_SGPR=96,
                 RING_SLOTS=5, LDS_BYTES=groups * SLAB_BYTES, WG_SIZE=256 * groups, ROW_G=groups,
                 ROW_LG=groups.bit_length() - 1, RING_RECOMPUTE_TWA=True)
        g.update(V_TWO=g["V_T"][1] + 1, V_TWA=g["V_T"][1] + 4, V_ZERO=g["V_T"][0] + 15)
        g["S_K0"].update(F0="s98", I0="s99")   # (s46 / s47 are stream 1's carry pair here)
        g["NEXT_SGPR"] = 100
    else:
        g["S_K0"].update(F0="s46", I0="s47")
        g.update(SINGLE_STREAM=True, V_BIDX=5, V_PHI=6, V_TWO=7, V_TWA=8, V_A=10, V_B=42, V_TW=74, V_T=[110, 110],
                 NEXT_VGPR=128, NEXT_SGPR=96, RING_SLOTS=9, LDS_BYTES=groups * SLAB_BYTES, WG_SIZE=256 * groups,
                 ROW_G=groups, ROW_LG=groups.bit_length() - 1, RING_RECOMPUTE_TWA=False)
        g.update(V_ZERO=g["V_T"][0] + 15)


def prologue16k(em, vm, stop=None, kind="polymul", key_row=False, compact_x=False):
    """1024 threads; v0 = tid on entry.  Leaves V_TID = tid & 255 (the thread's index inside its sub-group),
    V_OFF8 = tid*8, the LDS addresses of the sub-group's slab, all pass constants, and the row loads issued."""
    R = em.raw
    if stop == -3:
        R("s_endpgm")
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x20")           # mc
    R("s_load_dword s14, s[0:1], 0x28")                  # nm
    R("s_load_dword s88, s[0:1], 0x2c")                  # logn
    if kind == "fwd2":
        R("s_load_dword s96, s[0:1], 0x30")              # count: the workgroup transforms polynomials 2 wgx and 2 wgx + 1
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_OFF8, V_TID))                      # tid*8
    em.valu("v_lshrrev_b32_e32 v%d, 8, v%d" % (V_BIDX, V_TID))                      # q (wave-uniform)
    R("s_nop 1")                 # gfx950: a VALU VGPR write needs a wait state before v_readfirstlane reads it
    R("v_readfirstlane_b32 %s, v%d" % (S_Q, V_BIDX))
    R("s_nop 1")                 # ... and the SGPR it writes two before an SALU read
    em.valu("v_and_b32_e32 v%d, 0xff, v%d" % (V_TID, V_TID))                        # t = tid & 255
    R("s_mul_i32 %s, %s, 0x%x" % (S_SLAB, S_Q, SLAB_BYTES))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (V_BIDX, V_TID))                      # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (V_L1W, V_TID, V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1W, V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (V_L1R, V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_L1R, V_BIDX, V_L2R, V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1R, V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_L2R, V_TID, V_L2R))              # 17*t*8
    for reg in (V_L1W, V_L1R, V_L2R):
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (reg, S_SLAB, reg))                  # inside the sub-group's slab
    for t_ in sorted(set(V_T)):
        em.valu("v_mov_b32_e32 v%d, 0" % (t_ + 15,))                                # the persistent zero of each stream's ZP pair
    R("s_waitcnt lgkmcnt(0)")
    if kind == "fwd2":
        R("s_lshl_b32 s2, s2, 1")
    # G = ROW_G sub-groups, LG = log2 G: r = logn - 12 (>= LG); wgx = poly * 2^(r-LG) + blkG;
    # (4096 G)-word block = ((poly*nm + cm) << (r-LG)) + blkG
    R("s_sub_u32 s88, s88, 12")
    R("s_sub_u32 s86, s88, %d" % ROW_LG)                 # r - LG
    R("s_lshr_b32 s42, s2, s86")                         # poly
    R("s_lshl_b32 s43, s42, s86")
    R("s_sub_u32 s87, s2, s43")                          # blk16
    R("s_mul_i32 s42, s42, s14")
    R("s_add_u32 s42, s42, s3")                          # row
    R("s_lshl_b32 s42, s42, s86")
    R("s_add_u32 s42, s42, s87")                         # block index
    if "row0" in ABLATE:
        R("s_and_b32 s42, s42, 3")
    R("s_lshr_b32 s43, s42, %d" % (32 - 15 - ROW_LG,))
    R("s_lshl_b32 s42, s42, %d" % (15 + ROW_LG,))        # * 4096 G words * 8 bytes
    if ALIAS_ROWS:   # ablation "bprimeN": the scratch operand of the composed 32768-word product laid over N row blocks (cache-resident)
        R("s_lshr_b32 s44, s42, %d" % (15 + ROW_LG,))
        R("s_and_b32 s44, s44, %d" % (BPRIME_ALIAS - 1,))
        R("s_lshl_b32 s44, s44, %d" % (15 + ROW_LG,))
    for base, row in ((6, 16), (8, 18), (4, 20)):
        if row in ALIAS_ROWS:
            R("s_add_u32 s%d, s%d, s44" % (row, base))
            R("s_addc_u32 s%d, s%d, 0" % (row + 1, base + 1))
            continue
        R("s_add_u32 s%d, s%d, s42" % (row, base))
        R("s_addc_u32 s%d, s%d, s43" % (row + 1, base + 1))
    if compact_x:
        # operand a is ONE signed byte per coefficient (the samplers' compact output), the same for every modulus: row = a + poly * n
        R("s_lshr_b32 s42, s2, s86")                     # poly
        R("s_mov_b32 s43, 0")
        R("s_add_u32 s16, s88, 12")                      # logn
        R("s_lshl_b64 s[42:43], s[42:43], s16")
        R("s_add_u32 s16, s6, s42")
        R("s_addc_u32 s17, s7, s43")
    if key_row:
        # a third input row (the fused inverse kinds of build_row32k): its base at kernarg 0x30, and at 0x38 whether it advances
        # with the batch (1) or is ONE polynomial for every element (0: a key) -> s[98:99]
        R("s_load_dwordx2 s[98:99], s[0:1], 0x30")
        R("s_load_dword s100, s[0:1], 0x38")
        R("s_lshr_b32 s42, s2, s86")                     # poly
        R("s_waitcnt lgkmcnt(0)")
        R("s_mul_i32 s42, s42, s100")                    # ... or 0
        R("s_mul_i32 s42, s42, s14")
        R("s_add_u32 s42, s42, s3")                      # row
        R("s_lshl_b32 s42, s42, s86")
        R("s_add_u32 s42, s42, s87")                     # block index
        R("s_lshr_b32 s43, s42, %d" % (32 - 15 - ROW_LG,))
        R("s_lshl_b32 s42, s42, %d" % (15 + ROW_LG,))
        R("s_add_u32 s98, s98, s42")
        R("s_addc_u32 s99, s99, s43")
    if kind == "fwd2":
        # the second polynomial (same modulus: nm rows further), or the first one again for the odd one out at the end of
        # the batch (transformed twice, stored twice to the same place): source s[18:19], destination s[96:97]
        R("s_add_u32 s42, s2, 1")
        R("s_cmp_lt_u32 s42, s96")
        R("s_cselect_b32 s42, s14, 0")                   # rows to the second polynomial: nm or 0
        R("s_lshr_b32 s43, s42, %d" % (32 - 15 - ROW_LG,))
        R("s_lshl_b32 s42, s42, %d" % (15 + ROW_LG,))
        R("s_add_u32 s18, s16, s42")
        R("s_addc_u32 s19, s17, s43")
        R("s_add_u32 s96, s20, s42")
        R("s_addc_u32 s97, s21, s43")
    # tw = psi + (cm << (logn + 4))
    R("s_add_u32 s43, s88, 16")
    R("s_lshl_b32 s42, s3, s43")
    R("s_add_u32 s22, s10, s42")
    R("s_addc_u32 s23, s11, 0")
    # outer pass constants: K_F0 = 2^(r-LG) + blkG, K_I0 = 2^(r-LG+1) - blkG
    R("s_lshl_b32 %s, 1, s86" % (S_K0["F0"],))
    R("s_add_u32 %s, %s, s87" % (S_K0["F0"], S_K0["F0"]))
    R("s_lshl_b32 %s, 2, s86" % (S_K0["I0"],))
    R("s_sub_u32 %s, %s, s87" % (S_K0["I0"], S_K0["I0"]))
    # inner pass constants of block blk = G*blkG + q
    R("s_lshl_b32 s89, s87, %d" % ROW_LG)
    R("s_add_u32 s89, s89, %s" % (S_Q,))
    R("s_lshl_b32 s90, 1, s88")
    R("s_add_u32 s90, s90, s89")                         # Kf = 2^r + blk
    R("s_lshl_b32 s91, s90, 4")
    R("s_lshl_b32 s92, s90, 8")
    R("s_lshl_b32 s93, 0x200, s88")
    R("s_lshl_b32 s42, s89, 8")
    R("s_sub_u32 s93, s93, s42")                         # (512<<r) - 256*blk
    R("s_lshl_b32 s94, 32, s88")
    R("s_lshl_b32 s42, s89, 4")
    R("s_sub_u32 s94, s94, s42")                         # (32<<r) - 16*blk
    R("s_lshl_b32 s95, 2, s88")
    R("s_sub_u32 s95, s95, s89")                         # (2<<r) - blk
    R("s_mul_i32 s42, s3, 0x70")
    R("s_add_u32 s42, s12, s42")
    R("s_addc_u32 s43, s13, 0")
    R("s_load_dwordx16 s[56:71], s[42:43], 0x0")          # p p2 mu ninv ninv_sh w1ninv w1ninv_sh beta
    R("s_load_dwordx8 s[72:79], s[42:43], 0x40")          # beta_sh yinv yinv_sh mask
    R("s_load_dwordx4 s[80:83], s[42:43], 0x60")          # delta mu2
    if stop == -2:
        R("s_waitcnt vmcnt(0) lgkmcnt(0)")
        R("s_endpgm")
    def row_loads(dst, srow):                             # x[tid + 256 G k]: the layout F0 starts from
        R("s_mov_b64 s[86:87], %s" % (srow,))
        for k in range(16):
            vm.load("global_load_dwordx2 %s, v%d, s[86:87]" % (vp(dst + 2 * k), V_OFF8))
            if k < 15:
                R("s_add_u32 s86, s86, 0x%x" % (2048 * ROW_G,))
                R("s_addc_u32 s87, s87, 0")

    def block_base(srow):                                 # s[86:87] = first word of this sub-group's 4096-word block
        R("s_lshl_b32 s42, %s, 15" % (S_Q,))
        R("s_add_u32 s86, s%s, s42" % (srow[2:].split(":")[0],))
        R("s_addc_u32 s87, s%s, 0" % (srow.split(":")[1][:-1],))

    def thread16_loads(dst, srow):                        # words 16t .. 16t+15 of the block (NTT-form data, after F3)
        block_base(srow)
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(0, 0), V_TID))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, s[86:87] offset:%d" % (dst + 4 * i, dst + 4 * i + 3, T(0, 0), 16 * i))

    def lane_loads(dst, srow):                            # block element 1024w + 64j + l -> pair j (512 B per wave load)
        block_base(srow)
        g, _ = lane_contig_setup(em)
        for j in range(16):
            vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(dst + 2 * j), g, (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")

    if kind in ("polymul", "fwd2"):
        row_loads(V_A, S_AROW)
        row_loads(V_B, S_BROW)
    elif kind == "polymul_ntt":
        row_loads(V_A, S_AROW)
        thread16_loads(V_B, S_BROW)
    elif kind == "fwd":
        row_loads(V_A, S_AROW)
    elif kind == "none":      # (build_row32k issues its own loads)
        pass
    else:
        lane_loads(V_A, S_AROW)
    if stop == -1:
        R("s_waitcnt vmcnt(0) lgkmcnt(0)")
        R("s_endpgm")
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[24:25], s[56:57]")                    # p
    R("s_mov_b64 s[26:27], s[58:59]")                    # 2p
    R("s_add_u32 s28, s58, s56")                         # 3p
    R("s_addc_u32 s29, s59, s57")
    R("s_mov_b32 s30, s80")                              # delta
    R("s_mov_b32 s31, 0x3fffffff")
    R("s_mov_b32 s15, 0xc0000000")
    R("s_mov_b64 s[32:33], s[82:83]")                    # mu2
    R("s_mov_b64 s[34:35], s[62:63]")                    # ninv
    R("s_mov_b64 s[36:37], s[64:65]")                    # ninv_sh
    R("s_mov_b64 s[38:39], s[66:67]")                    # w1ninv
    R("s_mov_b64 s[40:41], s[68:69]")                    # w1ninv_sh
    em.valu("v_mov_b32_e32 v%d, s25" % (V_PHI,))
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        em.valu("v_mov_b32_e32 v%d, 0x3fffffff" % (v_mask(),))


def build_row16k(kind="polymul", stop=None):
    """kind: polymul | polymul_ntt (b already in NTT form) | fwd | inv -- over one 16384-word block per workgroup"""
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    passes = {"F0": (S_K0["F0"], None, False), "F1": (S_K["F1"], None, False), "F2": (S_K["F2"], V_BIDX, False),
              "F3": (S_K["F3"], V_TID, False), "I1": (S_K["I1"], V_TID, True), "I2": (S_K["I2"], V_BIDX, True),
              "I3": (S_K["I3"], None, True), "I0": (S_K0["I0"], None, True)}
    order = {"F0": tuple(range(ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0), "I0": tuple(range(ROW_LG - 1, -1, -1))}
    per = 16 // ROW_G          # register slots per 4096-word block in the row layout x[tid + 256 G k]
    has_fwd = kind != "inv"
    has_inv = kind not in ("fwd", "fwd2")
    names = (["F0", "F1", "F2", "F3"] if has_fwd else []) + (["I1", "I2", "I3", "I0"] if has_inv else [])
    uses = [(name, s, g) for name in names for s in order[name] for g in range(1 << s)]
    ring = Ring(em, vm, RING_SLOTS, uses, passes)
    fwd_bases = (V_A, V_B) if kind in ("polymul", "fwd2") else (V_A,)
    prologue16k(em, vm, stop, kind)
    n_before_ring = vm.issued
    ring.prime()

    def ck(n):   # debugging aid: build_row16k(stop=n) ends the kernel at checkpoint n
        if stop == n:
            R("s_waitcnt vmcnt(0) lgkmcnt(0)")
            R("s_endpgm")
    ck(0)

    def fwd_pass(name):
        em.comment("%s (operands share the twiddles)" % name)
        for s in order[name]:
            half = 8 >> s
            for g in range(1 << s):
                tw = ring.get((name, s, g))
                jobs = []
                for h in range(half):
                    i0 = g * 2 * half + h
                    for base in fwd_bases:
                        jobs.append(ct_bfly(base + 2 * i0, base + 2 * (i0 + half), tw))
                run_pairs(em, jobs)
                ring.done((name, s, g))

    def inv_pass(name, stages):
        em.comment(name)
        for s in stages:
            half = 8 >> s
            for g in range(1 << s):
                tw = ring.get((name, s, g))
                run_pairs(em, [gs_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw)
                               for h in range(half)])
                ring.done((name, s, g))

    AX = T(0, 0)   # exchange address scratch (the butterfly temporaries are idle during exchanges)
    # ---- two operands on shared twiddle records, exchanges under the arithmetic (kind "polymul"): in the last stage of a
    # pass and in the first two of the next one the butterflies of a run first, the records stay in the ring, then b's:
    #   last stage: a | W_a | b | barrier | R_a | barrier | W_b     next pass, stage 0: a | barrier | R_b     stage 1: a | b, b
    # so that a's writes, b's writes and b's reads are in flight under butterflies; only a's reads are waited for in the open
    # (consuming them word by word under a's stage 0 as well was measured: nothing, tools/sessions/gpu_round3_x.sh).
    # (Ring: the 8 records of a last stage are all live at once -- 9 slots; nothing is fetched twice.)
    def bflys(base, s_, g, tw):
        half = 8 >> s_
        return [ct_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)]

    def hold_a(name, s_):
        for g in range(1 << s_):
            run_pairs(em, bflys(V_A, s_, g, ring.get((name, s_, g))))

    def then_b(name, s_):
        for g in range(1 << s_):
            run_pairs(em, bflys(V_B, s_, g, ring.regs((name, s_, g))))
            ring.done((name, s_, g))

    def both(name, s_):
        for g in range(1 << s_):
            tw = ring.get((name, s_, g))
            jobs = []
            for ja, jb in zip(bflys(V_A, s_, g, tw), bflys(V_B, s_, g, tw)):
                jobs += [ja, jb]
            run_pairs(em, jobs)
            ring.done((name, s_, g))

    def x0_w(base):
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
        for k in range(16):
            qq, j = k // per, k % per
            R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * SLAB_BYTES + j * 2048 * ROW_G))

    def x0_r(base):
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))

    def fwd_split():
        W0, BAR = "s_waitcnt lgkmcnt(0)", "s_barrier"
        exch = {"F0": (x0_w, x0_r, True),
                "F1": (lambda b_: lds_write(em, V_L1W, b_, 2176), lambda b_: lds_read(em, V_L1R, b_, 136), True),
                "F2": (lambda b_: lds_write(em, V_L1R, b_, 136), lambda b_: lds_read(em, V_L2R, b_, 8), False)}   # E2: wave-local
        names = ["F0", "F1", "F2", "F3"]
        pending = None          # exchange of operand b still to be finished inside the next pass
        for name in names:
            stages = list(order[name])
            em.comment("%s (operands share the twiddle records; exchanges under the arithmetic)" % name)
            k = 0
            if pending is not None:
                w_, r_, cross = pending
                # stage 0: a alone while b's writes fly; then b's reads under stage 1 of a
                hold_a(name, stages[0])
                if cross:
                    R(W0)
                    R(BAR)
                    r_(V_B)
                hold_a(name, stages[1])
                R(W0)
                if cross:
                    R(BAR)           # every wave is done reading: the next exchange may write
                then_b(name, stages[0])
                then_b(name, stages[1])
                k = 2
                pending = None
            last = stages[-1] if name in exch else None
            for s_ in stages[k:]:
                if s_ != last:
                    both(name, s_)
            if last is not None:
                w_, r_, cross = exch[name]
                hold_a(name, last)
                w_(V_A)
                if not cross:        # wave-local transposes (LDS is in order per wave): a's reads follow its writes at once
                    r_(V_A)
                then_b(name, last)
                R(W0)
                if cross:
                    R(BAR)
                    r_(V_A)
                    R(W0)
                    R(BAR)
                w_(V_B)
                if not cross:
                    r_(V_B)
                pending = exch[name]
        assert pending is None

    def fwd_progressive():
        """one operand: every exchange written word by word out of a pass's last stage and read in the order the next
        pass's first stage consumes (see the inverse half below)"""
        def fwd_stage(name, s_, pre=None, post=None):
            half, i_ = 8 >> s_, 0
            for g in range(1 << s_):
                tw = ring.get((name, s_, g))
                for h in range(half):
                    x, y = g * 2 * half + h, g * 2 * half + h + half
                    if pre:
                        pre(i_)
                    run_pairs(em, [ct_bfly(V_A + 2 * x, V_A + 2 * y, tw)])
                    if post:
                        post(x)
                        post(y)
                    i_ += 1
                ring.done((name, s_, g))

        first = [k for h in range(8) for k in (h, h + 8)]        # visiting order of a pass's stage 0
        arrive = lambda i_: R("s_waitcnt lgkmcnt(%d)" % (14 - 2 * i_))
        AXP = V_TWA                                               # (idle in the forward passes)
        em.comment("F0; X0 written out of its last stage: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
        for s_ in order["F0"][:-1]:
            fwd_stage("F0", s_)
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AXP, 2 * SLAB_BYTES, V_OFF8))
        fwd_stage("F0", order["F0"][-1], post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (
            V_OFF8 if k // per < 2 else AXP, vp(V_A + 2 * k), ((k // per) & 1) * SLAB_BYTES + (k % per) * 2048 * ROW_G)))
        ck(1)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), AX, 2048 * k))
        ck(2)
        fwd_stage("F1", 0, pre=arrive)
        for s_ in (1, 2):
            fwd_stage("F1", s_)
        em.comment("E1 written out of F1's last stage")
        R("s_barrier")               # WAR: every wave is done reading X0
        fwd_stage("F1", 3, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (V_L1W, vp(V_A + 2 * k), 2176 * k)))
        ck(3)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_L1R, 136 * k))
        fwd_stage("F2", 0, pre=arrive)
        for s_ in (1, 2):
            fwd_stage("F2", s_)
        em.comment("E2 (wave-local 16-lane transposes) written out of F2's last stage")
        fwd_stage("F2", 3, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (V_L1R, vp(V_A + 2 * k), 136 * k)))
        ck(4)
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_L2R, 8 * k))
        fwd_stage("F3", 0, pre=arrive)
        for s_ in (1, 2, 3):
            fwd_stage("F3", s_)
        ck(5)

    if has_fwd and kind in ("polymul", "fwd2") and SPLIT32K:
        fwd_split()
    elif has_fwd and SPLIT32K:
        fwd_progressive()
    elif has_fwd:
            fwd_pass("F0")
            ck(1)
            for i, base in enumerate(fwd_bases):
                em.comment("X0: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
                if i:
                    R("s_barrier")       # WAR: the slabs are still being read for the previous operand
                em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
                for k in range(16):
                    qq, j = k // per, k % per
                    R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(base + 2 * k),
                                                           (qq & 1) * SLAB_BYTES + j * 2048 * ROW_G))
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
                for k in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
                R("s_waitcnt lgkmcnt(0)")
            ck(2)
            fwd_pass("F1")
            ck(3)
            for base in fwd_bases:
                em.comment("E1")
                R("s_barrier")           # WAR against the previous exchange through this slab
                lds_write(em, V_L1W, base, 2176)
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                lds_read(em, V_L1R, base, 136)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F2")
            ck(4)
            em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
            for base in fwd_bases:
                lds_write(em, V_L1R, base, 136)
                lds_read(em, V_L2R, base, 8)
            R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F3")
            ck(5)
    if kind == "fwd2":
        em.comment("two rows: canonical words, a wave-local LDS transpose per row so the stores are fully coalesced; the second"
                   " row's reduction runs under the first one's transposes")
        def transposes(base):
            lds_write(em, V_L2R, base, 8)
            _, l = lane_contig_setup(em)
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, S_SLAB, l))
            for j in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l, 544 * j))

        def stores(base, lo, hi):
            g, _ = lane_contig_setup(em)
            R("s_lshl_b32 s42, %s, 15" % (S_Q,))
            R("s_add_u32 s86, s%d, s42" % lo)
            R("s_addc_u32 s87, s%d, 0" % hi)
            for j in range(16):
                R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (g, vp(base + 2 * j), (j & 7) * 512))
                if j == 7:
                    R("s_add_u32 s86, s86, 0x1000")
                    R("s_addc_u32 s87, s87, 0")
        run_pairs(em, [canon(V_A + 2 * i) for i in range(16)])
        transposes(V_A)
        run_pairs(em, [canon(V_B + 2 * i) for i in range(16)])
        R("s_waitcnt lgkmcnt(0)")
        stores(V_A, 20, 21)
        transposes(V_B)
        R("s_waitcnt lgkmcnt(0)")
        stores(V_B, 96, 97)
        R("s_endpgm")
        return em
    if kind == "fwd":
        em.comment("canonical words, then a wave-local LDS transpose so the stores are fully coalesced")
        run_pairs(em, [canon(V_A + 2 * i) for i in range(16)])
        lds_write(em, V_L2R, V_A, 8)
        g, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, S_SLAB, l))
        for j in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * j), l, 544 * j))
        R("s_waitcnt lgkmcnt(0)")
        R("s_lshl_b32 s42, %s, 15" % (S_Q,))
        R("s_add_u32 s86, s20, s42")
        R("s_addc_u32 s87, s21, 0")
        for j in range(16):
            R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (g, vp(V_A + 2 * j), (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
        R("s_endpgm")
        return em

    if kind in ("polymul", "polymul_ntt"):
        em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
        if kind == "polymul_ntt":
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_before_ring))    # b's loads (issued before the ring's) have landed
        run_pairs(em, [pointwise(V_A + 2 * i, V_B + 2 * i, True, kind == "polymul") for i in range(16)])
    else:
        R("s_waitcnt vmcnt(%d)" % (vm.issued - n_before_ring))        # the block loads have landed
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, S_SLAB, l))
        for j in range(16):
            R("ds_write_b64 v%d, %s offset:%d" % (l, vp(V_A + 2 * j), 544 * j))
        lds_read(em, V_L2R, V_A, 8)
        R("s_waitcnt lgkmcnt(0)")
    if SPLIT32K:
        # ---- progressive exchanges of the inverse half (one operand, nothing else to run under an exchange): every word is
        # written to the LDS as soon as the pass's last stage has finished it, and the reads are issued in the order the next
        # pass's first stage consumes them, each butterfly waiting only for its own two (LDS returns in order).
        def inv_stage(name, s_, pre=None, post=None):
            half, i_ = 8 >> s_, 0
            for g in range(1 << s_):
                tw = ring.get((name, s_, g))
                for h in range(half):
                    x, y = g * 2 * half + h, g * 2 * half + h + half
                    if pre:
                        pre(i_)
                    run_pairs(em, [gs_bfly(V_A + 2 * x, V_A + 2 * y, tw)])
                    if post:
                        post(x)
                        post(y)
                    i_ += 1
                ring.done((name, s_, g))

        def visit(s_):
            half = 8 >> s_
            return [k for g in range(1 << s_) for h in range(half) for k in (g * 2 * half + h, g * 2 * half + h + half)]

        def arrive(i_):
            R("s_waitcnt lgkmcnt(%d)" % (14 - 2 * i_))

        AXP = V_TWA                                          # (idle in the uniform pass I3 and in I0)
        rstep = 2048 * ROW_G                                 # bytes between a reader's consecutive slots
        for s_ in (3, 2, 1):
            inv_stage("I1", s_)
        em.comment("E2' (wave-local): written word by word out of I1's last stage, read in I2's order")
        inv_stage("I1", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (V_L2R, vp(V_A + 2 * k), 8 * k)))
        ck(6)
        for k in visit(3):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_L1R, 136 * k))
        inv_stage("I2", 3, pre=arrive)
        for s_ in (2, 1):
            inv_stage("I2", s_)
        em.comment("E1': written out of I2's last stage (into positions only this wave has read), read in I3's order")
        inv_stage("I2", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (V_L1R, vp(V_A + 2 * k), 136 * k)))
        ck(7)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        for k in visit(3):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_L1W, 2176 * k))
        inv_stage("I3", 3, pre=arrive)
        for s_ in (2, 1):
            inv_stage("I3", s_)
        em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]; written out of I3's last stage")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AXP, V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AXP, AXP))                            # q*32768 + t*8
        inv_stage("I3", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (AXP, vp(V_A + 2 * k), (k // ROW_G) * 2048 * ROW_G + (k % ROW_G) * 2048)))
        ck(8)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, V_OFF8))
        first = order["I0"][:-1]
        for k in (visit(first[0]) if first else range(16)):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * rstep))
        if first:
            inv_stage("I0", first[0], pre=arrive)
            for s_ in first[1:]:
                inv_stage("I0", s_)
        else:
            R("s_waitcnt lgkmcnt(0)")
        ck(9)
    else:
        inv_pass("I1", (3, 2, 1, 0))
        ck(6)
        em.comment("E2'")
        lds_write(em, V_L2R, V_A, 8)
        lds_read(em, V_L1R, V_A, 136)
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I2", (3, 2, 1, 0))
        ck(7)
        em.comment("E1'")
        lds_write(em, V_L1R, V_A, 136)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, V_L1W, V_A, 2176)
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I3", (3, 2, 1, 0))
        ck(8)
        em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                              # q*32768 + t*8
        for k in range(16):
            g_, j = k % ROW_G, k // ROW_G
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(V_A + 2 * k), j * 2048 * ROW_G + g_ * 2048))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        rstep = 2048 * ROW_G                                 # bytes between a reader's consecutive slots
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * rstep))
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I0", order["I0"][:-1])
        ck(9)
    R("s_cmp_eq_u32 s88, %d" % ROW_LG)
    R("s_cbranch_scc1 .Lmerged_last_stage")
    em.comment("r > 2: plain global stage r-2; lazy output for the outer inverse passes")
    tw = ring.get(("I0", 0, 0))
    run_pairs(em, [gs_bfly(V_A + 2 * h, V_A + 2 * (h + 8), tw) for h in range(8)])
    R("s_branch .Lstore")
    em.lines.append(".Lmerged_last_stage:")
    em.comment("n == 16384: stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(V_A + 2 * h, V_A + 2 * (h + 8)) for h in range(8)])
    em.lines.append(".Lstore:")
    R("s_mov_b64 s[86:87], %s" % (S_CROW,))
    for k in range(16):
        R("global_store_dwordx2 v%d, %s, s[86:87]" % (V_OFF8, vp(V_A + 2 * k)))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * ROW_G,))
            R("s_addc_u32 s87, s87, 0")
    R("s_endpgm")
    return em


# ------------------------------------------------------------------ n = 16384, persistent with row prefetch
# One 1024-thread workgroup per CU means a row's loads, its arithmetic and its stores run one after the other (the
# skeleton without butterflies needs 58 % of the kernel's time, profiles/r03_longrow_ablation.txt).  This variant keeps
# the workgroup on the CU and moves the memory phases UNDER the arithmetic:
#   workgroup (x, cm) of a (G, nm) grid walks polynomials x, x + G, ...
#   b is transformed first and alone (file B) while a's row loads are in flight (file A);
#   after the point-wise step file B is free: b of the NEXT polynomial is loaded during the inverse transform;
#   the result is stored pair by pair out of the last stage, and drains under the next polynomial's first passes.
# vmcnt retires in order, so a block of row loads in front of a twiddle wait would make that wait absorb the HBM
# latency (what defeated round 2's persistent 4096-word kernel): the row loads are woven INTO the twiddle stream, one
# per ring issue, so each is waited for nine ring uses after it was issued.  Cost: the forward twiddles are fetched
# once per operand instead of once per pair (+48 records per wave and row).
# MEASURED (profiles/r03_persistent_rows.txt): bit-exact, and SLOWER -- n = 16384 x 8 moduli 459 k against 491 k products/s,
# n = 8192 x 2 moduli 4.06 M against 4.55 M.  The bound was there to read beforehand: with the rows served from the L2
# (no HBM phase at all, and its power back) the shipped kernels gain 18 % / 15 %, most of it clock; what an overlap of the
# memory phases alone can return is a few per cent, less than the second set of twiddle fetches costs.  Emitted only with
# NFL_GEN_EXPERIMENTS=1; tests/asm_emu.py run_block_kernel(grid_x=...) executes it.
# kernarg: c a b psi mc | nm logn | count G        grid (G, nm)
def build_row16k_loop():   # (also the 8192-word rows: ROW_G = 2, 512 threads, two workgroups per CU)
    assert ROW_G in (2, 4) and SINGLE_STREAM
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    base = {"F0": (S_K0["F0"], None, False), "F1": (S_K["F1"], None, False), "F2": (S_K["F2"], V_BIDX, False),
            "F3": (S_K["F3"], V_TID, False), "I1": (S_K["I1"], V_TID, True), "I2": (S_K["I2"], V_BIDX, True),
            "I3": (S_K["I3"], None, True), "I0": (S_K0["I0"], None, True)}
    order = {"F0": tuple(range(ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0), "I0": tuple(range(ROW_LG - 1, 0, -1))}
    per = 16 // ROW_G          # register slots per 4096-word block in the row layout x[tid + 256 G k]
    passes, uses = {}, []
    for tag in "ba":
        for name in ("F0", "F1", "F2", "F3"):
            passes[name + tag] = base[name]
            uses += [(name + tag, s_, g) for s_ in order[name] for g in range(1 << s_)]
    n_fwd = len(uses)
    for name in ("I1", "I2", "I3", "I0"):
        passes[name] = base[name]
        uses += [(name, s_, g) for s_ in order[name] for g in range(1 << s_)]
    S_I, S_G, S_COUNT, S_STRIDE, S_RUN = "s2", "s3", "s96", ("s98", "s99"), ("s100", "s101")

    def side_load(dst_pair):
        def f():
            vm.load("global_load_dwordx2 %s, v%d, s[100:101]" % (vp(dst_pair), V_OFF8))
            R("s_add_u32 s100, s100, 0x%x" % (2048 * ROW_G,))
            R("s_addc_u32 s101, s101, 0")
        return f
    side = {}
    for k in range(16):
        side[RING_SLOTS + k] = [side_load(V_A + 2 * k)]                    # a: under b's forward transform
        side[n_fwd + RING_SLOTS + k] = [side_load(V_B + 2 * k)]            # next b: under the inverse transform
    ring = Ring(em, vm, RING_SLOTS, uses, passes, side)
    R("s_load_dwordx2 s[96:97], s[0:1], 0x30")                             # count, G
    prologue16k(em, vm, None, "none")
    AX = T(0, 0)
    R("s_cmp_ge_u32 %s, %s" % (S_I, S_COUNT))
    R("s_cbranch_scc0 .Lhas_work")
    R("s_endpgm")
    em.lines.append(".Lhas_work:")
    R("s_mov_b32 %s, s97" % S_G)                                            # (cm is not needed any more)
    R("s_mul_i32 s42, %s, s14" % S_G)                                       # G * nm rows of 2^17 bytes between polynomials
    R("s_lshr_b32 %s, s42, %d" % (S_STRIDE[1], 32 - 15 - ROW_LG))
    R("s_lshl_b32 %s, s42, %d" % (S_STRIDE[0], 15 + ROW_LG))
    em.comment("b of the first polynomial (x[tid + 1024 k] -> slot k)")
    R("s_mov_b64 s[86:87], %s" % (S_BROW,))
    for k in range(16):
        vm.load("global_load_dwordx2 %s, v%d, s[86:87]" % (vp(V_B + 2 * k), V_OFF8))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * ROW_G,))
            R("s_addc_u32 s87, s87, 0")
    em.lines.append(".Lnext_polynomial:")
    em.comment("b row of the polynomial after this one (this one again if it is the last: a harmless reload)")
    R("s_add_u32 s52, %s, %s" % (S_I, S_G))
    R("s_cmp_lt_u32 s52, %s" % S_COUNT)
    R("s_cselect_b32 s52, %s, 0" % S_STRIDE[0])
    R("s_cselect_b32 s53, %s, 0" % S_STRIDE[1])
    R("s_add_u32 s18, s18, s52")
    R("s_addc_u32 s19, s19, s53")
    R("s_mov_b64 s[100:101], %s" % (S_AROW,))
    ring.prime()

    def X0(b_):
        em.comment("X0: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
        R("s_barrier")               # WAR: every wave is done reading the previous exchange
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
        for k in range(16):
            qq, j = k // per, k % per
            R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(b_ + 2 * k), (qq & 1) * SLAB_BYTES + j * 2048 * ROW_G))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(b_ + 2 * k), AX, 2048 * k))
        R("s_waitcnt lgkmcnt(0)")

    def E1(b_):
        em.comment("E1")
        R("s_barrier")               # WAR against the previous exchange through this slab
        lds_write(em, V_L1W, b_, 2176)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, V_L1R, b_, 136)
        R("s_waitcnt lgkmcnt(0)")

    def E2(b_):
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        lds_write(em, V_L1R, b_, 136)
        lds_read(em, V_L2R, b_, 8)
        R("s_waitcnt lgkmcnt(0)")

    def fwd_one(b_, tag):
        for name, after in (("F0", X0), ("F1", E1), ("F2", E2), ("F3", None)):
            em.comment("%s, operand %s" % (name, tag))
            for s_ in order[name]:
                half = 8 >> s_
                for g in range(1 << s_):
                    use = (name + tag, s_, g)
                    tw = ring.get(use)
                    run_pairs(em, [ct_bfly(b_ + 2 * (g * 2 * half + h), b_ + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done(use)
            if after:
                after(b_)

    def inv_pass(name):
        em.comment(name)
        for s_ in order[name]:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((name, s_, g))
                run_pairs(em, [gs_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((name, s_, g))

    fwd_one(V_B, "b")
    fwd_one(V_A, "a")
    em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
    run_pairs(em, [pointwise(V_A + 2 * i, V_B + 2 * i, True, True) for i in range(16)])
    R("s_mov_b64 s[100:101], %s" % (S_BROW,))          # file B is free: the ring's side loads now fetch the next b
    inv_pass("I1")
    em.comment("E2'")
    lds_write(em, V_L2R, V_A, 8)
    lds_read(em, V_L1R, V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2")
    em.comment("E1'")
    lds_write(em, V_L1R, V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    lds_read(em, V_L1W, V_A, 2176)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3")
    em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]")
    R("s_barrier")               # every wave is done reading E1'
    R("s_lshl_b32 s86, %s, 15" % (S_Q,))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
    em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                              # q*32768 + t*8
    for k in range(16):
        g_, j = k % ROW_G, k // ROW_G
        R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(V_A + 2 * k), j * 2048 * ROW_G + g_ * 2048))
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    rstep = 2048 * ROW_G
    em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, V_OFF8))
    for k in range(16):
        R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * rstep))
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I0")
    assert ring.next == len(uses) and len(ring.free) == RING_SLOTS
    em.comment("stage 0 with n^-1 folded in; every finished pair is stored at once (the next b has landed long ago)")
    R("s_waitcnt vmcnt(0)")
    for h in range(8):
        run_pairs(em, [final_bfly(V_A + 2 * h, V_A + 2 * (h + 8))])
        for k in (h, h + 8):
            R("s_add_u32 s86, s20, 0x%x" % (k * 2048 * ROW_G,))
            R("s_addc_u32 s87, s21, 0")
            R("global_store_dwordx2 v%d, %s, s[86:87]" % (V_OFF8, vp(V_A + 2 * k)))
    for lo in (16, 20):
        R("s_add_u32 s%d, s%d, %s" % (lo, lo, S_STRIDE[0]))
        R("s_addc_u32 s%d, s%d, %s" % (lo + 1, lo + 1, S_STRIDE[1]))
    R("s_add_u32 %s, %s, %s" % (S_I, S_I, S_G))
    R("s_cmp_lt_u32 %s, %s" % (S_I, S_COUNT))
    R("s_cbranch_scc1 .Lnext_polynomial")
    R("s_endpgm")
    return em


# ------------------------------------------------------------------ 32768-word rows: ONE operand register-resident
# A 32768-word row (256 KiB) is exactly the register footprint the 16384-word kernel manages for TWO operands: one
# 1024-thread workgroup, 32 words per thread in the two coefficient files v[V_A..] / v[V_B..] (64 VGPRs), one butterfly at
# a time, the twiddle records streaming through the 9-slot ring: 128 VGPRs, 4 waves per SIMD.  The row is HBM traffic
# exactly once per direction:
#   F0  radix-8 pass over all 32 slots (global stages r-3 .. r-1): thread tid holds x[tid + 1024 k], k = c + 4 m, i.e.
#       four columns c of the eight 4096-word blocks m; file A = blocks 0..3, file B = blocks 4..7
#   X0  through LDS in TWO rounds (a 32768-word row does not fit the 160 KiB): file A -> the four sub-groups' slabs ->
#       file A of sub-group q = block q; then file B -> block q + 4.  Same addresses as the 16384-word kernel's X0.
#   F1 F2 F3 / I1 I2 I3: the 4096-word passes of the block kernel, once per file (the files are different blocks of ONE
#       row here, so they do not share twiddles: file B's records are the block q + 4 ones, K offset by a constant)
#   X0' in two rounds, I0 radix-8 with the mirrored table, n^-1 folded into the last stage when the row is the whole row.
# kinds: fwd (canonical NTT-form words out), inv, polymul_ntt: c = INTT(NTT(a) (.) b') with b' (already transformed,
# canonical) STREAMED through the twiddle ring during the point-wise step -- the large-row product is then
# b' = fwd(b) (read + write) followed by polymul_ntt(a, b') (two reads + one write): 5 operand passes instead of 9.
def build_row32k(kind="fwd"):
    assert ROW_G == 4 and ROW_LG == 3 and NEXT_VGPR == 128
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    FILES = ((V_A, 0), (V_B, 4))                      # (register base, block offset inside the row)
    DK = {"F1": 1, "F2": 16, "F3": 256, "I1": -256, "I2": -16, "I3": -1}   # dK / d(block) of the pass constants
    inner = {"F1": (S_K["F1"], None, False), "F2": (S_K["F2"], V_BIDX, False), "F3": (S_K["F3"], V_TID, False),
             "I1": (S_K["I1"], V_TID, True), "I2": (S_K["I2"], V_BIDX, True), "I3": (S_K["I3"], None, True)}
    passes = {"F0": (S_K0["F0"], None, False), "I0": (S_K0["I0"], None, True)}
    for f, (_, boff) in enumerate(FILES):
        for name, (kreg, vidx, desc) in inner.items():
            passes[name + "ab"[f]] = (kreg, vidx, desc, DK[name] * boff)

    # b' of the composed product lives in the context's scratch in a layout of OUR choice ("_s" kinds): block-major, then the
    # slot pair i, then the thread -- [block q + boff][i][t] x 16 bytes -- so that the 64 lanes of a wave store / fetch 64
    # consecutive 16-byte pairs (8 cache lines).  In the reference's order (the user-visible one: kinds without "_s") thread t
    # owns words 16t .. 16t + 15, i.e. a lane's pair sits alone in its 128-byte line: 64 lines per load, 16 bytes used of each,
    # half of the product kernel's L1 fills -- and the forward kernel pays two LDS transposes to produce it with coalesced stores.
    scratch_layout = kind.endswith("_s")
    kind = kind[:-2] if scratch_layout else kind
    if scratch_layout:
        kind = {"polymul": "polymul_ntt"}.get(kind, kind)

    def bprime_loader(boff):
        def load(em_, r, s_, i, first):               # words 16t + 2i, 16t + 2i + 1 of block q + boff of b' -> one ring slot
            if first:
                em_.raw("s_lshl_b32 s42, %s, 15" % (S_Q,))
                if boff:
                    em_.raw("s_add_u32 s42, s42, 0x%x" % (boff << 15,))
                em_.raw("s_add_u32 s96, s18, s42")
                em_.raw("s_addc_u32 s97, s19, 0")
                em_.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_TWO, 4 if scratch_layout else 7, V_TID))
            if scratch_layout:
                em_.raw("s_add_u32 s86, s96, 0x%x" % (4096 * i,))
                em_.raw("s_addc_u32 s87, s97, 0")
                return "global_load_dwordx4 v[%d:%d], v%d, s[86:87] nt" % (r, r + 3, V_TWO)
            return "global_load_dwordx4 v[%d:%d], v%d, s[96:97] offset:%d" % (r, r + 3, V_TWO, 16 * i)
        return load
    passes["Ba"], passes["Bb"] = bprime_loader(0), bprime_loader(4)

    compact_x = kind in ("fwd_i8", "fma_fwd_i8", "enc2_i8")   # the row arrives as one signed byte per coefficient (a compact Gaussian polynomial)
    # fma_fwd_i8 / enc2_i8: the transformed row X never leaves the registers -- out0 = X k0 + e0' [, out1 = X k1 + e1'] with the key rows
    # (one polynomial for the batch) and the already transformed noise rows e' streamed through the ring's registers in the store layout
    enc_res = {"fma_fwd_i8": 1, "enc2_i8": 2}.get(kind, 0)
    if compact_x:
        kind = "fwd"
    fused_inv = kind in ("fms_inv", "fma_inv")       # INTT(b - a k) / INTT(b + a k): a at S_AROW, b at S_BROW, the key row at s[98:99]
    has_fwd, has_inv = kind != "inv" and not fused_inv, kind != "fwd"
    uses = []
    if has_fwd:
        uses += [("F0", s_, g) for s_ in range(3) for g in range(1 << s_)]
        for name in ("F1", "F2", "F3"):
            for f in range(2):
                uses += [(name + "ab"[f], s_, g) for s_ in range(4) for g in range(1 << s_)]
    if kind == "polymul_ntt":
        for f in range(2):
            uses += [("B" + "ab"[f], 0, i) for i in range(8)]
    if has_inv:
        for name in ("I1", "I2", "I3"):
            for f in range(2):
                uses += [(name + "ab"[f], s_, g) for s_ in (3, 2, 1, 0) for g in range(1 << s_)]
        uses += [("I0", s_, g) for s_ in (2, 1, 0) for g in range(1 << s_)]
    ring = Ring(em, vm, RING_SLOTS, uses, passes)
    global ALIAS_ROWS
    if BPRIME_ALIAS and scratch_layout:   # fwd_s writes b' through s20, polymul_ntt_s reads it through s18
        ALIAS_ROWS = (20,) if kind == "fwd" else (18,)
    prologue16k(em, vm, None, "none", key_row=fused_inv, compact_x=compact_x)
    ALIAS_ROWS = ()
    AX = T(0, 0)   # exchange address scratch (the butterfly temporaries are idle during exchanges)

    def block_base(srow, boff):                           # s[86:87] = first word of block q + boff of the row at srow
        lo, hi = srow[2:-1].split(":")
        R("s_lshl_b32 s42, %s, 15" % (S_Q,))
        if boff:
            R("s_add_u32 s42, s42, 0x%x" % (boff << 15,))
        R("s_add_u32 s86, s%s, s42" % lo)
        R("s_addc_u32 s87, s%s, 0" % hi)

    n_row_loads = 0
    if has_fwd:
        em.comment("the row: x[tid + 1024 k] -> slot k (8 KiB contiguous per workgroup load)")
        R("s_mov_b64 s[86:87], %s" % (S_AROW,))
        if compact_x:
            em.valu("v_lshrrev_b32_e32 v%d, 3, v%d" % (T(0, 0), V_OFF8))        # tid (V_TID is the index inside the sub-group)
            seq = None
            for k in range(32):
                seq = vm.load("global_load_sbyte v%d, v%d, s[86:87]" % (V_A + 2 * k, T(0, 0)))
                if k < 31:
                    R("s_add_u32 s86, s86, 0x400")
                    R("s_addc_u32 s87, s87, 0")
            vm.wait(seq)
            em.comment("x >= 0 stays, x < 0 becomes p + x: any word congruent to the coefficient is a legal input of the first butterfly")
            t = T(0, 4)
            for k in range(32):
                x = V_A + 2 * k
                em.valu("v_ashrrev_i32_e32 v%d, 31, v%d" % (x + 1, x))
                em.valu("v_and_b32_e32 v%d, s24, v%d" % (t, x + 1))
                em.valu("v_and_b32_e32 v%d, s25, v%d" % (t + 1, x + 1))
                em.valu("v_lshl_add_u64 %s, %s, 0, %s" % (vp(x), vp(x), vp(t)))
        for k in range(32 if not compact_x else 0):
            vm.load("global_load_dwordx2 %s, v%d, s[86:87] nt" % (vp(V_A + 2 * k), V_OFF8))
            if k < 31:
                R("s_add_u32 s86, s86, 0x2000")
                R("s_addc_u32 s87, s87, 0")
    else:
        em.comment("NTT-form words: block element 1024w + 64j + l -> pair j of the block's file (512 B per wave load)")
        g_, _ = lane_contig_setup(em)
        for base, boff in FILES:
            block_base(S_AROW, boff)
            for j in range(16):
                vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d nt" % (vp(base + 2 * j), g_, (j & 7) * 512))
                if j == 7:
                    R("s_add_u32 s86, s86, 0x1000")
                    R("s_addc_u32 s87, s87, 0")
    if fused_inv:
        em.comment("a <- fold(b -+ a k) word by word, in the load layout (the operation is element-wise): b and the key stream through"
                   " the twiddle ring's registers, eight words of each at a time, before the ring is primed")
        for f, (base, boff) in enumerate(FILES):
            for half in range(2):
                g_, _ = lane_contig_setup(em)
                seq = None
                for srow, dst0 in ((S_BROW, V_TW), ("s[98:99]", V_TW + 16)):
                    block_base(srow, boff)
                    if half:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
                    for jj in range(8):
                        seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst0 + 2 * jj), g_, jj * 512, " nt" if srow == S_BROW else ""))
                vm.wait(seq)
                run_pairs(em, [fms_job(base + 2 * (8 * half + jj), V_TW + 16 + 2 * jj, V_TW + 2 * jj, kind == "fms_inv") for jj in range(8)])
    n_row_loads = vm.issued
    ring.prime()

    def fwd_pass(name):
        for f, (base, _) in enumerate(FILES):
            nm_ = name + "ab"[f]
            em.comment("%s, file %s" % (name, "AB"[f]))
            for s_ in range(4):
                half = 8 >> s_
                for g in range(1 << s_):
                    tw = ring.get((nm_, s_, g))
                    run_pairs(em, [ct_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done((nm_, s_, g))

    def inv_pass(name):
        for f, (base, _) in enumerate(FILES):
            nm_ = name + "ab"[f]
            em.comment("%s, file %s" % (name, "AB"[f]))
            for s_ in (3, 2, 1, 0):
                half = 8 >> s_
                for g in range(1 << s_):
                    tw = ring.get((nm_, s_, g))
                    run_pairs(em, [gs_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done((nm_, s_, g))

    # ---- split-phase exchanges: the two files are independent between F0 and I0, so every LDS batch of one file (the
    # writes of an exchange, or its reads) is issued in FRONT of a half pass of arithmetic on the OTHER file and waited for
    # behind it; the arithmetic order -- and with it the order in which the ring consumes its records -- is unchanged.
    # Only the first forward round (file A after F0) and the last inverse round (file B before I0) stay exposed.
    W0 = "s_waitcnt lgkmcnt(0)"

    def fwd_stages(f, name, stages):
        base, nm_ = FILES[f][0], name + "ab"[f]
        em.comment("%s, file %s, stages %s" % (name, "AB"[f], stages))
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((nm_, s_, g))
                run_pairs(em, [ct_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((nm_, s_, g))

    def inv_stages(f, name, stages):
        base, nm_ = FILES[f][0], name + "ab"[f]
        em.comment("%s, file %s, stages %s" % (name, "AB"[f], stages))
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((nm_, s_, g))
                run_pairs(em, [gs_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((nm_, s_, g))

    def X0w(f):
        base = FILES[f][0]
        em.comment("X0 writes, file %s: thread (q, t) slot 4*m + c -> sub-group m, thread t, slot q + 4*c" % "AB"[f])
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
        for k in range(16):
            qq, j = k // 4, k % 4
            R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * SLAB_BYTES + j * 8192))

    def X0r(f):
        base = FILES[f][0]
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))

    def X0iw(f):
        base = FILES[f][0]
        em.comment("X0' writes, file %s: thread (q, t) slot g + 4*j -> thread (g, t) slot 4*q + j, layout [slot][tid]" % "AB"[f])
        R("s_lshl_b32 s86, %s, 15" % (S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                          # q*32768 + t*8
        for k in range(16):
            g_, j = k % 4, k // 4
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(base + 2 * k), j * 8192 + g_ * 2048))

    def X0ir(f):
        base = FILES[f][0]
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * 8192, V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * 8192))

    def seq(*items):          # strings are emitted as they are, callables are called
        for it in items:
            if isinstance(it, str):
                R(it)
            else:
                it()

    def split_phase_schedule():
        A_, B_ = FILES[0][0], FILES[1][0]
        BAR = "s_barrier"
        if has_fwd:
            em.comment("F0: radix-8 over the 32 slots (stage 0 couples the files)")
            for s_ in range(3):
                half = 16 >> s_
                for g in range(1 << s_):
                    tw = ring.get(("F0", s_, g))
                    run_pairs(em, [ct_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done(("F0", s_, g))
            seq(lambda: X0w(0), W0, BAR, lambda: X0r(0), W0, BAR,
                lambda: X0w(1), lambda: fwd_stages(0, "F1", (0, 1)), W0, BAR, lambda: X0r(1), lambda: fwd_stages(0, "F1", (2, 3)), W0, BAR,
                lambda: lds_write(em, V_L1W, A_, 2176), lambda: fwd_stages(1, "F1", (0, 1)), W0, BAR,
                lambda: lds_read(em, V_L1R, A_, 136), lambda: fwd_stages(1, "F1", (2, 3)), W0, BAR,
                lambda: lds_write(em, V_L1W, B_, 2176), lambda: fwd_stages(0, "F2", (0, 1)), W0, BAR,
                lambda: lds_read(em, V_L1R, B_, 136), lambda: fwd_stages(0, "F2", (2, 3)), W0,
                # E2 is wave-local (LDS is in order per wave): file A's transposes run under F2 of file B, B's under F3 of A
                lambda: lds_write(em, V_L1R, A_, 136), lambda: lds_read(em, V_L2R, A_, 8), lambda: fwd_stages(1, "F2", (0, 1, 2, 3)), W0,
                lambda: lds_write(em, V_L1R, B_, 136), lambda: lds_read(em, V_L2R, B_, 8), lambda: fwd_stages(0, "F3", (0, 1, 2, 3)), W0,
                lambda: fwd_stages(1, "F3", (0, 1, 2, 3)))
        if kind == "fwd" and scratch_layout:
            em.comment("canonical words straight into the product's scratch layout [block][pair i][thread]: no transposes")
            em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (V_TWO, V_TID))
            for base, boff in FILES:
                run_pairs(em, [canon(base + 2 * i) for i in range(16)])
                block_base(S_CROW, boff)
                for i in range(8):
                    R("global_store_dwordx4 v%d, v[%d:%d], s[86:87] nt" % (V_TWO, base + 4 * i, base + 4 * i + 3))
                    if i < 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            R("s_endpgm")
            return True
        if kind == "fwd" and enc_res:
            em.comment("X in the store layout (a wave-local LDS transpose per file), then per result: key and noise words in, X k + e' out")
            R("s_load_dwordx8 s[88:95], s[0:1], 0x30")                      # k0 k1 e1' out1 (an aligned group of eight)
            R("s_sub_u32 s42, s20, s4")                                     # the dense rows' offset (this element, this modulus)
            R("s_subb_u32 s43, s21, s5")
            R("s_mov_b32 s96, s3")                                          # the key rows' offset: modulus cm of ONE polynomial (n = 32768)
            R("s_mov_b32 s97, 0")
            R("s_lshl_b64 s[96:97], s[96:97], 18")
            R("s_waitcnt lgkmcnt(0)")
            for lo in (88, 90):
                R("s_add_u32 s%d, s%d, s96" % (lo, lo))
                R("s_addc_u32 s%d, s%d, s97" % (lo + 1, lo + 1))
            for lo in (92, 94):
                R("s_add_u32 s%d, s%d, s42" % (lo, lo))
                R("s_addc_u32 s%d, s%d, s43" % (lo + 1, lo + 1))
            def transposes(base):
                lds_write(em, V_L2R, base, 8)
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
            def fma_stores(base, boff):
                for res in range(enc_res):
                    krow, erow, orow = (("s[88:89]", S_BROW, S_CROW), ("s[90:91]", "s[92:93]", "s[94:95]"))[res]
                    for half in range(2):
                        g_, _ = lane_contig_setup(em)
                        seq = None
                        for srow, dst0, nt_ in ((krow, V_TW, ""), (erow, V_TW + 16, " nt")):
                            block_base(srow, boff)
                            if half:
                                R("s_add_u32 s86, s86, 0x1000")
                                R("s_addc_u32 s87, s87, 0")
                            for jj in range(8):
                                seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst0 + 2 * jj), g_, jj * 512, nt_))
                        vm.wait(seq)
                        run_pairs(em, [fma_job(V_TW + 2 * jj, base + 2 * (8 * half + jj), V_TW + 16 + 2 * jj, res == 0) for jj in range(8)])
                        g_, _ = lane_contig_setup(em)
                        block_base(orow, boff)
                        if half:
                            R("s_add_u32 s86, s86, 0x1000")
                            R("s_addc_u32 s87, s87, 0")
                        for jj in range(8):
                            R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(V_TW + 2 * jj), jj * 512))
            transposes(A_)
            R(W0)
            transposes(B_)
            fma_stores(A_, FILES[0][1])
            R(W0)
            fma_stores(B_, FILES[1][1])
            R("s_endpgm")
            return True
        if kind == "fwd":
            em.comment("canonical words, then a wave-local LDS transpose per file so the stores are fully coalesced; file B's"
                       " reduction runs under file A's transposes")
            def stores(base, boff):
                g_, _ = lane_contig_setup(em)
                block_base(S_CROW, boff)
                for j in range(16):
                    R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(base + 2 * j), (j & 7) * 512))
                    if j == 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            def transposes(base):
                lds_write(em, V_L2R, base, 8)
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
            run_pairs(em, [canon(A_ + 2 * i) for i in range(16)])
            transposes(A_)
            run_pairs(em, [canon(B_ + 2 * i) for i in range(16)])
            R(W0)
            stores(A_, FILES[0][1])
            transposes(B_)
            R(W0)
            stores(B_, FILES[1][1])
            R("s_endpgm")
            return True
        if kind == "polymul_ntt":
            em.comment("point-wise product with b' streamed through the ring: slot i of a file = words 16t + 2i, 16t + 2i + 1 of its block")
            for f, (base, _) in enumerate(FILES):
                for i in range(8):
                    use = ("B" + "ab"[f], 0, i)
                    ring.get(use)
                    r = V_TW + 4 * ring.slot_of[use]
                    run_pairs(em, [pointwise(base + 4 * i, r, True, False), pointwise(base + 4 * i + 2, r + 2, True, False)])
                    ring.done(use)
            seq(lambda: inv_stages(0, "I1", (3, 2, 1, 0)))
        else:
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_row_loads))          # the block loads have landed
            em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region; file B's under I1 of file A")
            def to_threads(base):
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, S_SLAB, l_))
                for j in range(16):
                    R("ds_write_b64 v%d, %s offset:%d" % (l_, vp(base + 2 * j), 544 * j))
                lds_read(em, V_L2R, base, 8)
            seq(lambda: to_threads(A_), W0, lambda: to_threads(B_), lambda: inv_stages(0, "I1", (3, 2, 1, 0)), W0)
        seq(# E2' is wave-local: file A's under I1 of file B, file B's under I2 of file A
            lambda: lds_write(em, V_L2R, A_, 8), lambda: lds_read(em, V_L1R, A_, 136), lambda: inv_stages(1, "I1", (3, 2, 1, 0)), W0,
            lambda: lds_write(em, V_L2R, B_, 8), lambda: lds_read(em, V_L1R, B_, 136), lambda: inv_stages(0, "I2", (3, 2, 1, 0)), W0,
            lambda: lds_write(em, V_L1R, A_, 136), lambda: inv_stages(1, "I2", (3, 2)), W0, BAR,
            lambda: lds_read(em, V_L1W, A_, 2176), lambda: inv_stages(1, "I2", (1, 0)), W0, BAR,
            lambda: lds_write(em, V_L1R, B_, 136), lambda: inv_stages(0, "I3", (3, 2)), W0, BAR,
            lambda: lds_read(em, V_L1W, B_, 2176), lambda: inv_stages(0, "I3", (1, 0)), W0, BAR,
            lambda: X0iw(0), lambda: inv_stages(1, "I3", (3, 2)), W0, BAR, lambda: X0ir(0), lambda: inv_stages(1, "I3", (1, 0)), W0, BAR,
            lambda: X0iw(1), W0, BAR, lambda: X0ir(1), W0)
        return False

    if SPLIT32K:
        if split_phase_schedule():
            return em
    else:
        if has_fwd:
            em.comment("F0: radix-8 over the 32 slots (stage 0 couples the files)")
            for s_ in range(3):
                half = 16 >> s_
                for g in range(1 << s_):
                    tw = ring.get(("F0", s_, g))
                    run_pairs(em, [ct_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done(("F0", s_, g))
            for i, (base, _) in enumerate(FILES):
                em.comment("X0 round %d: thread (q, t) slot 4*m + c of this file -> sub-group m, thread t, slot q + 4*c" % i)
                if i:
                    R("s_barrier")       # WAR: the slabs are still being read for the previous file
                em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
                for k in range(16):
                    qq, j = k // 4, k % 4
                    R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * SLAB_BYTES + j * 8192))
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
                for k in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F1")
            for base, _ in FILES:
                em.comment("E1")
                R("s_barrier")           # WAR against the previous exchange through this slab
                lds_write(em, V_L1W, base, 2176)
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                lds_read(em, V_L1R, base, 136)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F2")
            em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
            for base, _ in FILES:
                lds_write(em, V_L1R, base, 136)
                lds_read(em, V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F3")
        if kind == "fwd":
            em.comment("canonical words, then a wave-local LDS transpose per file so the stores are fully coalesced")
            run_pairs(em, [canon(V_A + 2 * i) for i in range(32)])
            for base, boff in FILES:
                lds_write(em, V_L2R, base, 8)
                g_, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
                R("s_waitcnt lgkmcnt(0)")
                block_base(S_CROW, boff)
                for j in range(16):
                    R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(base + 2 * j), (j & 7) * 512))
                    if j == 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            R("s_endpgm")
            return em

        if kind == "polymul_ntt":
            em.comment("point-wise product with b' streamed through the ring: slot i of a file = words 16t + 2i, 16t + 2i + 1 of its block")
            for f, (base, _) in enumerate(FILES):
                for i in range(8):
                    use = ("B" + "ab"[f], 0, i)
                    ring.get(use)
                    r = V_TW + 4 * ring.slot_of[use]
                    run_pairs(em, [pointwise(base + 4 * i, r, True, False), pointwise(base + 4 * i + 2, r + 2, True, False)])
                    ring.done(use)
        else:
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_row_loads))          # the block loads have landed
            em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region, file by file")
            _, l_ = lane_contig_setup(em)
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, S_SLAB, l_))
            for base, _ in FILES:
                for j in range(16):
                    R("ds_write_b64 v%d, %s offset:%d" % (l_, vp(base + 2 * j), 544 * j))
                lds_read(em, V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
        inv_pass("I1")
        em.comment("E2'")
        for base, _ in FILES:
            lds_write(em, V_L2R, base, 8)
            lds_read(em, V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        inv_pass("I2")
        for i, (base, _) in enumerate(FILES):
            em.comment("E1'")
            if i:
                R("s_barrier")           # WAR: the slab is still being read for the previous file
            lds_write(em, V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
        inv_pass("I3")
        for i, (base, _) in enumerate(FILES):
            em.comment("X0' round %d: thread (q, t) slot g + 4*j of this file -> thread (g, t) slot 4*q + j, layout [slot][tid]" % i)
            R("s_barrier")               # every wave is done reading the previous exchange
            R("s_lshl_b32 s86, %s, 15" % (S_Q,))
            em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
            em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                          # q*32768 + t*8
            for k in range(16):
                g_, j = k % 4, k // 4
                R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(base + 2 * k), j * 8192 + g_ * 2048))
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * 8192, V_OFF8))
            for k in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * 8192))
            R("s_waitcnt lgkmcnt(0)")
    em.comment("I0: radix-8 over the 32 slots, mirrored table")
    for s_ in (2, 1):
        half = 16 >> s_
        for g in range(1 << s_):
            tw = ring.get(("I0", s_, g))
            run_pairs(em, [gs_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
            ring.done(("I0", s_, g))
    R("s_cmp_eq_u32 s88, %d" % ROW_LG)
    R("s_cbranch_scc1 .Lmerged_last_stage")
    em.comment("r > 3: plain global stage r-3; lazy output for the outer inverse passes")
    tw = ring.get(("I0", 0, 0))
    run_pairs(em, [gs_bfly(V_A + 2 * h, V_A + 2 * (h + 16), tw) for h in range(16)])
    R("s_branch .Lstore")
    em.lines.append(".Lmerged_last_stage:")
    em.comment("n == 32768: stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(V_A + 2 * h, V_A + 2 * (h + 16)) for h in range(16)])
    em.lines.append(".Lstore:")
    R("s_mov_b64 s[86:87], %s" % (S_CROW,))
    for k in range(32):
        R("global_store_dwordx2 v%d, %s, s[86:87] nt" % (V_OFF8, vp(V_A + 2 * k)))
        if k < 31:
            R("s_add_u32 s86, s86, 0x2000")
            R("s_addc_u32 s87, s87, 0")
    R("s_endpgm")
    return em


# ------------------------------------------------------------------ n = 65536: the three-role pipeline kernel
# Long rows need streaming radix-16 passes around the fused 4096-word block kernel, and the two kinds of work bound
# different resources (HBM vs integer VALU).  Kernels from different streams do not interleave on a CU in practice
# (DESIGN.md), so ONE launch carries all three kinds of workgroups, interleaved by workgroup index:
#   role 0  V   fused product of one 4096-word block of chunk j-1   (operands already passed through role 1/2)
#   role 1,2 F  forward radix-16 pass (global stages 0-3) of 256 columns of operand a / b of chunk j   (src -> dst)
#   role 3  I   inverse radix-16 pass (global stages 3-0, n^-1 folded in) of 256 columns of c of chunk j-2, in place
# Consecutive launches on one stream form the pipeline; inside a launch the roles are independent.
# kernarg: c_v a_v b_v psi mc | nm (logn unused) | cntV cntF cntI pad | fa_src fa_dst fb_src fb_dst inv_data pad
# grid: (28 * max(cnt), nm): wgx = 28*poly + w.
PIPE_LOGN = 16


def emit_mc_load(em):
    R = em.raw
    R("s_mul_i32 s42, s3, 0x70")
    R("s_add_u32 s42, s12, s42")
    R("s_addc_u32 s43, s13, 0")
    R("s_load_dwordx16 s[56:71], s[42:43], 0x0")          # p p2 mu ninv ninv_sh w1ninv w1ninv_sh beta
    R("s_load_dwordx8 s[72:79], s[42:43], 0x40")          # beta_sh yinv yinv_sh mask
    R("s_load_dwordx4 s[80:83], s[42:43], 0x60")          # delta mu2


def emit_consts(em):
    R = em.raw
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[24:25], s[56:57]")                    # p
    R("s_mov_b64 s[26:27], s[58:59]")                    # 2p
    R("s_add_u32 s28, s58, s56")                         # 3p
    R("s_addc_u32 s29, s59, s57")
    R("s_mov_b32 s30, s80")                              # delta
    R("s_mov_b32 s31, 0x3fffffff")
    R("s_mov_b32 s15, 0xc0000000")
    R("s_mov_b64 s[32:33], s[82:83]")                    # mu2
    R("s_mov_b64 s[34:35], s[62:63]")                    # ninv
    R("s_mov_b64 s[36:37], s[64:65]")                    # ninv_sh
    R("s_mov_b64 s[38:39], s[66:67]")                    # w1ninv
    R("s_mov_b64 s[40:41], s[68:69]")                    # w1ninv_sh
    em.valu("v_mov_b32_e32 v%d, s25" % (V_PHI,))
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        em.valu("v_mov_b32_e32 v%d, 0x3fffffff" % (v_mask(),))


def legacy_role_map(em, PER_ROW, NV, NSW, b_ntt=False):
    R = em.raw
    # Dense role map, no idle workgroups (a workgroup launch costs ~35 ns of dispatcher time chip-wide, measured):
    # 28 workgroups per polynomial row -- w = wgx mod 28: 0..15 block products, 16..19 / 20..23 forward streaming of
    # a / b (four column groups each), 24..27 inverse streaming.  28 = 4 mod 8, so the XCD of a role rotates with the
    # polynomial index and every role is spread over all XCDs.
    R("s_mul_hi_u32 s86, s2, 0x%x" % ((1 << 32) // PER_ROW + 1,))   # poly = wgx / PER_ROW (exact below 1.7e8)
    R("s_mul_i32 s43, s86, %d" % PER_ROW)
    R("s_sub_u32 s89, s2, s43")                          # w
    R("s_mov_b32 s42, 0")                                # role 0: block product, blk = w
    R("s_cmp_lt_u32 s89, %d" % NV)
    R("s_cbranch_scc1 .Lrole_known")
    R("s_sub_u32 s89, s89, %d" % NV)
    R("s_lshr_b32 s42, s89, %d" % (NSW.bit_length() - 1))
    R("s_add_u32 s42, s42, 1")                           # role 1, 2, 3
    if b_ntt:                                            # (no forward role for b: the second streaming role is the inverse one)
        R("s_cmp_eq_u32 s42, 2")
        R("s_cselect_b32 s42, 3, s42")
    R("s_and_b32 s89, s89, %d" % (NSW - 1))              # q: column groups q, q+NSW, q+2 NSW, q+3 NSW
    em.lines.append(".Lrole_known:")
    R("s_mul_i32 s87, s86, s14")
    R("s_add_u32 s87, s87, s3")                          # row = poly*nm + cm
    R("s_lshl_b32 s43, s3, %d" % (PIPE_LOGN + 4,))       # tw = psi + cm * n * 16
    R("s_add_u32 s22, s10, s43")
    R("s_addc_u32 s23, s11, 0")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc1 .Lrole_v")
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc1 .Lrole_i")


# ------------------------------------------------------------------ one launch, rows pinned to an XCD
# The two-pass plan for rows that do not fit one CU moves every word 3 times (operand -> scratch -> scratch -> result):
# 9 word transfers per 3 algorithmic ones when the scratch lives in HBM.  Here the scratch of a row lives in the L2 of
# ONE XCD for the few microseconds between its producer and its consumer:
#   * row g of the batch (modulus-major: g = cm * batch + poly, so all XCDs work on the same modulus at the same time and
#     its twiddles stay in every L2) is job g / 8 of XCD g mod 8.  The grid is a fixed number of PERSISTENT workgroups;
#     each reads its XCC_ID once and then serves that XCD's jobs, whatever the placement of the workgroups.
#   * a job is 2 NSW forward streaming roles, then NV block products, then NSW inverse streaming roles.  Per XCD and kind
#     there is a CREDIT counter (roles that may start) and a TICKET counter (roles handed out, in job order).  A free
#     workgroup (its wave 0) reads the credits with one load, takes one with an atomic subtract (undone if it lost a race)
#     in the order inverse > product > forward -- inverse-first drains rows as fast as they mature -- and then draws the
#     next ticket of that kind.  Nothing spins on a shared word while work is available, and no atomic ever has to be
#     retried: hand-out is two fetch-and-adds.
#   * credits are posted by the role that completes a stage of a job (it sees the per-slot completion counter reach the
#     stage's size): forward -> NV product credits, product -> NSW inverse credits, inverse -> 2 NSW forward credits for
#     the job that reuses the scratch slot (R slots per XCD, job j uses slot j mod R).  Stages may complete out of job
#     order while tickets are in job order, so a role re-checks its own job's inputs before touching them; if k stages
#     have completed, the tickets of the first k jobs' roles of that stage have all been handed out (tickets are in
#     order), hence such a wait is only ever for roles that are already running: no deadlock.
#   * a role publishes "done" with one atomic add after all its stores were acknowledged by the L2 (s_waitcnt vmcnt(0) +
#     s_barrier); producer and consumer share the L2, nothing is written back in between.  The consumer's L1 is the one
#     cache that is not coherent with it, and it is kept out of the way by construction instead of by invalidation: every
#     row has its OWN scratch rows (the scratch mirrors the batch), so within a launch a scratch word is loaded by exactly
#     one workgroup after its last write, on a CU that either never touched the line or wrote it itself (block product:
#     reads a'[k], writes c'[k] over it -- write-through keeps its own L1 current); L1s start a launch invalidated.
#     What was measured on the way (tools/probes/l2_flag_probe.hip, profiles/README): workgroup-scope (sc0) loads hit
#     in the L1 and never see another CU's update; device-scope (sc1) loads and atomics are served memory-side (0.15 -
#     0.5 us) -- scratch read with sc1 loads was correct but moved MORE HBM bytes than the chunked pipeline (3.8x vs 3.3x
#     the algorithmic bytes); `buffer_inv sc0` does not reliably drop stale lines (wrong words in 4 of 9 runs).
#     The ring only bounds the rows in flight (R per domain): its slots index the completion counters.
# Kernel arguments after the standard seven: rows, batch, ceil(2^32 / batch), log2 D | Rlog, -, spin limit, - | scrA, scrB,
# ctl, trace buffer (or null).  D = scheduling domains per XCD (each with its own record, jobs and ring; they only share the
# L2).  ctl: +64 + 4 xcd: workgroups that joined; the record of domain d = xcd + 8 sub at byte 4096 + 69632 d (zeroed by the host):
#   +0 credits {forward (biased by the initial min(R, jobs) * 2 NSW), product, inverse}, +12 exit flag, +16 trace count
#   +128 tickets {forward, product, inverse}      +256 + 16 slot: completed {forward, product, inverse} roles (all epochs)
FUSED_NT = int(os.environ.get("NFL_FUSED_NT", "1"))
FUSED_LIFO = False                       # scratch rows come from a per-XCD pool, lowest free slot first (lifo_* below)
FUSED_LOADS = ""                         # modifier of the scratch loads: " sc1" = device scope (L1 bypass), "" = plain after a buffer_inv sc0
LDS_TICKET = (4096 + 256) * 8           # 64 B behind the exchange slab: wave 0's decision and the running role's completion record



# ---- per-XCD scratch pool (FUSED_LIFO): 32 row slots per XCD, a free mask at ctl + 128 + 4 xcd.  A row's a' and b' live
# in two slots from its first forward role until every block product has LOADED them (signalled a few microseconds into
# the product, after its first barrier), c' in a third from the first product's store to the last inverse role's end.
# "Lowest free slot first" keeps the set of slots in use as small as the concurrency allows, so a slot is rewritten
# while its previous (dead, dirty) contents still sit in the L2 -- the lines are overwritten there instead of being
# written back.  Consumers read the scratch with `nt` loads: measured (tools/probes) to miss the L1 and see other CUs'
# stores, which a slot that is reused within a launch needs.
# Aux record of a job at record + 1024 + 16 slot: {1 + job, (a slot + 1) | (b slot + 1) << 8, c word, products that loaded};
# c word: 0 none, bit 31 = a product is allocating it, low byte = c slot + 1.
def lifo_mask_addr(em, dst_pair, ctl_pair, tmp):
    """dst = ctl + 128 + 4 * xcd"""
    R = em.raw
    lo = int(dst_pair[2:].split(":")[0])
    clo = int(ctl_pair[2:].split(":")[0])
    R("s_and_b32 %s, s98, 7" % tmp)
    R("s_lshl_b32 %s, %s, 2" % (tmp, tmp))
    R("s_add_u32 %s, %s, 128" % (tmp, tmp))
    R("s_add_u32 s%d, s%d, %s" % (lo, clo, tmp))
    R("s_addc_u32 s%d, s%d, 0" % (lo + 1, clo + 1))


def lifo_pop(em, name, vt, mask_pair, out, t0, t1, spin):
    """out = index of a free slot, now taken (one lane active); bounded"""
    R = em.raw
    L = em.lines.append
    R("s_mov_b32 %s, 0" % spin)
    L(".Lpop_%s:" % name)
    R("global_load_dword v%d, v%d, %s sc1" % (vt, V_ZERO, mask_pair))
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 %s, v%d" % (t0, vt))
    R("s_cmp_lg_u32 %s, 0" % t0)
    R("s_cbranch_scc1 .Lpop_%s_try" % name)
    R("s_sleep 8")
    R("s_add_u32 %s, %s, 1" % (spin, spin))
    R("s_cmp_lt_u32 %s, 0x200000" % spin)
    R("s_cbranch_scc1 .Lpop_%s" % name)
    R("s_trap 2")                                        # the pool never refills: fail loudly
    L(".Lpop_%s_try:" % name)
    R("s_ff1_i32_b32 %s, %s" % (out, t0))
    R("s_lshl_b32 %s, 1, %s" % (t1, out))
    R("s_not_b32 %s, %s" % (t0, t1))
    R("v_mov_b32_e32 v%d, %s" % (vt, t0))
    R("global_atomic_and v%d, v%d, v%d, %s sc0" % (vt, V_ZERO, vt, mask_pair))
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 %s, v%d" % (t0, vt))
    R("s_and_b32 %s, %s, %s" % (t0, t0, t1))
    R("s_cmp_lg_u32 %s, 0" % t0)
    R("s_cbranch_scc0 .Lpop_%s" % name)                  # somebody else took that slot first


def lifo_slot_addr(em, dst_lo, slot_sgpr, scr_pair, tmp, NB):
    """s[dst_lo:dst_lo+1] = scr + ((xcd * 32 + slot) << NB)"""
    R = em.raw
    slo = int(scr_pair[2:].split(":")[0])
    R("s_and_b32 %s, s98, 7" % tmp)
    R("s_lshl_b32 %s, %s, 5" % (tmp, tmp))
    R("s_add_u32 %s, %s, %s" % (tmp, tmp, slot_sgpr))
    R("s_lshr_b32 s%d, %s, %d" % (dst_lo + 1, tmp, 32 - NB))
    R("s_lshl_b32 s%d, %s, %d" % (dst_lo, tmp, NB))
    R("s_add_u32 s%d, s%d, s%d" % (dst_lo, dst_lo, slo))
    R("s_addc_u32 s%d, s%d, s%d" % (dst_lo + 1, dst_lo + 1, slo + 1))


def lifo_product_loaded(em, NV):
    """injected after the block product's first barrier: its a' / b' blocks are in registers.  The last product of the
    job to get here returns both slots to the pool.  s[96:97] = the job's aux record, s100 = the two slots' bits."""
    R = em.raw
    L = em.lines.append
    R("v_readfirstlane_b32 s42, v%d" % V_TID)
    R("s_cmp_lg_u32 s42, 0")
    R("s_cbranch_scc1 .Lvl_done")
    R("s_mov_b64 exec, 1")
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v7, v%d, v7, s[96:97] offset:12 sc0" % V_ZERO)
    R("s_load_dwordx2 s[46:47], s[0:1], 0x60")           # ctl
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v7")
    R("s_cmp_eq_u32 s42, %d" % (NV - 1))
    R("s_cbranch_scc0 .Lvl_restore")
    lifo_mask_addr(em, "s[46:47]", "s[46:47]", "s42")
    R("v_mov_b32_e32 v7, s100")
    R("global_atomic_or v%d, v7, s[46:47]" % V_ZERO)
    L(".Lvl_restore:")
    R("s_mov_b64 exec, -1")
    L(".Lvl_done:")


def lifo_product_store(em, NB):
    """injected in front of the block product's stores: learn (or allocate) the job's c' slot, point S_CROW at this
    product's block of it.  Every wave runs it (no workgroup exchange needed): v[40:41] are free by now (b is consumed)."""
    R = em.raw
    L = em.lines.append
    R("s_load_dwordx2 s[46:47], s[0:1], 0x60")           # ctl
    R("s_load_dwordx2 s[52:53], s[0:1], 0x50")           # scratch pool
    R("s_mov_b64 exec, 1")
    R("v_bfrev_b32_e32 v40, 1")                          # 0x80000000
    R("global_atomic_or v40, v%d, v40, s[96:97] offset:8 sc0" % V_ZERO)
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v40")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc0 .Lcs_wait")
    # first product of the job to finish: take a slot, publish it
    lifo_mask_addr(em, "s[46:47]", "s[46:47]", "s43")
    lifo_pop(em, "c", 40, "s[46:47]", "s44", "s42", "s43", "s45")
    R("s_add_u32 s42, s44, 1")
    R("v_mov_b32_e32 v40, s42")
    R("global_atomic_or v%d, v40, s[96:97] offset:8" % V_ZERO)
    R("s_branch .Lcs_known")
    L(".Lcs_wait:")
    R("s_mov_b32 s45, 0")
    L(".Lcs_poll:")
    R("s_and_b32 s44, s42, 0xff")
    R("s_cmp_lg_u32 s44, 0")
    R("s_cbranch_scc1 .Lcs_have")
    R("s_sleep 2")
    R("global_load_dword v40, v%d, s[96:97] offset:8 sc1" % V_ZERO)
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s42, v40")
    R("s_add_u32 s45, s45, 1")
    R("s_cmp_lt_u32 s45, 0x200000")
    R("s_cbranch_scc1 .Lcs_poll")
    R("s_trap 2")
    L(".Lcs_have:")
    R("s_sub_u32 s44, s44, 1")
    L(".Lcs_known:")
    R("s_mov_b64 exec, -1")
    lifo_slot_addr(em, 20, "s44", "s[52:53]", "s42", NB)
    R("s_lshl_b32 s42, s89, 15")
    R("s_add_u32 s20, s20, s42")
    R("s_addc_u32 s21, s21, 0")                          # S_CROW: block s89 of the c' slot


def fused_header(em, PER_ROW, NV, NSW, CG_LOG):
    R = em.raw
    L = em.lines.append
    NB = PIPE_LOGN + 3                                    # log2 bytes of a row
    LI, LV, LF = NSW.bit_length() - 1, NV.bit_length() - 1, NSW.bit_length()   # log2 roles per job: inverse, product, forward
    Z = V_ZERO
    T = LDS_TICKET      # +0 kind, +4 ticket | +16 counter offset (0: none), +20 target, +24 credit offset, +28 amount | +32 id, +36 t0, +40 t1

    def lane0():
        R("s_mov_b64 exec, 1")

    def all_lanes():
        R("s_mov_b64 exec, -1")

    def poll(name, off_sgpr, want_sgpr):
        """wait until the dword at record + off_sgpr equals want_sgpr (normally true at once); bounded"""
        R("s_add_u32 s84, s72, %s" % off_sgpr)
        R("s_addc_u32 s85, s73, 0")
        R("s_mov_b32 s92, 0")
        L(".Lpoll_%s:" % name)
        R("global_load_dword v7, v%d, s[84:85] sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s91, v7")
        R("s_cmp_eq_u32 s91, %s" % want_sgpr)
        R("s_cbranch_scc1 .Lpoll_%s_done" % name)
        R("s_sleep 4")
        R("s_add_u32 s92, s92, 1")
        R("s_cmp_lt_u32 s92, s62")
        R("s_cbranch_scc1 .Lpoll_%s" % name)
        R("s_trap 2")                                    # an input that never completes: fail loudly, do not hang
        L(".Lpoll_%s_done:" % name)

    def take(kind, cdw, bias, nxt):
        """wave 0, lane 0 active: take one credit of counter cdw (effective value = stored + bias SGPR or 0), then a ticket"""
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_sub v12, v%d, v12, s[72:73] offset:%d sc0" % (Z, 4 * cdw))
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s84, v12")
        if bias:
            R("s_add_u32 s84, s84, %s" % bias)
        R("s_cmp_gt_i32 s84, 0")
        R("s_cbranch_scc1 .Ltook_%d" % kind)
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_add v%d, v12, s[72:73] offset:%d" % (Z, 4 * cdw))   # lost the race for the last credit: give it back
        R("s_branch %s" % nxt)
        L(".Ltook_%d:" % kind)
        R("v_mov_b32_e32 v12, 1")
        R("global_atomic_add v12, v%d, v12, s[72:73] offset:%d sc0" % (Z, 128 + 4 * cdw))
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s83, v12")
        R("s_mov_b32 s82, %d" % kind)
        R("s_branch .Ldecided")

    def stamp_t1(name):
        """trace: the role's inputs are ready (wave 0 keeps the stamp in LDS)"""
        R("v_readfirstlane_b32 s84, v%d" % V_TID)
        R("s_cmp_lg_u32 s84, 0")
        R("s_cbranch_scc1 .Lt1_%s" % name)
        R("s_memtime s[84:85]")
        lane0()
        R("s_waitcnt lgkmcnt(0)")
        R("v_mov_b32_e32 v8, s84")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 40))
        all_lanes()
        L(".Lt1_%s:" % name)

    # scheduling domain of this workgroup: 2^s59 independent domains per XCD (own record, own jobs, own ring) keep the
    # atomic traffic per record line low; workgroups of an XCD join them round-robin.  s98 = domain = xcd + 8 * sub
    R("s_getreg_b32 s98, hwreg(HW_REG_XCC_ID, 0, 4)")
    R("s_and_b32 s98, s98, 7")
    R("s_load_dwordx16 s[56:71], s[0:1], 0x30")
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s42, s98, 2")
    R("s_add_u32 s42, s42, 64")
    R("s_add_u32 s72, s68, s42")
    R("s_addc_u32 s73, s69, 0")                          # ctl + 64 + 4 xcd: workgroups of this XCD seen so far
    R("v_readfirstlane_b32 s74, v%d" % V_TID)
    R("s_cmp_lg_u32 s74, 0")
    R("s_cbranch_scc1 .Ldom_wait")
    lane0()
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v7, v%d, v7, s[72:73] sc0" % Z)
    R("v_mov_b32_e32 v8, 0")
    R("s_waitcnt vmcnt(0)")
    R("ds_write_b32 v%d, v7 offset:%d" % (Z, T))
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 16))    # no completion to publish yet
    R("s_waitcnt lgkmcnt(0)")
    all_lanes()
    L(".Ldom_wait:")
    R("s_barrier")
    R("ds_read_b32 v7, v%d offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s74, v7")
    R("s_lshl_b32 s75, 1, s59")
    R("s_sub_u32 s75, s75, 1")
    R("s_and_b32 s74, s74, s75")                         # sub
    R("s_lshl_b32 s74, s74, 3")
    R("s_add_u32 s98, s98, s74")
    R("s_barrier")                                       # (wave 0 reuses the LDS word)
    R("s_branch .Lticket")
    L(".Lnext:")
    R("s_waitcnt vmcnt(0) lgkmcnt(0)")                   # this wave's stores are in the L2
    R("s_barrier")
    L(".Lticket:")
    R("s_load_dwordx16 s[56:71], s[0:1], 0x30")
    R("s_waitcnt lgkmcnt(0)")
    R("s_mul_i32 s42, s98, 0x11000")
    R("s_add_u32 s42, s42, 4096")                        # records 68 KiB apart (different memory channels)
    R("s_add_u32 s72, s68, s42")
    R("s_addc_u32 s73, s69, 0")                          # s[72:73]: this XCD's record
    R("s_add_u32 s93, s59, 3")                           # log2 of the number of domains
    R("s_lshl_b32 s42, 1, s93")
    R("s_sub_u32 s42, s42, 1")
    R("s_sub_u32 s99, s56, s98")
    R("s_add_u32 s99, s99, s42")
    R("s_lshr_b32 s99, s99, s93")                        # jobs of this domain: rows dom, dom + 8 D, ...  (rows >= 8 D checked by the host)
    R("v_readfirstlane_b32 s74, v%d" % V_TID)
    R("s_cmp_lg_u32 s74, 0")
    R("s_cbranch_scc1 .Lsched_done")                     # waves 1..3 wait at the barrier for wave 0's decision
    # ---- wave 0: publish the finished role (and the credits it releases), then find the next one
    R("s_memtime s[86:87]")
    lane0()
    R("ds_read_b128 v[8:11], v%d offset:%d" % (Z, T + 16))  # counter offset, target, credit offset, amount
    R("ds_read_b128 v[14:17], v%d offset:%d" % (Z, T + 32)) # id, t0, t1, -
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s74, v8")
    R("s_cmp_eq_u32 s74, 0")
    R("s_cbranch_scc1 .Lt_noflag")
    R("v_readfirstlane_b32 s75, v9")
    R("v_readfirstlane_b32 s76, v10")
    R("v_readfirstlane_b32 s77, v11")
    R("s_add_u32 s84, s72, s74")
    R("s_addc_u32 s85, s73, 0")
    R("v_mov_b32_e32 v12, 1")
    R("global_atomic_add v12, v%d, v12, s[84:85] sc0" % Z)  # the role just finished: one more "done"
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s78, v12")
    R("s_add_u32 s78, s78, 1")
    R("s_cmp_eq_u32 s78, s75")
    R("s_cbranch_scc0 .Lt_posted")                       # not the last role of its stage
    if FUSED_LIFO:
        R("v_readfirstlane_b32 s79, v17")                # (v[14:17] = id, t0, t1, slots to free)
        R("s_cmp_eq_u32 s79, 0")
        R("s_cbranch_scc1 .Lt_nofree")
        lifo_mask_addr(em, "s[84:85]", "s[68:69]", "s80")
        R("v_mov_b32_e32 v12, s79")
        R("global_atomic_or v%d, v12, s[84:85]" % Z)     # the inverse stage is complete: its c' slot returns to the pool
        L(".Lt_nofree:")
    R("s_cmp_eq_u32 s77, 0")
    R("s_cbranch_scc1 .Lt_posted")
    R("s_add_u32 s84, s72, s76")
    R("s_addc_u32 s85, s73, 0")
    R("v_mov_b32_e32 v12, s77")
    R("global_atomic_add v%d, v12, s[84:85]" % Z)        # the next stage of that job (or the slot's next job) may start
    L(".Lt_posted:")
    # optional trace record {ticket | kind << 28, t0, t1, t2} (low words of s_memtime), 16 B per role, 2^16 per XCD
    R("s_cmp_eq_u64 s[70:71], 0")
    R("s_cbranch_scc1 .Lt_noflag")
    R("v_mov_b32_e32 v12, 1")
    R("global_atomic_add v12, v%d, v12, s[72:73] offset:16 sc0" % Z)
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s74, v12")
    R("s_and_b32 s74, s74, 0xffff")
    R("s_lshl_b32 s75, s98, 16")
    R("s_or_b32 s74, s74, s75")
    R("s_lshl_b32 s74, s74, 4")
    R("s_add_u32 s74, s70, s74")
    R("s_addc_u32 s75, s71, 0")
    R("v_mov_b32_e32 v17, s86")
    R("global_store_dwordx4 v%d, v[14:17], s[74:75]" % Z)
    L(".Lt_noflag:")
    R("v_mov_b32_e32 v8, 0")
    R("v_mov_b32_e32 v9, s86")
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 16))    # nothing to publish until a role is set up
    R("ds_write_b32 v%d, v9 offset:%d" % (Z, T + 36))    # t0: this workgroup is free
    R("s_mov_b32 s88, 0")                                # polls so far
    R("s_lshl_b32 s79, 1, s60")                          # R
    R("s_min_u32 s79, s79, s99")
    R("s_lshl_b32 s79, s79, %d" % LF)                    # forward credits the host's zero stands for: min(R, jobs) * 2 NSW
    L(".Lsched:")
    R("global_load_dwordx4 v[8:11], v%d, s[72:73] sc1" % Z)   # sc1: device scope; plain and sc0 loads hit in the L1
    R("s_waitcnt vmcnt(0)")
    R("v_readfirstlane_b32 s77, v10")                    # inverse credits
    R("s_cmp_gt_i32 s77, 0")
    R("s_cbranch_scc0 .Lsee_v")
    take(3, 2, None, ".Lsee_v")
    L(".Lsee_v:")
    R("v_readfirstlane_b32 s76, v9")                     # product credits
    R("s_cmp_gt_i32 s76, 0")
    R("s_cbranch_scc0 .Lsee_f")
    take(0, 1, None, ".Lsee_f")
    L(".Lsee_f:")
    R("v_readfirstlane_b32 s75, v8")                     # forward credits (biased)
    R("s_add_u32 s75, s75, s79")
    R("s_cmp_gt_i32 s75, 0")
    R("s_cbranch_scc0 .Lsee_exit")
    take(1, 0, "s79", ".Lsee_exit")
    L(".Lsee_exit:")
    R("v_readfirstlane_b32 s78, v11")
    R("s_cmp_eq_u32 s78, 0")
    R("s_cbranch_scc1 .Lnothing")
    R("s_mov_b32 s82, 4")                                # every inverse role has been handed out: done
    R("s_mov_b32 s83, 0")
    R("s_branch .Ldecided")
    L(".Lnothing:")
    R("s_sleep 8")
    R("s_cmp_lt_u32 s88, 8")
    R("s_cbranch_scc1 .Lnothing_short")
    R("s_sleep 60")                                      # nothing for a while: poll every ~2 us
    L(".Lnothing_short:")
    R("s_add_u32 s88, s88, 1")
    R("s_cmp_lt_u32 s88, s62")
    R("s_cbranch_scc1 .Lsched")
    R("s_trap 2")                                        # nothing became ready for seconds: fail loudly, do not hang
    L(".Ldecided:")
    R("v_mov_b32_e32 v10, s82")
    R("v_mov_b32_e32 v11, s83")
    R("ds_write_b64 v%d, v[10:11] offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    all_lanes()
    L(".Lsched_done:")
    R("s_barrier")
    R("ds_read_b64 v[10:11], v%d offset:%d" % (Z, T))
    R("s_waitcnt lgkmcnt(0)")
    R("v_readfirstlane_b32 s42, v10")                    # kind: 0 product, 1 forward, 3 inverse, 4 exit
    R("v_readfirstlane_b32 s2, v11")                     # role number within its kind
    R("s_cmp_eq_u32 s42, 4")
    R("s_cbranch_scc0 .Lwork")
    R("S_EXIT")
    L(".Lwork:")
    # ---- job, slot and sub-index of the role; its completion record
    #      s74 job, s77 slot, s78 epoch, s89 sub-index; s75 byte offset of the counter to bump, s76 its value when the stage
    #      is complete, s80 the credit word that stage completion feeds, s81 how many credits
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc0 .Ldec_not_i")
    R("s_lshr_b32 s74, s2, %d" % LI)
    R("s_and_b32 s89, s2, %d" % (NSW - 1))
    R("s_mov_b32 s75, 8")
    R("s_mov_b32 s76, %d" % NSW)
    R("s_mov_b32 s80, 0")                                # -> forward credits of the job that reuses the slot ...
    R("s_lshl_b32 s81, 1, s60")
    R("s_add_u32 s81, s81, s74")
    R("s_cmp_lt_u32 s81, s99")                           # ... if there is one
    R("s_cselect_b32 s81, %d, 0" % (2 * NSW))
    R("s_add_u32 s43, s2, 1")
    R("s_lshl_b32 s83, s99, %d" % LI)
    R("s_cmp_eq_u32 s43, s83")                           # the XCD's last inverse role: tell the idle workgroups to leave
    R("s_cbranch_scc0 .Ldec_done")
    R("v_readfirstlane_b32 s43, v%d" % V_TID)
    R("s_cmp_lg_u32 s43, 0")
    R("s_cbranch_scc1 .Ldec_done")
    lane0()
    R("v_mov_b32_e32 v7, 1")
    R("global_atomic_add v%d, v7, s[72:73] offset:12" % Z)
    all_lanes()
    R("s_branch .Ldec_done")
    L(".Ldec_not_i:")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc0 .Ldec_f")
    R("s_lshr_b32 s74, s2, %d" % LV)
    R("s_and_b32 s89, s2, %d" % (NV - 1))
    R("s_mov_b32 s75, 4")
    R("s_mov_b32 s76, %d" % NV)
    R("s_mov_b32 s80, 8")                                # -> inverse credits
    R("s_mov_b32 s81, %d" % NSW)
    R("s_branch .Ldec_done")
    L(".Ldec_f:")
    R("s_lshr_b32 s74, s2, %d" % LF)
    R("s_and_b32 s89, s2, %d" % (2 * NSW - 1))
    R("s_mov_b32 s75, 0")
    R("s_mov_b32 s76, %d" % (2 * NSW))
    R("s_mov_b32 s80, 4")                                # -> product credits
    R("s_mov_b32 s81, %d" % NV)
    L(".Ldec_done:")
    R("s_lshl_b32 s43, 1, s60")
    R("s_sub_u32 s43, s43, 1")
    R("s_and_b32 s77, s74, s43")                         # slot
    R("s_lshr_b32 s78, s74, s60")                        # epoch
    R("s_lshl_b32 s43, s77, 4")
    R("s_add_u32 s43, s43, 256")                         # the slot's counters
    R("s_add_u32 s75, s75, s43")
    R("s_add_u32 s83, s78, 1")
    R("s_mul_i32 s76, s76, s83")                         # the counter's value when this job's stage is complete
    R("v_readfirstlane_b32 s84, v%d" % V_TID)
    R("s_cmp_lg_u32 s84, 0")
    R("s_cbranch_scc1 .Lrec_done")
    lane0()
    R("v_mov_b32_e32 v8, s75")
    R("v_mov_b32_e32 v9, s76")
    R("v_mov_b32_e32 v10, s80")
    R("v_mov_b32_e32 v11, s81")
    R("ds_write_b128 v%d, v[8:11] offset:%d" % (Z, T + 16))
    R("s_lshl_b32 s84, s42, 28")
    R("s_and_b32 s85, s2, 0xfffffff")
    R("s_or_b32 s84, s84, s85")
    R("v_mov_b32_e32 v8, s84")
    R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 32))
    if FUSED_LIFO:
        R("v_mov_b32_e32 v8, 0")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 44))  # pool slots to free when this role completes its stage (set by the inverse role)
    all_lanes()
    L(".Lrec_done:")
    # ---- the job's row: g = 8 D job + domain (modulus-major)
    R("s_add_u32 s84, s59, 3")
    R("s_lshl_b32 s84, s74, s84")
    R("s_add_u32 s84, s84, s98")
    R("s_mul_hi_u32 s3, s84, s58")                       # cm = g / batch
    R("s_mul_i32 s43, s3, s57")
    R("s_sub_u32 s86, s84, s43")                         # poly
    R("s_mul_i32 s87, s86, s14")
    R("s_add_u32 s87, s87, s3")                          # row = poly*nm + cm
    R("s_lshl_b32 s43, s3, %d" % (PIPE_LOGN + 4,))
    R("s_add_u32 s22, s10, s43")
    R("s_addc_u32 s23, s11, 0")                          # twiddles of the modulus
    R("s_lshr_b32 s83, s87, %d" % (32 - NB))
    R("s_lshl_b32 s82, s87, %d" % NB)                    # s[82:83]: byte offset of the row in the batch ...
    R("s_mov_b64 s[80:81], s[82:83]")                    # ... and in the scratch, which mirrors the batch (see above)
    R("s_lshl_b32 s43, s77, 4")
    R("s_add_u32 s79, s43, 256")                         # byte offset of the slot's counters in the record
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc1 .Lprep_v")
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc1 .Lprep_i")
    # ---- forward streaming role: operand s89 >> log NSW, column groups q = s89 mod NSW; the slot must be drained
    R("s_lshl_b32 s76, s78, %d" % LI)                    # inverse roles completed on the slot by earlier epochs
    R("s_add_u32 s75, s79, 8")
    poll("slot", "s75", "s76")
    if FUSED_LIFO:
        # the job's first forward role takes two slots from the XCD's pool and publishes them; everybody reads them
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")                      # s[96:97]: the job's aux record
        R("s_add_u32 s76, s74, 1")                       # tag = job + 1
        R("s_cmp_lg_u32 s89, 0")
        R("s_cbranch_scc1 .Lf_slots")
        R("v_readfirstlane_b32 s43, v%d" % V_TID)
        R("s_cmp_lg_u32 s43, 0")
        R("s_cbranch_scc1 .Lf_slots")
        lane0()
        lifo_mask_addr(em, "s[84:85]", "s[68:69]", "s43")
        lifo_pop(em, "a", 7, "s[84:85]", "s80", "s42", "s43", "s81")
        lifo_pop(em, "b", 7, "s[84:85]", "s91", "s42", "s43", "s81")
        R("s_add_u32 s80, s80, 1")
        R("s_add_u32 s91, s91, 1")
        R("s_lshl_b32 s91, s91, 8")
        R("s_or_b32 s80, s80, s91")
        R("v_mov_b32_e32 v8, 0")
        R("v_mov_b32_e32 v9, 0")
        R("global_atomic_swap_x2 v%d, v[8:9], s[96:97] offset:8" % Z)   # c word, products that loaded
        R("v_mov_b32_e32 v7, s80")
        R("global_atomic_swap v%d, v7, s[96:97] offset:4" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_mov_b32_e32 v7, s76")
        R("global_atomic_swap v%d, v7, s[96:97]" % Z)    # the tag last: the record is valid for this job
        all_lanes()
        L(".Lf_slots:")
        R("s_mov_b32 s92, 0")
        L(".Lf_slots_poll:")
        R("global_load_dwordx2 v[10:11], v%d, s[96:97] sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s91, v10")
        R("v_readfirstlane_b32 s80, v11")
        R("s_cmp_eq_u32 s91, s76")
        R("s_cbranch_scc1 .Lf_slots_known")
        R("s_sleep 2")
        R("s_add_u32 s92, s92, 1")
        R("s_cmp_lt_u32 s92, s62")
        R("s_cbranch_scc1 .Lf_slots_poll")
        R("s_trap 2")
        L(".Lf_slots_known:")
        R("s_lshr_b32 s42, s89, %d" % LI)                # operand: 0 = a, 1 = b
        R("s_and_b32 s89, s89, %d" % (NSW - 1))
        R("s_lshl_b32 s43, s42, 3")
        R("s_lshr_b32 s80, s80, s43")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")                       # the operand's slot
        lifo_slot_addr(em, 20, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)            # the bytes of q column groups
        R("s_add_u32 s20, s20, s43")
        R("s_addc_u32 s21, s21, 0")
        R("s_add_u32 s82, s82, s43")                     # (no carry: the low bits were zero)
        R("s_cmp_eq_u32 s42, 0")
        R("s_cselect_b64 s[16:17], s[6:7], s[8:9]")
        R("s_add_u32 s16, s16, s82")
        R("s_addc_u32 s17, s17, s83")
    else:
        R("s_lshr_b32 s42, s89, %d" % LI)
        R("s_and_b32 s89, s89, %d" % (NSW - 1))
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)                # the bytes of q column groups
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s82, s82, s43")                         # (no carries: the low bits were zero)
        R("s_cmp_eq_u32 s42, 0")
        R("s_cselect_b64 s[16:17], s[6:7], s[8:9]")
        R("s_cselect_b64 s[20:21], s[64:65], s[66:67]")
        R("s_add_u32 s16, s16, s82")
        R("s_addc_u32 s17, s17, s83")
        R("s_add_u32 s20, s20, s80")
        R("s_addc_u32 s21, s21, s81")
    R("s_mov_b32 s90, 1")
    stamp_t1("f")
    R("s_branch .Lbody_f")
    L(".Lprep_i:")
    R("s_add_u32 s76, s78, 1")
    R("s_lshl_b32 s76, s76, %d" % LV)                    # every block product of the job
    R("s_add_u32 s75, s79, 4")
    poll("vdone", "s75", "s76")
    if FUSED_LIFO:
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")
        R("global_load_dword v7, v%d, s[96:97] offset:8 sc1" % Z)       # the c word (published before any product completed)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s80, v7")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")
        R("v_readfirstlane_b32 s43, v%d" % V_TID)
        R("s_cmp_lg_u32 s43, 0")
        R("s_cbranch_scc1 .Li_free_noted")
        lane0()
        R("s_lshl_b32 s43, 1, s80")
        R("v_mov_b32_e32 v8, s43")
        R("ds_write_b32 v%d, v8 offset:%d" % (Z, T + 44))   # returned to the pool by whoever completes the inverse stage
        all_lanes()
        L(".Li_free_noted:")
        lifo_slot_addr(em, 16, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)
        R("s_add_u32 s16, s16, s43")
        R("s_addc_u32 s17, s17, 0")
        R("s_add_u32 s82, s82, s43")
    else:
        R("s_lshl_b32 s43, s89, %d" % CG_LOG)
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s82, s82, s43")
        R("s_add_u32 s16, s64, s80")
        R("s_addc_u32 s17, s65, s81")
    R("s_add_u32 s20, s4, s82")
    R("s_addc_u32 s21, s5, s83")
    R("s_mov_b32 s95, 2")
    stamp_t1("i")
    R("s_branch .Lbody_i")
    L(".Lprep_v:")
    R("s_add_u32 s76, s78, 1")
    R("s_lshl_b32 s76, s76, %d" % LF)                    # every forward role of the job
    R("s_mov_b32 s75, s79")
    poll("fdone", "s75", "s76")
    if FUSED_LIFO:
        R("s_lshl_b32 s43, s77, 4")
        R("s_add_u32 s43, s43, 1024")
        R("s_add_u32 s96, s72, s43")
        R("s_addc_u32 s97, s73, 0")                      # s[96:97]: the job's aux record (kept through the role)
        R("global_load_dword v7, v%d, s[96:97] offset:4 sc1" % Z)
        R("s_waitcnt vmcnt(0)")
        R("v_readfirstlane_b32 s80, v7")
        R("s_and_b32 s81, s80, 0xff")
        R("s_sub_u32 s81, s81, 1")                       # a' slot
        R("s_lshr_b32 s80, s80, 8")
        R("s_and_b32 s80, s80, 0xff")
        R("s_sub_u32 s80, s80, 1")                       # b' slot
        R("s_lshl_b32 s100, 1, s81")
        R("s_lshl_b32 s43, 1, s80")
        R("s_or_b32 s100, s100, s43")                    # both bits: returned to the pool once every product has loaded
        lifo_slot_addr(em, 16, "s81", "s[64:65]", "s43", NB)
        lifo_slot_addr(em, 18, "s80", "s[64:65]", "s43", NB)
        R("s_lshl_b32 s43, s89, 15")
        R("s_add_u32 s16, s16, s43")
        R("s_addc_u32 s17, s17, 0")
        R("s_add_u32 s18, s18, s43")
        R("s_addc_u32 s19, s19, 0")
        R("s_mov_b64 s[20:21], 0")                       # (the c' block is known when the stores start)
    else:
        R("s_lshl_b32 s43, s89, 15")
        R("s_add_u32 s80, s80, s43")
        R("s_add_u32 s16, s64, s80")
        R("s_addc_u32 s17, s65, s81")
        R("s_add_u32 s18, s66, s80")
        R("s_addc_u32 s19, s67, s81")
        R("s_mov_b64 s[20:21], s[16:17]")                    # the block product overwrites its a' block
    stamp_t1("v")
    R("s_branch .Lbody_v")


def build_pipe(logn=None, fused=False, b_ntt=False):
    """n = 65536 (logn 16): radix-16 streaming roles, 16 + 3 x 4 = 28 workgroups per row.
    n = 32768 (logn 15): radix-8 streaming roles (a thread's 16 registers hold two columns of 8 words), 8 + 3 x 2 = 14.
    fused: ONE launch of persistent workgroups for the whole batch; the three roles of a row run on ONE XCD, ordered by
    a per-XCD ticket queue and per-row completion counters, so the intermediates travel through that XCD's L2
    (see fused_header below).
    b_ntt: operand b is ALREADY transformed (canonical words in the reference's order): there is no forward streaming role
    for it -- NV + 2 NSW workgroups per row -- and the block products read its block as it lies (16 consecutive words per
    thread: what the inner forward passes would have left in the registers), like the stand-alone polymul_ntt kernel."""
    global PIPE_LOGN
    if logn is not None:
        PIPE_LOGN = logn
    em = Emitter()
    R = em.raw
    n_words = 1 << PIPE_LOGN
    RL = PIPE_LOGN - 12                                   # global stages done by the streaming roles: 4 (radix 16) or 3 (radix 8)
    RADIX = 1 << RL
    NV = n_words // 4096                                  # block products per row
    NSW = 4 if RL == 4 else 2                             # streaming workgroups per row and operand
    PER_ROW = NV + (2 if b_ntt else 3) * NSW
    assert not (fused and b_ntt)
    CG_LOG = 11 if RL == 4 else 12                        # bytes (log2) of one column group: 256 columns x (16 / RADIX) x 8 B
    stride = n_words // RADIX * 8                         # bytes between x[o + k n/RADIX]
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c_v, a_v, b_v, psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x20")           # mc
    R("s_load_dword s14, s[0:1], 0x28")                  # nm
    if not fused:
        R("s_load_dwordx16 s[56:71], s[0:1], 0x30")      # cntV cntF cntI pad | fa_src fa_dst fb_src fb_dst inv pad
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_OFF8, V_TID))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (V_BIDX, V_TID))                     # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (V_L1W, V_TID, V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1W, V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (V_L1R, V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_L1R, V_BIDX, V_L2R, V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1R, V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_L2R, V_TID, V_L2R))              # 17*t*8
    for t0 in sorted(set(V_T)):
        em.valu("v_mov_b32_e32 v%d, 0" % (t0 + 15,))                                # the persistent zero of ZP
    R("s_waitcnt lgkmcnt(0)")
    if fused:
        fused_header(em, PER_ROW, NV, NSW, CG_LOG)
    else:
        # Workgroups are dealt to the 8 XCDs round-robin by their linear index, so with the plain (wgx, cm) grid every XCD works
        # on every modulus and every private L2 fetches every twiddle table once per pass: at n = 65536 x 30 moduli that is
        # 8 x 31 MB per pass, a third of the operand bytes again (3.34 x the algorithmic traffic where the plan moves 3.0).
        # Remap: unit u = cm gx + wgx (modulus-major); XCD slot k = L mod 8 takes the CONTIGUOUS units [k U/8, (k+1) U/8):
        # an XCD then walks through ~nm/8 moduli, one after the other.  kernarg: gx (0 = off), U/8, ceil(2^32 / gx).
        R("s_cmp_eq_u32 s59, 0")
        R("s_cbranch_scc1 .Lno_remap")
        R("s_mul_i32 s42, s3, s59")
        R("s_add_u32 s42, s42, s2")                          # L
        R("s_and_b32 s43, s42, 7")
        R("s_lshr_b32 s42, s42, 3")
        R("s_mul_i32 s43, s43, s70")
        R("s_add_u32 s42, s42, s43")                         # u
        R("s_mul_hi_u32 s3, s42, s71")                       # cm = u / gx
        R("s_mul_i32 s43, s3, s59")
        R("s_sub_u32 s2, s42, s43")                          # wgx = u mod gx
        em.lines.append(".Lno_remap:")
        legacy_role_map(em, PER_ROW, NV, NSW, b_ntt)
    stream_setup = {}
    # ---------------------------------------------------------------- streaming roles
    # A streaming workgroup owns the four column groups sub, sub+4, sub+8, sub+12 of its row (sub < 4; the others
    # exit at once) and double-buffers them through the a / b register files: the loads of group g+1 are in flight
    # while group g is transformed, and the 15 twiddle records of the pass are loaded once.
    GROUPS, GSTEP = 4, NSW << CG_LOG

    def group_io(buf, srow, offset, store=False):
        """one column group <-> 16 register pairs: radix 16: x[t + k n/16], k < 16; radix 8: x[t + k n/8] in pairs 0..7 and
        x[t + 256 + k n/8] in pairs 8..15"""
        if RL == 4:
            return strided_rows(em, vm_cur[0], buf, srow, stride, store=store, offset=offset)
        seq = 0
        for half in range(2):
            seq = strided_rows(em, vm_cur[0], buf + 16 * half, srow, stride, store=store, offset=offset + 2048 * half, nwords=8)
        return seq

    def radix_stage_fwd(buf, st):
        if RL == 4:
            return ct_stage(em, [buf], st)
        half = 4 >> st
        jobs = []
        for g in range(1 << st):
            tw = twreg(tw_slot(st, g))
            for h in range(half):
                for grp in (0, 8):
                    i0 = grp + g * 2 * half + h
                    jobs.append(ct_bfly(buf + 2 * i0, buf + 2 * (i0 + half), tw))
        run_pairs(em, jobs)

    def radix_stage_inv(buf, st):
        if RL == 4:
            return gs_stage(em, buf, st)
        half = 4 >> st
        jobs = []
        for g in range(1 << st):
            tw = twreg(tw_slot(st, g))
            for h in range(half):
                for grp in (0, 8):
                    i0 = grp + g * 2 * half + h
                    jobs.append(gs_bfly(buf + 2 * i0, buf + 2 * (i0 + half), tw))
        run_pairs(em, jobs)

    vm_cur = [None]

    def stream_role(kind):
        vm = VmCounter(em)
        vm_cur[0] = vm
        bufs = [V_A, V_B]
        seq_of = {0: group_io(bufs[0], S_AROW, 0)}
        tw_last = 0
        for st in (tuple(range(RL)) if kind == "F" else tuple(range(RL - 1, -1, -1))):
            tw_last = PASS_TW["F1" if kind == "F" else "I3"](em, vm, st)
        emit_consts(em)
        for gi in range(GROUPS):
            buf = bufs[gi & 1]
            if gi + 1 < GROUPS:
                seq_of[gi + 1] = group_io(bufs[(gi + 1) & 1], S_AROW, (gi + 1) * GSTEP)
            vm.wait(max(seq_of[gi], tw_last))
            if kind == "F":
                for st in range(RL):
                    radix_stage_fwd(buf, st)
            else:
                for st in range(RL - 1, 0, -1):
                    radix_stage_inv(buf, st)
                if RL == 4:
                    run_pairs(em, [final_bfly(buf + 2 * h, buf + 2 * (h + 8)) for h in range(8)])
                else:
                    run_pairs(em, [final_bfly(buf + 2 * (grp + h), buf + 2 * (grp + h + 4)) for h in range(4) for grp in (0, 8)])
            group_io(buf, S_CROW, gi * GSTEP, store=True)
        R("s_endpgm")

    em.comment("role F: x[o + k n/16] -> radix-16 over global stages 0..3 -> lazy words (the block kernel takes any word)")
    if fused:
        em.lines.append(".Lbody_f:")
    else:
        R("s_cmp_ge_u32 s86, s57")
        R("s_cbranch_scc1 .Lidle")
        R("s_cmp_eq_u32 s42, 1")
        R("s_cselect_b64 s[16:17], s[60:61], s[64:65]")      # src
        R("s_cselect_b64 s[20:21], s[62:63], s[66:67]")      # dst
        R("s_lshr_b32 s43, s87, %d" % (32 - (PIPE_LOGN + 3),))
        R("s_lshl_b32 s42, s87, %d" % (PIPE_LOGN + 3,))      # row * n * 8
        R("s_lshl_b32 s86, s89, %d" % CG_LOG)
        R("s_add_u32 s42, s42, s86")                         # + the bytes of q column groups (no carry: the low bits were zero)
        if SCRATCH_ALIAS:   # ablation: the scratch rows of the whole batch laid over a window of SCRATCH_ALIAS rows (cache-resident)
            R("s_add_u32 s16, s16, s42")
            R("s_addc_u32 s17, s17, s43")
            R("s_and_b32 s44, s87, %d" % (SCRATCH_ALIAS - 1,))
            R("s_lshl_b32 s44, s44, %d" % (PIPE_LOGN + 3,))
            R("s_add_u32 s44, s44, s86")
            R("s_add_u32 s20, s20, s44")
            R("s_addc_u32 s21, s21, 0")
        else:
            for row in (16, 20):
                R("s_add_u32 s%d, s%d, s42" % (row, row))
                R("s_addc_u32 s%d, s%d, s43" % (row + 1, row + 1))
        R("s_mov_b32 s90, 1")                                # K_F1 of the row's first four stages
    emit_mc_load(em)
    mark = len(em.lines)
    stream_role("F")
    if fused and FUSED_NT:   # the operands are read once; the scratch they are written to is what the L2 should keep
        em.lines[mark:] = [l + " nt" if "global_load_dwordx2" in l else l for l in em.lines[mark:]]

    em.lines.append(".Lrole_i:")
    em.comment("role I: lazy words of the block kernel -> global stages 3..0 with n^-1 -> canonical x[o + k n/16]")
    if fused:
        em.lines.append(".Lbody_i:")
    else:
        R("s_cmp_ge_u32 s86, s58")
        R("s_cbranch_scc1 .Lidle")
        R("s_lshr_b32 s43, s87, %d" % (32 - (PIPE_LOGN + 3),))
        R("s_lshl_b32 s42, s87, %d" % (PIPE_LOGN + 3,))
        R("s_lshl_b32 s86, s89, %d" % CG_LOG)
        R("s_add_u32 s42, s42, s86")
        R("s_add_u32 s16, s68, s42")
        R("s_addc_u32 s17, s69, s43")
        R("s_mov_b64 s[20:21], s[16:17]")
        R("s_mov_b32 s95, 2")                                # K_I3 of the row's last four stages
    emit_mc_load(em)
    mark = len(em.lines)
    stream_role("I")
    if fused:   # the scratch comes from another CU of the XCD: read it from the L2, not from this CU's L1
        em.lines[mark:] = [l + (" nt" if FUSED_LIFO else FUSED_LOADS) if "global_load_dwordx2" in l else l for l in em.lines[mark:]]
    if fused and FUSED_NT:   # ... and the result is written once
        em.lines[mark:] = [l + " nt" if "global_store_dwordx2" in l else l for l in em.lines[mark:]]

    # ---------------------------------------------------------------- role 0: the fused block product
    em.lines.append(".Lrole_v:")
    em.comment("role V: one 4096-word block, exactly the stand-alone block kernel (r = 4, blk = s89)")
    if fused:
        em.lines.append(".Lbody_v:")
    else:
        R("s_cmp_ge_u32 s86, s56")
        R("s_cbranch_scc1 .Lidle")
        R("s_lshl_b32 s42, s87, %d" % RL)
        R("s_add_u32 s42, s42, s89")                         # block index = row * (n / 4096) + blk
        R("s_lshr_b32 s43, s42, 17")
        R("s_lshl_b32 s42, s42, 15")
        if SCRATCH_ALIAS and not b_ntt:
            R("s_and_b32 s44, s87, %d" % (SCRATCH_ALIAS - 1,))
            R("s_lshl_b32 s44, s44, %d" % RL)
            R("s_add_u32 s44, s44, s89")
            R("s_lshl_b32 s44, s44, 15")
            for base, row in ((6, 16), (8, 18)):
                R("s_add_u32 s%d, s%d, s44" % (row, base))
                R("s_addc_u32 s%d, s%d, 0" % (row + 1, base + 1))
            R("s_add_u32 s20, s4, s42")
            R("s_addc_u32 s21, s5, s43")
        else:
            for base, row in ((6, 16), (8, 18), (4, 20)):
                R("s_add_u32 s%d, s%d, s42" % (row, base))
                R("s_addc_u32 s%d, s%d, s43" % (row + 1, base + 1))
    R("s_mov_b32 s88, %d" % (PIPE_LOGN - 12,))
    R("s_lshl_b32 s90, 1, s88")
    R("s_add_u32 s90, s90, s89")                         # Kf = 2^r + blk
    R("s_lshl_b32 s91, s90, 4")
    R("s_lshl_b32 s92, s90, 8")
    R("s_lshl_b32 s93, 0x200, s88")
    R("s_lshl_b32 s42, s89, 8")
    R("s_sub_u32 s93, s93, s42")                         # (512<<r) - 256*blk
    R("s_lshl_b32 s94, 32, s88")
    R("s_lshl_b32 s42, s89, 4")
    R("s_sub_u32 s94, s94, s42")                         # (32<<r) - 16*blk
    R("s_lshl_b32 s95, 2, s88")
    R("s_sub_u32 s95, s95, s89")                         # (2<<r) - blk
    emit_mc_load(em)
    vm = VmCounter(em)
    mark_v = len(em.lines)
    strided_rows(em, vm, V_A, S_AROW, 2048)
    if b_ntt:
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_TID))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (V_B + 4 * i, V_B + 4 * i + 3, T(1, 0), S_BROW, 16 * i))
    else:
        strided_rows(em, vm, V_B, S_BROW, 2048)
    tw_seq = {}
    for st in range(4):
        tw_seq[("F1", st)] = PASS_TW["F1"](em, vm, st)
    emit_consts(em)
    mark = len(em.lines)
    build_body(em, vm, "polymul_ntt" if b_ntt else "polymul", tw_seq, "_v")
    if fused:
        mod = " nt" if FUSED_LIFO else FUSED_LOADS
        em.lines[mark_v:mark] = [l + mod if "global_load_dwordx2" in l else l for l in em.lines[mark_v:mark]]
    if fused and FUSED_LIFO:
        # splice the pool protocol into the product: "loaded" after its first barrier, the c' slot in front of its stores
        body = em.lines[mark:]
        e1, e2 = Emitter(), Emitter()
        lifo_product_loaded(e1, NV)
        lifo_product_store(e2, PIPE_LOGN + 3)
        b = next(i for i, l in enumerate(body) if l.strip() == "s_barrier")
        body[b + 1:b + 1] = e1.lines
        st = next(i for i, l in enumerate(body) if l.strip() == ".Lstore:")
        body[st + 1:st + 1] = e2.lines
        em.lines[mark:] = body
    em.lines.append(".Lidle:")
    R("s_endpgm")
    if fused:   # every role ends by drawing the next ticket; the only exit is the VOID inverse role of the header
        em.lines = ["\ts_branch .Lnext" if l.strip() == "s_endpgm" else l for l in em.lines]
        em.lines = ["\ts_endpgm" if l.strip() == "S_EXIT" else l for l in em.lines]
    return em


# ------------------------------------------------------------------ transform-fused pipelines (n = 4096, one row per workgroup)
# What callers of the reference run around the transforms (tests/nfllib_demo_main_op.cpp:26-58: three Gaussian polynomials,
# three forward transforms and two multiply-adds per encryption; one multiply-subtract and one inverse transform per
# decryption) as ONE launch per batch: [expand | load] -> forward passes in registers -> point-wise step against key rows ->
# store, or load -> point-wise step -> inverse passes -> store.  The intermediate polynomials never reach HBM.
#   fma_fwd   out0 = NTT(x0) * k0 + NTT(x1)
#   enc2      out0 = NTT(x0) * k0 + NTT(x1),  out1 = NTT(x0) * k1 + NTT(x2)      (x0' stays in registers for the second half)
#   fms_inv   out0 = INTT(x1 - x0 * k0)        fma_inv   out0 = INTT(x1 + x0 * k0)
# Every operand advances by its own stride (in polynomials) from one batch element to the next: 0 = one polynomial for the
# whole batch (a key), 1 = dense.  A forward input x is either full residue words (format 0: [nm][n] words in coefficient
# form) or ONE signed integer per coefficient shared by all moduli (formats 1 / 2 / 3: int8 / int16 / int32 -- what the
# samplers produce before they are spread over the moduli, core.hpp:230-277; x < 0 is expanded to p + x).
# kernarg: out0 out1 x0 x1 x2 k0 k1 psi mc | nm logn fmt (4 bits per x) | strides x0 x1 x2 k0 k1 | count magic (prologue_fused)
ARGS_FUSED = [("ptr", 8 * i) for i in range(9)] + [("i32", 72 + 4 * i) for i in range(10)]
S_FMT, S_F = "s4", "s5"
S_X2ROW, S_K0ROW, S_K1ROW, S_O1ROW = "s[54:55]", "s[96:97]", "s[98:99]", "s[100:101]"


def prologue_fused(em, vm, kind):
    """256 threads, workgroup (x, y) = (batch element, modulus).  Leaves the row pointers, the pass constants (r = 0), the
    ModConst record requested, the first pass's twiddle loads issued and -- forward kinds -- x0 / x1 on their way into V_A /
    V_B; returns (tw_seq, sequence number of the last operand load)"""
    R = em.raw
    fwd = kind in ("enc2", "fma_fwd")
    R("s_load_dwordx16 s[56:71], s[0:1], 0x0")           # out0 out1 x0 x1 x2 k0 k1 psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x40")           # mc
    R("s_load_dwordx2 s[72:73], s[0:1], 0x48")           # nm, logn
    R("s_load_dwordx8 s[76:83], s[0:1], 0x50")           # fmt, strides x0 x1 x2 k0 k1, count, magic
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_OFF8, V_TID))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (V_BIDX, V_TID))                     # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (V_L1W, V_TID, V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1W, V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (V_L1R, V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_L1R, V_BIDX, V_L2R, V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1R, V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_L2R, V_TID, V_L2R))              # 17*t*8
    for s in sorted(set(V_T)):
        em.valu("v_mov_b32_e32 v%d, 0" % (s + 15,))                                 # the persistent zero of ZP
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b32 s14, s72")                              # nm
    # workgroup -> (batch element, modulus).  magic = 0: the grid is (batch, nm).  Otherwise a 1-D grid of nm * 8 * ceil(count / 8)
    # workgroups dealt so that the nm rows of one batch element run BACK TO BACK ON ONE XCD (workgroups go to the XCDs round-robin
    # by linear index): L = 8 q + xcd, q = nm j + cm, element = 8 j + xcd -- the compact inputs the nm rows share are then
    # fetched from HBM once, by that XCD's L2 (j = q / nm by one multiply: magic = 2^32 / nm + 1, exact below 2^32 / nm)
    R("s_cmp_eq_u32 s83, 0")
    R("s_cbranch_scc1 .Lplain_grid")
    R("s_and_b32 s42, s2, 7")                            # xcd
    R("s_lshr_b32 s43, s2, 3")                           # q
    R("s_mul_hi_u32 s44, s43, s83")                      # j
    R("s_mul_i32 s45, s44, s14")
    R("s_sub_u32 s3, s43, s45")                          # cm
    R("s_lshl_b32 s44, s44, 3")
    R("s_add_u32 s2, s44, s42")                          # element
    R("s_cmp_lt_u32 s2, s82")
    R("s_cbranch_scc1 .Lplain_grid")
    R("s_endpgm")                                        # padding of the last group of eight
    em.lines.append(".Lplain_grid:")
    R("s_sub_u32 s88, s73, 12")                          # r = 0: rows of exactly 4096 words
    R("s_mov_b32 s89, 0")                                # blk
    R("s_mov_b32 %s, s76" % S_FMT)
    R("s_mov_b64 s[10:11], s[70:71]")                    # psi

    def word_row(dst, base, stride):
        """s[dst:dst+1] = base + (((x * stride) * nm + y) << 15); stride None = dense"""
        R("s_mul_i32 s42, s2, s%d" % stride if stride is not None else "s_mov_b32 s42, s2")
        R("s_mul_hi_u32 s43, s42, s14")
        R("s_mul_i32 s42, s42, s14")
        R("s_add_u32 s42, s42, s3")
        R("s_addc_u32 s43, s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], 15")
        R("s_add_u32 s%d, s%d, s42" % (dst, base))
        R("s_addc_u32 s%d, s%d, s43" % (dst + 1, base + 1))

    def x_row(dst, base, stride, k):
        """the same for a forward input: its format decides between word rows and the compact (x * stride) << (11 + f)"""
        if not fwd:
            return word_row(dst, base, stride)
        R("s_bfe_u32 %s, %s, 0x%x" % (S_F, S_FMT, (4 << 16) | (4 * k)))
        R("s_mul_i32 s42, s2, s%d" % stride)
        R("s_mul_hi_u32 s45, s42, s14")
        R("s_mul_i32 s44, s42, s14")
        R("s_add_u32 s44, s44, s3")
        R("s_addc_u32 s45, s45, 0")
        R("s_lshl_b64 s[44:45], s[44:45], 15")
        R("s_add_u32 s87, %s, 11" % S_F)
        R("s_mov_b32 s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], s87")
        R("s_cmp_eq_u32 %s, 0" % S_F)
        R("s_cselect_b64 s[42:43], s[44:45], s[42:43]")
        R("s_add_u32 s%d, s%d, s42" % (dst, base))
        R("s_addc_u32 s%d, s%d, s43" % (dst + 1, base + 1))

    x_row(16, 60, 77, 0)                                 # x0 -> S_AROW
    x_row(18, 62, 78, 1)                                 # x1 -> S_BROW
    word_row(20, 56, None)                               # out0 -> S_CROW (results are dense)
    word_row(96, 66, 80)                                 # k0
    if kind == "enc2":
        x_row(54, 64, 79, 2)                             # x2
        word_row(98, 68, 81)                             # k1
        word_row(100, 58, None)                          # out1
    # tw = psi + (cm << (logn + 4)); pass constants K (prologue())
    R("s_add_u32 s43, s88, 16")
    R("s_lshl_b32 s42, s3, s43")
    R("s_add_u32 s22, s10, s42")
    R("s_addc_u32 s23, s11, 0")
    R("s_lshl_b32 s90, 1, s88")
    R("s_add_u32 s90, s90, s89")                         # Kf = 2^r + blk
    R("s_lshl_b32 s91, s90, 4")
    R("s_lshl_b32 s92, s90, 8")
    R("s_lshl_b32 s93, 0x200, s88")
    R("s_lshl_b32 s42, s89, 8")
    R("s_sub_u32 s93, s93, s42")                         # (512<<r) - 256*blk
    R("s_lshl_b32 s94, 32, s88")
    R("s_lshl_b32 s42, s89, 4")
    R("s_sub_u32 s94, s94, s42")                         # (32<<r) - 16*blk
    R("s_lshl_b32 s95, 2, s88")
    R("s_sub_u32 s95, s95, s89")                         # (2<<r) - blk
    R("s_mul_i32 s42, s3, 0x70")
    R("s_add_u32 s42, s12, s42")
    R("s_addc_u32 s43, s13, 0")
    R("s_load_dwordx16 s[56:71], s[42:43], 0x0")          # p p2 mu ninv ninv_sh w1ninv w1ninv_sh beta   (the kernarg copies are spent)
    R("s_load_dwordx8 s[72:79], s[42:43], 0x40")          # beta_sh yinv yinv_sh mask
    R("s_load_dwordx4 s[80:83], s[42:43], 0x60")          # delta mu2
    seq = 0
    if fwd:
        seq = fused_x_loads(em, vm, V_A, S_AROW, 0, "x0")
        seq = fused_x_loads(em, vm, V_B, S_BROW, 1, "x1")
        first = "F1"
    else:
        # x0 and the key row first (the product needs them), x1 behind them; then the part of I1's first sub-stage that
        # fits beside the key row (records g = 1..7 in slots 8..14; the key row occupies slots 0..7 until it is consumed)
        fused_lane_loads(em, vm, V_A, S_AROW)
        seq = (fused_lane_loads(em, vm, V_TW, S_K0ROW, stream=False)[-1], fused_lane_loads(em, vm, V_B, S_BROW))
        first = None
    tw_seq = {}
    if first:
        for s in (0, 1, 2, 3):
            tw_seq[(first, s)] = PASS_TW[first](em, vm, s)
    else:
        tw_seq[("I1", 3, "late")] = tw_lane_stage(em, vm, 3, V_TID, S_K["I1"], True, groups=range(1, 8))
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[24:25], s[56:57]")                    # p
    R("s_mov_b64 s[26:27], s[58:59]")                    # 2p
    R("s_add_u32 s28, s58, s56")                         # 3p
    R("s_addc_u32 s29, s59, s57")
    R("s_mov_b32 s30, s80")                              # delta
    R("s_mov_b32 s31, 0x3fffffff")
    R("s_mov_b32 s15, 0xc0000000")
    R("s_mov_b64 s[32:33], s[82:83]")                    # mu2
    R("s_mov_b64 s[34:35], s[62:63]")                    # ninv
    R("s_mov_b64 s[36:37], s[64:65]")                    # ninv_sh
    R("s_mov_b64 s[38:39], s[66:67]")                    # w1ninv
    R("s_mov_b64 s[40:41], s[68:69]")                    # w1ninv_sh
    em.valu("v_mov_b32_e32 v%d, s25" % (V_PHI,))
    return tw_seq, seq


def fused_x_loads(em, vm, dst, srow, k, tag):
    """x[t + 256 j] -> register pair j (the layout F1 starts from), whatever the operand's format: 16 vector loads on every
    path, so the static load count of the VmCounter holds"""
    R = em.raw
    A = T(1, 0)
    R("s_bfe_u32 %s, %s, 0x%x" % (S_F, S_FMT, (4 << 16) | (4 * k)))
    R("s_mov_b64 s[86:87], %s" % (srow,))
    R("s_cmp_eq_u32 %s, 0" % S_F)
    R("s_cbranch_scc0 .L%s_compact" % tag)
    seq = 0
    for j in range(16):
        seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d nt" % (vp(dst + 2 * j), V_OFF8, (j & 1) * 2048))
        if j & 1:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_compact:" % tag)
    R("s_cmp_eq_u32 %s, 1" % S_F)
    R("s_cbranch_scc0 .L%s_i16" % tag)
    for j in range(16):
        R("global_load_sbyte v%d, v%d, s[86:87] offset:%d" % (dst + 2 * j, V_TID, 256 * j))
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_i16:" % tag)
    R("s_cmp_eq_u32 %s, 2" % S_F)
    R("s_cbranch_scc0 .L%s_i32" % tag)
    em.valu("v_lshlrev_b32_e32 v%d, 1, v%d" % (A, V_TID))
    for j in range(16):
        R("global_load_sshort v%d, v%d, s[86:87] offset:%d" % (dst + 2 * j, A, 512 * (j & 7)))
        if j == 7:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_i32:" % tag)
    em.valu("v_lshlrev_b32_e32 v%d, 2, v%d" % (A, V_TID))
    for j in range(16):
        R("global_load_dword v%d, v%d, s[86:87] offset:%d" % (dst + 2 * j, A, 1024 * (j & 3)))
        if j & 3 == 3:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    em.lines.append(".L%s_issued:" % tag)
    return seq


def fused_x_expand(em, dst, k, tag):
    """compact formats: the sign-extended integer x becomes x (x >= 0) or p + x (x < 0) -- any 64-bit word congruent to the
    coefficient is a legal input of the first butterfly"""
    R = em.raw
    R("s_bfe_u32 %s, %s, 0x%x" % (S_F, S_FMT, (4 << 16) | (4 * k)))
    R("s_cmp_eq_u32 %s, 0" % S_F)
    R("s_cbranch_scc1 .L%s_words" % tag)
    t = T(0, 4)
    for j in range(16):
        x = dst + 2 * j
        em.valu("v_ashrrev_i32_e32 v%d, 31, v%d" % (x + 1, x))
        em.valu("v_and_b32_e32 v%d, s24, v%d" % (t, x + 1))
        em.valu("v_and_b32_e32 v%d, s25, v%d" % (t + 1, x + 1))
        em.valu("v_lshl_add_u64 %s, %s, 0, %s" % (vp(x), vp(x), vp(t)))
    em.lines.append(".L%s_words:" % tag)


def fused_lane_loads(em, vm, dst, srow, stream=True):
    """element 1024w + 64j + l of the row -> register pair j (512 B per wave instruction): any layout serves a point-wise
    step as long as all operands share it.  stream: a row nobody reads again (`nt`); the key row stays in the caches"""
    g, _ = lane_contig_setup(em)
    em.raw("s_mov_b64 s[86:87], %s" % (srow,))
    seqs = []
    for j in range(16):
        seqs.append(vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst + 2 * j), g, (j & 7) * 512, " nt" if stream else "")))
        if j == 7:
            em.raw("s_add_u32 s86, s86, 0x1000")
            em.raw("s_addc_u32 s87, s87, 0")
    return seqs


def fma_job(k, a, b, fold_a):
    """k = canonical(k * a + b): k a canonical key word, a / b lazily reduced words (a is folded in place the first time)"""
    def gen(s):
        yield from pointwise(k, a, False, fold_a)(s)
        yield from fold2(s, b, b)
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(k), vp(k), vp(b)), None, None
        yield from fold2(s, k, k)
        yield from csub_p(s, k)
    return gen


def fms_job(a, k, b, subtract):
    """a = fold(b -+ a * k) < p + 4 delta, all inputs canonical (the contract of the reference's operators, ops.hpp:131,211)"""
    def gen(s):
        yield from pointwise(a, k, False, False)(s)
        if subtract:
            E = T(s, 12)
            yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(b), S_P2), None, None
            yield "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (a, S_BORROW[s], E, a), S_BORROW[s], None
            yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (a + 1, S_DUMMY, E + 1, a + 1, S_BORROW[s]), None, S_BORROW[s]
        else:
            yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(a), vp(a), vp(b)), None, None
        yield from fold2(s, a, a)
    return gen


def build_fused(kind):
    """kind: enc2 | fma_fwd | fms_inv | fma_inv"""
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    tw_seq, seq_x = prologue_fused(em, vm, kind)
    V_K = V_TW                      # key words: twiddle slots 0..7

    def forward(bases, k_row, first_pass_ready):
        """F1 E1 F2 E2 F3 over `bases` (shared twiddle records); the key row's loads are woven into F3: seven of its eight
        16-byte loads as soon as F3's sub-stage 2 is done with slots 0..6, the last one behind sub-stage 3.  Returns the
        sequence numbers of the key loads"""
        for name, nxt in (("F1", "F2"), ("F2", "F3"), ("F3", None)):
            em.comment("%s%s" % (name, "; prefetching " + nxt if nxt else "; then the key row"))
            kseq = []
            for s in range(4):
                vm.wait(tw_seq[(name, s)])
                ct_stage(em, bases, s)
                if nxt is not None:
                    tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)
                elif s >= 2:
                    em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_TID))     # (butterfly scratch: recomputed per batch of loads)
                    for i in (range(7) if s == 2 else (7,)):
                        kseq.append(vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d"
                                            % (V_K + 4 * i, V_K + 4 * i + 3, T(1, 0), k_row, 16 * i)))
            if name == "F1":
                for i, base in enumerate(bases):
                    em.comment("E1")
                    if i or not first_pass_ready:
                        R("s_barrier")       # WAR: the slab is still being read (previous operand / the first half's store transposes)
                    lds_write(em, V_L1W, base, 2176)
                    R("s_waitcnt lgkmcnt(0)")
                    R("s_barrier")
                    lds_read(em, V_L1R, base, 136)
                    R("s_waitcnt lgkmcnt(0)")
            elif name == "F2":
                em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
                for base in bases:
                    lds_write(em, V_L1R, base, 136)
                    lds_read(em, V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
        return kseq

    def fma_store(xb, fold_a, kseq, dst_row, early=None):
        """V_K = canonical(V_K * V_A + xb) -> dst_row (NTT form: thread q holds words 16q..16q+15; a wave-local LDS transpose
        makes the stores 512 B per wave instruction)"""
        em.comment("point-wise multiply-add against the key row")
        for i in range(8):
            vm.wait(kseq[i])
            run_pairs(em, [fma_job(V_K + 4 * i, V_A + 4 * i, xb + 4 * i, fold_a), fma_job(V_K + 4 * i + 2, V_A + 4 * i + 2, xb + 4 * i + 2, fold_a)])
        if early is not None:
            early()
        lds_write(em, V_L2R, V_K, 8)
        g, l = lane_contig_setup(em)
        for j in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_K + 2 * j), l, 544 * j))
        R("s_waitcnt lgkmcnt(0)")
        R("s_mov_b64 s[86:87], %s" % (dst_row,))
        for j in range(16):
            vm.load("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g, vp(V_K + 2 * j), (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")

    if kind in ("enc2", "fma_fwd"):
        vm.wait(seq_x)
        fused_x_expand(em, V_A, 0, "e0")
        fused_x_expand(em, V_B, 1, "e1")
        kseq = forward([V_A, V_B], S_K0ROW, True)
        if kind == "fma_fwd":
            fma_store(V_B, True, kseq, S_CROW)
            R("s_endpgm")
            return em
        state = {}

        def early():   # x2 is requested as soon as V_B is free: its latency hides behind the store of out0
            state["x2"] = fused_x_loads(em, vm, V_B, S_X2ROW, 2, "x2")
        fma_store(V_B, True, kseq, S_CROW, early)
        em.comment("second half: x2 alone, x0' stays in V_A")
        for s in (0, 1, 2, 3):
            tw_seq[("F1", s)] = PASS_TW["F1"](em, vm, s)
        vm.wait(state["x2"])
        fused_x_expand(em, V_B, 2, "e2")
        kseq = forward([V_B], S_K1ROW, False)
        fma_store(V_B, False, kseq, S_O1ROW)
        R("s_endpgm")
        return em

    # ---- fms_inv / fma_inv: point-wise step in the loaded (lane-contiguous) layout, then the inverse passes of build_body
    seq_k, seq_b = seq_x
    vm.wait(seq_k)
    em.comment("x1 -+ x0 * k0 (x1 is consumed word by word as it lands)")
    for i in range(0, 16, 2):
        vm.wait(seq_b[i + 1])
        run_pairs(em, [fms_job(V_A + 2 * j, V_K + 2 * j, V_B + 2 * j, kind == "fms_inv") for j in (i, i + 1)])
    tw_seq[("I1", 3)] = tw_lane_stage(em, vm, 3, V_TID, S_K["I1"], True, groups=(0,))   # the record the key row was in the way of
    for s in (2, 1, 0):
        tw_seq[("I1", s)] = PASS_TW["I1"](em, vm, s)
    em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
    _, l = lane_contig_setup(em)
    for j in range(16):
        R("ds_write_b64 v%d, %s offset:%d" % (l, vp(V_A + 2 * j), 544 * j))
    lds_read(em, V_L2R, V_A, 8)
    R("s_waitcnt lgkmcnt(0)")

    def first_stage():
        """I1's sub-stage 3 with the late record's butterfly last: groups 1..7 run on the records fetched beside the key row"""
        jobs = [gs_bfly(V_A + 4 * g, V_A + 4 * g + 2, twreg(tw_slot(3, g))) for g in (1, 2, 3, 4, 5, 6, 7, 0)]
        vm.wait(tw_seq[("I1", 3, "late")])
        run_pairs(em, jobs[:6])
        vm.wait(tw_seq[("I1", 3)])
        run_pairs(em, jobs[6:])
    inverse_half(em, vm, tw_seq, first_stage)
    return em


def inverse_half(em, vm, tw_seq, first_stage=None):
    """I1 E2' I2 E1' I3 and the merged last stage over V_A (thread-contiguous words in), store to S_CROW"""
    R = em.raw
    order = ["I1", "I2", "I3"]
    for name in order:
        nxt = order[order.index(name) + 1] if name != "I3" else None
        em.comment("%s%s" % (name, "; prefetching " + nxt if nxt else ""))
        for s in ((3, 2, 1, 0) if name != "I3" else (3, 2, 1)):
            if name == "I1" and s == 3 and first_stage is not None:
                first_stage()
            else:
                vm.wait(tw_seq[(name, s)])
                gs_stage(em, V_A, s)
            if nxt is not None:
                tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)
        if name == "I1":
            em.comment("E2'")
            lds_write(em, V_L2R, V_A, 8)
            lds_read(em, V_L1R, V_A, 136)
            R("s_waitcnt lgkmcnt(0)")
        elif name == "I2":
            em.comment("E1'")
            lds_write(em, V_L1R, V_A, 136)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, V_L1W, V_A, 2176)
            R("s_waitcnt lgkmcnt(0)")

    def last_plain():
        vm.wait(tw_seq[("I3", 0)])
        gs_stage(em, V_A, 0)
    epilogue_inverse(em, vm, last_plain)


# ------------------------------------------------------------------ transform-fused pipelines, rows of 8192 / 16384 words
# The same four pipelines on the row-resident register map of build_row16k (ring mode: 128 VGPRs, one butterfly at a time,
# twiddle records streaming through the 9-slot ring; ROW_G sub-groups of 256 threads, one outer radix-ROW_G pass F0 / I0
# around the 4096-word passes).  One workgroup per (batch element, modulus) row of exactly 4096 ROW_G words.  The ring is
# empty between a transform and the next one, so the 32 registers of a key row's 16 words live in ITS slots: the key is
# loaded behind the last forward record, the multiply-add lands in the key's registers (x' stays for the second result), and
# the next transform's ring is primed once the result's stores have been issued.  The exchanges are the plain ones of
# build_row16k (write, barrier, read): the split-phase schedules of the product kernels are tied to their two-operand shape.
# kernarg as ARGS_FUSED; both grids of prologue_fused.
S_X2ROW16 = "s[52:53]"     # (stream 1's borrow pair: idle in single-stream mode)


def build_fused_rows(kind):
    """kind: enc2 | fma_fwd | fms_inv | fma_inv -- over one 4096 * ROW_G-word row per workgroup (configure("ring", ROW_G));
    polymul (experiment, ROW_G = 1): out0 = INTT(NTT(x0) (.) NTT(x1)), the metric product on the ring-mode map"""
    assert SINGLE_STREAM and ROW_G in (1, 2, 4)      # (1: a 4096-word row on the ring-mode map -- 128 VGPRs, four workgroups per CU)
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    fwd = kind in ("enc2", "fma_fwd", "polymul")
    passes = {"F0": (S_K0["F0"], None, False), "F1": (S_K["F1"], None, False), "F2": (S_K["F2"], V_BIDX, False),
              "F3": (S_K["F3"], V_TID, False), "I1": (S_K["I1"], V_TID, True), "I2": (S_K["I2"], V_BIDX, True),
              "I3": (S_K["I3"], None, True), "I0": (S_K0["I0"], None, True)}
    order = {"F0": tuple(range(ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0) if ROW_G > 1 else (3, 2, 1), "I0": tuple(range(ROW_LG - 1, 0, -1))}
    per = 16 // ROW_G
    AX = T(0, 0)
    V_K = V_TW
    row_bytes_log = 15 + ROW_LG

    # ---------------- prologue: thread map of prologue16k, operands of prologue_fused
    R("s_load_dwordx16 s[56:71], s[0:1], 0x0")           # out0 out1 x0 x1 x2 k0 k1 psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x40")           # mc
    R("s_load_dwordx2 s[72:73], s[0:1], 0x48")           # nm, logn
    R("s_load_dwordx8 s[76:83], s[0:1], 0x50")           # fmt, strides x0 x1 x2 k0 k1, count, magic
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_OFF8, V_TID))                      # tid*8
    em.valu("v_mov_b32_e32 v%d, v%d" % (V_TWA, V_TID))                              # the workgroup-wide thread index (compact inputs)
    em.valu("v_lshrrev_b32_e32 v%d, 8, v%d" % (V_BIDX, V_TID))                      # q (wave-uniform)
    R("s_nop 1")
    R("v_readfirstlane_b32 %s, v%d" % (S_Q, V_BIDX))
    R("s_nop 1")
    em.valu("v_and_b32_e32 v%d, 0xff, v%d" % (V_TID, V_TID))                        # t = tid & 255
    R("s_mul_i32 %s, %s, 0x%x" % (S_SLAB, S_Q, SLAB_BYTES))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (V_BIDX, V_TID))                      # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (V_L1W, V_TID, V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1W, V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (V_L1R, V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_L1R, V_BIDX, V_L2R, V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_L1R, V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_L2R, V_TID, V_L2R))              # 17*t*8
    for reg in (V_L1W, V_L1R, V_L2R):
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (reg, S_SLAB, reg))                  # inside the sub-group's slab
    for t_ in sorted(set(V_T)):
        em.valu("v_mov_b32_e32 v%d, 0" % (t_ + 15,))                                # the persistent zero of ZP
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b32 s14, s72")                              # nm
    # magic != 0: the 1-D grid of prologue_fused (the nm rows of a batch element back to back on one XCD)
    R("s_cmp_eq_u32 s83, 0")
    R("s_cbranch_scc1 .Lplain_grid")
    R("s_and_b32 s42, s2, 7")                            # xcd
    R("s_lshr_b32 s43, s2, 3")                           # q
    R("s_mul_hi_u32 s44, s43, s83")                      # j
    R("s_mul_i32 s45, s44, s14")
    R("s_sub_u32 s3, s43, s45")                          # cm
    R("s_lshl_b32 s44, s44, 3")
    R("s_add_u32 s2, s44, s42")                          # element
    R("s_cmp_lt_u32 s2, s82")
    R("s_cbranch_scc1 .Lplain_grid")
    R("s_endpgm")                                        # padding of the last group of eight
    em.lines.append(".Lplain_grid:")
    R("s_sub_u32 s88, s73, 12")                          # r = ROW_LG: rows of exactly 4096 ROW_G words
    R("s_mov_b32 %s, s76" % S_FMT)
    R("s_mov_b64 s[10:11], s[70:71]")                    # psi (the lane-major copy)

    def word_row(dst, base, stride):
        R("s_mul_i32 s42, s2, s%d" % stride if stride is not None else "s_mov_b32 s42, s2")
        R("s_mul_hi_u32 s43, s42, s14")
        R("s_mul_i32 s42, s42, s14")
        R("s_add_u32 s42, s42, s3")
        R("s_addc_u32 s43, s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], %d" % row_bytes_log)
        R("s_add_u32 s%d, s%d, s42" % (dst, base))
        R("s_addc_u32 s%d, s%d, s43" % (dst + 1, base + 1))

    def x_row(dst, base, stride, k):
        if not fwd:
            return word_row(dst, base, stride)
        R("s_bfe_u32 %s, %s, 0x%x" % (S_F, S_FMT, (4 << 16) | (4 * k)))
        R("s_mul_i32 s42, s2, s%d" % stride)
        R("s_mul_hi_u32 s45, s42, s14")
        R("s_mul_i32 s44, s42, s14")
        R("s_add_u32 s44, s44, s3")
        R("s_addc_u32 s45, s45, 0")
        R("s_lshl_b64 s[44:45], s[44:45], %d" % row_bytes_log)
        R("s_add_u32 s87, %s, %d" % (S_F, 11 + ROW_LG))      # compact: (x * stride) << (log2 n + f - 1)
        R("s_mov_b32 s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], s87")
        R("s_cmp_eq_u32 %s, 0" % S_F)
        R("s_cselect_b64 s[42:43], s[44:45], s[42:43]")
        R("s_add_u32 s%d, s%d, s42" % (dst, base))
        R("s_addc_u32 s%d, s%d, s43" % (dst + 1, base + 1))

    x_row(16, 60, 77, 0)
    x_row(18, 62, 78, 1)
    word_row(20, 56, None)
    word_row(96, 66, 80)                                 # k0
    if kind == "enc2":
        x_row(52, 64, 79, 2)                             # x2
        word_row(98, 68, 81)                             # k1
        word_row(100, 58, None)                          # out1
    # tw = psi + (cm << (logn + 4)); pass constants of the row's only block group (blkG = 0) and of block q
    R("s_add_u32 s43, s88, 16")
    R("s_lshl_b32 s42, s3, s43")
    R("s_add_u32 s22, s10, s42")
    R("s_addc_u32 s23, s11, 0")
    R("s_mov_b32 %s, 1" % (S_K0["F0"],))
    R("s_mov_b32 %s, 2" % (S_K0["I0"],))
    R("s_mov_b32 s89, %s" % (S_Q,))                      # blk = q
    R("s_lshl_b32 s90, 1, s88")
    R("s_add_u32 s90, s90, s89")                         # Kf = 2^r + blk
    R("s_lshl_b32 s91, s90, 4")
    R("s_lshl_b32 s92, s90, 8")
    R("s_lshl_b32 s93, 0x200, s88")
    R("s_lshl_b32 s42, s89, 8")
    R("s_sub_u32 s93, s93, s42")                         # (512<<r) - 256*blk
    R("s_lshl_b32 s94, 32, s88")
    R("s_lshl_b32 s42, s89, 4")
    R("s_sub_u32 s94, s94, s42")                         # (32<<r) - 16*blk
    R("s_lshl_b32 s95, 2, s88")
    R("s_sub_u32 s95, s95, s89")                         # (2<<r) - blk
    R("s_mul_i32 s42, s3, 0x70")
    R("s_add_u32 s42, s12, s42")
    R("s_addc_u32 s43, s13, 0")
    R("s_load_dwordx16 s[56:71], s[42:43], 0x0")          # the ModConst record (the kernarg copies are spent)
    R("s_load_dwordx8 s[72:79], s[42:43], 0x40")
    R("s_load_dwordx4 s[80:83], s[42:43], 0x60")

    def x_loads(dst, srow, k, tag):
        """x[tid + 256 G j] -> register pair j (the layout F0 starts from): word rows or compact; 16 loads on every path"""
        step = 256 * ROW_G
        R("s_bfe_u32 %s, %s, 0x%x" % (S_F, S_FMT, (4 << 16) | (4 * k)))
        R("s_mov_b64 s[86:87], %s" % (srow,))
        R("s_cmp_eq_u32 %s, 0" % S_F)
        R("s_cbranch_scc0 .L%s_compact" % tag)
        seq = 0
        for j in range(16):
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] nt" % (vp(dst + 2 * j), V_OFF8))
            if j < 15:
                R("s_add_u32 s86, s86, 0x%x" % (8 * step,))
                R("s_addc_u32 s87, s87, 0")
        R("s_branch .L%s_issued" % tag)
        em.lines.append(".L%s_compact:" % tag)
        for f, (es, op) in enumerate(((1, "global_load_sbyte"), (2, "global_load_sshort"), (4, "global_load_dword")), 1):
            if f < 3:
                R("s_cmp_eq_u32 %s, %d" % (S_F, f))
                R("s_cbranch_scc0 .L%s_f%d" % (tag, f + 1))
            if es > 1:
                em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (AX, es.bit_length() - 1, V_TWA))
            for j in range(16):
                R("%s v%d, v%d, s[86:87]" % (op, dst + 2 * j, V_TWA if es == 1 else AX))
                if j < 15:
                    R("s_add_u32 s86, s86, 0x%x" % (es * step,))
                    R("s_addc_u32 s87, s87, 0")
            if f < 3:
                R("s_branch .L%s_issued" % tag)
                em.lines.append(".L%s_f%d:" % (tag, f + 1))
        em.lines.append(".L%s_issued:" % tag)
        return seq

    def block_base(srow):                                 # s[86:87] = first word of this sub-group's 4096-word block
        lo, hi = srow[2:-1].split(":")
        R("s_lshl_b32 s42, %s, 15" % (S_Q,))
        R("s_add_u32 s86, s%s, s42" % lo)
        R("s_addc_u32 s87, s%s, 0" % hi)

    def lane_loads(dst, srow, stream=True):
        block_base(srow)
        g, _ = lane_contig_setup(em)
        seqs = []
        for j in range(16):
            seqs.append(vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst + 2 * j), g, (j & 7) * 512, " nt" if stream else "")))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
        return seqs

    if fwd:
        x_loads(V_A, S_AROW, 0, "x0")
        seq_x = x_loads(V_B, S_BROW, 1, "x1")
    else:
        lane_loads(V_A, S_AROW)
        seq_k = lane_loads(V_K, S_K0ROW, stream=False)[-1]
        seq_b = lane_loads(V_B, S_BROW)
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[24:25], s[56:57]")                    # p
    R("s_mov_b64 s[26:27], s[58:59]")                    # 2p
    R("s_add_u32 s28, s58, s56")                         # 3p
    R("s_addc_u32 s29, s59, s57")
    R("s_mov_b32 s30, s80")                              # delta
    R("s_mov_b32 s31, 0x3fffffff")
    R("s_mov_b32 s15, 0xc0000000")
    R("s_mov_b64 s[32:33], s[82:83]")                    # mu2
    R("s_mov_b64 s[34:35], s[62:63]")                    # ninv
    R("s_mov_b64 s[36:37], s[64:65]")                    # ninv_sh
    R("s_mov_b64 s[38:39], s[66:67]")                    # w1ninv
    R("s_mov_b64 s[40:41], s[68:69]")                    # w1ninv_sh
    em.valu("v_mov_b32_e32 v%d, s25" % (V_PHI,))

    def make_ring(names):
        uses = [(name, s_, g) for name in names for s_ in order[name] for g in range(1 << s_)]
        ring = Ring(em, vm, RING_SLOTS, uses, passes)
        ring.prime()
        return ring

    def forward(bases, first):
        """F0 X0 F1 E1 F2 E2 F3 over `bases` on shared twiddle records (the plain exchanges of build_row16k)"""
        ring = make_ring(["F0", "F1", "F2", "F3"])

        def fwd_pass(name):
            em.comment("%s" % name)
            for s_ in order[name]:
                half = 8 >> s_
                for g in range(1 << s_):
                    tw = ring.get((name, s_, g))
                    jobs = []
                    for h in range(half):
                        i0 = g * 2 * half + h
                        for base in bases:
                            jobs.append(ct_bfly(base + 2 * i0, base + 2 * (i0 + half), tw))
                    run_pairs(em, jobs)
                    ring.done((name, s_, g))
        fwd_pass("F0")
        for i, base in enumerate(bases if ROW_G > 1 else ()):
            em.comment("X0: thread (q, t) slot per*qq + j  ->  sub-group qq, thread t, slot q + G*j")
            if i or not first:
                R("s_barrier")       # WAR: the slabs are still being read (previous operand / the first result's store transposes)
            em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * SLAB_BYTES, V_OFF8))
            for k in range(16):
                qq, j = k // per, k % per
                R("ds_write_b64 v%d, %s offset:%d" % (V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * SLAB_BYTES + j * 2048 * ROW_G))
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, S_SLAB, AX))
            for k in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F1")
        for base in bases:
            em.comment("E1")
            R("s_barrier")           # WAR against the previous exchange through this slab
            lds_write(em, V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F2")
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        for base in bases:
            lds_write(em, V_L1R, base, 136)
            lds_read(em, V_L2R, base, 8)
        R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F3")

    def x_expand(dst, k, tag):
        fused_x_expand(em, dst, k, tag)

    def fma_store(xb, fold_a, k_row, dst_row, early=None):
        """V_K = canonical(key * V_A + xb) -> dst_row; the key's 16 words go into the (empty) ring's registers"""
        em.comment("the key row's block: words 16t .. 16t+15 of block q")
        block_base(k_row)
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (AX, V_TID))
        kseq = [vm.load("global_load_dwordx4 v[%d:%d], v%d, s[86:87] offset:%d" % (V_K + 4 * i, V_K + 4 * i + 3, AX, 16 * i)) for i in range(8)]
        for i in range(8):
            vm.wait(kseq[i])
            run_pairs(em, [fma_job(V_K + 4 * i, V_A + 4 * i, xb + 4 * i, fold_a), fma_job(V_K + 4 * i + 2, V_A + 4 * i + 2, xb + 4 * i + 2, fold_a)])
        if early is not None:
            early()
        lds_write(em, V_L2R, V_K, 8)
        g, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, S_SLAB, l))
        for j in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_K + 2 * j), l, 544 * j))
        R("s_waitcnt lgkmcnt(0)")
        block_base(dst_row)
        for j in range(16):
            vm.load("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g, vp(V_K + 2 * j), (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")

    if fwd:
        vm.wait(seq_x)
        x_expand(V_A, 0, "e0")
        x_expand(V_B, 1, "e1")
        forward([V_A, V_B], True)
    if kind == "polymul":
        ring = make_ring(["I1", "I2", "I3", "I0"])       # (the inverse passes' first records fly under the product)
        em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
        run_pairs(em, [pointwise(V_A + 2 * i, V_B + 2 * i, True, True) for i in range(16)])
    elif fwd:
        if kind == "fma_fwd":
            fma_store(V_B, True, S_K0ROW, S_CROW)
            R("s_endpgm")
            return em
        state = {}

        def early():
            state["x2"] = x_loads(V_B, S_X2ROW16, 2, "x2")
        fma_store(V_B, True, S_K0ROW, S_CROW, early)
        em.comment("second half: x2 alone, x0' stays in V_A")
        vm.wait(state["x2"])
        x_expand(V_B, 2, "e2")
        forward([V_B], False)
        fma_store(V_B, False, S_K1ROW, S_O1ROW)
        R("s_endpgm")
        return em

    # ---- fms_inv / fma_inv (and the second half of the product)
    if kind != "polymul":
        vm.wait(seq_k)
        em.comment("x1 -+ x0 * k0 in the loaded (lane-contiguous) layout; x1 is consumed as it lands")
        for i in range(0, 16, 2):
            vm.wait(seq_b[i + 1])
            run_pairs(em, [fms_job(V_A + 2 * j, V_K + 2 * j, V_B + 2 * j, kind == "fms_inv") for j in (i, i + 1)])
        ring = make_ring(["I1", "I2", "I3", "I0"])
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, S_SLAB, l))
        for j in range(16):
            R("ds_write_b64 v%d, %s offset:%d" % (l, vp(V_A + 2 * j), 544 * j))
        lds_read(em, V_L2R, V_A, 8)
        R("s_waitcnt lgkmcnt(0)")

    def inv_pass(name, stages):
        em.comment(name)
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((name, s_, g))
                run_pairs(em, [gs_bfly(V_A + 2 * (g * 2 * half + h), V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((name, s_, g))
    inv_pass("I1", (3, 2, 1, 0))
    em.comment("E2'")
    lds_write(em, V_L2R, V_A, 8)
    lds_read(em, V_L1R, V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2", (3, 2, 1, 0))
    em.comment("E1'")
    lds_write(em, V_L1R, V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    lds_read(em, V_L1W, V_A, 2176)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3", (3, 2, 1, 0) if ROW_G > 1 else (3, 2, 1))
    if ROW_G > 1:
        em.comment("X0': thread (q, t) slot g + G*j  ->  thread (g, t) slot per*q + j, reader-major layout [slot][tid]")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                              # q*32768 + t*8
        for k in range(16):
            g_, j = k % ROW_G, k // ROW_G
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(V_A + 2 * k), j * 2048 * ROW_G + g_ * 2048))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        rstep = 2048 * ROW_G
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(V_A + 2 * k), V_OFF8 if k < 8 else AX, (k & 7) * rstep))
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I0", order["I0"])
    em.comment("stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(V_A + 2 * h, V_A + 2 * (h + 8)) for h in range(8)])
    R("s_mov_b64 s[86:87], %s" % (S_CROW,))
    for k in range(16):
        R("global_store_dwordx2 v%d, %s, s[86:87] nt" % (V_OFF8, vp(V_A + 2 * k)))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * ROW_G,))
            R("s_addc_u32 s87, s87, 0")
    R("s_endpgm")
    return em


KERNELS_FUSED = {
    "enc2": ("fused_enc2_4096", "nflhip_fused_enc2_4096_asm"),
    "fma_fwd": ("fused_fma_fwd4096", "nflhip_fused_fma_fwd4096_asm"),
    "fms_inv": ("fused_fms_inv4096", "nflhip_fused_fms_inv4096_asm"),
    "fma_inv": ("fused_fma_inv4096", "nflhip_fused_fma_inv4096_asm"),
}


HEADER = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.globl\t%(k)s
\t.p2align\t8
\t.type\t%(k)s,@function
%(k)s:
"""

FOOTER = """.Lfunc_end0:
\t.size\t%(k)s, .Lfunc_end0-%(k)s
\t.rodata
\t.p2align\t6
\t.amdhsa_kernel %(k)s
\t\t.amdhsa_group_segment_fixed_size %(lds)d
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size %(karg)d
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 1
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr %(vgpr)d
\t\t.amdhsa_next_free_sgpr %(sgpr)d
\t\t.amdhsa_accum_offset %(accum)d
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .args:
%(args)s    .group_segment_fixed_size: %(lds)d
    .kernarg_segment_align: 8
    .kernarg_segment_size: %(karg)d
    .max_flat_workgroup_size: %(wg)d
    .name:           %(k)s
    .private_segment_fixed_size: 0
    .sgpr_count:     %(sgprc)d
    .symbol:         %(k)s.kd
    .vgpr_count:     %(vgpr)d
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata
"""


KERNELS = {   # kind -> (file suffix, kernel symbol)
    "polymul": ("polymul4096", "nflhip_polymul4096_asm"),
    "polymul_ntt": ("polymul_ntt4096", "nflhip_polymul_ntt4096_asm"),
    "fwd": ("ntt_fwd4096", "nflhip_ntt_fwd4096_asm"),
    "inv": ("ntt_inv4096", "nflhip_ntt_inv4096_asm"),
    "inv_mul": ("ntt_inv_mul4096", "nflhip_ntt_inv_mul4096_asm"),
    "fwd2": ("ntt_fwd4096x2", "nflhip_ntt_fwd4096x2_asm"),
    "inv2": ("ntt_inv4096x2", "nflhip_ntt_inv4096x2_asm"),
}


ARGS_STD = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44)]
ARGS_PIPE = ARGS_STD + [("i32", 48), ("i32", 52), ("i32", 56), ("i32", 60), ("ptr", 64), ("ptr", 72), ("ptr", 80), ("ptr", 88),
                        ("ptr", 96), ("i32", 104), ("i32", 108)]


def args_yaml(spec):
    out = []
    for kind, off in spec:
        if kind == "ptr":
            out.append("      - {.address_space: global, .offset: %d, .size: 8, .value_kind: global_buffer}" % off)
        else:
            out.append("      - {.offset: %d, .size: 4, .value_kind: by_value}" % off)
    return "\n".join(out) + "\n"


def emit_file(path, kname, em, args=None, lds=None):
    accum = (NEXT_VGPR + 3) // 4 * 4
    args = ARGS_STD if args is None else args
    karg = args[-1][1] + (8 if args[-1][0] == "ptr" else 4)
    params = dict(k=kname, lds=LDS_BYTES if lds is None else lds, vgpr=NEXT_VGPR, sgpr=NEXT_SGPR, accum=accum, sgprc=NEXT_SGPR + 6, wg=WG_SIZE,
                  karg=karg, args=args_yaml(args))
    with open(path, "w") as f:
        f.write("; GENERATED by tools/gen_polymul_asm.py -- do not edit.\n")
        f.write(HEADER % params)
        f.write("\n".join(em.lines) + "\n")
        f.write(FOOTER % params)
    print("wrote %s: %d VALU instructions (static), %d hazard nops, %d lines" % (path, em.n_valu, em.n_nop, len(em.lines)))


KERNELS16K = {
    "polymul": ("polymul16384", "nflhip_polymul16384_asm"),
    "polymul_ntt": ("polymul_ntt16384", "nflhip_polymul_ntt16384_asm"),
    "fwd": ("ntt_fwd16384", "nflhip_ntt_fwd16384_asm"),
    "inv": ("ntt_inv16384", "nflhip_ntt_inv16384_asm"),
}


def main():
    outdir = os.path.dirname(OUT)
    experiments = bool(os.environ.get("NFL_GEN_EXPERIMENTS"))   # also emit the variants that were measured and not kept
    nt = lambda em_: [l + " nt" if ("global_load_dwordx2" in l or "global_store_dwordx2" in l) else l for l in em_.lines]
    configure("pair")
    for kind, (stem, kname) in KERNELS.items():
        args = ARGS_STD + [("i32", 48)] if kind in ("fwd2", "inv2") else None
        if kind in ("polymul", "fwd2", "inv2"):
            # the n = 4096 product and the two-row transforms stream their coefficients with `nt` (+1 % on workload B)
            em_nt = build(kind)
            em_nt.lines = nt(em_nt)
            emit_file(os.path.join(outdir, stem + "nt_gfx950.s"), kname.replace("_asm", "nt_asm"), em_nt, args=args)
            if not experiments:
                continue
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, build(kind), args=args)
    # transform-fused pipelines (n = 4096): word-row streams `nt`, key rows and compact inputs through the caches
    g = globals()
    g.update(NEXT_SGPR=102)
    for kind, (stem, kname) in KERNELS_FUSED.items():
        emf = build_fused(kind)
        emf.lines = [l + " nt" if "global_store_dwordx2" in l and not l.endswith(" nt") else l for l in emf.lines]   # (the inverse kinds' result rows)
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, emf, args=ARGS_FUSED)
    g.update(NEXT_SGPR=96)
    # n = 65536: the three-role pipeline kernel; coefficient streams `nt`: 3 x 15.7 MB of data per product pass through each
    # XCD's 4 MiB L2 exactly once, the 31 MB of twiddle tables are what is worth keeping there (+3 % on workload E)
    em_nt = build_pipe()
    em_nt.lines = nt(em_nt)
    emit_file(os.path.join(outdir, "polymul_pipe65536nt_gfx950.s"), "nflhip_polymul_pipe65536nt_asm", em_nt, args=ARGS_PIPE)
    em_b = build_pipe(b_ntt=True)     # operand b already transformed: two streaming roles per row, b' read block-wise as it lies
    em_b.lines = nt(em_b)
    emit_file(os.path.join(outdir, "polymul_pipe65536ntb_gfx950.s"), "nflhip_polymul_pipe65536ntb_asm", em_b, args=ARGS_PIPE)
    if experiments:
        emit_file(os.path.join(outdir, "polymul_pipe65536_gfx950.s"), "nflhip_polymul_pipe65536_asm", build_pipe(), args=ARGS_PIPE)
        em15 = build_pipe(15)     # n = 32768 on the same kernel with radix-8 streaming roles (superseded by build_row32k)
        em15.lines = nt(em15)
        emit_file(os.path.join(outdir, "polymul_pipe32768_gfx950.s"), "nflhip_polymul_pipe32768_asm", em15, args=ARGS_PIPE)
    # one-launch variants: rows pinned to an XCD, intermediates through its L2 (fused_header); "l" = the pooled-scratch
    # experiment (measured, not kept)
    global FUSED_LOADS, FUSED_LIFO
    for lg, mod, sfx in ((16, "", ""), (15, "", "")) + (((16, "", "l"), (15, "", "l")) if experiments else ()):
        FUSED_LOADS = mod
        FUSED_LIFO = sfx == "l"
        emf = build_pipe(lg, fused=True)
        emit_file(os.path.join(outdir, "polymul_xcd%d%s_gfx950.s" % (1 << lg, sfx)), "nflhip_polymul_xcd%d%s_asm" % (1 << lg, sfx), emf,
                  args=ARGS_PIPE, lds=LDS_BYTES + 64)
    FUSED_LIFO = False
    build_pipe(16)   # (leave the module-level PIPE_LOGN as it was)
    ring = "ringpair" if os.environ.get("NFL_GEN_RINGPAIR") else "ring"
    configure(ring, 4)
    for kind, (stem, kname) in KERNELS16K.items():
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, build_row16k(kind))
    g = globals()
    g.update(NEXT_SGPR=98)     # two rows of one modulus per workgroup on shared twiddle records (stand-alone forward transform)
    emit_file(os.path.join(outdir, "ntt_fwd16384x2_gfx950.s"), "nflhip_ntt_fwd16384x2_asm", build_row16k("fwd2"), args=ARGS_STD + [("i32", 48)])
    g.update(NEXT_SGPR=96)
    if experiments:            # persistent workgroups with row prefetch: measured -6.5 % (n = 16384) / -11 % (n = 8192), not kept
        g.update(NEXT_SGPR=102)
        emit_file(os.path.join(outdir, "polymul16384p_gfx950.s"), "nflhip_polymul16384p_asm", build_row16k_loop(),
                  args=ARGS_STD + [("i32", 48), ("i32", 52)])
    configure(ring, 2)         # 8192-word rows: two sub-groups, 512 threads, one radix-2 stage around the blocks
    for kind, (stem, kname) in KERNELS16K.items():
        emit_file(os.path.join(outdir, stem.replace("16384", "8192") + "_gfx950.s"), kname.replace("16384", "8192"),
                  build_row16k(kind))
    g.update(NEXT_SGPR=98)
    emit_file(os.path.join(outdir, "ntt_fwd8192x2_gfx950.s"), "nflhip_ntt_fwd8192x2_asm", build_row16k("fwd2"), args=ARGS_STD + [("i32", 48)])
    g.update(NEXT_SGPR=96)
    if experiments:
        g.update(NEXT_SGPR=102)
        emit_file(os.path.join(outdir, "polymul8192p_gfx950.s"), "nflhip_polymul8192p_asm", build_row16k_loop(),
                  args=ARGS_STD + [("i32", 48), ("i32", 52)])
    # experiment (nflhip_debug_fused_grid(3)): the inverse pipelines of a 4096-word row on the ring-mode map (128 VGPRs: four
    # workgroups per CU instead of three, one butterfly at a time)
    configure("ring", 1)
    g.update(NEXT_SGPR=102, LDS_BYTES=SLAB_BYTES)
    for kind, (stem, kname) in KERNELS_FUSED.items():
        emit_file(os.path.join(outdir, stem + "r_gfx950.s"), kname.replace("_asm", "r_asm"), build_fused_rows(kind), args=ARGS_FUSED)
    # ... and the metric product itself on that map: measured same-box against nflhip_polymul4096nt_asm (profiles/
    # r04_ring_vs_pair_4096.txt): 2.89 ms per 16 384 products either way -- the product is bound by its arithmetic, a fourth
    # workgroup per CU buys nothing.  Only emitted with NFL_GEN_EXPERIMENTS=1.
    if experiments:
        emit_file(os.path.join(outdir, "fused_polymul4096r_gfx950.s"), "nflhip_fused_polymul4096r_asm", build_fused_rows("polymul"), args=ARGS_FUSED)
    g.update(NEXT_SGPR=96)
    # transform-fused pipelines on the row-resident map: rows of 16384 and 8192 words
    for groups, words in ((4, 16384), (2, 8192)):
        configure("ring", groups)
        g.update(NEXT_SGPR=102)
        for kind, (stem, kname) in KERNELS_FUSED.items():
            emit_file(os.path.join(outdir, stem.replace("4096", str(words)).replace("_%d" % words, "_%d" % words) + "_gfx950.s"),
                      kname.replace("4096", str(words)), build_fused_rows(kind), args=ARGS_FUSED)
        g.update(NEXT_SGPR=96)
    configure(ring, 4)
    # 32768-word rows: one operand register-resident in a 1024-thread workgroup (4 sub-groups x 2 blocks)
    g = globals()
    g.update(ROW_LG=3, NEXT_SGPR=max(NEXT_SGPR, 98))
    for kind, stem in (("fwd", "ntt_fwd32768"), ("inv", "ntt_inv32768"), ("polymul_ntt", "polymul_ntt32768"),
                       ("fwd_s", "ntt_fwd32768s"), ("polymul_s", "polymul_ntt32768s")):
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), "nflhip_%s_asm" % stem, build_row32k(kind))
    # ... and the fused inverse pipelines of such a row: INTT(b -+ a k), the key row's base and stride flag behind the standard arguments
    # ... the forward transform of a compact (int8) Gaussian polynomial: one byte per coefficient in, NTT words of every modulus out
    emit_file(os.path.join(outdir, "ntt_fwd32768i8_gfx950.s"), "nflhip_ntt_fwd32768i8_asm", build_row32k("fwd_i8"))
    for kind, stem in (("fma_fwd_i8", "fused_fma_fwd32768i8"), ("enc2_i8", "fused_enc2_32768i8")):   # ... and the forward pipelines on such a polynomial: out0 = NTT(x) k0 + e0' [, out1 = NTT(x) k1 + e1']
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), "nflhip_%s_asm" % stem, build_row32k(kind),
                  args=ARGS_STD + [("ptr", 48), ("ptr", 56), ("ptr", 64), ("ptr", 72)])
    g.update(NEXT_SGPR=102)
    for kind in ("fms_inv", "fma_inv"):
        emit_file(os.path.join(outdir, "fused_%s32768_gfx950.s" % kind), "nflhip_fused_%s32768_asm" % kind, build_row32k(kind),
                  args=ARGS_STD + [("ptr", 48), ("i32", 56)])
    g.update(NEXT_SGPR=98)
    configure("ring", 4)
    if os.environ.get("NFL_DEBUG16K"):   # checkpoint variants for bisecting a fault: kernel ends after phase n
        kind = os.environ["NFL_DEBUG16K"]
        for n in range(-3, 10):
            emit_file("/tmp/dbg16k_p%d.s" % (n + 3), "nflhip_dbg16k_%d" % (n + 3), build_row16k(kind, stop=n))


if __name__ == "__main__":
    main()
