"""The 62-bit modular arithmetic as instruction generators: Shoup quotient / low chains, the two-bit fold, Cooley-Tukey and
Gentleman-Sande butterflies, conditional subtractions, the point-wise product."""
import os

from . import state as cfg
from .emitter import vp

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
    return cfg.V_T[s] + k


def quotient(s, y, tw, exact):
    """Q = floor(y*w'/2^64) (exact) or that minus e, e in {0,1} (not exact). y = VGPR pair base of
    the multiplicand; tw = (w0, w1, a0, a1) operand strings (VGPR or SGPR)."""
    w0, w1, a0, a1 = tw
    A, P, Q, ZP = T(s, 6), T(s, 2), T(s, 8), T(s, 14)
    if exact:
        yield "v_mul_hi_u32 v%d, v%d, %s" % (ZP, y, a0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), cfg.S_DUMMY, y, a1, vp(ZP)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), cfg.S_CARRY[s], y + 1, a0, vp(A)), cfg.S_CARRY[s], None
    else:
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(A), cfg.S_DUMMY, y + 1, a0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(A), cfg.S_CARRY[s], y, a1, vp(A)), cfg.S_CARRY[s], None
    yield "v_mov_b32_e32 v%d, v%d" % (P, A + 1), None, None
    yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, cfg.S_DUMMY, cfg.S_CARRY[s]), None, cfg.S_CARRY[s]
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), cfg.S_DUMMY, y + 1, a1, vp(P)), None, None


def lowchain(s, y, tw, acc, seed, after_low=None):
    """acc = seed + y*w - Q*p (mod 2^64) using p = 2^62 - delta: y*w + Q*delta - (Q << 62).
    The high-dword terms are accumulated first; after_low is an instruction that needs only
    acc's LOW dword and may overwrite y's low dword (slotted in once both are settled)."""
    w0, w1, a0, a1 = tw
    Q, H = T(s, 8), T(s, 10)
    yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, y, w1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, y + 1, w0, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, Q + 1, cfg.S_DELTA, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, Q, cfg.S_C0, vp(H)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(acc), cfg.S_DUMMY, y, w0, seed), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(acc), cfg.S_DUMMY, Q, cfg.S_DELTA, vp(acc)), None, None
    if after_low is not None:
        yield after_low
    yield "v_add_u32_e32 v%d, v%d, v%d" % (acc + 1, acc + 1, H), None, None


def v_mask():
    """VGPR holding 0x3fffffff: the one temporary slot (T + 1 of stream 0) no butterfly uses"""
    return cfg.V_T[0] + 1


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
        yield "v_and_b32_e32 v%d, %s, v%d" % (src + 1, cfg.S_MASK, src + 1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(dst), cfg.S_DUMMY, t, cfg.S_DELTA, vp(src)), None, None


def ct_bfly(x, y, tw):
    """Cooley-Tukey: x' = x + w*y, y' = x - w*y (any 64-bit words in, any 64-bit words out)."""
    def gen(s):
        if "nobfly" in cfg.ABLATE:
            return
        U, Y2 = T(s, 4), T(s, 12)
        yield from fold2(s, U, x)
        yield from quotient(s, y, tw, exact=False)
        yield "v_lshl_add_u64 %s, %s, 1, %s" % (vp(Y2), vp(U), cfg.S_P3), None, None
        # x' = U + m (m < 3p) lands in x; y' = (2U + 3p) - x'.  The low-dword subtract is issued as soon as
        # x' low is final, so its borrow is old enough when v_subb consumes it (no hazard nop).
        sub_lo = ("v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (y, cfg.S_BORROW[s], Y2, x), cfg.S_BORROW[s], None)
        yield from lowchain(s, y, tw, x, vp(U), after_low=sub_lo)
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (y + 1, cfg.S_DUMMY, Y2 + 1, x + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
    return gen


def gs_bfly(x, y, tw):
    """Gentleman-Sande with the negated mirrored twiddle: x' = fold(x + y), y' = (y - x)*w; inputs < 2p."""
    def gen(s):
        if "nobfly" in cfg.ABLATE:
            return
        E, D, SUM = T(s, 12), T(s, 4), T(s, 16)
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(y), cfg.S_P2), None, None
        yield "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (D, cfg.S_BORROW[s], E, x), cfg.S_BORROW[s], None
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(SUM), vp(x), vp(y)), None, None
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (D + 1, cfg.S_DUMMY, E + 1, x + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
        yield from fold2(s, x, SUM)
        yield from quotient(s, D, tw, exact=True)
        yield from lowchain(s, D, tw, y, "0")
    return gen


def csub_p(s, reg):
    """reg = reg >= p ? reg - p : reg  (borrow trick)."""
    E = T(s, 12)
    yield "v_sub_co_u32_e64 v%d, %s, v%d, %s" % (E, cfg.S_BORROW[s], reg, "s24"), cfg.S_BORROW[s], None
    # subb with an SGPR subtrahend needs it in src0 of the *rev* form: use a VGPR copy of p's high dword
    yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (E + 1, cfg.S_BORROW[s], reg + 1, cfg.V_PHI, cfg.S_BORROW[s]), cfg.S_BORROW[s], cfg.S_BORROW[s]
    yield "v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (reg, E, reg, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
    yield "v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (reg + 1, E + 1, reg + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]




def canon(reg):
    """any 64-bit word -> canonical [0,p): two-bit fold (< p + 4*delta) then one conditional subtract."""
    def gen(s):
        yield from fold2(s, reg, reg)
        yield from csub_p(s, reg)
    return gen


def final_bfly(x, y):
    """Last inverse stage with n^-1 folded in; canonical outputs."""
    tw_n = ("s%d" % cfg.S_NINV[0], "s%d" % cfg.S_NINV[1], "s%d" % cfg.S_NINVSH[0], "s%d" % cfg.S_NINVSH[1])
    tw_w = ("s%d" % cfg.S_W1N[0], "s%d" % cfg.S_W1N[1], "s%d" % cfg.S_W1NSH[0], "s%d" % cfg.S_W1NSH[1])

    def gen(s):
        E, D = T(s, 12), T(s, 4)
        yield "v_lshl_add_u64 %s, %s, 0, %s" % (vp(E), vp(y), cfg.S_P2), None, None
        yield "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (D, cfg.S_BORROW[s], E, x), cfg.S_BORROW[s], None
        yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (D + 1, cfg.S_DUMMY, E + 1, x + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
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
    mu0, mu1 = "s%d" % cfg.S_MU2[0], "s%d" % cfg.S_MU2[1]

    def gen(s):
        L, A, P, Q, H, E, ZP = T(s, 16), T(s, 6), T(s, 2), T(s, 8), T(s, 10), T(s, 12), T(s, 14)
        if fold_a:
            yield from fold2(s, xa, xa)
        if fold_b:
            yield from fold2(s, xb, xb)
        # T = xa*xb as four dwords: T0 = L.lo, T1 = A.lo, T2 = E.lo, T3 = E.hi
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (vp(L), cfg.S_DUMMY, xa, xb), None, None
        yield "v_mov_b32_e32 v%d, v%d" % (ZP, L + 1), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), cfg.S_DUMMY, xa, xb + 1, vp(ZP)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), cfg.S_CARRY[s], xa + 1, xb, vp(A)), cfg.S_CARRY[s], None
        yield "v_mov_b32_e32 v%d, v%d" % (P, A + 1), None, None
        yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, cfg.S_DUMMY, cfg.S_CARRY[s]), None, cfg.S_CARRY[s]
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(E), cfg.S_DUMMY, xa + 1, xb + 1, vp(P)), None, None
        # th = T >> 61 -> D pair
        D = T(s, 4)
        yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D, E, A), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D + 1, E + 1, E), None, None
        # q ~ floor(th*mu2/2^64), one-off allowed (r < 4p, folded below)
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, D + 1, mu0), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_CARRY[s], D, mu1, vp(H)), cfg.S_CARRY[s], None
        yield "v_mov_b32_e32 v%d, v%d" % (P, H + 1), None, None
        yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, cfg.S_DUMMY, cfg.S_CARRY[s]), None, cfg.S_CARRY[s]
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), cfg.S_DUMMY, D + 1, mu1, vp(P)), None, None
        # r = lo64(T) + q*delta - (q << 62)
        yield "v_mov_b32_e32 v%d, v%d" % (L + 1, A), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(L), cfg.S_DUMMY, Q, cfg.S_DELTA, vp(L)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, Q + 1, cfg.S_DELTA), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, Q, cfg.S_C0, vp(H)), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (L + 1, L + 1, H), None, None
        yield from fold2(s, xa, L)
    return gen
