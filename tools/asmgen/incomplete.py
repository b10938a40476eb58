"""The metric product with INCOMPLETE transforms (round 6): both forward transforms stop L stages early (L = 1 or 2), the
point-wise product becomes a product of degree-(2^L - 1) residues mod (X^(2^L) - zeta), and the inverse starts L stages
late.  Coefficient form in and out, so the result is bit-identical to the complete kernel's (every word canonical).

After global stage S - 1 (S = 12 - L) thread q holds, in register pairs G g .. G g + G - 1 (G = 2^L), the residue of its
operand modulo X^G - zeta_g, coefficients in natural order, with zeta_g = +w for even g and -w for odd g, w = the twiddle
of the last retained stage that produced the pair of groups (the same algebra on Python integers:
tests/test_asm_emulated.py::test_incomplete_transform_algebra_in_integers).
Base multiplication, per group (c_k = sum_{i+j=k} a_i b_j + zeta sum_{i+j=k+G} a_i b_j):
  * the 2 G inputs are folded (< 2^62 + 3 delta); bz_j = fold(zeta b_j) by one Shoup product (one-off quotient), IN PLACE, as soon
    as the raw b_j has been used for the last time (c_k is computed for k = G-1 .. 0);
  * c_k is ONE G-term dot product accumulated lazily in 128 bits (column-wise v_mad_u64_u32 chains; the carries out of the 64-bit
    column accumulators are counted by v_addc from two alternating SGPR pairs, which keeps the VALU-writes-SGPR hazard covered by
    the interleaved second stream) and ONE Barrett reduction for sums below 2^127:
        th = T >> 63,  mu = floor(2^127 / p) = 2^65 + m  (m < 2^35),  q^ = 2 th + floor(th m / 2^64)   (mod 2^64: T / p may
        reach 2^64 for lazily reduced operands, and only q^ mod 2^64 enters r),  r = T - q^ p  in [0, 4p)  (q - q^ <= 3),
    then the two-bit fold, like the point-wise product of the complete kernel.
The host hands this kernel a ModConst record whose n^-1 fields are (n / G)^-1 and whose mu2 field is m (api.hip build_tables).
Instruction count per thread against the complete kernel: -(2 x 16 x 18 + 16 x 20) per dropped stage, + the base multiplication
instead of 16 point-wise products (tools/asm_cost.py prints both)."""
from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, fold2, lowchain, quotient
from .block4096 import epilogue_inverse, lds_read, lds_write, prologue
from .twiddles import PASS_TW, ct_stage, gs_stage, tw_lane_stage, tw_slot, twreg


def barrett127(s, dst):
    """dst = T - q^ p (< 4p) for T = (L.lo, A.lo, E.lo, E.hi) < 2^127; s32 / s33 = m0 / m1"""
    L, A, Q, H, E, ZP, D = T(s, 16), T(s, 6), T(s, 8), T(s, 10), T(s, 12), T(s, 14), T(s, 4)
    m0, m1 = "s%d" % cfg.S_MU2[0], "s%d" % cfg.S_MU2[1]
    yield "v_alignbit_b32 v%d, v%d, v%d, 31" % (D, E, A), None, None
    yield "v_alignbit_b32 v%d, v%d, v%d, 31" % (D + 1, E + 1, E), None, None
    yield "v_mul_hi_u32 v%d, v%d, %s" % (ZP, D, m0), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, D + 1, m0, vp(ZP)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, D, m1, vp(H)), None, None
    yield "v_mov_b32_e32 v%d, v%d" % (ZP, H + 1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), cfg.S_DUMMY, D + 1, m1, vp(ZP)), None, None
    yield "v_lshl_add_u64 %s, %s, 1, %s" % (vp(Q), vp(D), vp(Q)), None, None
    # r = lo64(T) + q*delta - (q << 62)
    yield "v_mov_b32_e32 v%d, v%d" % (L + 1, A), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(dst), cfg.S_DUMMY, Q, cfg.S_DELTA, vp(L)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, Q + 1, cfg.S_DELTA), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, Q, cfg.S_C0, vp(H)), None, None
    yield "v_add_u32_e32 v%d, v%d, v%d" % (dst + 1, dst + 1, H), None, None


def dot(s, xs, ys):
    """T = sum x_i y_i as dwords (L.lo, A.lo, E.lo, E.hi); x_i, y_i: register pairs of folded words (high dwords <= 2^30 + 2)"""
    L, P, A, Q, E = T(s, 16), T(s, 2), T(s, 6), T(s, 8), T(s, 12)
    cregs = [cfg.S_CARRY[s], cfg.S_BORROW[s]]
    G = len(xs)

    def chain(acc, addend, terms, capture_from, cnt):
        pending, ncap, first = None, 0, True
        for idx, (x, y) in enumerate(terms):
            cap = idx >= capture_from
            creg = cregs[ncap & 1] if cap else cfg.S_DUMMY
            yield ("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(acc), creg, x, y, addend if idx == 0 else vp(acc)), creg if cap else None, None)
            if pending is not None:
                yield ("v_addc_co_u32_e64 v%d, %s, %s, 0, %s" % (cnt, cfg.S_DUMMY, "0" if first else "v%d" % cnt, pending), None, pending)
                first = False
            pending = creg if cap else None
            ncap += cap
        # the accumulator's high dword moves into the next column's addend, then the last carry is counted
        yield "v_mov_b32_e32 v%d, v%d" % (cnt - 1, acc + 1), None, None
        yield ("v_addc_co_u32_e64 v%d, %s, %s, 0, %s" % (cnt, cfg.S_DUMMY, "0" if first else "v%d" % cnt, pending), None, pending)

    yield from chain(L, "0", [(xs[i], ys[i]) for i in range(G)], 1, P + 1)
    cross = []
    for i in range(G):
        cross += [(xs[i], ys[i] + 1), (xs[i] + 1, ys[i])]
    yield from chain(A, vp(P), cross, 3, Q + 1)
    for i in range(G):
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(E), cfg.S_DUMMY, xs[i] + 1, ys[i] + 1, vp(Q) if i == 0 else vp(E)), None, None


def base_mul(a0, b0, G, tw, negate, rtmp, negtmp):
    """one group: pairs a0 .. a0 + 2G - 2 and b0 ..; tw = the VGPR record (w0, w1, w'0, w'1) of +zeta; negate: zeta = -w;
    rtmp: G - 1 spare register pairs per stream; negtmp: 4 spare registers per stream"""
    def gen(s):
        # (a0 / b0: the first of G consecutive register pairs, or the list of the G pairs -- rows32k streams b through ring slots)
        a = list(a0) if isinstance(a0, (list, tuple)) else [a0 + 2 * i for i in range(G)]
        b = list(b0) if isinstance(b0, (list, tuple)) else [b0 + 2 * i for i in range(G)]
        z = tw
        if negate:
            nt = negtmp[s]
            w0, w1, s0, s1 = (int(r[1:]) for r in tw)
            yield "v_sub_co_u32_e64 v%d, %s, s24, v%d" % (nt, cfg.S_BORROW[s], w0), cfg.S_BORROW[s], None
            yield "v_not_b32_e32 v%d, v%d" % (nt + 2, s0), None, None
            yield "v_not_b32_e32 v%d, v%d" % (nt + 3, s1), None, None
            yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (nt + 1, cfg.S_DUMMY, cfg.V_PHI, w1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
            z = ("v%d" % nt, "v%d" % (nt + 1), "v%d" % (nt + 2), "v%d" % (nt + 3))
        for r in a + b:
            yield from fold2(s, r, r)
        res = {}
        for k in range(G - 1, -1, -1):
            ys = [b[k - i] if k - i >= 0 else b[k - i + G] for i in range(G)]   # (b[j] for j > k already holds zeta b_j)
            yield from dot(s, a, ys)
            dst = T(s, 16) if k == 0 else (rtmp[s][k - 1] if isinstance(rtmp[s], (list, tuple)) else rtmp[s] + 2 * (k - 1))
            yield from barrett127(s, dst)
            res[k] = dst
            if k:
                yield from quotient(s, b[k], z, exact=False)
                yield from lowchain(s, b[k], z, b[k], "0")
                yield from fold2(s, b[k], b[k])
        for k in range(G):
            yield from fold2(s, a[k], res[k])
    return gen


def build_incomplete(level=2):
    """nflhip_polymul4096i{level}_asm: kind "polymul" of block4096.build with L = level stages dropped each way"""
    em = Emitter()
    vm = VmCounter(em)
    tw_seq = prologue(em, vm, "polymul")
    return body_incomplete(em, vm, tw_seq, "", level)


def body_incomplete(em, vm, tw_seq, suffix, level):
    """block4096.build_body(kind "polymul") on incomplete transforms: everything behind the prologue (operand rows and F1's
    twiddle records requested, constants in place).  Also role V of the n = 65536 / 32768 pipeline kernels (r > 0: the
    epilogue's plain last stage; the scale of the shorter inverse is folded in by the streaming inverse role)."""
    assert level in (1, 2)
    G = 1 << level
    keep = 4 - level                      # sub-stages of F3 / I1 that remain
    R = em.raw
    bases = [cfg.V_A, cfg.V_B]
    # twiddle slots: F3's last retained sub-stage stays resident through the base multiplication (it is zeta); I1's records of
    # that sub-stage therefore land in the slots of the first dropped sub-stage, which nothing else uses
    last = keep - 1
    i1_slot = lambda s, g: tw_slot(s, g) if s != last else tw_slot(s + 1, g)
    load_i1 = lambda s: tw_lane_stage(em, vm, s, cfg.V_TID, cfg.S_K["I1"], True, slot=i1_slot)
    # scratch of the base multiplication: the top slots of the twiddle file (free between F2 and I2's prefetch)
    top = cfg.V_TW + 60
    negtmp = [top - 4, top - 8]
    rtmp = [top - 8 - 2 * (G - 1), top - 8 - 4 * (G - 1)]

    def fwd_pass(name, nxt, stages=(0, 1, 2, 3)):
        em.comment("%s; prefetching %s" % (name, nxt))
        for s in stages:
            vm.wait(tw_seq[(name, s)])
            ct_stage(em, bases, s)
            if nxt == "I1":
                tw_seq[("I1", s)] = load_i1(s)
            elif nxt == "F3" and s >= keep:
                pass
            else:
                tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)

    fwd_pass("F1", "F2")
    for i, base in enumerate(bases):
        em.comment("E1")
        if i:
            R("s_barrier")
        lds_write(em, cfg.V_L1W, base, 2176)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, cfg.V_L1R, base, 136)
        R("s_waitcnt lgkmcnt(0)")
    fwd_pass("F2", "F3")
    em.comment("E2: wave-local 16-lane transposes")
    for base in bases:
        lds_write(em, cfg.V_L1R, base, 136)
        lds_read(em, cfg.V_L2R, base, 8)
    R("s_waitcnt lgkmcnt(0)")
    fwd_pass("F3", "I1", stages=tuple(range(keep)))

    em.comment("base multiplication mod X^%d -+ zeta (thread q holds words 16q..16q+15 of both operands)" % G)
    jobs = []
    for g in range(16 // G):
        tw = twreg(tw_slot(last, g // 2))
        jobs.append(base_mul(cfg.V_A + 2 * G * g, cfg.V_B + 2 * G * g, G, tw, bool(g & 1), rtmp, negtmp))
    # stream 0 takes the even (+zeta) groups' partner order so that both streams have the same length: pair (g, g + 1)
    run_pairs(em, jobs)

    # inverse: I1 starts at its sub-stage `last`; I2's records are requested in the order I2 consumes them (3, 2, 1, 0), each as
    # soon as its slots are free: level 2 -- sub-stage 3 (slots 7-14: the base multiplication's scratch) right away, 2 and 1 after
    # I1's first stage (slots 3, 4 held I1's records); level 1 -- 3 and 2 after I1's first stage (its records sat in slots 7-10)
    em.comment("I1 (sub-stages %s); prefetching I2" % (tuple(range(last, -1, -1)),))
    after = {2: {"start": [3], 1: [2, 1], 0: [0]}, 1: {"start": [], 2: [3, 2], 1: [1], 0: [0]}}[level]
    for s2 in after["start"]:
        tw_seq[("I2", s2)] = PASS_TW["I2"](em, vm, s2)
    for s in range(last, -1, -1):
        vm.wait(tw_seq[("I1", s)])
        gs_stage(em, cfg.V_A, s, slot=i1_slot)
        for s2 in after[s]:
            tw_seq[("I2", s2)] = PASS_TW["I2"](em, vm, s2)
    em.comment("E2'")
    lds_write(em, cfg.V_L2R, cfg.V_A, 8)
    lds_read(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    em.comment("I2; prefetching I3")
    for s in (3, 2, 1, 0):
        vm.wait(tw_seq[("I2", s)])
        gs_stage(em, cfg.V_A, s)
        tw_seq[("I3", s)] = PASS_TW["I3"](em, vm, s)
    em.comment("E1'")
    lds_write(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    lds_read(em, cfg.V_L1W, cfg.V_A, 2176)
    R("s_waitcnt lgkmcnt(0)")
    em.comment("I3")
    for s in (3, 2, 1):
        vm.wait(tw_seq[("I3", s)])
        gs_stage(em, cfg.V_A, s)

    def last_plain():
        vm.wait(tw_seq[("I3", 0)])
        gs_stage(em, cfg.V_A, 0)
    epilogue_inverse(em, vm, last_plain, suffix)
    return em
