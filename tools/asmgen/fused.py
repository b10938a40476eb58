"""Transform-fused pipelines (the LWE demo's encrypt / decrypt bodies): n = 4096 on the pair map (build_fused), 4096 / 8192 /
16384 on the row-resident map (build_fused_rows)."""
import os

from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, csub_p, ct_bfly, final_bfly, fold2, gs_bfly, pointwise
from .twiddles import PASS_TW, Ring, ct_stage, gs_stage, tw_lane_stage, tw_slot, twreg
from .block4096 import epilogue_inverse, lane_contig_setup, lds_read, lds_write

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
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_OFF8, cfg.V_TID))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (cfg.V_BIDX, cfg.V_TID))                     # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (cfg.V_L1W, cfg.V_TID, cfg.V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_L1W, cfg.V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (cfg.V_L1R, cfg.V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (cfg.V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (cfg.V_L1R, cfg.V_BIDX, cfg.V_L2R, cfg.V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_L1R, cfg.V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (cfg.V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (cfg.V_L2R, cfg.V_TID, cfg.V_L2R))              # 17*t*8
    for s in sorted(set(cfg.V_T)):
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
    R("s_mov_b32 %s, s76" % cfg.S_FMT)
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
        R("s_bfe_u32 %s, %s, 0x%x" % (cfg.S_F, cfg.S_FMT, (4 << 16) | (4 * k)))
        R("s_mul_i32 s42, s2, s%d" % stride)
        R("s_mul_hi_u32 s45, s42, s14")
        R("s_mul_i32 s44, s42, s14")
        R("s_add_u32 s44, s44, s3")
        R("s_addc_u32 s45, s45, 0")
        R("s_lshl_b64 s[44:45], s[44:45], 15")
        R("s_add_u32 s87, %s, 11" % cfg.S_F)
        R("s_mov_b32 s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], s87")
        R("s_cmp_eq_u32 %s, 0" % cfg.S_F)
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
        seq = fused_x_loads(em, vm, cfg.V_A, cfg.S_AROW, 0, "x0")
        seq = fused_x_loads(em, vm, cfg.V_B, cfg.S_BROW, 1, "x1")
        first = "F1"
    else:
        # x0 and the key row first (the product needs them), x1 behind them; then the part of I1's first sub-stage that
        # fits beside the key row (records g = 1..7 in slots 8..14; the key row occupies slots 0..7 until it is consumed)
        fused_lane_loads(em, vm, cfg.V_A, cfg.S_AROW)
        seq = (fused_lane_loads(em, vm, cfg.V_TW, cfg.S_K0ROW, stream=False)[-1], fused_lane_loads(em, vm, cfg.V_B, cfg.S_BROW))
        first = None
    tw_seq = {}
    if first:
        for s in (0, 1, 2, 3):
            tw_seq[(first, s)] = PASS_TW[first](em, vm, s)
    else:
        tw_seq[("I1", 3, "late")] = tw_lane_stage(em, vm, 3, cfg.V_TID, cfg.S_K["I1"], True, groups=range(1, 8))
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
    em.valu("v_mov_b32_e32 v%d, s25" % (cfg.V_PHI,))
    return tw_seq, seq


def fused_x_loads(em, vm, dst, srow, k, tag):
    """x[t + 256 j] -> register pair j (the layout F1 starts from), whatever the operand's format: 16 vector loads on every
    path, so the static load count of the VmCounter holds"""
    R = em.raw
    A = T(1, 0)
    R("s_bfe_u32 %s, %s, 0x%x" % (cfg.S_F, cfg.S_FMT, (4 << 16) | (4 * k)))
    R("s_mov_b64 s[86:87], %s" % (srow,))
    R("s_cmp_eq_u32 %s, 0" % cfg.S_F)
    R("s_cbranch_scc0 .L%s_compact" % tag)
    seq = 0
    for j in range(16):
        seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d nt" % (vp(dst + 2 * j), cfg.V_OFF8, (j & 1) * 2048))
        if j & 1:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_compact:" % tag)
    R("s_cmp_eq_u32 %s, 1" % cfg.S_F)
    R("s_cbranch_scc0 .L%s_i16" % tag)
    for j in range(16):
        R("global_load_sbyte v%d, v%d, s[86:87] offset:%d" % (dst + 2 * j, cfg.V_TID, 256 * j))
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_i16:" % tag)
    R("s_cmp_eq_u32 %s, 2" % cfg.S_F)
    R("s_cbranch_scc0 .L%s_i32" % tag)
    em.valu("v_lshlrev_b32_e32 v%d, 1, v%d" % (A, cfg.V_TID))
    for j in range(16):
        R("global_load_sshort v%d, v%d, s[86:87] offset:%d" % (dst + 2 * j, A, 512 * (j & 7)))
        if j == 7:
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
    R("s_branch .L%s_issued" % tag)
    em.lines.append(".L%s_i32:" % tag)
    em.valu("v_lshlrev_b32_e32 v%d, 2, v%d" % (A, cfg.V_TID))
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
    R("s_bfe_u32 %s, %s, 0x%x" % (cfg.S_F, cfg.S_FMT, (4 << 16) | (4 * k)))
    R("s_cmp_eq_u32 %s, 0" % cfg.S_F)
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


def mac128(s, x, y, addend, dst):
    """dst = fold(x * y + addend) < p + 4 delta: ONE 128-bit chain.  x < 2^62 (canonical, or p - canonical), y folded or canonical
    (< 2^62 + 3 delta), addend ANY 64-bit word: it is the addend of the low product, whose carry (weight 2^64) joins the cross sum's
    high dword -- the cross sum stays below 2^63.1, so neither that addition nor the sum itself can overflow; T = x y + addend < 2^124.1
    takes the point-wise product's Barrett step (T < 2^125: arith.pointwise), r < 4p, then the two-bit fold"""
    mu0, mu1 = "s%d" % cfg.S_MU2[0], "s%d" % cfg.S_MU2[1]
    L, A, P, Q, H, E, ZP, D = T(s, 16), T(s, 6), T(s, 2), T(s, 8), T(s, 10), T(s, 12), T(s, 14), T(s, 4)
    # T as four dwords: T0 = L.lo, T1 = A.lo, T2 = E.lo, T3 = E.hi
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(L), cfg.S_BORROW[s], x, y, vp(addend)), cfg.S_BORROW[s], None
    yield "v_mov_b32_e32 v%d, v%d" % (ZP, L + 1), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), cfg.S_DUMMY, x, y + 1, vp(ZP)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(A), cfg.S_DUMMY, x + 1, y, vp(A)), None, None
    yield "v_addc_co_u32_e64 v%d, %s, v%d, 0, %s" % (ZP, cfg.S_DUMMY, A + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(E), cfg.S_DUMMY, x + 1, y + 1, vp(ZP)), None, None
    # th = T >> 61; q ~ floor(th*mu2/2^64), one-off allowed; r = lo64(T) + q*delta - (q << 62)
    yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D, E, A), None, None
    yield "v_alignbit_b32 v%d, v%d, v%d, 29" % (D + 1, E + 1, E), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, D + 1, mu0), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_CARRY[s], D, mu1, vp(H)), cfg.S_CARRY[s], None
    yield "v_mov_b32_e32 v%d, v%d" % (P, H + 1), None, None
    yield "v_addc_co_u32_e64 v%d, %s, 0, 0, %s" % (P + 1, cfg.S_DUMMY, cfg.S_CARRY[s]), None, cfg.S_CARRY[s]
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(Q), cfg.S_DUMMY, D + 1, mu1, vp(P)), None, None
    yield "v_mov_b32_e32 v%d, v%d" % (L + 1, A), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(L), cfg.S_DUMMY, Q, cfg.S_DELTA, vp(L)), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, 0" % (vp(H), cfg.S_DUMMY, Q + 1, cfg.S_DELTA), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, %s, %s" % (vp(H), cfg.S_DUMMY, Q, cfg.S_C0, vp(H)), None, None
    yield "v_add_u32_e32 v%d, v%d, v%d" % (L + 1, L + 1, H), None, None
    yield from fold2(s, dst, L)


def fma_job(k, a, b, fold_a):
    """k = canonical(k * a + b): k a canonical key word, a a lazily reduced word (folded in place the first time), b ANY 64-bit word:
    27 instructions where product, fold, add, fold, conditional subtraction took 33"""
    def gen(s):
        if fold_a:
            yield from fold2(s, a, a)
        yield from mac128(s, k, a, b, k)
        yield from csub_p(s, k)
    return gen


def fms_job(a, k, b, subtract):
    """a = fold(b -+ a * k) < p + 4 delta, all inputs canonical (the contract of the reference's operators, ops.hpp:131,211): b - a k is
    (p - a) k + b, one chain"""
    def gen(s):
        if subtract:
            yield "v_sub_co_u32_e64 v%d, %s, s24, v%d" % (a, cfg.S_BORROW[s], a), cfg.S_BORROW[s], None
            yield "v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (a + 1, cfg.S_DUMMY, cfg.V_PHI, a + 1, cfg.S_BORROW[s]), None, cfg.S_BORROW[s]
        yield from mac128(s, a, k, b, a)
    return gen


def build_fused(kind):
    """kind: enc2 | fma_fwd | fms_inv | fma_inv"""
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    tw_seq, seq_x = prologue_fused(em, vm, kind)
    V_K = cfg.V_TW                      # key words: twiddle slots 0..7

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
                    em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), cfg.V_TID))     # (butterfly scratch: recomputed per batch of loads)
                    for i in (range(7) if s == 2 else (7,)):
                        kseq.append(vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d"
                                            % (V_K + 4 * i, V_K + 4 * i + 3, T(1, 0), k_row, 16 * i)))
            if name == "F1":
                for i, base in enumerate(bases):
                    em.comment("E1")
                    if i or not first_pass_ready:
                        R("s_barrier")       # WAR: the slab is still being read (previous operand / the first half's store transposes)
                    lds_write(em, cfg.V_L1W, base, 2176)
                    R("s_waitcnt lgkmcnt(0)")
                    R("s_barrier")
                    lds_read(em, cfg.V_L1R, base, 136)
                    R("s_waitcnt lgkmcnt(0)")
            elif name == "F2":
                em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
                for base in bases:
                    lds_write(em, cfg.V_L1R, base, 136)
                    lds_read(em, cfg.V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
        return kseq

    def fma_store(xb, fold_a, kseq, dst_row, early=None):
        """V_K = canonical(V_K * V_A + xb) -> dst_row (NTT form: thread q holds words 16q..16q+15; a wave-local LDS transpose
        makes the stores 512 B per wave instruction)"""
        em.comment("point-wise multiply-add against the key row")
        for i in range(8):
            vm.wait(kseq[i])
            run_pairs(em, [fma_job(V_K + 4 * i, cfg.V_A + 4 * i, xb + 4 * i, fold_a), fma_job(V_K + 4 * i + 2, cfg.V_A + 4 * i + 2, xb + 4 * i + 2, fold_a)])
        if early is not None:
            early()
        lds_write(em, cfg.V_L2R, V_K, 8)
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
        fused_x_expand(em, cfg.V_A, 0, "e0")
        fused_x_expand(em, cfg.V_B, 1, "e1")
        kseq = forward([cfg.V_A, cfg.V_B], cfg.S_K0ROW, True)
        if kind == "fma_fwd":
            fma_store(cfg.V_B, True, kseq, cfg.S_CROW)
            R("s_endpgm")
            return em
        state = {}

        def early():   # x2 is requested as soon as V_B is free: its latency hides behind the store of out0
            state["x2"] = fused_x_loads(em, vm, cfg.V_B, cfg.S_X2ROW, 2, "x2")
        fma_store(cfg.V_B, True, kseq, cfg.S_CROW, early)
        em.comment("second half: x2 alone, x0' stays in V_A")
        for s in (0, 1, 2, 3):
            tw_seq[("F1", s)] = PASS_TW["F1"](em, vm, s)
        vm.wait(state["x2"])
        fused_x_expand(em, cfg.V_B, 2, "e2")
        kseq = forward([cfg.V_B], cfg.S_K1ROW, False)
        fma_store(cfg.V_B, False, kseq, cfg.S_O1ROW)
        R("s_endpgm")
        return em

    # ---- fms_inv / fma_inv: point-wise step in the loaded (lane-contiguous) layout, then the inverse passes of build_body
    seq_k, seq_b = seq_x
    vm.wait(seq_k)
    em.comment("x1 -+ x0 * k0 (x1 is consumed word by word as it lands)")
    for i in range(0, 16, 2):
        vm.wait(seq_b[i + 1])
        run_pairs(em, [fms_job(cfg.V_A + 2 * j, V_K + 2 * j, cfg.V_B + 2 * j, kind == "fms_inv") for j in (i, i + 1)])
    tw_seq[("I1", 3)] = tw_lane_stage(em, vm, 3, cfg.V_TID, cfg.S_K["I1"], True, groups=(0,))   # the record the key row was in the way of
    for s in (2, 1, 0):
        tw_seq[("I1", s)] = PASS_TW["I1"](em, vm, s)
    em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
    _, l = lane_contig_setup(em)
    for j in range(16):
        R("ds_write_b64 v%d, %s offset:%d" % (l, vp(cfg.V_A + 2 * j), 544 * j))
    lds_read(em, cfg.V_L2R, cfg.V_A, 8)
    R("s_waitcnt lgkmcnt(0)")

    def first_stage():
        """I1's sub-stage 3 with the late record's butterfly last: groups 1..7 run on the records fetched beside the key row"""
        jobs = [gs_bfly(cfg.V_A + 4 * g, cfg.V_A + 4 * g + 2, twreg(tw_slot(3, g))) for g in (1, 2, 3, 4, 5, 6, 7, 0)]
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
                gs_stage(em, cfg.V_A, s)
            if nxt is not None:
                tw_seq[(nxt, s)] = PASS_TW[nxt](em, vm, s)
        if name == "I1":
            em.comment("E2'")
            lds_write(em, cfg.V_L2R, cfg.V_A, 8)
            lds_read(em, cfg.V_L1R, cfg.V_A, 136)
            R("s_waitcnt lgkmcnt(0)")
        elif name == "I2":
            em.comment("E1'")
            lds_write(em, cfg.V_L1R, cfg.V_A, 136)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, cfg.V_L1W, cfg.V_A, 2176)
            R("s_waitcnt lgkmcnt(0)")

    def last_plain():
        vm.wait(tw_seq[("I3", 0)])
        gs_stage(em, cfg.V_A, 0)
    epilogue_inverse(em, vm, last_plain)


def build_fused_rows(kind):
    """kind: enc2 | fma_fwd | fms_inv | fma_inv -- over one 4096 * ROW_G-word row per workgroup (configure("ring", ROW_G));
    polymul (experiment, ROW_G = 1): out0 = INTT(NTT(x0) (.) NTT(x1)), the metric product on the ring-mode map"""
    assert cfg.SINGLE_STREAM and cfg.ROW_G in (1, 2, 4)      # (1: a 4096-word row on the ring-mode map -- 128 VGPRs, four workgroups per CU)
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    fwd = kind in ("enc2", "fma_fwd", "polymul")
    passes = {"F0": (cfg.S_K0["F0"], None, False), "F1": (cfg.S_K["F1"], None, False), "F2": (cfg.S_K["F2"], cfg.V_BIDX, False),
              "F3": (cfg.S_K["F3"], cfg.V_TID, False), "I1": (cfg.S_K["I1"], cfg.V_TID, True), "I2": (cfg.S_K["I2"], cfg.V_BIDX, True),
              "I3": (cfg.S_K["I3"], None, True), "I0": (cfg.S_K0["I0"], None, True)}
    order = {"F0": tuple(range(cfg.ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0) if cfg.ROW_G > 1 else (3, 2, 1), "I0": tuple(range(cfg.ROW_LG - 1, 0, -1))}
    per = 16 // cfg.ROW_G
    AX = T(0, 0)
    V_K = cfg.V_TW
    row_bytes_log = 15 + cfg.ROW_LG

    # ---------------- prologue: thread map of prologue16k, operands of prologue_fused
    R("s_load_dwordx16 s[56:71], s[0:1], 0x0")           # out0 out1 x0 x1 x2 k0 k1 psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x40")           # mc
    R("s_load_dwordx2 s[72:73], s[0:1], 0x48")           # nm, logn
    R("s_load_dwordx8 s[76:83], s[0:1], 0x50")           # fmt, strides x0 x1 x2 k0 k1, count, magic
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_OFF8, cfg.V_TID))                      # tid*8
    em.valu("v_mov_b32_e32 v%d, v%d" % (cfg.V_TWA, cfg.V_TID))                              # the workgroup-wide thread index (compact inputs)
    em.valu("v_lshrrev_b32_e32 v%d, 8, v%d" % (cfg.V_BIDX, cfg.V_TID))                      # q (wave-uniform)
    R("s_nop 1")
    R("v_readfirstlane_b32 %s, v%d" % (cfg.S_Q, cfg.V_BIDX))
    R("s_nop 1")
    em.valu("v_and_b32_e32 v%d, 0xff, v%d" % (cfg.V_TID, cfg.V_TID))                        # t = tid & 255
    R("s_mul_i32 %s, %s, 0x%x" % (cfg.S_SLAB, cfg.S_Q, cfg.SLAB_BYTES))
    em.valu("v_lshrrev_b32_e32 v%d, 4, v%d" % (cfg.V_BIDX, cfg.V_TID))                      # B = t >> 4
    em.valu("v_add_u32_e32 v%d, v%d, v%d" % (cfg.V_L1W, cfg.V_TID, cfg.V_BIDX))
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_L1W, cfg.V_L1W))                       # (t + B)*8
    em.valu("v_and_b32_e32 v%d, 15, v%d" % (cfg.V_L1R, cfg.V_TID))                          # r
    em.valu("v_mov_b32_e32 v%d, 0x110" % (cfg.V_L2R,))                                  # 272
    em.valu("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (cfg.V_L1R, cfg.V_BIDX, cfg.V_L2R, cfg.V_L1R))     # 272*B + r
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_L1R, cfg.V_L1R))
    em.valu("v_mov_b32_e32 v%d, 0x88" % (cfg.V_L2R,))                                   # 17*8
    em.valu("v_mul_u32_u24_e32 v%d, v%d, v%d" % (cfg.V_L2R, cfg.V_TID, cfg.V_L2R))              # 17*t*8
    for reg in (cfg.V_L1W, cfg.V_L1R, cfg.V_L2R):
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (reg, cfg.S_SLAB, reg))                  # inside the sub-group's slab
    for t_ in sorted(set(cfg.V_T)):
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
    R("s_mov_b32 %s, s76" % cfg.S_FMT)
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
        R("s_bfe_u32 %s, %s, 0x%x" % (cfg.S_F, cfg.S_FMT, (4 << 16) | (4 * k)))
        R("s_mul_i32 s42, s2, s%d" % stride)
        R("s_mul_hi_u32 s45, s42, s14")
        R("s_mul_i32 s44, s42, s14")
        R("s_add_u32 s44, s44, s3")
        R("s_addc_u32 s45, s45, 0")
        R("s_lshl_b64 s[44:45], s[44:45], %d" % row_bytes_log)
        R("s_add_u32 s87, %s, %d" % (cfg.S_F, 11 + cfg.ROW_LG))      # compact: (x * stride) << (log2 n + f - 1)
        R("s_mov_b32 s43, 0")
        R("s_lshl_b64 s[42:43], s[42:43], s87")
        R("s_cmp_eq_u32 %s, 0" % cfg.S_F)
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
    R("s_mov_b32 %s, 1" % (cfg.S_K0["F0"],))
    R("s_mov_b32 %s, 2" % (cfg.S_K0["I0"],))
    R("s_mov_b32 s89, %s" % (cfg.S_Q,))                      # blk = q
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
        step = 256 * cfg.ROW_G
        R("s_bfe_u32 %s, %s, 0x%x" % (cfg.S_F, cfg.S_FMT, (4 << 16) | (4 * k)))
        R("s_mov_b64 s[86:87], %s" % (srow,))
        R("s_cmp_eq_u32 %s, 0" % cfg.S_F)
        R("s_cbranch_scc0 .L%s_compact" % tag)
        seq = 0
        for j in range(16):
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] nt" % (vp(dst + 2 * j), cfg.V_OFF8))
            if j < 15:
                R("s_add_u32 s86, s86, 0x%x" % (8 * step,))
                R("s_addc_u32 s87, s87, 0")
        R("s_branch .L%s_issued" % tag)
        em.lines.append(".L%s_compact:" % tag)
        for f, (es, op) in enumerate(((1, "global_load_sbyte"), (2, "global_load_sshort"), (4, "global_load_dword")), 1):
            if f < 3:
                R("s_cmp_eq_u32 %s, %d" % (cfg.S_F, f))
                R("s_cbranch_scc0 .L%s_f%d" % (tag, f + 1))
            if es > 1:
                em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (AX, es.bit_length() - 1, cfg.V_TWA))
            for j in range(16):
                R("%s v%d, v%d, s[86:87]" % (op, dst + 2 * j, cfg.V_TWA if es == 1 else AX))
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
        R("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
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
        x_loads(cfg.V_A, cfg.S_AROW, 0, "x0")
        seq_x = x_loads(cfg.V_B, cfg.S_BROW, 1, "x1")
    else:
        lane_loads(cfg.V_A, cfg.S_AROW)
        seq_k = lane_loads(V_K, cfg.S_K0ROW, stream=False)[-1]
        seq_b = lane_loads(cfg.V_B, cfg.S_BROW)
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
    em.valu("v_mov_b32_e32 v%d, s25" % (cfg.V_PHI,))

    def make_ring(names):
        uses = [(name, s_, g) for name in names for s_ in order[name] for g in range(1 << s_)]
        ring = Ring(em, vm, cfg.RING_SLOTS, uses, passes)
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
        for i, base in enumerate(bases if cfg.ROW_G > 1 else ()):
            em.comment("X0: thread (q, t) slot per*qq + j  ->  sub-group qq, thread t, slot q + G*j")
            if i or not first:
                R("s_barrier")       # WAR: the slabs are still being read (previous operand / the first result's store transposes)
            em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
            for k in range(16):
                qq, j = k // per, k % per
                R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * cfg.SLAB_BYTES + j * 2048 * cfg.ROW_G))
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
            for k in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F1")
        for base in bases:
            em.comment("E1")
            R("s_barrier")           # WAR against the previous exchange through this slab
            lds_write(em, cfg.V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, cfg.V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F2")
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        for base in bases:
            lds_write(em, cfg.V_L1R, base, 136)
            lds_read(em, cfg.V_L2R, base, 8)
        R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F3")

    def x_expand(dst, k, tag):
        fused_x_expand(em, dst, k, tag)

    def fma_store(xb, fold_a, k_row, dst_row, early=None):
        """V_K = canonical(key * V_A + xb) -> dst_row; the key's 16 words go into the (empty) ring's registers"""
        em.comment("the key row's block: words 16t .. 16t+15 of block q")
        block_base(k_row)
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (AX, cfg.V_TID))
        kseq = [vm.load("global_load_dwordx4 v[%d:%d], v%d, s[86:87] offset:%d" % (V_K + 4 * i, V_K + 4 * i + 3, AX, 16 * i)) for i in range(8)]
        for i in range(8):
            vm.wait(kseq[i])
            run_pairs(em, [fma_job(V_K + 4 * i, cfg.V_A + 4 * i, xb + 4 * i, fold_a), fma_job(V_K + 4 * i + 2, cfg.V_A + 4 * i + 2, xb + 4 * i + 2, fold_a)])
        if early is not None:
            early()
        lds_write(em, cfg.V_L2R, V_K, 8)
        g, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, cfg.S_SLAB, l))
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
        x_expand(cfg.V_A, 0, "e0")
        x_expand(cfg.V_B, 1, "e1")
        forward([cfg.V_A, cfg.V_B], True)
    if kind == "polymul":
        ring = make_ring(["I1", "I2", "I3", "I0"])       # (the inverse passes' first records fly under the product)
        em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
        run_pairs(em, [pointwise(cfg.V_A + 2 * i, cfg.V_B + 2 * i, True, True) for i in range(16)])
    elif fwd:
        if kind == "fma_fwd":
            fma_store(cfg.V_B, True, cfg.S_K0ROW, cfg.S_CROW)
            R("s_endpgm")
            return em
        state = {}

        def early():
            state["x2"] = x_loads(cfg.V_B, cfg.S_X2ROW16, 2, "x2")
        fma_store(cfg.V_B, True, cfg.S_K0ROW, cfg.S_CROW, early)
        em.comment("second half: x2 alone, x0' stays in V_A")
        vm.wait(state["x2"])
        x_expand(cfg.V_B, 2, "e2")
        forward([cfg.V_B], False)
        fma_store(cfg.V_B, False, cfg.S_K1ROW, cfg.S_O1ROW)
        R("s_endpgm")
        return em

    # ---- fms_inv / fma_inv (and the second half of the product)
    if kind != "polymul":
        vm.wait(seq_k)
        em.comment("x1 -+ x0 * k0 in the loaded (lane-contiguous) layout; x1 is consumed as it lands")
        for i in range(0, 16, 2):
            vm.wait(seq_b[i + 1])
            run_pairs(em, [fms_job(cfg.V_A + 2 * j, V_K + 2 * j, cfg.V_B + 2 * j, kind == "fms_inv") for j in (i, i + 1)])
        ring = make_ring(["I1", "I2", "I3", "I0"])
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, cfg.S_SLAB, l))
        for j in range(16):
            R("ds_write_b64 v%d, %s offset:%d" % (l, vp(cfg.V_A + 2 * j), 544 * j))
        lds_read(em, cfg.V_L2R, cfg.V_A, 8)
        R("s_waitcnt lgkmcnt(0)")

    def inv_pass(name, stages):
        em.comment(name)
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((name, s_, g))
                run_pairs(em, [gs_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((name, s_, g))
    inv_pass("I1", (3, 2, 1, 0))
    em.comment("E2'")
    lds_write(em, cfg.V_L2R, cfg.V_A, 8)
    lds_read(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2", (3, 2, 1, 0))
    em.comment("E1'")
    lds_write(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    lds_read(em, cfg.V_L1W, cfg.V_A, 2176)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3", (3, 2, 1, 0) if cfg.ROW_G > 1 else (3, 2, 1))
    if cfg.ROW_G > 1:
        em.comment("X0': thread (q, t) slot g + G*j  ->  thread (g, t) slot per*q + j, reader-major layout [slot][tid]")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (cfg.S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                              # q*32768 + t*8
        for k in range(16):
            g_, j = k % cfg.ROW_G, k // cfg.ROW_G
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(cfg.V_A + 2 * k), j * 2048 * cfg.ROW_G + g_ * 2048))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        rstep = 2048 * cfg.ROW_G
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, cfg.V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_OFF8 if k < 8 else AX, (k & 7) * rstep))
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I0", order["I0"])
    em.comment("stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    R("s_mov_b64 s[86:87], %s" % (cfg.S_CROW,))
    for k in range(16):
        R("global_store_dwordx2 v%d, %s, s[86:87] nt" % (cfg.V_OFF8, vp(cfg.V_A + 2 * k)))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * cfg.ROW_G,))
            R("s_addc_u32 s87, s87, 0")
    R("s_endpgm")
    return em
