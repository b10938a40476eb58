"""n = 65536 / 32768: the three-role pipeline kernel (build_pipe; with fused=True the roles are handed out by xcd.fused_header)."""
import os

from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs
from .arith import T, ct_bfly, final_bfly, gs_bfly, v_mask
from .twiddles import PASS_TW, ct_stage, gs_stage, tw_slot, twreg
from .block4096 import build_body, strided_rows
from .xcd import fused_header, lifo_product_loaded, lifo_product_store

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
    em.valu("v_mov_b32_e32 v%d, s25" % (cfg.V_PHI,))
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
    R("s_lshl_b32 s43, s3, %d" % (cfg.PIPE_LOGN + 4,))       # tw = psi + cm * n * 16
    R("s_add_u32 s22, s10, s43")
    R("s_addc_u32 s23, s11, 0")
    R("s_cmp_eq_u32 s42, 0")
    R("s_cbranch_scc1 .Lrole_v")
    R("s_cmp_eq_u32 s42, 3")
    R("s_cbranch_scc1 .Lrole_i")


def build_pipe(logn=None, fused=False, b_ntt=False, level=0):
    """n = 65536 (logn 16): radix-16 streaming roles, 16 + 3 x 4 = 28 workgroups per row.
    n = 32768 (logn 15): radix-8 streaming roles (a thread's 16 registers hold two columns of 8 words), 8 + 3 x 2 = 14.
    fused: ONE launch of persistent workgroups for the whole batch; the three roles of a row run on ONE XCD, ordered by
    a per-XCD ticket queue and per-row completion counters, so the intermediates travel through that XCD's L2
    (see fused_header below).
    b_ntt: operand b is ALREADY transformed (canonical words in the reference's order): there is no forward streaming role
    for it -- NV + 2 NSW workgroups per row -- and the block products read its block as it lies (16 consecutive words per
    thread: what the inner forward passes would have left in the registers), like the stand-alone polymul_ntt kernel."""
    if logn is not None:
        cfg.PIPE_LOGN = logn
    em = Emitter()
    R = em.raw
    n_words = 1 << cfg.PIPE_LOGN
    RL = cfg.PIPE_LOGN - 12                                   # global stages done by the streaming roles: 4 (radix 16) or 3 (radix 8)
    RADIX = 1 << RL
    NV = n_words // 4096                                  # block products per row
    NSW = 4 if RL == 4 else 2                             # streaming workgroups per row and operand
    PER_ROW = NV + (2 if b_ntt else 3) * NSW
    assert not (fused and b_ntt)
    assert not (level and b_ntt)                          # level: the block products run on incomplete transforms (incomplete.py);
                                                          # the host then passes the records with (n / 2^level)^-1 for role I
    CG_LOG = 11 if RL == 4 else 12                        # bytes (log2) of one column group: 256 columns x (16 / RADIX) x 8 B
    stride = n_words // RADIX * 8                         # bytes between x[o + k n/RADIX]
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c_v, a_v, b_v, psi
    R("s_load_dwordx2 s[12:13], s[0:1], 0x20")           # mc
    R("s_load_dword s14, s[0:1], 0x28")                  # nm
    if not fused:
        R("s_load_dwordx16 s[56:71], s[0:1], 0x30")      # cntV cntF cntI pad | fa_src fa_dst fb_src fb_dst inv pad
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
    for t0 in sorted(set(cfg.V_T)):
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
        bufs = [cfg.V_A, cfg.V_B]
        seq_of = {0: group_io(bufs[0], cfg.S_AROW, 0)}
        tw_last = 0
        for st in (tuple(range(RL)) if kind == "F" else tuple(range(RL - 1, -1, -1))):
            tw_last = PASS_TW["F1" if kind == "F" else "I3"](em, vm, st)
        emit_consts(em)
        for gi in range(GROUPS):
            buf = bufs[gi & 1]
            if gi + 1 < GROUPS:
                seq_of[gi + 1] = group_io(bufs[(gi + 1) & 1], cfg.S_AROW, (gi + 1) * GSTEP)
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
            group_io(buf, cfg.S_CROW, gi * GSTEP, store=True)
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
        R("s_lshr_b32 s43, s87, %d" % (32 - (cfg.PIPE_LOGN + 3),))
        R("s_lshl_b32 s42, s87, %d" % (cfg.PIPE_LOGN + 3,))      # row * n * 8
        R("s_lshl_b32 s86, s89, %d" % CG_LOG)
        R("s_add_u32 s42, s42, s86")                         # + the bytes of q column groups (no carry: the low bits were zero)
        if cfg.SCRATCH_ALIAS:   # ablation: the scratch rows of the whole batch laid over a window of SCRATCH_ALIAS rows (cache-resident)
            R("s_add_u32 s16, s16, s42")
            R("s_addc_u32 s17, s17, s43")
            R("s_and_b32 s44, s87, %d" % (cfg.SCRATCH_ALIAS - 1,))
            R("s_lshl_b32 s44, s44, %d" % (cfg.PIPE_LOGN + 3,))
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
    if fused and cfg.FUSED_NT:   # the operands are read once; the scratch they are written to is what the L2 should keep
        em.lines[mark:] = [l + " nt" if "global_load_dwordx2" in l else l for l in em.lines[mark:]]

    em.lines.append(".Lrole_i:")
    em.comment("role I: lazy words of the block kernel -> global stages 3..0 with n^-1 -> canonical x[o + k n/16]")
    if fused:
        em.lines.append(".Lbody_i:")
    else:
        R("s_cmp_ge_u32 s86, s58")
        R("s_cbranch_scc1 .Lidle")
        R("s_lshr_b32 s43, s87, %d" % (32 - (cfg.PIPE_LOGN + 3),))
        R("s_lshl_b32 s42, s87, %d" % (cfg.PIPE_LOGN + 3,))
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
        em.lines[mark:] = [l + (" nt" if cfg.FUSED_LIFO else cfg.FUSED_LOADS) if "global_load_dwordx2" in l else l for l in em.lines[mark:]]
    if fused and cfg.FUSED_NT:   # ... and the result is written once
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
        if cfg.SCRATCH_ALIAS and not b_ntt:
            R("s_and_b32 s44, s87, %d" % (cfg.SCRATCH_ALIAS - 1,))
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
    R("s_mov_b32 s88, %d" % (cfg.PIPE_LOGN - 12,))
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
    strided_rows(em, vm, cfg.V_A, cfg.S_AROW, 2048)
    if b_ntt:
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), cfg.V_TID))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (cfg.V_B + 4 * i, cfg.V_B + 4 * i + 3, T(1, 0), cfg.S_BROW, 16 * i))
    else:
        strided_rows(em, vm, cfg.V_B, cfg.S_BROW, 2048)
    tw_seq = {}
    for st in range(4):
        tw_seq[("F1", st)] = PASS_TW["F1"](em, vm, st)
    emit_consts(em)
    mark = len(em.lines)
    build_body(em, vm, "polymul_ntt" if b_ntt else "polymul", tw_seq, "_v", level=level)
    if fused:
        mod = " nt" if cfg.FUSED_LIFO else cfg.FUSED_LOADS
        em.lines[mark_v:mark] = [l + mod if "global_load_dwordx2" in l else l for l in em.lines[mark_v:mark]]
    if fused and cfg.FUSED_LIFO:
        # splice the pool protocol into the product: "loaded" after its first barrier, the c' slot in front of its stores
        body = em.lines[mark:]
        e1, e2 = Emitter(), Emitter()
        lifo_product_loaded(e1, NV)
        lifo_product_store(e2, cfg.PIPE_LOGN + 3)
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
