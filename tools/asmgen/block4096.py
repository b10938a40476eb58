"""The 4096-word block kernels (one 256-thread workgroup per row): prologue, the three radix-16 passes each way, the LDS
exchanges, epilogues -- the metric kernel nflhip_polymul4096[nt]_asm and its siblings."""
import os

from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, canon, final_bfly, pointwise, v_mask
from .twiddles import PASS_TW, ct_stage, gs_stage

def lds_write(em, addr, base, stride):
    if "nolds" in cfg.ABLATE:
        return
    for k in range(16):
        em.raw("ds_write_b64 v%d, %s offset:%d" % (addr, vp(base + 2 * k), stride * k))


def lds_read(em, addr, base, stride):
    if "nolds" in cfg.ABLATE:
        return
    for k in range(16):
        em.raw("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), addr, stride * k))


def lane_contig_setup(em):
    """T(1,0) = byte offset of element 1024*w + l inside a 4096-word block (w = t>>6, l = t&63);
    T(1,1) = padded LDS byte address of the same element.  Per j the element 1024w + 64j + l sits at
    +512*j bytes in global memory and +544*j bytes in the padded slab."""
    g, l = T(1, 0), T(1, 6)       # (T + 1 of stream 0 holds the fold mask: in single-stream mode both streams share the temporaries)
    em.valu("v_lshrrev_b32_e32 v%d, 6, v%d" % (g, cfg.V_TID))                 # w
    em.valu("v_and_b32_e32 v%d, 63, v%d" % (l, cfg.V_TID))                    # l
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
    if "row0" in cfg.ABLATE:
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
        if "x4" in cfg.ABLATE:
            # timing ablation (round 6, wrong results by construction): the same 32 KiB row fetched as 8 x 16 bytes per thread
            # (word pairs 2t, 2t+1 of each 512-word slice) instead of 16 x 8 bytes -- does the fetch WIDTH matter to the product?
            em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (T(1, 0), cfg.V_TID))
            for k in range(8):
                seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, s[86:87] offset:%d nt" % (dst_base + 4 * k, dst_base + 4 * k + 3, T(1, 0), 0))
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
            return seq
        for k in range(16):
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(dst_base + 2 * k), cfg.V_OFF8, (k & 1) * 2048))
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
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), cfg.V_TID))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (dst_base + 4 * i, dst_base + 4 * i + 3, T(1, 0), srow, 16 * i))

    if kind == "polymul":
        row_loads(cfg.V_A, cfg.S_AROW)
        row_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "polymul_ntt":
        row_loads(cfg.V_A, cfg.S_AROW)
        thread16_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "fwd":
        row_loads(cfg.V_A, cfg.S_AROW)
    elif kind == "fwd2":
        row_loads(cfg.V_A, cfg.S_AROW)
        row_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "inv":
        lane_loads(cfg.V_A, cfg.S_AROW)
    elif kind == "inv2":
        lane_loads(cfg.V_A, cfg.S_AROW)
        lane_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "inv_mul":
        lane_loads(cfg.V_A, cfg.S_AROW)
        lane_loads(cfg.V_B, cfg.S_BROW)
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
    em.valu("v_mov_b32_e32 v%d, s25" % (cfg.V_PHI,))
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        em.valu("v_mov_b32_e32 v%d, 0x3fffffff" % (v_mask(),))

    return tw_seq


def strided_rows(em, vm, base, srow, stride, store=False, offset=0, nwords=16):
    """16 words x[t + k*stride/8] of the row at srow (+ offset bytes) <-> register pairs base+2k; returns the number
    of the last memory instruction issued (stores are counted too when a VmCounter is given)"""
    R = em.raw
    seq = 0
    R("s_mov_b64 s[86:87], %s" % (srow,))
    if "x4" in cfg.ABLATE and store and stride == 2048 and nwords == 16 and not offset:
        em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (T(1, 0), cfg.V_TID))
        for k in range(8):
            R("global_store_dwordx4 v%d, v[%d:%d], s[86:87] nt" % (T(1, 0), base + 4 * k, base + 4 * k + 3))
            R("s_add_u32 s86, s86, 0x1000")
            R("s_addc_u32 s87, s87, 0")
        return seq
    if offset:
        R("s_add_u32 s86, s86, 0x%x" % offset)
        R("s_addc_u32 s87, s87, 0")
    for k in range(nwords):
        if stride == 2048:
            off = (k & 1) * 2048
        else:
            off = 0
        if store:
            text = "global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (cfg.V_OFF8, vp(base + 2 * k), off)
            if vm is None:
                R(text)
            else:
                seq = vm.load(text)
        else:
            seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(base + 2 * k), cfg.V_OFF8, off))
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
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    em.lines.append(".Lstore%s:" % suffix)
    # ---------------- store c (x[t + 256k])
    strided_rows(em, None, cfg.V_A, cfg.S_CROW, stride, store=True)
    R("s_endpgm")


def epilogue_forward(em, vm, end=True, base=None, dst=None):
    """canonical words, then a wave-local LDS transpose so the stores are fully coalesced"""
    R = em.raw
    base = cfg.V_A if base is None else base
    run_pairs(em, [canon(base + 2 * i) for i in range(16)])
    lds_write(em, cfg.V_L2R, base, 8)
    g, l = lane_contig_setup(em)
    for j in range(16):
        R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l, 544 * j))
    R("s_waitcnt lgkmcnt(0)")
    R("s_mov_b64 s[86:87], %s" % (cfg.S_CROW if dst is None else dst,))
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


def build_body(em, vm, kind, tw_seq, suffix="", level=0):
    """level (kind "polymul" only): 1 / 2 = the product on incomplete transforms (incomplete.py)"""
    if level:
        assert kind == "polymul"
        from .incomplete import body_incomplete
        return body_incomplete(em, vm, tw_seq, suffix, level)
    R = em.raw
    has_fwd = kind in ("polymul", "polymul_ntt", "fwd", "fwd2")
    has_inv = kind not in ("fwd", "fwd2")
    fwd_bases = [cfg.V_A, cfg.V_B] if kind in ("polymul", "fwd2") else [cfg.V_A]
    passes = (["F1", "F2", "F3"] if has_fwd else []) + (["I1", "I2", "I3"] if has_inv else [])
    inv_bases = [cfg.V_A, cfg.V_B] if kind == "inv2" else [cfg.V_A]

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
            lds_write(em, cfg.V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, cfg.V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F2")
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        for base in fwd_bases:
            lds_write(em, cfg.V_L1R, base, 136)
            lds_read(em, cfg.V_L2R, base, 8)
        R("s_waitcnt lgkmcnt(0)")
        fwd_pass("F3")
    if kind == "fwd":
        epilogue_forward(em, vm)
        return em
    if kind == "fwd2":
        epilogue_forward(em, vm, end=False)
        epilogue_forward(em, vm, base=cfg.V_B, dst="s[54:55]")
        return em

    if kind in ("polymul", "polymul_ntt"):
        em.comment("point-wise product (thread q holds words 16q..16q+15 of both operands)")
        run_pairs(em, [pointwise(cfg.V_A + 2 * i, cfg.V_B + 2 * i, True, kind == "polymul") for i in range(16)])
    else:
        R("s_waitcnt vmcnt(%d)" % (vm.issued - (32 if kind in ("inv_mul", "inv2") else 16)))   # the row loads have landed
        if kind == "inv_mul":
            em.comment("point-wise product of canonical NTT-form operands (any common layout works)")
            run_pairs(em, [pointwise(cfg.V_A + 2 * i, cfg.V_B + 2 * i, False, False) for i in range(16)])
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        for base in inv_bases:
            for j in range(16):
                R("ds_write_b64 v%d, %s offset:%d" % (l, vp(base + 2 * j), 544 * j))
            lds_read(em, cfg.V_L2R, base, 8)
            R("s_waitcnt lgkmcnt(0)")

    inv_pass("I1")
    em.comment("E2'")
    for base in inv_bases:
        lds_write(em, cfg.V_L2R, base, 8)
        lds_read(em, cfg.V_L1R, base, 136)
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2")
    em.comment("E1'")
    for i, base in enumerate(inv_bases):
        if i:
            R("s_barrier")       # WAR: the slab is still being read for the previous row
        lds_write(em, cfg.V_L1R, base, 136)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, cfg.V_L1W, base, 2176)
        R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3", stages=(3, 2, 1))
    if kind == "inv2":
        em.comment("stage 0 with n^-1 folded in, both rows (n = 4096 only)")
        for base, dst in ((cfg.V_A, cfg.S_CROW), (cfg.V_B, "s[54:55]")):
            run_pairs(em, [final_bfly(base + 2 * h, base + 2 * (h + 8)) for h in range(8)])
            strided_rows(em, None, base, dst, 2048, store=True)
        R("s_endpgm")
        return em

    def last_plain():
        vm.wait(tw_seq[("I3", 0)])
        gs_stage(em, cfg.V_A, 0)
    # I3's sub-stage-0 record is only used by the r > 0 tail; make sure it was requested
    if ("I3", 0) not in tw_seq:
        tw_seq[("I3", 0)] = PASS_TW["I3"](em, vm, 0)
    epilogue_inverse(em, vm, last_plain)
    return em
