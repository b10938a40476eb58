"""Row-resident kernels: 8192 / 16384-word rows (512 / 1024 threads, an outer radix-2 / radix-4 pass around the 4096-word
passes); prologue16k is shared with the 32768-word rows and the row-resident fused pipelines."""
import os

from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, canon, ct_bfly, final_bfly, gs_bfly, pointwise, v_mask
from .twiddles import Ring
from .block4096 import lane_contig_setup, lds_read, lds_write

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
    em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (cfg.V_OFF8, cfg.V_TID))                      # tid*8
    em.valu("v_lshrrev_b32_e32 v%d, 8, v%d" % (cfg.V_BIDX, cfg.V_TID))                      # q (wave-uniform)
    R("s_nop 1")                 # gfx950: a VALU VGPR write needs a wait state before v_readfirstlane reads it
    R("v_readfirstlane_b32 %s, v%d" % (cfg.S_Q, cfg.V_BIDX))
    R("s_nop 1")                 # ... and the SGPR it writes two before an SALU read
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
        em.valu("v_mov_b32_e32 v%d, 0" % (t_ + 15,))                                # the persistent zero of each stream's ZP pair
    R("s_waitcnt lgkmcnt(0)")
    if kind == "fwd2":
        R("s_lshl_b32 s2, s2, 1")
    # G = ROW_G sub-groups, LG = log2 G: r = logn - 12 (>= LG); wgx = poly * 2^(r-LG) + blkG;
    # (4096 G)-word block = ((poly*nm + cm) << (r-LG)) + blkG
    R("s_sub_u32 s88, s88, 12")
    R("s_sub_u32 s86, s88, %d" % cfg.ROW_LG)                 # r - LG
    R("s_lshr_b32 s42, s2, s86")                         # poly
    R("s_lshl_b32 s43, s42, s86")
    R("s_sub_u32 s87, s2, s43")                          # blk16
    R("s_mul_i32 s42, s42, s14")
    R("s_add_u32 s42, s42, s3")                          # row
    R("s_lshl_b32 s42, s42, s86")
    R("s_add_u32 s42, s42, s87")                         # block index
    if "row0" in cfg.ABLATE:
        R("s_and_b32 s42, s42, 3")
    R("s_lshr_b32 s43, s42, %d" % (32 - 15 - cfg.ROW_LG,))
    R("s_lshl_b32 s42, s42, %d" % (15 + cfg.ROW_LG,))        # * 4096 G words * 8 bytes
    if cfg.ALIAS_ROWS:   # ablation "bprimeN": the scratch operand of the composed 32768-word product laid over N row blocks (cache-resident)
        R("s_lshr_b32 s44, s42, %d" % (15 + cfg.ROW_LG,))
        R("s_and_b32 s44, s44, %d" % (cfg.BPRIME_ALIAS - 1,))
        R("s_lshl_b32 s44, s44, %d" % (15 + cfg.ROW_LG,))
    for base, row in ((6, 16), (8, 18), (4, 20)):
        if row in cfg.ALIAS_ROWS:
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
        R("s_lshr_b32 s43, s42, %d" % (32 - 15 - cfg.ROW_LG,))
        R("s_lshl_b32 s42, s42, %d" % (15 + cfg.ROW_LG,))
        R("s_add_u32 s98, s98, s42")
        R("s_addc_u32 s99, s99, s43")
    if kind == "fwd2":
        # the second polynomial (same modulus: nm rows further), or the first one again for the odd one out at the end of
        # the batch (transformed twice, stored twice to the same place): source s[18:19], destination s[96:97]
        R("s_add_u32 s42, s2, 1")
        R("s_cmp_lt_u32 s42, s96")
        R("s_cselect_b32 s42, s14, 0")                   # rows to the second polynomial: nm or 0
        R("s_lshr_b32 s43, s42, %d" % (32 - 15 - cfg.ROW_LG,))
        R("s_lshl_b32 s42, s42, %d" % (15 + cfg.ROW_LG,))
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
    R("s_lshl_b32 %s, 1, s86" % (cfg.S_K0["F0"],))
    R("s_add_u32 %s, %s, s87" % (cfg.S_K0["F0"], cfg.S_K0["F0"]))
    R("s_lshl_b32 %s, 2, s86" % (cfg.S_K0["I0"],))
    R("s_sub_u32 %s, %s, s87" % (cfg.S_K0["I0"], cfg.S_K0["I0"]))
    # inner pass constants of block blk = G*blkG + q
    R("s_lshl_b32 s89, s87, %d" % cfg.ROW_LG)
    R("s_add_u32 s89, s89, %s" % (cfg.S_Q,))
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
            vm.load("global_load_dwordx2 %s, v%d, s[86:87]" % (vp(dst + 2 * k), cfg.V_OFF8))
            if k < 15:
                R("s_add_u32 s86, s86, 0x%x" % (2048 * cfg.ROW_G,))
                R("s_addc_u32 s87, s87, 0")

    def block_base(srow):                                 # s[86:87] = first word of this sub-group's 4096-word block
        R("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
        R("s_add_u32 s86, s%s, s42" % (srow[2:].split(":")[0],))
        R("s_addc_u32 s87, s%s, 0" % (srow.split(":")[1][:-1],))

    def thread16_loads(dst, srow):                        # words 16t .. 16t+15 of the block (NTT-form data, after F3)
        block_base(srow)
        em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(0, 0), cfg.V_TID))
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
        row_loads(cfg.V_A, cfg.S_AROW)
        row_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "polymul_ntt":
        row_loads(cfg.V_A, cfg.S_AROW)
        thread16_loads(cfg.V_B, cfg.S_BROW)
    elif kind == "fwd":
        row_loads(cfg.V_A, cfg.S_AROW)
    elif kind == "none":      # (build_row32k issues its own loads)
        pass
    else:
        lane_loads(cfg.V_A, cfg.S_AROW)
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
    em.valu("v_mov_b32_e32 v%d, s25" % (cfg.V_PHI,))
    if os.environ.get("NFL_GEN_VGPR_OPERANDS"):
        em.valu("v_mov_b32_e32 v%d, 0x3fffffff" % (v_mask(),))


def build_row16k(kind="polymul", stop=None, level=0):
    """kind: polymul | polymul_ntt (b already in NTT form) | fwd | inv -- over one 16384-word block per workgroup
    level (kind polymul, split schedules only): 1 / 2 = the product on incomplete transforms (incomplete.py): F3 and I1 keep
    their first 4 - level sub-stages, the point-wise step is the base multiplication mod X^(2^level) -+ zeta; the host passes
    the ModConst records with (n / 2^level)^-1"""
    assert not level or (kind == "polymul" and cfg.SPLIT32K and cfg.SINGLE_STREAM)
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    passes = {"F0": (cfg.S_K0["F0"], None, False), "F1": (cfg.S_K["F1"], None, False), "F2": (cfg.S_K["F2"], cfg.V_BIDX, False),
              "F3": (cfg.S_K["F3"], cfg.V_TID, False), "I1": (cfg.S_K["I1"], cfg.V_TID, True), "I2": (cfg.S_K["I2"], cfg.V_BIDX, True),
              "I3": (cfg.S_K["I3"], None, True), "I0": (cfg.S_K0["I0"], None, True)}
    order = {"F0": tuple(range(cfg.ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0), "I0": tuple(range(cfg.ROW_LG - 1, -1, -1))}
    per = 16 // cfg.ROW_G          # register slots per 4096-word block in the row layout x[tid + 256 G k]
    has_fwd = kind != "inv"
    has_inv = kind not in ("fwd", "fwd2")
    names = (["F0", "F1", "F2", "F3"] if has_fwd else []) + (["I1", "I2", "I3", "I0"] if has_inv else [])
    keep_last = None               # level: (pass, sub-stage) whose records stay in the ring through the base multiplication (zeta)
    if level:
        keep = 4 - level
        order["F3"], order["I1"] = tuple(range(keep)), tuple(range(keep - 1, -1, -1))
        keep_last = ("F3", keep - 1)
        passes["TMP"] = None       # three reserved ring slots: the base multiplication's scratch (every register of the 128 is taken)
        names = ["F0", "F1", "F2", "F3", "TMP", "I1", "I2", "I3", "I0"]
        order["TMP"] = (0,)
    uses = [(name, s, g) for name in names for s in order[name] for g in (range(3) if name == "TMP" else range(1 << s))]
    ring = Ring(em, vm, cfg.RING_SLOTS, uses, passes)
    fwd_bases = (cfg.V_A, cfg.V_B) if kind in ("polymul", "fwd2") else (cfg.V_A,)
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
                run_pairs(em, [gs_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw)
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
            run_pairs(em, bflys(cfg.V_A, s_, g, ring.get((name, s_, g))))

    def then_b(name, s_):
        for g in range(1 << s_):
            run_pairs(em, bflys(cfg.V_B, s_, g, ring.regs((name, s_, g))))
            if (name, s_) != keep_last:
                ring.done((name, s_, g))

    def both(name, s_):
        for g in range(1 << s_):
            tw = ring.get((name, s_, g))
            jobs = []
            for ja, jb in zip(bflys(cfg.V_A, s_, g, tw), bflys(cfg.V_B, s_, g, tw)):
                jobs += [ja, jb]
            run_pairs(em, jobs)
            if (name, s_) != keep_last:
                ring.done((name, s_, g))

    def x0_w(base):
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
        for k in range(16):
            qq, j = k // per, k % per
            R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * cfg.SLAB_BYTES + j * 2048 * cfg.ROW_G))

    def x0_r(base):
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))

    def fwd_split():
        W0, BAR = "s_waitcnt lgkmcnt(0)", "s_barrier"
        exch = {"F0": (x0_w, x0_r, True),
                "F1": (lambda b_: lds_write(em, cfg.V_L1W, b_, 2176), lambda b_: lds_read(em, cfg.V_L1R, b_, 136), True),
                "F2": (lambda b_: lds_write(em, cfg.V_L1R, b_, 136), lambda b_: lds_read(em, cfg.V_L2R, b_, 8), False)}   # E2: wave-local
        pending = None          # exchange of operand b still to be finished inside the next pass
        for name in ["F0", "F1", "F2", "F3"]:
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
                    r_(cfg.V_B)
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
                w_(cfg.V_A)
                if not cross:        # wave-local transposes (LDS is in order per wave): a's reads follow its writes at once
                    r_(cfg.V_A)
                then_b(name, last)
                R(W0)
                if cross:
                    R(BAR)
                    r_(cfg.V_A)
                    R(W0)
                    R(BAR)
                w_(cfg.V_B)
                if not cross:
                    r_(cfg.V_B)
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
                    run_pairs(em, [ct_bfly(cfg.V_A + 2 * x, cfg.V_A + 2 * y, tw)])
                    if post:
                        post(x)
                        post(y)
                    i_ += 1
                ring.done((name, s_, g))

        first = [k for h in range(8) for k in (h, h + 8)]        # visiting order of a pass's stage 0
        arrive = lambda i_: R("s_waitcnt lgkmcnt(%d)" % (14 - 2 * i_))
        AXP = cfg.V_TWA                                               # (idle in the forward passes)
        em.comment("F0; X0 written out of its last stage: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
        for s_ in order["F0"][:-1]:
            fwd_stage("F0", s_)
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AXP, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
        fwd_stage("F0", order["F0"][-1], post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (
            cfg.V_OFF8 if k // per < 2 else AXP, vp(cfg.V_A + 2 * k), ((k // per) & 1) * cfg.SLAB_BYTES + (k % per) * 2048 * cfg.ROW_G)))
        ck(1)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), AX, 2048 * k))
        ck(2)
        fwd_stage("F1", 0, pre=arrive)
        for s_ in (1, 2):
            fwd_stage("F1", s_)
        em.comment("E1 written out of F1's last stage")
        R("s_barrier")               # WAR: every wave is done reading X0
        fwd_stage("F1", 3, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_L1W, vp(cfg.V_A + 2 * k), 2176 * k)))
        ck(3)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_L1R, 136 * k))
        fwd_stage("F2", 0, pre=arrive)
        for s_ in (1, 2):
            fwd_stage("F2", s_)
        em.comment("E2 (wave-local 16-lane transposes) written out of F2's last stage")
        fwd_stage("F2", 3, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_L1R, vp(cfg.V_A + 2 * k), 136 * k)))
        ck(4)
        for k in first:
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_L2R, 8 * k))
        fwd_stage("F3", 0, pre=arrive)
        for s_ in (1, 2, 3):
            fwd_stage("F3", s_)
        ck(5)

    if has_fwd and kind in ("polymul", "fwd2") and cfg.SPLIT32K:
        fwd_split()
    elif has_fwd and cfg.SPLIT32K:
        fwd_progressive()
    elif has_fwd:
            fwd_pass("F0")
            ck(1)
            for i, base in enumerate(fwd_bases):
                em.comment("X0: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
                if i:
                    R("s_barrier")       # WAR: the slabs are still being read for the previous operand
                em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
                for k in range(16):
                    qq, j = k // per, k % per
                    R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(base + 2 * k),
                                                           (qq & 1) * cfg.SLAB_BYTES + j * 2048 * cfg.ROW_G))
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
                for k in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
                R("s_waitcnt lgkmcnt(0)")
            ck(2)
            fwd_pass("F1")
            ck(3)
            for base in fwd_bases:
                em.comment("E1")
                R("s_barrier")           # WAR against the previous exchange through this slab
                lds_write(em, cfg.V_L1W, base, 2176)
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                lds_read(em, cfg.V_L1R, base, 136)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F2")
            ck(4)
            em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
            for base in fwd_bases:
                lds_write(em, cfg.V_L1R, base, 136)
                lds_read(em, cfg.V_L2R, base, 8)
            R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F3")
            ck(5)
    if kind == "fwd2":
        em.comment("two rows: canonical words, a wave-local LDS transpose per row so the stores are fully coalesced; the second"
                   " row's reduction runs under the first one's transposes")
        def transposes(base):
            lds_write(em, cfg.V_L2R, base, 8)
            _, l = lane_contig_setup(em)
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, cfg.S_SLAB, l))
            for j in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l, 544 * j))

        def stores(base, lo, hi):
            g, _ = lane_contig_setup(em)
            R("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
            R("s_add_u32 s86, s%d, s42" % lo)
            R("s_addc_u32 s87, s%d, 0" % hi)
            for j in range(16):
                R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (g, vp(base + 2 * j), (j & 7) * 512))
                if j == 7:
                    R("s_add_u32 s86, s86, 0x1000")
                    R("s_addc_u32 s87, s87, 0")
        run_pairs(em, [canon(cfg.V_A + 2 * i) for i in range(16)])
        transposes(cfg.V_A)
        run_pairs(em, [canon(cfg.V_B + 2 * i) for i in range(16)])
        R("s_waitcnt lgkmcnt(0)")
        stores(cfg.V_A, 20, 21)
        transposes(cfg.V_B)
        R("s_waitcnt lgkmcnt(0)")
        stores(cfg.V_B, 96, 97)
        R("s_endpgm")
        return em
    if kind == "fwd":
        em.comment("canonical words, then a wave-local LDS transpose so the stores are fully coalesced")
        run_pairs(em, [canon(cfg.V_A + 2 * i) for i in range(16)])
        lds_write(em, cfg.V_L2R, cfg.V_A, 8)
        g, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, cfg.S_SLAB, l))
        for j in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * j), l, 544 * j))
        R("s_waitcnt lgkmcnt(0)")
        R("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
        R("s_add_u32 s86, s20, s42")
        R("s_addc_u32 s87, s21, 0")
        for j in range(16):
            R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (g, vp(cfg.V_A + 2 * j), (j & 7) * 512))
            if j == 7:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
        R("s_endpgm")
        return em

    if kind in ("polymul", "polymul_ntt"):
        em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
        if kind == "polymul_ntt":
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_before_ring))    # b's loads (issued before the ring's) have landed
        if level:
            from .incomplete import base_mul
            G = 1 << level
            em.comment("base multiplication mod X^%d -+ zeta; scratch = three reserved ring slots" % G)
            tmp = [int(ring.get(("TMP", 0, k))[0][1:]) for k in range(3)]          # first register of each reserved slot
            assert tmp[0] % 2 == 0
            rt = [tmp[0], tmp[0] + 2, tmp[1]][:G - 1] if G == 4 else [tmp[0]]
            # (base_mul addresses its G - 1 result pairs as rtmp + 2 (k - 1): hand it a contiguous-looking map)
            jobs = []
            for g in range(16 // G):
                tw = ring.regs((keep_last[0], keep_last[1], g // 2))
                jobs.append(base_mul(cfg.V_A + 2 * G * g, cfg.V_B + 2 * G * g, G, tw, bool(g & 1), [rt, rt], [tmp[2], tmp[2]]))
            run_pairs(em, jobs)
            for g in range(1 << keep_last[1]):
                ring.done((keep_last[0], keep_last[1], g))
            for k in range(3):
                ring.done(("TMP", 0, k))
        else:
            run_pairs(em, [pointwise(cfg.V_A + 2 * i, cfg.V_B + 2 * i, True, kind == "polymul") for i in range(16)])
    else:
        R("s_waitcnt vmcnt(%d)" % (vm.issued - n_before_ring))        # the block loads have landed
        em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region")
        _, l = lane_contig_setup(em)
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (l, cfg.S_SLAB, l))
        for j in range(16):
            R("ds_write_b64 v%d, %s offset:%d" % (l, vp(cfg.V_A + 2 * j), 544 * j))
        lds_read(em, cfg.V_L2R, cfg.V_A, 8)
        R("s_waitcnt lgkmcnt(0)")
    if cfg.SPLIT32K:
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
                    run_pairs(em, [gs_bfly(cfg.V_A + 2 * x, cfg.V_A + 2 * y, tw)])
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

        AXP = cfg.V_TWA                                          # (idle in the uniform pass I3 and in I0)
        rstep = 2048 * cfg.ROW_G                                 # bytes between a reader's consecutive slots
        for s_ in order["I1"][:-1]:
            inv_stage("I1", s_)
        em.comment("E2' (wave-local): written word by word out of I1's last stage, read in I2's order")
        inv_stage("I1", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_L2R, vp(cfg.V_A + 2 * k), 8 * k)))
        ck(6)
        for k in visit(3):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_L1R, 136 * k))
        inv_stage("I2", 3, pre=arrive)
        for s_ in (2, 1):
            inv_stage("I2", s_)
        em.comment("E1': written out of I2's last stage (into positions only this wave has read), read in I3's order")
        inv_stage("I2", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_L1R, vp(cfg.V_A + 2 * k), 136 * k)))
        ck(7)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        for k in visit(3):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_L1W, 2176 * k))
        inv_stage("I3", 3, pre=arrive)
        for s_ in (2, 1):
            inv_stage("I3", s_)
        em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]; written out of I3's last stage")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (cfg.S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AXP, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AXP, AXP))                            # q*32768 + t*8
        inv_stage("I3", 0, post=lambda k: R("ds_write_b64 v%d, %s offset:%d" % (AXP, vp(cfg.V_A + 2 * k), (k // cfg.ROW_G) * 2048 * cfg.ROW_G + (k % cfg.ROW_G) * 2048)))
        ck(8)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, cfg.V_OFF8))
        first = order["I0"][:-1]
        for k in (visit(first[0]) if first else range(16)):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_OFF8 if k < 8 else AX, (k & 7) * rstep))
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
        lds_write(em, cfg.V_L2R, cfg.V_A, 8)
        lds_read(em, cfg.V_L1R, cfg.V_A, 136)
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I2", (3, 2, 1, 0))
        ck(7)
        em.comment("E1'")
        lds_write(em, cfg.V_L1R, cfg.V_A, 136)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, cfg.V_L1W, cfg.V_A, 2176)
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I3", (3, 2, 1, 0))
        ck(8)
        em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]")
        R("s_barrier")               # every wave is done reading E1'
        R("s_lshl_b32 s86, %s, 15" % (cfg.S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                              # q*32768 + t*8
        for k in range(16):
            g_, j = k % cfg.ROW_G, k // cfg.ROW_G
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(cfg.V_A + 2 * k), j * 2048 * cfg.ROW_G + g_ * 2048))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        rstep = 2048 * cfg.ROW_G                                 # bytes between a reader's consecutive slots
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * rstep, cfg.V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(cfg.V_A + 2 * k), cfg.V_OFF8 if k < 8 else AX, (k & 7) * rstep))
        R("s_waitcnt lgkmcnt(0)")
        inv_pass("I0", order["I0"][:-1])
        ck(9)
    R("s_cmp_eq_u32 s88, %d" % cfg.ROW_LG)
    R("s_cbranch_scc1 .Lmerged_last_stage")
    em.comment("r > 2: plain global stage r-2; lazy output for the outer inverse passes")
    tw = ring.get(("I0", 0, 0))
    run_pairs(em, [gs_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8), tw) for h in range(8)])
    R("s_branch .Lstore")
    em.lines.append(".Lmerged_last_stage:")
    em.comment("n == 16384: stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    em.lines.append(".Lstore:")
    R("s_mov_b64 s[86:87], %s" % (cfg.S_CROW,))
    for k in range(16):
        R("global_store_dwordx2 v%d, %s, s[86:87]" % (cfg.V_OFF8, vp(cfg.V_A + 2 * k)))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * cfg.ROW_G,))
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
    assert cfg.ROW_G in (2, 4) and cfg.SINGLE_STREAM
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    base = {"F0": (cfg.S_K0["F0"], None, False), "F1": (cfg.S_K["F1"], None, False), "F2": (cfg.S_K["F2"], cfg.V_BIDX, False),
            "F3": (cfg.S_K["F3"], cfg.V_TID, False), "I1": (cfg.S_K["I1"], cfg.V_TID, True), "I2": (cfg.S_K["I2"], cfg.V_BIDX, True),
            "I3": (cfg.S_K["I3"], None, True), "I0": (cfg.S_K0["I0"], None, True)}
    order = {"F0": tuple(range(cfg.ROW_LG)), "F1": (0, 1, 2, 3), "F2": (0, 1, 2, 3), "F3": (0, 1, 2, 3), "I1": (3, 2, 1, 0),
             "I2": (3, 2, 1, 0), "I3": (3, 2, 1, 0), "I0": tuple(range(cfg.ROW_LG - 1, 0, -1))}
    per = 16 // cfg.ROW_G          # register slots per 4096-word block in the row layout x[tid + 256 G k]
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
            vm.load("global_load_dwordx2 %s, v%d, s[100:101]" % (vp(dst_pair), cfg.V_OFF8))
            R("s_add_u32 s100, s100, 0x%x" % (2048 * cfg.ROW_G,))
            R("s_addc_u32 s101, s101, 0")
        return f
    side = {}
    for k in range(16):
        side[cfg.RING_SLOTS + k] = [side_load(cfg.V_A + 2 * k)]                    # a: under b's forward transform
        side[n_fwd + cfg.RING_SLOTS + k] = [side_load(cfg.V_B + 2 * k)]            # next b: under the inverse transform
    ring = Ring(em, vm, cfg.RING_SLOTS, uses, passes, side)
    R("s_load_dwordx2 s[96:97], s[0:1], 0x30")                             # count, G
    prologue16k(em, vm, None, "none")
    AX = T(0, 0)
    R("s_cmp_ge_u32 %s, %s" % (S_I, S_COUNT))
    R("s_cbranch_scc0 .Lhas_work")
    R("s_endpgm")
    em.lines.append(".Lhas_work:")
    R("s_mov_b32 %s, s97" % S_G)                                            # (cm is not needed any more)
    R("s_mul_i32 s42, %s, s14" % S_G)                                       # G * nm rows of 2^17 bytes between polynomials
    R("s_lshr_b32 %s, s42, %d" % (S_STRIDE[1], 32 - 15 - cfg.ROW_LG))
    R("s_lshl_b32 %s, s42, %d" % (S_STRIDE[0], 15 + cfg.ROW_LG))
    em.comment("b of the first polynomial (x[tid + 1024 k] -> slot k)")
    R("s_mov_b64 s[86:87], %s" % (cfg.S_BROW,))
    for k in range(16):
        vm.load("global_load_dwordx2 %s, v%d, s[86:87]" % (vp(cfg.V_B + 2 * k), cfg.V_OFF8))
        if k < 15:
            R("s_add_u32 s86, s86, 0x%x" % (2048 * cfg.ROW_G,))
            R("s_addc_u32 s87, s87, 0")
    em.lines.append(".Lnext_polynomial:")
    em.comment("b row of the polynomial after this one (this one again if it is the last: a harmless reload)")
    R("s_add_u32 s52, %s, %s" % (S_I, S_G))
    R("s_cmp_lt_u32 s52, %s" % S_COUNT)
    R("s_cselect_b32 s52, %s, 0" % S_STRIDE[0])
    R("s_cselect_b32 s53, %s, 0" % S_STRIDE[1])
    R("s_add_u32 s18, s18, s52")
    R("s_addc_u32 s19, s19, s53")
    R("s_mov_b64 s[100:101], %s" % (cfg.S_AROW,))
    ring.prime()

    def X0(b_):
        em.comment("X0: thread (q, t) slot 4*qq + j  ->  sub-group qq, thread t, slot q + 4*j")
        R("s_barrier")               # WAR: every wave is done reading the previous exchange
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
        for k in range(16):
            qq, j = k // per, k % per
            R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(b_ + 2 * k), (qq & 1) * cfg.SLAB_BYTES + j * 2048 * cfg.ROW_G))
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(b_ + 2 * k), AX, 2048 * k))
        R("s_waitcnt lgkmcnt(0)")

    def E1(b_):
        em.comment("E1")
        R("s_barrier")               # WAR against the previous exchange through this slab
        lds_write(em, cfg.V_L1W, b_, 2176)
        R("s_waitcnt lgkmcnt(0)")
        R("s_barrier")
        lds_read(em, cfg.V_L1R, b_, 136)
        R("s_waitcnt lgkmcnt(0)")

    def E2(b_):
        em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
        lds_write(em, cfg.V_L1R, b_, 136)
        lds_read(em, cfg.V_L2R, b_, 8)
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
                run_pairs(em, [gs_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((name, s_, g))

    fwd_one(cfg.V_B, "b")
    fwd_one(cfg.V_A, "a")
    em.comment("point-wise product (thread t of sub-group q holds words 16t..16t+15 of block q of both operands)")
    run_pairs(em, [pointwise(cfg.V_A + 2 * i, cfg.V_B + 2 * i, True, True) for i in range(16)])
    R("s_mov_b64 s[100:101], %s" % (cfg.S_BROW,))          # file B is free: the ring's side loads now fetch the next b
    inv_pass("I1")
    em.comment("E2'")
    lds_write(em, cfg.V_L2R, cfg.V_A, 8)
    lds_read(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I2")
    em.comment("E1'")
    lds_write(em, cfg.V_L1R, cfg.V_A, 136)
    R("s_waitcnt lgkmcnt(0)")
    R("s_barrier")
    lds_read(em, cfg.V_L1W, cfg.V_A, 2176)
    R("s_waitcnt lgkmcnt(0)")
    inv_pass("I3")
    em.comment("X0': thread (q, t) slot g + 4*j  ->  thread (g, t) slot 4*q + j, reader-major layout [slot][tid]")
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
    inv_pass("I0")
    assert ring.next == len(uses) and len(ring.free) == cfg.RING_SLOTS
    em.comment("stage 0 with n^-1 folded in; every finished pair is stored at once (the next b has landed long ago)")
    R("s_waitcnt vmcnt(0)")
    for h in range(8):
        run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8))])
        for k in (h, h + 8):
            R("s_add_u32 s86, s20, 0x%x" % (k * 2048 * cfg.ROW_G,))
            R("s_addc_u32 s87, s21, 0")
            R("global_store_dwordx2 v%d, %s, s[86:87]" % (cfg.V_OFF8, vp(cfg.V_A + 2 * k)))
    for lo in (16, 20):
        R("s_add_u32 s%d, s%d, %s" % (lo, lo, S_STRIDE[0]))
        R("s_addc_u32 s%d, s%d, %s" % (lo + 1, lo + 1, S_STRIDE[1]))
    R("s_add_u32 %s, %s, %s" % (S_I, S_I, S_G))
    R("s_cmp_lt_u32 %s, %s" % (S_I, S_COUNT))
    R("s_cbranch_scc1 .Lnext_polynomial")
    R("s_endpgm")
    return em
