"""64-bit rows of 1024 / 2048 words: ONE WAVE per 1024-word row (two per 2048), no workgroup barrier at 1024 -- the generated
twins of kernels_wave.hip k_row<Pol64, MODE, LB> (round 6).  The shape of the reference's own micro-benchmark
(tests/ntt_perfs.cpp:178: run<1024, 124, uint64_t>).

Lane mapping, LDS layouts and twiddle indices are those of tools/gen_row1024_u32_asm.py (read its build() first); the arithmetic
is the delta-form 62-bit arithmetic of arith.py on the "pair" register map (two interleaved butterflies, 15 resident twiddle
records, 168 VGPRs -> 3 waves per SIMD), and the product runs on INCOMPLETE transforms (incomplete.py, level 2): a row of 1024
words keeps stages 0 .. 7 -- the third pass of either transform disappears altogether -- and multiplies residues mod X^4 -+ zeta.

  pass 1 (uniform records tw[(1 << s) + g])            lane t holds x[t + W q], q < 16                     W = 16 LB lanes per row
  E1      through the row's LDS slab (row barrier when the row has two waves)
  pass 2 (tw[((16 + B) << s) + g], B = t >> lg LB)     16 words LB apart inside block B
  E2      wave-local 16-lane transposes
  pass 3 (tw[(256 << i) + G t + g])                    lane t holds words 16 t .. 16 t + 15: the last lg LB stages
  inverse: the mirror image over the same table (descending indices), n^-1 folded into the last stage.
kernarg: c a b psi mc | nm magic (ceil(2^32 / nm), 0 when nm = 1) | rows (64 bit)        grid: ceil(rows / (4 / waves per row))"""
from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, canon, final_bfly, pointwise
from .incomplete import base_mul
from .pipe import emit_consts, emit_mc_load
from .twiddles import ct_stage, gs_stage, tw_slot, twreg

V_LANE = 7
SLAB = 1088 * 8          # bytes of LDS per 1024 row words (padding of either exchange layout included)
ARGS_ROW = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("i32", 48), ("i32", 52)]


def build_row1k(LB=4, mode="polymul", level=2):
    """mode: polymul (c = INTT(NTT(a) (.) NTT(b)), transforms incomplete by `level` stages) | fwd (canonical NTT words) | inv"""
    assert LB in (4, 8) and mode in ("polymul", "fwd", "inv") and level in (0, 2)
    if mode != "polymul":
        level = 0
    W, LG = 16 * LB, LB.bit_length() - 1
    LOGN, NS3, WAVES = 8 + LG, LG, W // 64
    em = Emitter()
    vm = VmCounter(em)
    R, V, L = em.raw, em.valu, em.lines.append
    V_GOFF, V_A1, V_A2, V_A3, V_B_ = cfg.V_OFF8, cfg.V_L1W, cfg.V_L1R, cfg.V_L2R, cfg.V_BIDX
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic (s15 becomes a constant once magic has been used)
    R("s_load_dwordx2 s[92:93], s[0:1], 0x30")           # rows
    V("v_and_b32_e32 v%d, %d, v%d" % (V_LANE, W - 1, cfg.V_TID))               # t: lane of the row
    V("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_GOFF, V_LANE))
    V("v_readfirstlane_b32 s91, v%d" % cfg.V_TID)
    R("s_lshr_b32 s91, s91, %d" % (6 + (WAVES.bit_length() - 1)))            # row of the workgroup
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s94, s2, %d" % (2 - (WAVES.bit_length() - 1)))
    R("s_add_u32 s94, s94, s91")                         # row
    R("s_mov_b32 s95, 1")                                # store the result
    R("s_cmp_lt_u32 s94, s92")
    R("s_cbranch_scc1 .Llive")
    if LB == 4:
        R("s_endpgm")                                    # a surplus wave of the last workgroup (no workgroup barrier anywhere)
    else:
        R("s_sub_u32 s94, s92, 1")                       # a surplus row: walk through every barrier on the last row, store nothing
        R("s_mov_b32 s95, 0")
    L(".Llive:")
    R("s_mul_hi_u32 s3, s94, s15")
    R("s_mul_i32 s3, s3, s14")
    R("s_sub_u32 s3, s94, s3")                           # cm = row mod nm
    R("s_cmp_eq_u32 s14, 1")
    R("s_cselect_b32 s3, 0, s3")
    R("s_lshl_b32 s42, s3, %d" % (LOGN + 4))             # twiddles of the modulus: psi + cm * n * 16
    R("s_add_u32 s22, s10, s42")
    R("s_addc_u32 s23, s11, 0")
    R("s_lshr_b32 s43, s94, %d" % (32 - (LOGN + 3)))
    R("s_lshl_b32 s42, s94, %d" % (LOGN + 3))            # row * n * 8 bytes
    for base, dst in ((6, 16), (8, 18), (4, 20)):
        R("s_add_u32 s%d, s%d, s42" % (dst, base))
        R("s_addc_u32 s%d, s%d, s43" % (dst + 1, base + 1))
    emit_mc_load(em)                                     # (cm in s3, table in s[12:13]) -> s[56:83]

    def row_io(base, srow, store=False):
        """lane t <-> x[t + W q] in pair q (the immediate offset reaches 4095 bytes: the pointer steps every 4096)"""
        per = 4096 // (8 * W)
        R("s_mov_b64 s[86:87], %s" % srow)
        for q in range(16):
            if q and q % per == 0:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
            off = 8 * W * (q % per)
            if store:
                R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (V_GOFF, vp(base + 2 * q), off))
            else:
                vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(base + 2 * q), V_GOFF, off))

    if mode == "inv":     # NTT-form input: lane t holds words 16 t .. 16 t + 15
        V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
        for i in range(8):
            vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (cfg.V_A + 4 * i, cfg.V_A + 4 * i + 3, T(1, 0), cfg.S_AROW, 16 * i))
    else:
        row_io(cfg.V_A, cfg.S_AROW)
        if mode == "polymul":
            row_io(cfg.V_B, cfg.S_BROW)
    n_rows_loaded = vm.issued
    # LDS addresses of the row's slab: A1 = 8 t (+ 8 (W + LB) q), A2 = 8 ((W + LB) B + l) (+ 8 LB q [+ 8 (q >> (4 - lg LB))]), A3 = 136 t (+ 8 q)
    R("s_mul_i32 s90, s91, %d" % (SLAB * WAVES))
    V("v_add_u32_e32 v%d, s90, v%d" % (V_A1, V_GOFF))
    V("v_lshrrev_b32_e32 v%d, %d, v%d" % (V_B_, LG, V_LANE))                 # B
    V("v_and_b32_e32 v%d, %d, v%d" % (V_A2, LB - 1, V_LANE))                 # l
    V("v_mov_b32_e32 v%d, %d" % (V_A3, W + LB))
    V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_A2, V_B_, V_A3, V_A2))         # (W + LB) B + l
    V("v_lshlrev_b32_e32 v%d, 3, v%d" % (V_A2, V_A2))
    V("v_add_u32_e32 v%d, s90, v%d" % (V_A2, V_A2))
    V("v_mov_b32_e32 v%d, 136" % V_A3)
    V("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_A3, V_LANE, V_A3))              # 136 t
    V("v_add_u32_e32 v%d, s90, v%d" % (V_A3, V_A3))
    for t0 in sorted(set(cfg.V_T)):
        V("v_mov_b32_e32 v%d, 0" % (t0 + 15,))                               # the persistent zero of ZP
    emit_consts(em)                                      # waits for the ModConst record; s15 = 0xc0000000 from here on

    # ---------------------------------------------------------------- twiddle records: 15 slots, group s = slots of sub-stage s
    def load_uniform(first_of):
        def f(s, slot):
            seq = 0
            for g in range(1 << s):
                r = cfg.V_TW + 4 * slot(s, g)
                seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (r, r + 3, cfg.V_ZERO, 16 * first_of(s, g)))
            return seq
        return f

    def load_lane(index_expr, nrec_of, rec_of_group):
        """per-lane records: index_expr(s) leaves 16 x (index of the lane's first record) in V_TWO; record j of the block serves group
        rec_of_group^-1 (ascending passes: j = g; descending: j = n - 1 - g)"""
        def f(s, slot):
            index_expr(s)
            seq = 0
            n = nrec_of(s)
            for j in range(n):
                r = cfg.V_TW + 4 * slot(s, rec_of_group(n, j))
                seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (r, r + 3, cfg.V_TWO, 16 * j))
            return seq
        return f

    def idx_pass2(s):      # 16 ((16 + B) << s)
        V("v_add_u32_e32 v%d, 16, v%d" % (cfg.V_TWO, V_B_))
        V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, cfg.V_TWO))

    def idx_inv2(s):       # 16 ((31 - B) << s)
        V("v_sub_u32_e32 v%d, 31, v%d" % (cfg.V_TWO, V_B_))
        V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, cfg.V_TWO))

    s3_first = 4 - NS3                                    # pass 3 = sub-stages s3_first .. 3 of the 16-point structure

    def idx_pass3(s):      # 16 ((256 << i) + G t), i = s - s3_first, G = 2^s groups
        i = s - s3_first
        V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s, V_LANE))
        V("v_add_u32_e32 v%d, 0x%x, v%d" % (cfg.V_TWO, 256 << i, cfg.V_TWO))
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (cfg.V_TWO, cfg.V_TWO))

    def idx_inv1(s):       # 16 ((512 << i) - G (t + 1))
        i = s - s3_first
        V("v_add_u32_e32 v%d, 1, v%d" % (cfg.V_TWO, V_LANE))
        V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s, cfg.V_TWO))
        V("v_sub_u32_e32 v%d, 0x%x, v%d" % (cfg.V_TWO, 512 << i, cfg.V_TWO))
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (cfg.V_TWO, cfg.V_TWO))

    def idx_zeta(s):       # level 2, rows of 1024 words: zeta = -+ tw[128 + 2 t + g] (the last retained stage is pass 2's)
        V("v_lshlrev_b32_e32 v%d, 1, v%d" % (cfg.V_TWO, V_LANE))
        V("v_add_u32_e32 v%d, 0x%x, v%d" % (cfg.V_TWO, 1 << (LOGN - 3), cfg.V_TWO))
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (cfg.V_TWO, cfg.V_TWO))

    asc, desc = (lambda n, j: j), (lambda n, j: n - 1 - j)
    LOAD = {"F1": load_uniform(lambda s, g: (1 << s) + g), "I3": load_uniform(lambda s, g: (2 << s) - 1 - g),
            "F2": load_lane(idx_pass2, lambda s: 1 << s, asc), "I2": load_lane(idx_inv2, lambda s: 1 << s, desc),
            "F3": load_lane(idx_pass3, lambda s: 1 << s, asc), "I1": load_lane(idx_inv1, lambda s: 1 << s, desc),
            "BM": load_lane(idx_zeta, lambda s: 2, asc)}

    # ---------------------------------------------------------------- exchanges
    def row_sync():
        if WAVES > 1:
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")

    def exchange(bases, waddr, woff, raddr, roff, sync_between=False, sync_before=False):
        for n_, b in enumerate(bases):
            if sync_before or (n_ and sync_between):
                row_sync()
            for q in range(16):
                R("ds_write_b64 v%d, %s offset:%d" % (waddr, vp(b + 2 * q), woff(q)))
            if sync_between:
                row_sync()
            for q in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(b + 2 * q), raddr, roff(q)))
            R("s_waitcnt lgkmcnt(0)")

    e1_row = lambda q: 8 * (W + LB) * q
    e1_blk = lambda q: 8 * LB * q
    e2_blk = lambda q: 8 * (LB * q + (q >> (4 - LG)))
    e2_thr = lambda q: 8 * q
    both = [cfg.V_A, cfg.V_B] if mode == "polymul" else [cfg.V_A]

    # ---------------------------------------------------------------- the schedule: passes in order, each a list of sub-stages
    keep3 = NS3 - level                                   # sub-stages of pass 3 that remain (level 2: 0 at 1024, 1 at 2048)
    f3 = list(range(s3_first, s3_first + keep3))
    i1 = f3[::-1]
    G = 4
    plan = []
    if mode != "inv":
        plan += [("F1", [0, 1, 2, 3]), ("F2", [0, 1, 2, 3]), ("F3", f3)]
    if level:
        plan += [("BM", [1])]                             # zeta records sit in slot group 1 (what base_mul's caller expects)
    if mode != "fwd":
        plan += [("I1", i1), ("I2", [3, 2, 1, 0]), ("I3", [3, 2, 1])]
    zeta_from_f3 = bool(level and f3)                     # 2048: pass 3's retained stage IS the last retained stage
    # slot group of (pass, sub-stage): its own, except (a) the zeta records live in group 1, (b) I1's records of a retained sub-stage
    # whose group is still held by zeta go one group up (free: the dropped sub-stages' slots)
    def grp(name, s):
        if name == "I1" and zeta_from_f3 and s == f3[-1]:
            return s + 1
        return s
    slot_of = lambda name, s: (lambda s_, g: tw_slot(grp(name, s_), g))
    order = [(name, s) for name, stages in plan for s in stages]
    if zeta_from_f3:
        assert f3[-1] == 1
    seq, loaded = {}, set()

    def load(name, s):
        if (name, s) in loaded:
            return
        loaded.add((name, s))
        if name == "BM" and zeta_from_f3:
            seq[(name, s)] = seq[("F3", f3[-1])]          # the records are already there
            return
        seq[(name, s)] = LOAD[name](s, slot_of(name, s))

    state = {"bm_done": not level, "deferred": []}

    def release(pos):
        """slot group of order[pos] is free: request the records of its next user -- except that the top slots (group 3) are the
        base multiplication's scratch until it is done: an inverse pass's records wait for that"""
        name, s = order[pos]
        g = grp(name, s)
        for name2, s2 in order[pos + 1:]:
            if grp(name2, s2) == g and (name2, s2) not in loaded:
                if g == 3 and name2[0] == "I" and not state["bm_done"]:
                    state["deferred"].append((name2, s2))
                else:
                    load(name2, s2)
                return

    # the first user of every slot group
    for g in range(4):
        for name, s in order:
            if grp(name, s) == g:
                load(name, s)
                break
    if mode == "inv":
        R("s_waitcnt vmcnt(%d)" % (vm.issued - n_rows_loaded))    # the row has landed (pass I1 works on the lane's 16 consecutive words)
    rtmp_top = cfg.V_TW + 60
    negtmp = [rtmp_top - 4, rtmp_top - 8]
    rtmp = [rtmp_top - 8 - 2 * (G - 1), rtmp_top - 8 - 4 * (G - 1)]

    pos_of = {ns: k for k, ns in enumerate(order)}
    stages_of = dict(plan)

    def run_pass(name):
        if not stages_of.get(name):
            return
        em.comment(name)
        for s in stages_of[name]:
            vm.wait(seq[(name, s)])
            if name[0] == "F":
                ct_stage(em, both, s, slot=slot_of(name, s))
            else:
                gs_stage(em, cfg.V_A, s, slot=slot_of(name, s))
            if not (zeta_from_f3 and name == "F3" and s == f3[-1]):      # (zeta: released after the base multiplication)
                release(pos_of[(name, s)])

    if mode != "inv":
        R("s_waitcnt vmcnt(%d)" % (vm.issued - n_rows_loaded))             # operands landed (the twiddle prefetch may still fly)
        run_pass("F1")
        exchange(both, V_A1, e1_row, V_A2, e1_blk, sync_between=True)
        run_pass("F2")
        exchange(both, V_A2, e2_blk, V_A3, e2_thr, sync_before=True)
        run_pass("F3")
    if mode == "fwd":
        em.comment("canonical words 16 t .. 16 t + 15: eight 16-byte stores per lane")
        run_pairs(em, [canon(cfg.V_A + 2 * q) for q in range(16)])
        V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
        if WAVES > 1:
            R("s_cmp_eq_u32 s95, 0")
            R("s_cbranch_scc1 .Ldone")
        for i_ in range(8):
            R("global_store_dwordx4 v%d, v[%d:%d], %s offset:%d" % (T(1, 0), cfg.V_A + 4 * i_, cfg.V_A + 4 * i_ + 3, cfg.S_CROW, 16 * i_))
        L(".Ldone:")
        R("s_endpgm")
        return em
    if mode == "polymul":
        if level:
            em.comment("base multiplication mod X^4 -+ zeta (lane t holds words 16 t .. 16 t + 15 of both operands: four residues)")
            load("BM", 1)                 # (2048-word rows: the records are pass 3's retained stage's, already resident)
            vm.wait(seq[("BM", 1)])
            run_pairs(em, [base_mul(cfg.V_A + 2 * G * g, cfg.V_B + 2 * G * g, G, twreg(tw_slot(1, g // 2)), bool(g & 1), rtmp, negtmp)
                           for g in range(16 // G)])
            state["bm_done"] = True
            for name2, s2 in state["deferred"]:
                load(name2, s2)
            release(pos_of[("BM", 1)])    # (2048-word rows: this also frees pass 3's retained records -- they ARE zeta)
        else:
            em.comment("point-wise product")
            run_pairs(em, [pointwise(cfg.V_A + 2 * q, cfg.V_B + 2 * q, True, True) for q in range(16)])
    one = [cfg.V_A]
    run_pass("I1")
    exchange(one, V_A3, e2_thr, V_A2, e2_blk)
    run_pass("I2")
    exchange(one, V_A2, e1_blk, V_A1, e1_row, sync_between=True, sync_before=True)
    run_pass("I3")
    em.comment("stage 0 with n^-1 folded in")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    if WAVES > 1:
        R("s_cmp_eq_u32 s95, 0")
        R("s_cbranch_scc1 .Ldone")
    row_io(cfg.V_A, cfg.S_CROW, store=True)
    L(".Ldone:")
    R("s_endpgm")
    return em
