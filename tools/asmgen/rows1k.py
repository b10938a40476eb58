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
from .fused import fma_job, fms_job
from .pipe import emit_consts, emit_mc_load
from .twiddles import ct_stage, gs_stage, tw_slot, twreg

V_LANE = 7
SLAB = 1088 * 8          # bytes of LDS per 1024 row words (padding of either exchange layout included)
ARGS_ROW = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("i32", 48), ("i32", 52)]


class RowKernel:
    """what every wave-per-row kernel shares: lane constants, LDS addresses of the row's slab, the exchanges, the loaders of the
    twiddle records and the scheduler that requests a slot group's next records as soon as its current user is done"""

    def __init__(self, LB):
        self.LB, self.W, self.LG = LB, 16 * LB, LB.bit_length() - 1
        self.LOGN, self.NS3, self.WAVES = 8 + self.LG, self.LG, self.W // 64
        self.em = Emitter()
        self.vm = VmCounter(self.em)
        self.V_GOFF, self.V_A1, self.V_A2, self.V_A3, self.V_B_ = cfg.V_OFF8, cfg.V_L1W, cfg.V_L1R, cfg.V_L2R, cfg.V_BIDX
        self.s3_first = 4 - self.NS3                      # pass 3 = sub-stages s3_first .. 3 of the 16-point structure
        self.plan, self.order, self.seq, self.loaded = [], [], {}, set()
        self.hold3 = None                                 # (label, s): group 3 (the top slots) is that step's scratch -- records of LATER steps wait for unhold3()
        self.deferred = []
        self.group_of = {}                                # (label, s) -> slot group, where it is not s

    # ------------------------------------------------------------ prologue pieces
    def lane_and_row(self, rows_sgpr="s92"):
        """V_LANE, V_GOFF; s91 = row of the workgroup, s94 = row (clamped: s95 = 0 then), s3 = cm, s93' = el (row / nm, in s43 on
        return -- callers that need it copy it at once)"""
        R, V, L = self.em.raw, self.em.valu, self.em.lines.append
        W, WAVES, LB = self.W, self.WAVES, self.LB
        V("v_and_b32_e32 v%d, %d, v%d" % (V_LANE, W - 1, cfg.V_TID))               # t: lane of the row
        V("v_lshlrev_b32_e32 v%d, 3, v%d" % (self.V_GOFF, V_LANE))
        V("v_readfirstlane_b32 s91, v%d" % cfg.V_TID)
        R("s_lshr_b32 s91, s91, %d" % (6 + (WAVES.bit_length() - 1)))            # row of the workgroup
        R("s_waitcnt lgkmcnt(0)")
        R("s_lshl_b32 s94, s2, %d" % (2 - (WAVES.bit_length() - 1)))
        R("s_add_u32 s94, s94, s91")                         # row
        R("s_mov_b32 s95, 1")                                # store the result
        R("s_cmp_lt_u32 s94, %s" % rows_sgpr)
        R("s_cbranch_scc1 .Llive")
        if LB == 4:
            R("s_endpgm")                                    # a surplus wave of the last workgroup (no workgroup barrier anywhere)
        else:
            R("s_sub_u32 s94, %s, 1" % rows_sgpr)            # a surplus row: walk through every barrier on the last row, store nothing
            R("s_mov_b32 s95, 0")
        L(".Llive:")
        R("s_mul_hi_u32 s43, s94, s15")                      # el = row / nm (magic = ceil(2^32 / nm); 0 when nm = 1)
        R("s_mul_i32 s3, s43, s14")
        R("s_sub_u32 s3, s94, s3")                           # cm = row mod nm
        R("s_cmp_eq_u32 s14, 1")
        R("s_cselect_b32 s3, 0, s3")
        R("s_cselect_b32 s43, s94, s43")
        R("s_lshl_b32 s42, s3, %d" % (self.LOGN + 4))        # twiddles of the modulus: psi + cm * n * 16
        R("s_add_u32 s22, s10, s42")
        R("s_addc_u32 s23, s11, 0")

    def add_index(self, dst, base, idx, shift):
        """s[dst:dst+1] = s[base:base+1] + (idx << shift)   (idx: an SGPR holding a 32-bit count)"""
        R = self.em.raw
        R("s_lshr_b32 s87, %s, %d" % (idx, 32 - shift))
        R("s_lshl_b32 s86, %s, %d" % (idx, shift))
        R("s_add_u32 s%d, s%d, s86" % (dst, base))
        R("s_addc_u32 s%d, s%d, s87" % (dst + 1, base + 1))

    def lds_addresses(self):
        """A1 = 8 t (+ 8 (W + LB) q), A2 = 8 ((W + LB) B + l) (+ 8 LB q [+ 8 (q >> (4 - lg LB))]), A3 = 136 t (+ 8 q); s90 = the row's slab"""
        R, V = self.em.raw, self.em.valu
        W, LB, LG = self.W, self.LB, self.LG
        R("s_mul_i32 s90, s91, %d" % (SLAB * self.WAVES))
        V("v_add_u32_e32 v%d, s90, v%d" % (self.V_A1, self.V_GOFF))
        V("v_lshrrev_b32_e32 v%d, %d, v%d" % (self.V_B_, LG, V_LANE))                 # B
        V("v_and_b32_e32 v%d, %d, v%d" % (self.V_A2, LB - 1, V_LANE))                 # l
        V("v_mov_b32_e32 v%d, %d" % (self.V_A3, W + LB))
        V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (self.V_A2, self.V_B_, self.V_A3, self.V_A2))   # (W + LB) B + l
        V("v_lshlrev_b32_e32 v%d, 3, v%d" % (self.V_A2, self.V_A2))
        V("v_add_u32_e32 v%d, s90, v%d" % (self.V_A2, self.V_A2))
        V("v_mov_b32_e32 v%d, 136" % self.V_A3)
        V("v_mul_u32_u24_e32 v%d, v%d, v%d" % (self.V_A3, V_LANE, self.V_A3))         # 136 t
        V("v_add_u32_e32 v%d, s90, v%d" % (self.V_A3, self.V_A3))
        for t0 in sorted(set(cfg.V_T)):
            V("v_mov_b32_e32 v%d, 0" % (t0 + 15,))                                    # the persistent zero of ZP

    def row_io(self, base, srow, store=False):
        """lane t <-> x[t + W q] in pair q (the immediate offset reaches 4095 bytes: the pointer steps every 4096)"""
        R, W = self.em.raw, self.W
        per = 4096 // (8 * W)
        R("s_mov_b64 s[86:87], %s" % srow)
        for q in range(16):
            if q and q % per == 0:
                R("s_add_u32 s86, s86, 0x1000")
                R("s_addc_u32 s87, s87, 0")
            off = 8 * W * (q % per)
            if store:
                R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d" % (self.V_GOFF, vp(base + 2 * q), off))
            else:
                self.vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d" % (vp(base + 2 * q), self.V_GOFF, off))

    def lane16_io(self, base, srow, store=False, addr=None):
        """lane t <-> words 16 t .. 16 t + 15 (NTT-form layout): eight 16-byte accesses; addr: VGPR with 128 t"""
        A = T(1, 0) if addr is None else addr
        if addr is None:
            self.em.valu("v_lshlrev_b32_e32 v%d, 7, v%d" % (A, V_LANE))
        seq = 0
        for i in range(8):
            if store:
                self.em.raw("global_store_dwordx4 v%d, v[%d:%d], %s offset:%d" % (A, base + 4 * i, base + 4 * i + 3, srow, 16 * i))
            else:
                seq = self.vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (base + 4 * i, base + 4 * i + 3, A, srow, 16 * i))
        return seq

    # ------------------------------------------------------------ exchanges
    def row_sync(self):
        if self.WAVES > 1:
            self.em.raw("s_waitcnt lgkmcnt(0)")
            self.em.raw("s_barrier")

    def exchange(self, bases, which, sync_between=False, sync_before=False):
        """which: "E1" row -> block, "E2" block -> thread, "E2i" thread -> block, "E1i" block -> row"""
        R, W, LB, LG = self.em.raw, self.W, self.LB, self.LG
        e1_row = lambda q: 8 * (W + LB) * q
        e1_blk = lambda q: 8 * LB * q
        e2_blk = lambda q: 8 * (LB * q + (q >> (4 - LG)))
        e2_thr = lambda q: 8 * q
        waddr, woff, raddr, roff = {"E1": (self.V_A1, e1_row, self.V_A2, e1_blk), "E2": (self.V_A2, e2_blk, self.V_A3, e2_thr),
                                    "E2i": (self.V_A3, e2_thr, self.V_A2, e2_blk), "E1i": (self.V_A2, e1_blk, self.V_A1, e1_row)}[which]
        for n_, b in enumerate(bases):
            if sync_before or (n_ and sync_between):
                self.row_sync()
            for q in range(16):
                R("ds_write_b64 v%d, %s offset:%d" % (waddr, vp(b + 2 * q), woff(q)))
            if sync_between:
                self.row_sync()
            for q in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(b + 2 * q), raddr, roff(q)))
            R("s_waitcnt lgkmcnt(0)")

    # ------------------------------------------------------------ twiddle records: 15 slots, group s = slots of sub-stage s
    def _load_uniform(self, first_of):
        def f(s, slot):
            seq = 0
            for g in range(1 << s):
                r = cfg.V_TW + 4 * slot(s, g)
                seq = self.vm.load("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (r, r + 3, cfg.V_ZERO, 16 * first_of(s, g)))
            return seq
        return f

    def _load_lane(self, index_expr, nrec_of, rec_of_group):
        """per-lane records: index_expr(s) leaves 16 x (index of the lane's first record) in V_TWO; record j of the block serves group
        rec_of_group(n, j) (ascending passes: j; descending: n - 1 - j)"""
        def f(s, slot):
            index_expr(s)
            seq = 0
            n = nrec_of(s)
            for j in range(n):
                r = cfg.V_TW + 4 * slot(s, rec_of_group(n, j))
                seq = self.vm.load("global_load_dwordx4 v[%d:%d], v%d, s[22:23] offset:%d" % (r, r + 3, cfg.V_TWO, 16 * j))
            return seq
        return f

    def loaders(self):
        V, V_B_, s3_first, LOGN = self.em.valu, self.V_B_, self.s3_first, self.LOGN

        def idx_pass2(s):      # 16 ((16 + B) << s)
            V("v_add_u32_e32 v%d, 16, v%d" % (cfg.V_TWO, V_B_))
            V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, cfg.V_TWO))

        def idx_inv2(s):       # 16 ((31 - B) << s)
            V("v_sub_u32_e32 v%d, 31, v%d" % (cfg.V_TWO, V_B_))
            V("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, cfg.V_TWO))

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
        return {"F1": self._load_uniform(lambda s, g: (1 << s) + g), "I3": self._load_uniform(lambda s, g: (2 << s) - 1 - g),
                "F2": self._load_lane(idx_pass2, lambda s: 1 << s, asc), "I2": self._load_lane(idx_inv2, lambda s: 1 << s, desc),
                "F3": self._load_lane(idx_pass3, lambda s: 1 << s, asc), "I1": self._load_lane(idx_inv1, lambda s: 1 << s, desc),
                "BM": self._load_lane(idx_zeta, lambda s: 2, asc)}

    # ------------------------------------------------------------ the schedule
    def set_plan(self, plan, custom=None):
        """plan: [(label, kind, [sub-stages in execution order])]; kind names a loader (F1 F2 F3 I1 I2 I3 BM) or a key of `custom`
        ({kind: f(s, slot) -> sequence number})"""
        self.plan = plan
        self.kind_of = {label: kind for label, kind, _ in plan}
        self.stages_of = {label: stages for label, _, stages in plan}
        self.order = [(label, s) for label, _, stages in plan for s in stages]
        self.pos_of = {ls: k for k, ls in enumerate(self.order)}
        self.LOAD = self.loaders()
        self.LOAD.update(custom or {})

    def grp(self, label, s):
        return self.group_of.get((label, s), s)

    def slot_of(self, label):
        return lambda s_, g: tw_slot(self.grp(label, s_), g)

    def load(self, label, s):
        if (label, s) in self.loaded:
            return
        self.loaded.add((label, s))
        self.seq[(label, s)] = self.LOAD[self.kind_of[label]](s, self.slot_of(label))

    def alias(self, label, s, other):
        """(label, s) uses records that are already resident: those of `other`"""
        self.loaded.add((label, s))
        self.seq[(label, s)] = self.seq[other]

    def release(self, label, s):
        """the slot group of (label, s) is free: request the records of its next user -- unless the top slots (group 3) are held
        as scratch: then that request waits for unhold3()"""
        g = self.grp(label, s)
        for label2, s2 in self.order[self.pos_of[(label, s)] + 1:]:
            if self.grp(label2, s2) == g and (label2, s2) not in self.loaded:
                if g == 3 and self.hold3 is not None and self.pos_of[(label2, s2)] > self.pos_of[self.hold3]:
                    self.deferred.append((label2, s2))
                else:
                    self.load(label2, s2)
                return

    def unhold3(self):
        self.hold3 = None
        for label2, s2 in self.deferred:
            self.load(label2, s2)
        self.deferred = []

    def prime(self):
        """the first user of every slot group"""
        for g in range(4):
            for label, s in self.order:
                if self.grp(label, s) == g:
                    if g == 3 and self.hold3 is not None and self.pos_of[(label, s)] > self.pos_of[self.hold3]:
                        self.deferred.append((label, s))
                    else:
                        self.load(label, s)
                    break

    def run_pass(self, label, bases, keep=()):
        """the butterfly stages of one pass over the register files `bases`; keep: sub-stages whose records stay (released by the caller)"""
        if not self.stages_of.get(label):
            return
        self.em.comment(label)
        for s in self.stages_of[label]:
            self.vm.wait(self.seq[(label, s)])
            if self.kind_of[label][0] == "F":
                ct_stage(self.em, bases, s, slot=self.slot_of(label))
            else:
                gs_stage(self.em, bases[0], s, slot=self.slot_of(label))
            if s not in keep:
                self.release(label, s)


def build_row1k(LB=4, mode="polymul", level=2):
    """mode: polymul (c = INTT(NTT(a) (.) NTT(b)), transforms incomplete by `level` stages) | fwd (canonical NTT words) | inv"""
    assert LB in (4, 8) and mode in ("polymul", "fwd", "inv") and level in (0, 2)
    if mode != "polymul":
        level = 0
    K = RowKernel(LB)
    em, vm = K.em, K.vm
    R, V, L = em.raw, em.valu, em.lines.append
    NS3, WAVES = K.NS3, K.WAVES
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic (s15 becomes a constant once magic has been used)
    R("s_load_dwordx2 s[92:93], s[0:1], 0x30")           # rows
    K.lane_and_row()
    for base, dst in ((6, 16), (8, 18), (4, 20)):        # row * n * 8 bytes into a, b, c
        K.add_index(dst, base, "s94", K.LOGN + 3)
    emit_mc_load(em)                                     # (cm in s3, table in s[12:13]) -> s[56:83]
    if mode == "inv":     # NTT-form input: lane t holds words 16 t .. 16 t + 15
        K.lane16_io(cfg.V_A, cfg.S_AROW)
    else:
        K.row_io(cfg.V_A, cfg.S_AROW)
        if mode == "polymul":
            K.row_io(cfg.V_B, cfg.S_BROW)
    n_rows_loaded = vm.issued
    K.lds_addresses()
    emit_consts(em)                                      # waits for the ModConst record; s15 = 0xc0000000 from here on
    both = [cfg.V_A, cfg.V_B] if mode == "polymul" else [cfg.V_A]

    # ---------------------------------------------------------------- the schedule: passes in order, each a list of sub-stages
    keep3 = NS3 - level                                   # sub-stages of pass 3 that remain (level 2: 0 at 1024, 1 at 2048)
    f3 = list(range(K.s3_first, K.s3_first + keep3))
    i1 = f3[::-1]
    G = 4
    plan = []
    if mode != "inv":
        plan += [("F1", "F1", [0, 1, 2, 3]), ("F2", "F2", [0, 1, 2, 3]), ("F3", "F3", f3)]
    if level:
        plan += [("BM", "BM", [1])]                       # zeta records sit in slot group 1 (what base_mul's caller expects)
    if mode != "fwd":
        plan += [("I1", "I1", i1), ("I2", "I2", [3, 2, 1, 0]), ("I3", "I3", [3, 2, 1])]
    zeta_from_f3 = bool(level and f3)                     # 2048: pass 3's retained stage IS the last retained stage
    K.set_plan(plan)
    if zeta_from_f3:
        # I1's records of the retained sub-stage go one group up (free: the dropped sub-stages' slots) while zeta holds group 1
        assert f3[-1] == 1
        K.group_of[("I1", 1)] = 2
    K.hold3 = ("BM", 1) if level else None                # the top slots are the base multiplication's scratch until it is done
    K.prime()
    rtmp_top = cfg.V_TW + 60
    negtmp = [rtmp_top - 4, rtmp_top - 8]
    rtmp = [rtmp_top - 8 - 2 * (G - 1), rtmp_top - 8 - 4 * (G - 1)]

    R("s_waitcnt vmcnt(%d)" % (vm.issued - n_rows_loaded))                 # operands landed (the twiddle prefetch may still fly)
    if mode != "inv":
        K.run_pass("F1", both)
        K.exchange(both, "E1", sync_between=True)
        K.run_pass("F2", both)
        K.exchange(both, "E2", sync_before=True)
        K.run_pass("F3", both, keep=(f3[-1],) if zeta_from_f3 else ())
    if mode == "fwd":
        em.comment("canonical words 16 t .. 16 t + 15: eight 16-byte stores per lane")
        run_pairs(em, [canon(cfg.V_A + 2 * q) for q in range(16)])
        V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
        if WAVES > 1:
            R("s_cmp_eq_u32 s95, 0")
            R("s_cbranch_scc1 .Ldone")
        K.lane16_io(cfg.V_A, cfg.S_CROW, store=True, addr=T(1, 0))
        L(".Ldone:")
        R("s_endpgm")
        return em
    if mode == "polymul":
        if level:
            em.comment("base multiplication mod X^4 -+ zeta (lane t holds words 16 t .. 16 t + 15 of both operands: four residues)")
            if zeta_from_f3:
                K.alias("BM", 1, ("F3", f3[-1]))          # (2048-word rows: the records are pass 3's retained stage's, already resident)
            vm.wait(K.seq[("BM", 1)])
            run_pairs(em, [base_mul(cfg.V_A + 2 * G * g, cfg.V_B + 2 * G * g, G, twreg(tw_slot(1, g // 2)), bool(g & 1), rtmp, negtmp)
                           for g in range(16 // G)])
            K.unhold3()
            K.release("BM", 1)            # (2048-word rows: this also frees pass 3's retained records -- they ARE zeta)
        else:
            em.comment("point-wise product")
            run_pairs(em, [pointwise(cfg.V_A + 2 * q, cfg.V_B + 2 * q, True, True) for q in range(16)])
    one = [cfg.V_A]
    K.run_pass("I1", one)
    K.exchange(one, "E2i")
    K.run_pass("I2", one)
    K.exchange(one, "E1i", sync_between=True, sync_before=True)
    K.run_pass("I3", one)
    em.comment("stage 0 with n^-1 folded in")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    if WAVES > 1:
        R("s_cmp_eq_u32 s95, 0")
        R("s_cbranch_scc1 .Ldone")
    K.row_io(cfg.V_A, cfg.S_CROW, store=True)
    L(".Ldone:")
    R("s_endpgm")
    return em


# ------------------------------------------------------------------ transform-fused pipelines on rows of 1024 / 2048 words
# What callers of the reference run around the transforms (tests/nfllib_demo_main_op.cpp:26-58), one wave(s) per row -- the generated
# twins of kernels_wave.hip k_row_fwd_fma / k_row_fma_inv:
#   fwd_fma   out0 = NTT(x) k0 + NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)]      x transformed once and kept in registers; each noise row
#             transformed in the same wave(s); the key row (NTT form, words 16 t .. 16 t + 15 per lane) lands in the top twiddle
#             slots once pass 3 has released them, and the canonical result leaves from there
#   fma_inv   c = INTT(b -+ a k)                                               the point-wise step in the loaded layout, then the inverse
# Operands: x / e of ONE format -- "w" residue words [nm][n] or "i8" one signed byte per coefficient shared by the moduli (v < 0
# stands for p + v) -- every stride 0 or 1 (in polynomials); keys words in NTT form, stride 0 (one polynomial) or 1.
ARGS_ROW_FWD = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("ptr", 48), ("ptr", 56),
                ("ptr", 64), ("ptr", 72), ("i32", 80), ("i32", 84), ("i32", 88), ("i32", 92), ("i32", 96), ("i32", 100), ("i32", 104), ("i32", 108)]
ARGS_ROW_INV = ARGS_ROW + [("ptr", 56), ("i32", 64), ("i32", 68)]
KEY = lambda: cfg.V_TW + 28          # slot group 3 (slots 7 .. 14): 32 registers = the key row's 16 words


def build_row1k_fma_inv(LB=4, subtract=True):
    """kernarg: c a b psi mc | nm magic | rows | key kstride"""
    K = RowKernel(LB)
    em, vm = K.em, K.vm
    R, V, L = em.raw, em.valu, em.lines.append
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic
    R("s_load_dwordx2 s[92:93], s[0:1], 0x30")           # rows
    R("s_load_dwordx2 s[84:85], s[0:1], 0x38")           # key
    R("s_load_dword s96, s[0:1], 0x40")                  # its stride: 0 = one polynomial for the batch, 1 = dense
    K.lane_and_row()
    R("s_cmp_eq_u32 s96, 0")
    R("s_cselect_b32 s97, s3, s94")                      # key row: cm or row
    for base, dst in ((6, 16), (8, 18), (4, 20)):
        K.add_index(dst, base, "s94", K.LOGN + 3)
    K.add_index(84, 84, "s97", K.LOGN + 3)
    emit_mc_load(em)
    V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
    seq_k = K.lane16_io(KEY(), "s[84:85]", addr=T(1, 0))
    seq_a = K.lane16_io(cfg.V_A, cfg.S_AROW, addr=T(1, 0))
    seq_b = K.lane16_io(cfg.V_B, cfg.S_BROW, addr=T(1, 0))
    K.lds_addresses()
    emit_consts(em)
    i1 = list(range(K.s3_first, 4))[::-1]
    K.set_plan([("PW", "PW", [3]), ("I1", "I1", i1), ("I2", "I2", [3, 2, 1, 0]), ("I3", "I3", [3, 2, 1])],
               custom={"PW": lambda s, slot: seq_k})
    K.hold3 = ("PW", 3)                                  # the key row sits in the top slots until the point-wise step is done
    K.prime()
    em.comment("b -+ a * k in the loaded layout (lane t: words 16 t .. 16 t + 15)")
    vm.wait(seq_b)
    run_pairs(em, [fms_job(cfg.V_A + 2 * q, KEY() + 2 * q, cfg.V_B + 2 * q, subtract) for q in range(16)])
    K.unhold3()
    K.release("PW", 3)
    one = [cfg.V_A]
    K.run_pass("I1", one)
    K.exchange(one, "E2i")
    K.run_pass("I2", one)
    K.exchange(one, "E1i", sync_between=True, sync_before=True)
    K.run_pass("I3", one)
    em.comment("stage 0 with n^-1 folded in")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 8)) for h in range(8)])
    if K.WAVES > 1:
        R("s_cmp_eq_u32 s95, 0")
        R("s_cbranch_scc1 .Ldone")
    K.row_io(cfg.V_A, cfg.S_CROW, store=True)
    L(".Ldone:")
    R("s_endpgm")
    return em


def build_row1k_fwd_fma(LB=4, two=True, fmt="i8"):
    """kernarg: out0 out1 x psi mc | nm magic | k0 e0 k1 e1 | rows (64 bit) | strides x k0 e0 k1 e1 (in polynomials: 0 or 1)"""
    assert fmt in ("w", "i8")
    K = RowKernel(LB)
    em, vm = K.em, K.vm
    R, V, L = em.raw, em.valu, em.lines.append
    W, LOGN = K.W, K.LOGN
    R("s_load_dwordx4 s[4:7], s[0:1], 0x0")              # out0, out1
    R("s_load_dwordx8 s[8:15], s[0:1], 0x10")            # x, psi, mc, nm, magic
    R("s_load_dwordx4 s[16:19], s[0:1], 0x30")           # k0, e0
    R("s_load_dwordx2 s[20:21], s[0:1], 0x40")           # k1
    R("s_load_dwordx2 s[84:85], s[0:1], 0x48")           # e1
    R("s_load_dwordx2 s[92:93], s[0:1], 0x50")           # rows
    R("s_load_dwordx4 s[96:99], s[0:1], 0x58")           # strides: x, k0, e0, k1
    R("s_load_dword s100, s[0:1], 0x68")                 # ... e1
    K.lane_and_row()
    R("s_mov_b32 s101, s43")                             # el = row / nm: the batch element
    def operand_row(pair, stride, words):
        """s[pair] += ((stride * el) * (words ? nm : 1) + (words ? cm : 0)) << log2(bytes per row)"""
        R("s_mul_i32 s42, %s, s101" % stride)
        if words:
            R("s_mul_i32 s42, s42, s14")
            R("s_add_u32 s42, s42, s3")
        K.add_index(pair, pair, "s42", LOGN + (3 if words else 0))
    operand_row(8, "s96", fmt == "w")
    operand_row(18, "s98", fmt == "w")
    operand_row(16, "s97", True)
    if two:
        operand_row(84, "s100", fmt == "w")
        operand_row(20, "s99", True)
        K.add_index(6, 6, "s94", LOGN + 3)
    K.add_index(4, 4, "s94", LOGN + 3)
    emit_mc_load(em)
    K.lds_addresses()

    STG = T(1, 4)         # four staging registers of a compact row (stream 1's temporaries: idle outside the butterflies)

    def request(base, pair):
        """start fetching an operand row: words go straight to their register file (x[t + W q] -> pair q); a compact row's
        16 bytes per lane wait in the first four registers of the (idle) file"""
        if fmt == "w":
            K.row_io(base, "s[%d:%d]" % (pair, pair + 1))
            return vm.issued
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (T(1, 0), V_LANE))
        return vm.load("global_load_dwordx4 v[%d:%d], v%d, s[%d:%d]" % (base, base + 3, T(1, 0), pair, pair + 1))

    def land(base, seq, first):
        """the row in the first pass's lane map, as words any butterfly takes"""
        vm.wait(seq)
        if fmt == "w":
            return
        if not first:
            K.row_sync()                                                          # (the other wave of the row may still read the slab)
        A, S, D = T(1, 0), T(1, 1), T(1, 2)
        V("v_lshlrev_b32_e32 v%d, 4, v%d" % (A, V_LANE))
        V("v_add_u32_e32 v%d, s90, v%d" % (A, A))                                 # slab + 16 t
        R("ds_write_b128 v%d, v[%d:%d]" % (A, base, base + 3))
        V("v_and_b32_e32 v%d, 0xfffffffc, v%d" % (D, V_LANE))
        V("v_add_u32_e32 v%d, s90, v%d" % (D, D))                                 # slab + (t & ~3)
        V("v_and_b32_e32 v%d, 3, v%d" % (S, V_LANE))
        V("v_lshlrev_b32_e32 v%d, 3, v%d" % (S, S))                               # 8 (t & 3)
        R("s_waitcnt lgkmcnt(0)")
        if K.WAVES > 1:
            R("s_barrier")
        for q in range(16):
            R("ds_read_b32 v%d, v%d offset:%d" % (base + 2 * q, D, W * q))        # the dword that holds byte t + W q
        R("s_waitcnt lgkmcnt(0)")
        t = T(0, 4)
        for q in range(16):
            x = base + 2 * q
            V("v_bfe_i32 v%d, v%d, v%d, 8" % (x, x, S))                           # the signed byte
            V("v_ashrrev_i32_e32 v%d, 31, v%d" % (x + 1, x))                      # x < 0 stands for p + x
            V("v_and_b32_e32 v%d, s24, v%d" % (t, x + 1))
            V("v_and_b32_e32 v%d, s25, v%d" % (t + 1, x + 1))
            V("v_lshl_add_u64 %s, %s, 0, %s" % (vp(x), vp(x), vp(t)))
        K.row_sync()                                                              # (the slab is the transform's exchange buffer next)

    seq_x = request(cfg.V_A, 8)
    seq_e = request(cfg.V_B, 18)
    emit_consts(em)
    f3 = list(range(K.s3_first, 4))
    plan, custom = [], {}
    for tag in ["x", "e0"] + (["e1"] if two else []):
        plan += [("F1" + tag, "F1", [0, 1, 2, 3]), ("F2" + tag, "F2", [0, 1, 2, 3]), ("F3" + tag, "F3", f3)]
        if tag != "x":
            plan += [("K" + tag, "K" + tag, [3])]
            kp = 16 if tag == "e0" else 20
            def key_loader(s, slot, kp=kp):
                V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
                return K.lane16_io(KEY(), "s[%d:%d]" % (kp, kp + 1), addr=T(1, 0))
            custom["K" + tag] = key_loader
    K.set_plan(plan, custom)
    K.prime()

    def forward(tag, base):
        K.run_pass("F1" + tag, [base])
        K.exchange([base], "E1", sync_between=True, sync_before=tag != "x")
        K.run_pass("F2" + tag, [base])
        K.exchange([base], "E2", sync_before=True)
        K.run_pass("F3" + tag, [base])

    land(cfg.V_A, seq_x, True)
    forward("x", cfg.V_A)
    for h, tag in enumerate(["e0"] + (["e1"] if two else [])):
        land(cfg.V_B, seq_e, False)
        forward(tag, cfg.V_B)
        em.comment("out%d = X * k%d + E%d against the key row (words 16 t .. 16 t + 15), canonical, stored from the key's registers" % (h, h, h))
        vm.wait(K.seq[("K" + tag, 3)])
        run_pairs(em, [fma_job(KEY() + 2 * q, cfg.V_A + 2 * q, cfg.V_B + 2 * q, h == 0) for q in range(16)])
        if two and h == 0:
            seq_e = request(cfg.V_B, 84)                 # the second noise row is on its way while the first result leaves
        V("v_lshlrev_b32_e32 v%d, 7, v%d" % (T(1, 0), V_LANE))
        if K.WAVES > 1:
            R("s_cmp_eq_u32 s95, 0")
            R("s_cbranch_scc1 .Lskip%d" % h)
        K.lane16_io(KEY(), "s[%d:%d]" % ((4, 5) if h == 0 else (6, 7)), store=True, addr=T(1, 0))
        L(".Lskip%d:" % h)
        K.release("K" + tag, 3)
    R("s_endpgm")
    return em
