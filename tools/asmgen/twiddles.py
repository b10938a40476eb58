"""Twiddle plans: which twiddle record a butterfly stage reads, from where (uniform / per-lane, lane-major copy), the resident
records of the pair map and the ring of the row kernels."""
import os

from . import state as cfg
from .emitter import run_pairs, vp
from .arith import ct_bfly, gs_bfly

# ------------------------------------------------------------------ passes
def twreg(i):
    b = cfg.V_TW + 4 * i
    return ("v%d" % b, "v%d" % (b + 1), "v%d" % (b + 2), "v%d" % (b + 3))


def tw_slot(s, g):
    return (1 << s) - 1 + g          # 15 records: sub-stage s (0..3), group g (0..2^s-1)


def ct_stage(em, bases, s, slot=tw_slot):
    if "nobfly" in cfg.ABLATE:
        return
    half = 8 >> s
    jobs = []
    for g in range(1 << s):
        tw = twreg(slot(s, g))
        for h in range(half):
            i0 = g * 2 * half + h
            for base in bases:
                jobs.append(ct_bfly(base + 2 * i0, base + 2 * (i0 + half), tw))
    run_pairs(em, jobs)


def gs_stage(em, base, s, slot=tw_slot):
    if "nobfly" in cfg.ABLATE:
        return
    half = 8 >> s
    jobs = []
    for g in range(1 << s):
        tw = twreg(slot(s, g))
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


def tw_base_lm(em, kreg, s, g, descending, koff=0):
    c = (2 << s) - 2 - g if descending else (1 << s) - 1 + g
    if c:
        em.raw("s_lshl_b32 s86, 0x%x, %s" % (256 * c, cfg.S_R))
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
    em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (cfg.V_TWO, cfg.V_TID))
    if descending:
        em.valu("v_sub_u32_e32 v%d, 0xff0, v%d" % (cfg.V_TWO, cfg.V_TWO))
    if "tw0" in cfg.ABLATE:
        em.valu("v_mov_b32_e32 v%d, 0" % (cfg.V_TWO,))


def tw_uniform_stage(em, vm, s, kreg, descending):
    """Twiddle records of sub-stage s at wave-uniform indices (K << s) + g  /  (K << s) - 1 - g."""
    tw_base(em, kreg, s, descending)
    seq = 0
    for g in range(1 << s):
        r = cfg.V_TW + 4 * tw_slot(s, g)
        off = -g * 16 if descending else g * 16
        seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, cfg.V_ZERO, cfg.S_BASE2, off))
    return seq


def tw_lane_stage(em, vm, s, vidx, kreg, descending, groups=None, slot=tw_slot):
    """Per-lane twiddle records of sub-stage s.  Ascending (forward): index = (K << s) + (vidx << s) + g.
    Descending (inverse, mirrored): index = (K << s) - 1 - (vidx << s) - g.  vidx: VGPR with B or t.
    groups: only these g (default: all 2^s)"""
    seq = 0
    groups = range(1 << s) if groups is None else groups
    tw_base(em, kreg, s, descending)
    em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, vidx))
    if "tw0" in cfg.ABLATE:
        em.valu("v_mov_b32_e32 v%d, 0" % (cfg.V_TWO,))
    if not descending:
        for g in groups:
            r = cfg.V_TW + 4 * slot(s, g)
            seq = vm.load("global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, cfg.V_TWO, cfg.S_BASE2, g * 16))
    else:
        em.valu("v_mov_b32_e32 v%d, s84" % (cfg.V_TWA,))
        em.valu("v_mov_b32_e32 v%d, s85" % (cfg.V_TWA + 1,))
        em.valu("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (cfg.V_TWA, cfg.V_TWA, cfg.V_TWO), "vcc", None)
        em.valu("v_subbrev_co_u32_e32 v%d, vcc, 0, v%d, vcc" % (cfg.V_TWA + 1, cfg.V_TWA + 1), "vcc", "vcc")
        for g in groups:
            r = cfg.V_TW + 4 * slot(s, g)
            seq = vm.load("global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (r, r + 3, vp(cfg.V_TWA), -g * 16))
    return seq


# twiddle index of (pass, sub-stage s, group g); see fwd_head / fwd_tail / inv_core of kernels_fast.hip:
# with Kf = 2^r + blk the forward passes use K = Kf, 16*Kf, 256*Kf; the mirrored inverse passes use
# K = (512<<r) - 256*blk, (32<<r) - 16*blk, (2<<r) - blk.
PASS_TW = {
    "F1": lambda em, vm, s: tw_uniform_stage(em, vm, s, cfg.S_K["F1"], False),
    "F2": lambda em, vm, s: tw_lane_stage(em, vm, s, cfg.V_BIDX, cfg.S_K["F2"], False),
    "F3": lambda em, vm, s: tw_lane_stage(em, vm, s, cfg.V_TID, cfg.S_K["F3"], False),
    "I1": lambda em, vm, s: tw_lane_stage(em, vm, s, cfg.V_TID, cfg.S_K["I1"], True),
    "I2": lambda em, vm, s: tw_lane_stage(em, vm, s, cfg.V_BIDX, cfg.S_K["I2"], True),
    "I3": lambda em, vm, s: tw_uniform_stage(em, vm, s, cfg.S_K["I3"], True),
}


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
        b = cfg.V_TW + 4 * self.slot_of[use]
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
        em, r = self.em, cfg.V_TW + 4 * slot
        if self.passes[name] is None:     # a RESERVED slot: four scratch registers handed out in ring order, nothing is loaded
            self.seq_of[use] = self.vm.issued
            return
        if callable(self.passes[name]):   # not a twiddle record: the pass supplies the load (row32k streams b' this way)
            text = self.passes[name](em, r, s, g, self.cur != (name, s))
            self.cur = (name, s)
            self.seq_of[use] = self.vm.load(text)
            return
        kreg, vidx, desc = self.passes[name][:3]
        koff = self.passes[name][3] if len(self.passes[name]) > 3 else 0
        fresh = self.cur != (name, s)
        if cfg.LANE_MAJOR and vidx is not None and vidx == cfg.V_TID:
            if self.cur is None or self.cur[0] != name:
                tw_lane_offset_lm(em, desc)
            self.cur = (name, s)
            tw_base_lm(em, kreg, s, g, desc, koff)
            self.seq_of[use] = self.vm.load("global_load_dwordx4 v[%d:%d], v%d, %s" % (r, r + 3, cfg.V_TWO, cfg.S_BASE2))
            return
        if fresh:
            tw_base(em, kreg, s, desc, koff)
            if vidx is not None:
                em.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, s + 4, vidx))
                if "tw0" in cfg.ABLATE:
                    em.valu("v_mov_b32_e32 v%d, 0" % (cfg.V_TWO,))
            self.cur = (name, s)
        if vidx is not None and desc and (fresh or cfg.RING_RECOMPUTE_TWA):   # (the address pair is butterfly scratch in ringpair mode)
            em.valu("v_mov_b32_e32 v%d, s84" % (cfg.V_TWA,))
            em.valu("v_mov_b32_e32 v%d, s85" % (cfg.V_TWA + 1,))
            em.valu("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (cfg.V_TWA, cfg.V_TWA, cfg.V_TWO), "vcc", None)
            em.valu("v_subbrev_co_u32_e32 v%d, vcc, 0, v%d, vcc" % (cfg.V_TWA + 1, cfg.V_TWA + 1), "vcc", "vcc")
        off = -g * 16 if desc else g * 16
        if vidx is None:
            text = "global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, cfg.V_ZERO, cfg.S_BASE2, off)
        elif not desc:
            text = "global_load_dwordx4 v[%d:%d], v%d, %s offset:%d" % (r, r + 3, cfg.V_TWO, cfg.S_BASE2, off)
        else:
            text = "global_load_dwordx4 v[%d:%d], %s, off offset:%d" % (r, r + 3, vp(cfg.V_TWA), off)
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
