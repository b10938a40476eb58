"""The instruction emitter (hazard tracking, s_nop insertion, interleaving of instruction streams) and the vmcnt bookkeeping."""
import os

from . import state as cfg

def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(pair):
    return "s[%d:%d]" % pair


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
        if ("nolds" in cfg.ABLATE and text.startswith("ds_")) or ("nobar" in cfg.ABLATE and text.startswith("s_barrier")):
            return
        self.lines.append("\t" + text)
        self.pos += 1

    def comment(self, text):
        self.lines.append("\t; " + text)

    def valu(self, text, wr=None, rd=None):
        if cfg.SWAP_MULHI and text.startswith("v_mul_hi_u32 "):   # experiment: the same question for v_mul_hi_u32
            ops = [o.strip() for o in text[len("v_mul_hi_u32 "):].split(",")]
            if len(ops) == 3:
                text = "v_mul_hi_u32 %s, %s, %s" % (ops[0], ops[2], ops[1])
        if cfg.SWAP_MAD != "0" and text.startswith("v_mad_u64_u32 "):
            # The butterfly code is written "data x constant" (v_mad_u64_u32 D, carry, data, twiddle-or-constant, addend); what is
            # EMITTED is "constant x data": same result, same issue cost, and 1.1 - 2.3 % more products/s on the metric kernel --
            # the kernels run at the package power limit and the multiplier draws less with the sparse constants (delta < 2^27,
            # the 2^62 term) in its first operand (same-box A/B, equal checksums: profiles/r03_mad_operand_order.txt; exchanging
            # only the SGPR-constant ones +1.8 %, only the twiddle ones +0.4 %, all +2.3 %).  NFL_GEN_SWAP_MAD=0 / sgpr / vgpr.
            ops = [o.strip() for o in text[len("v_mad_u64_u32 "):].split(",")]
            # operands: vdst ("v[a:b]"), sdst ("s[a:b]"), src0, src1, src2
            sgpr = ops[3].startswith("s") if len(ops) == 5 else False
            if len(ops) == 5 and (cfg.SWAP_MAD in ("", "1") or (cfg.SWAP_MAD == "sgpr" and sgpr) or (cfg.SWAP_MAD == "vgpr" and not sgpr)):
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


def run_pairs(em, jobs):
    """jobs: list of callables(stream) -> generator; executed two at a time, interleaved."""
    if cfg.SINGLE_STREAM:
        for j in jobs:
            interleave(em, [j(0)])
        return
    for i in range(0, len(jobs), 2):
        gens = [jobs[i](0)]
        if i + 1 < len(jobs):
            gens.append(jobs[i + 1](1))
        interleave(em, gens)


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
