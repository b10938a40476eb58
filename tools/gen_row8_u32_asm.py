#!/usr/bin/env python3
"""Generate nfllib_amd/csrc/row8_u32_gfx950.s -- the fused product for 32-bit limbs, n = 8: the reference's
(8, 60, uint32_t) test config.  c = INTT(NTT(a) (.) NTT(b)), ONE LANE PER RNS ROW.

A row is 8 words = 32 bytes: a lane loads its row with two 16-byte loads (consecutive lanes = consecutive rows: the wave
reads 2 KiB contiguous), runs the three stages in its own registers -- no exchange of any kind -- and stores two 16-byte
vectors.  Same multiply-add butterflies on coefficient pairs as tools/gen_row1024_u32_asm.py (imported); moduli
constants and the 7 twiddle records are per-lane registers (row mod nm selects them).
Run by nfllib_amd/csrc/Makefile after the other generators.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_polymul_asm as G          # noqa: E402
import gen_row1024_u32_asm as U      # noqa: E402  (pair(), interleaving run())

KNAME = "nflhip_row8_u32_asm"
OUT = os.path.join(G.ROOT, "nfllib_amd", "csrc", "row8_u32_gfx950.s")

V_TID, V_ROW, V_OFF, V_TWOFF, V_MCOFF = 0, 1, 2, 3, 4
V_P, V_2P, V_NEGP, V_MU, V_NINV, V_NINVSH, V_W1N, V_W1NSH = 6, 7, 8, 9, 10, 11, 12, 13   # (v[6:13]: one 32-byte load + negp)
V_CA, V_CB = 16, 24          # the loaded rows (8 consecutive registers each); V_CA receives the result
V_A, V_B = 32, 48            # 8 even-aligned coefficient pairs each
V_TW = 64                    # tw[1..7]: record k = {w, w'} at v[64 + 2k : 65 + 2k]  (tw[0] loaded too, unused)
V_S = [80, 84]
V_PW = 88
NEXT_VGPR = 92
NEXT_SGPR = 40
S_DUMMY = "s[30:31]"


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def tw(k):
    return V_TW + 2 * k, V_TW + 2 * k + 1


def ct(x, y, k, xsrc=None, ysrc=None):
    w, wp = tw(k)
    xs, ys = (x if xsrc is None else xsrc), (y if ysrc is None else ysrc)

    def gen(s):
        T0, Q, T2 = V_S[s], V_S[s] + 1, V_S[s] + 2
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, xs, V_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, xs, T0), None, None
        yield "v_mul_hi_u32 v%d, v%d, v%d" % (Q, ys, wp), None, None
        yield "v_lshl_add_u32 v%d, v%d, 1, v%d" % (T2, x, V_2P), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(x), S_DUMMY, Q, V_NEGP, pair(x)), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(x), S_DUMMY, ys, w, pair(x)), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (y, T2, x), None, None
    return gen


def gs(x, y, k):
    w, wp = tw(k)

    def gen(s):
        T0, Q, D, S = V_S[s], V_S[s] + 1, V_S[s] + 2, V_S[s] + 3
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, x, y), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, y, x), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (D, D, V_2P), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, S, V_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, S, T0), None, None
        yield "v_mul_hi_u32 v%d, v%d, v%d" % (Q, D, wp), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (pair(y), S_DUMMY, Q, V_NEGP), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(y), S_DUMMY, D, w, pair(y)), None, None
    return gen


def csub(reg, dst, bound, s):
    T0 = V_S[s]
    yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, reg, bound), None, None
    yield "v_min_u32_e32 v%d, v%d, v%d" % (dst, reg, T0), None, None


def pointwise(a, b):
    def gen(s):
        Q, TH = V_S[s] + 1, V_S[s] + 2
        P0 = V_PW + 2 * s
        for r in (a, b):
            yield from csub(r, r, V_2P, s)
            yield from csub(r, r, V_P, s)
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (pair(P0), S_DUMMY, a, b), None, None
        yield "v_alignbit_b32 v%d, v%d, v%d, 28" % (TH, P0 + 1, P0), None, None
        yield "v_mul_hi_u32 v%d, v%d, v%d" % (Q, TH, V_MU), None, None
        yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(P0), S_DUMMY, Q, V_NEGP, pair(P0)), None, None
        yield from csub(P0, a, V_2P, s)
    return gen


def mul_shoup_exact(y, tmp_pair, dst, w, wp, s):
    Q = V_S[s] + 1
    yield "v_mul_hi_u32 v%d, v%d, v%d" % (Q, y, wp), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, 0" % (pair(tmp_pair), S_DUMMY, Q, V_NEGP), None, None
    yield "v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (pair(tmp_pair), S_DUMMY, y, w, pair(tmp_pair)), None, None
    yield from csub(tmp_pair, dst, V_P, s)


def last(u, x, du, dx):
    """stage 0 of the inverse with n^-1 folded in; canonical results into the consecutive registers du, dx"""
    def gen(s):
        D, S = V_S[s] + 2, V_S[s] + 3
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, u, x), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, x, u), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (D, D, V_2P), None, None
        yield from mul_shoup_exact(S, u, du, V_NINV, V_NINVSH, s)
        yield from mul_shoup_exact(D, x, dx, V_W1N, V_W1NSH, s)
    return gen


def build(mode="polymul"):
    """mode: polymul | polymul_ntt (b already in NTT form) | fwd (c = NTT(a), canonical) | inv (c = INTT(a))"""
    em = G.Emitter()
    R = em.raw
    V = em.valu
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, magic = ceil(2^32 / nm) (0 when nm = 1)
    R("s_load_dwordx2 s[16:17], s[0:1], 0x30")           # rows
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s18, s2, 8")                           # first row of the workgroup
    V("v_add_u32_e32 v%d, s18, v%d" % (V_ROW, V_TID))
    R("s_sub_u32 s19, s16, 1")
    V("v_min_u32_e32 v%d, s19, v%d" % (V_OFF, V_ROW))     # surplus lanes repeat the last row and store nothing
    V("v_mul_hi_u32 v%d, v%d, s15" % (V_TWOFF, V_OFF))
    V("v_mul_lo_u32 v%d, v%d, s14" % (V_TWOFF, V_TWOFF))
    V("v_sub_u32_e32 v%d, v%d, v%d" % (V_TWOFF, V_OFF, V_TWOFF))   # cm = row mod nm
    R("s_cmp_eq_u32 s14, 1")
    R("s_cselect_b64 vcc, -1, 0")
    V("v_cndmask_b32_e64 v%d, v%d, 0, vcc" % (V_TWOFF, V_TWOFF))
    V("v_mul_u32_u24_e32 v%d, 56, v%d" % (V_MCOFF, V_TWOFF))       # its ModConst<u32> record
    V("v_lshlrev_b32_e32 v%d, 6, v%d" % (V_TWOFF, V_TWOFF))        # its twiddle table: 8 records of 8 bytes
    V("v_lshlrev_b32_e32 v%d, 5, v%d" % (V_OFF, V_OFF))            # the row: 32 bytes
    has_b = mode in ("polymul", "polymul_ntt")
    for i in range(2):
        R("global_load_dwordx4 v[%d:%d], v%d, s[6:7] offset:%d" % (V_CA + 4 * i, V_CA + 4 * i + 3, V_OFF, 16 * i))
        if has_b:
            R("global_load_dwordx4 v[%d:%d], v%d, s[8:9] offset:%d" % (V_CB + 4 * i, V_CB + 4 * i + 3, V_OFF, 16 * i))
    # p 2p mu ninv | ninv_sh w1ninv w1ninv_sh (beta): 28 bytes used
    R("global_load_dwordx4 v[%d:%d], v%d, s[12:13]" % (V_S[0], V_S[0] + 3, V_MCOFF))
    R("global_load_dwordx3 v[%d:%d], v%d, s[12:13] offset:16" % (V_S[1], V_S[1] + 2, V_MCOFF))
    for i in range(4):
        R("global_load_dwordx4 v[%d:%d], v%d, s[10:11] offset:%d" % (V_TW + 4 * i, V_TW + 4 * i + 3, V_TWOFF, 16 * i))
    R("s_waitcnt vmcnt(4)")                              # rows and constants are here (the twiddles may still fly)
    V("v_mov_b32_e32 v%d, v%d" % (V_P, V_S[0]))
    V("v_mov_b32_e32 v%d, v%d" % (V_2P, V_S[0] + 1))
    V("v_mov_b32_e32 v%d, v%d" % (V_MU, V_S[0] + 2))
    V("v_mov_b32_e32 v%d, v%d" % (V_NINV, V_S[0] + 3))
    V("v_mov_b32_e32 v%d, v%d" % (V_NINVSH, V_S[1]))
    V("v_mov_b32_e32 v%d, v%d" % (V_W1N, V_S[1] + 1))
    V("v_mov_b32_e32 v%d, v%d" % (V_W1NSH, V_S[1] + 2))
    V("v_sub_u32_e32 v%d, 0, v%d" % (V_NEGP, V_P))
    R("s_waitcnt vmcnt(0)")
    # ---------------------------------------------------------------- forward (both operands of a full product): tw[(1 << s) + g]
    fwd_ops = {"polymul": ((V_CA, V_A), (V_CB, V_B)), "polymul_ntt": ((V_CA, V_A),), "fwd": ((V_CA, V_A),), "inv": ()}[mode]
    for s in range(3):
        half = 4 >> s
        jobs = []
        for g in range(1 << s):
            for h in range(half):
                for src, dst in fwd_ops:
                    i0 = g * 2 * half + h
                    if s == 0:   # the loaded words sit in consecutive registers
                        jobs.append(ct(dst + 2 * i0, dst + 2 * (i0 + half), (1 << s) + g, xsrc=src + i0, ysrc=src + i0 + half))
                    else:
                        jobs.append(ct(dst + 2 * i0, dst + 2 * (i0 + half), (1 << s) + g))
        if jobs:
            U.run(em, jobs)
    if mode == "fwd":            # canonical words into the consecutive registers, then out
        def canon(q):
            def gen(s_):
                yield from csub(V_A + 2 * q, V_A + 2 * q, V_2P, s_)
                yield from csub(V_A + 2 * q, V_CA + q, V_P, s_)
            return gen
        U.run(em, [canon(q) for q in range(8)])
    elif mode == "inv":          # NTT-form words (canonical) into the coefficient pairs
        for q in range(8):
            V("v_mov_b32_e32 v%d, v%d" % (V_A + 2 * q, V_CA + q))
    elif mode == "polymul_ntt":  # b arrives transformed: its words are the loaded registers themselves
        U.run(em, [pointwise(V_A + 2 * q, V_CB + q) for q in range(8)])
    else:
        U.run(em, [pointwise(V_A + 2 * q, V_B + 2 * q) for q in range(8)])
    if mode != "fwd":
        # ------------------------------------------------------------ inverse: tw[(2 << s) - 1 - g], then n^-1
        for s in (2, 1):
            half = 4 >> s
            jobs = []
            for g in range(1 << s):
                for h in range(half):
                    i0 = g * 2 * half + h
                    jobs.append(gs(V_A + 2 * i0, V_A + 2 * (i0 + half), (2 << s) - 1 - g))
            U.run(em, jobs)
        U.run(em, [last(V_A + 2 * h, V_A + 2 * (h + 4), V_CA + h, V_CA + h + 4) for h in range(4)])
    V("v_cmp_gt_u32_e32 vcc, s16, v%d" % V_ROW)
    R("s_and_saveexec_b64 s[32:33], vcc")
    for i in range(2):
        R("global_store_dwordx4 v%d, v[%d:%d], s[4:5] offset:%d" % (V_OFF, V_CA + 4 * i, V_CA + 4 * i + 3, 16 * i))
    R("s_endpgm")
    return em


ARGS = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("ptr", 48)]


def main():
    for mode, sfx in (("polymul", ""), ("polymul_ntt", "_ntt"), ("fwd", "_fwd"), ("inv", "_inv")):
        em = build(mode)
        kname = "nflhip_row8%s_u32_asm" % sfx
        out = os.path.join(G.ROOT, "nfllib_amd", "csrc", "row8%s_u32_gfx950.s" % sfx)
        accum = (NEXT_VGPR + 3) // 4 * 4
        params = dict(k=kname, lds=0, vgpr=NEXT_VGPR, sgpr=NEXT_SGPR, accum=accum, sgprc=NEXT_SGPR + 6, wg=256,
                      karg=56, args=G.args_yaml(ARGS).replace("{.address_space: global, .offset: 48, .size: 8, .value_kind: global_buffer}",
                                                                "{.offset: 48, .size: 8, .value_kind: by_value}"))
        with open(out, "w") as f:
            f.write("; GENERATED by tools/gen_row8_u32_asm.py -- do not edit.\n")
            f.write(G.HEADER % params)
            f.write("\n".join(em.lines) + "\n")
            f.write(G.FOOTER % params)
        print("wrote %s: %d VALU instructions (static), %d lines" % (out, em.n_valu, len(em.lines)))


if __name__ == "__main__":
    main()
