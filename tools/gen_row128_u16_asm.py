#!/usr/bin/env python3
"""Generate nfllib_amd/csrc/row128_u16_gfx950.s -- the fused product for 16-bit limbs (14-bit moduli), n = 128: the
reference's (128, 14, uint16_t) test config.  c = INTT(NTT(a) (.) NTT(b)), EIGHT ROWS PER WAVE.

A row is 128 words = 8 lanes x 16 words; lane (r, l) of a wave (r = lane / 8 the row, l = lane % 8) holds x[l + 8 q] in
register q: four stages in registers (distances 64 .. 8), one wave-local LDS exchange, then the lane holds the 16
consecutive words 16 l .. 16 l + 15 = two 8-word blocks: the last three stages in registers.  Every word is loaded with
one zero-extending 2-byte load and lives in a 32-bit register; 14-bit residues make every product fit 32 bits, so the
butterflies are full-rate 24-bit multiplies:
    Cooley-Tukey (8):  t = x - 2p ; X = min(x, t) ; u = 2X + 2p ; h = y w' ; g = (h >> 16) p [SDWA] ; m = y w + X ;
                       x' = m - g ; y' = u - x'
    Gentleman-Sande (9), Barrett point-wise product (15), last inverse stage with n^-1 (15): the same arithmetic as the
generic kernels (Harvey lazy ranges 4p < 2^16, Shoup constants floor(w 2^16 / p) from the device tables).
Rows of one wave may belong to different moduli (row mod nm), so moduli constants and twiddles are per-lane registers.
Run by nfllib_amd/csrc/Makefile (after tools/gen_polymul_asm.py, whose emitter and file templates it reuses).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_polymul_asm as G   # noqa: E402


V_TID, V_LANE, V_L, V_R, V_GOFF, V_LW, V_LR, V_TWOFF = 0, 1, 2, 3, 4, 5, 6, 7
V_P, V_2P, V_MU, V_NINV, V_NINVSH, V_W1N, V_W1NSH, V_ROW = 8, 9, 10, 11, 12, 13, 14, 15
V_A, V_B = 16, 32            # 16 words each
V_W1, V_WP1 = 48, 64          # records 1 .. 15 of the row-uniform passes (index k at +k): w, w'
V_W2, V_WP2 = 80, 94          # 14 per-lane records of the other pass
V_S = [108, 114]              # per-stream temporaries (6 each)
V_X = 120                     # address temporaries (2)
V_OFF2 = 122                  # byte offset of the lane's 16 consecutive words (NTT-form I/O of the stand-alone transforms)
NEXT_VGPR = 124
NEXT_SGPR = 40
ROW_WORDS = 136               # LDS words per row: 128 + one pad word per 16
SLAB = 8 * ROW_WORDS * 4      # bytes per wave


def ct(x, y, w, wp):
    def gen(s):
        T0, U, H, Gq, M = (V_S[s] + i for i in range(5))
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, x, V_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, x, T0), None, None
        yield "v_lshl_add_u32 v%d, v%d, 1, v%d" % (U, x, V_2P), None, None
        yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (H, y, wp), None, None
        yield "v_mul_u32_u24_sdwa v%d, v%d, v%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" % (Gq, H, V_P), None, None
        yield "v_mad_u32_u24 v%d, v%d, v%d, v%d" % (M, y, w, x), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (x, M, Gq), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (y, U, x), None, None
    return gen


def gs(x, y, w, wp):
    def gen(s):
        T0, S, D, H, Gq, M = (V_S[s] + i for i in range(6))
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, x, y), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, y, x), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (D, D, V_2P), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, S, V_2P), None, None
        yield "v_min_u32_e32 v%d, v%d, v%d" % (x, S, T0), None, None
        yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (H, D, wp), None, None
        yield "v_mul_u32_u24_sdwa v%d, v%d, v%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" % (Gq, H, V_P), None, None
        yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (M, D, w), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (y, M, Gq), None, None
    return gen


def csub(reg, dst, bound, s):
    T0 = V_S[s]
    yield "v_sub_u32_e32 v%d, v%d, v%d" % (T0, reg, bound), None, None
    yield "v_min_u32_e32 v%d, v%d, v%d" % (dst, reg, T0), None, None


def pointwise(a, b):
    """a = a b mod p in [0, 2p): barrett<uint16_t>::mul on canonical operands, without its last subtract"""
    def gen(s):
        T, TH, Q, Gq = (V_S[s] + i for i in range(1, 5))
        for r in (a, b):
            yield from csub(r, r, V_2P, s)
            yield from csub(r, r, V_P, s)
        yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (T, a, b), None, None
        yield "v_lshrrev_b32_e32 v%d, 12, v%d" % (TH, T), None, None
        yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (Q, TH, V_MU), None, None
        yield "v_mul_u32_u24_sdwa v%d, v%d, v%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" % (Gq, Q, V_P), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (T, T, Gq), None, None       # in [0, 3p): the quotient never overshoots
        yield from csub(T, a, V_2P, s)
    return gen


def mul_shoup_exact(y, dst, w, wp, s):
    H, Gq, M = V_S[s] + 3, V_S[s] + 4, V_S[s] + 5
    yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (H, y, wp), None, None
    yield "v_mul_u32_u24_sdwa v%d, v%d, v%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" % (Gq, H, V_P), None, None
    yield "v_mul_u32_u24_e32 v%d, v%d, v%d" % (M, y, w), None, None
    yield "v_sub_u32_e32 v%d, v%d, v%d" % (M, M, Gq), None, None
    yield from csub(M, dst, V_P, s)


def last(u, x):
    def gen(s):
        S, D = V_S[s] + 1, V_S[s] + 2
        yield "v_add_u32_e32 v%d, v%d, v%d" % (S, u, x), None, None
        yield "v_sub_u32_e32 v%d, v%d, v%d" % (D, x, u), None, None
        yield "v_add_u32_e32 v%d, v%d, v%d" % (D, D, V_2P), None, None
        yield from mul_shoup_exact(S, u, V_NINV, V_NINVSH, s)
        yield from mul_shoup_exact(D, x, V_W1N, V_W1NSH, s)
    return gen


def run(em, jobs):
    for i in range(0, len(jobs), 2):
        gens = [jobs[i](0)]
        if i + 1 < len(jobs):
            gens.append(jobs[i + 1](1))
        G.interleave(em, gens)


def build(mode="polymul"):
    """mode: polymul | polymul_ntt (b already in NTT form) | fwd (c = NTT(a), canonical) | inv (c = INTT(a))"""
    em = G.Emitter()
    R = em.raw
    L = em.lines.append
    V = em.valu
    R("s_load_dwordx8 s[4:11], s[0:1], 0x0")             # c, a, b, psi
    R("s_load_dwordx4 s[12:15], s[0:1], 0x20")           # mc, nm, -
    R("s_load_dwordx2 s[16:17], s[0:1], 0x30")           # rows
    V("v_and_b32_e32 v%d, 63, v%d" % (V_LANE, V_TID))
    V("v_and_b32_e32 v%d, 7, v%d" % (V_L, V_TID))
    V("v_lshrrev_b32_e32 v%d, 3, v%d" % (V_R, V_LANE))
    V("v_readfirstlane_b32 s18, v%d" % V_TID)
    R("s_lshr_b32 s18, s18, 6")                          # wave of the workgroup
    R("s_waitcnt lgkmcnt(0)")
    R("s_lshl_b32 s19, s2, 2")
    R("s_add_u32 s19, s19, s18")
    R("s_lshl_b32 s19, s19, 3")                          # first row of the wave
    R("s_cmp_lt_u32 s19, s16")
    R("s_cbranch_scc1 .Llive")
    R("s_endpgm")
    L(".Llive:")
    R("s_lshr_b32 s21, s19, 24")
    R("s_lshl_b32 s20, s19, 8")                          # first row * 256 bytes
    for base, dst in ((6, 24), (8, 26), (4, 28)):
        R("s_add_u32 s%d, s%d, s20" % (dst, base))
        R("s_addc_u32 s%d, s%d, s21" % (dst + 1, base + 1))
    R("s_sub_u32 s22, s16, 1")                           # last valid row
    R("s_sub_u32 s23, s14, 1")                           # nm - 1 (nm is a power of two: 1 or 2 moduli of this size exist)
    V("v_add_u32_e32 v%d, s19, v%d" % (V_ROW, V_R))       # the lane's row ...
    V("v_min_u32_e32 v%d, s22, v%d" % (V_X, V_ROW))       # ... clamped for the loads (surplus lanes repeat the last row, store nothing)
    V("v_and_b32_e32 v%d, s23, v%d" % (V_TWOFF, V_X))     # cm
    V("v_mul_u32_u24_e32 v%d, 28, v%d" % (V_X + 1, V_TWOFF))   # its ModConst<u16> record
    for k, dst in enumerate((V_P, V_2P, V_MU, V_NINV, V_NINVSH, V_W1N, V_W1NSH)):
        R("global_load_ushort v%d, v%d, s[12:13] offset:%d" % (dst, V_X + 1, 2 * k))
    V("v_lshlrev_b32_e32 v%d, 9, v%d" % (V_TWOFF, V_TWOFF))    # its twiddle table: 128 records of 4 bytes
    V("v_subrev_u32_e32 v%d, s19, v%d" % (V_GOFF, V_X))
    V("v_lshlrev_b32_e32 v%d, 8, v%d" % (V_GOFF, V_GOFF))
    V("v_lshl_add_u32 v%d, v%d, 1, v%d" % (V_GOFF, V_L, V_GOFF))   # (row - first row) * 256 + 2 l
    V("v_mov_b32_e32 v%d, 30" % V_OFF2)
    V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_OFF2, V_L, V_OFF2, V_GOFF))   # NTT-form words 16 l + j: (row - first row) * 256 + 32 l
    if mode == "inv":
        for j in range(16):
            R("global_load_ushort v%d, v%d, s[24:25] offset:%d" % (V_A + j, V_OFF2, 2 * j))
    else:
        for q in range(16):
            R("global_load_ushort v%d, v%d, s[24:25] offset:%d" % (V_A + q, V_GOFF, 16 * q))
        if mode == "polymul":
            for q in range(16):
                R("global_load_ushort v%d, v%d, s[26:27] offset:%d" % (V_B + q, V_GOFF, 16 * q))
        elif mode == "polymul_ntt":   # NTT-form words 16 l + j: the layout the point-wise step works in
            for j in range(16):
                R("global_load_ushort v%d, v%d, s[26:27] offset:%d" % (V_B + j, V_OFF2, 2 * j))
    for k in range(1, 16):                               # the row-uniform records, raw {w | w' << 16} into the w' registers
        R("global_load_dword v%d, v%d, s[10:11] offset:%d" % (V_WP1 + k, V_TWOFF, 4 * k))
    # LDS: word e of row r at 4 (136 r + e + (e >> 4)); write base (e = l + 8 q), read base (e = 16 l + j)
    R("s_mul_i32 s30, s18, %d" % SLAB)
    V("v_mov_b32_e32 v%d, %d" % (V_X, ROW_WORDS))
    V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_LW, V_R, V_X, V_L))
    V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_LW, V_LW))
    V("v_add_u32_e32 v%d, s30, v%d" % (V_LW, V_LW))
    V("v_mul_u32_u24_e32 v%d, v%d, v%d" % (V_LR, V_R, V_X))
    V("v_mov_b32_e32 v%d, 17" % V_X)
    V("v_mad_u32_u24 v%d, v%d, v%d, v%d" % (V_LR, V_L, V_X, V_LR))
    V("v_lshlrev_b32_e32 v%d, 2, v%d" % (V_LR, V_LR))
    V("v_add_u32_e32 v%d, s30, v%d" % (V_LR, V_LR))

    def unpack(w_base, wp_base, idx):
        for k in idx:
            V("v_and_b32_e32 v%d, 0xffff, v%d" % (w_base + k, wp_base + k))
            V("v_lshrrev_b32_e32 v%d, 16, v%d" % (wp_base + k, wp_base + k))

    def lane_records(i, inverse):
        """the 2 / 4 / 8 per-lane records of stage i of the lane-local pass -> raw into WP2[off .. off + G)"""
        Gn = 2 << i
        off = Gn - 2                                     # 0, 2, 6
        if not inverse:                                  # tw[(16 << i) + G l + g]
            V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_X, 3 + i, V_L))
            V("v_add_u32_e32 v%d, v%d, v%d" % (V_X, V_X, V_TWOFF))
            imm = 4 * (16 << i)
        else:                                            # tw[(32 << i) - 1 - (G l + g)]: the block [(32 << i) - G (l + 1), + G)
            V("v_add_u32_e32 v%d, 1, v%d" % (V_X, V_L))
            V("v_lshlrev_b32_e32 v%d, %d, v%d" % (V_X, 3 + i, V_X))
            V("v_sub_u32_e32 v%d, v%d, v%d" % (V_X, V_TWOFF, V_X))
            V("v_add_u32_e32 v%d, 0x%x, v%d" % (V_X, 4 * (32 << i), V_X))   # (the register offset is unsigned: keep it non-negative)
            imm = 0
        if Gn == 2:
            R("global_load_dwordx2 v[%d:%d], v%d, s[10:11] offset:%d" % (V_WP2 + off, V_WP2 + off + 1, V_X, imm))
        else:
            for c in range(Gn // 4):
                R("global_load_dwordx4 v[%d:%d], v%d, s[10:11] offset:%d" % (V_WP2 + off + 4 * c, V_WP2 + off + 4 * c + 3, V_X, imm + 16 * c))
        return off, Gn

    def exchange(bases, to_blocks):
        """to_blocks: x[l + 8 q] -> words 16 l + j; else the way back"""
        for b in bases:
            for q in range(16):
                if to_blocks:
                    R("ds_write_b32 v%d, v%d offset:%d" % (V_LW, b + q, 4 * (8 * q + (q >> 1))))
                else:
                    R("ds_write_b32 v%d, v%d offset:%d" % (V_LR, b + q, 4 * q))
            for q in range(16):
                if to_blocks:
                    R("ds_read_b32 v%d, v%d offset:%d" % (b + q, V_LR, 4 * q))
                else:
                    R("ds_read_b32 v%d, v%d offset:%d" % (b + q, V_LW, 4 * (8 * q + (q >> 1))))
            R("s_waitcnt lgkmcnt(0)")

    both = [V_A, V_B] if mode == "polymul" else [V_A]
    R("s_waitcnt vmcnt(0)")
    unpack(V_W1, V_WP1, range(1, 16))
    if mode != "inv":
        # ------------------------------------------------------------ forward (both operands of a product together)
        for s in range(4):
            half = 8 >> s
            jobs = []
            for g in range(1 << s):
                k = (1 << s) + g
                for h in range(half):
                    for b in both:
                        i0 = g * 2 * half + h
                        jobs.append(ct(b + i0, b + i0 + half, V_W1 + k, V_WP1 + k))
            run(em, jobs)
        for i in range(3):                               # (all three stages' records: 14 registers)
            lane_records(i, False)
        exchange(both, True)
        R("s_waitcnt vmcnt(0)")
        unpack(V_W2, V_WP2, range(14))
        for i in range(3):
            d = 4 >> i
            Gn = 2 << i
            off = Gn - 2
            jobs = []
            for g in range(Gn):
                for h in range(d):
                    for b in both:
                        jobs.append(ct(b + 2 * d * g + h, b + 2 * d * g + h + d, V_W2 + off + g, V_WP2 + off + g))
            run(em, jobs)
    V("v_cmp_gt_u32_e32 vcc, s16, v%d" % V_ROW)          # rows > row: a real row (surplus lanes store nothing)
    if mode == "fwd":
        def canon(q):
            def gen(s):
                yield from csub(V_A + q, V_A + q, V_2P, s)
                yield from csub(V_A + q, V_A + q, V_P, s)
            return gen
        run(em, [canon(q) for q in range(16)])
        R("s_and_saveexec_b64 s[32:33], vcc")
        for j in range(16):
            R("global_store_short v%d, v%d, s[28:29] offset:%d" % (V_OFF2, V_A + j, 2 * j))
        R("s_endpgm")
        return em
    for i in range(3):
        lane_records(i, True)
    if mode in ("polymul", "polymul_ntt"):
        # ------------------------------------------------------------ point-wise product -> a, in [0, 2p)
        run(em, [pointwise(V_A + q, V_B + q) for q in range(16)])
    # ---------------------------------------------------------------- inverse
    R("s_waitcnt vmcnt(0)")
    unpack(V_W2, V_WP2, range(14))
    for i in (2, 1, 0):
        d = 4 >> i
        Gn = 2 << i
        off = Gn - 2
        jobs = []
        for g in range(Gn):
            for h in range(d):
                jobs.append(gs(V_A + 2 * d * g + h, V_A + 2 * d * g + h + d, V_W2 + off + (Gn - 1 - g), V_WP2 + off + (Gn - 1 - g)))
        run(em, jobs)
    exchange([V_A], False)
    for s in (3, 2, 1):                                  # row-uniform: tw[(2 << s) - 1 - g]
        half = 8 >> s
        jobs = []
        for g in range(1 << s):
            k = (2 << s) - 1 - g
            for h in range(half):
                i0 = g * 2 * half + h
                jobs.append(gs(V_A + i0, V_A + i0 + half, V_W1 + k, V_WP1 + k))
        run(em, jobs)
    run(em, [last(V_A + h, V_A + h + 8) for h in range(8)])
    V("v_cmp_gt_u32_e32 vcc, s16, v%d" % V_ROW)
    R("s_and_saveexec_b64 s[32:33], vcc")
    for q in range(16):
        R("global_store_short v%d, v%d, s[28:29] offset:%d" % (V_GOFF, V_A + q, 16 * q))
    R("s_endpgm")
    return em


ARGS = [("ptr", 0), ("ptr", 8), ("ptr", 16), ("ptr", 24), ("ptr", 32), ("i32", 40), ("i32", 44), ("ptr", 48)]


def main():
    for mode in ("polymul", "polymul_ntt", "fwd", "inv"):
        em = build(mode)
        sfx = {"polymul": "", "polymul_ntt": "_ntt"}.get(mode, "_" + mode)
        kname = "nflhip_row128%s_u16_asm" % sfx
        out = os.path.join(G.ROOT, "nfllib_amd", "csrc", "row128%s_u16_gfx950.s" % sfx)
        accum = (NEXT_VGPR + 3) // 4 * 4
        params = dict(k=kname, lds=4 * SLAB, vgpr=NEXT_VGPR, sgpr=NEXT_SGPR, accum=accum, sgprc=NEXT_SGPR + 6, wg=256,
                      karg=56, args=G.args_yaml(ARGS).replace("{.address_space: global, .offset: 48, .size: 8, .value_kind: global_buffer}",
                                                                "{.offset: 48, .size: 8, .value_kind: by_value}"))
        with open(out, "w") as f:
            f.write("; GENERATED by tools/gen_row128_u16_asm.py -- do not edit.\n")
            f.write(G.HEADER % params)
            f.write("\n".join(em.lines) + "\n")
            f.write(G.FOOTER % params)
        print("wrote %s: %d VALU instructions (static), %d lines" % (out, em.n_valu, len(em.lines)))


if __name__ == "__main__":
    main()
