"""32768-word rows: ONE operand register-resident in a 1024-thread workgroup (transforms, the composed product's two kernels,
the fused inverse pipelines and the int8 forward pipelines)."""
import os

from . import state as cfg
from .emitter import Emitter, VmCounter, run_pairs, vp
from .arith import T, canon, ct_bfly, final_bfly, gs_bfly, pointwise
from .twiddles import Ring
from .block4096 import lane_contig_setup, lds_read, lds_write
from .fused import fma_job, fms_job
from .rows import prologue16k

# ------------------------------------------------------------------ 32768-word rows: ONE operand register-resident
# A 32768-word row (256 KiB) is exactly the register footprint the 16384-word kernel manages for TWO operands: one
# 1024-thread workgroup, 32 words per thread in the two coefficient files v[V_A..] / v[V_B..] (64 VGPRs), one butterfly at
# a time, the twiddle records streaming through the 9-slot ring: 128 VGPRs, 4 waves per SIMD.  The row is HBM traffic
# exactly once per direction:
#   F0  radix-8 pass over all 32 slots (global stages r-3 .. r-1): thread tid holds x[tid + 1024 k], k = c + 4 m, i.e.
#       four columns c of the eight 4096-word blocks m; file A = blocks 0..3, file B = blocks 4..7
#   X0  through LDS in TWO rounds (a 32768-word row does not fit the 160 KiB): file A -> the four sub-groups' slabs ->
#       file A of sub-group q = block q; then file B -> block q + 4.  Same addresses as the 16384-word kernel's X0.
#   F1 F2 F3 / I1 I2 I3: the 4096-word passes of the block kernel, once per file (the files are different blocks of ONE
#       row here, so they do not share twiddles: file B's records are the block q + 4 ones, K offset by a constant)
#   X0' in two rounds, I0 radix-8 with the mirrored table, n^-1 folded into the last stage when the row is the whole row.
# kinds: fwd (canonical NTT-form words out), inv, polymul_ntt: c = INTT(NTT(a) (.) b') with b' (already transformed,
# canonical) STREAMED through the twiddle ring during the point-wise step -- the large-row product is then
# b' = fwd(b) (read + write) followed by polymul_ntt(a, b') (two reads + one write): 5 operand passes instead of 9.
def build_row32k(kind="fwd", level=0):
    assert cfg.ROW_G == 4 and cfg.ROW_LG == 3 and cfg.NEXT_VGPR == 128
    em = Emitter()
    vm = VmCounter(em)
    R = em.raw
    FILES = ((cfg.V_A, 0), (cfg.V_B, 4))                      # (register base, block offset inside the row)
    DK = {"F1": 1, "F2": 16, "F3": 256, "I1": -256, "I2": -16, "I3": -1}   # dK / d(block) of the pass constants
    inner = {"F1": (cfg.S_K["F1"], None, False), "F2": (cfg.S_K["F2"], cfg.V_BIDX, False), "F3": (cfg.S_K["F3"], cfg.V_TID, False),
             "I1": (cfg.S_K["I1"], cfg.V_TID, True), "I2": (cfg.S_K["I2"], cfg.V_BIDX, True), "I3": (cfg.S_K["I3"], None, True)}
    passes = {"F0": (cfg.S_K0["F0"], None, False), "I0": (cfg.S_K0["I0"], None, True)}
    for f, (_, boff) in enumerate(FILES):
        for name, (kreg, vidx, desc) in inner.items():
            passes[name + "ab"[f]] = (kreg, vidx, desc, DK[name] * boff)

    # b' of the composed product lives in the context's scratch in a layout of OUR choice ("_s" kinds): block-major, then the
    # slot pair i, then the thread -- [block q + boff][i][t] x 16 bytes -- so that the 64 lanes of a wave store / fetch 64
    # consecutive 16-byte pairs (8 cache lines).  In the reference's order (the user-visible one: kinds without "_s") thread t
    # owns words 16t .. 16t + 15, i.e. a lane's pair sits alone in its 128-byte line: 64 lines per load, 16 bytes used of each,
    # half of the product kernel's L1 fills -- and the forward kernel pays two LDS transposes to produce it with coalesced stores.
    scratch_layout = kind.endswith("_s")
    kind = kind[:-2] if scratch_layout else kind
    if scratch_layout:
        kind = {"polymul": "polymul_ntt"}.get(kind, kind)
    # level 2 (the "_s" pair only -- b' is OURS, so it may be stored incomplete): both forward transforms stop two stages early (F3 keeps
    # its sub-stages 0 and 1; fwd_s stores the words as the butterflies leave them, no canonical step), the point-wise step becomes the
    # base multiplication mod X^4 -+ zeta (incomplete.py) on a's registers and the streamed b' -- a group of four words is two ring slots,
    # zeta the two records of F3's sub-stage 1, which stay in the ring through the group products of their file, the scratch three
    # RESERVED slots: 2 + 3 + two groups of b' in flight = the nine slots -- and the inverse starts two stages late (I1 runs its
    # sub-stages 1 and 0).  The host hands both kernels of the pair the level-2 ModConst records ((n / 4)^-1, the mu2 field).
    assert level in (0, 2) and (not level or (scratch_layout and kind in ("fwd", "polymul_ntt") and cfg.SPLIT32K and cfg.SINGLE_STREAM))
    f3_stages = (0, 1) if level else (0, 1, 2, 3)
    i1_stages = (1, 0) if level else (3, 2, 1, 0)

    def bprime_loader(boff):
        def load(em_, r, s_, i, first):               # words 16t + 2i, 16t + 2i + 1 of block q + boff of b' -> one ring slot
            if first:
                em_.raw("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
                if boff:
                    em_.raw("s_add_u32 s42, s42, 0x%x" % (boff << 15,))
                em_.raw("s_add_u32 s96, s18, s42")
                em_.raw("s_addc_u32 s97, s19, 0")
                em_.valu("v_lshlrev_b32_e32 v%d, %d, v%d" % (cfg.V_TWO, 4 if scratch_layout else 7, cfg.V_TID))
            if scratch_layout:
                em_.raw("s_add_u32 s86, s96, 0x%x" % (4096 * i,))
                em_.raw("s_addc_u32 s87, s97, 0")
                return "global_load_dwordx4 v[%d:%d], v%d, s[86:87] nt" % (r, r + 3, cfg.V_TWO)
            return "global_load_dwordx4 v[%d:%d], v%d, s[96:97] offset:%d" % (r, r + 3, cfg.V_TWO, 16 * i)
        return load
    passes["Ba"], passes["Bb"] = bprime_loader(0), bprime_loader(4)

    compact_x = kind in ("fwd_i8", "fma_fwd_i8", "enc2_i8")   # the row arrives as one signed byte per coefficient (a compact Gaussian polynomial)
    # fma_fwd_i8 / enc2_i8: the transformed row X never leaves the registers -- out0 = X k0 + e0' [, out1 = X k1 + e1'] with the key rows
    # (one polynomial for the batch) and the already transformed noise rows e' streamed through the ring's registers in the store layout
    enc_res = {"fma_fwd_i8": 1, "enc2_i8": 2}.get(kind, 0)
    if compact_x:
        kind = "fwd"
    fused_inv = kind in ("fms_inv", "fma_inv")       # INTT(b - a k) / INTT(b + a k): a at S_AROW, b at S_BROW, the key row at s[98:99]
    has_fwd, has_inv = kind != "inv" and not fused_inv, kind != "fwd"
    uses = []
    if has_fwd:
        uses += [("F0", s_, g) for s_ in range(3) for g in range(1 << s_)]
        for name in ("F1", "F2", "F3"):
            for f in range(2):
                uses += [(name + "ab"[f], s_, g) for s_ in (f3_stages if name == "F3" else range(4)) for g in range(1 << s_)]
                if name == "F3" and level and kind == "polymul_ntt":     # the file's group products follow its F3 at once
                    uses += [("TMP", f, k) for k in range(3)] + [("B" + "ab"[f], 0, i) for i in range(8)]
    if kind == "polymul_ntt" and not level:
        for f in range(2):
            uses += [("B" + "ab"[f], 0, i) for i in range(8)]
    if has_inv:
        for name in ("I1", "I2", "I3"):
            for f in range(2):
                uses += [(name + "ab"[f], s_, g) for s_ in (i1_stages if name == "I1" else (3, 2, 1, 0)) for g in range(1 << s_)]
        uses += [("I0", s_, g) for s_ in (2, 1, 0) for g in range(1 << s_)]
    if level:
        passes["TMP"] = None           # reserved ring slots: four scratch registers each, nothing is loaded
    ring = Ring(em, vm, cfg.RING_SLOTS, uses, passes)
    if cfg.BPRIME_ALIAS and scratch_layout:   # fwd_s writes b' through s20, polymul_ntt_s reads it through s18
        cfg.ALIAS_ROWS = (20,) if kind == "fwd" else (18,)
    prologue16k(em, vm, None, "none", key_row=fused_inv, compact_x=compact_x)
    cfg.ALIAS_ROWS = ()
    AX = T(0, 0)   # exchange address scratch (the butterfly temporaries are idle during exchanges)

    def block_base(srow, boff):                           # s[86:87] = first word of block q + boff of the row at srow
        lo, hi = srow[2:-1].split(":")
        R("s_lshl_b32 s42, %s, 15" % (cfg.S_Q,))
        if boff:
            R("s_add_u32 s42, s42, 0x%x" % (boff << 15,))
        R("s_add_u32 s86, s%s, s42" % lo)
        R("s_addc_u32 s87, s%s, 0" % hi)

    n_row_loads = 0
    if has_fwd:
        em.comment("the row: x[tid + 1024 k] -> slot k (8 KiB contiguous per workgroup load)")
        R("s_mov_b64 s[86:87], %s" % (cfg.S_AROW,))
        if compact_x:
            em.valu("v_lshrrev_b32_e32 v%d, 3, v%d" % (T(0, 0), cfg.V_OFF8))        # tid (V_TID is the index inside the sub-group)
            seq = None
            for k in range(32):
                seq = vm.load("global_load_sbyte v%d, v%d, s[86:87]" % (cfg.V_A + 2 * k, T(0, 0)))
                if k < 31:
                    R("s_add_u32 s86, s86, 0x400")
                    R("s_addc_u32 s87, s87, 0")
            vm.wait(seq)
            em.comment("x >= 0 stays, x < 0 becomes p + x: any word congruent to the coefficient is a legal input of the first butterfly")
            t = T(0, 4)
            for k in range(32):
                x = cfg.V_A + 2 * k
                em.valu("v_ashrrev_i32_e32 v%d, 31, v%d" % (x + 1, x))
                em.valu("v_and_b32_e32 v%d, s24, v%d" % (t, x + 1))
                em.valu("v_and_b32_e32 v%d, s25, v%d" % (t + 1, x + 1))
                em.valu("v_lshl_add_u64 %s, %s, 0, %s" % (vp(x), vp(x), vp(t)))
        for k in range(32 if not compact_x else 0):
            vm.load("global_load_dwordx2 %s, v%d, s[86:87] nt" % (vp(cfg.V_A + 2 * k), cfg.V_OFF8))
            if k < 31:
                R("s_add_u32 s86, s86, 0x2000")
                R("s_addc_u32 s87, s87, 0")
    else:
        em.comment("NTT-form words: block element 1024w + 64j + l -> pair j of the block's file (512 B per wave load)")
        g_, _ = lane_contig_setup(em)
        for base, boff in FILES:
            block_base(cfg.S_AROW, boff)
            for j in range(16):
                vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d nt" % (vp(base + 2 * j), g_, (j & 7) * 512))
                if j == 7:
                    R("s_add_u32 s86, s86, 0x1000")
                    R("s_addc_u32 s87, s87, 0")
    if fused_inv:
        em.comment("a <- fold(b -+ a k) word by word, in the load layout (the operation is element-wise): b and the key stream through"
                   " the twiddle ring's registers, eight words of each at a time, before the ring is primed")
        for f, (base, boff) in enumerate(FILES):
            for half in range(2):
                g_, _ = lane_contig_setup(em)
                seq = None
                for srow, dst0 in ((cfg.S_BROW, cfg.V_TW), ("s[98:99]", cfg.V_TW + 16)):
                    block_base(srow, boff)
                    if half:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
                    for jj in range(8):
                        seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst0 + 2 * jj), g_, jj * 512, " nt" if srow == cfg.S_BROW else ""))
                vm.wait(seq)
                run_pairs(em, [fms_job(base + 2 * (8 * half + jj), cfg.V_TW + 16 + 2 * jj, cfg.V_TW + 2 * jj, kind == "fms_inv") for jj in range(8)])
    n_row_loads = vm.issued
    ring.prime()

    def fwd_pass(name):
        for f, (base, _) in enumerate(FILES):
            nm_ = name + "ab"[f]
            em.comment("%s, file %s" % (name, "AB"[f]))
            for s_ in range(4):
                half = 8 >> s_
                for g in range(1 << s_):
                    tw = ring.get((nm_, s_, g))
                    run_pairs(em, [ct_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done((nm_, s_, g))

    def inv_pass(name):
        for f, (base, _) in enumerate(FILES):
            nm_ = name + "ab"[f]
            em.comment("%s, file %s" % (name, "AB"[f]))
            for s_ in (3, 2, 1, 0):
                half = 8 >> s_
                for g in range(1 << s_):
                    tw = ring.get((nm_, s_, g))
                    run_pairs(em, [gs_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done((nm_, s_, g))

    # ---- split-phase exchanges: the two files are independent between F0 and I0, so every LDS batch of one file (the
    # writes of an exchange, or its reads) is issued in FRONT of a half pass of arithmetic on the OTHER file and waited for
    # behind it; the arithmetic order -- and with it the order in which the ring consumes its records -- is unchanged.
    # Only the first forward round (file A after F0) and the last inverse round (file B before I0) stay exposed.
    W0 = "s_waitcnt lgkmcnt(0)"

    def fwd_stages(f, name, stages):
        base, nm_ = FILES[f][0], name + "ab"[f]
        em.comment("%s, file %s, stages %s" % (name, "AB"[f], stages))
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((nm_, s_, g))
                run_pairs(em, [ct_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                if not (level and kind == "polymul_ntt" and name == "F3" and s_ == 1):    # (zeta: released by group_products)
                    ring.done((nm_, s_, g))

    def group_products(f):
        """level 2, the product kernel: file f = a' (four groups of four words per thread) times the streamed b' modulo X^4 -+ zeta"""
        if not (level and kind == "polymul_ntt"):
            return
        from .incomplete import base_mul
        base = FILES[f][0]
        em.comment("base multiplication mod X^4 -+ zeta, file %s: b' streams through the ring two slots per group; scratch = three reserved slots" % "AB"[f])
        tmp = [int(ring.get(("TMP", f, k))[0][1:]) for k in range(3)]
        rt = [tmp[0], tmp[0] + 2, tmp[1]]
        for g4 in range(4):
            u0, u1 = ("B" + "ab"[f], 0, 2 * g4), ("B" + "ab"[f], 0, 2 * g4 + 1)
            ring.get(u0)
            ring.get(u1)
            r0, r1 = cfg.V_TW + 4 * ring.slot_of[u0], cfg.V_TW + 4 * ring.slot_of[u1]
            tw = ring.regs(("F3" + "ab"[f], 1, g4 // 2))
            run_pairs(em, [base_mul([base + 8 * g4 + 2 * i for i in range(4)], [r0, r0 + 2, r1, r1 + 2], 4, tw, bool(g4 & 1), [rt, rt], [tmp[2], tmp[2]])])
            ring.done(u0)
            ring.done(u1)
        for g in range(2):
            ring.done(("F3" + "ab"[f], 1, g))
        for k in range(3):
            ring.done(("TMP", f, k))

    def inv_stages(f, name, stages):
        base, nm_ = FILES[f][0], name + "ab"[f]
        em.comment("%s, file %s, stages %s" % (name, "AB"[f], stages))
        for s_ in stages:
            half = 8 >> s_
            for g in range(1 << s_):
                tw = ring.get((nm_, s_, g))
                run_pairs(em, [gs_bfly(base + 2 * (g * 2 * half + h), base + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                ring.done((nm_, s_, g))

    def X0w(f):
        base = FILES[f][0]
        em.comment("X0 writes, file %s: thread (q, t) slot 4*m + c -> sub-group m, thread t, slot q + 4*c" % "AB"[f])
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
        for k in range(16):
            qq, j = k // 4, k % 4
            R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * cfg.SLAB_BYTES + j * 8192))

    def X0r(f):
        base = FILES[f][0]
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))

    def X0iw(f):
        base = FILES[f][0]
        em.comment("X0' writes, file %s: thread (q, t) slot g + 4*j -> thread (g, t) slot 4*q + j, layout [slot][tid]" % "AB"[f])
        R("s_lshl_b32 s86, %s, 15" % (cfg.S_Q,))
        em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
        em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                          # q*32768 + t*8
        for k in range(16):
            g_, j = k % 4, k // 4
            R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(base + 2 * k), j * 8192 + g_ * 2048))

    def X0ir(f):
        base = FILES[f][0]
        em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * 8192, cfg.V_OFF8))
        for k in range(16):
            R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), cfg.V_OFF8 if k < 8 else AX, (k & 7) * 8192))

    def seq(*items):          # strings are emitted as they are, callables are called
        for it in items:
            if isinstance(it, str):
                R(it)
            else:
                it()

    def split_phase_schedule():
        A_, B_ = FILES[0][0], FILES[1][0]
        BAR = "s_barrier"
        if has_fwd:
            em.comment("F0: radix-8 over the 32 slots (stage 0 couples the files)")
            for s_ in range(3):
                half = 16 >> s_
                for g in range(1 << s_):
                    tw = ring.get(("F0", s_, g))
                    run_pairs(em, [ct_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done(("F0", s_, g))
            seq(lambda: X0w(0), W0, BAR, lambda: X0r(0), W0, BAR,
                lambda: X0w(1), lambda: fwd_stages(0, "F1", (0, 1)), W0, BAR, lambda: X0r(1), lambda: fwd_stages(0, "F1", (2, 3)), W0, BAR,
                lambda: lds_write(em, cfg.V_L1W, A_, 2176), lambda: fwd_stages(1, "F1", (0, 1)), W0, BAR,
                lambda: lds_read(em, cfg.V_L1R, A_, 136), lambda: fwd_stages(1, "F1", (2, 3)), W0, BAR,
                lambda: lds_write(em, cfg.V_L1W, B_, 2176), lambda: fwd_stages(0, "F2", (0, 1)), W0, BAR,
                lambda: lds_read(em, cfg.V_L1R, B_, 136), lambda: fwd_stages(0, "F2", (2, 3)), W0,
                # E2 is wave-local (LDS is in order per wave): file A's transposes run under F2 of file B, B's under F3 of A
                lambda: lds_write(em, cfg.V_L1R, A_, 136), lambda: lds_read(em, cfg.V_L2R, A_, 8), lambda: fwd_stages(1, "F2", (0, 1, 2, 3)), W0,
                lambda: lds_write(em, cfg.V_L1R, B_, 136), lambda: lds_read(em, cfg.V_L2R, B_, 8), lambda: fwd_stages(0, "F3", f3_stages),
                lambda: group_products(0), W0, lambda: fwd_stages(1, "F3", f3_stages), lambda: group_products(1))
        if kind == "fwd" and scratch_layout:
            em.comment("canonical words straight into the product's scratch layout [block][pair i][thread]: no transposes")
            em.valu("v_lshlrev_b32_e32 v%d, 4, v%d" % (cfg.V_TWO, cfg.V_TID))
            for base, boff in FILES:
                if not level:          # (level 2: the product's base multiplication folds whatever word it is handed)
                    run_pairs(em, [canon(base + 2 * i) for i in range(16)])
                block_base(cfg.S_CROW, boff)
                for i in range(8):
                    R("global_store_dwordx4 v%d, v[%d:%d], s[86:87] nt" % (cfg.V_TWO, base + 4 * i, base + 4 * i + 3))
                    if i < 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            R("s_endpgm")
            return True
        if kind == "fwd" and enc_res:
            em.comment("X in the store layout (a wave-local LDS transpose per file), then per result: key and noise words in, X k + e' out")
            R("s_load_dwordx8 s[88:95], s[0:1], 0x30")                      # k0 k1 e1' out1 (an aligned group of eight)
            R("s_sub_u32 s42, s20, s4")                                     # the dense rows' offset (this element, this modulus)
            R("s_subb_u32 s43, s21, s5")
            R("s_mov_b32 s96, s3")                                          # the key rows' offset: modulus cm of ONE polynomial (n = 32768)
            R("s_mov_b32 s97, 0")
            R("s_lshl_b64 s[96:97], s[96:97], 18")
            R("s_waitcnt lgkmcnt(0)")
            for lo in (88, 90):
                R("s_add_u32 s%d, s%d, s96" % (lo, lo))
                R("s_addc_u32 s%d, s%d, s97" % (lo + 1, lo + 1))
            for lo in (92, 94):
                R("s_add_u32 s%d, s%d, s42" % (lo, lo))
                R("s_addc_u32 s%d, s%d, s43" % (lo + 1, lo + 1))
            def transposes(base):
                lds_write(em, cfg.V_L2R, base, 8)
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, cfg.S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
            def fma_stores(base, boff):
                for res in range(enc_res):
                    krow, erow, orow = (("s[88:89]", cfg.S_BROW, cfg.S_CROW), ("s[90:91]", "s[92:93]", "s[94:95]"))[res]
                    for half in range(2):
                        g_, _ = lane_contig_setup(em)
                        seq = None
                        for srow, dst0, nt_ in ((krow, cfg.V_TW, ""), (erow, cfg.V_TW + 16, " nt")):
                            block_base(srow, boff)
                            if half:
                                R("s_add_u32 s86, s86, 0x1000")
                                R("s_addc_u32 s87, s87, 0")
                            for jj in range(8):
                                seq = vm.load("global_load_dwordx2 %s, v%d, s[86:87] offset:%d%s" % (vp(dst0 + 2 * jj), g_, jj * 512, nt_))
                        vm.wait(seq)
                        run_pairs(em, [fma_job(cfg.V_TW + 2 * jj, base + 2 * (8 * half + jj), cfg.V_TW + 16 + 2 * jj, res == 0) for jj in range(8)])
                        g_, _ = lane_contig_setup(em)
                        block_base(orow, boff)
                        if half:
                            R("s_add_u32 s86, s86, 0x1000")
                            R("s_addc_u32 s87, s87, 0")
                        for jj in range(8):
                            R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(cfg.V_TW + 2 * jj), jj * 512))
            transposes(A_)
            R(W0)
            transposes(B_)
            fma_stores(A_, FILES[0][1])
            R(W0)
            fma_stores(B_, FILES[1][1])
            R("s_endpgm")
            return True
        if kind == "fwd":
            em.comment("canonical words, then a wave-local LDS transpose per file so the stores are fully coalesced; file B's"
                       " reduction runs under file A's transposes")
            def stores(base, boff):
                g_, _ = lane_contig_setup(em)
                block_base(cfg.S_CROW, boff)
                for j in range(16):
                    R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(base + 2 * j), (j & 7) * 512))
                    if j == 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            def transposes(base):
                lds_write(em, cfg.V_L2R, base, 8)
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, cfg.S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
            run_pairs(em, [canon(A_ + 2 * i) for i in range(16)])
            transposes(A_)
            run_pairs(em, [canon(B_ + 2 * i) for i in range(16)])
            R(W0)
            stores(A_, FILES[0][1])
            transposes(B_)
            R(W0)
            stores(B_, FILES[1][1])
            R("s_endpgm")
            return True
        if kind == "polymul_ntt" and level:
            seq(lambda: inv_stages(0, "I1", i1_stages))
        elif kind == "polymul_ntt":
            em.comment("point-wise product with b' streamed through the ring: slot i of a file = words 16t + 2i, 16t + 2i + 1 of its block")
            for f, (base, _) in enumerate(FILES):
                for i in range(8):
                    use = ("B" + "ab"[f], 0, i)
                    ring.get(use)
                    r = cfg.V_TW + 4 * ring.slot_of[use]
                    run_pairs(em, [pointwise(base + 4 * i, r, True, False), pointwise(base + 4 * i + 2, r + 2, True, False)])
                    ring.done(use)
            seq(lambda: inv_stages(0, "I1", (3, 2, 1, 0)))
        else:
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_row_loads))          # the block loads have landed
            em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region; file B's under I1 of file A")
            def to_threads(base):
                _, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, cfg.S_SLAB, l_))
                for j in range(16):
                    R("ds_write_b64 v%d, %s offset:%d" % (l_, vp(base + 2 * j), 544 * j))
                lds_read(em, cfg.V_L2R, base, 8)
            seq(lambda: to_threads(A_), W0, lambda: to_threads(B_), lambda: inv_stages(0, "I1", (3, 2, 1, 0)), W0)
        seq(# E2' is wave-local: file A's under I1 of file B, file B's under I2 of file A
            lambda: lds_write(em, cfg.V_L2R, A_, 8), lambda: lds_read(em, cfg.V_L1R, A_, 136), lambda: inv_stages(1, "I1", i1_stages), W0,
            lambda: lds_write(em, cfg.V_L2R, B_, 8), lambda: lds_read(em, cfg.V_L1R, B_, 136), lambda: inv_stages(0, "I2", (3, 2, 1, 0)), W0,
            lambda: lds_write(em, cfg.V_L1R, A_, 136), lambda: inv_stages(1, "I2", (3, 2)), W0, BAR,
            lambda: lds_read(em, cfg.V_L1W, A_, 2176), lambda: inv_stages(1, "I2", (1, 0)), W0, BAR,
            lambda: lds_write(em, cfg.V_L1R, B_, 136), lambda: inv_stages(0, "I3", (3, 2)), W0, BAR,
            lambda: lds_read(em, cfg.V_L1W, B_, 2176), lambda: inv_stages(0, "I3", (1, 0)), W0, BAR,
            lambda: X0iw(0), lambda: inv_stages(1, "I3", (3, 2)), W0, BAR, lambda: X0ir(0), lambda: inv_stages(1, "I3", (1, 0)), W0, BAR,
            lambda: X0iw(1), W0, BAR, lambda: X0ir(1), W0)
        return False

    if cfg.SPLIT32K:
        if split_phase_schedule():
            return em
    else:
        if has_fwd:
            em.comment("F0: radix-8 over the 32 slots (stage 0 couples the files)")
            for s_ in range(3):
                half = 16 >> s_
                for g in range(1 << s_):
                    tw = ring.get(("F0", s_, g))
                    run_pairs(em, [ct_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
                    ring.done(("F0", s_, g))
            for i, (base, _) in enumerate(FILES):
                em.comment("X0 round %d: thread (q, t) slot 4*m + c of this file -> sub-group m, thread t, slot q + 4*c" % i)
                if i:
                    R("s_barrier")       # WAR: the slabs are still being read for the previous file
                em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 2 * cfg.SLAB_BYTES, cfg.V_OFF8))
                for k in range(16):
                    qq, j = k // 4, k % 4
                    R("ds_write_b64 v%d, %s offset:%d" % (cfg.V_OFF8 if qq < 2 else AX, vp(base + 2 * k), (qq & 1) * cfg.SLAB_BYTES + j * 8192))
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (AX, cfg.S_SLAB, AX))
                for k in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), AX, 2048 * k))
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F1")
            for base, _ in FILES:
                em.comment("E1")
                R("s_barrier")           # WAR against the previous exchange through this slab
                lds_write(em, cfg.V_L1W, base, 2176)
                R("s_waitcnt lgkmcnt(0)")
                R("s_barrier")
                lds_read(em, cfg.V_L1R, base, 136)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F2")
            em.comment("E2: wave-local 16-lane transposes (LDS is in order per wave)")
            for base, _ in FILES:
                lds_write(em, cfg.V_L1R, base, 136)
                lds_read(em, cfg.V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
            fwd_pass("F3")
        if kind == "fwd":
            em.comment("canonical words, then a wave-local LDS transpose per file so the stores are fully coalesced")
            run_pairs(em, [canon(cfg.V_A + 2 * i) for i in range(32)])
            for base, boff in FILES:
                lds_write(em, cfg.V_L2R, base, 8)
                g_, l_ = lane_contig_setup(em)
                em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, cfg.S_SLAB, l_))
                for j in range(16):
                    R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * j), l_, 544 * j))
                R("s_waitcnt lgkmcnt(0)")
                block_base(cfg.S_CROW, boff)
                for j in range(16):
                    R("global_store_dwordx2 v%d, %s, s[86:87] offset:%d nt" % (g_, vp(base + 2 * j), (j & 7) * 512))
                    if j == 7:
                        R("s_add_u32 s86, s86, 0x1000")
                        R("s_addc_u32 s87, s87, 0")
            R("s_endpgm")
            return em

        if kind == "polymul_ntt":
            em.comment("point-wise product with b' streamed through the ring: slot i of a file = words 16t + 2i, 16t + 2i + 1 of its block")
            for f, (base, _) in enumerate(FILES):
                for i in range(8):
                    use = ("B" + "ab"[f], 0, i)
                    ring.get(use)
                    r = cfg.V_TW + 4 * ring.slot_of[use]
                    run_pairs(em, [pointwise(base + 4 * i, r, True, False), pointwise(base + 4 * i + 2, r + 2, True, False)])
                    ring.done(use)
        else:
            R("s_waitcnt vmcnt(%d)" % (vm.issued - n_row_loads))          # the block loads have landed
            em.comment("lane-contiguous -> thread-contiguous through the wave's own LDS region, file by file")
            _, l_ = lane_contig_setup(em)
            em.valu("v_add_u32_e32 v%d, %s, v%d" % (l_, cfg.S_SLAB, l_))
            for base, _ in FILES:
                for j in range(16):
                    R("ds_write_b64 v%d, %s offset:%d" % (l_, vp(base + 2 * j), 544 * j))
                lds_read(em, cfg.V_L2R, base, 8)
                R("s_waitcnt lgkmcnt(0)")
        inv_pass("I1")
        em.comment("E2'")
        for base, _ in FILES:
            lds_write(em, cfg.V_L2R, base, 8)
            lds_read(em, cfg.V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
        inv_pass("I2")
        for i, (base, _) in enumerate(FILES):
            em.comment("E1'")
            if i:
                R("s_barrier")           # WAR: the slab is still being read for the previous file
            lds_write(em, cfg.V_L1R, base, 136)
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            lds_read(em, cfg.V_L1W, base, 2176)
            R("s_waitcnt lgkmcnt(0)")
        inv_pass("I3")
        for i, (base, _) in enumerate(FILES):
            em.comment("X0' round %d: thread (q, t) slot g + 4*j of this file -> thread (g, t) slot 4*q + j, layout [slot][tid]" % i)
            R("s_barrier")               # every wave is done reading the previous exchange
            R("s_lshl_b32 s86, %s, 15" % (cfg.S_Q,))
            em.valu("v_lshlrev_b32_e32 v%d, 3, v%d" % (AX, cfg.V_TID))
            em.valu("v_add_u32_e32 v%d, s86, v%d" % (AX, AX))                          # q*32768 + t*8
            for k in range(16):
                g_, j = k % 4, k // 4
                R("ds_write_b64 v%d, %s offset:%d" % (AX, vp(base + 2 * k), j * 8192 + g_ * 2048))
            R("s_waitcnt lgkmcnt(0)")
            R("s_barrier")
            em.valu("v_add_u32_e32 v%d, 0x%x, v%d" % (AX, 8 * 8192, cfg.V_OFF8))
            for k in range(16):
                R("ds_read_b64 %s, v%d offset:%d" % (vp(base + 2 * k), cfg.V_OFF8 if k < 8 else AX, (k & 7) * 8192))
            R("s_waitcnt lgkmcnt(0)")
    em.comment("I0: radix-8 over the 32 slots, mirrored table")
    for s_ in (2, 1):
        half = 16 >> s_
        for g in range(1 << s_):
            tw = ring.get(("I0", s_, g))
            run_pairs(em, [gs_bfly(cfg.V_A + 2 * (g * 2 * half + h), cfg.V_A + 2 * (g * 2 * half + h + half), tw) for h in range(half)])
            ring.done(("I0", s_, g))
    R("s_cmp_eq_u32 s88, %d" % cfg.ROW_LG)
    R("s_cbranch_scc1 .Lmerged_last_stage")
    em.comment("r > 3: plain global stage r-3; lazy output for the outer inverse passes")
    tw = ring.get(("I0", 0, 0))
    run_pairs(em, [gs_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 16), tw) for h in range(16)])
    R("s_branch .Lstore")
    em.lines.append(".Lmerged_last_stage:")
    em.comment("n == 32768: stage 0 with n^-1 folded in")
    R("s_waitcnt vmcnt(0)")
    run_pairs(em, [final_bfly(cfg.V_A + 2 * h, cfg.V_A + 2 * (h + 16)) for h in range(16)])
    em.lines.append(".Lstore:")
    R("s_mov_b64 s[86:87], %s" % (cfg.S_CROW,))
    for k in range(32):
        R("global_store_dwordx2 v%d, %s, s[86:87] nt" % (cfg.V_OFF8, vp(cfg.V_A + 2 * k)))
        if k < 31:
            R("s_add_u32 s86, s86, 0x2000")
            R("s_addc_u32 s87, s87, 0")
    R("s_endpgm")
    return em
