"""Which kernels are generated, under which register map: the driver behind tools/gen_polymul_asm.py."""
import os

from . import state as cfg
from .block4096 import build
from .incomplete import build_incomplete
from .rows1k import ARGS_ROW, ARGS_ROW_FWD, ARGS_ROW_INV, build_row1k, build_row1k_fma_inv, build_row1k_fwd_fma
from .fused import build_fused, build_fused_rows
from .rows import build_row16k, build_row16k_loop
from .rows32k import build_row32k
from .pipe import build_pipe
from .objfile import ARGS_FUSED, ARGS_PIPE, ARGS_STD, emit_file

KERNELS_FUSED = {
    "enc2": ("fused_enc2_4096", "nflhip_fused_enc2_4096_asm"),
    "fma_fwd": ("fused_fma_fwd4096", "nflhip_fused_fma_fwd4096_asm"),
    "fms_inv": ("fused_fms_inv4096", "nflhip_fused_fms_inv4096_asm"),
    "fma_inv": ("fused_fma_inv4096", "nflhip_fused_fma_inv4096_asm"),
}


KERNELS = {   # kind -> (file suffix, kernel symbol)
    "polymul": ("polymul4096", "nflhip_polymul4096_asm"),
    "polymul_ntt": ("polymul_ntt4096", "nflhip_polymul_ntt4096_asm"),
    "fwd": ("ntt_fwd4096", "nflhip_ntt_fwd4096_asm"),
    "inv": ("ntt_inv4096", "nflhip_ntt_inv4096_asm"),
    "inv_mul": ("ntt_inv_mul4096", "nflhip_ntt_inv_mul4096_asm"),
    "fwd2": ("ntt_fwd4096x2", "nflhip_ntt_fwd4096x2_asm"),
    "inv2": ("ntt_inv4096x2", "nflhip_ntt_inv4096x2_asm"),
}


KERNELS16K = {
    "polymul": ("polymul16384", "nflhip_polymul16384_asm"),
    "polymul_ntt": ("polymul_ntt16384", "nflhip_polymul_ntt16384_asm"),
    "fwd": ("ntt_fwd16384", "nflhip_ntt_fwd16384_asm"),
    "inv": ("ntt_inv16384", "nflhip_ntt_inv16384_asm"),
}


def main():
    outdir = os.path.dirname(cfg.OUT)
    experiments = bool(os.environ.get("NFL_GEN_EXPERIMENTS"))   # also emit the variants that were measured and not kept
    nt = lambda em_: [l + " nt" if ("global_load_dwordx2" in l or "global_store_dwordx2" in l) else l for l in em_.lines]
    cfg.configure("pair")
    for kind, (stem, kname) in KERNELS.items():
        args = ARGS_STD + [("i32", 48)] if kind in ("fwd2", "inv2") else None
        if kind in ("polymul", "fwd2", "inv2"):
            # the n = 4096 product and the two-row transforms stream their coefficients with `nt` (+1 % on workload B)
            em_nt = build(kind)
            em_nt.lines = nt(em_nt)
            emit_file(os.path.join(outdir, stem + "nt_gfx950.s"), kname.replace("_asm", "nt_asm"), em_nt, args=args)
            if not experiments:
                continue
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, build(kind), args=args)
    # the metric product on incomplete transforms (round 6, incomplete.py): 1 or 2 stages dropped each way, base multiplication
    # mod X^2 / X^4 -+ zeta; same arguments, the host passes the ModConst records with (n / G)^-1 and the 2^127 Barrett constant
    for level in (1, 2):
        emi = build_incomplete(level)
        emi.lines = nt(emi)
        emit_file(os.path.join(outdir, "polymul4096i%d_gfx950.s" % level), "nflhip_polymul4096i%d_asm" % level, emi)
    # 64-bit rows of 1024 / 2048 words, one wave / two waves per row (rows1k.py, round 6): product on incomplete transforms,
    # stand-alone transforms; "l0" = the product on complete transforms (the A/B partner and the cross-check)
    for LB, words in ((4, 1024), (8, 2048)):
        for mode, level, sfx in (("polymul", 2, ""), ("polymul", 0, "_l0"), ("fwd", 0, "_fwd"), ("inv", 0, "_inv")):
            emit_file(os.path.join(outdir, "row%d%s_u64_gfx950.s" % (words, sfx)), "nflhip_row%d%s_u64_asm" % (words, sfx),
                      build_row1k(LB, mode, level), args=ARGS_ROW)
    # ... and the transform-fused pipelines on those rows (the LWE demo's bodies, one wave(s) per row)
    cfg.set(NEXT_SGPR=102)
    for LB, words in ((4, 1024), (8, 2048)):
        for sub, nm_ in ((True, "fmsinv"), (False, "fmainv")):
            emit_file(os.path.join(outdir, "row%d_%s_u64_gfx950.s" % (words, nm_)), "nflhip_row%d_%s_u64_asm" % (words, nm_),
                      build_row1k_fma_inv(LB, sub), args=ARGS_ROW_INV)
        for two, nm_ in ((True, "enc2"), (False, "fmafwd")):
            for fmt in ("w", "i8"):
                emit_file(os.path.join(outdir, "row%d_%s%s_u64_gfx950.s" % (words, nm_, fmt)), "nflhip_row%d_%s%s_u64_asm" % (words, nm_, fmt),
                          build_row1k_fwd_fma(LB, two, fmt), args=ARGS_ROW_FWD)
    cfg.set(NEXT_SGPR=96)
    # transform-fused pipelines (n = 4096): word-row streams `nt`, key rows and compact inputs through the caches
    cfg.set(NEXT_SGPR=102)
    for kind, (stem, kname) in KERNELS_FUSED.items():
        emf = build_fused(kind)
        emf.lines = [l + " nt" if "global_store_dwordx2" in l and not l.endswith(" nt") else l for l in emf.lines]   # (the inverse kinds' result rows)
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, emf, args=ARGS_FUSED)
    cfg.set(NEXT_SGPR=96)
    # n = 65536: the three-role pipeline kernel; coefficient streams `nt`: 3 x 15.7 MB of data per product pass through each
    # XCD's 4 MiB L2 exactly once, the 31 MB of twiddle tables are what is worth keeping there (+3 % on workload E)
    em_nt = build_pipe()
    em_nt.lines = nt(em_nt)
    emit_file(os.path.join(outdir, "polymul_pipe65536nt_gfx950.s"), "nflhip_polymul_pipe65536nt_asm", em_nt, args=ARGS_PIPE)
    em_b = build_pipe(b_ntt=True)     # operand b already transformed: two streaming roles per row, b' read block-wise as it lies
    em_b.lines = nt(em_b)
    emit_file(os.path.join(outdir, "polymul_pipe65536ntb_gfx950.s"), "nflhip_polymul_pipe65536ntb_asm", em_b, args=ARGS_PIPE)
    em_i = build_pipe(level=2)        # ... its block products on incomplete transforms (round 6)
    em_i.lines = nt(em_i)
    emit_file(os.path.join(outdir, "polymul_pipe65536nti2_gfx950.s"), "nflhip_polymul_pipe65536nti2_asm", em_i, args=ARGS_PIPE)
    if experiments:
        emit_file(os.path.join(outdir, "polymul_pipe65536_gfx950.s"), "nflhip_polymul_pipe65536_asm", build_pipe(), args=ARGS_PIPE)
        em15 = build_pipe(15)     # n = 32768 on the same kernel with radix-8 streaming roles (superseded by build_row32k)
        em15.lines = nt(em15)
        emit_file(os.path.join(outdir, "polymul_pipe32768_gfx950.s"), "nflhip_polymul_pipe32768_asm", em15, args=ARGS_PIPE)
    # one-launch variants: rows pinned to an XCD, intermediates through its L2 (fused_header); "l" = the pooled-scratch
    # experiment (measured, not kept)
    for lg, mod, sfx in ((16, "", ""), (15, "", "")) + (((16, "", "l"), (15, "", "l")) if experiments else ()):
        cfg.FUSED_LOADS = mod
        cfg.FUSED_LIFO = sfx == "l"
        emf = build_pipe(lg, fused=True)
        emit_file(os.path.join(outdir, "polymul_xcd%d%s_gfx950.s" % (1 << lg, sfx)), "nflhip_polymul_xcd%d%s_asm" % (1 << lg, sfx), emf,
                  args=ARGS_PIPE, lds=cfg.LDS_BYTES + 64)
    cfg.FUSED_LIFO = False
    for lg in (16, 15):               # ... and on incomplete transforms
        cfg.FUSED_LOADS = ""
        emf = build_pipe(lg, fused=True, level=2)
        emit_file(os.path.join(outdir, "polymul_xcd%di2_gfx950.s" % (1 << lg)), "nflhip_polymul_xcd%di2_asm" % (1 << lg), emf,
                  args=ARGS_PIPE, lds=cfg.LDS_BYTES + 64)
    build_pipe(16)   # (leave the module-level PIPE_LOGN as it was)
    ring = "ringpair" if os.environ.get("NFL_GEN_RINGPAIR") else "ring"
    cfg.configure(ring, 4)
    for kind, (stem, kname) in KERNELS16K.items():
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), kname, build_row16k(kind))
    emit_file(os.path.join(outdir, "polymul16384i2_gfx950.s"), "nflhip_polymul16384i2_asm", build_row16k("polymul", level=2))   # incomplete transforms (round 6)
    cfg.set(NEXT_SGPR=98)     # two rows of one modulus per workgroup on shared twiddle records (stand-alone forward transform)
    emit_file(os.path.join(outdir, "ntt_fwd16384x2_gfx950.s"), "nflhip_ntt_fwd16384x2_asm", build_row16k("fwd2"), args=ARGS_STD + [("i32", 48)])
    cfg.set(NEXT_SGPR=96)
    if experiments:            # persistent workgroups with row prefetch: measured -6.5 % (n = 16384) / -11 % (n = 8192), not kept
        cfg.set(NEXT_SGPR=102)
        emit_file(os.path.join(outdir, "polymul16384p_gfx950.s"), "nflhip_polymul16384p_asm", build_row16k_loop(),
                  args=ARGS_STD + [("i32", 48), ("i32", 52)])
    cfg.configure(ring, 2)         # 8192-word rows: two sub-groups, 512 threads, one radix-2 stage around the blocks
    for kind, (stem, kname) in KERNELS16K.items():
        emit_file(os.path.join(outdir, stem.replace("16384", "8192") + "_gfx950.s"), kname.replace("16384", "8192"),
                  build_row16k(kind))
    emit_file(os.path.join(outdir, "polymul8192i2_gfx950.s"), "nflhip_polymul8192i2_asm", build_row16k("polymul", level=2))
    cfg.set(NEXT_SGPR=98)
    emit_file(os.path.join(outdir, "ntt_fwd8192x2_gfx950.s"), "nflhip_ntt_fwd8192x2_asm", build_row16k("fwd2"), args=ARGS_STD + [("i32", 48)])
    cfg.set(NEXT_SGPR=96)
    if experiments:
        cfg.set(NEXT_SGPR=102)
        emit_file(os.path.join(outdir, "polymul8192p_gfx950.s"), "nflhip_polymul8192p_asm", build_row16k_loop(),
                  args=ARGS_STD + [("i32", 48), ("i32", 52)])
    # experiment (nflhip_debug_fused_grid(3)): the inverse pipelines of a 4096-word row on the ring-mode map (128 VGPRs: four
    # workgroups per CU instead of three, one butterfly at a time)
    cfg.configure("ring", 1)
    cfg.set(NEXT_SGPR=102, LDS_BYTES=cfg.SLAB_BYTES)
    for kind, (stem, kname) in KERNELS_FUSED.items():
        emit_file(os.path.join(outdir, stem + "r_gfx950.s"), kname.replace("_asm", "r_asm"), build_fused_rows(kind), args=ARGS_FUSED)
    # ... and the metric product itself on that map: measured same-box against nflhip_polymul4096nt_asm (profiles/
    # r04_ring_vs_pair_4096.txt): 2.89 ms per 16 384 products either way -- the product is bound by its arithmetic, a fourth
    # workgroup per CU buys nothing.  Only emitted with NFL_GEN_EXPERIMENTS=1.
    if experiments:
        emit_file(os.path.join(outdir, "fused_polymul4096r_gfx950.s"), "nflhip_fused_polymul4096r_asm", build_fused_rows("polymul"), args=ARGS_FUSED)
    cfg.set(NEXT_SGPR=96)
    # transform-fused pipelines on the row-resident map: rows of 16384 and 8192 words
    for groups, words in ((4, 16384), (2, 8192)):
        cfg.configure("ring", groups)
        cfg.set(NEXT_SGPR=102)
        for kind, (stem, kname) in KERNELS_FUSED.items():
            emit_file(os.path.join(outdir, stem.replace("4096", str(words)).replace("_%d" % words, "_%d" % words) + "_gfx950.s"),
                      kname.replace("4096", str(words)), build_fused_rows(kind), args=ARGS_FUSED)
        cfg.set(NEXT_SGPR=96)
    cfg.configure(ring, 4)
    # 32768-word rows: one operand register-resident in a 1024-thread workgroup (4 sub-groups x 2 blocks)
    cfg.set(ROW_LG=3, NEXT_SGPR=max(cfg.NEXT_SGPR, 98))
    for kind, stem in (("fwd", "ntt_fwd32768"), ("inv", "ntt_inv32768"), ("polymul_ntt", "polymul_ntt32768"),
                       ("fwd_s", "ntt_fwd32768s"), ("polymul_s", "polymul_ntt32768s")):
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), "nflhip_%s_asm" % stem, build_row32k(kind))
    for kind, stem in (("fwd_s", "ntt_fwd32768si2"), ("polymul_s", "polymul_ntt32768si2")):   # the composed product's pair on incomplete transforms (round 6)
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), "nflhip_%s_asm" % stem, build_row32k(kind, level=2))
    # ... and the fused inverse pipelines of such a row: INTT(b -+ a k), the key row's base and stride flag behind the standard arguments
    # ... the forward transform of a compact (int8) Gaussian polynomial: one byte per coefficient in, NTT words of every modulus out
    emit_file(os.path.join(outdir, "ntt_fwd32768i8_gfx950.s"), "nflhip_ntt_fwd32768i8_asm", build_row32k("fwd_i8"))
    for kind, stem in (("fma_fwd_i8", "fused_fma_fwd32768i8"), ("enc2_i8", "fused_enc2_32768i8")):   # ... and the forward pipelines on such a polynomial: out0 = NTT(x) k0 + e0' [, out1 = NTT(x) k1 + e1']
        emit_file(os.path.join(outdir, stem + "_gfx950.s"), "nflhip_%s_asm" % stem, build_row32k(kind),
                  args=ARGS_STD + [("ptr", 48), ("ptr", 56), ("ptr", 64), ("ptr", 72)])
    cfg.set(NEXT_SGPR=102)
    for kind in ("fms_inv", "fma_inv"):
        emit_file(os.path.join(outdir, "fused_%s32768_gfx950.s" % kind), "nflhip_fused_%s32768_asm" % kind, build_row32k(kind),
                  args=ARGS_STD + [("ptr", 48), ("i32", 56)])
    cfg.set(NEXT_SGPR=98)
    cfg.configure("ring", 4)
    if os.environ.get("NFL_DEBUG16K"):   # checkpoint variants for bisecting a fault: kernel ends after phase n
        kind = os.environ["NFL_DEBUG16K"]
        for n in range(-3, 10):
            emit_file("/tmp/dbg16k_p%d.s" % (n + 3), "nflhip_dbg16k_%d" % (n + 3), build_row16k(kind, stop=n))
