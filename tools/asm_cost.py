#!/usr/bin/env python3
"""ANALYSIS AID (CPU only): dynamic instruction counts of the generated assembly kernels, taken by executing them on the
gfx950 interpreter of tests/asm_emu.py, turned into the VALU-issue bound of each product kernel: a SIMD issues one VALU
instruction of a wave64 per ~4 cycles (profiles/r02_pmc_sq_A.txt), the chip has 1024 SIMDs at
~2.03 GHz under load.  Prints, per kernel: VALU / SALU / vector-memory / LDS instructions per product and the bound in
products per second; the measured rates are quoted next to it from profiles/r03_final_bench_*.json.
usage: tools/asm_cost.py > profiles/r02_valu_issue_model.txt"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import asm_emu                                 # noqa: E402
from nfllib_amd.params import params           # noqa: E402

SIMDS, CLOCK, CYCLES = 1024, 2.0e9, 3.86   # sclk under these kernels (1400 W package limit), issue cost of the butterfly mix (profiles/r03_ubench_issue.txt)
CSRC = os.path.join(ROOT, "nfllib_amd", "csrc")
asm_emu.STRICT = False
asm_emu.COUNT = True


def operands(bits, n, nm, batch):
    prm = params(bits)
    rng = np.random.default_rng(5)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a = (rng.integers(0, 1 << 62, size=(batch, nm, n), dtype=np.uint64) % P[None, :, None]).astype(prm.dtype)
    return prm, a


def measured(workload):
    try:
        with open(os.path.join(ROOT, "profiles", "r03_final_bench_%s.json" % workload)) as f:
            return json.loads(f.read().strip().splitlines()[-1])["value"]
    except Exception:
        return None


rows = []


lds_rows = []


def case(name, workload, nm_workload, run, products):
    asm_emu.Wave.counts = {}
    asm_emu.LDS_STATS.clear()
    run()
    c = asm_emu.Wave.counts
    for ins, (n_ins, cycles, extra) in sorted(asm_emu.LDS_STATS.items()):
        lds_rows.append((name, ins, n_ins, cycles, extra))
    per = {k: v / products for k, v in c.items()}           # per product of ONE modulus
    valu = per.get("valu", 0) * nm_workload
    bound = SIMDS * CLOCK / CYCLES / valu
    m = measured(workload)
    rows.append((name, workload, valu, per.get("salu", 0) * nm_workload, per.get("vmem", 0) * nm_workload,
                 per.get("lds", 0) * nm_workload, bound, m))


def block(stem, n, block_log, workload, nm_w, incomplete=0):
    prm, a = operands(64, n, 1, 1)
    case(stem, workload, nm_w, lambda: asm_emu.run_block_kernel(os.path.join(CSRC, stem + "_gfx950.s"), n, 1, prm, a, a, block_log,
                                                                incomplete=incomplete), 1)


def row(stem, bits, n, rows_per_wg, magic, workload, nm_w, batch, incomplete=0):
    prm, a = operands(bits, n, 1, batch)
    case(stem, workload, nm_w, lambda: asm_emu.run_row_kernel(os.path.join(CSRC, stem + "_gfx950.s"), bits, n, 1, prm, a, a, rows_per_wg, magic,
                                                              incomplete=incomplete), batch)


def pipe(stem, n, workload, nm_w, incomplete=0):
    prm, a = operands(64, n, 1, 1)
    case(stem, workload, nm_w, lambda: asm_emu.run_pipe_product(os.path.join(CSRC, stem + "_gfx950.s"), n, 1, prm, a, a, incomplete=incomplete), 1)


block("polymul4096nt", 4096, 12, "B", 4)
block("polymul4096i1", 4096, 12, "B", 4, incomplete=1)   # round 6: 1 / 2 stages dropped each way, base multiplication mod X^2 / X^4 -+ zeta
block("polymul4096i2", 4096, 12, "B", 4, incomplete=2)
block("polymul8192", 8192, 13, "G", 2)
block("polymul8192i2", 8192, 13, "G", 2, incomplete=2)
block("polymul16384", 16384, 14, "C", 8)
block("polymul16384i2", 16384, 14, "C", 8, incomplete=2)
def rows32k(workload, nm_w, sfx="s", incomplete=0):   # b' = NTT(b) into the scratch, then c = INTT(NTT(a) (.) b'): the register-resident pair of n = 32768
    prm, a = operands(64, 32768, 1, 1)
    k = lambda stem: asm_emu.run_block_kernel(os.path.join(CSRC, stem + "_gfx950.s"), 32768, 1, prm, a, a, 15, words_per_thread=32, incomplete=incomplete)
    case("fwd32768%s + polymul_ntt32768%s" % (sfx, sfx), workload, nm_w, lambda: (k("ntt_fwd32768" + sfx), k("polymul_ntt32768" + sfx)), 1)


rows32k("F", 2)
rows32k("F", 2, "si2", 2)                                  # round 6, third session: the pair on incomplete transforms
pipe("polymul_pipe65536nt", 65536, "E", 30)
pipe("polymul_pipe65536nti2", 65536, "E", 30, incomplete=2)
row("row1024_u32", 32, 1024, 4, True, "A", 1, 4)
row("row1024_l0_u64", 64, 1024, 4, True, "-", 2, 4)                   # round 6: 64-bit rows of 1024 / 2048 words, one / two waves per row
row("row1024_u64", 64, 1024, 4, True, "-", 2, 4, incomplete=2)
row("row2048_l0_u64", 64, 2048, 2, True, "-", 2, 2)
row("row2048_u64", 64, 2048, 2, True, "-", 2, 2, incomplete=2)
row("row128_u16", 16, 128, 32, False, "H", 1, 32)
row("row8_u32", 32, 8, 256, True, "T", 2, 256)

print("VALU-issue bound of the generated product kernels (dynamic counts from the interpreter; one VALU instruction per SIMD per %.2f"
      " cycles, %d SIMDs, %.2f GHz)" % (CYCLES, SIMDS, CLOCK / 1e9))
print("%-22s %-3s %12s %9s %9s %9s %14s %14s %7s" % ("kernel", "wl", "VALU/product", "SALU", "VMEM", "LDS", "bound [1/s]", "measured [1/s]", "ratio"))
for name, wl, valu, salu, vmem, lds, bound, m in rows:
    print("%-22s %-3s %12.0f %9.0f %9.0f %9.0f %14.4g %14s %7s" % (name, wl, valu, salu, vmem, lds, bound, "%.4g" % m if m else "-",
                                                                  "%.2f" % (m / bound) if m else "-"))
print("(wave-instructions per product of the workload's shape, all moduli; measured = profiles/r03_final_bench_<wl>.json)")
print()
print("LDS bank conflicts of the same runs (bank model of MI355X_MICROARCH.md: lane groups of 32 / 32 / 32 / 16 lanes, 32 / 32 / 64 / 32 banks")
print("for ds_read_b32 / ds_write_b32 / ds_read_b64 / ds_write_b64; extra = LDS cycles added by distinct addresses on one bank within a group)")
print("%-22s %-14s %12s %16s %12s" % ("kernel", "instruction", "wave-instr.", "group cycles", "extra"))
for name, ins, n_ins, cycles, extra in lds_rows:
    print("%-22s %-14s %12d %16d %12d" % (name, ins, n_ins, cycles, extra))
