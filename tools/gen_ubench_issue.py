#!/usr/bin/env python3
"""Emit tools/ubench_issue_gen.inc for tools/ubench_issue.hip: every measured instruction stream as ONE inline-assembly
block on FIXED registers (separate `asm volatile` statements make hipcc pad every instruction with an s_nop, and AMDGPU
inline asm has no sub-register operand modifiers), plus the table of kernels the host side runs.

  op_<name>_<acc>      one opcode, `acc` independent accumulators (dependency distance = acc instructions), unrolled x8
  bfly32[i]_<acc>      the 7-instruction 30-bit Cooley-Tukey butterfly of gen_row1024_u32_asm.py
  bfly64[i][v]_<acc>   the 18-instruction 62-bit one of gen_polymul_asm.py; i = two butterflies interleaved instruction by
                       instruction (what the product kernel does); v = twiddle and constants read from VGPRs as in the kernel
                       (the plain variant keeps them in SGPRs)
  bfly64v_long<k>      the same butterflies as STRAIGHT-LINE code of k KiB (no loop inside; an outer loop repeats it): the
                       product kernel is ~50 KiB of straight-line code per wave, the instruction cache is 64 KiB per 2 CUs
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = 64          # first fixed VGPR


def pr(r):
    return "v[%d:%d]" % (r, r + 1)


# ---- single opcodes: (name, text with {d} = accumulator register (32-bit) or {D} = accumulator pair, {x} {y} = VGPR inputs,
#      %0 = an SGPR input), is64
OPS = [
    ("add", "v_add_u32_e32 v{d}, v{x}, v{d}", 0),
    ("add_sgpr", "v_add_u32_e32 v{d}, %0, v{d}", 0),
    ("sub", "v_sub_u32_e32 v{d}, v{x}, v{d}", 0),
    ("and", "v_and_b32_e32 v{d}, v{x}, v{d}", 0),
    ("mov", "v_mov_b32_e32 v{d}, v{x}", 0),
    ("lshl", "v_lshlrev_b32_e32 v{d}, 3, v{d}", 0),
    ("lshr", "v_lshrrev_b32_e32 v{d}, 3, v{d}", 0),
    ("min", "v_min_u32_e32 v{d}, v{x}, v{d}", 0),
    ("mul_lo", "v_mul_lo_u32 v{d}, v{x}, v{d}", 0),
    ("mul_hi", "v_mul_hi_u32 v{d}, v{x}, v{d}", 0),
    ("mad_u24", "v_mad_u32_u24 v{d}, v{x}, v{y}, v{d}", 0),
    ("lshl_add", "v_lshl_add_u32 v{d}, v{d}, 1, v{x}", 0),
    ("add3", "v_add3_u32 v{d}, v{x}, v{y}, v{d}", 0),
    ("add_co", "v_add_co_u32_e32 v{d}, vcc, v{x}, v{d}", 0),
    ("add_co_s", "v_add_co_u32_e64 v{d}, s[40:41], v{x}, v{d}", 0),
    ("cndmask_s", "v_cndmask_b32_e64 v{d}, v{x}, v{d}, s[42:43]", 0),
    ("mad64", "v_mad_u64_u32 {D}, vcc, v{x}, v{y}, {D}", 1),
    ("mad64_s", "v_mad_u64_u32 {D}, s[40:41], v{x}, %0, {D}", 1),
    ("mad64_0", "v_mad_u64_u32 {D}, s[40:41], v{x}, v{y}, 0", 1),
    ("lshl_add64", "v_lshl_add_u64 {D}, {D}, 0, v[60:61]", 1),
    ("fma_f64", "v_fma_f64 {D}, v[60:61], v[60:61], {D}", 1),
    ("pk_fma_f32", "v_pk_fma_f32 {D}, v[60:61], v[60:61], {D}", 1),
    ("fma_f32", "v_fma_f32 v{d}, v{x}, v{y}, v{d}", 0),
]


def op_body(text, is64, acc, unroll=8):
    lines = []
    for _ in range(unroll):
        for k in range(acc):
            d = BASE + 2 * k
            lines.append(text.format(d=d, D=pr(d), x=60, y=61))
    return lines, unroll * acc


def bfly32(base):
    x, y, t, q, u = base, base + 2, base + 4, base + 5, base + 6
    return ["v_subrev_u32_e32 v%d, %%0, v%d" % (t, x),
            "v_min_u32_e32 v%d, v%d, v%d" % (x, x, t),
            "v_mul_hi_u32 v%d, v%d, %%3" % (q, y),
            "v_lshl_add_u32 v%d, v%d, 1, %%0" % (u, x),
            "v_mad_u64_u32 v[%d:%d], vcc, v%d, %%1, v[%d:%d]" % (x, x + 1, q, x, x + 1),
            "v_mad_u64_u32 v[%d:%d], vcc, v%d, %%2, v[%d:%d]" % (x, x + 1, y, x, x + 1),
            "v_sub_u32_e32 v%d, v%d, v%d" % (y, u, x)]


def bfly64(base, vtw=False):
    """ct_bfly of gen_polymul_asm.py.  SGPR form: %0 delta, %1 mask, %2 w0, %3 w1, %4 a0, %5 a1, %6 3p (pair), %7 0xC0000000.
    VGPR form: the twiddle record (w0 w1 a0 a1) in v[56:59] as in the kernel (delta, mask, 3p, C0 stay scalar there too)."""
    x, y, t, U, A, P, Q, H, Y2 = base, base + 2, base + 4, base + 6, base + 8, base + 10, base + 12, base + 14, base + 16
    w0, w1, a0, a1 = ("v56", "v57", "v58", "v59") if vtw else ("%2", "%3", "%4", "%5")
    return ["v_lshrrev_b32_e32 v%d, 30, v%d" % (t, x + 1),
            "v_and_b32_e32 v%d, %%1, v%d" % (x + 1, x + 1),
            "v_mad_u64_u32 %s, s[40:41], v%d, %%0, %s" % (pr(U), t, pr(x)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %s, 0" % (pr(A), y + 1, a0),
            "v_mad_u64_u32 %s, s[42:43], v%d, %s, %s" % (pr(A), y, a1, pr(A)),
            "v_mov_b32_e32 v%d, v%d" % (P, A + 1),
            "v_addc_co_u32_e64 v%d, s[40:41], 0, 0, s[42:43]" % (P + 1),
            "v_mad_u64_u32 %s, s[40:41], v%d, %s, %s" % (pr(Q), y + 1, a1, pr(P)),
            "v_lshl_add_u64 %s, %s, 1, %%6" % (pr(Y2), pr(U)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %s, 0" % (pr(H), y, w1),
            "v_mad_u64_u32 %s, s[40:41], v%d, %s, %s" % (pr(H), y + 1, w0, pr(H)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %%0, %s" % (pr(H), Q + 1, pr(H)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %%7, %s" % (pr(H), Q, pr(H)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %s, %s" % (pr(x), y, w0, pr(U)),
            "v_mad_u64_u32 %s, s[40:41], v%d, %%0, %s" % (pr(x), Q, pr(x)),
            "v_sub_co_u32_e64 v%d, s[44:45], v%d, v%d" % (y, Y2, x),
            "v_add_u32_e32 v%d, v%d, v%d" % (x + 1, x + 1, H),
            "v_subb_co_u32_e64 v%d, s[40:41], v%d, v%d, s[44:45]" % (y + 1, Y2 + 1, x + 1)]


def mads_only(base, vtw=True):
    return [i for i in bfly64(base, vtw) if "v_mad_u64_u32" in i]


def light_only(base, vtw=True):
    return [i for i in bfly64(base, vtw) if "v_mad_u64_u32" not in i]


def interleave(streams):
    out = []
    for k in range(max(len(s) for s in streams)):
        for s in streams:
            if k < len(s):
                out.append(s[k])
    return out


def bfly_body(gen, acc, pairwise, reps=1, **kw):
    streams = [gen(BASE + k * 20, **kw) for k in range(acc)]
    if pairwise:
        body = []
        for k in range(0, acc, 2):
            body += interleave(streams[k:k + 2])
    else:
        body = [i for s in streams for i in s]
    return body * reps, acc * reps


KERNELS = []     # (name, units per body execution, kind)


def emit(out, name, body, units, kind, nreg):
    clob = ", ".join('"v%d"' % r for r in range(56, BASE + nreg)) + ', "vcc", "s40", "s41", "s42", "s43", "s44", "s45"'
    init = "\\n\"\n        \"".join("v_mov_b32_e32 v%d, v0" % r for r in range(56, BASE + nreg))
    text = "\\n\"\n        \"".join(body)
    if kind == "b64":
        args = "uint32_t delta, uint32_t mask, uint32_t w0, uint32_t w1, uint32_t a0, uint32_t a1, uint64_t p3, uint32_t c0k"
        ins = '"s"(delta), "s"(mask), "s"(w0), "s"(w1), "s"(a0), "s"(a1), "s"(p3), "s"(c0k)'
    elif kind == "b32":
        args = "uint32_t p2, uint32_t negp, uint32_t w, uint32_t wp"
        ins = '"s"(p2), "s"(negp), "s"(w), "s"(wp)'
    else:
        args = "uint32_t sb"
        ins = '"s"(sb)'
    out.append("""
__global__ void k_%s(uint32_t *out, int iters, %s, Clk *clk) {
  asm volatile("%s" ::: %s);
  const long long c0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "%s"
        :
        : %s
        : %s);
  }
  const long long c1 = clock64(), r1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = Clk{c1 - c0, r1 - r0};
  if (c1 == 1) out[0] = 1;
}
""" % (name, args, init, clob, text, ins, clob))
    KERNELS.append((name, units, kind, len(body)))


def main():
    out = ["// GENERATED by tools/gen_ubench_issue.py -- do not edit\n"]
    for name, text, is64 in OPS:
        for acc in (1, 4):
            body, units = op_body(text, is64, acc)
            emit(out, "op_%s_%d" % (name, acc), body, units, "op", 2 * acc)
    # mixes on independent chains: H = mad64 on pairs, L = add on singles
    for nh, nl in ((1, 1), (1, 2), (2, 1), (1, 3), (10, 8)):
        body = []
        for _ in range(4):
            for k in range(2):
                d = BASE + 4 * k
                body += ["v_mad_u64_u32 %s, s[40:41], v60, v61, %s" % (pr(d), pr(d))] * nh
                body += ["v_add_u32_e32 v%d, v60, v%d" % (d + 2, d + 2)] * nl
        emit(out, "mix_h%dl%d" % (nh, nl), body, 8 * (nh + nl), "op", 8)
    for acc in (1, 2, 4):
        b, u = bfly_body(bfly32, acc, False)
        emit(out, "bfly32_%d" % acc, b, u, "b32", 20 * acc)
    b, u = bfly_body(bfly32, 2, True)
    emit(out, "bfly32i_2", b, u, "b32", 40)
    for acc in (1, 2, 4):
        b, u = bfly_body(bfly64, acc, False)
        emit(out, "bfly64_%d" % acc, b, u, "b64", 20 * acc)
    for acc in (2, 4):
        b, u = bfly_body(bfly64, acc, True)
        emit(out, "bfly64i_%d" % acc, b, u, "b64", 20 * acc)
        b, u = bfly_body(bfly64, acc, True, vtw=True)
        emit(out, "bfly64iv_%d" % acc, b, u, "b64", 20 * acc)
    b, u = bfly_body(mads_only, 2, True)
    emit(out, "bfly64_mads_only", b, u, "b64", 40)
    b, u = bfly_body(light_only, 2, True)
    emit(out, "bfly64_light_only", b, u, "b64", 40)
    # straight-line code: 18 instructions x 8 bytes = 144 B per butterfly
    for kib in (16, 48, 96, 192):
        reps = kib * 1024 // (2 * 18 * 8)
        b, u = bfly_body(bfly64, 2, True, reps=reps, vtw=True)
        emit(out, "bfly64iv_long%d" % kib, b, u, "b64", 40)
    out.append("\nstruct KernelRow { const char *name; int units, ninstr; const char *kind; void *fn; };\n")
    out.append("static const KernelRow kRows[] = {\n")
    for name, units, kind, ninstr in KERNELS:
        out.append('  {"%s", %d, %d, "%s", (void *)k_%s},\n' % (name, units, ninstr, kind, name))
    out.append("};\n")
    open(os.path.join(ROOT, "tools", "ubench_issue_gen.inc"), "w").write("".join(out))


if __name__ == "__main__":
    main()
