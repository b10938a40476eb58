"""CPU-side parity of the GENERATED assembly row kernels: the text tools/gen_row*_asm.py emit is executed by the small
gfx950 interpreter in tests/asm_emu.py and compared with the oracle, so the `-m "not gpu"` gate covers the arithmetic of
the assembly kernels too (register allocation, butterfly order, twiddle indexing, exchange addressing, tail handling).
The GPU tests (tests/test_gpu_u32_asm.py, tests/test_gpu_u16_asm.py) remain the parity tests proper.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import asm_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nfllib_amd", "csrc")


GENERATORS = ("gen_polymul_asm.py", "gen_row1024_u32_asm.py", "gen_row128_u16_asm.py", "gen_row8_u32_asm.py")


@pytest.fixture(scope="module")
def generated():
    """the generated sources (normally made by the library's Makefile; regenerated here when absent or older than the
    generators, as on a fresh checkout: only the metric kernel's listing is kept in history)"""
    state = {"ran": False}

    def get(stem):
        path = os.path.join(CSRC, stem + "_gfx950.s")
        srcs = [os.path.join(ROOT, "tools", g) for g in GENERATORS]
        srcs += [os.path.join(ROOT, "tools", "asmgen", f) for f in os.listdir(os.path.join(ROOT, "tools", "asmgen")) if f.endswith(".py")]
        newest = max(os.path.getmtime(f) for f in srcs)
        if not state["ran"] and (not os.path.exists(path) or os.path.getmtime(path) < newest):
            for g in GENERATORS:
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", g)], stdout=subprocess.DEVNULL)
            state["ran"] = True
        return path
    return get


def operands(o, limb_bits, n, nm, batch, seed):
    from nfllib_amd.params import params
    prm = params(limb_bits)
    rng = np.random.default_rng(seed)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a = (rng.integers(0, 1 << 62, size=(batch, nm, n), dtype=np.uint64) % P[None, :, None]).astype(prm.dtype)
    b = (rng.integers(0, 1 << 62, size=(batch, nm, n), dtype=np.uint64) % P[None, :, None]).astype(prm.dtype)
    Pw = P.astype(prm.dtype)
    a[0, :, 0], a[0, :, 1], a[0, :, 2], a[0, :, n - 1] = 0, 1, Pw - 1, Pw - 1
    b[0, :, 0], b[0, :, 1], b[0, :, 2], b[0, :, n - 1] = Pw - 1, Pw - 1, Pw - 1, Pw - 1
    return prm, a, b


@pytest.mark.parametrize("nm,batch", [(1, 3), (2, 5), (3, 90)])
def test_emulated_lane_per_row_product_n8(nm, batch, generated, oracle_factory):
    o = oracle_factory(32, 8, nm)
    prm, a, b = operands(o, 32, 8, nm, batch, 11)
    got = asm_emu.run_row_kernel(generated("row8_u32"), 32, 8, nm, prm, a, b, 256, True)
    assert np.array_equal(got, o.polymul(a, b))
    # the stand-alone transforms and the product with b already transformed
    fa, fb = o.ntt(a), o.ntt(b)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row8_fwd_u32"), 32, 8, nm, prm, a, a, 256, True), fa)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row8_inv_u32"), 32, 8, nm, prm, fa, fa, 256, True), a)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row8_ntt_u32"), 32, 8, nm, prm, a, fb, 256, True), o.polymul(a, b))


@pytest.mark.parametrize("n,nm,batch", [(1024, 1, 1), (1024, 3, 3), (2048, 2, 1), (4096, 1, 1), (4096, 3, 1)])
def test_emulated_u32_product(n, nm, batch, generated, oracle_factory):
    o = oracle_factory(32, n, nm)
    prm, a, b = operands(o, 32, n, nm, batch, 12)
    got = asm_emu.run_row_kernel(generated("row%d_u32" % n), 32, n, nm, prm, a, b, 4096 // n, True)
    assert np.array_equal(got, o.polymul(a, b))


@pytest.mark.parametrize("n,nm,batch", [(1024, 3, 3), (2048, 2, 2), (4096, 1, 2)])
def test_emulated_u32_product_on_incomplete_transforms(n, nm, batch, generated, oracle_factory):
    """32-bit limbs (tools/gen_row1024_u32_asm.py base_mul): two stages dropped each way, four-term sums of canonical 30-bit residues in
    ONE carry-free 64-bit chain, Barrett with floor(2^62 / p) = 2^32 + m; all-(p-1) rows are the largest sums"""
    o = oracle_factory(32, n, nm)
    prm, a, b = operands(o, 32, n, nm, batch, 27)
    P = np.asarray(prm.P[:nm], dtype=np.uint64).astype(prm.dtype)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    got = asm_emu.run_row_kernel(generated("row%d_i2_u32" % n), 32, n, nm, prm, a, b, 4096 // n, True, incomplete=2)
    assert np.array_equal(got, o.polymul(a, b))


@pytest.mark.parametrize("n,nm,batch", [(1024, 2, 3), (2048, 1, 1), (4096, 2, 1)])
def test_emulated_u32_transforms(n, nm, batch, generated, oracle_factory):
    o = oracle_factory(32, n, nm)
    prm, a, _ = operands(o, 32, n, nm, batch, 13)
    f = asm_emu.run_row_kernel(generated("row%d_fwd_u32" % n), 32, n, nm, prm, a, a, 4096 // n, True)
    assert np.array_equal(f, o.ntt(a))
    back = asm_emu.run_row_kernel(generated("row%d_inv_u32" % n), 32, n, nm, prm, f, f, 4096 // n, True)
    assert np.array_equal(back, a)


@pytest.mark.parametrize("nm,batch", [(1, 1), (1, 9), (2, 21)])
def test_emulated_u16_product_and_transforms(nm, batch, generated, oracle_factory):
    o = oracle_factory(16, 128, nm)
    prm, a, b = operands(o, 16, 128, nm, batch, 14)
    got = asm_emu.run_row_kernel(generated("row128_u16"), 16, 128, nm, prm, a, b, 32, False)
    assert np.array_equal(got, o.polymul(a, b))
    f = asm_emu.run_row_kernel(generated("row128_fwd_u16"), 16, 128, nm, prm, a, a, 32, False)
    assert np.array_equal(f, o.ntt(a))
    back = asm_emu.run_row_kernel(generated("row128_inv_u16"), 16, 128, nm, prm, f, f, 32, False)
    assert np.array_equal(back, a)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row128_ntt_u16"), 16, 128, nm, prm, a, o.ntt(b), 32, False), o.polymul(a, b))


@pytest.mark.parametrize("stem,n,block_log,nm,batch", [("polymul4096nt", 4096, 12, 2, 2), ("polymul4096nt", 4096, 12, 1, 1),
                                                        ("polymul8192", 8192, 13, 1, 1), ("polymul16384", 16384, 14, 1, 1)])
def test_emulated_u64_block_product(stem, n, block_log, nm, batch, generated, oracle_factory):
    """the metric kernel (workloads B / D) and its 8192- and 16384-word siblings: one row per workgroup"""
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, batch, 15)
    got = asm_emu.run_block_kernel(generated(stem), n, nm, prm, a, b, block_log)
    assert np.array_equal(got, o.polymul(a, b))


@pytest.mark.parametrize("level", [1, 2])
def test_emulated_u64_product_on_incomplete_transforms(level, generated, oracle_factory):
    """nflhip_polymul4096i{1,2}_asm (tools/asmgen/incomplete.py): forward transforms stop `level` stages early, base
    multiplication mod X^(2^level) -+ zeta on lazily accumulated 128-bit sums, inverse starts `level` stages late -- the words
    out are the complete kernel's.  Random operands with the boundary values of operands(), all-(p-1) rows (the largest sums
    the 2^127 Barrett step sees) and unit impulses (X^(n-1) * X^(n-1): the negacyclic sign through the base multiplication)."""
    nm = 2
    o = oracle_factory(64, 4096, nm)
    prm, a, b = operands(o, 64, 4096, nm, 3, 20 + level)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a[1], b[1] = (P - 1)[:, None], (P - 1)[:, None]
    a[2], b[2] = 0, 0
    a[2, :, 4095], b[2, :, 4095] = 1, 1
    b[2, :, 1] = P - 1
    stem = generated("polymul4096i%d" % level)
    got = asm_emu.run_block_kernel(stem, 4096, nm, prm, a, b, 12, incomplete=level)
    assert np.array_equal(got, o.polymul(a, b))
    # the records matter: with the complete kernel's ModConst (n^-1, mu2) the same listing must NOT produce the product
    bad = asm_emu.run_block_kernel(stem, 4096, nm, prm, a[:1], b[:1], 12, incomplete=0)
    assert not np.array_equal(bad, got[:1])


@pytest.mark.parametrize("n,block_log", [(8192, 13), (16384, 14)])
def test_emulated_u64_row_products_on_incomplete_transforms(n, block_log, generated, oracle_factory):
    """the row-resident products (ring-mode register map: one butterfly at a time, every register taken -- the base
    multiplication's scratch is three reserved slots of the twiddle ring) with 2 stages dropped each way"""
    o = oracle_factory(64, n, 1)
    prm, a, b = operands(o, 64, n, 1, 2, 25)
    P = np.asarray(prm.P[:1], dtype=np.uint64)
    a[1], b[1] = (P - 1)[:, None], (P - 1)[:, None]
    got = asm_emu.run_block_kernel(generated("polymul%di2" % n), n, 1, prm, a, b, block_log, incomplete=2)
    assert np.array_equal(got, o.polymul(a, b))


@pytest.mark.parametrize("n,nm,batch", [(1024, 3, 3), (2048, 2, 2)])
def test_emulated_u64_wave_per_row_kernels(n, nm, batch, generated, oracle_factory):
    """tools/asmgen/rows1k.py: 64-bit rows of 1024 words (one wave per row, no workgroup barrier) and 2048 (two waves): the
    product on incomplete transforms (level 2: at 1024 the third pass of either transform is gone), the product on complete
    transforms, and the stand-alone transforms; odd row counts leave surplus waves / rows in the last workgroup"""
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, batch, 26)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    rpw = 4096 // n
    want = o.polymul(a, b)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row%d_u64" % n), 64, n, nm, prm, a, b, rpw, True, incomplete=2), want)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row%d_l0_u64" % n), 64, n, nm, prm, a, b, rpw, True), want)
    fa = o.ntt(a)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row%d_fwd_u64" % n), 64, n, nm, prm, a, a, rpw, True), fa)
    assert np.array_equal(asm_emu.run_row_kernel(generated("row%d_inv_u64" % n), 64, n, nm, prm, fa, fa, rpw, True), a)


@pytest.mark.parametrize("n,nm,batch", [(1024, 3, 3), (2048, 2, 2)])
def test_emulated_u64_wave_per_row_fused_pipelines(n, nm, batch, generated, oracle_factory):
    """tools/asmgen/rows1k.py build_row1k_fwd_fma / build_row1k_fma_inv: out0 = NTT(x) k0 + NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)] with
    word and int8 operands, shared (stride 0) and dense keys / x, and INTT(b -+ a k) -- against the operator-by-operator oracle"""
    ADD, SUB, MUL = 0, 1, 2
    o = oracle_factory(64, n, nm)
    from nfllib_amd.params import params
    prm = params(64)
    rng = np.random.default_rng(31)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    rnd = lambda B: (rng.integers(0, 1 << 62, size=(B, nm, n), dtype=np.uint64) % P[None, :, None])
    small = lambda B: rng.integers(-128, 128, size=(B, n)).astype(np.int8)
    expand = lambda v: np.where(v.astype(np.int64)[:, None, :] < 0, P.astype(np.int64)[None, :, None] + v.astype(np.int64)[:, None, :],
                                v.astype(np.int64)[:, None, :]).astype(np.uint64)
    rpw = 4096 // n
    a, b = rnd(batch), rnd(batch)
    # the extreme words of the one-chain multiply-add (tools/asmgen/fused.py mac128): a = 0 (p - a = p), a = b = k = p - 1
    a[0, :, 0], a[0, :, 1], a[0, :, 2], a[0, :, 3] = 0, P - 1, P - 1, 0
    b[0, :, 0], b[0, :, 1], b[0, :, 2], b[0, :, 3] = P - 1, P - 1, 0, 0
    for key in (rnd(1), rnd(batch)):
        key[0, :, :4] = (P - 1)[:, None]
        kk = np.broadcast_to(key, (batch, nm, n)).copy()
        for sub, stem in ((True, "fmsinv"), (False, "fmainv")):
            want = o.intt(o.pointwise(SUB if sub else ADD, b, o.pointwise(MUL, a, kk)))
            got = asm_emu.run_row_fused(generated("row%d_%s_u64" % (n, stem)), n, nm, prm, rpw, "inv", a=a, b=b, key=key)
            assert np.array_equal(got, want), (stem, key.shape)
    for fmt in ("w", "i8"):
        for two, xb in ((True, 1), (False, batch)):
            if fmt == "w":
                x, e0, e1 = rnd(xb), rnd(batch), rnd(batch)
                X, E0, E1 = x, e0, e1
            else:
                x, e0, e1 = small(xb), small(batch), small(batch)
                X, E0, E1 = expand(x), expand(e0), expand(e1)
            Xb = np.broadcast_to(X, (batch, nm, n)).copy()
            k0, k1 = rnd(1), rnd(batch)
            k0[0, :, :8], k1[0, :, :8] = (P - 1)[:, None], (P - 1)[:, None]
            K0 = np.broadcast_to(k0, (batch, nm, n)).copy()
            w0 = o.pointwise(ADD, o.pointwise(MUL, o.ntt(Xb), K0), o.ntt(E0))
            w1 = o.pointwise(ADD, o.pointwise(MUL, o.ntt(Xb), k1), o.ntt(E1))
            stem = ("enc2" if two else "fmafwd") + fmt
            r = asm_emu.run_row_fused(generated("row%d_%s_u64" % (n, stem)), n, nm, prm, rpw, "fwd", x=x, e0=e0, k0=k0,
                                      e1=e1 if two else None, k1=k1 if two else None, batch=batch)
            assert np.array_equal(r[0], w0) and (not two or np.array_equal(r[1], w1)), stem


@pytest.mark.parametrize("n,nm,batch", [(1024, 3, 3), (2048, 3, 3), (4096, 1, 2)])
def test_emulated_u32_wave_per_row_fused_pipelines(n, nm, batch, generated, oracle_factory):
    """tools/gen_row1024_u32_asm.py build_fwd_fma / build_fma_inv (32-bit limbs, one / two / four waves per row): out0 = NTT(x) k0 +
    NTT(e0) [, out1 = NTT(x) k1 + NTT(e1)] with word and int8 operands (extreme bytes included), shared (stride 0) and dense keys / x,
    and INTT(b -+ a k) -- against the operator-by-operator oracle; odd row counts leave surplus waves / rows in the last workgroup"""
    ADD, SUB, MUL = 0, 1, 2
    o = oracle_factory(32, n, nm)
    from nfllib_amd.params import params
    prm = params(32)
    rng = np.random.default_rng(47)
    P = np.asarray(prm.P[:nm], dtype=np.uint32)
    rnd = lambda B: (rng.integers(0, 1 << 30, size=(B, nm, n), dtype=np.uint32) % P[None, :, None])
    def small(B):
        v = rng.integers(-128, 128, size=(B, n)).astype(np.int8)
        v[:, :4] = (-128, 127, -1, 0)
        return v
    expand = lambda v: np.where(v.astype(np.int64)[:, None, :] < 0, P.astype(np.int64)[None, :, None] + v.astype(np.int64)[:, None, :],
                                v.astype(np.int64)[:, None, :]).astype(np.uint32)
    rpw = 4096 // n
    a, b = rnd(batch), rnd(batch)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    for key in (rnd(1), rnd(batch)):
        key[0, :, :2] = (P - 1)[:, None]
        kk = np.broadcast_to(key, (batch, nm, n)).copy()
        for sub, stem in ((True, "fmsinv"), (False, "fmainv")):
            want = o.intt(o.pointwise(SUB if sub else ADD, b, o.pointwise(MUL, a, kk)))
            got = asm_emu.run_row_fused(generated("row%d_%s_u32" % (n, stem)), n, nm, prm, rpw, "inv", limb_bits=32, a=a, b=b, key=key)
            assert np.array_equal(got, want), (stem, key.shape)
    for fmt in ("w", "i8"):
        for two, xb in ((True, 1), (False, batch)):
            if fmt == "w":
                x, e0, e1 = rnd(xb), rnd(batch), rnd(batch)
                X, E0, E1 = x, e0, e1
            else:
                x, e0, e1 = small(xb), small(batch), small(batch)
                X, E0, E1 = expand(x), expand(e0), expand(e1)
            Xb = np.broadcast_to(X, (batch, nm, n)).copy()
            k0, k1 = rnd(1), rnd(batch)
            k0[0, :, :2] = (P - 1)[:, None]
            K0 = np.broadcast_to(k0, (batch, nm, n)).copy()
            w0 = o.pointwise(ADD, o.pointwise(MUL, o.ntt(Xb), K0), o.ntt(E0))
            w1 = o.pointwise(ADD, o.pointwise(MUL, o.ntt(Xb), k1), o.ntt(E1))
            stem = ("enc2" if two else "fmafwd") + fmt
            r = asm_emu.run_row_fused(generated("row%d_%s_u32" % (n, stem)), n, nm, prm, rpw, "fwd", limb_bits=32, incomplete=2, x=x, e0=e0, k0=k0,
                                      e1=e1 if two else None, k1=k1 if two else None, batch=batch)
            assert np.array_equal(r[0], w0) and (not two or np.array_equal(r[1], w1)), stem


@pytest.mark.parametrize("level", [1, 2])
def test_incomplete_transform_algebra_in_integers(level):
    """what tools/asmgen/incomplete.py relies on, on Python integers at n = 64: after S = log2(n) - level stages of the merged
    Cooley-Tukey network over psi_br[k] = phi^bitrev(k), group g of 2^level consecutive words is the residue modulo
    X^(2^level) - zeta_g with zeta_g = +psi_br[2^(S-1) + g / 2] for even g and its negative for odd g; multiplying residues group by
    group and running the mirrored inverse over S stages (scale 2^-S) gives the negacyclic product"""
    import random
    from nfllib_amd.params import params
    prm = params(64)
    n, logn = 64, 6
    p, phi = int(prm.P[0]), int(prm.primitive_roots[0])
    for _ in range(prm.kmax_log2 - logn):
        phi = phi * phi % p
    br = lambda k: int(format(k, "0%db" % logn)[::-1], 2)
    psi = [pow(phi, br(k), p) for k in range(n)]

    def fwd(a, stages):
        a, t, m = a[:], n, 1
        for _ in range(stages):
            t //= 2
            for i in range(m):
                for j in range(i * 2 * t, i * 2 * t + t):
                    u, v = a[j], a[j + t] * psi[m + i] % p
                    a[j], a[j + t] = (u + v) % p, (u - v) % p
            m *= 2
        return a

    def inv(a, stages):
        a, m, t = a[:], 1 << stages, n >> stages
        for _ in range(stages):
            m //= 2
            for i in range(m):
                w = pow(psi[m + i], p - 2, p)
                for j in range(i * 2 * t, i * 2 * t + t):
                    u, v = a[j], a[j + t]
                    a[j], a[j + t] = (u + v) % p, (u - v) * w % p
            t *= 2
        sc = pow(1 << stages, p - 2, p)
        return [x * sc % p for x in a]

    rnd = random.Random(3)
    a, b = [rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]
    want = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                want[k] = (want[k] + a[i] * b[j]) % p
            else:
                want[k - n] = (want[k - n] - a[i] * b[j]) % p
    S, G = logn - level, 1 << level
    fa, fb, out = fwd(a, S), fwd(b, S), [0] * n
    for g in range(n // G):
        w = psi[(1 << (S - 1)) + g // 2]
        zeta = w if g % 2 == 0 else p - w
        A, B = fa[g * G:(g + 1) * G], fb[g * G:(g + 1) * G]
        for k in range(G):
            acc = sum(A[i] * B[k - i] for i in range(k + 1)) + zeta * sum(A[i] * B[k + G - i] for i in range(k + 1, G))
            out[g * G + k] = acc % p
    assert inv(out, S) == want


def test_barrett_step_of_the_base_multiplication_in_integers():
    """the reduction incomplete.py emits for sums T < 2^127 of products of folded words, restated on Python integers:
    th = T >> 63, q^ = 2 th + floor(th m / 2^64) with m = floor(2^127 / p) - 2^65, r = T - q^ p must lie in [0, 2^64) with
    q - q^ <= 3 -- at the largest operands the two-bit fold can leave (2^62 + 3 delta - 1) and at random ones, for the first and
    the last delta-form moduli (delta up to 2^32)"""
    from nfllib_amd.params import params
    prm = params(64)
    rng = np.random.default_rng(99)
    for ci in (0, 1, 29, 63, 91):
        p = int(prm.P[ci])
        d = (1 << 62) - p
        m = (1 << 127) // p - (1 << 65)
        assert 0 <= m < (1 << 35)
        m0, m1 = m & 0xFFFFFFFF, m >> 32
        hi = (1 << 62) + 3 * d - 1
        cases = [([hi] * 4, [hi] * 4), ([p - 1] * 4, [p - 1] * 4), ([0] * 4, [hi] * 4), ([hi, 0, 0, 0], [hi, 0, 0, 0])]
        cases += [([int(x) % (hi + 1) for x in rng.integers(0, 1 << 63, 4)], [int(x) % (hi + 1) for x in rng.integers(0, 1 << 63, 4)])
                  for _ in range(400)]
        for xs, ys in cases:
            T = sum(x * y for x, y in zip(xs, ys))
            th = T >> 63
            tl, thh = th & 0xFFFFFFFF, th >> 32
            H = thh * m0 + ((tl * m0) >> 32)
            H += tl * m1
            assert H < (1 << 64)
            q = 2 * th + thh * m1 + (H >> 32)
            r = T - q * p
            assert 0 <= T // p - q <= 3 and 0 <= r < (1 << 64)
            f = (r & ((1 << 62) - 1)) + (r >> 62) * d
            assert f % p == T % p and f <= hi


@pytest.mark.parametrize("suffix,n,block_log", [("4096", 4096, 12), ("8192", 8192, 13), ("16384", 16384, 14)])
def test_emulated_u64_block_transforms(suffix, n, block_log, generated, oracle_factory):
    """stand-alone forward / inverse, the product with b already in NTT form, and (4096) the inverse with a fused
    pointwise multiply"""
    nm = 2 if n == 4096 else 1
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, 1, 16)
    fa, fb = o.ntt(a), o.ntt(b)
    want = o.polymul(a, b)
    assert np.array_equal(asm_emu.run_block_kernel(generated("ntt_fwd" + suffix), n, nm, prm, a, a, block_log), fa)
    assert np.array_equal(asm_emu.run_block_kernel(generated("ntt_inv" + suffix), n, nm, prm, fa, fa, block_log), a)
    assert np.array_equal(asm_emu.run_block_kernel(generated("polymul_ntt" + suffix), n, nm, prm, a, fb, block_log), want)
    if n == 4096:
        assert np.array_equal(asm_emu.run_block_kernel(generated("ntt_inv_mul4096"), n, nm, prm, fa, fb, block_log), want)


def _signed_rows(x, P):
    """(count, n) signed integers -> (count, nm, n) canonical residue words (what the samplers store: core.hpp:230-277)"""
    v = x.astype(np.int64)[:, None, :]
    return np.where(v < 0, (P[None, :, None].astype(np.int64) + v), v).astype(np.uint64)


@pytest.mark.parametrize("fmt,nm,batch", [(np.int8, 2, 2), (np.int16, 1, 1), (np.int32, 1, 1), (np.uint64, 2, 1)])
def test_emulated_u64_transform_fused_forward_multiply_add(fmt, nm, batch, generated, oracle_factory):
    """out0 = NTT(x0) * k0 + NTT(x1), out1 = NTT(x0) * k1 + NTT(x2) in ONE launch (the LWE demo's encrypt(),
    tests/nfllib_demo_main_op.cpp:26-46), inputs as residue words or as one signed integer per coefficient; the keys are one
    polynomial for the whole batch"""
    n = 4096
    o = oracle_factory(64, n, nm)
    prm, ka, kb = operands(o, 64, n, nm, 1, 31)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    rng = np.random.default_rng(32)
    if fmt is np.uint64:
        _, x0, x1 = operands(o, 64, n, nm, batch, 33)
        _, x2, _ = operands(o, 64, n, nm, batch, 34)
        w = [x0, x1, x2]
        xs = w
    else:
        info = np.iinfo(fmt)
        xs = [rng.integers(info.min, info.max, size=(batch, n), endpoint=True).astype(fmt) for _ in range(3)]
        xs[0][0, :4] = (info.min, info.max, 0, -1)
        w = [_signed_rows(x, P) for x in xs]
    f = [o.ntt(x) for x in w]
    KA, KB = (np.ascontiguousarray(np.broadcast_to(k_, f[0].shape)) for k_ in (ka, kb))
    want0 = o.pointwise(0, o.pointwise(2, f[0], KA), f[1])
    want1 = o.pointwise(0, o.pointwise(2, f[0], KB), f[2])
    got = asm_emu.run_fused_kernel(generated("fused_enc2_4096"), n, nm, prm, xs, [ka, kb], batch, 2)
    assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1)
    got = asm_emu.run_fused_kernel(generated("fused_fma_fwd4096"), n, nm, prm, xs[:2], [ka], batch, 1)
    assert np.array_equal(got[0], want0)
    if fmt is np.int8:   # the 1-D grid that keeps the nm rows of an element on one XCD (padding workgroups exit at once)
        got = asm_emu.run_fused_kernel(generated("fused_enc2_4096"), n, nm, prm, xs, [ka, kb], batch, 2, remap=True)
        assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1)
        # ... and the ring-mode variants of the two forward kernels (lane-major twiddle copy), both grids
        got = asm_emu.run_fused_kernel(generated("fused_enc2_4096r"), n, nm, prm, xs, [ka, kb], batch, 2, remap=True, lane_major=True)
        assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1)
        got = asm_emu.run_fused_kernel(generated("fused_fma_fwd4096r"), n, nm, prm, xs[:2], [ka], batch, 1, lane_major=True)
        assert np.array_equal(got[0], want0)


@pytest.mark.parametrize("nm,batch", [(2, 2), (1, 1)])
def test_emulated_u64_transform_fused_multiply_subtract_inverse(nm, batch, generated, oracle_factory):
    """out = INTT(x1 - x0 * k) (the demo's decrypt(), tests/nfllib_demo_main_op.cpp:49-58) and INTT(x1 + x0 * k)"""
    n = 4096
    o = oracle_factory(64, n, nm)
    prm, x0, x1 = operands(o, 64, n, nm, batch, 35)
    _, k, _ = operands(o, 64, n, nm, 1, 36)
    K = np.ascontiguousarray(np.broadcast_to(k, x0.shape))
    prod = o.pointwise(2, x0, K)
    got = asm_emu.run_fused_kernel(generated("fused_fms_inv4096"), n, nm, prm, [x0, x1], [k], batch, 1)
    assert np.array_equal(got[0], o.intt(o.pointwise(1, x1, prod)))
    got = asm_emu.run_fused_kernel(generated("fused_fma_inv4096"), n, nm, prm, [x0, x1], [k], batch, 1)
    assert np.array_equal(got[0], o.intt(o.pointwise(0, x1, prod)))
    # the ring-mode variants of the same two kernels (one butterfly at a time, 128 VGPRs, lane-major twiddle copy)
    got = asm_emu.run_fused_kernel(generated("fused_fms_inv4096r"), n, nm, prm, [x0, x1], [k], batch, 1, lane_major=True)
    assert np.array_equal(got[0], o.intt(o.pointwise(1, x1, prod)))
    got = asm_emu.run_fused_kernel(generated("fused_fma_inv4096r"), n, nm, prm, [x0, x1], [k], batch, 1, lane_major=True)
    assert np.array_equal(got[0], o.intt(o.pointwise(0, x1, prod)))


@pytest.mark.parametrize("n,nm,batch,fmt", [(8192, 2, 2, np.int8), (8192, 1, 1, np.uint64), (16384, 1, 1, np.int16), (16384, 2, 1, np.uint64)])
def test_emulated_u64_transform_fused_pipelines_on_row_resident_kernels(n, nm, batch, fmt, generated, oracle_factory):
    """the same four pipelines for rows of 8192 / 16384 words (tools/gen_polymul_asm.py build_fused_rows: ring-mode register
    map, the key row in the empty ring's registers): both forward kernels and both inverse kernels against the oracle"""
    o = oracle_factory(64, n, nm)
    prm, ka, kb = operands(o, 64, n, nm, 1, 41)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    rng = np.random.default_rng(42)
    if fmt is np.uint64:
        _, x0, x1 = operands(o, 64, n, nm, batch, 43)
        _, x2, _ = operands(o, 64, n, nm, batch, 44)
        xs = w = [x0, x1, x2]
    else:
        info = np.iinfo(fmt)
        xs = [rng.integers(info.min, info.max, size=(batch, n), endpoint=True).astype(fmt) for _ in range(3)]
        w = [_signed_rows(x, P) for x in xs]
    f = [o.ntt(x) for x in w]
    KA, KB = (np.ascontiguousarray(np.broadcast_to(k_, f[0].shape)) for k_ in (ka, kb))
    want0 = o.pointwise(0, o.pointwise(2, f[0], KA), f[1])
    want1 = o.pointwise(0, o.pointwise(2, f[0], KB), f[2])
    G = n // 4096
    got = asm_emu.run_fused_kernel(generated("fused_enc2_%d" % n), n, nm, prm, xs, [ka, kb], batch, 2, groups=G)
    assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1)
    got = asm_emu.run_fused_kernel(generated("fused_fma_fwd%d" % n), n, nm, prm, xs[:2], [ka], batch, 1, groups=G)
    assert np.array_equal(got[0], want0)
    prod = o.pointwise(2, want0, KB)
    got = asm_emu.run_fused_kernel(generated("fused_fms_inv%d" % n), n, nm, prm, [want0, want1], [kb], batch, 1, groups=G)
    assert np.array_equal(got[0], o.intt(o.pointwise(1, want1, prod)))
    got = asm_emu.run_fused_kernel(generated("fused_fma_inv%d" % n), n, nm, prm, [want0, want1], [kb], batch, 1, groups=G)
    assert np.array_equal(got[0], o.intt(o.pointwise(0, want1, prod)))


@pytest.mark.parametrize("n,nm,batch", [(16384, 2, 3), (8192, 3, 4), (8192, 1, 1)])
def test_emulated_u64_two_rows_per_workgroup_forward_transform_of_long_rows(n, nm, batch, generated, oracle_factory):
    """stand-alone forward transform at n = 16384 / 8192: two polynomials of one modulus per workgroup on shared twiddle
    records (the forward half of the fused product without the product); the odd one out is transformed twice"""
    o = oracle_factory(64, n, nm)
    prm, a, _ = operands(o, 64, n, nm, batch, 27)
    got = asm_emu.run_block_kernel(generated("ntt_fwd%dx2" % n), n, nm, prm, a, a, n.bit_length() - 1, count=batch)
    assert np.array_equal(got, o.ntt(a))


@pytest.mark.parametrize("nm,batch", [(1, 1), (2, 2)])
def test_emulated_u64_register_resident_32768_word_rows(nm, batch, generated, oracle_factory):
    """n = 32768 (workload F, the reference's largest test config): ONE operand register-resident in a 1024-thread
    workgroup, 32 words per thread -- forward, inverse, and the product with b' streamed through the point-wise step"""
    n = 32768
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, batch, 21)
    fa, fb = o.ntt(a), o.ntt(b)
    run = lambda stem, x, y: asm_emu.run_block_kernel(generated(stem), n, nm, prm, x, y, 15, words_per_thread=32)
    assert np.array_equal(run("ntt_fwd32768", a, a), fa)
    if batch == 1:   # (the second parameter set spends its time on the shipped pair below)
        assert np.array_equal(run("ntt_inv32768", fa, fa), a)
        assert np.array_equal(run("polymul_ntt32768", a, fb), o.polymul(a, b))
        # the pair of the composed product: b' travels through the scratch as [block][slot pair i][thread] x 16 bytes (coalesced
        # on both sides), not in the reference's order
        scratch = run("ntt_fwd32768s", b, b)
        assert np.array_equal(scratch.reshape(batch, nm, 8, 8, 256, 2), fb.reshape(batch, nm, 8, 256, 8, 2).transpose(0, 1, 2, 4, 3, 5))
        assert np.array_equal(run("polymul_ntt32768s", a, scratch), o.polymul(a, b))
    # ... and the same pair on incomplete transforms (level 2: b' is stored two stages short and unreduced, the point-wise step is the
    # base multiplication mod X^4 -+ zeta against the streamed groups, the inverse starts two stages late; all-(p - 1) rows included)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a2, b2 = a.copy(), b.copy()
    a2[batch - 1], b2[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    run2 = lambda stem, x, y: asm_emu.run_block_kernel(generated(stem), n, nm, prm, x, y, 15, words_per_thread=32, incomplete=2)
    scratch2 = run2("ntt_fwd32768si2", b2, b2)
    assert np.array_equal(run2("polymul_ntt32768si2", a2, scratch2), o.polymul(a2, b2))


@pytest.mark.parametrize("nm,batch", [(2, 2)])
def test_emulated_u64_forward_transform_of_a_compact_32768_word_row(nm, batch, generated, oracle_factory):
    """n = 32768: a compact Gaussian polynomial (one signed byte per coefficient, the same for every modulus) in, the NTT words
    of every modulus out -- the expansion v < 0 -> p + v happens in the registers the transform starts from"""
    n = 32768
    o = oracle_factory(64, n, nm)
    prm, a, _ = operands(o, 64, n, nm, batch, 31)
    rng = np.random.default_rng(5)
    x = rng.integers(-128, 127, size=(batch, n), endpoint=True).astype(np.int8)
    x[0, :4] = (-128, 127, 0, -1)
    P = np.asarray(o.P[:nm], dtype=np.uint64)
    words = np.where(x[:, None, :] < 0, (P[None, :, None].astype(np.int64) + x[:, None, :].astype(np.int64)).astype(np.uint64),
                     x[:, None, :].astype(np.int64).astype(np.uint64))
    got = asm_emu.run_block_kernel(generated("ntt_fwd32768i8"), n, nm, prm, a, a, 15, words_per_thread=32, compact=x)
    assert np.array_equal(got, o.ntt(np.ascontiguousarray(words)))


@pytest.mark.parametrize("nm,batch,two", [(2, 2, True), (1, 1, False)])
def test_emulated_u64_fused_forward_pipelines_of_32768_word_rows(nm, batch, two, generated, oracle_factory):
    """n = 32768: out0 = NTT(x) k0 + e0' [, out1 = NTT(x) k1 + e1'] with x one signed byte per coefficient, transformed once and
    kept in registers; the keys (one polynomial for the batch) and the transformed noise rows stream through the idle twiddle ring"""
    n = 32768
    o = oracle_factory(64, n, nm)
    prm, e0p, e1p = operands(o, 64, n, nm, batch, 37)
    _, k0, k1 = operands(o, 64, n, nm, 1, 41)
    rng = np.random.default_rng(9)
    x = rng.integers(-128, 127, size=(batch, n), endpoint=True).astype(np.int8)
    P = np.asarray(o.P[:nm], dtype=np.uint64)
    words = np.where(x[:, None, :] < 0, (P[None, :, None].astype(np.int64) + x[:, None, :].astype(np.int64)).astype(np.uint64),
                     x[:, None, :].astype(np.int64).astype(np.uint64))
    X = o.ntt(np.ascontiguousarray(words))
    bc = lambda k: np.ascontiguousarray(np.broadcast_to(k, X.shape))
    want0 = o.pointwise(0, o.pointwise(2, X, bc(k0)), e0p)
    if two:
        got = asm_emu.run_row32k_forward_pipeline(generated("fused_enc2_32768i8"), nm, prm, x, k0, e0p, k1, e1p)
        assert np.array_equal(got[0], want0)
        assert np.array_equal(got[1], o.pointwise(0, o.pointwise(2, X, bc(k1)), e1p))
    else:
        got = asm_emu.run_row32k_forward_pipeline(generated("fused_fma_fwd32768i8"), nm, prm, x, k0, e0p)
        assert np.array_equal(got[0], want0)


@pytest.mark.parametrize("nm,batch,shared_key,subtract", [(1, 1, True, True), (2, 2, False, False)])
def test_emulated_u64_fused_inverse_pipeline_of_32768_word_rows(nm, batch, shared_key, subtract, generated, oracle_factory):
    """n = 32768: INTT(b - a k) / INTT(b + a k) in ONE register-resident kernel (the decryption of the reference's demo at its
    largest test configuration, tests/nfllib_demo_main_op.cpp:51-57): b and the key stream through the idle twiddle ring"""
    n = 32768
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, batch, 23)
    _, k, _ = operands(o, 64, n, nm, 1 if shared_key else batch, 29)
    K = np.ascontiguousarray(np.broadcast_to(k, a.shape)) if shared_key else k
    prod = o.pointwise(2, a, K)
    want = o.intt(o.pointwise(1 if subtract else 0, b, prod))
    got = asm_emu.run_block_kernel(generated("fused_fms_inv32768" if subtract else "fused_fma_inv32768"), n, nm, prm, a, b, 15,
                                   words_per_thread=32, key=k)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nt", ["nt"])
@pytest.mark.parametrize("nm,batch", [(1, 3), (2, 2)])
def test_emulated_u64_two_rows_per_workgroup_transforms(nt, nm, batch, generated, oracle_factory):
    """n = 4096 stand-alone transforms, two polynomials per workgroup (an odd count leaves half a workgroup)"""
    n = 4096
    o = oracle_factory(64, n, nm)
    prm, a, _ = operands(o, 64, n, nm, batch, 17)
    fa = o.ntt(a)
    assert np.array_equal(asm_emu.run_block_kernel(generated("ntt_fwd4096x2" + nt), n, nm, prm, a, a, 12, count=batch), fa)
    assert np.array_equal(asm_emu.run_block_kernel(generated("ntt_inv4096x2" + nt), n, nm, prm, fa, fa, 12, count=batch), a)


@pytest.mark.parametrize("stem,n", [("polymul_pipe65536nt", 65536)])
def test_emulated_u64_three_role_kernel(stem, n, generated, oracle_factory):
    """n = 65536 (workload E): forward streaming, block products and inverse streaming roles"""
    o = oracle_factory(64, n, 1)
    prm, a, b = operands(o, 64, n, 1, 1, 18)
    assert np.array_equal(asm_emu.run_pipe_product(generated(stem), n, 1, prm, a, b), o.polymul(a, b))
    # two moduli with the launcher's XCD remap: the workgroup with linear index L takes unit (L mod 8) U/8 + L div 8 of the
    # modulus-major order, so that one XCD walks through a contiguous range of moduli (their twiddle tables stay in ITS L2)
    o2 = oracle_factory(64, n, 2)
    prm, a, b = operands(o2, 64, n, 2, 1, 19)
    assert np.array_equal(asm_emu.run_pipe_product(generated(stem), n, 2, prm, a, b, remap=True), o2.polymul(a, b))
    # operand b already transformed: the variant without a forward role for it (24 workgroups per row), b' read block-wise
    fb = o2.ntt(b)
    assert np.array_equal(asm_emu.run_pipe_product(generated(stem + "b"), n, 2, prm, a, fb, remap=True, b_ntt=True), o2.polymul(a, b))


def test_emulated_u64_three_role_kernel_on_incomplete_transforms(generated, oracle_factory):
    """workload E with its block products on incomplete transforms (role V = incomplete.body_incomplete at r = 4, the scale
    (n / 4)^-1 folded in by the streaming inverse role from the level-2 ModConst records), two moduli, XCD remap, and the same
    kernel driven as the chunked pipeline drives it (all three roles in one launch)"""
    n = 65536
    o2 = oracle_factory(64, n, 2)
    prm, a, b = operands(o2, 64, n, 2, 1, 23)
    stem = generated("polymul_pipe65536nti2")
    want = o2.polymul(a, b)
    assert np.array_equal(asm_emu.run_pipe_product(stem, n, 2, prm, a, b, remap=True, incomplete=2), want)
    if os.environ.get("NFL_EMU_FULL"):     # (12 s more on the interpreter; the roles are the same code either way)
        assert np.array_equal(asm_emu.run_pipe_product_pipelined(stem, n, 2, prm, a, b, incomplete=2), want)


def test_emulated_one_launch_plan_on_incomplete_transforms(generated, oracle_factory):
    """the persistent one-launch plan (rows of 32768 words) with its block products on incomplete transforms"""
    o = oracle_factory(64, 32768, 1)
    prm, a, b = operands(o, 64, 32768, 1, 8, 24)
    got = asm_emu.run_xcd_product(generated("polymul_xcd32768i2"), 32768, 1, prm, a, b, 0, 3, 0, 40, _picker("round-robin"), incomplete=2)
    assert np.array_equal(got, o.polymul(a, b))


def _picker(kind):
    import random
    rnd = random.Random(7)
    state = {"i": 0}

    def pick(live):
        if kind == "random":
            return rnd.choice(live)
        if kind == "highest":                 # the workgroup that joined last always goes first
            return live[-1]
        state["i"] += 1                       # round robin
        return live[state["i"] % len(live)]
    return pick


@pytest.mark.parametrize("stem,n,nm,batch,dlog,rlog,pooled,wgs,order", [
    ("polymul_xcd32768", 32768, 1, 8, 0, 3, 0, 40, "round-robin"),  # five workgroups per XCD
    # (the other two interleavings take 35 s and 60 s on the interpreter: with NFL_EMU_FULL=1, as the n = 65536 case below)
    pytest.param("polymul_xcd32768", 32768, 2, 5, 0, 2, 0, 8, "highest",       # two moduli, batch not a power of two, one workgroup per XCD
                 marks=pytest.mark.skipif(not os.environ.get("NFL_EMU_FULL"), reason="set NFL_EMU_FULL=1 (35 s)")),
    pytest.param("polymul_xcd32768", 32768, 1, 16, 1, 1, 0, 24, "random",      # two scheduling domains per XCD (needs 16 rows)
                 marks=pytest.mark.skipif(not os.environ.get("NFL_EMU_FULL"), reason="set NFL_EMU_FULL=1 (60 s)")),
    # n = 65536 (same generator, 16 block products and radix-16 streaming roles per row): 50 s on the interpreter, so only
    # with NFL_EMU_FULL=1; its three roles run in every suite through test_emulated_u64_three_role_kernel
    pytest.param("polymul_xcd65536", 65536, 1, 8, 0, 1, 0, 16, "random",
                 marks=pytest.mark.skipif(not os.environ.get("NFL_EMU_FULL"), reason="set NFL_EMU_FULL=1 (50 s)")),
])
def test_emulated_one_launch_plan(stem, n, nm, batch, dlog, rlog, pooled, wgs, order, generated, oracle_factory):
    """the persistent one-launch plan (credit / ticket scheduler, per-XCD domains, completion counters): all workgroups
    resident, interleaved at their polls in the given order; a wait that never ends raises"""
    o = oracle_factory(64, n, nm)
    prm, a, b = operands(o, 64, n, nm, batch, 19)
    got = asm_emu.run_xcd_product(generated(stem), n, nm, prm, a, b, dlog, rlog, pooled, wgs, _picker(order))
    assert np.array_equal(got, o.polymul(a, b))


def test_emulated_one_launch_plan_fails_loudly_when_it_cannot_progress(generated, oracle_factory, tmp_path):
    """a scheduler that never posts the forward credits of later rows (the store that posts them dropped): the kernel's
    bounded wait traps (or the interpreter sees that nothing changes any more) instead of hanging"""
    import re
    o = oracle_factory(64, 32768, 1)
    prm, a, b = operands(o, 64, 32768, 1, 32, 20)
    with open(generated("polymul_xcd32768")) as f:
        text = f.read()
    mutant = tmp_path / "no_credits.s"
    mutant.write_text(re.sub(r"global_atomic_add[^\n]*\n", "", text))     # every credit / ticket post dropped
    with pytest.raises((RuntimeError, asm_emu.StrictError), match="s_trap|stuck|outside|differs|stale"):
        got = asm_emu.run_xcd_product(str(mutant), 32768, 1, prm, a, b, 0, 1, 0, 8, _picker("random"), spin=50)
        assert np.array_equal(got, o.polymul(a, b)), "differs"


def _mutant(path, tmp_path, pick, edit):
    """a copy of a generated source with ONE line changed: pick(lines) -> index, edit(line) -> replacement (None = drop)"""
    with open(path) as f:
        lines = f.read().split("\n")
    i = pick(lines)
    new = edit(lines[i])
    lines[i:i + 1] = [] if new is None else [new]
    out = tmp_path / os.path.basename(path)
    out.write_text("\n".join(lines))
    return str(out)


def test_strict_mode_catches_what_the_gpu_might_forgive(generated, oracle_factory, tmp_path):
    """the interpreter is not only an arithmetic model: a relaxed wait count, a dropped hazard nop and a dropped barrier --
    each of which would usually still pass on hardware -- are errors"""
    import re
    o = oracle_factory(32, 2048, 1)
    prm, a, b = operands(o, 32, 2048, 1, 1, 21)
    src = generated("row2048_u32")

    def nth(pattern, k):
        return lambda lines: [i for i, l in enumerate(lines) if re.search(pattern, l)][k]

    assert np.array_equal(asm_emu.run_row_kernel(src, 32, 2048, 1, prm, a, b, 2, True), o.polymul(a, b))
    relaxed = _mutant(src, tmp_path, nth(r"s_waitcnt vmcnt\(\d+\)", 3),
                      lambda l: re.sub(r"vmcnt\((\d+)\)", lambda m: "vmcnt(%d)" % (int(m.group(1)) + 1), l))
    with pytest.raises(asm_emu.StrictError, match="still outstanding"):
        asm_emu.run_row_kernel(relaxed, 32, 2048, 1, prm, a, b, 2, True)
    no_barrier = _mutant(src, tmp_path, nth(r"s_barrier", 2), lambda l: None)
    with pytest.raises(asm_emu.StrictError, match="no barrier in between"):
        asm_emu.run_row_kernel(no_barrier, 32, 2048, 1, prm, a, b, 2, True)
    # the 64-bit metric kernel: a carry written by one VALU instruction and consumed two slots later
    o64 = oracle_factory(64, 4096, 1)
    prm64, a64, b64 = operands(o64, 64, 4096, 1, 1, 22)
    no_nop = _mutant(generated("polymul4096nt"), tmp_path, nth(r"s_nop", 5), lambda l: None)
    with pytest.raises(asm_emu.StrictError, match="wait state"):
        asm_emu.run_block_kernel(no_nop, 4096, 1, prm64, a64, b64, 12)


def test_strict_mode_models_the_measured_visibility_rules(generated, oracle_factory, tmp_path):
    """polls of the one-launch plan read with plain loads instead of sc1: the second poll of a counter may be served by the
    CU's L1 (profiles/r02_l2_flag_probe.txt) -- reported, although the interpreter's own memory is coherent"""
    import re
    o = oracle_factory(64, 32768, 1)
    prm, a, b = operands(o, 64, 32768, 1, 8, 23)
    with open(generated("polymul_xcd32768")) as f:
        text = f.read()
    mutant = tmp_path / "plain_polls.s"
    mutant.write_text(re.sub(r"(global_load_dword .*) sc1\n", r"\1\n", text))
    with pytest.raises(asm_emu.StrictError, match="stale cache line"):
        asm_emu.run_xcd_product(str(mutant), 32768, 1, prm, a, b, 0, 1, 0, 16, _picker("random"))
