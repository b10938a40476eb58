"""GPU parity of the transform-fused pipelines (include/nflhip.h "transform-fused pipelines"): the LWE demo's encrypt() /
decrypt() bodies (reference tests/nfllib_demo_main_op.cpp:26-58) as ONE device pass each, against

  * the CPU oracle run operator by operator (ntt_pow_phi, operator*, operator+ / operator-, invntt_pow_invphi), and
  * the engine's own unfused sequence (nflhip_ntt_fwd_dev / nflhip_eval_dev / nflhip_ntt_inv_dev), and
  * the context created under NFLHIP_VARIANT=hipcc, which composes the same result from the compiled kernels,

bit for bit.  u64 / 4096, 8192 and 16384 run the generated gfx950 kernels (tools/gen_polymul_asm.py build_fused /
build_fused_rows), u64 / 32768 the inverse pipelines' (build_row32k); every other shape the composed plan behind the same entry points.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))
EXPR_ADD, EXPR_SUB, EXPR_MUL = 0x10, 0x11, 0x12
SHAPES = [(64, 4096, 4), (64, 4096, 1), (64, 4096, 3), (64, 1024, 2), (64, 8192, 2), (64, 8192, 1), (64, 16384, 8), (64, 16384, 1),
          (64, 32768, 2), (32, 1024, 1), (32, 4096, 2), (16, 128, 1), (64, 2048, 3), (32, 2048, 2), (32, 1024, 2)]


def _words(o, batch, seed, operand=0):
    return o.fill_uniform(batch, seed, operand)


def _signed_rows(x, P, dtype):
    v = x.astype(np.int64)[:, None, :]
    return np.where(v < 0, P[None, :, None].astype(np.int64) + v, v).astype(dtype)


def _compact(rng, np_fmt, batch, n, bound):
    info = np.iinfo(np_fmt)
    lo, hi = max(info.min, -bound), min(info.max, bound)
    x = rng.integers(lo, hi, size=(batch, n), endpoint=True).astype(np_fmt)
    x[0, :4] = (lo, hi, 0, -1)
    return x


def _bc(k, like):
    return np.ascontiguousarray(np.broadcast_to(k, like.shape))


@pytest.mark.parametrize("lb,n,nm", SHAPES)
@pytest.mark.parametrize("fmt", ["words", "i8", "i16", "i32"])
def test_forward_multiply_add_pipelines(lb, n, nm, fmt, engine_factory, compiled_engine_factory, oracle_factory):
    import torch
    e, o = engine_factory(lb, n, nm), oracle_factory(lb, n, nm)
    P = np.asarray(e.P, dtype=np.uint64)
    rng = np.random.default_rng(n * 31 + nm)
    for batch in (1, 5, 37):
        ka, kb = _words(o, 1, 7, 0), _words(o, 1, 7, 1)
        if fmt == "words":
            xs_h = [_words(o, batch, 11 + i, i & 1) for i in range(3)]
            w = xs_h
            xs_d = [e.to_device(x) for x in xs_h]
        else:
            np_fmt = {"i8": np.int8, "i16": np.int16, "i32": np.int32}[fmt]
            bound = min(int(P.min()) - 1, np.iinfo(np_fmt).max)
            xs_h = [_compact(rng, np_fmt, batch, n, bound) for _ in range(3)]
            w = [_signed_rows(x, P, e.np_dtype) for x in xs_h]
            xs_d = [torch.from_numpy(x).to("cuda:0") for x in xs_h]
            assert np.array_equal(e.to_host(e.expand_small(xs_d[1])), w[1])
        f = [o.ntt(x) for x in w]
        want0 = o.pointwise(0, o.pointwise(2, f[0], _bc(ka, f[0])), f[1])
        want1 = o.pointwise(0, o.pointwise(2, f[0], _bc(kb, f[0])), f[2])
        dka, dkb = e.to_device(ka), e.to_device(kb)
        got0, got1 = e.fwd_fma2(xs_d[0], dka, xs_d[1], dkb, xs_d[2])
        assert np.array_equal(e.to_host(got0), want0) and np.array_equal(e.to_host(got1), want1)
        assert np.array_equal(e.to_host(e.fwd_fma(xs_d[0], dkb, xs_d[2])), want1)
        # dense keys (stride 1) and a shared input (stride 0)
        dense_k = e.to_device(_words(o, batch, 19, 1))
        want = o.pointwise(0, o.pointwise(2, f[0], e.to_host(dense_k)), f[1])
        assert np.array_equal(e.to_host(e.fwd_fma(xs_d[0], dense_k, xs_d[1])), want)
        one = xs_d[2][:1].contiguous()
        f2 = o.ntt(_bc(w[2][:1], w[2]))
        want = o.pointwise(0, o.pointwise(2, f[0], _bc(ka, f[0])), f2)
        assert np.array_equal(e.to_host(e.fwd_fma(xs_d[0], dka, one, batch=batch)), want)
        # the engine's own operator-by-operator sequence and the compiled-kernel context
        exp = [e.ntt_(e.expand_small(x) if fmt != "words" else x.clone()) for x in xs_d]
        seq0 = e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [exp[0], e.broadcast(dka, batch), exp[1]])
        assert torch.equal(seq0, got0)
        ce = compiled_engine_factory(lb, n, nm)
        c0, c1 = ce.fwd_fma2(xs_d[0], dka, xs_d[1], dkb, xs_d[2])
        assert torch.equal(c0, got0) and torch.equal(c1, got1)
        if fmt == "words":   # a result may alias a dense input
            alias = xs_d[1].clone()
            e.fwd_fma(xs_d[0], dka, alias, out=alias)
            assert torch.equal(alias, got0)
            # ... including an input of the OTHER result: out0 over e1's array and out1 over e0's,
            e0c, e1c = xs_d[1].clone(), xs_d[2].clone()
            e.fwd_fma2(xs_d[0], dka, e0c, dkb, e1c, out0=e1c, out1=e0c)
            assert torch.equal(e1c, got0) and torch.equal(e0c, got1)
            # ... and out0 over a per-element k1 (the composed plan holds the first result back until the second is done)
            k1d = dense_k.clone()
            want1d = o.pointwise(0, o.pointwise(2, f[0], e.to_host(dense_k)), f[2])
            _, o1 = e.fwd_fma2(xs_d[0], dka, xs_d[1], k1d, xs_d[2], out0=k1d)
            assert torch.equal(k1d, got0) and np.array_equal(e.to_host(o1), want1d)


@pytest.mark.parametrize("lb,n,nm", SHAPES)
def test_multiply_subtract_inverse_pipeline(lb, n, nm, engine_factory, compiled_engine_factory, oracle_factory):
    import torch
    e, o = engine_factory(lb, n, nm), oracle_factory(lb, n, nm)
    for batch in (1, 6, 33):
        a, b, s = _words(o, batch, 3, 0), _words(o, batch, 3, 1), _words(o, 1, 5, 0)
        prod = o.pointwise(2, a, _bc(s, a))
        da, db, ds = e.to_device(a), e.to_device(b), e.to_device(s)
        got = e.fma_inv(da, ds, db, subtract=True)
        assert np.array_equal(e.to_host(got), o.intt(o.pointwise(1, b, prod)))
        got_add = e.fma_inv(da, ds, db, subtract=False)
        assert np.array_equal(e.to_host(got_add), o.intt(o.pointwise(0, b, prod)))
        seq = e.intt_(e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_SUB]), [db, da, e.broadcast(ds, batch)]))
        assert torch.equal(seq, got)
        ce = compiled_engine_factory(lb, n, nm)
        assert torch.equal(ce.fma_inv(da, ds, db, subtract=True), got)
        alias = db.clone()
        e.fma_inv(da, ds, alias, subtract=True, out=alias)
        assert torch.equal(alias, got)
        kk = _words(o, batch, 7, 1)                                   # a key per element (stride 1), result over the first operand
        alias_a = da.clone()
        e.fma_inv(alias_a, e.to_device(kk), db, subtract=False, out=alias_a)
        assert np.array_equal(e.to_host(alias_a), o.intt(o.pointwise(0, b, o.pointwise(2, a, kk))))


@pytest.mark.parametrize("lb,n,nm", [(64, 4096, 4), (64, 1024, 2), (32, 1024, 1)])
def test_compact_gaussian_polynomials_expand_to_the_full_sampler(lb, n, nm, engine_factory):
    """nflhip_sample_gauss_small[_seq]_dev + nflhip_expand_small_dev == nflhip_sample_gauss[_seq]_dev, bit for bit, in every
    format the samples fit; formats they do not fit are refused"""
    import torch
    from nfllib_amd import NflHipError
    from nfllib_amd._lib import FMT_I8, FMT_I16, FMT_I32
    e = engine_factory(lb, n, nm)
    g = e.gauss_create(3.19, 128, 1 << 10)
    try:
        for batch, amp in ((1, 1), (9, 2)):
            full = e.sample_gauss(e.empty(batch), g, KEY, stream_id=77, amplifier=amp)
            seq = e.sample_gauss_seq(e.empty(batch), g, KEY, 500, 3, amplifier=amp)
            for fmt in (FMT_I8, FMT_I16, FMT_I32):
                small = e.sample_gauss_small(e.empty_small(batch, fmt), g, KEY, stream_id=77, amplifier=amp)
                assert torch.equal(e.expand_small(small), full)
                small = e.sample_gauss_small_seq(e.empty_small(batch, fmt), g, KEY, 500, 3, amplifier=amp)
                assert torch.equal(e.expand_small(small), seq)
            # shards of one logical batch
            part = e.sample_gauss_small(e.empty_small(batch, FMT_I8), g, KEY, stream_id=77, amplifier=amp, first_poly=4)
            whole = e.sample_gauss_small(e.empty_small(batch + 4, FMT_I8), g, KEY, stream_id=77, amplifier=amp)
            assert torch.equal(part, whole[4:])
        with pytest.raises(NflHipError):
            e.sample_gauss_small(e.empty_small(1, FMT_I8), g, KEY, stream_id=1, amplifier=1000)
    finally:
        e.gauss_destroy(g)


def test_lwe_encrypt_decrypt_through_the_fused_pipelines(engine_factory, oracle_factory):
    """the reference's demo end to end (tests/nfllib_demo_main_op.cpp:260-332) on compact noise and two launches per
    encryption batch, one per decryption batch: every decrypted coefficient is even and small"""
    import torch
    from nfllib_amd import DIST_UNIFORM
    from nfllib_amd._lib import FMT_I8
    e = engine_factory(64, 4096, 4)
    g = e.gauss_create(3.19, 128, 1 << 10)
    try:
        B = 64
        s = e.ntt_(e.sample_gauss(e.empty(1), g, KEY, 1))
        pka = e.sample(e.empty(1), DIST_UNIFORM, KEY, 2)
        pkb = e.ntt_(e.sample_gauss(e.empty(1), g, KEY, 3, amplifier=2))
        pkb = e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_ADD]), [pkb, pka, s])
        u = e.sample_gauss_small(e.empty_small(B, FMT_I8), g, KEY, 10)
        e1 = e.sample_gauss_small(e.empty_small(B, FMT_I8), g, KEY, 11, amplifier=2)
        e2 = e.sample_gauss_small(e.empty_small(B, FMT_I8), g, KEY, 12, amplifier=2)
        resa, resb = e.fwd_fma2(u, pka, e1, pkb, e2)
        dec = e.to_host(e.fma_inv(resa, s, resb, subtract=True))
        P0 = e.P[0]
        v = dec[:, 0, :].astype(object)
        bits = np.where(v < P0 // 2, v % 2, 1 - v % 2)
        assert (bits == 0).all()
        noise = np.where(v < P0 // 2, v, v - P0).astype(np.float64)
        assert np.abs(noise).max() < 1 << 20
        # and the same ciphertexts from the unfused operator sequence
        U, E1 = e.ntt_(e.expand_small(u)), e.ntt_(e.expand_small(e1))
        ref = e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [U, e.broadcast(pka, B), E1])
        assert torch.equal(ref, resa)
    finally:
        e.gauss_destroy(g)


def test_fused_entries_reject_bad_arguments(engine_factory):
    from nfllib_amd import NflHipError
    e = engine_factory(64, 4096, 4)
    x = e.empty(2)
    import torch
    with pytest.raises(ValueError):
        e.fma_inv(torch.zeros((2, 4096), dtype=torch.int8, device="cuda:0"), x, x)   # compact operands only feed the forward entries
    with pytest.raises(NflHipError):
        e._chk(e.lib.nflhip_fwd_fma_dev(e.ctx, None, None, None, None, 2, None))


def test_both_grids_of_the_fused_kernels_give_the_same_words(engine_factory):
    """the 2-D (element, modulus) grid and the 1-D grid that deals the nm rows of an element to one XCD (include/nflhip_debug.h
    nflhip_debug_fused_grid), every fused entry, ragged batches (the 1-D grid pads to groups of eight elements)"""
    import torch
    from nfllib_amd import _lib
    from nfllib_amd._lib import FMT_I16
    e = engine_factory(64, 4096, 4)
    g = e.gauss_create(3.19, 128, 1 << 10)
    try:
        for batch in (1, 7, 8, 9, 100):
            x = e.sample_gauss_small(e.empty_small(batch, FMT_I16), g, KEY, 40, amplifier=3)
            w = e.fill_uniform(e.empty(batch), 5, 0)
            k0, k1 = e.fill_uniform(e.empty(1), 6, 0), e.fill_uniform(e.empty(batch), 6, 1)
            got = []
            for mode in (1, 2, 3, 0):
                _lib.lib.nflhip_debug_fused_grid(mode)
                a0, a1 = e.fwd_fma2(x, k0, w, k1, x)
                got.append((a0, a1, e.fwd_fma(w, k1, x), e.fma_inv(a0, k0, a1, subtract=True), e.fma_inv(a0, k1, a1)))
            for other in got[1:]:
                assert all(torch.equal(p, q) for p, q in zip(got[0], other)), batch
    finally:
        _lib.lib.nflhip_debug_fused_grid(0)
        e.gauss_destroy(g)


@pytest.mark.parametrize("lb,n,nm", [(32, 1024, 2), (32, 2048, 3), (32, 4096, 2), (64, 1024, 2), (64, 2048, 1)])
def test_generated_and_compiled_wave_per_row_pipelines_give_the_same_words(lb, n, nm, engine_factory):
    """rows of 1024 / 2048 (/ 4096 at 32-bit limbs) words: the generated fused kernels (tools/gen_row1024_u32_asm.py build_fwd_fma /
    build_fma_inv, tools/asmgen/rows1k.py) and the compiled one-pass template they replace (nflhip_debug_fused_grid(4)), int8 and word
    operands, shared and dense keys, ragged batches (surplus waves / rows in the last workgroup)"""
    import torch
    from nfllib_amd import _lib
    from nfllib_amd._lib import FMT_I8
    e = engine_factory(lb, n, nm)
    g = e.gauss_create(3.19, 128, 1 << 10)
    try:
        for batch in (1, 3, 10, 65):
            xs = [e.sample_gauss_small(e.empty_small(batch, FMT_I8), g, KEY, 40 + i) for i in range(3)]
            ws = [e.fill_uniform(e.empty(batch), 5, i) for i in range(3)]
            k0, k1 = e.fill_uniform(e.empty(1), 6, 0), e.fill_uniform(e.empty(batch), 6, 1)
            got = []
            for mode in (4, 0):
                _lib.lib.nflhip_debug_fused_grid(mode)
                a0, a1 = e.fwd_fma2(xs[0], k0, xs[1], k1, xs[2])
                b0, b1 = e.fwd_fma2(ws[0], k1, ws[1], k0, ws[2])
                got.append((a0, a1, b0, b1, e.fwd_fma(xs[2], k1, xs[0]), e.fwd_fma(ws[1], k0, ws[0]),
                            e.fma_inv(a0, k0, a1, subtract=True), e.fma_inv(b0, k1, b1, subtract=False)))
            assert all(torch.equal(p, q) for p, q in zip(got[0], got[1])), batch
    finally:
        _lib.lib.nflhip_debug_fused_grid(0)
        e.gauss_destroy(g)


def test_ambiguous_operand_shapes_are_refused(engine_factory):
    """torch has no unsigned 16 / 32-bit types: a u32 ring's words and the compact int32 format share a dtype, so the shape
    must name the format -- [count][nmoduli][degree] words, [count][degree] compact -- and anything else raises."""
    import torch
    e = engine_factory(32, 1024, 2)
    w, k = e.empty(4), e.empty(1)
    good = e.fwd_fma(w, k, w.clone())
    assert good.shape == w.shape
    compact = torch.zeros((4, 1024), dtype=torch.int32, device=w.device)
    e.fwd_fma(compact, k, compact)                                    # 2-D int32: one signed integer per coefficient
    for bad in (w.view(-1), w.view(4, 2048), w.view(4, 2, 32, 32), compact.view(-1), compact.view(4, 2, 512)):
        with pytest.raises(ValueError):
            e.fwd_fma(bad, k, w)
    with pytest.raises(ValueError):
        e.fwd_fma(w, compact, w)                                       # keys are words


def test_compact_rows_at_an_odd_offset_take_the_composed_plan(engine_factory, oracle_factory):
    """the wave-per-row forward kernels fetch compact rows 16 bytes per lane; an int8 array that starts at an odd byte is still a
    legal operand (one signed integer per coefficient) and gives the same words through the composed plan"""
    import torch
    lb, n, nm, batch = 32, 1024, 2, 5
    e, o = engine_factory(lb, n, nm), oracle_factory(lb, n, nm)
    P = np.asarray(e.P, dtype=np.uint64)
    rng = np.random.default_rng(3)
    raw = [torch.from_numpy(rng.integers(-100, 100, size=batch * n + 16, dtype=np.int8)).to("cuda:0") for _ in range(3)]
    odd = [r[3:3 + batch * n].view(batch, n) for r in raw]              # storage offset 3 bytes
    assert all(t.data_ptr() % 16 == 3 and t.is_contiguous() for t in odd)
    even = [t.clone() for t in odd]
    ka, kb = e.to_device(o.fill_uniform(1, 7, 0)), e.to_device(o.fill_uniform(1, 7, 1))
    g0, g1 = e.fwd_fma2(odd[0], ka, odd[1], kb, odd[2])
    w0, w1 = e.fwd_fma2(even[0], ka, even[1], kb, even[2])
    assert torch.equal(g0, w0) and torch.equal(g1, w1)
    w = [_signed_rows(t.cpu().numpy(), P, e.np_dtype) for t in even]
    f = [o.ntt(x) for x in w]
    assert np.array_equal(e.to_host(g0), o.pointwise(0, o.pointwise(2, f[0], _bc(e.to_host(ka), f[0])), f[1]))
