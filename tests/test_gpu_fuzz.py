"""Shape / batch / operation fuzz of the C ABI against the oracle (hypothesis seeded with conftest.FUZZ_SEED: NFL_FUZZ_SEED, or a
value derived from the tree under test -- the same examples for one tree, new ones for the next; printed in the pytest header).
The parametrised parity tests pin the shapes of BASELINE.json; this one walks the space between them -- every limb width,
degrees 4 .. 8192, 1 .. 32 moduli, ragged batches, random expression programs, aliasing of the destination."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, seed as hyp_seed, settings, strategies as st

from conftest import FUZZ_SEED

pytestmark = pytest.mark.gpu

ADD, SUB, MUL = 0x10, 0x11, 0x12
_RAN = []

_shape = st.one_of(
    st.tuples(st.just(64), st.integers(2, 13), st.integers(1, 6)),
    st.tuples(st.just(32), st.integers(2, 12), st.integers(1, 5)),
    st.tuples(st.just(16), st.integers(2, 9), st.integers(1, 2)),
    st.tuples(st.just(64), st.integers(2, 8), st.integers(7, 32)),    # many moduli (CRT tables up to 32 limbs), short rows
    st.tuples(st.just(32), st.integers(2, 8), st.integers(6, 32)),
)


def _program(draw_ops, noperands):
    """a random valid postfix program: a left-leaning tree over the operands with random operators"""
    prog = [0]
    for k, op in enumerate(draw_ops, start=1):
        prog += [k % noperands, op]
    return prog


@settings(max_examples=300, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@hyp_seed(FUZZ_SEED)
@given(shape=_shape, batch=st.integers(1, 7), seed=st.integers(0, 2**31), ops=st.lists(st.sampled_from([ADD, SUB, MUL]), min_size=1, max_size=6),
       alias=st.integers(0, 2))
def test_random_shapes_and_operations(shape, batch, seed, ops, alias, oracle_factory, engine_factory):
    from nfllib_amd import OP_ADD, OP_MUL, OP_SUB
    _RAN.append(shape)
    lb, logn, m = shape
    n = 1 << logn
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a = o.fill_uniform(batch, seed, 0)
    b = o.fill_uniform(batch, seed, 1)
    c = o.fill_uniform(batch, seed + 1, 0)
    da, db, dc = e.to_device(a), e.to_device(b), e.to_device(c)
    # transforms and the fused products, with the destination aliasing nothing / a / b
    want = o.polymul(a, b)
    out = (e.empty(batch), da.clone(), db.clone())[alias]
    src_a = out if alias == 1 else da
    src_b = out if alias == 2 else db
    assert np.array_equal(e.to_host(e.polymul(src_a, src_b, out=out)), want)
    fb = e.ntt_(db.clone())
    assert np.array_equal(e.to_host(fb), o.ntt(b))
    assert np.array_equal(e.to_host(e.polymul(da, fb, b_is_ntt=True)), want)
    assert np.array_equal(e.to_host(e.intt_(fb)), b)
    assert np.array_equal(e.to_host(e.intt_(da.clone())), o.intt(a))
    # a random expression tree in one pass vs the oracle op by op
    opmap = {ADD: OP_ADD, SUB: OP_SUB, MUL: OP_MUL}
    hosts, devs = [a, b, c], [da, db, dc]
    acc = hosts[0]
    for k, op in enumerate(ops, start=1):
        acc = o.pointwise(opmap[op], acc, hosts[k % 3])
    prog = _program(ops, 3)
    if n * (lb // 8) >= 16:   # rows shorter than one 16-byte vector are declined by nflhip_eval (the header goes node by node)
        assert np.array_equal(e.to_host(e.eval(prog, devs)), acc)
    # CRT round trip
    if True:
        assert np.array_equal(e.to_host(e.crt_project(e.crt_lift(da))), a)


@settings(max_examples=120, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@hyp_seed(FUZZ_SEED)
@given(shape=_shape, batch=st.integers(1, 6), first=st.integers(0, 1000), sid=st.integers(0, 2**64 - 1), ub=st.integers(1, 1 << 13),
       amp=st.integers(1, 4), rho=st.integers(0, 255), key=st.binary(min_size=32, max_size=32), sigma=st.sampled_from([2.0, 3.19, 20.0]),
       sec=st.sampled_from([20, 64, 100]))
def test_random_sampler_calls(shape, batch, first, sid, ub, amp, rho, key, sigma, sec, engine_factory):
    """The random constructors on random shapes, shards (first_poly), stream ids and keys against the restated rules
    fed with the very keystream words (oracle/samplers.py, pinned on the real reference)."""
    from nfllib_amd import DIST_BOUNDED, DIST_UNIFORM, DIST_ZO
    from oracle import samplers as S
    lb, logn, m = shape
    n = 1 << logn
    e = engine_factory(lb, n, m)
    P = [int(e.table(1, cm)[0]) for cm in range(m)]
    dt = e.np_dtype
    mask = {16: 0xFFFF, 32: 0xFFFFFFFF, 64: 0xFFFFFFFFFFFFFFFF}[lb]
    words = (S.chacha20_words(key, sid, first * m * n, batch * m * n, counter_base=S.domain_base("uniform")) & np.uint64(mask)).astype(dt).reshape(batch, m, n)
    got = e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, key, stream_id=sid, first_poly=first))
    assert np.array_equal(got, S.uniform(words, P))
    cw = S.chacha20_words(key, sid, first * n, batch * n, counter_base=S.domain_base("bounded")).reshape(batch, n)
    zw = S.chacha20_words(key, sid, first * n, batch * n, counter_base=S.domain_base("zo")).reshape(batch, n)
    if ub * amp < min(P) // 2:
        d = e.sample(e.empty(batch), DIST_BOUNDED, key, stream_id=sid, param0=ub, param1=amp, first_poly=first)
        assert np.array_equal(e.to_host(d), S.non_uniform(cw, P, ub, amp, dtype=dt))
    d = e.sample(e.empty(batch), DIST_ZO, key, stream_id=sid, param0=rho, first_poly=first)
    assert np.array_equal(e.to_host(d), S.zo_dist(zw & np.uint64(0xFF), P, rho, canonical=True, dtype=dt))
    if lb == 16 and sigma > 3.2:
        return                                           # +-13 sigma does not fit below p/2 of a 14-bit modulus
    g = e.gauss_create(sigma, security=sec, samples=1024)
    info = e.gauss_info(g)
    v = S.centered(e.to_host(e.sample_gauss(e.empty(batch), g, key, stream_id=sid, first_poly=first)), P)[:, 0].reshape(-1)
    r = S.gaussian_words(key, sid, first * n, batch * n, info["words"])
    assert np.array_equal(v, S.gaussian_from_table(r, info["table"], info["x_min"]))
    e.gauss_destroy(g)


_fused_shape = st.one_of(
    st.tuples(st.just(32), st.integers(10, 12), st.integers(1, 4)),     # the wave-per-row kernels (u32: 1024 / 2048 / 4096 words)
    st.tuples(st.just(64), st.integers(10, 11), st.integers(1, 5)),     # ... u64: 1024 / 2048
    st.tuples(st.just(64), st.integers(12, 13), st.integers(1, 3)),     # the generated kernels (4096 / 8192)
    st.tuples(st.just(32), st.integers(5, 9), st.integers(1, 3)),       # composed plans
    st.tuples(st.just(16), st.integers(5, 9), st.integers(1, 2)),
)


@settings(max_examples=90, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@hyp_seed(FUZZ_SEED)
@given(shape=_fused_shape, batch=st.integers(1, 9), seed=st.integers(0, 2**31), fmt=st.sampled_from(["words", "i8", "i16", "i32"]),
       xs=st.integers(0, 1), es=st.integers(0, 1), ks=st.integers(0, 1), two=st.booleans(), alias=st.integers(0, 3), sub=st.booleans())
def test_random_fused_pipeline_calls(shape, batch, seed, fmt, xs, es, ks, two, alias, sub, oracle_factory, engine_factory):
    """nflhip_fwd_fma[2]_dev / nflhip_fma_inv_dev on random shapes, batches, operand formats, strides (0 = one polynomial for the batch)
    and result aliasing, against the oracle run operator by operator: the one-pass kernels of every row family and the composed
    plan must agree with it bit for bit."""
    import torch
    lb, logn, m = shape
    n = 1 << logn
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    P = np.asarray(e.P, dtype=np.uint64)
    rng = np.random.default_rng(seed)

    def operand(count, which):
        if fmt == "words":
            h = o.fill_uniform(count, seed + which, which & 1)
            return h, e.to_device(h)
        np_fmt = {"i8": np.int8, "i16": np.int16, "i32": np.int32}[fmt]
        bound = min(int(P.min()) - 1, np.iinfo(np_fmt).max)
        c = rng.integers(-bound, bound, size=(count, n), endpoint=True).astype(np_fmt)
        v = c.astype(np.int64)[:, None, :]
        return np.where(v < 0, P[None, :, None].astype(np.int64) + v, v).astype(e.np_dtype), torch.from_numpy(c).to("cuda:0")

    bc = lambda h: np.ascontiguousarray(np.broadcast_to(h, (batch,) + h.shape[1:]))
    xh, xd = operand(batch if xs else 1, 0)
    e0h, e0d = operand(batch if es else 1, 1)
    e1h, e1d = operand(batch if es else 1, 2)
    k0h, k1h = o.fill_uniform(batch if ks else 1, seed + 7, 0), o.fill_uniform(batch if ks else 1, seed + 7, 1)
    k0d, k1d = e.to_device(k0h), e.to_device(k1h)
    fx = o.ntt(bc(xh))
    want0 = o.pointwise(0, o.pointwise(2, fx, bc(k0h)), o.ntt(bc(e0h)))
    want1 = o.pointwise(0, o.pointwise(2, fx, bc(k1h)), o.ntt(bc(e1h)))
    # a result over a dense word input of its own result (legal aliasing), or fresh
    out0 = out1 = None
    if fmt == "words" and alias == 1 and es:
        e0d = e0d.clone(); out0 = e0d
    if fmt == "words" and alias == 2 and xs and not two:
        xd = xd.clone(); out0 = xd
    if fmt == "words" and alias == 3 and ks and two:
        k1d = k1d.clone(); out1 = k1d
    if two:
        g0, g1 = e.fwd_fma2(xd, k0d, e0d, k1d, e1d, out0=out0, out1=out1, batch=batch)
        assert np.array_equal(e.to_host(g0), want0) and np.array_equal(e.to_host(g1), want1)
    else:
        g0 = e.fwd_fma(xd, k0d, e0d, out=out0, batch=batch)
        assert np.array_equal(e.to_host(g0), want0)
    # the inverse entry on the same operands (NTT-form words): INTT(b -+ a k)
    ah, bh = o.fill_uniform(batch, seed + 11, 0), o.fill_uniform(batch, seed + 11, 1)
    prod = o.pointwise(2, ah, bc(k0h))
    wanti = o.intt(o.pointwise(1 if sub else 0, bh, prod))
    ad, bd = e.to_device(ah), e.to_device(bh)
    outi = bd if alias == 1 else (ad if alias == 2 else None)
    assert np.array_equal(e.to_host(e.fma_inv(ad, k0d, bd, subtract=sub, out=outi, batch=batch)), wanti)


@settings(max_examples=60, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@hyp_seed(FUZZ_SEED)
@given(shape=_shape, batch=st.integers(1, 6), first=st.integers(0, 1000), sid=st.integers(0, 2**64 - 1), key=st.binary(min_size=32, max_size=32),
       sigma=st.sampled_from([2.0, 3.19, 20.0]), sec=st.sampled_from([20, 64, 100]), amp=st.integers(1, 3))
def test_random_narrow_draw_calls(shape, batch, first, sid, key, sigma, sec, amp, engine_factory):
    """the narrow draws (NFLHIP_DIST_NARROW, nflhip_gauss_set_draw_bits(g, 32)) on random shapes, shards, stream ids and keys against
    the restated rules fed with the very keystream lanes"""
    import torch
    from nfllib_amd import DIST_UNIFORM
    from oracle import samplers as S
    lb, logn, m = shape
    n = 1 << logn
    e = engine_factory(lb, n, m)
    P = [int(e.table(1, cm)[0]) for cm in range(m)]
    lanes = S.uniform_narrow_words(key, sid, first * m * n, batch * m * n, lb).reshape(batch, m, n)
    got = e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, key, stream_id=sid, first_poly=first, narrow=True))
    assert np.array_equal(got, S.uniform(lanes, P))
    if lb == 16 and sigma > 3.2:
        return
    g = e.gauss_create(sigma, security=sec, samples=1024, draw_bits=32)
    info = e.gauss_info(g)
    want = S.gaussian_from_table(S.gaussian_words_narrow(key, sid, first * n, batch * n, info["words"]), info["table"], info["x_min"])
    v = S.centered(e.to_host(e.sample_gauss(e.empty(batch), g, key, stream_id=sid, first_poly=first)), P)[:, 0].reshape(-1)
    assert np.array_equal(v, want)
    if amp * max(abs(info["x_min"]), info["x_min"] + info["entries"] - 1) < min(min(P), 1 << 31):
        sm = e.sample_gauss_small(torch.empty((batch, n), dtype=torch.int32, device="cuda:0"), g, key, stream_id=sid, amplifier=amp, first_poly=first)
        assert np.array_equal(sm.cpu().numpy().reshape(-1).astype(np.int64), amp * want)
    e.gauss_destroy(g)


def test_the_walk_was_wide():
    """(runs after the fuzz above) every limb width and both ends of the degree range were visited"""
    assert len(_RAN) >= 100
    assert {lb for lb, _, _ in _RAN} == {16, 32, 64}
    assert min(l for _, l, _ in _RAN) <= 3 and max(l for _, l, _ in _RAN) >= 11 and max(m for _, _, m in _RAN) >= 16


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_crt_matrix_core_kernels_every_modulus_count_and_width(seed, oracle_factory, engine_factory):
    """GMP::poly2mpz / mpz2poly (gmp.hpp:183-219) on the matrix cores (kernels_crt_mfma.hip): every modulus count they serve
    (lift 21 .. 32, projection 17 .. 32), every input width of the projection (5 .. 64 words: one GEMM pass up to 32, two beyond), random residues plus lifted
    values whose digits are runs of ones / zeros placed at random (the carry / borrow paths between the kernel's parts)."""
    import torch
    rng = np.random.default_rng(seed)
    for m in range(17, 33):
        n, batch = 256, 3
        o, e = oracle_factory(64, n, m), engine_factory(64, n, m)
        Q = o.crt_modulus()
        a = o.fill_uniform(batch, int(rng.integers(1, 2**40)), 0)
        adv = [int(rng.integers(0, 2**62)) << int(rng.integers(0, 64 * o.crt_limbs)) for _ in range(60)]
        adv += [(1 << int(rng.integers(1, Q.bit_length()))) - 1 for _ in range(60)]
        adv += [Q - 1 - ((1 << int(rng.integers(1, Q.bit_length() - 1))) - 1) for _ in range(60)]
        adv += [((1 << 512) - 1) << (32 * int(rng.integers(0, 2 * o.crt_limbs - 16))) for _ in range(40)]
        adv = [x % Q for x in adv][:n]
        for idx, x in enumerate(adv):
            a[0, :, idx] = [x % int(p) for p in o.P[:m]]
        limbs = e.crt_lift(e.to_device(a))
        got = e.to_host(limbs).view(np.uint64)
        assert np.array_equal(got, o.crt_lift(a)), "lift, %d moduli" % m
        for idx, x in enumerate(adv):
            assert int.from_bytes(got[0, idx].tobytes(), "little") == x
        assert np.array_equal(e.to_host(e.crt_project(limbs)), a), "round trip, %d moduli" % m
        for lin in range(5, 65):
            wide = rng.integers(0, 2**63, size=(1, n, lin), dtype=np.uint64) * np.uint64(2) + rng.integers(
                0, 2, size=(1, n, lin), dtype=np.uint64)
            wide[0, 0, :] = np.uint64(0xFFFFFFFFFFFFFFFF)
            wide[0, 1, :] = 0
            wide[0, 2, :] = np.uint64(0x8080808080808080)
            wide[0, 3, :] = np.uint64(0x7F7F7F7F7F7F7F7F)
            dw = torch.from_numpy(wide.view(np.int64)).to(limbs.device)
            assert np.array_equal(e.to_host(e.crt_project(dw)), o.crt_project(wide)), "project, %d moduli, %d words" % (m, lin)
