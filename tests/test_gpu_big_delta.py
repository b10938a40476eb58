"""-m gpu: u64 contexts that reach past the 92nd mirrored 62-bit modulus.

The reference allows nfl::poly<uint64_t, n, k> for k up to kMaxNbModuli = 1000 (params.hpp:82-119).  Its primes are
2^62 - delta with delta growing along the table; from the 93rd on (index 92) delta >= 2^32, Shape::small_delta is cleared at
context creation and EVERY transform, product and fused entry of such a context is served by the general-modulus kernel
family (kernels_fast.hip k_polymul4096<..,0,2> / k_ntt_fwd4096<0> / k_ntt_inv4096<0,..>, kernels_generic.hip) instead
of the delta-form assembly -- DESIGN.md section 5 "contexts with moduli beyond #92" lists which kernel serves which shape.
These tests put every entry point of that family against the CPU oracle, which tests/test_oracle_vs_ref.py pins
memcmp-equal on the real reference at poly<uint64_t,64,96> and poly<uint64_t,1024,94>, and tests/golden holds the real
reference's digests for (test_gpu_golden.py runs them on the device too).

Inputs: the seeded uniform stream, all-(p-1) rows (largest products, folds, quotients), alternating p-1 / 0.
"""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

# (degree, nmoduli, batch): 93..96 moduli, every row length the u64 launchers distinguish
SHAPES = [(64, 96, 3), (1024, 94, 3), (2048, 93, 3), (4096, 95, 3), (8192, 94, 2), (16384, 96, 2), (32768, 93, 2), (65536, 94, 1)]
IDS = ["n%d-m%d" % s[:2] for s in SHAPES]
ADD, SUB, MUL, MSH, CSH = 0x10, 0x11, 0x12, 0x13, 0x14


def _ctx(n, m, oracle_factory, engine_factory):
    o, e = oracle_factory(64, n, m), engine_factory(64, n, m)
    assert (1 << 62) - int(o.P[91]) < (1 << 32) <= (1 << 62) - int(o.P[92]), "the mirrored table no longer splits at #92"
    return o, e


def _inputs(o, n, batch):
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    pm1 = (np.array([int(p) for p in o.P], dtype=np.uint64) - 1)[:, None]
    a[0], b[0] = pm1, pm1
    if batch > 1:
        b[1] = np.where(np.arange(n) % 2 == 0, pm1, 0)
    return a, b


def _bc(k, like):
    return np.ascontiguousarray(np.broadcast_to(k, like.shape))


@pytest.mark.parametrize("n,m,batch", SHAPES, ids=IDS)
def test_transforms_and_products(n, m, batch, oracle_factory, engine_factory):
    import torch
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    a, b = _inputs(o, n, batch)
    da, db = e.to_device(a), e.to_device(b)
    want = o.polymul(a, b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), want)
    fa = e.ntt_(da.clone())
    assert np.array_equal(e.to_host(fa), o.ntt(a)), "ntt_pow_phi"
    assert np.array_equal(e.to_host(e.intt_(da.clone())), o.intt(a)), "invntt_pow_invphi"
    assert np.array_equal(e.to_host(e.intt_(fa.clone())), a), "round trip"
    assert np.array_equal(e.to_host(e.polymul(db, fa, b_is_ntt=True)), want), "b_is_ntt"
    # in place: the product over either operand
    x = da.clone(); e.polymul(x, db, out=x)
    assert np.array_equal(e.to_host(x), want)
    x = db.clone(); e.polymul(da, x, out=x)
    assert np.array_equal(e.to_host(x), want)
    x = da.clone(); e.polymul(x, x, out=x)
    assert np.array_equal(e.to_host(x), o.polymul(a, a)), "square, in place"
    assert torch.equal(e.polymul(da, db), e.polymul(db, da))


@pytest.mark.parametrize("n,m,batch", SHAPES, ids=IDS)
def test_pointwise_and_expression_trees(n, m, batch, oracle_factory, engine_factory):
    from nfllib_amd import OP_ADD, OP_COMPUTE_SHOUP, OP_MUL, OP_MUL_SHOUP, OP_SUB
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    a, b = _inputs(o, n, batch)
    c = o.fill_uniform(batch, 77, 0)
    da, db, dc = e.to_device(a), e.to_device(b), e.to_device(c)
    for op in (OP_ADD, OP_SUB, OP_MUL):
        assert np.array_equal(e.to_host(e.pointwise(op, da, db)), o.pointwise(op, a, b)), op
    bp = o.pointwise(OP_COMPUTE_SHOUP, b)
    dbp = e.pointwise(OP_COMPUTE_SHOUP, db)
    assert np.array_equal(e.to_host(dbp), bp)
    assert np.array_equal(e.to_host(e.pointwise(OP_MUL_SHOUP, da, db, dbp)), o.pointwise(OP_MUL_SHOUP, a, b, bp))
    out = da.clone(); e.pointwise(OP_SUB, out, db, out=out)
    assert np.array_equal(e.to_host(out), o.pointwise(OP_SUB, a, b)), "aliasing"
    want = o.pointwise(OP_SUB, b, o.pointwise(OP_MUL, a, c))                                   # b - a*c
    assert np.array_equal(e.to_host(e.eval([1, 0, 2, MUL, SUB], [da, db, dc])), want)
    want = o.pointwise(OP_ADD, o.pointwise(OP_MUL, o.pointwise(OP_ADD, a, b), o.pointwise(OP_SUB, a, c)), o.pointwise(OP_MUL, b, c))
    out = da.clone()
    e.eval([0, 1, ADD, 0, 2, SUB, MUL, 1, 2, MUL, ADD], [out, db, dc], out=out)
    assert np.array_equal(e.to_host(out), want)
    want = o.pointwise(OP_ADD, o.pointwise(OP_MUL, a, b), c)
    assert np.array_equal(e.to_host(e.eval([0, 1, 1, CSH, MSH, 2, ADD], [da, db, dc])), want)
    assert e.any_neq(da, dc) and not e.any_neq(da, da) and e.any_eq(da, da)
    d2 = dc.clone(); d2[d2 == da] += 1
    assert not e.any_eq(da, d2)
    d2[batch - 1, m - 1, n - 1] = da[batch - 1, m - 1, n - 1]                                   # one equal lane in the LAST (big-delta) row
    assert e.any_eq(da, d2)


@pytest.mark.parametrize("n,m,batch", SHAPES, ids=IDS)
@pytest.mark.parametrize("fmt", ["words", "i8"])
def test_fused_pipelines(n, m, batch, fmt, oracle_factory, engine_factory):
    """nflhip_fwd_fma[2]_dev / nflhip_fma_inv_dev (the LWE encrypt / decrypt bodies, tests/nfllib_demo_main_op.cpp:26-58)."""
    import torch
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    P = np.asarray(e.P, dtype=np.uint64)
    ka, kb = o.fill_uniform(1, 7, 0), o.fill_uniform(1, 7, 1)
    if fmt == "words":
        w = [o.fill_uniform(batch, 11 + i, i & 1) for i in range(3)]
        w[0][0] = (P - 1)[:, None]
        xs = [e.to_device(x) for x in w]
    else:
        rng = np.random.default_rng(n + m)
        small = [rng.integers(-128, 127, size=(batch, n), endpoint=True).astype(np.int8) for _ in range(3)]
        small[0][0, :4] = (-128, 127, 0, -1)
        v = [x.astype(np.int64)[:, None, :] for x in small]
        w = [np.where(x < 0, P[None, :, None].astype(np.int64) + x, x).astype(np.uint64) for x in v]
        xs = [torch.from_numpy(x).to("cuda:0") for x in small]
        assert np.array_equal(e.to_host(e.expand_small(xs[1])), w[1])
    f = [o.ntt(x) for x in w]
    want0 = o.pointwise(0, o.pointwise(2, f[0], _bc(ka, f[0])), f[1])
    want1 = o.pointwise(0, o.pointwise(2, f[0], _bc(kb, f[0])), f[2])
    dka, dkb = e.to_device(ka), e.to_device(kb)
    got0, got1 = e.fwd_fma2(xs[0], dka, xs[1], dkb, xs[2])
    assert np.array_equal(e.to_host(got0), want0) and np.array_equal(e.to_host(got1), want1)
    assert np.array_equal(e.to_host(e.fwd_fma(xs[0], dkb, xs[2])), want1)
    if fmt == "words":
        dense_k = o.fill_uniform(batch, 19, 1)
        want = o.pointwise(0, o.pointwise(2, f[0], dense_k), f[1])
        assert np.array_equal(e.to_host(e.fwd_fma(xs[0], e.to_device(dense_k), xs[1])), want), "a key per element"
        alias = xs[1].clone()
        e.fwd_fma(xs[0], dka, alias, out=alias)
        assert torch.equal(alias, got0), "result over a dense input"
        e0c, e1c = xs[1].clone(), xs[2].clone()
        e.fwd_fma2(xs[0], dka, e0c, dkb, e1c, out0=e1c, out1=e0c)
        assert torch.equal(e1c, got0) and torch.equal(e0c, got1), "results over the other result's input"
        # decrypt body: INTT(b -/+ a*s)
        a, b = w[0], w[1]
        prod = o.pointwise(2, a, _bc(ka, a))
        assert np.array_equal(e.to_host(e.fma_inv(xs[0], dka, xs[1], subtract=True)), o.intt(o.pointwise(1, b, prod)))
        assert np.array_equal(e.to_host(e.fma_inv(xs[0], dka, xs[1], subtract=False)), o.intt(o.pointwise(0, b, prod)))
        alias = xs[1].clone()
        e.fma_inv(xs[0], dka, alias, subtract=True, out=alias)
        assert np.array_equal(e.to_host(alias), o.intt(o.pointwise(1, b, prod))), "in place over b"


@pytest.mark.parametrize("n,m", [(64, 96), (1024, 94), (4096, 95), (16384, 96)])
def test_cyclic_rows_and_tables_of_the_late_moduli(n, m, oracle_factory, engine_factory):
    """core::ntt (core.hpp:455-532) per row and the reference-layout tables, for the moduli the delta form excludes."""
    from nfllib_amd import engine as E
    from oracle import oracle as O
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    a = o.fill_uniform(2, SEED, 0)
    a[0] = (np.array([int(p) for p in o.P], dtype=np.uint64) - 1)[:, None]
    for cm in (91, 92, m - 1):
        x = np.ascontiguousarray(a[:, cm, :])
        for inv in (False, True):
            want = np.stack([o.ntt_row(r, cm, inv) for r in x])
            assert np.array_equal(e.to_host(e.ntt_row_(e.to_device(x), cm, inverse_tables=inv)), want), (cm, inv)
        for et, ot in ((E.TAB_PHIS, O.TAB_PHIS), (E.TAB_SHOUPPHIS, O.TAB_SHOUPPHIS), (E.TAB_INVPOLY_INVPHIS, O.TAB_INVPOLY_INVPHIS),
                       (E.TAB_OMEGAS, O.TAB_OMEGAS), (E.TAB_INVOMEGAS, O.TAB_INVOMEGAS)):
            assert np.array_equal(e.table(et, cm), o.table(ot, cm)), (cm, et)


@pytest.mark.parametrize("n,m", [(1024, 94), (4096, 95)])
def test_host_pointer_entries_and_crt(n, m, oracle_factory, engine_factory):
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    a, b = _inputs(o, n, 2)
    assert np.array_equal(e.h_polymul(a, b), o.polymul(a, b))
    assert np.array_equal(e.h_ntt(a), o.ntt(a)) and np.array_equal(e.h_intt(a), o.intt(a))
    lifted = e.h_crt_lift(a)
    assert np.array_equal(lifted, o.crt_lift(a))
    assert np.array_equal(e.h_crt_project(lifted), a)            # tests/poly_mpz.cpp:19-29


def test_same_rows_as_a_small_delta_context(oracle_factory, engine_factory):
    """Rows 0..91 of a 95-modulus context are the rows of the 92-modulus context: two kernel families, one answer."""
    n = 4096
    o95, e95 = _ctx(n, 95, oracle_factory, engine_factory)
    e92 = engine_factory(64, n, 92)
    a, b = _inputs(o95, n, 2)
    got95 = e95.to_host(e95.polymul(e95.to_device(a), e95.to_device(b)))
    a92, b92 = np.ascontiguousarray(a[:, :92]), np.ascontiguousarray(b[:, :92])
    got92 = e92.to_host(e92.polymul(e92.to_device(a92), e92.to_device(b92)))
    assert np.array_equal(got95[:, :92], got92)


def test_samplers_on_the_late_moduli(oracle_factory, engine_factory):
    """the random constructors are modulus-generic (mask, one conditional subtraction; v < 0 -> p + v): wide and narrow draws on a
    94-modulus context against the restated rules fed with the very keystream words"""
    from nfllib_amd import DIST_UNIFORM
    from oracle import samplers as S
    n, m, batch = 1024, 94, 2
    o, e = _ctx(n, m, oracle_factory, engine_factory)
    P = [int(p) for p in o.P]
    key = bytes(range(32))
    words = S.chacha20_words(key, 3, 0, batch * m * n, counter_base=S.domain_base("uniform")).reshape(batch, m, n)
    assert np.array_equal(e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, key, stream_id=3)), S.uniform(words, P))
    lanes = S.uniform_narrow_words(key, 3, 0, batch * m * n, 64).reshape(batch, m, n)
    assert np.array_equal(e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, key, stream_id=3, narrow=True)), S.uniform(lanes, P))
    for bits, rule in ((64, S.gaussian_words), (32, S.gaussian_words_narrow)):
        g = e.gauss_create(3.19, 128, n, draw_bits=bits)
        info = e.gauss_info(g)
        want = S.gaussian_from_table(rule(key, 5, 0, batch * n, info["words"]), info["table"], info["x_min"])
        got = S.centered(e.to_host(e.sample_gauss(e.empty(batch), g, key, stream_id=5)), P)
        assert all(np.array_equal(got[:, cm].reshape(-1), want) for cm in (0, 91, 92, m - 1)), bits
        e.gauss_destroy(g)
