"""-m gpu: the cyclic row transform core::ntt (core.hpp:455-532) and the reference-layout table views.

tests/ntt_perfs.cpp:121-171 (BASELINE configs[0]) reaches core::ntt and core::base through the poly_tests_proxy
friend; nflhip_ntt_row[_dev] and the NFLHIP_TAB_* views are that path through the C ABI.  The expected values come
from the oracle's ntt_row / tables, which tests/test_oracle_vs_ref.py pins memcmp-equal on the real reference."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

SHAPES = [(32, 1024, 1, 5), (64, 1024, 2, 4), (16, 128, 1, 3), (32, 8, 2, 3), (64, 4096, 4, 3), (64, 2048, 1, 2),
          (32, 4096, 2, 2), (64, 16384, 2, 2), (64, 65536, 2, 1), (32, 16384, 1, 2), (64, 64, 3, 2)]
IDS = ["u%d-n%d-m%d" % s[:3] for s in SHAPES]


@pytest.mark.parametrize("lb,n,m,rows", SHAPES, ids=IDS)
def test_cyclic_row_transform_is_core_ntt(lb, n, m, rows, oracle_factory, engine_factory):
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a = o.fill_uniform(rows, SEED, 0)
    for cm in range(m):
        x = np.ascontiguousarray(a[:, cm, :])
        for inv in (False, True):
            want = np.stack([o.ntt_row(r, cm, inv) for r in x])
            d = e.to_device(x)
            got = e.to_host(e.ntt_row_(d, cm, inverse_tables=inv))
            assert np.array_equal(got, want), (cm, inv)
            assert np.array_equal(e.h_ntt_row(x, cm, inverse_tables=inv), want), "host-pointer entry"
        # core::inv_ntt (core.hpp:539-557) = permut, ntt with the inverse tables, permut: it undoes core::ntt up to n
        f = e.ntt_row_(e.to_device(x), cm)
        back = e.to_host(e.ntt_row_(f, cm, inverse_tables=True, bitrev_io=True)).astype(object)
        p = int(o.P[cm])
        assert np.array_equal(back % p, (x.astype(object) * n) % p)


def test_negacyclic_transform_is_twist_then_cyclic(oracle_factory, engine_factory):
    """core::ntt_pow_phi (core.hpp:594-600) = multiply by phis, then core::ntt per row: the two device paths agree."""
    from nfllib_amd import OP_MUL_SHOUP
    from nfllib_amd.engine import TAB_PHIS, TAB_SHOUPPHIS
    lb, n, m = 64, 1024, 2
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a = o.fill_uniform(3, SEED, 0)
    phis = np.stack([e.table(TAB_PHIS, cm) for cm in range(m)])[None].repeat(3, 0)
    sphis = np.stack([e.table(TAB_SHOUPPHIS, cm) for cm in range(m)])[None].repeat(3, 0)
    tw = e.pointwise(OP_MUL_SHOUP, e.to_device(a), e.to_device(phis), e.to_device(sphis))
    for cm in range(m):
        rows = tw[:, cm, :].contiguous()
        e.ntt_row_(rows, cm)
        tw[:, cm, :] = rows
    assert np.array_equal(e.to_host(tw), o.ntt(a))


@pytest.mark.parametrize("lb,n,m", [(64, 1024, 2), (32, 1024, 1), (16, 128, 1), (64, 4096, 4), (32, 8, 2)])
def test_reference_layout_tables(lb, n, m, oracle_factory, engine_factory):
    from nfllib_amd import engine as E
    from oracle import oracle as O
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    pairs = [(E.TAB_PHIS, O.TAB_PHIS), (E.TAB_SHOUPPHIS, O.TAB_SHOUPPHIS), (E.TAB_INVPOLY_INVPHIS, O.TAB_INVPOLY_INVPHIS),
             (E.TAB_SHOUPINVPOLY_INVPHIS, O.TAB_SHOUPINVPOLY_INVPHIS), (E.TAB_OMEGAS, O.TAB_OMEGAS),
             (E.TAB_INVOMEGAS, O.TAB_INVOMEGAS)]
    for cm in range(m):
        for te, to in pairs:
            assert np.array_equal(e.table(te, cm), o.table(to, cm)), (cm, te)


def test_rows_argument_errors(engine_factory):
    from nfllib_amd import NflHipError
    e = engine_factory(64, 1024, 2)
    d = e.empty(1)[0, 0].contiguous()
    with pytest.raises(NflHipError, match="modulus index"):
        e.ntt_row_(d, 2)
    with pytest.raises(NflHipError):
        e._chk(e.lib.nflhip_ntt_row_dev(e.ctx, d.data_ptr(), 0, 7, 1, None))


def test_broadcast_and_random_bytes(engine_factory):
    import ctypes as C
    from oracle import samplers as S
    e = engine_factory(32, 1024, 2)
    one = e.fill_uniform(e.empty(1), SEED, 0)
    many = e.to_host(e.broadcast(one, 37))
    assert many.shape[0] == 37 and all(np.array_equal(many[k], e.to_host(one)[0]) for k in range(37))
    key = bytes(range(32))
    for nbytes in (1, 8, 63, 64, 1000, 4099):
        buf = (C.c_ubyte * nbytes)()
        e._chk(e.lib.nflhip_random_bytes(0, buf, nbytes, C.create_string_buffer(key, 32), 5))
        want = S.chacha20_words(key, 5, 0, (nbytes + 7) // 8).tobytes()[:nbytes]
        assert bytes(buf) == want


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2), (16, 128, 1), (64, 4, 2)])
def test_reference_words_mode_of_zo_and_hwt(lb, n, m, engine_factory):
    """NFLHIP_DIST_REFERENCE_WORDS: +1 is stored as p + 1 like the reference (core.hpp:341, 387); everything else,
    incl. the keystream use, is the canonical mode's."""
    from nfllib_amd import DIST_HWT, DIST_ZO
    from nfllib_amd.engine import DIST_REFERENCE_WORDS
    from oracle import samplers as S
    e = engine_factory(lb, n, m)
    P = [int(x) for x in e.P]
    key, batch = bytes(31 - i for i in range(32)), 3
    zw = S.chacha20_words(key, 4, 0, batch * n, counter_base=S.domain_base("zo")).reshape(batch, n)
    for rho in (0x7F, 255, 3):
        d = e.to_host(e.sample(e.empty(batch), DIST_ZO | DIST_REFERENCE_WORDS, key, stream_id=4, param0=rho))
        assert np.array_equal(d, S.zo_dist(zw & np.uint64(0xFF), P, rho, canonical=False, dtype=e.np_dtype)), rho
    h = max(1, n // 4)
    can = e.to_host(e.sample(e.empty(batch), DIST_HWT, key, stream_id=6, param0=h))
    ref = e.to_host(e.sample(e.empty(batch), DIST_HWT | DIST_REFERENCE_WORDS, key, stream_id=6, param0=h))
    for cm, p in enumerate(P):
        c, r = can[:, cm].astype(object), ref[:, cm].astype(object)
        assert np.array_equal(np.where(r == p + 1, 1, r), c) and set(np.unique(r).tolist()) <= {0, p - 1, p + 1}
        assert (r == p + 1).sum() == (c == 1).sum() and ((c == 1).sum() > 0 or n < 128)


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2), (16, 128, 1), (64, 8, 3), (64, 2048, 5)])
def test_sequence_samplers_equal_a_loop_of_single_calls(lb, n, m, engine_factory):
    """nflhip_sample_seq_dev / nflhip_sample_gauss_seq_dev: one launch over a dense batch == per-polynomial calls with
    stream ids first + b*stride (what the header's deferred random constructors are coalesced into)."""
    from nfllib_amd import DIST_BOUNDED, DIST_HWT, DIST_UNIFORM, DIST_ZO
    e = engine_factory(lb, n, m)
    key, batch = bytes(range(32)), 7
    for dist, p0, p1 in ((DIST_UNIFORM, 0, 1), (DIST_BOUNDED, 9, 2), (DIST_ZO, 0x7F, 1), (DIST_HWT, max(1, n // 8), 1)):
        for first, stride in ((100, 1), (5, 3)):
            got = e.to_host(e.sample_seq(e.empty(batch), dist, key, first, stride, param0=p0, param1=p1))
            if dist in (DIST_UNIFORM, DIST_BOUNDED):   # ... and against the CPU statement of the rule on the same keystream
                from oracle import samplers as S
                for b in (0, batch - 1):
                    sid = first + b * stride
                    if dist == DIST_UNIFORM:
                        w = S.chacha20_words(key, sid, 0, m * n, counter_base=S.domain_base("uniform"))
                        w = (w & np.uint64((1 << lb) - 1)).astype(e.np_dtype).reshape(1, m, n)
                        want = S.uniform(w, [int(x) for x in e.P])
                    else:
                        w = S.chacha20_words(key, sid, 0, n, counter_base=S.domain_base("bounded")).reshape(1, n)
                        want = S.non_uniform(w, [int(x) for x in e.P], p0, p1, dtype=e.np_dtype)
                    assert np.array_equal(got[b:b + 1], want), (dist, first, stride, b)
            for b in range(batch):
                one = e.to_host(e.sample(e.empty(1), dist, key, stream_id=first + b * stride, param0=p0, param1=p1))
                assert np.array_equal(got[b:b + 1], one), (dist, first, stride, b)
    if lb == 16:
        return
    for sigma, sec in ((3.19, 128), (20.0, 64)):
        g = e.gauss_create(sigma, security=sec, samples=n)
        got = e.to_host(e.sample_gauss_seq(e.empty(batch), g, key, 40, 2, amplifier=2))
        for b in range(batch):
            one = e.to_host(e.sample_gauss(e.empty(1), g, key, stream_id=40 + 2 * b, amplifier=2))
            assert np.array_equal(got[b:b + 1], one), (sigma, b)
        e.gauss_destroy(g)


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2), (16, 128, 1), (64, 8, 3)])
def test_strided_expression_batches(lb, n, m, oracle_factory, engine_factory):
    """nflhip_eval_strided_dev: dense, shared (stride 0) and interleaved (stride 2) operands in one launch, against the
    dense evaluation of the gathered operands."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    k = 6
    u = e.to_device(o.fill_uniform(k, SEED, 0))
    key = e.to_device(o.fill_uniform(1, SEED + 1, 0))
    inter = e.to_device(o.fill_uniform(2 * k, SEED + 2, 1))          # e1_0, e2_0, e1_1, e2_1, ...
    out = e.empty(2 * k)
    out.zero_()
    prog = [0, 1, 0x12, 2, 0x10]                                       # u * key + e
    e.eval_strided(prog, [u, key, inter], [1, 0, 2], out, out_stride=2, batch=k)
    e.eval_strided(prog, [u, key, inter[1:]], [1, 0, 2], out[1:], out_stride=2, batch=k)
    dense_key = key.expand(k, -1, -1).contiguous()
    want1 = e.to_host(e.eval(prog, [u, dense_key, inter[0::2].contiguous()]))
    want2 = e.to_host(e.eval(prog, [u, dense_key, inter[1::2].contiguous()]))
    got = e.to_host(out)
    assert np.array_equal(got[0::2], want1) and np.array_equal(got[1::2], want2)
    # ... and against the CPU checker directly: u * key + e with the reference's mulmod / addmod (ops.hpp:124-135, 183-219)
    from nfllib_amd import OP_ADD, OP_MUL
    hu, hkey, hinter = e.to_host(u), e.to_host(dense_key), e.to_host(inter)
    for half, col in ((0, got[0::2]), (1, got[1::2])):
        assert np.array_equal(col, o.pointwise(OP_ADD, o.pointwise(OP_MUL, hu, hkey), np.ascontiguousarray(hinter[half::2])))
    # aliasing: in place on the interleaved operand
    e.eval_strided(prog, [u, key, inter], [1, 0, 2], inter, out_stride=2, batch=k)
    assert np.array_equal(e.to_host(inter)[0::2], want1)
