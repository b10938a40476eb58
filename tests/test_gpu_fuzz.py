"""Shape / batch / operation fuzz of the C ABI against the oracle (derandomised hypothesis: the same examples every run).
The parametrised parity tests pin the shapes of BASELINE.json; this one walks the space between them -- every limb width,
degrees 4 .. 8192, 1 .. 32 moduli, ragged batches, random expression programs, aliasing of the destination."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

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


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
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


def test_the_walk_was_wide():
    """(runs after the fuzz above) every limb width and both ends of the degree range were visited"""
    assert len(_RAN) >= 100
    assert {lb for lb, _, _ in _RAN} == {16, 32, 64}
    assert min(l for _, l, _ in _RAN) <= 3 and max(l for _, l, _ in _RAN) >= 11 and max(m for _, _, m in _RAN) >= 16
