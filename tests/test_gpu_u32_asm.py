"""The generated gfx950 assembly kernels of the 32-bit fused product (tools/gen_row1024_u32_asm.py: n = 1024 / 2048 /
4096, one / two / four waves per row) against the compiled kernels they replace (a context created under
NFLHIP_VARIANT=hipcc) and against the
oracle: row counts that leave surplus waves / rows in the last workgroup, several moduli per polynomial, boundary
words."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,batch", [(1024, 1, 7), (1024, 3, 5), (1024, 2, 1), (2048, 2, 5), (2048, 3, 3), (2048, 1, 1),
                                        (4096, 3, 3), (4096, 1, 2), (1024, 4, 257), (2048, 1, 129), (4096, 2, 65)])
def test_assembly_product_matches_compiled_kernel_and_oracle(n, m, batch, oracle_factory, engine_factory, compiled_engine_factory):
    o, e, ec = oracle_factory(32, n, m), engine_factory(32, n, m), compiled_engine_factory(32, n, m)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    ha, hb = e.to_host(a), e.to_host(b)
    # boundary words: 0, 1, p - 1 in the first polynomial
    P = np.asarray(o.P[:m], dtype=ha.dtype)
    ha[0, :, 0], ha[0, :, 1], ha[0, :, 2] = 0, 1, P - 1
    hb[0, :, 0], hb[0, :, 1], hb[0, :, 2] = P - 1, P - 1, P - 1
    ha[0, :, n - 1], hb[0, :, n - 1] = P - 1, P - 1
    a, b = e.to_device(ha), e.to_device(hb)
    want = ec.to_host(ec.polymul(a, b))
    want_f = ec.to_host(ec.ntt_(a.clone()))
    want_i = ec.to_host(ec.intt_(ec.ntt_(b.clone())))
    got = e.to_host(e.polymul(a, b))
    assert np.array_equal(got, want)
    # the stand-alone transforms (in place): forward = the compiled kernel's and the oracle's words, inverse undoes it
    fa = e.ntt_(a.clone())
    assert np.array_equal(e.to_host(fa), want_f)
    assert np.array_equal(want_f[:1], o.ntt(ha[:1]))
    assert np.array_equal(e.to_host(e.intt_(fa)), ha)
    assert np.array_equal(e.to_host(e.intt_(e.ntt_(b.clone()))), want_i) and np.array_equal(want_i, hb)
    k = min(batch, 3)
    assert np.array_equal(got[:k], o.polymul(ha[:k], hb[:k]))
    # in place on either operand, and commuted
    a2, b2 = a.clone(), b.clone()
    e.polymul(a2, b, out=a2)
    e.polymul(a, b2, out=b2)
    assert np.array_equal(e.to_host(a2), want) and np.array_equal(e.to_host(b2), want)
    assert np.array_equal(e.to_host(e.polymul(b, a)), want)


def test_assembly_product_every_modulus_of_the_table(oracle_factory, engine_factory):
    """all 32-bit moduli the context can hold at once, one row each"""
    from nfllib_amd.params import params
    m = min(64, params(32).max_moduli)
    o, e = oracle_factory(32, 1024, m), engine_factory(32, 1024, m)
    a = e.fill_uniform(e.empty(2), SEED + 3, 0)
    b = e.fill_uniform(e.empty(2), SEED + 3, 1)
    got = e.to_host(e.polymul(a, b))
    assert np.array_equal(got, o.polymul(e.to_host(a), e.to_host(b)))


@pytest.mark.parametrize("m,batch", [(1, 1), (2, 1), (2, 127), (1, 257), (2, 300), (3, 85), (2, 70001)])
def test_lane_per_row_product_n8(m, batch, oracle_factory, engine_factory, compiled_engine_factory):
    """n = 8 (the reference's (8, 60, uint32_t) config): one lane per row (tools/gen_row8_u32_asm.py)"""
    o, e, ec = oracle_factory(32, 8, m), engine_factory(32, 8, m), compiled_engine_factory(32, 8, m)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    ha, hb = e.to_host(a), e.to_host(b)
    P = np.asarray(o.P[:m], dtype=ha.dtype)
    ha[0, :, 0], ha[0, :, 1], ha[0, :, 2] = 0, 1, P - 1
    hb[0, :, 0], hb[0, :, 1], hb[0, :, 7] = P - 1, P - 1, P - 1
    a, b = e.to_device(ha), e.to_device(hb)
    want = ec.to_host(ec.polymul(a, b))
    got = e.to_host(e.polymul(a, b))
    assert np.array_equal(got, want)
    k = min(batch, 300)
    assert np.array_equal(got[:k], o.polymul(ha[:k], hb[:k]))
    # stand-alone transforms and the product with b already transformed: generated kernels of their own too
    fa, fb = e.ntt_(a.clone()), e.ntt_(b.clone())
    assert np.array_equal(e.to_host(fa), ec.to_host(ec.ntt_(a.clone())))
    assert np.array_equal(e.to_host(fa)[:k], o.ntt(ha[:k]))
    assert np.array_equal(e.to_host(e.intt_(fa.clone())), ha)
    assert np.array_equal(e.to_host(e.polymul(a, fb, b_is_ntt=True)), want)
    a2 = a.clone()
    e.polymul(a2, b, out=a2)
    assert np.array_equal(e.to_host(a2), want)
