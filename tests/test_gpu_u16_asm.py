"""The generated assembly product for 16-bit limbs, n = 128 (tools/gen_row128_u16_asm.py: the reference's
(128, 14, uint16_t) config, eight rows per wave) against the composed plan on the generic kernels (a context created under
NFLHIP_VARIANT=hipcc)
and against the oracle: one and two moduli, row counts that leave surplus lanes, boundary words."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,batch", [(1, 1), (1, 7), (1, 8), (1, 33), (2, 1), (2, 5), (2, 16), (2, 129), (1, 4099)])
def test_assembly_product_matches_generic_kernels_and_oracle(m, batch, oracle_factory, engine_factory, compiled_engine_factory):
    o, e, ec = oracle_factory(16, 128, m), engine_factory(16, 128, m), compiled_engine_factory(16, 128, m)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    ha, hb = e.to_host(a), e.to_host(b)
    P = np.asarray(o.P[:m], dtype=ha.dtype)
    ha[0, :, 0], ha[0, :, 1], ha[0, :, 2] = 0, 1, P - 1
    hb[0, :, 0], hb[0, :, 1], hb[0, :, 2] = P - 1, P - 1, P - 1
    ha[0, :, 127], hb[0, :, 127] = P - 1, P - 1
    a, b = e.to_device(ha), e.to_device(hb)
    want = ec.to_host(ec.polymul(a, b))
    want_f = ec.to_host(ec.ntt_(a.clone()))
    got = e.to_host(e.polymul(a, b))
    assert np.array_equal(got, want)
    # the stand-alone transforms (in place)
    fa = e.ntt_(a.clone())
    assert np.array_equal(e.to_host(fa), want_f)
    assert np.array_equal(want_f[:1], o.ntt(ha[:1]))
    assert np.array_equal(e.to_host(e.intt_(fa)), ha)
    assert np.array_equal(e.to_host(e.intt_(e.ntt_(b.clone()))), hb)
    # the product with b already transformed (keys kept in NTT form: tests/nfllib_demo_main_op.cpp:26-46)
    fb = e.ntt_(b.clone())
    assert np.array_equal(e.to_host(e.polymul(a, fb, b_is_ntt=True)), want)
    assert np.array_equal(ec.to_host(ec.polymul(a, fb, b_is_ntt=True)), want)
    k = min(batch, 9)
    assert np.array_equal(got[:k], o.polymul(ha[:k], hb[:k]))
    a2, b2 = a.clone(), b.clone()
    e.polymul(a2, b, out=a2)
    e.polymul(a, b2, out=b2)
    assert np.array_equal(e.to_host(a2), want) and np.array_equal(e.to_host(b2), want)
