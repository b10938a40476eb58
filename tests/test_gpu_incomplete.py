"""-m gpu: the metric product on INCOMPLETE transforms (nflhip_polymul4096i{1,2}_asm, tools/asmgen/incomplete.py) through the
C ABI against the CPU oracle, bit-exact -- the same words as the complete kernel: nflhip_polymul_dev with
nflhip_debug_polymul_level 1 / 2 (the level the library ships with is whichever measured faster; all three must agree).
Semantics matched: a.ntt_pow_phi(); b.ntt_pow_phi(); c = a * b; c.invntt_pow_invphi() (poly.hpp:167-168, 350;
core.hpp:594-614)."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


@pytest.fixture()
def level():
    from nfllib_amd import _lib
    saved = _lib.lib.nflhip_debug_polymul_level(-1)

    def set_level(v):
        _lib.lib.nflhip_debug_polymul_level(v)
    yield set_level
    _lib.lib.nflhip_debug_polymul_level(saved)


@pytest.mark.parametrize("nm,batch", [(1, 1), (4, 6), (30, 3), (92, 2)])
def test_incomplete_products_equal_the_oracle(nm, batch, level, oracle_factory, engine_factory):
    o, e = oracle_factory(64, 4096, nm), engine_factory(64, 4096, nm)
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    want = o.polymul(a, b)
    da, db = e.to_device(a), e.to_device(b)
    for lv in (0, 1, 2):
        level(lv)
        assert np.array_equal(e.to_host(e.polymul(da, db)), want), "level %d" % lv


@pytest.mark.parametrize("lv", [1, 2])
def test_incomplete_products_at_the_boundaries(lv, level, oracle_factory, engine_factory):
    """all-(p-1) rows (the largest 128-bit sums), zeros, unit impulses at 0 and n-1 (negacyclic wrap through the base
    multiplication), alternating 0 / p-1, operands aliased with the result"""
    nm = 4
    o, e = oracle_factory(64, 4096, nm), engine_factory(64, 4096, nm)
    P = np.asarray(e.params.P[:nm], dtype=np.uint64)
    a = np.zeros((6, nm, 4096), dtype=np.uint64)
    b = np.zeros_like(a)
    a[0], b[0] = (P - 1)[:, None], (P - 1)[:, None]
    a[1, :, 4095], b[1, :, 4095] = 1, 1
    a[2, :, 0], b[2] = 1, o.fill_uniform(1, SEED, 1)[0]
    a[3, :, ::2], b[3, :, 1::2] = (P - 1)[:, None], (P - 1)[:, None]
    a[4], b[4] = o.fill_uniform(1, 7, 0)[0], 0
    a[5, :, 4095], b[5, :, 1] = (P - 1), (P - 1)
    want = o.polymul(a, b)
    level(lv)
    da, db = e.to_device(a), e.to_device(b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), want)
    assert np.array_equal(e.to_host(e.polymul(da, db, out=da)), want)      # c over a
    da = e.to_device(a)
    assert np.array_equal(e.to_host(e.polymul(da, db, out=db)), want)      # c over b
    sq = e.to_device(a)
    assert np.array_equal(e.to_host(e.polymul(sq, sq)), o.polymul(a, a))   # a * a (both operands one buffer)


@pytest.mark.parametrize("lv", [1, 2])
def test_incomplete_products_at_full_batch(lv, level, engine_factory, oracle_factory):
    """BASELINE configs[1] at its bench batch: digest equal to the complete kernel's, commutes, sampled polynomials vs the oracle"""
    e = engine_factory(64, 4096, 4)
    batch = 16384
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    level(0)
    ref = e.digest(e.polymul(a, b))
    level(lv)
    c = e.polymul(a, b)
    assert e.digest(c) == ref
    assert not e.any_neq(c, e.polymul(b, a))
    o = oracle_factory(64, 4096, 4)
    for i in (0, 5461, batch - 1):
        assert np.array_equal(e.to_host(c[i:i + 1]), o.polymul(e.to_host(a[i:i + 1]), e.to_host(b[i:i + 1])))


@pytest.mark.parametrize("n,m,batch,xcd", [(65536, 3, 11, False), (65536, 3, 11, True), (65536, 30, 2, False), (32768, 2, 37, True)])
def test_long_row_plans_on_incomplete_block_products(n, m, batch, xcd, level, oracle_factory):
    """rows of 65536 / 32768 words: the chunked three-role pipeline and the one-launch plan with their 4096-word block
    products on incomplete transforms (level 2; the streaming inverse role folds in (n / 4)^-1) -- the complete plans' words,
    the oracle's on the first and the last polynomial, in place"""
    from test_gpu_xcd import _engine
    o, e = oracle_factory(64, n, m), _engine(n, m, xcd)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    level(0)
    want = e.to_host(e.polymul(a, b))
    level(2)
    got = e.to_host(e.polymul(a, b))
    assert np.array_equal(got, want)
    for i in (0, batch - 1):
        assert np.array_equal(got[i:i + 1], o.polymul(e.to_host(a[i:i + 1]), e.to_host(b[i:i + 1])))
    a2 = a.clone()
    e.polymul(a2, b, out=a2)
    assert np.array_equal(e.to_host(a2), want)


@pytest.mark.parametrize("n,m,batch", [(8192, 2, 5), (16384, 8, 3), (16384, 1, 1)])
def test_row_resident_products_on_incomplete_transforms(n, m, batch, level, oracle_factory, engine_factory):
    """rows of 8192 / 16384 words (nflhip_polymul{8192,16384}i2_asm): the oracle's words at level 2 and 0, all-(p-1) rows, in place"""
    o, e = oracle_factory(64, n, m), engine_factory(64, n, m)
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    P = np.asarray(e.params.P[:m], dtype=np.uint64)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    want = o.polymul(a, b)
    for lv in (0, 2):
        level(lv)
        da, db = e.to_device(a), e.to_device(b)
        assert np.array_equal(e.to_host(e.polymul(da, db)), want), "level %d" % lv
        assert np.array_equal(e.to_host(e.polymul(da, db, out=da)), want), "level %d in place" % lv


@pytest.mark.parametrize("m,batch", [(2, 5), (1, 3)])
def test_register_resident_32768_word_rows_on_incomplete_transforms(m, batch, level, oracle_factory):
    """rows of 32768 words, the composed product (nflhip_ntt_fwd32768s[i2]_asm -> scratch -> nflhip_polymul_ntt32768s[i2]_asm): at level 2
    b' is stored two stages short and unreduced and meets a' in a base multiplication mod X^4 -+ zeta -- the oracle's words at level 2
    and 0, all-(p-1) rows, result over a and over b"""
    from test_gpu_xcd import _engine
    n = 32768
    o, e = oracle_factory(64, n, m), _engine(n, m, False)
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    P = np.asarray(e.params.P[:m], dtype=np.uint64)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    want = o.polymul(a, b)
    for lv in (0, 2, 0, 2):
        level(lv)
        da, db = e.to_device(a), e.to_device(b)
        assert np.array_equal(e.to_host(e.polymul(da, db)), want), "level %d" % lv
        assert np.array_equal(e.to_host(e.polymul(da, db, out=da)), want), "level %d, result over a" % lv
        da = e.to_device(a)
        assert np.array_equal(e.to_host(e.polymul(da, db, out=db)), want), "level %d, result over b" % lv


@pytest.mark.parametrize("n,m,batch", [(1024, 2, 5), (1024, 1, 1), (2048, 3, 3), (2048, 2, 64), (1024, 30, 7)])
def test_wave_per_row_u64_kernels(n, m, batch, level, oracle_factory, engine_factory, compiled_engine_factory):
    """64-bit rows of 1024 / 2048 words on the generated kernels (tools/asmgen/rows1k.py): product at level 2 and 0, forward, inverse,
    in place -- the oracle's words and the compiled template's (NFLHIP_VARIANT=hipcc contexts)"""
    o, e, ec = oracle_factory(64, n, m), engine_factory(64, n, m), compiled_engine_factory(64, n, m)
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    P = np.asarray(e.params.P[:m], dtype=np.uint64)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    want = o.polymul(a, b)
    for lv in (0, 2):
        level(lv)
        da, db = e.to_device(a), e.to_device(b)
        assert np.array_equal(e.to_host(e.polymul(da, db)), want), "level %d" % lv
        assert np.array_equal(e.to_host(e.polymul(da, db, out=da)), want), "level %d in place" % lv
    assert np.array_equal(ec.to_host(ec.polymul(ec.to_device(a), ec.to_device(b))), want)
    fa = e.to_host(e.ntt_(e.to_device(a)))
    assert np.array_equal(fa, o.ntt(a)) and np.array_equal(fa, ec.to_host(ec.ntt_(ec.to_device(a))))
    assert np.array_equal(e.to_host(e.intt_(e.to_device(fa))), a)


@pytest.mark.parametrize("n,m,batch", [(1024, 1, 5), (1024, 2, 3), (2048, 2, 5), (4096, 3, 3)])
def test_u32_products_on_incomplete_transforms(n, m, batch, level, oracle_factory, engine_factory):
    """32-bit limbs, rows of 1024 / 2048 / 4096 words (nflhip_row*_i2_u32_asm): the oracle's words at level 2 and 0, all-(p-1) rows, in place"""
    o, e = oracle_factory(32, n, m), engine_factory(32, n, m)
    a, b = o.fill_uniform(batch, SEED, 0), o.fill_uniform(batch, SEED, 1)
    P = np.asarray(e.params.P[:m], dtype=np.uint64).astype(a.dtype)
    a[batch - 1], b[batch - 1] = (P - 1)[:, None], (P - 1)[:, None]
    want = o.polymul(a, b)
    for lv in (0, 2):
        level(lv)
        da, db = e.to_device(a), e.to_device(b)
        assert np.array_equal(e.to_host(e.polymul(da, db)), want), "level %d" % lv
        assert np.array_equal(e.to_host(e.polymul(da, db, out=da)), want), "level %d in place" % lv
