"""-m gpu: BASELINE.json configs[3] at its REAL per-GPU size.

configs[3] is nfl::poly<uint64_t,4096,4> with a batch of 2^20 polynomials split over 8 GPUs: every
rank owns 2^17 polynomials = 16 GiB per operand, and rank r generates its shard in place from the
shared counter stream at first_poly = r * 2^17 (no data-path collective).  This test
runs the LAST rank's shard (first_poly = 7 * 2^17) on one GPU: word offsets past 2^31, byte offsets
past 2^34, whole-shard size-independent properties of the reference sequence
a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b; c.invntt_pow_invphi() (poly.hpp:167-168, 350), and sampled
polynomials regenerated on the CPU oracle from the same counter stream.
"""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

LB, N, NM = 64, 4096, 4
SHARD = 1 << 17
FIRST = 7 * SHARD


def test_configs3_last_rank_shard(oracle_factory, engine_factory):
    import torch
    from nfllib_amd import OP_ADD
    free, _total = torch.cuda.mem_get_info(0)
    need = 6 * SHARD * NM * N * 8
    if free < need + (8 << 30):
        pytest.fail("configs[3] needs %d GiB of HBM, only %d GiB free" % (need >> 30, free >> 30))
    o, e = oracle_factory(LB, N, NM), engine_factory(LB, N, NM)
    a = e.fill_uniform(e.empty(SHARD), SEED, 0, first_poly=FIRST)
    b = e.fill_uniform(e.empty(SHARD), SEED, 1, first_poly=FIRST)
    assert a.numel() == 1 << 31 and a.numel() * 8 == 16 << 30
    c = e.polymul(a, b)
    t = e.polymul(b, a)
    assert not e.any_neq(c, t), "commutativity over the whole shard"
    # the b-pre-transformed kernel agrees word for word with the fused one
    t.copy_(b)
    e.ntt_(t)
    u = e.polymul(a, t, b_is_ntt=True)
    assert not e.any_neq(c, u)
    # INTT(NTT(b)) == b over the whole shard (t holds NTT(b))
    e.intt_(t)
    assert not e.any_neq(t, b), "round trip over the whole shard"
    # linearity: (a+b)*b == a*b + b*b
    e.pointwise(OP_ADD, a, b, out=t)
    lhs = e.polymul(t, b, out=t)
    e.polymul(b, b, out=u)
    rhs = e.pointwise(OP_ADD, c, u, out=u)
    assert not e.any_neq(lhs, rhs), "linearity over the whole shard"
    # any_neq must really see the far end of a 16 GiB operand
    u.copy_(c)
    u[SHARD - 1, NM - 1, N - 1] += 1
    assert e.any_neq(c, u)
    del t, u, lhs, rhs
    # sampled polynomials, incl. both ends of the shard (the last one starts at word 2^31 - 16384)
    rng = np.random.default_rng(3)
    picks = [0, 1, SHARD // 2 - 1, SHARD // 2, SHARD - 2, SHARD - 1] + [int(x) for x in rng.choice(SHARD, size=12, replace=False)]
    for idx in picks:
        ha = o.fill_uniform(1, SEED, 0, first_poly=FIRST + idx)
        hb = o.fill_uniform(1, SEED, 1, first_poly=FIRST + idx)
        assert np.array_equal(e.to_host(a[idx:idx + 1]), ha), "operand a, poly %d" % idx
        assert np.array_equal(e.to_host(b[idx:idx + 1]), hb), "operand b, poly %d" % idx
        assert np.array_equal(e.to_host(c[idx:idx + 1]), o.polymul(ha, hb)), "product, poly %d" % idx
    # the shard checksum a rank would contribute to the checksum-of-checksums is shard-position dependent
    a0 = e.fill_uniform(e.empty(4), SEED, 0, first_poly=0)
    assert e.any_neq(a0, a[:4])
