"""Rows of 32768 / 65536 words in ONE launch of persistent workgroups (csrc/kernels_fast.hip launch_polymul_xcd_u64,
tools/gen_polymul_asm.py fused_header): every row's three roles run on one XCD, handed out by per-domain credit /
ticket counters.  Checked here: the words are those of the chunked pipeline (which test_gpu_parity.py holds against
the oracle) and of the oracle itself on a sample row, for batches whose row counts are not multiples of anything, for
every ring / domain setting, in place, and on two streams; and that the plan under test is the one that ran."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

KNOBS = ("NFLHIP_XCD", "NFLHIP_XCD_RLOG", "NFLHIP_XCD_DLOG", "NFLHIP_XCD_WGS", "NFLHIP_XCD_MAX_ROWS", "NFLHIP_XCD_POOL")


@pytest.fixture(autouse=True)
def _restore_env():
    saved = {k: os.environ.get(k) for k in KNOBS}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _launches(e):
    e.lib.nflhip_debug_xcd_launches.restype = C.c_ulonglong
    return int(e.lib.nflhip_debug_xcd_launches())


def _product(e, a, b, xcd, **knobs):
    os.environ["NFLHIP_XCD"] = "1" if xcd else "0"
    for k in ("NFLHIP_XCD_RLOG", "NFLHIP_XCD_DLOG", "NFLHIP_XCD_WGS", "NFLHIP_XCD_POOL"):
        os.environ.pop(k, None)
    for k, v in knobs.items():
        os.environ["NFLHIP_XCD_" + k.upper()] = str(v)
    before = _launches(e)
    c = e.polymul(a, b)
    import torch
    torch.cuda.synchronize()
    assert (_launches(e) - before == 1) == bool(xcd), "the plan under test did not run"
    return c


@pytest.mark.parametrize("n,m,batch", [(32768, 2, 37), (32768, 2, 64), (32768, 1, 33), (65536, 3, 11), (65536, 30, 2),
                                        (65536, 2, 16), (65536, 5, 13)])
def test_one_launch_plan_matches_pipeline_and_oracle(n, m, batch, oracle_factory, engine_factory):
    o, e = oracle_factory(64, n, m), engine_factory(64, n, m)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    want = e.to_host(_product(e, a, b, xcd=False))
    got = e.to_host(_product(e, a, b, xcd=True))
    assert np.array_equal(got, want)
    ha, hb = e.to_host(a[batch - 1:batch]), e.to_host(b[batch - 1:batch])
    assert np.array_equal(got[batch - 1:batch], o.polymul(ha, hb)), "last polynomial differs from the oracle"
    # commutes, and works in place on either operand
    assert np.array_equal(e.to_host(_product(e, b, a, xcd=True)), want)
    os.environ["NFLHIP_XCD"] = "1"
    a2, b2 = a.clone(), b.clone()
    e.polymul(a2, b, out=a2)
    e.polymul(a, b2, out=b2)
    assert np.array_equal(e.to_host(a2), want) and np.array_equal(e.to_host(b2), want)


@pytest.mark.parametrize("rlog,dlog,wgs", [(1, 0, 256), (1, 3, 768), (2, 1, 512), (3, 2, 768), (4, 0, 768), (5, 3, 1024),
                                            (3, 2, 300)])
def test_every_ring_and_domain_setting(rlog, dlog, wgs, engine_factory):
    """rows in flight per domain (2^rlog), scheduling domains per XCD (2^dlog) and grid size only change the schedule"""
    for n, m, batch in ((32768, 2, 41), (65536, 3, 23)):
        e = engine_factory(64, n, m)
        a = e.fill_uniform(e.empty(batch), SEED + 1, 0)
        b = e.fill_uniform(e.empty(batch), SEED + 1, 1)
        want = e.to_host(_product(e, a, b, xcd=False))
        assert np.array_equal(e.to_host(_product(e, a, b, xcd=True, rlog=rlog, dlog=dlog, wgs=wgs)), want)


@pytest.mark.parametrize("rlog,dlog", [(1, 2), (2, 1), (1, 0), (3, 0)])
def test_pooled_scratch_variant(rlog, dlog, engine_factory):
    """NFLHIP_XCD_POOL=1 (experiment kept for its measurements): scratch rows come from a per-XCD pool of 32
    slots that are reused within the launch, consumers read them with `nt` loads -- same words"""
    for n, m, batch in ((32768, 2, 45), (65536, 3, 19), (65536, 30, 4)):
        e = engine_factory(64, n, m)
        a = e.fill_uniform(e.empty(batch), SEED + 2, 0)
        b = e.fill_uniform(e.empty(batch), SEED + 2, 1)
        want = e.to_host(_product(e, a, b, xcd=False))
        for rep in range(3):   # (slot reuse across launches too)
            assert np.array_equal(e.to_host(_product(e, a, b, xcd=True, pool=1, rlog=rlog, dlog=dlog)), want)


def test_small_batches_fall_back(engine_factory):
    """fewer rows than 8 per domain set: the chunked plans serve the call (same words, no one-launch kernel)"""
    e = engine_factory(64, 65536, 2)
    a = e.fill_uniform(e.empty(3), SEED, 0)
    b = e.fill_uniform(e.empty(3), SEED, 1)
    os.environ["NFLHIP_XCD"] = "0"
    want = e.to_host(e.polymul(a, b))
    os.environ["NFLHIP_XCD"] = "1"
    before = _launches(e)
    assert np.array_equal(e.to_host(e.polymul(a, b)), want)
    assert _launches(e) == before


def test_default_policy_by_rows(engine_factory):
    """unset NFLHIP_XCD: batches of at most NFLHIP_XCD_MAX_ROWS rows (default 1024 at n = 32768, 256 at 65536) take the
    one-launch plan, larger ones the pipeline"""
    os.environ.pop("NFLHIP_XCD", None)
    os.environ.pop("NFLHIP_XCD_MAX_ROWS", None)
    e = engine_factory(64, 32768, 2)
    for batch, expect in ((64, 1), (513, 0)):
        a = e.fill_uniform(e.empty(batch), SEED, 0)
        b = e.fill_uniform(e.empty(batch), SEED, 1)
        before = _launches(e)
        c = e.polymul(a, b)
        assert _launches(e) - before == expect, batch
        os.environ["NFLHIP_XCD"] = "0" if expect else "1"
        assert not e.any_neq(e.polymul(a, b), c)
        os.environ.pop("NFLHIP_XCD", None)


def test_two_streams_share_the_scratch(engine_factory):
    """successive one-launch products on different streams reuse the context's scratch: ordered by the library's events"""
    import torch
    e = engine_factory(64, 32768, 2)
    batch = 48
    os.environ["NFLHIP_XCD"] = "0"
    xs = [(e.fill_uniform(e.empty(batch), SEED + k, 0), e.fill_uniform(e.empty(batch), SEED + k, 1)) for k in range(4)]
    wants = [e.to_host(e.polymul(a, b)) for a, b in xs]
    torch.cuda.synchronize()
    os.environ["NFLHIP_XCD"] = "1"
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for rep in range(3):
        for k, (a, b) in enumerate(xs):
            st = streams[k & 1]
            with torch.cuda.stream(st):
                outs.append((k, e.polymul(a, b, stream=st)))
    torch.cuda.synchronize()
    for k, c in outs:
        assert np.array_equal(e.to_host(c), wants[k]), k


def test_graph_capture_of_one_launch_plan(engine_factory):
    """memset + persistent kernel are plain stream work: a captured graph replays the product"""
    import torch
    e = engine_factory(64, 32768, 2)
    batch = 40
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    os.environ["NFLHIP_XCD"] = "0"
    want = e.to_host(e.polymul(a, b))
    os.environ["NFLHIP_XCD"] = "1"
    c = e.empty(batch)
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    before = _launches(e)
    with torch.cuda.stream(st):
        e.polymul(a, b, out=c)   # warm-up outside the capture: scratch allocation, module load
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            e.polymul(a, b, out=c)
    assert _launches(e) - before == 2
    for _ in range(3):
        c.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(e.to_host(c), want)


def test_two_persistent_kernels_at_once(engine_factory):
    """two contexts launch their one-launch plans on two streams: 2 x 768 persistent workgroups cannot all be resident;
    nothing may depend on workgroups that are not (each kernel's roles are served by whichever of ITS workgroups run)"""
    import torch
    e1, e2 = engine_factory(64, 65536, 3), engine_factory(64, 32768, 2)
    a1, b1 = e1.fill_uniform(e1.empty(24), SEED, 0), e1.fill_uniform(e1.empty(24), SEED, 1)
    a2, b2 = e2.fill_uniform(e2.empty(96), SEED, 2), e2.fill_uniform(e2.empty(96), SEED, 3)
    os.environ["NFLHIP_XCD"] = "0"
    w1, w2 = e1.to_host(e1.polymul(a1, b1)), e2.to_host(e2.polymul(a2, b2))
    os.environ["NFLHIP_XCD"] = "1"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(6):
        with torch.cuda.stream(s1):
            c1 = e1.polymul(a1, b1, stream=s1)
        with torch.cuda.stream(s2):
            c2 = e2.polymul(a2, b2, stream=s2)
        outs.append((c1, c2))
    torch.cuda.synchronize()
    for c1, c2 in outs:
        assert np.array_equal(e1.to_host(c1), w1) and np.array_equal(e2.to_host(c2), w2)
