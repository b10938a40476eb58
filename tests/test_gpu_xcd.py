"""Rows of 32768 / 65536 words in ONE launch of persistent workgroups (csrc/kernels_fast.hip launch_polymul_xcd_u64,
tools/gen_polymul_asm.py fused_header): every row's three roles run on one XCD, handed out by per-domain credit /
ticket counters.  Checked here: the words are those of the chunked pipeline (which test_gpu_parity.py holds against
the oracle) and of the oracle itself on a sample row, for batches whose row counts are not multiples of anything, in
place, and on two streams; and that the plan under test is the one that ran.  The plan is pinned per CONTEXT
(NFLHIP_XCD read once when it is created)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

_ENGINES = {}


def _engine(n, m, xcd):
    """a context whose plan for the long rows is pinned when it is created: NFLHIP_XCD=1 the one-launch plan, 0 the other
    plans (pipeline at 65536, register-resident rows at 32768), None = the default policy by batch size"""
    from nfllib_amd import Engine
    key = (n, m, xcd)
    if key not in _ENGINES:
        saved = os.environ.get("NFLHIP_XCD")
        if xcd is None:
            os.environ.pop("NFLHIP_XCD", None)
        else:
            os.environ["NFLHIP_XCD"] = "1" if xcd else "0"
        try:
            _ENGINES[key] = Engine(64, n, m)
        finally:
            if saved is None:
                os.environ.pop("NFLHIP_XCD", None)
            else:
                os.environ["NFLHIP_XCD"] = saved
    return _ENGINES[key]


@pytest.fixture(scope="module", autouse=True)
def _close_engines():
    yield
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def _launches(e):
    e.lib.nflhip_debug_xcd_launches.restype = C.c_ulonglong
    return int(e.lib.nflhip_debug_xcd_launches())


def _product(n, m, a, b, xcd, **kw):
    e = _engine(n, m, xcd)
    before = _launches(e)
    c = e.polymul(a, b, **kw)
    import torch
    torch.cuda.synchronize()
    assert (_launches(e) - before == 1) == bool(xcd), "the plan under test did not run"
    return c


@pytest.mark.parametrize("n,m,batch", [(32768, 2, 37), (32768, 2, 64), (32768, 1, 33), (65536, 3, 11), (65536, 30, 2),
                                        (65536, 2, 16), (65536, 5, 13), (32768, 2, 300)])
def test_one_launch_plan_matches_the_other_plans_and_the_oracle(n, m, batch, oracle_factory):
    o, e = oracle_factory(64, n, m), _engine(n, m, True)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    want = e.to_host(_product(n, m, a, b, xcd=False))
    got = e.to_host(_product(n, m, a, b, xcd=True))
    assert np.array_equal(got, want)
    ha, hb = e.to_host(a[batch - 1:batch]), e.to_host(b[batch - 1:batch])
    assert np.array_equal(got[batch - 1:batch], o.polymul(ha, hb)), "last polynomial differs from the oracle"
    # commutes, and works in place on either operand
    assert np.array_equal(e.to_host(_product(n, m, b, a, xcd=True)), want)
    a2, b2 = a.clone(), b.clone()
    e.polymul(a2, b, out=a2)
    e.polymul(a, b2, out=b2)
    assert np.array_equal(e.to_host(a2), want) and np.array_equal(e.to_host(b2), want)


def test_small_batches_fall_back():
    """fewer rows than 8 per domain set: the chunked plans serve the call (same words, no one-launch kernel)"""
    e0, e1 = _engine(65536, 2, False), _engine(65536, 2, True)
    a = e0.fill_uniform(e0.empty(3), SEED, 0)
    b = e0.fill_uniform(e0.empty(3), SEED, 1)
    want = e0.to_host(e0.polymul(a, b))
    before = _launches(e1)
    assert np.array_equal(e1.to_host(e1.polymul(a, b)), want)
    assert _launches(e1) == before


def test_default_policy_by_rows():
    """NFLHIP_XCD unset when the context is created: batches of fewer than 256 rows (n = 32768) / at most 256 rows
    (n = 65536) take the one-launch plan, larger ones the register-resident rows / the pipeline"""
    for n, m, cases in ((32768, 2, ((64, 1), (127, 1), (128, 0), (513, 0))), (65536, 2, ((128, 1), (129, 0)))):
        e = _engine(n, m, None)
        for batch, expect in cases:
            a = e.fill_uniform(e.empty(batch), SEED, 0)
            b = e.fill_uniform(e.empty(batch), SEED, 1)
            before = _launches(e)
            c = e.polymul(a, b)
            assert _launches(e) - before == expect, (n, batch)
            other = _engine(n, m, not expect)
            assert not e.any_neq(other.polymul(a, b), c)


def test_two_streams_share_the_scratch():
    """successive one-launch products on different streams reuse the context's scratch: ordered by the library's events"""
    import torch
    e0, e = _engine(32768, 2, False), _engine(32768, 2, True)
    batch = 48
    xs = [(e.fill_uniform(e.empty(batch), SEED + k, 0), e.fill_uniform(e.empty(batch), SEED + k, 1)) for k in range(4)]
    wants = [e0.to_host(e0.polymul(a, b)) for a, b in xs]
    torch.cuda.synchronize()
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


def test_graph_capture_of_one_launch_plan():
    """memset + persistent kernel are plain stream work: a captured graph replays the product"""
    import torch
    e0, e = _engine(32768, 2, False), _engine(32768, 2, True)
    batch = 40
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    want = e0.to_host(e0.polymul(a, b))
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


def test_two_persistent_kernels_at_once():
    """two contexts launch their one-launch plans on two streams: 2 x 768 persistent workgroups cannot all be resident;
    nothing may depend on workgroups that are not (each kernel's roles are served by whichever of ITS workgroups run)"""
    import torch
    e1, e2 = _engine(65536, 3, True), _engine(32768, 2, True)
    a1, b1 = e1.fill_uniform(e1.empty(24), SEED, 0), e1.fill_uniform(e1.empty(24), SEED, 1)
    a2, b2 = e2.fill_uniform(e2.empty(96), SEED, 2), e2.fill_uniform(e2.empty(96), SEED, 3)
    f1, f2 = _engine(65536, 3, False), _engine(32768, 2, False)
    w1, w2 = f1.to_host(f1.polymul(a1, b1)), f2.to_host(f2.polymul(a2, b2))
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
