"""hipGraph capture of the C-ABI entry points ("capture launch-bound inner loops in hipGraphs"): every *_dev call
only enqueues work on the caller's stream, so a sequence of them can be captured once and replayed.  Covers the
single-launch kernels and the multi-launch plans (helper streams forked from / joined into the capturing stream at
n = 32768, the three-role pipeline at n = 65536, the composed plan for 16-bit limbs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lb,n,m,batch", [(64, 4096, 4, 8), (32, 1024, 1, 8), (64, 1024, 2, 8), (64, 8192, 2, 4),
                                          (64, 16384, 2, 4), (64, 32768, 2, 4), (64, 65536, 2, 4), (16, 128, 1, 8)])
def test_captured_sequence_replays_bit_identically(lb, n, m, batch, oracle_factory, engine_factory):
    import torch
    from nfllib_amd import OP_ADD
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a = e.fill_uniform(e.empty(batch), 7, 0)
    b = e.fill_uniform(e.empty(batch), 7, 1)
    c, d, f = e.empty(batch), e.empty(batch), e.empty(batch)
    ha, hb = e.to_host(a), e.to_host(b)
    want_c = o.polymul(ha, hb)
    want_d = o.pointwise(OP_ADD, want_c, ha)
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        e.polymul(a, b, out=c)  # warm-up outside the capture: scratch allocation, module load
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            e.polymul(a, b, out=c)
            e.pointwise(OP_ADD, c, a, out=d)
            f.copy_(d)
            e.ntt_(f)
            e.intt_(f)
    for _ in range(3):  # replays see fresh destinations every time
        c.zero_(); d.zero_(); f.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(e.to_host(c), want_c)
        assert np.array_equal(e.to_host(d), want_d)
        assert np.array_equal(e.to_host(f), want_d)
    # eager calls on the same context still work (and still order themselves on the scratch) after a capture
    assert np.array_equal(e.to_host(e.polymul(a, b)), want_c)
