"""-m gpu: the HIP engine against the committed golden fixtures (outputs of the
REAL reference, tools/gen_golden.py) -- no oracle in the loop except as the
seeded input generator."""
import numpy as np
import pytest

import golden_util as G
from conftest import SEED

pytestmark = pytest.mark.gpu
INDEX, ARR = G.load()
OP = {"ADD": 0, "SUB": 1, "MUL": 2, "MUL_SHOUP": 3, "COMPUTE_SHOUP": 4}


class _Hip:
    """numpy-in/numpy-out adaptor over the device-pointer C ABI."""

    def __init__(self, e):
        self.e = e

    def ntt(self, a):
        return self.e.to_host(self.e.ntt_(self.e.to_device(a)))

    def intt(self, a):
        return self.e.to_host(self.e.intt_(self.e.to_device(a)))

    def pointwise(self, op, a, b=None, bp=None):
        d = self.e.to_device
        return self.e.to_host(self.e.pointwise(op, d(a), None if b is None else d(b), None if bp is None else d(bp)))

    def polymul(self, a, b):
        return self.e.to_host(self.e.polymul(self.e.to_device(a), self.e.to_device(b)))


@pytest.mark.parametrize("key", [k for k in INDEX["shapes"] if INDEX["shapes"][k]["degree"] >= 4])
def test_hip_matches_reference_fixtures(key, oracle_factory, engine_factory):
    ent = INDEX["shapes"][key]
    lb, n, m = ent["limb_bits"], ent["degree"], ent["nmoduli"]
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = o.fill_uniform(1, SEED, 0), o.fill_uniform(1, SEED, 1)
    assert G.sha(a) == ent["input_sha256"]["a"]
    out = G.compute_all(_Hip(e), a, b, OP)
    for k in G.OPS:
        assert G.sha(out[k]) == ent["sha256"][k], (key, k)
    if ent["mode"] == "full":
        h = _Hip(e)
        for ename, ev in G.edge_inputs(o.P, o.dtype, n, m).items():
            assert np.array_equal(h.ntt(ev), ARR["%s/edge_%s_ntt" % (key, ename)]), ename
            assert np.array_equal(h.intt(ev), ARR["%s/edge_%s_intt" % (key, ename)]), ename
    c = ent["crt"]
    lifted = e.crt_lift(e.to_device(a))
    hl = lifted.cpu().numpy().view(np.uint64)
    assert G.sha(hl) == c["lift_sha256"]
    assert np.array_equal(hl[0, ARR["%s/crt_idx" % key]], ARR["%s/crt_lift_sample" % key])
    import torch
    wide = G.wide_integers(n, c["project_wide_limbs"])
    dw = torch.from_numpy(wide.view(np.int64)).to(lifted.device)
    assert G.sha(e.to_host(e.crt_project(dw))) == c["project_sha256"]
