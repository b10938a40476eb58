"""CPU: the oracle (oracle/nfl_oracle.c) against the golden fixtures that were
generated from the REAL reference (tools/gen_golden.py).  This is what pins the
oracle on hosts where /root/reference and oracle/_ref do not exist."""
import numpy as np
import pytest

import golden_util as G
from conftest import SEED

INDEX, ARR = G.load()
OP = {"ADD": 0, "SUB": 1, "MUL": 2, "MUL_SHOUP": 3, "COMPUTE_SHOUP": 4}
# E (65536 x 30) takes ~10 s on the CPU: keep it, it is BASELINE configs[4]
KEYS = list(INDEX["shapes"].keys())


@pytest.mark.parametrize("key", KEYS)
def test_oracle_matches_reference_outputs(key, oracle_factory):
    ent = INDEX["shapes"][key]
    lb, n, m = ent["limb_bits"], ent["degree"], ent["nmoduli"]
    o = oracle_factory(lb, n, m)
    a, b = o.fill_uniform(1, SEED, 0), o.fill_uniform(1, SEED, 1)
    assert G.sha(a) == ent["input_sha256"]["a"] and G.sha(b) == ent["input_sha256"]["b"], "input generator drifted"
    out = G.compute_all(o, a, b, OP)
    for k in G.OPS:
        assert G.sha(out[k]) == ent["sha256"][k], (key, k)
        if ent["mode"] == "full":
            assert np.array_equal(out[k], ARR["%s/%s" % (key, k)]), (key, k)
        else:
            idx = ARR["%s/sample_idx" % key]
            assert np.array_equal(out[k].reshape(-1)[idx], ARR["%s/sample_%s" % (key, k)]), (key, k)


@pytest.mark.parametrize("key", G.shape_keys(INDEX, mode="full"))
def test_oracle_tables_and_edges(key, oracle_factory):
    from oracle import oracle as O
    ent = INDEX["shapes"][key]
    lb, n, m = ent["limb_bits"], ent["degree"], ent["nmoduli"]
    o = oracle_factory(lb, n, m)
    for which, name in ((O.TAB_PHIS, "phis"), (O.TAB_SHOUPPHIS, "shoupphis"), (O.TAB_INVPOLY_INVPHIS, "invpoly_times_invphis"),
                        (O.TAB_OMEGAS, "omegas"), (O.TAB_INVOMEGAS, "invomegas")):
        assert np.array_equal(o.table(which, 0), ARR["%s/table_%s" % (key, name)]), name
    for ename, ev in G.edge_inputs(o.P, o.dtype, n, m).items():
        assert np.array_equal(o.ntt(ev), ARR["%s/edge_%s_ntt" % (key, ename)]), ename
        assert np.array_equal(o.intt(ev), ARR["%s/edge_%s_intt" % (key, ename)]), ename


@pytest.mark.parametrize("key", KEYS)
def test_oracle_crt_matches_reference(key, oracle_factory):
    import hashlib
    ent = INDEX["shapes"][key]
    lb, n, m = ent["limb_bits"], ent["degree"], ent["nmoduli"]
    o = oracle_factory(lb, n, m)
    c = ent["crt"]
    assert o.crt_bits == c["bits"] and o.crt_shift == c["shift"] and o.crt_limbs == c["limbs"]
    assert hex(o.crt_modulus()) == c["modulus"] and hex(o.crt_modulus_shoup()) == c["modulus_shoup"]
    lh = hashlib.sha256(b"".join(o.crt_lifting(cm).to_bytes(8 * (40 if m <= 32 else 100), "little") for cm in range(m))).hexdigest()
    assert lh == c["lifting_sha256"]
    a = o.fill_uniform(1, SEED, 0)
    lifted = o.crt_lift(a)
    assert G.sha(lifted) == c["lift_sha256"]
    assert np.array_equal(lifted[0, ARR["%s/crt_idx" % key]], ARR["%s/crt_lift_sample" % key])
    assert np.array_equal(o.crt_project(lifted), a)  # tests/poly_mpz.cpp:19-29
    assert G.sha(o.crt_project(G.wide_integers(n, c["project_wide_limbs"]))) == c["project_sha256"]


def test_polymul_is_negacyclic_convolution(oracle_factory):
    """Second, independent oracle: schoolbook mod (X^n+1, p)."""
    for lb, n, m in [(64, 64, 3), (32, 8, 2), (16, 128, 1)]:
        o = oracle_factory(lb, n, m)
        a, b = o.fill_uniform(1, 99, 0), o.fill_uniform(1, 99, 1)
        got = o.polymul(a, b)
        for cm in range(m):
            p = o.P[cm]
            x, y = [int(v) for v in a[0, cm]], [int(v) for v in b[0, cm]]
            z = [0] * n
            for i in range(n):
                for j in range(n):
                    if i + j < n:
                        z[i + j] = (z[i + j] + x[i] * y[j]) % p
                    else:
                        z[i + j - n] = (z[i + j - n] - x[i] * y[j]) % p
            assert [int(v) for v in got[0, cm]] == z


def test_reference_eq_quirk(oracle_factory):
    """`a == b` is "any lane equal", `a != b` is "any lane differs" (ops.hpp:81-117)."""
    o = oracle_factory(64, 64, 3)
    a, b = o.fill_uniform(1, 5, 0), o.fill_uniform(1, 5, 1)
    b[b == a] += 1
    assert not o.any_eq(a, b) and o.any_neq(a, b)
    b[0, 2, 63] = a[0, 2, 63]
    assert o.any_eq(a, b) and o.any_neq(a, b)
    assert o.any_eq(a, a) and not o.any_neq(a, a)


def test_degree2_quirk_is_restated(oracle_factory):
    """core::ntt returns before the strict reduction for degree 2 (core.hpp:469-481)."""
    o = oracle_factory(64, 2, 1)
    p = o.P[0]
    row = np.array([p - 1, p - 1], dtype=np.uint64)
    out = o.ntt_row(row, 0)
    assert int(out[0]) == 2 * p - 2 and int(out[1]) == 0
