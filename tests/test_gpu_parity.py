"""-m gpu: the HIP engine (through the C ABI) against the CPU oracle, bit-exact.

Mirrors what the reference's own tests pin -- scalar-formula agreement of + - *
(tests/test_binary_op.h:9-32), shoup-mul (tests/nfllib_demo_main_op.cpp:76-84),
NTT/INTT (tests/poly_p.cpp:51-58), CRT round trip (tests/poly_mpz.cpp:19-64) --
but with memcmp equality instead of the reference's "any lane" operator==
(ops.hpp:81-95), over the reference's test configs (tests/CMakeLists.txt:19-48)
and the BASELINE.json shapes.
"""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

# (limb_bits, degree, nmoduli, batch)
SHAPES = [
    (16, 128, 1, 5),      # reference CONFIG (128,14,uint16_t)
    (16, 512, 2, 2),
    (32, 8, 2, 9),        # reference CONFIG (8,60,uint32_t)
    (32, 1024, 1, 3),     # BASELINE configs[0]
    (32, 1024, 2, 3),     # reference CONFIG (1024,60,uint32_t)
    (32, 4096, 3, 3),     # 30-bit moduli, one workgroup per row
    (32, 8192, 2, 3),     # 30-bit moduli, streaming pass + 4096-word blocks
    (32, 32768, 1, 1),    # u32 kMaxPolyDegree
    (64, 4, 1, 7),
    (64, 8, 2, 5),
    (64, 64, 3, 4),
    (64, 1024, 2, 3),     # tests/ntt_perfs.cpp shape
    (64, 2048, 1, 2),
    (64, 2048, 3, 3),     # two-wave rows, odd row count (a surplus row in the last workgroup)
    (32, 2048, 2, 5),
    (64, 4096, 4, 6),     # BASELINE configs[1] -- the metric shape
    (64, 8192, 2, 2),     # reference CONFIG (8192,124,uint64_t)
    (64, 16384, 8, 2),    # BASELINE configs[2]
    (64, 32768, 2, 1),    # reference CONFIG (32768,124,uint64_t)
    (64, 65536, 30, 1),   # BASELINE configs[4]
]
IDS = ["u%d-n%d-m%d" % s[:3] for s in SHAPES]


def _inputs(o, batch, seed=SEED):
    return o.fill_uniform(batch, seed, 0), o.fill_uniform(batch, seed, 1)


@pytest.mark.parametrize("lb,n,m,batch", SHAPES, ids=IDS)
def test_fill_uniform_matches_oracle(lb, n, m, batch, oracle_factory, engine_factory):
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    for operand in (0, 1):
        d = e.fill_uniform(e.empty(batch), SEED, operand, first_poly=3)
        assert np.array_equal(e.to_host(d), o.fill_uniform(batch, SEED, operand, first_poly=3))


@pytest.mark.parametrize("lb,n,m,batch", SHAPES, ids=IDS)
def test_ntt_forward_inverse(lb, n, m, batch, oracle_factory, engine_factory):
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, _ = _inputs(o, batch)
    fa = e.to_host(e.ntt_(e.to_device(a)))
    assert np.array_equal(fa, o.ntt(a)), "ntt_pow_phi differs from the reference algorithm"
    ia = e.to_host(e.intt_(e.to_device(a)))
    assert np.array_equal(ia, o.intt(a)), "invntt_pow_invphi differs from the reference algorithm"
    rt = e.to_host(e.intt_(e.ntt_(e.to_device(a))))
    assert np.array_equal(rt, a), "INTT(NTT(a)) != a"


@pytest.mark.parametrize("lb,n,m,batch", SHAPES, ids=IDS)
def test_pointwise_ops(lb, n, m, batch, oracle_factory, engine_factory):
    from nfllib_amd import OP_ADD, OP_COMPUTE_SHOUP, OP_MUL, OP_MUL_SHOUP, OP_SUB
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, batch)
    da, db = e.to_device(a), e.to_device(b)
    for op in (OP_ADD, OP_SUB, OP_MUL):
        assert np.array_equal(e.to_host(e.pointwise(op, da, db)), o.pointwise(op, a, b)), op
    bp = o.pointwise(OP_COMPUTE_SHOUP, b)
    dbp = e.pointwise(OP_COMPUTE_SHOUP, db)
    assert np.array_equal(e.to_host(dbp), bp)
    assert np.array_equal(e.to_host(e.pointwise(OP_MUL_SHOUP, da, db, dbp)), o.pointwise(OP_MUL_SHOUP, a, b, bp))
    # aliasing: a = a + b is legal in the reference (core.hpp:24-37)
    dc = da.clone()
    e.pointwise(OP_ADD, dc, db, out=dc)
    assert np.array_equal(e.to_host(dc), o.pointwise(OP_ADD, a, b))


@pytest.mark.parametrize("lb,n,m,batch", SHAPES, ids=IDS)
def test_polymul(lb, n, m, batch, oracle_factory, engine_factory):
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, batch)
    want = o.polymul(a, b)
    da, db = e.to_device(a), e.to_device(b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), want)
    # inputs untouched
    assert np.array_equal(e.to_host(da), a) and np.array_equal(e.to_host(db), b)
    # one operand pre-transformed
    dbn = e.ntt_(db.clone())
    assert np.array_equal(e.to_host(e.polymul(da, dbn, b_is_ntt=True)), want)
    # c aliasing a
    dc = da.clone()
    e.polymul(dc, db, out=dc)
    assert np.array_equal(e.to_host(dc), want)


def test_polymul_is_negacyclic_convolution(oracle_factory, engine_factory):
    """Independent second oracle: schoolbook product mod (X^n + 1, p)."""
    for lb, n, m in [(64, 64, 3), (32, 8, 2), (16, 128, 1)]:
        o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
        a, b = _inputs(o, 2, seed=7)
        got = e.to_host(e.polymul(e.to_device(a), e.to_device(b)))
        for bi in range(2):
            for cm in range(m):
                p = o.P[cm]
                x = [int(v) for v in a[bi, cm]]
                y = [int(v) for v in b[bi, cm]]
                z = [0] * n
                for i in range(n):
                    for j in range(n):
                        k = i + j
                        if k < n:
                            z[k] = (z[k] + x[i] * y[j]) % p
                        else:
                            z[k - n] = (z[k - n] - x[i] * y[j]) % p
                assert [int(v) for v in got[bi, cm]] == z


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2), (16, 128, 1)])
def test_edge_vectors(lb, n, m, oracle_factory, engine_factory):
    """all-zero, all-(p-1), unit impulses, X^(n-1)*X wrap (negacyclic sign)."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    dt = o.dtype
    zero = np.zeros((1, m, n), dtype=dt)
    pm1 = np.stack([np.full(n, o.P[cm] - 1, dtype=dt) for cm in range(m)])[None]
    imp0 = zero.copy(); imp0[0, :, 0] = 1
    impl = zero.copy(); impl[0, :, n - 1] = 1
    x1 = zero.copy(); x1[0, :, 1] = 1
    for v in (zero, pm1, imp0, impl, x1):
        assert np.array_equal(e.to_host(e.ntt_(e.to_device(v))), o.ntt(v))
        assert np.array_equal(e.to_host(e.intt_(e.to_device(v))), o.intt(v))
    # X^(n-1) * X = X^n = -1
    got = e.to_host(e.polymul(e.to_device(impl), e.to_device(x1)))
    want = zero.copy()
    for cm in range(m):
        want[0, cm, 0] = o.P[cm] - 1
    assert np.array_equal(got, want)
    assert np.array_equal(e.to_host(e.polymul(e.to_device(pm1), e.to_device(pm1))), o.polymul(pm1, pm1))


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2)])
def test_any_eq_neq_reference_semantics(lb, n, m, oracle_factory, engine_factory):
    """`a == b` in the reference is "ANY lane equal" (ops.hpp:81-107)."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, 2)
    b[b == a] += 1  # make every lane differ
    da, db = e.to_device(a), e.to_device(b)
    assert e.any_eq(da, da) and not e.any_neq(da, da)
    assert not e.any_eq(da, db) and e.any_neq(da, db)
    b2 = b.copy(); b2[1, m - 1, n - 1] = a[1, m - 1, n - 1]
    db2 = e.to_device(b2)
    assert e.any_eq(da, db2) and e.any_neq(da, db2)
    assert e.any_eq(da, db2) == o.any_eq(a, b2) and e.any_neq(da, db2) == o.any_neq(a, b2)


@pytest.mark.parametrize("lb,n,m,batch", [(64, 4096, 4, 2), (64, 16384, 8, 1), (64, 65536, 30, 1), (32, 1024, 2, 2),
                                           (16, 128, 1, 3), (64, 64, 3, 3), (64, 256, 5, 2), (64, 128, 13, 2),
                                           (64, 64, 21, 1), (64, 32, 32, 2), (64, 16, 1, 3)])
def test_crt_lift_project(lb, n, m, batch, oracle_factory, engine_factory):
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    assert e.crt_limbs == o.crt_limbs
    assert e.crt_constant(0) == o.crt_modulus()
    for cm in range(m):
        assert e.crt_constant(1, cm) == o.crt_lifting(cm)
    a, _ = _inputs(o, batch)
    a[0, :, 0] = 0  # the reference skips zero residues (gmp.hpp:193)
    if n > 1:
        a[0, :, 1] = np.asarray(o.P[:m], dtype=a.dtype) - 1  # all residues p-1: X = Q-1, the largest lift
    da = e.to_device(a)
    limbs = e.crt_lift(da)
    got = limbs.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, o.crt_lift(a)), "poly2mpz differs"
    back = e.crt_project(limbs)
    assert np.array_equal(e.to_host(back), a), "mpz2poly(poly2mpz(a)) != a (tests/poly_mpz.cpp:19-29)"
    # reduction of wide non-negative integers (tests/poly_mpz.cpp:44-64): noise of 4 limbs, of the widest input
    # the multiply-accumulate table covers (2L+2), and of one beyond it (the Horner kernel)
    import torch
    rng = np.random.default_rng(0)
    L = e.crt_limbs
    for lin in ((4, 2 * L + 2, 2 * L + 5) if n <= 4096 else (4,)):
        wide = rng.integers(0, 2**63, size=(batch, n, lin), dtype=np.uint64) * np.uint64(2) + rng.integers(
            0, 2, size=(batch, n, lin), dtype=np.uint64)
        wide[0, 0, :] = np.uint64(0xFFFFFFFFFFFFFFFF)  # all-ones words: the accumulators' worst case
        dw = torch.from_numpy(wide.view(np.int64)).to(da.device)
        assert np.array_equal(e.to_host(e.crt_project(dw)), o.crt_project(wide)), "L_in = %d" % lin


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 1), (64, 8, 2)])
def test_host_pointer_entry_points(lb, n, m, oracle_factory, engine_factory):
    """The un-suffixed C entry points (host buffers in/out) used by the per-poly nfl::poly surface."""
    from nfllib_amd import OP_ADD, OP_MUL
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, 2)
    assert np.array_equal(e.h_ntt(a), o.ntt(a))
    assert np.array_equal(e.h_intt(a), o.intt(a))
    assert np.array_equal(e.h_pointwise(OP_ADD, a, b), o.pointwise(OP_ADD, a, b))
    assert np.array_equal(e.h_pointwise(OP_MUL, a, b), o.pointwise(OP_MUL, a, b))
    assert np.array_equal(e.h_polymul(a, b), o.polymul(a, b))
    assert e.h_any_eq(a, a) and not e.h_any_neq(a, a)
    assert np.array_equal(e.h_crt_project(e.h_crt_lift(a)), a)


def test_device_tables_match_reference_tables(oracle_factory, engine_factory):
    """psi_br[k] = phi^bitrev(k): cross-check against the reference's phis table (core.hpp:649-656)."""
    from nfllib_amd import TAB_INVDEGREE, TAB_MODULUS, TAB_PSI
    from oracle import oracle as O
    for lb, n, m in [(64, 4096, 4), (32, 1024, 2), (16, 128, 1)]:
        o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
        logn = n.bit_length() - 1
        br = np.array([int(format(k, "0%db" % logn)[::-1], 2) for k in range(n)])
        for cm in range(m):
            psi = e.table(TAB_PSI, cm).reshape(n, 2)
            assert np.array_equal(psi[:, 0], o.table(O.TAB_PHIS, cm)[br])
            assert np.array_equal(psi[:, 1], o.table(O.TAB_SHOUPPHIS, cm)[br])
            assert int(e.table(TAB_MODULUS, cm)[0]) == o.P[cm]
            assert int(e.table(TAB_INVDEGREE, cm)[0]) == int(o.table(O.TAB_INVPOLYDEGREE, cm)[0])


def test_full_size_properties_metric_shape(oracle_factory, engine_factory):
    """BASELINE configs[1] at a bench-sized batch: size-independent properties +
    sampled compare against the oracle."""
    lb, n, m, batch = 64, 4096, 4, 4096
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    c = e.polymul(a, b)
    # commutativity, and INTT(NTT(a)) == a over the whole batch
    assert not e.any_neq(c, e.polymul(b, a))
    # the "b already transformed" kernel must agree word for word with the fused one on the whole batch
    assert not e.any_neq(c, e.polymul(a, e.ntt_(b.clone()), b_is_ntt=True))
    rt = e.intt_(e.ntt_(a.clone()))
    assert not e.any_neq(rt, a)
    # linearity: (a+b)*b == a*b + b*b
    from nfllib_amd import OP_ADD
    lhs = e.polymul(e.pointwise(OP_ADD, a, b), b)
    rhs = e.pointwise(OP_ADD, c, e.polymul(b, b))
    assert not e.any_neq(lhs, rhs)
    # sampled polys regenerated on the CPU from the same counter stream
    rng = np.random.default_rng(1)
    hc = e.to_host(c)
    for idx in rng.choice(batch, size=8, replace=False):
        ha = o.fill_uniform(1, SEED, 0, first_poly=int(idx))
        hb = o.fill_uniform(1, SEED, 1, first_poly=int(idx))
        assert np.array_equal(hc[idx:idx + 1], o.polymul(ha, hb))


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (32, 1024, 2), (16, 128, 1), (64, 8, 2)])
def test_fused_expression_trees(lb, n, m, oracle_factory, engine_factory):
    """nflhip_eval: whole expression trees in one pass (core.hpp:24-37), vs the oracle's op-by-op result."""
    from nfllib_amd import NflHipError, OP_ADD, OP_COMPUTE_SHOUP, OP_MUL, OP_MUL_SHOUP, OP_SUB
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, 3)
    c = o.fill_uniform(3, 77, 0)
    da, db, dc = e.to_device(a), e.to_device(b), e.to_device(c)
    ADD, SUB, MUL, MSH, CSH = 0x10, 0x11, 0x12, 0x13, 0x14
    # a + b*c  (tests/poly_p.cpp:62-66)
    want = o.pointwise(OP_ADD, a, o.pointwise(OP_MUL, b, c))
    assert np.array_equal(e.to_host(e.eval([0, 1, 2, MUL, ADD], [da, db, dc])), want)
    # b - a*c  (tests/nfllib_demo_main_op.cpp:51)
    want = o.pointwise(OP_SUB, b, o.pointwise(OP_MUL, a, c))
    assert np.array_equal(e.to_host(e.eval([1, 0, 2, MUL, SUB], [da, db, dc])), want)
    # (a+b)*(a-c) + b*c, result aliasing an operand
    want = o.pointwise(OP_ADD, o.pointwise(OP_MUL, o.pointwise(OP_ADD, a, b), o.pointwise(OP_SUB, a, c)), o.pointwise(OP_MUL, b, c))
    out = da.clone()
    e.eval([0, 1, ADD, 0, 2, SUB, MUL, 1, 2, MUL, ADD], [out, db, dc], out=out)
    assert np.array_equal(e.to_host(out), want)
    # shoup(a*b, compute_shoup(b)) + c
    want = o.pointwise(OP_ADD, o.pointwise(OP_MUL, a, b), c)
    assert np.array_equal(e.to_host(e.eval([0, 1, 1, CSH, MSH, 2, ADD], [da, db, dc])), want)
    # host-pointer variant
    assert np.array_equal(e.h_eval([0, 1, 2, MUL, ADD], [a, b, c]), o.pointwise(OP_ADD, a, o.pointwise(OP_MUL, b, c)))
    # ... with four distinct operands: c + shoup(a * b, b') (tests/nfllib_demo_main_op.cpp:254; the result is formed over the first
    # operand's staging buffer), at one polynomial (pinned staging) and at a batch past it
    bp = o.pointwise(OP_COMPUTE_SHOUP, b)
    assert np.array_equal(e.h_eval([0, 1, 2, 3, MSH, ADD], [c, a, b, bp]), o.pointwise(OP_ADD, c, o.pointwise(OP_MUL, a, b)))
    assert np.array_equal(e.h_eval([0, 1, 2, 3, MSH, ADD], [c[:1], a[:1], b[:1], bp[:1]]), o.pointwise(OP_ADD, c[:1], o.pointwise(OP_MUL, a[:1], b[:1])))
    with pytest.raises(NflHipError):
        e.h_eval([0, 1, ADD, 2, ADD, 3, ADD, 4, ADD], [a, b, c, bp, a.copy()])     # five distinct operands: the host-pointer variant declines
    # malformed programs are rejected, not executed
    for bad in ([0, ADD], [0, 1], [0, 1, 2, 0, 1, ADD], [9, 0, ADD], [0, 1, 0x7f]):
        with pytest.raises(NflHipError):
            e.eval(bad, [da, db, dc])


_VARIANT_CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import torch
from nfllib_amd import Engine
from nfllib_amd.sharding import digest_words
out = {}
for n, m, batch in ((4096, 4, 512), (16384, 8, 4)):
    e = Engine(64, n, m)
    a = e.fill_uniform(e.empty(batch), int(sys.argv[2]), 0)
    b = e.fill_uniform(e.empty(batch), int(sys.argv[2]), 1)
    fb = e.ntt_(b.clone())
    d = {"ntt": digest_words(e.to_host(fb)),
         "intt": digest_words(e.to_host(e.intt_(a.clone()))),
         "polymul": digest_words(e.to_host(e.polymul(a, b))),
         "polymul_ntt": digest_words(e.to_host(e.polymul(a, fb, b_is_ntt=True)))}
    out["%d_%d" % (n, m)] = d
print(json.dumps(out))
"""


def test_assembly_and_compiled_kernels_agree():
    """Two independently produced device programs -- the generated assembly (tools/gen_*_asm.py, default) and the
    hipcc-compiled templates (NFLHIP_VARIANT=hipcc when the context is created) -- must give the same words for every
    entry point they serve, over whole batches; for the long rows that compares the plans too (register-resident rows
    / three-role pipeline / one launch of persistent workgroups against streaming passes around compiled blocks)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for v in ("asm", "hipcc"):
        env = dict(os.environ, NFLHIP_VARIANT=v)
        r = subprocess.run([sys.executable, "-c", _VARIANT_CHILD, root, str(SEED)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[v] = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["asm"] == got["hipcc"]


_ROWS_CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import torch
from nfllib_amd import Engine
from nfllib_amd.sharding import digest_words
out = {}
for n, m, batch in ((8192, 2, 5), (8192, 1, 1), (16384, 8, 5), (16384, 1, 1), (32768, 2, 3), (32768, 2, 130), (65536, 3, 2),
                    (65536, 2, 5), (65536, 1, 1)):
    e = Engine(64, n, m)
    a = e.fill_uniform(e.empty(batch), int(sys.argv[2]), 0)
    b = e.fill_uniform(e.empty(batch), int(sys.argv[2]), 1)
    c = e.polymul(a, b)
    fb = e.ntt_(b.clone())
    d = {"polymul": digest_words(e.to_host(c)), "ntt": digest_words(e.to_host(fb)),
         "intt": digest_words(e.to_host(e.intt_(a.clone()))),
         "polymul_ntt": digest_words(e.to_host(e.polymul(a, fb, b_is_ntt=True)))}
    assert not e.any_neq(e.intt_(fb.clone()), b)
    b2 = b.clone()
    e.polymul(a, b2, out=b2)        # in place on the second operand
    assert not e.any_neq(b2, c)
    fb2, a2 = fb.clone(), a.clone()
    e.polymul(a, fb2, out=fb2, b_is_ntt=True)   # pre-transformed operand, in place on either side
    assert not e.any_neq(fb2, c)
    e.polymul(a2, fb, out=a2, b_is_ntt=True)
    assert not e.any_neq(a2, c)
    e.polymul(a, b, out=a)          # in place on the first operand
    assert not e.any_neq(a, c)
    out["%d_%d" % (n, m) + ("_%d" % batch if batch > 100 else "")] = d
print(json.dumps(out))
"""


def test_every_plan_of_the_long_rows_gives_the_same_words():
    """Rows of 8192 ... 65536 words: the default plan of every shape (row-resident kernels at 8192 / 16384; at 32768 the
    one-launch plan for the small batch and the register-resident rows for the large one; the pipeline at 65536),
    NFLHIP_XCD=0 / 1 (never / always the one-launch plan at 32768 and 65536) and the compiled kernels
    (NFLHIP_VARIANT=hipcc): identical digests, in-place variants included; test_polymul compares the default against
    the oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, extra in (("default", {}), ("xcd0", {"NFLHIP_XCD": "0"}), ("xcd1", {"NFLHIP_XCD": "1"}), ("hipcc", {"NFLHIP_VARIANT": "hipcc"})):
        env = dict(os.environ)
        env.pop("NFLHIP_XCD", None)
        env.pop("NFLHIP_VARIANT", None)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", _ROWS_CHILD, root, str(SEED)], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["default"] == got["xcd0"] == got["xcd1"] == got["hipcc"]


def test_concurrent_host_threads_on_distinct_streams(oracle_factory, engine_factory):
    """The threading contract of include/nflhip.h: a context is immutable after creation, so host threads may call
    the *_dev entry points concurrently on distinct streams (SURVEY.md 8(b) "Threading").  The two largest shapes
    run the multi-launch plans, which share the context's scratch: successive calls from different threads and
    streams (fused plan, then the b-pre-transformed composed plan) are ordered by the library's own events."""
    import threading
    import torch
    for lb, n, m, batch in ((64, 4096, 4, 64), (64, 16384, 2, 8), (32, 1024, 2, 64), (64, 32768, 2, 8), (64, 65536, 1, 8)):
        e = engine_factory(lb, n, m)
        a = e.fill_uniform(e.empty(batch), SEED, 0)
        b = e.fill_uniform(e.empty(batch), SEED, 1)
        want = e.to_host(e.polymul(a, b))
        torch.cuda.synchronize()
        results, errors = {}, []

        def work(tid):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    aa, bb = a.clone(), b.clone()
                    fb = e.ntt_(b.clone(), stream=st)
                    out = out2 = None
                    for _ in range(10):
                        out = e.polymul(aa, bb, stream=st)
                        out2 = e.polymul(aa, fb, b_is_ntt=True, stream=st)
                        f = e.ntt_(aa.clone(), stream=st)
                        aa = e.intt_(f, stream=st)
                    st.synchronize()
                    results[tid] = e.to_host(out)
                    assert np.array_equal(e.to_host(out2), results[tid])
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errors, errors
        for tid in range(4):
            assert np.array_equal(results[tid], want), (lb, n, m, tid)


def test_helper_stream_plan_then_pipeline_on_another_stream(engine_factory):
    """n = 65536: polymul_ntt_dev (helper-stream plan, reads the context's scratch on aux streams) issued on stream A,
    immediately followed by polymul_dev (single-stream pipeline, rewrites the same scratch) on stream B: the library
    must order B's scratch writes after A's helper streams (api.hip pipe64k branch)."""
    import torch
    e = engine_factory(64, 65536, 2)
    batch = 16
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    b = e.fill_uniform(e.empty(batch), SEED, 1)
    fb = e.ntt_(b.clone())
    want = e.to_host(e.polymul(a, b))
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(8):
        with torch.cuda.stream(sa):
            o1 = e.polymul(a, fb, b_is_ntt=True, stream=sa)
        with torch.cuda.stream(sb):
            o2 = e.polymul(a, b, stream=sb)
        with torch.cuda.stream(sa):
            o3 = e.polymul(a, fb, b_is_ntt=True, stream=sa)
        sa.synchronize(); sb.synchronize()
        for o in (o1, o2, o3):
            assert np.array_equal(e.to_host(o), want)


def test_concurrent_comparisons_own_their_result_slot(engine_factory):
    """nflhip_any_eq_dev / nflhip_any_neq_dev from several host threads on distinct streams (allowed by
    include/nflhip.h) while the host-pointer compare runs too: every call has its own device result slot, so no
    call can see another call's memset / atomicOr."""
    import threading
    import torch
    e = engine_factory(64, 4096, 4)
    batch = 256
    a = e.fill_uniform(e.empty(batch), SEED, 0)
    same = a.clone()
    diff = a.clone()
    diff[batch - 1, 3, 4095] ^= 1
    ha, hd = e.to_host(a[:2]), e.to_host(diff[batch - 2:])
    torch.cuda.synchronize()
    errors = []

    def work(tid):
        try:
            st = torch.cuda.Stream()
            for k in range(200):
                if tid == 0:
                    assert e.h_any_neq(ha, ha) is False and e.h_any_eq(ha, ha) is True
                    assert e.h_any_neq(e.to_host(a[batch - 2:]), hd) is True
                elif (tid + k) & 1:
                    assert e.any_neq(a, same, stream=st) is False, "false positive"
                    assert e.any_eq(a, same, stream=st) is True
                else:
                    assert e.any_neq(a, diff, stream=st) is True, "missed the one differing word"
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


@pytest.mark.parametrize("lb,n,m", [(64, 4096, 4), (64, 8192, 2), (64, 16384, 2), (64, 65536, 2), (32, 1024, 2), (64, 1024, 2),
                                    (64, 2048, 1), (32, 2048, 1), (32, 4096, 2)])
def test_adversarial_coefficient_values(lb, n, m, oracle_factory, engine_factory):
    """The tuned kernels lean on approximate quotients, two-bit folds and lazy ranges whose proofs are about extreme
    words: feed polynomials built from boundary values (0, 1, p-1, p-2, 2^k, 2^k - 1, runs and alternations of them)
    instead of uniform noise and require bit-equality with the oracle on every entry point."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    rng = np.random.default_rng(lb * n + m)
    batch = 4
    a = np.zeros((batch, m, n), dtype=o.dtype)
    b = np.zeros_like(a)
    for cm in range(m):
        p = int(o.P[cm])
        specials = [0, 1, 2, p - 1, p - 2, p // 2, p // 2 + 1, (1 << (lb - 2)) % p, ((1 << (lb - 2)) - 1) % p,
                    (1 << (lb // 2)) % p, ((1 << (lb // 2)) - 1) % p, (1 << (lb // 2 - 1)) % p, (p - (1 << (lb // 2))) % p]
        sp = np.array(specials, dtype=np.uint64).astype(o.dtype)
        for x in (a, b):
            x[0, cm, :] = p - 1                                            # all p-1
            x[1, cm, :] = sp[rng.integers(0, len(sp), n)]                  # random mix of boundary values
            x[2, cm, :] = np.where(np.arange(n) % 2 == 0, p - 1, 0).astype(o.dtype)
            x[3, cm, :] = sp[(np.arange(n) // 16 + (x is b)) % len(sp)]    # runs of 16 equal boundary values
    da, db = e.to_device(a), e.to_device(b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), o.polymul(a, b))
    fa = e.ntt_(da.clone())
    assert np.array_equal(e.to_host(fa), o.ntt(a))
    assert np.array_equal(e.to_host(e.intt_(da.clone())), o.intt(a))
    assert np.array_equal(e.to_host(e.intt_(fa.clone())), a)
    assert np.array_equal(e.to_host(e.polymul(db, fa, b_is_ntt=True)), o.polymul(a, b))


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 65536])
def test_every_mirrored_62_bit_modulus_runs_the_delta_form_kernels(n, oracle_factory, engine_factory):
    """The first 92 mirrored 62-bit moduli in one context: every prime whose 2^62 - p still fits 32 bits (the 93rd on run the
    general-modulus kernels, tests/test_gpu_big_delta.py).  From the 46th on 2^62 - p needs the full 32 bits (c >= 1024 in
    params.hpp:94-97's c*2^21 - 1), which is the widest delta the multiply-add butterflies accept: uniform words and
    the extreme ones (p-1 everywhere: largest products, folds and quotients) must still match the oracle bit for bit
    on the wave, block, row-resident and pipeline kernels.  (CRT needs <= 32 moduli and is covered elsewhere.)"""
    from nfllib_amd.params import params
    lb, m, batch = 64, 92, 3 if n < 65536 else 2
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    assert (1 << 62) - int(o.P[45]) >= (1 << 31) and (1 << 62) - int(o.P[91]) < (1 << 32) <= (1 << 62) - int(params(64).P[92])
    a, b = _inputs(o, batch)
    pm1 = (np.array([int(p) for p in o.P], dtype=np.uint64) - 1).astype(o.dtype)[:, None]
    a[0], b[0] = pm1, pm1
    b[1] = np.where(np.arange(n) % 2 == 0, pm1, 0)
    da, db = e.to_device(a), e.to_device(b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), o.polymul(a, b))
    fa = e.ntt_(da.clone())
    assert np.array_equal(e.to_host(fa), o.ntt(a))
    assert np.array_equal(e.to_host(e.intt_(da.clone())), o.intt(a))
    assert np.array_equal(e.to_host(e.polymul(db, fa, b_is_ntt=True)), o.polymul(a, b))
    assert np.array_equal(e.to_host(e.intt_(fa)), a)
    from nfllib_amd import OP_MUL
    assert np.array_equal(e.to_host(e.pointwise(OP_MUL, da, db)), o.pointwise(OP_MUL, a, b))


def test_host_pointer_calls_of_many_sizes_match_the_resident_path(engine_factory):
    """The host-pointer entry points stage pageable caller memory through context-owned device buffers that grow on
    demand: calls from 32 bytes to 50 MiB per operand, growing and shrinking, must equal the resident path."""
    lb, n, m = 64, 4096, 4   # 128 KiB per polynomial
    e = engine_factory(lb, n, m)
    for batch in (1, 32, 33, 400, 131):
        a = e.fill_uniform(e.empty(batch), SEED + batch, 0)
        b = e.fill_uniform(e.empty(batch), SEED + batch, 1)
        ha, hb = e.to_host(a), e.to_host(b)
        assert np.array_equal(e.h_polymul(ha, hb), e.to_host(e.polymul(a, b))), batch
        assert np.array_equal(e.h_ntt(ha), e.to_host(e.ntt_(a.clone()))), batch
        assert e.h_any_neq(ha, hb) and not e.h_any_neq(ha, ha.copy())
    e2 = engine_factory(16, 16, 1)  # 32-byte polynomials
    a = e2.fill_uniform(e2.empty(3), SEED, 0)
    assert np.array_equal(e2.h_intt(e2.h_ntt(e2.to_host(a))), e2.to_host(a))


@pytest.mark.parametrize("lb,n,m,batches", [(64, 4096, 4, (128, 129, 200, 453)), (32, 1024, 2, (2048, 5000)), (16, 128, 1, (65536, 70001)),
                                            (64, 32768, 2, (32, 37))])
def test_large_host_pointer_calls_through_the_pinned_pipeline(lb, n, m, batches, oracle_factory, engine_factory):
    """Host arrays of two or more 8 MiB chunks per operand take the pipelined path (pinned staging slots, host threads
    copying chunk k + 1 while chunk k crosses PCIe, chunk k - 1 is computed and chunk k - 2 returns): every entry point
    that uses it against the resident path over the same words (chunk boundaries, the short last chunk, in-place
    transforms, operands that alias each other) -- and a sample against the oracle."""
    from nfllib_amd import OP_ADD, OP_MUL, OP_MUL_SHOUP, OP_COMPUTE_SHOUP
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    for batch in batches:
        a = e.fill_uniform(e.empty(batch), SEED + batch, 0)
        b = e.fill_uniform(e.empty(batch), SEED + batch, 1)
        ha, hb = e.to_host(a), e.to_host(b)
        hc = e.h_polymul(ha, hb)
        assert np.array_equal(hc, e.to_host(e.polymul(a, b))), batch
        for k in (0, batch // 2, batch - 1):   # first, a middle and the last (short) chunk against the CPU checker
            assert np.array_equal(hc[k:k + 1], o.polymul(ha[k:k + 1], hb[k:k + 1])), (batch, k)
        assert np.array_equal(e.h_polymul(ha, ha), e.to_host(e.polymul(a, a))), batch          # aliased operands
        f = e.h_ntt(ha)
        assert np.array_equal(f, e.to_host(e.ntt_(a.clone()))), batch
        assert np.array_equal(e.h_intt(f), ha), batch
        assert np.array_equal(e.h_pointwise(OP_ADD, ha, hb), e.to_host(e.pointwise(OP_ADD, a, b))), batch
        assert np.array_equal(e.h_pointwise(OP_MUL, ha, hb), e.to_host(e.pointwise(OP_MUL, a, b))), batch
        hbp = e.h_pointwise(OP_COMPUTE_SHOUP, hb)
        assert np.array_equal(e.h_pointwise(OP_MUL_SHOUP, ha, hb, hbp), e.to_host(e.pointwise(OP_MUL, a, b))), batch
        prog = [0, 1, 0x12, 2, 0x10]                                                              # a * b + c
        assert np.array_equal(e.h_eval(prog, [ha, hb, hc]), e.to_host(e.eval(prog, [a, b, e.to_device(hc)]))), batch


def test_concurrent_host_pointer_calls_on_two_contexts(engine_factory):
    """Two host threads, each with its own context, inside the pinned pipeline at once (one thread per GPU of the batch
    split; two ring types of one program): the process-wide pool of copying threads serves one job at a time, so neither
    call sees the other's slices.  Every product against the resident path over the same words, several rounds."""
    import threading
    e1, e2 = engine_factory(64, 4096, 4), engine_factory(64, 4096, 2)
    jobs = []
    for e, batch in ((e1, 200), (e2, 330)):       # 3.1 and 2.6 chunks of 8 MiB per operand
        a = e.fill_uniform(e.empty(batch), SEED + batch, 0)
        b = e.fill_uniform(e.empty(batch), SEED + batch, 1)
        jobs.append((e, e.to_host(a), e.to_host(b), e.to_host(e.polymul(a, b))))
    errors = []

    def worker(e, ha, hb, want):
        try:
            for _ in range(6):
                if not np.array_equal(e.h_polymul(ha, hb), want):
                    errors.append("a concurrent host-pointer product differs from the resident one")
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))
    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors and not any(t.is_alive() for t in threads), errors


@pytest.mark.parametrize("lb,n,m,batch", [(64, 131072, 2, 2), (64, 262144, 1, 1), (64, 1048576, 1, 1), (32, 16384, 2, 2)])
def test_degrees_beyond_the_baseline_shapes(lb, n, m, batch, oracle_factory, engine_factory):
    """Up to params<uint64_t>::kMaxPolyDegree = 2^20 (params.hpp:98-99): two and more streaming passes around the
    4096-word blocks; product, both transforms and the b-pre-transformed product against the oracle."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    a, b = _inputs(o, batch)
    da, db = e.to_device(a), e.to_device(b)
    want = o.polymul(a, b)
    assert np.array_equal(e.to_host(e.polymul(da, db)), want)
    f = e.ntt_(da.clone())
    assert np.array_equal(e.to_host(f), o.ntt(a))
    assert np.array_equal(e.to_host(e.intt_(f)), a)
    assert np.array_equal(e.to_host(e.polymul(da, e.ntt_(db.clone()), b_is_ntt=True)), want)


@pytest.mark.parametrize("n,m,batch", [(256, 21, 3), (64, 22, 1), (128, 25, 2), (32, 29, 2), (256, 31, 1), (64, 30, 5),
                                       (4096, 23, 40), (65536, 30, 2), (64, 20, 2), (64, 32, 1)])
def test_crt_lift_on_the_matrix_cores(n, m, batch, oracle_factory, engine_factory):
    """GMP::poly2mpz (gmp.hpp:183-209) with 21 .. 32 62-bit moduli runs as an int8 GEMM (kernels_crt_mfma.hip): extreme
    residues (X = Q - 1, 0, 1, one residue set), several modulus counts' zero padding, more tiles than workgroups; 32 moduli
    take the instantiation without a quotient row, 20 the VALU kernel below the range."""
    o, e = oracle_factory(64, n, m), engine_factory(64, n, m)
    a = o.fill_uniform(batch, SEED, 0)
    P = np.asarray(o.P[:m], dtype=a.dtype)
    a[0, :, 0] = P - 1
    a[0, :, 1] = 0
    a[0, :, 2] = 1
    a[0, :, 3] = 0; a[0, m - 1, 3] = P[m - 1] - 1
    a[0, :, 4] = P - 1; a[0, 0, 4] = 0
    a[0, :, 5] = P >> 1
    # lifted values that put runs of all-ones / all-zero digits against the boundaries of the kernel's four 512-bit parts
    # (the carry / borrow that runs THROUGH a part is a ballot-guarded slow path random residues never take); small X also
    # makes the quotient estimate land one below the floor, so both S - tQ = X and X + Q come by
    Q = o.crt_modulus()
    adv = [(1 << (32 * j)) - 1 for j in range(1, 2 * o.crt_limbs)] + [1 << (512 * k) for k in (1, 2, 3)]
    adv += [(1 << (512 * k)) + 1 for k in (1, 2, 3)] + [Q - (1 << (512 * k)) for k in (1, 2, 3)] + [Q - 2, Q >> 1, (Q >> 1) + 1]
    adv += [((1 << 512) - 1) << 512, ((1 << 512) - 1) << 1024, (1 << 1024) - (1 << 512)]
    adv = [x for x in adv if 0 <= x < Q][: max(0, n - 6)]
    for idx, x in enumerate(adv):
        a[0, :, 6 + idx] = [x % int(p) for p in o.P[:m]]
    limbs = e.crt_lift(e.to_device(a))
    got = e.to_host(limbs).view(np.uint64)
    for idx, x in enumerate(adv):
        assert int.from_bytes(got[0, 6 + idx].tobytes(), "little") == x, "adversarial value %d (%d bits)" % (idx, x.bit_length())
    if n <= 4096:
        assert np.array_equal(got, o.crt_lift(a)), "poly2mpz differs"
    else:                                            # the oracle's big integers on a sample; everything through the round trip
        o2 = oracle_factory(64, 512, m)              # the lift is per coefficient: a 512-coefficient oracle on a sample
        assert o2.P[:m] == o.P[:m]
        assert np.array_equal(got[:, :512], o2.crt_lift(np.ascontiguousarray(a[:, :, :512])))
    assert int.from_bytes(got[0, 0].tobytes(), "little") == Q - 1
    assert int.from_bytes(got[0, 1].tobytes(), "little") == 0 and int.from_bytes(got[0, 2].tobytes(), "little") == 1
    assert np.array_equal(e.to_host(e.crt_project(limbs)), a)
    # mpz2poly of wide non-negative integers (tests/poly_mpz.cpp:44-64) is the same GEMM the other way round, 5 .. 64 words
    import torch
    rng = np.random.default_rng(1)
    nn = min(n, 512)
    o2 = o if n == nn else oracle_factory(64, nn, m)
    for lin in (5, e.crt_limbs, 32, 33, 2 * e.crt_limbs, 64):      # (33 .. 64 words: two GEMM passes)
        wide = rng.integers(0, 2**63, size=(batch, n, lin), dtype=np.uint64) * np.uint64(2) + rng.integers(
            0, 2, size=(batch, n, lin), dtype=np.uint64)
        wide[0, 0, :] = np.uint64(0xFFFFFFFFFFFFFFFF)   # every byte 255: the accumulators' worst case
        wide[0, 1, :] = 0
        wide[0, 2, :] = np.uint64(0x8080808080808080)   # every byte the bias itself
        dw = torch.from_numpy(wide.view(np.int64)).to(limbs.device)
        got_p = e.to_host(e.crt_project(dw))
        assert np.array_equal(got_p[:, :, :nn], o2.crt_project(np.ascontiguousarray(wide[:, :nn]))), "L_in = %d" % lin


@pytest.mark.parametrize("lb,n,m,batch", [(64, 64, 33, 2), (64, 256, 40, 1), (64, 16, 100, 2), (32, 64, 64, 2), (32, 32, 291, 1),
                                          (64, 8, 1000, 1)])
def test_crt_beyond_32_moduli(lb, n, m, batch, oracle_factory, engine_factory):
    """GMP::poly2mpz / mpz2poly have no modulus-count cap in the reference (gmp.hpp:113-219, any NbModuli up to
    params<T>::kMaxNbModuli = 1000 / 291): the limb-serial lift and the generic projection cover every count."""
    o, e = oracle_factory(lb, n, m), engine_factory(lb, n, m)
    assert e.crt_limbs == o.crt_limbs and e.crt_constant(0) == o.crt_modulus()
    a = o.fill_uniform(batch, SEED, 0)
    a[0, :, 0] = [p - 1 for p in o.P]          # X = Q - 1
    a[0, :, 1] = 0
    a[0, :, 2] = 1
    limbs = e.crt_lift(e.to_device(a))
    assert np.array_equal(e.to_host(limbs).view(np.uint64), o.crt_lift(a)), "poly2mpz differs"
    Q = o.crt_modulus()
    x0 = int.from_bytes(e.to_host(limbs)[0, 0].view(np.uint64).tobytes(), "little")
    assert x0 == Q - 1
    assert np.array_equal(e.to_host(e.crt_project(limbs)), a)
    assert np.array_equal(e.h_crt_project(e.h_crt_lift(a)), a)
