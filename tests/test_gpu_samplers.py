"""-m gpu: the device samplers (nflhip_sample*_dev) through the C ABI.

uniform / non_uniform / ZO_dist are checked EXACTLY: the device output must equal the reference's rule
(oracle/samplers.py, pinned against the real reference by tests/test_samplers_cpu.py) applied to the very keystream
words the device consumed (ChaCha20, restated in numpy).  hwt_dist and gaussian are checked structurally, exactly
against the cumulative table, and statistically against histograms of the real reference's samples."""
import os

import numpy as np
import pytest

from nfllib_amd import DIST_BOUNDED, DIST_HWT, DIST_UNIFORM, DIST_ZO, NflHipError
from oracle import samplers as S

pytestmark = pytest.mark.gpu
KEY = bytes((7 * i + 3) & 0xFF for i in range(32))
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers.npz"))
SHAPES = [(64, 4096, 4, 3), (64, 64, 3, 5), (32, 1024, 2, 3), (16, 128, 1, 4)]
IDS = ["u%d-n%d-m%d" % s[:3] for s in SHAPES]
_MASK = {16: 0xFFFF, 32: 0xFFFFFFFF, 64: 0xFFFFFFFFFFFFFFFF}


def _P(e):
    from nfllib_amd.params import params
    return [int(x) for x in params(e.limb_bits).P[:e.nmoduli]]


def test_keystream_matches_chacha20(engine_factory):
    e = engine_factory(64, 64, 3)
    for first, count in ((0, 64), (5, 100), (8, 8), (1023, 3), (0, 1)):
        got = e.random_words(count, KEY, stream_id=11, first_word=first).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, S.chacha20_words(KEY, 11, first, count)), (first, count)
    zero = e.random_words(8, bytes(32), 0).cpu().numpy().view(np.uint64)
    assert zero[0] == 0x903DF1A0ADE0B876  # first word of the all-zero ChaCha20 test vector


@pytest.mark.parametrize("lb,n,m,batch", SHAPES, ids=IDS)
def test_uniform_bounded_zo_are_the_reference_rules(lb, n, m, batch, engine_factory):
    e = engine_factory(lb, n, m)
    P, dt = _P(e), e.np_dtype
    d = e.sample(e.empty(batch), DIST_UNIFORM, KEY, stream_id=1)
    words = (S.chacha20_words(KEY, 1, 0, batch * m * n, counter_base=S.domain_base("uniform")) & np.uint64(_MASK[lb])).astype(dt).reshape(batch, m, n)
    got = e.to_host(d)
    assert np.array_equal(got, S.uniform(words, P))
    assert all((got[:, cm] < P[cm]).all() for cm in range(m))
    # sharded generation: any split of the batch gives the same words
    lo = e.sample(e.empty(1), DIST_UNIFORM, KEY, stream_id=1, first_poly=0)
    hi = e.sample(e.empty(batch - 1), DIST_UNIFORM, KEY, stream_id=1, first_poly=1)
    assert np.array_equal(np.concatenate([e.to_host(lo), e.to_host(hi)]), got)
    # same key and stream id for the bounded and the ZO calls below: the distribution tag in the block counter keeps
    # their keystream words apart (a public polynomial must not share words with the noise next to it)
    cw = S.chacha20_words(KEY, 2, 0, batch * n, counter_base=S.domain_base("bounded")).reshape(batch, n)
    zw = S.chacha20_words(KEY, 2, 0, batch * n, counter_base=S.domain_base("zo")).reshape(batch, n)
    for ub, amp in ((1, 1), (2, 1), (5, 3), (1000, 1), (1 << 12, 1)):
        if ub * amp >= min(P) // 2:
            continue
        d = e.sample(e.empty(batch), DIST_BOUNDED, KEY, stream_id=2, param0=ub, param1=amp)
        assert np.array_equal(e.to_host(d), S.non_uniform(cw, P, ub, amp, dtype=dt)), (ub, amp)
    for rho in (0x7F, 0, 255, 10):
        d = e.sample(e.empty(batch), DIST_ZO, KEY, stream_id=2, param0=rho)
        assert np.array_equal(e.to_host(d), S.zo_dist(zw & np.uint64(0xFF), P, rho, canonical=True, dtype=dt)), rho
    # a different stream id is a different polynomial
    assert not np.array_equal(e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, KEY, stream_id=3)), got)


def test_argument_errors_follow_the_reference(engine_factory):
    e = engine_factory(32, 1024, 2)
    d = e.empty(1)
    with pytest.raises(NflHipError, match="upper_bound is larger than the modulus"):   # core.hpp:205-210
        e.sample(d, DIST_BOUNDED, KEY, param0=min(_P(e)), param1=1)
    for bad in (dict(dist=DIST_HWT, param0=0), dict(dist=DIST_HWT, param0=1025), dict(dist=DIST_ZO, param0=256),
                dict(dist=9)):
        with pytest.raises(NflHipError):
            e.sample(d, bad.pop("dist"), KEY, **bad)
    with pytest.raises(NflHipError):
        e.gauss_create(-1.0)


def _chi2_two_sample(a, b):
    """chi-square statistic / dof for two histograms over the same bins (bins with few counts pooled)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    keep = (a + b) >= 20
    a2, b2 = np.append(a[keep], a[~keep].sum()), np.append(b[keep], b[~keep].sum())
    k1, k2 = np.sqrt(b2.sum() / a2.sum()), np.sqrt(a2.sum() / b2.sum())
    nz = (a2 + b2) > 0
    stat = (((k1 * a2 - k2 * b2) ** 2)[nz] / (a2 + b2)[nz]).sum()
    return stat / max(int(nz.sum()) - 1, 1)


def test_hamming_weight_distribution(engine_factory):
    e = engine_factory(64, 1024, 2)
    P = _P(e)
    h, batch = 64, 400
    d = e.to_host(e.sample(e.empty(batch), DIST_HWT, KEY, stream_id=5, param0=h))
    nz = d[:, 0] != 0
    assert (nz.sum(axis=1) == h).all(), "exactly h non-zero coefficients per polynomial (core.hpp:347-391)"
    assert np.array_equal(nz, d[:, 1] != 0), "the same support in every residue row"
    c = S.centered(d, P)
    assert np.array_equal(c[:, 0], c[:, 1]) and set(np.unique(c).tolist()) == {-1, 0, 1}
    plus = int((c[:, 0] == 1).sum())
    assert abs(plus - batch * h / 2) < 5 * np.sqrt(batch * h / 4), "signs are fair"
    # supports are uniform: position histogram against the real reference's (two-sample) and against flat
    pos = nz.sum(axis=0)
    assert _chi2_two_sample(pos, GOLD["hwt_64/pos_hist"]) < 1.35
    assert abs(int(GOLD["hwt_64/plus"]) - int(GOLD["hwt_64/reps"]) * h / 2) < 5 * np.sqrt(int(GOLD["hwt_64/reps"]) * h / 4)
    # deterministic, shardable, full-weight edge
    again = e.to_host(e.sample(e.empty(batch), DIST_HWT, KEY, stream_id=5, param0=h))
    assert np.array_equal(again, d)
    part = e.to_host(e.sample(e.empty(10), DIST_HWT, KEY, stream_id=5, param0=h, first_poly=100))
    assert np.array_equal(part, d[100:110])
    full = e.to_host(e.sample(e.empty(2), DIST_HWT, KEY, stream_id=6, param0=1024))
    assert (full[:, 0] != 0).all()


@pytest.mark.parametrize("sigma", [3.2, 20.0])
def test_gaussian(sigma, engine_factory):
    e = engine_factory(64, 1024, 2)
    P = _P(e)
    g = e.gauss_create(sigma, security=128, samples=1024)
    info = e.gauss_info(g)
    # the reference's parameters for these arguments (FastGaussianNoise.hpp:239-272): k = 128 + 1 + 10
    k = 139.0
    assert abs(info["tail"] ** 2 - 2 * np.log(info["tail"]) - 1 - 2 * k * np.log(2)) < 0.2
    assert info["entries"] == 2 * int(np.ceil(info["tail"] * sigma)) + 1 and info["x_min"] == -(info["entries"] // 2)
    assert info["bit_precision"] == int(np.ceil(k + np.log2(2 * info["tail"] * sigma))) and info["words"] == 3
    tab = info["table"]
    top = tab[:, 0].astype(np.float64) / 2.0 ** 64
    pmf = S.gaussian_pmf(sigma, 0.0, info["x_min"], info["entries"])
    assert np.abs(np.diff(np.concatenate([[0.0], top])) - pmf).max() < 1e-15, "table = cumulative discrete Gaussian"
    assert (tab[-1] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    flat = [int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big") for row in tab]
    assert all(x <= y for x, y in zip(flat, flat[1:]))
    batch = 200
    d = e.to_host(e.sample_gauss(e.empty(batch), g, KEY, stream_id=9))
    c = S.centered(d, P)
    assert np.array_equal(c[:, 0], c[:, 1]), "one integer per coefficient, replicated over the moduli"
    v = c[:, 0]
    # exact: inversion of the very keystream words through the table
    r = S.gaussian_words(KEY, 9, 0, 4 * 1024, 3)
    assert np.array_equal(S.gaussian_from_table(r, tab, info["x_min"]), v[:4].reshape(-1))
    # statistics: moments, tail, and a two-sample test against the real reference's samples
    flatv = v.reshape(-1)
    assert abs(flatv.mean()) < 5 * sigma / np.sqrt(flatv.size) and abs(flatv.var() / sigma ** 2 - 1) < 0.02
    assert np.abs(flatv).max() <= np.ceil(info["tail"] * sigma)
    rh, rlo = GOLD["gauss_%g/hist" % sigma], int(GOLD["gauss_%g/lo" % sigma])
    lo, hi = min(rlo, int(flatv.min())), max(rlo + rh.size - 1, int(flatv.max()))
    a = np.bincount(flatv - lo, minlength=hi - lo + 1)
    b = np.zeros(hi - lo + 1, dtype=np.int64)
    b[rlo - lo:rlo - lo + rh.size] = rh
    assert _chi2_two_sample(a, b) < 1.5
    # FastGaussianNoise::getNoise: the raw integers behind the same keystream positions, device and host entry
    raw = e.gauss_noise(g, 3 * 1024 + 5, KEY, stream_id=9, first_sample=1024 - 3).cpu().numpy()
    assert np.array_equal(raw, flatv[1024 - 3:4 * 1024 + 2])
    assert np.array_equal(e.h_gauss_noise(g, 777, KEY, stream_id=9), flatv[:777])
    # amplifier and sharding
    amp = e.to_host(e.sample_gauss(e.empty(2), g, KEY, stream_id=9, amplifier=5))
    assert np.array_equal(S.centered(amp, P)[:, 0], 5 * v[:2])
    part = e.to_host(e.sample_gauss(e.empty(3), g, KEY, stream_id=9, first_poly=50))
    assert np.array_equal(part, d[50:53])
    e.gauss_destroy(g)
    # off-centre table
    g2 = e.gauss_create(sigma, security=64, samples=1024, center=2.5)
    i2 = e.gauss_info(g2)
    v2 = S.centered(e.to_host(e.sample_gauss(e.empty(100), g2, KEY, stream_id=4)), P)[:, 0].reshape(-1)
    assert abs(v2.mean() - 2.5) < 5 * sigma / np.sqrt(v2.size) and i2["words"] == 2
    e.gauss_destroy(g2)


_TIE_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from nfllib_amd import Engine
from oracle import samplers as S
import ctypes
from nfllib_amd import _lib
_lib.lib.nflhip_debug_gauss_tie_shift.argtypes = [ctypes.c_int]
_lib.lib.nflhip_debug_gauss_tie_shift.restype = None
_lib.lib.nflhip_debug_gauss_tie_shift(int(sys.argv[1]))
KEY = bytes(range(32))
for lb, n, m, batch, sigma, sec, first in ((64, 1024, 2, 3, 3.2, 128, 0), (64, 64, 1, 13, 20.0, 128, 5), (32, 256, 1, 5, 3.2, 64, 2),
                                          (64, 4, 1, 9, 3.2, 128, 3), (16, 128, 1, 7, 2.0, 20, 0), (64, 4096, 1, 2, 215.0, 100, 1)):
    e = Engine(lb, n, m)
    P = [int(e.table(1, cm)[0]) for cm in range(m)]
    g = e.gauss_create(sigma, security=sec, samples=1024)
    info = e.gauss_info(g)
    d = e.to_host(e.sample_gauss(e.empty(batch), g, KEY, stream_id=21, first_poly=first))
    v = S.centered(d, P)[:, 0].reshape(-1)
    r = S.gaussian_words(KEY, 21, first * n, batch * n, info["words"])
    want = S.gaussian_from_table(r, info["table"], info["x_min"])
    assert np.array_equal(v, want), (lb, n, m, sigma, sec, int((v != want).sum()))
    e.gauss_destroy(g)
print("TIE_OK")
"""


@pytest.mark.parametrize("tie_shift", ["0", "56", "63"])
def test_gaussian_lazy_precision_equals_full_precision_inversion(tie_shift):
    """One keystream word per sample; the lower words of the W-word uniform number come from the secondary stream only
    when the first word ties with a table entry (2^-64 per entry in production).  The debug entry point
    nflhip_debug_gauss_tie_shift (include/nflhip_debug.h; never called in production) widens what
    counts as a tie (56: equal top bytes; 63: equal top bits, i.e. nearly every comparison) without changing the value,
    so the tie path is exercised: every shape must still equal the oracle's full-precision inversion of the same words
    (1-, 2- and 3-word tables, the 8-per-thread kernel with ragged wave tiles and the per-coefficient one, shards)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _TIE_CHILD % {"root": root}, tie_shift], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TIE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("sigma,security,words", [(3.19, 256, 5), (20.0, 300, 6), (3.19, 200, 4)])
def test_gaussian_beyond_192_bits(sigma, security, words, engine_factory):
    """Tables of 4, 5 and 6 words per entry (the reference takes its precision from MPFR and has no cap,
    FastGaussianNoise.hpp:239-272): the device sampler is the exact inversion of the very keystream words, incl. with
    widened ties resolved by the lower words (here: plain run, the tie path has its own test)."""
    e = engine_factory(64, 1024, 2)
    P = _P(e)
    g = e.gauss_create(sigma, security=security, samples=1024)
    info = e.gauss_info(g)
    assert info["words"] == words and info["bit_precision"] > 192
    tab = info["table"]
    flat = [int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big") for row in tab]
    assert all(x <= y for x, y in zip(flat, flat[1:])) and (tab[-1] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    v = S.centered(e.to_host(e.sample_gauss(e.empty(64), g, KEY, stream_id=31)), P)[:, 0]
    r = S.gaussian_words(KEY, 31, 0, 8 * 1024, words)
    assert np.array_equal(S.gaussian_from_table(r, tab, info["x_min"]), v[:8].reshape(-1))
    flatv = v.reshape(-1)
    assert abs(flatv.mean()) < 5 * sigma / np.sqrt(flatv.size) and abs(flatv.var() / sigma ** 2 - 1) < 0.05
    raw = e.gauss_noise(g, 2048, KEY, stream_id=31).cpu().numpy()
    assert np.array_equal(raw, flatv[:2048])
    e.gauss_destroy(g)


# ---- the narrow draws (NFLHIP_DIST_NARROW, nflhip_gauss_set_draw_bits(g, 32)): keystream lanes instead of 64-bit words ----------
NARROW_SHAPES = [(64, 4096, 4, 3), (64, 4, 3, 7), (32, 1024, 2, 3), (32, 8, 2, 5), (32, 16, 3, 4), (16, 128, 1, 4), (16, 16, 2, 9),
                 (16, 32, 2, 5), (16, 512, 2, 2), (32, 32768, 1, 1)]


@pytest.mark.parametrize("lb,n,m,batch", NARROW_SHAPES, ids=["u%d-n%d-m%d" % s[:3] for s in NARROW_SHAPES])
def test_narrow_uniform_is_the_reference_rule_on_keystream_lanes(lb, n, m, batch, engine_factory):
    """poly(uniform) (core.hpp:152-188: mask to floor(log2 p) + 1 bits, one conditional subtraction) applied to the limb-width
    LANE g of the keystream instead of the 64-bit word g: exact against the numpy restatement of the lanes, for rows longer
    and shorter than a keystream block (the per-thread-block kernel and the lane-by-lane one), shards and the sequence form."""
    e = engine_factory(lb, n, m)
    P, dt = _P(e), e.np_dtype
    got = e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, KEY, stream_id=1, narrow=True))
    lanes = S.uniform_narrow_words(KEY, 1, 0, batch * m * n, lb).reshape(batch, m, n)
    assert np.array_equal(got, S.uniform(lanes, P))
    assert all((got[:, cm] < P[cm]).all() for cm in range(m))
    wide = e.to_host(e.sample(e.empty(batch), DIST_UNIFORM, KEY, stream_id=1))
    assert not np.array_equal(wide, got), "its own keystream domain: the wide rule's values keep their meaning"
    # any split of the batch gives the same words
    if batch >= 2:
        cut = batch // 2
        lo = e.sample(e.empty(cut), DIST_UNIFORM, KEY, stream_id=1, first_poly=0, narrow=True)
        hi = e.sample(e.empty(batch - cut), DIST_UNIFORM, KEY, stream_id=1, first_poly=cut, narrow=True)
        assert np.array_equal(np.concatenate([e.to_host(lo), e.to_host(hi)]), got)
    # sequence form: polynomial b = a one-polynomial call with stream id first + b * stride
    if n >= 8:
        seq = e.to_host(e.sample_seq(e.empty(batch), DIST_UNIFORM, KEY, 100, 3, narrow=True))
        for b in range(batch):
            one = S.uniform_narrow_words(KEY, 100 + 3 * b, 0, m * n, lb).reshape(1, m, n)
            assert np.array_equal(seq[b:b + 1], S.uniform(one, P)), b
    # the flag belongs to the uniform rule only
    for dist, p0 in ((DIST_BOUNDED, 5), (DIST_ZO, 100), (DIST_HWT, 1)):
        with pytest.raises(NflHipError, match="narrow draw"):
            e.sample(e.empty(1), dist, KEY, param0=p0, narrow=True)
    # statistics of the lanes: every residue row uniform on [0, p)
    if batch * n >= 4096:
        for cm in range(m):
            x = got[:, cm].astype(np.float64).reshape(-1) / P[cm]
            assert abs(x.mean() - 0.5) < 5 / np.sqrt(12 * x.size) and abs(x.var() - 1 / 12) < 0.01


_NARROW_TIE_CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from nfllib_amd import Engine
from oracle import samplers as S
import ctypes
from nfllib_amd import _lib
_lib.lib.nflhip_debug_gauss_tie_shift.argtypes = [ctypes.c_int]
_lib.lib.nflhip_debug_gauss_tie_shift.restype = None
_lib.lib.nflhip_debug_gauss_tie_shift(int(sys.argv[1]))
KEY = bytes(range(32))
for lb, n, m, batch, sigma, sec, first in ((64, 1024, 2, 3, 3.2, 128, 0), (64, 64, 1, 13, 20.0, 128, 5), (32, 256, 1, 5, 3.2, 64, 2),
                                          (64, 4, 1, 9, 3.2, 128, 3), (64, 8, 2, 5, 3.2, 128, 1), (16, 128, 1, 7, 2.0, 20, 0),
                                          (64, 4096, 1, 2, 215.0, 100, 1), (64, 16, 1, 6, 3.19, 128, 0)):
    e = Engine(lb, n, m)
    P = [int(e.table(1, cm)[0]) for cm in range(m)]
    g = e.gauss_create(sigma, security=sec, samples=1024, draw_bits=32)
    info = e.gauss_info(g)
    d = e.to_host(e.sample_gauss(e.empty(batch), g, KEY, stream_id=21, first_poly=first))
    v = S.centered(d, P)[:, 0].reshape(-1)
    r = S.gaussian_words_narrow(KEY, 21, first * n, batch * n, info["words"])
    want = S.gaussian_from_table(r, info["table"], info["x_min"])
    assert np.array_equal(v, want), ("words", lb, n, m, sigma, sec, int((v != want).sum()))
    # the compact form and the raw samples read the same lanes
    amp = 2 if sigma < 10 else 1
    fmt = torch.int8 if sigma < 10 else torch.int32
    sm = e.sample_gauss_small(torch.empty((batch, n), dtype=fmt, device="cuda:0"), g, KEY, stream_id=21, amplifier=amp, first_poly=first)
    assert np.array_equal(sm.cpu().numpy().reshape(-1).astype(np.int64), amp * want), ("compact", lb, n, sigma)
    raw = e.gauss_noise(g, batch * n - 3, KEY, stream_id=21, first_sample=first * n + 1).cpu().numpy()
    assert np.array_equal(raw, want[1:batch * n - 2]), ("noise", lb, n)
    if n >= 16:   # sequence forms: one keystream per polynomial
        seq = e.to_host(e.sample_gauss_seq(e.empty(batch), g, KEY, 500, 7))
        sms = e.sample_gauss_small_seq(torch.empty((batch, n), dtype=fmt, device="cuda:0"), g, KEY, 500, 7, amplifier=amp).cpu().numpy()
        for b in range(batch):
            rb = S.gaussian_words_narrow(KEY, 500 + 7 * b, 0, n, info["words"])
            wb = S.gaussian_from_table(rb, info["table"], info["x_min"])
            assert np.array_equal(S.centered(seq[b:b + 1], P)[0, 0], wb) and np.array_equal(sms[b].astype(np.int64), amp * wb), ("seq", lb, n, b)
    e.gauss_destroy(g)
print("TIE_OK")
"""


@pytest.mark.parametrize("tie_shift", ["0", "20", "31", "56", "63"])
def test_narrow_gaussian_equals_full_precision_inversion(tie_shift):
    """The 32-bit draw: a sample is decided by the 32-bit lane g of its stream unless that lane equals the top half of a table
    entry the search meets; then the lower half (a second domain) and, on a further tie, the secondary words are read.  The
    value is exactly the oracle's full-precision inversion of (lane, lower lane, secondary words) for every table width, the
    16-per-thread kernels and the per-coefficient ones, residue words, compact polynomials, raw samples, shards and the
    sequence forms -- with the debug tie shift (never set in production) widening what counts as a tie at BOTH stages so that
    the refinement paths run (20: one 32-bit tie in 4096 comparisons; 31 / 63: nearly every comparison)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _NARROW_TIE_CHILD % {"root": root}, tie_shift], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "TIE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_narrow_gaussian_statistics_and_arguments(engine_factory):
    e = engine_factory(64, 1024, 2)
    P = _P(e)
    sigma = 3.2
    g = e.gauss_create(sigma, security=128, samples=1024, draw_bits=32)
    assert e.lib.nflhip_gauss_draw_bits(g) == 32
    v = S.centered(e.to_host(e.sample_gauss(e.empty(200), g, KEY, stream_id=9)), P)[:, 0].reshape(-1)
    assert abs(v.mean()) < 5 * sigma / np.sqrt(v.size) and abs(v.var() / sigma ** 2 - 1) < 0.02
    rh, rlo = GOLD["gauss_%g/hist" % sigma], int(GOLD["gauss_%g/lo" % sigma])
    lo, hi = min(rlo, int(v.min())), max(rlo + rh.size - 1, int(v.max()))
    a = np.bincount(v - lo, minlength=hi - lo + 1)
    b = np.zeros(hi - lo + 1, dtype=np.int64)
    b[rlo - lo:rlo - lo + rh.size] = rh
    assert _chi2_two_sample(a, b) < 1.5, "two-sample test against the real reference's samples"
    g64 = e.gauss_create(sigma, security=128, samples=1024)
    assert e.lib.nflhip_gauss_draw_bits(g64) == 64
    v64 = S.centered(e.to_host(e.sample_gauss(e.empty(200), g64, KEY, stream_id=9)), P)[:, 0].reshape(-1)
    assert not np.array_equal(v, v64), "other keystream domains: the 64-bit draw's values keep their meaning"
    assert e.lib.nflhip_gauss_set_draw_bits(g64, 16) != 0 and e.lib.nflhip_gauss_set_draw_bits(g64, 32) == 0
    assert np.array_equal(S.centered(e.to_host(e.sample_gauss(e.empty(200), g64, KEY, stream_id=9)), P)[:, 0].reshape(-1), v)
    e8 = engine_factory(64, 8, 2)
    g8 = e8.gauss_create(sigma, security=128, samples=8, draw_bits=32)
    with pytest.raises(NflHipError, match="degree >= 8"):
        e8.sample_gauss_seq(e8.empty(2), g8, KEY, 1)
    for h in (g, g64):
        e.gauss_destroy(h)
    e8.gauss_destroy(g8)


@pytest.mark.parametrize("lb,n,m,draw_bits", [(64, 4096, 4, 32), (64, 1024, 2, 32), (32, 1024, 2, 32), (64, 16, 2, 32), (64, 8, 2, 64), (64, 2048, 2, 64)],
                         ids=["u64-4096", "u64-1024", "u32-1024", "u64-16", "u64-8-wide", "u64-2048-wide"])
def test_several_compact_draws_in_one_launch_are_the_single_draws(lb, n, m, draw_bits, engine_factory):
    """nflhip_sample_gauss_small_multi_dev (what an LWE encryption's x, e0, e1 become): draw j is, byte for byte, the single
    sequence call (per-polynomial stream ids first + i * stride) -- or, without strides, the single batch call -- with draw j's
    amplifier and stream id; one to four draws, int8 / int16 / int32, also where the one-launch kernel does not apply (64-bit
    draw, degree < 16: the library loops) and at a batch that does not fill the grid evenly"""
    import torch
    e = engine_factory(lb, n, m)
    g = e.gauss_create(3.2, security=128, samples=n, draw_bits=draw_bits)
    for batch in (1, 37):
        for fmt, amps in ((torch.int8, [1, 2, 2, 1]), (torch.int16, [100, 1, 250, 7]), (torch.int32, [1 << 20, 5, 1, 12345])):
            for count in (1, 2, 3, 4):
                sids, strides = [500, 501, 77, 1 << 40][:count], [3, 3, 0, 11][:count]
                if n >= 16 or draw_bits == 64 and n >= 8:      # sequence form
                    got = e.sample_gauss_small_multi([torch.zeros((batch, n), dtype=fmt, device="cuda:0") for _ in range(count)], g, KEY, sids, strides,
                                                     amps[:count])
                    for j in range(count):
                        want = e.sample_gauss_small_seq(torch.zeros((batch, n), dtype=fmt, device="cuda:0"), g, KEY, sids[j], strides[j], amplifier=amps[j])
                        assert torch.equal(got[j], want), ("seq", batch, fmt, count, j)
                got = e.sample_gauss_small_multi([torch.zeros((batch, n), dtype=fmt, device="cuda:0") for _ in range(count)], g, KEY, sids, None, amps[:count])
                for j in range(count):
                    want = e.sample_gauss_small(torch.zeros((batch, n), dtype=fmt, device="cuda:0"), g, KEY, stream_id=sids[j], amplifier=amps[j])
                    assert torch.equal(got[j], want), ("batch", batch, fmt, count, j)
    with pytest.raises(NflHipError, match="one to four"):
        e.sample_gauss_small_multi([torch.zeros((1, n), dtype=torch.int8, device="cuda:0")] * 5, g, KEY, [1] * 5, None, [1] * 5)
    with pytest.raises(NflHipError, match="do not fit"):
        e.sample_gauss_small_multi([torch.zeros((1, n), dtype=torch.int8, device="cuda:0")] * 2, g, KEY, [1, 2], None, [1, 120])
    e.gauss_destroy(g)
