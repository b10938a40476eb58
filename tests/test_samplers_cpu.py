"""CPU: the restated sampler rules (oracle/samplers.py) against fixtures captured from the REAL reference's random
constructors (tests/golden/samplers.npz, made by tools/gen_golden_samplers.py through the fork-replay of
oracle/ref_shim.cpp), the ChaCha20 restatement against its known-answer vector, and -- when the real reference is
available (build container) -- a fresh replay."""
import os

import numpy as np
import pytest

from nfllib_amd.params import params
from oracle import oracle as O
from oracle import samplers as S

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers.npz"))
_DT = {16: np.uint16, 32: np.uint32, 64: np.uint64}


def _shapes():
    return sorted({k.split("/")[0] for k in GOLD.files if k.startswith("u")})


def _moduli(tag):
    lb, n, m = (int(x) for x in tag[1:].split("_"))
    return lb, n, m, [int(x) for x in params(lb).P[:m]]


@pytest.mark.parametrize("tag", _shapes())
def test_rules_match_reference_fixtures(tag):
    lb, n, m, P = _moduli(tag)
    dt = _DT[lb]
    seen = 0
    for key in GOLD.files:
        if not key.startswith(tag + "/") or not key.endswith("/raw"):
            continue
        kind = key.split("/")[1]
        raw, want = GOLD[key], GOLD[key[:-3] + "out"]
        if kind == "uniform":
            got = S.uniform(raw.view(dt).reshape(m, n), P)              # core.hpp:152-188
        elif kind.startswith("bounded_"):
            ub, amp = (int(x) for x in kind.split("_")[1:])
            got = S.non_uniform(raw.view(dt), P, ub, amp, dtype=dt)     # core.hpp:195-277
        else:
            rho = int(kind.split("_")[1])
            got = S.zo_dist(raw, P, rho, canonical=False, dtype=dt)     # core.hpp:330-340 (p+1 for +1, as stored)
            can = S.zo_dist(raw, P, rho, canonical=True, dtype=dt)
            assert np.array_equal(S.centered(can[None], P), S.centered((want.astype(object) % np.array(P, dtype=object)[:, None]).astype(dt)[None], P))
        assert np.array_equal(got, want), key
        seen += 1
    assert seen >= 6


def test_chacha20_known_answer():
    # D. J. Bernstein's ChaCha20, all-zero key and nonce, block 0 (the classic test vector)
    w = S.chacha20_words(bytes(32), 0, 0, 8)
    ks = b"".join(int(v).to_bytes(8, "little") for v in w).hex()
    assert ks == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                  "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    # random access: any window equals the corresponding slice of a long run, and streams differ
    key = bytes(range(32))
    long = S.chacha20_words(key, 7, 0, 100)
    assert np.array_equal(S.chacha20_words(key, 7, 13, 50), long[13:63])
    assert not np.array_equal(S.chacha20_words(key, 8, 0, 100), long)
    # the secondary stream (block counters from 2^63) and the W-word numbers of the lazy-precision Gaussian
    sec = S.chacha20_words(key, 7, 0, 100, counter_base=S.SECONDARY_COUNTER)
    assert not np.array_equal(sec, long) and np.array_equal(S.chacha20_words(key, 7, 16, 8, counter_base=S.SECONDARY_COUNTER), sec[16:24])
    # the Gaussian sampler reads its own domain of (key, stream id): distribution tag in bits 56..62 of the block counter
    gb = S.domain_base("gauss")
    glong = S.chacha20_words(key, 7, 0, 100, counter_base=gb)
    gsec = S.chacha20_words(key, 7, 0, 100, counter_base=S.SECONDARY_COUNTER | gb)
    r = S.gaussian_words(key, 7, 5, 20, 3)
    assert np.array_equal(r[:, 0], glong[5:25]) and np.array_equal(r[:, 1:].reshape(-1), gsec[10:50])
    assert np.array_equal(S.gaussian_words(key, 7, 5, 20, 1)[:, 0], glong[5:25])
    # no two distributions share a keystream word for the same (key, stream id)
    runs = [S.chacha20_words(key, 7, 0, 64, counter_base=S.domain_base(d)) for d in S.DOMAIN]
    assert len({r_.tobytes() for r_ in runs}) == len(S.DOMAIN) and np.array_equal(runs[0], long[:64])


def test_distribution_fixtures_are_sane():
    for sigma in (3.2, 20.0):
        hist, lo = GOLD["gauss_%g/hist" % sigma], int(GOLD["gauss_%g/lo" % sigma])
        x = np.arange(lo, lo + hist.size)
        mean = (hist * x).sum() / hist.sum()
        var = (hist * (x - mean) ** 2).sum() / hist.sum()
        assert abs(mean) < 4 * sigma / np.sqrt(hist.sum()) and abs(var / sigma ** 2 - 1) < 0.02
    assert int(GOLD["hwt_64/pos_hist"].sum()) == 64 * int(GOLD["hwt_64/reps"])


@pytest.mark.skipif(not O.ref_available() or not hasattr(O.Reference(64, 64, 3).lib, "nflref_sample_replay"),
                    reason="real reference not built here")
def test_rules_match_live_reference():
    ref = O.Reference(64, 64, 3)
    P = [int(x) for x in params(64).P[:3]]
    for _ in range(3):
        out, raw = ref.sample_replay(0)
        assert np.array_equal(S.uniform(raw.view(np.uint64).reshape(3, 64), P), out)
        out, raw = ref.sample_replay(1, 37, 5)
        assert np.array_equal(S.non_uniform(raw.view(np.uint64), P, 37, 5), out)
        out, raw = ref.sample_replay(2, 0x7F)
        assert np.array_equal(S.zo_dist(raw, P, 0x7F, canonical=False), out)


@pytest.mark.skipif(not O.ref_available(), reason="needs oracle/_ref (the real reference)")
@pytest.mark.parametrize("sigma,security,center", [(3.19, 128, 0.0), (3.19, 64, 0.0), (20.0, 128, 0.0), (215.0, 100, 0.0), (4.0, 80, 2.5),
                                                   (3.19, 256, 0.0), (20.0, 300, 1.5)])   # 274 / 321 bits: 5 / 6 words per entry
def test_gaussian_table_against_the_real_references_barriers(sigma, security, center):
    """The engine's cumulative table (host arithmetic of gauss_table.cpp: floor(2^(64 W) * CDF), 448-bit fixed point) vs
    the table the REAL reference computes with MPFR at its bit precision (round(CDF * (2^bp - 1)), one rounding per
    accumulated term): same support, same bit precision, and every entry equal to within the reference's own rounding
    noise -- a few units in the LAST of its bp bits per accumulated term."""
    from nfllib_amd.engine import gauss_table
    ref = O.ref_gauss_barriers(sigma, security, 1024, center)
    if ref is None:
        pytest.skip("this prebuilt reference library predates nflref_gauss_barriers")
    bp, rounded_center, bar = ref
    t = gauss_table(sigma, security, 1024, center)
    # (the reference rounds its precision up to whole in_class words -- bytes here, FastGaussianNoise.hpp:262-272)
    assert -(-t["bit_precision"] // 8) * 8 == bp and t["entries"] == len(bar)
    assert t["x_min"] == rounded_center - (len(bar) - 1) // 2          # FastGaussianNoise.hpp:323
    W = t["words"]
    ours = [int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big") >> (64 * W - bp) for row in t["table"]]
    worst = max(abs(a - b) for a, b in zip(ours[:-1], bar[:-1]))       # (the engine's last entry is all ones by definition)
    assert worst <= len(bar) + 4, worst
    assert bar[-1] >= (1 << bp) - len(bar) - 4                         # the reference's last barrier is ~ 2^bp - 1


@pytest.mark.skipif(not O.ref_available(), reason="needs oracle/_ref (the real reference)")
@pytest.mark.parametrize("sigma,security,center", [(3.19, 128, 0.0), (20.0, 64, 0.0), (4.0, 80, 2.5)])
def test_reference_getnoise_is_the_inversion_the_engine_implements(sigma, security, center):
    """Fork-replay of the REAL getNoise: (1) its samples are exactly the inversion of its own cumulative table on the
    bytes it consumed (so the restated rule is the reference's), and (2) the engine's rule -- inversion through the
    engine's table of a W-word number -- gives the same integer for EVERY completion of those bytes (checked on the
    all-zeros and all-ones completions), i.e. both samplers compute the same function of the uniform number."""
    from nfllib_amd.engine import gauss_table
    rep = O.ref_gauss_replay(sigma, security, 1024, center, 4096)
    ref = O.ref_gauss_barriers(sigma, security, 1024, center)
    if rep is None or ref is None:
        pytest.skip("this prebuilt reference library predates the gaussian entry points")
    got, raw, call_words = rep
    bp, rounded_center, bar = ref
    wp = bp // 8
    bars = [b.to_bytes(wp, "big") for b in bar]
    x0 = rounded_center - (len(bar) - 1) // 2
    want, prefixes = S.gaussian_reference_decode(raw, call_words, bars, wp, got.size, x0)
    assert np.array_equal(got, want), "the reference's samples are not the inversion of its table on these bytes"
    assert {len(p) for p in prefixes} >= {1, 2}            # both short paths were taken (the full one is rare)
    t = gauss_table(sigma, security, 1024, center)
    W = t["words"]
    for pad in (b"\x00", b"\xff"):
        r = np.array([[int.from_bytes((p + pad * (8 * W - len(p)))[8 * k:8 * k + 8], "big") for k in range(W)] for p in prefixes],
                     dtype=np.uint64)
        assert np.array_equal(S.gaussian_from_table(r, t["table"], t["x_min"]), got)


def test_gaussian_fixtures_from_the_real_reference():
    """The same two checks on committed fixtures (tests/golden/gauss_replay.npz, tools/gen_golden_gauss.py), so they run
    on hosts without the reference: table against the real barriers, rule against a real getNoise replay."""
    from nfllib_amd.engine import gauss_table
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gauss_replay.npz"))
    for k, (sigma, security, center) in enumerate(G["sets"]):
        bp, rounded_center, call_words = (int(x) for x in G["%d/meta" % k])
        bars = [bytes(r) for r in G["%d/barriers" % k]]
        wp, nb = bp // 8, len(bars)
        got = G["%d/out" % k]
        x0 = rounded_center - (nb - 1) // 2
        want, prefixes = S.gaussian_reference_decode(G["%d/raw" % k], call_words, bars, wp, got.size, x0)
        assert np.array_equal(got, want)
        t = gauss_table(float(sigma), int(security), 1024, float(center))
        W = t["words"]
        assert -(-t["bit_precision"] // 8) * 8 == bp and t["entries"] == nb and t["x_min"] == x0
        ours = [int.from_bytes(b"".join(int(v).to_bytes(8, "big") for v in row), "big") >> (64 * W - bp) for row in t["table"]]
        assert max(abs(a - int.from_bytes(b, "big")) for a, b in zip(ours[:-1], bars[:-1])) <= nb + 4
        for pad in (b"\x00", b"\xff"):
            r = np.array([[int.from_bytes((p + pad * (8 * W - len(p)))[8 * j:8 * j + 8], "big") for j in range(W)] for p in prefixes],
                         dtype=np.uint64)
            assert np.array_equal(S.gaussian_from_table(r, t["table"], t["x_min"]), got)


# ---- the narrow draws (oracle/samplers.py chacha20_lanes / uniform_narrow_words / gaussian_words_narrow) -----------------------
KEY = bytes((7 * i + 3) & 0xFF for i in range(32))


def test_keystream_lanes_are_the_little_endian_bytes_of_the_words():
    w = S.chacha20_words(KEY, 5, 3, 40, counter_base=S.domain_base("uniform_narrow"))
    raw = w.tobytes()
    for lane_bytes, dt in ((1, np.uint8), (2, np.uint16), (4, np.uint32), (8, np.uint64)):
        per = 8 // lane_bytes
        lanes = S.chacha20_lanes(KEY, 5, 3 * per, 40 * per, lane_bytes, counter_base=S.domain_base("uniform_narrow"))
        assert lanes.tobytes() == raw and lanes.dtype == dt
        # any window, aligned or not
        part = S.chacha20_lanes(KEY, 5, 3 * per + 5, 17, lane_bytes, counter_base=S.domain_base("uniform_narrow"))
        assert np.array_equal(part, lanes[5:22])


@pytest.mark.parametrize("tag", _shapes())
def test_the_narrow_uniform_rule_is_the_references_rule_on_its_own_byte_stream(tag):
    """The reference's poly(uniform) fills _data with fastrandombytes and reduces every limb-width word in place (core.hpp:152-188):
    residue word g is made from bytes [g w, (g + 1) w) of the stream.  That IS the narrow rule; the fixture captured from the real
    reference (raw bytes, resulting polynomial) pins S.uniform on exactly that reading of the bytes."""
    lb, n, m, P = _moduli(tag)
    key = next((k for k in GOLD.files if k.startswith(tag + "/uniform") and k.endswith("/raw")), None)
    if key is None:
        pytest.skip("no uniform fixture for this shape")
    raw, want = GOLD[key], GOLD[key[:-3] + "out"]
    lanes = np.frombuffer(raw.tobytes(), dtype={16: "<u2", 32: "<u4", 64: "<u8"}[lb]).reshape(m, n)
    assert np.array_equal(S.uniform(lanes, P), want.reshape(m, n))
    # and the device's narrow keystream is read the same way: lane g = bytes [g w, (g + 1) w)
    words = S.uniform_narrow_words(KEY, 9, 0, m * n, lb)
    stream = S.chacha20_words(KEY, 9, 0, (m * n * lb // 8 + 7) // 8, counter_base=S.domain_base("uniform_narrow")).tobytes()
    assert words.tobytes() == stream[:m * n * lb // 8]


@pytest.mark.parametrize("sigma,security", [(3.19, 128), (20.0, 128), (2.0, 20)])
def test_the_narrow_gaussian_draw_has_the_tables_distribution(sigma, security):
    """(32-bit lane, 32-bit lower lane, secondary words) is a uniform W-word number, so its inversion through the table has the
    table's distribution: chi-square against the tail-cut discrete Gaussian, and agreement with the wide draw's statistics.
    The table is the engine's host-built one (pinned against the real reference's MPFR table above)."""
    from nfllib_amd.engine import gauss_table
    info = gauss_table(sigma, security, 1024)
    N = 1 << 16
    r = S.gaussian_words_narrow(KEY, 77, 0, N, info["words"])
    assert r.shape == (N, info["words"]) and r.dtype == np.uint64
    x = S.gaussian_from_table(r, info["table"], info["x_min"])
    pmf = S.gaussian_pmf(sigma, 0.0, info["x_min"], info["entries"])
    obs = np.bincount(x - info["x_min"], minlength=info["entries"]).astype(np.float64)
    exp = pmf * N
    keep = exp >= 8
    stat = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum() + (obs[~keep].sum() - exp[~keep].sum()) ** 2 / max(exp[~keep].sum(), 1e-9)
    dof = int(keep.sum())
    assert stat < dof + 6 * np.sqrt(2 * dof), (stat, dof)
    assert abs(x.mean()) < 5 * sigma / np.sqrt(N) and abs(x.var() / sigma ** 2 - 1) < 0.03
    # first words: top half = lane g of domain 7, lower half = lane g of domain 8 -- independent streams
    hi = S.chacha20_lanes(KEY, 77, 0, N, 4, counter_base=S.domain_base("gauss32")).astype(np.uint64)
    lo = S.chacha20_lanes(KEY, 77, 0, N, 4, counter_base=S.domain_base("gauss32_ref")).astype(np.uint64)
    assert np.array_equal(r[:, 0], (hi << np.uint64(32)) | lo) and not np.array_equal(hi, lo)
