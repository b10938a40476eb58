#!/usr/bin/env python3
"""Generate tests/golden/samplers.npz from the REAL reference's random constructors (build container only).

* rule fixtures: for uniform / non_uniform / ZO_dist, pairs (raw bytes the constructor consumed, the polynomial it
  built) captured by oracle/ref_shim.cpp `nflref_sample_replay` -- they pin oracle/samplers.py, the CPU restatement
  the device samplers are tested against;
* distribution fixtures: histograms of many reference samples of hwt_dist and gaussian (whose procedures are not a
  function of one replayable byte string), for two-sample tests of the device samplers.

    python tools/gen_golden_samplers.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nfllib_amd.params import params  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import samplers as S  # noqa: E402

RULE_SHAPES = [(64, 64, 3), (32, 8, 2), (16, 128, 1), (64, 1024, 2)]
BOUNDED = [(1, 1), (2, 1), (5, 3), (1000, 1), (1 << 13, 1)]
RHOS = [0x7F, 0, 255, 10]


def main():
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libnflref.so missing: run `make -C oracle` in the build container")
    arrays = {}
    for lb, n, m in RULE_SHAPES:
        ref = O.Reference(lb, n, m)
        tag = "u%d_%d_%d" % (lb, n, m)
        out, raw = ref.sample_replay(0)
        arrays[tag + "/uniform/raw"], arrays[tag + "/uniform/out"] = raw, out
        pmin = min(int(x) for x in params(lb).P[:m])
        for ub, amp in BOUNDED:
            if ub * amp >= pmin // 2:
                continue
            out, raw = ref.sample_replay(1, ub, amp)
            arrays["%s/bounded_%d_%d/raw" % (tag, ub, amp)], arrays["%s/bounded_%d_%d/out" % (tag, ub, amp)] = raw, out
        for rho in RHOS:
            out, raw = ref.sample_replay(2, rho)
            arrays["%s/zo_%d/raw" % (tag, rho)], arrays["%s/zo_%d/out" % (tag, rho)] = raw, out
    # distribution fixtures on u64/1024/2
    ref = O.Reference(64, 1024, 2)
    P = [int(x) for x in params(64).P[:2]]
    for sigma, reps in ((3.2, 200), (20.0, 200)):
        vals = []
        for _ in range(reps):
            out, _ = ref.sample_replay(4, 128, 1, sigma)
            c = S.centered(out[None], P)[0]
            assert np.array_equal(c[0], c[1])
            vals.append(c[0])
        v = np.concatenate(vals)
        lo, hi = int(v.min()), int(v.max())
        arrays["gauss_%g/lo" % sigma] = np.int64(lo)
        arrays["gauss_%g/hist" % sigma] = np.bincount(v - lo, minlength=hi - lo + 1).astype(np.int64)
    h, reps = 64, 400
    pos_hist = np.zeros(1024, dtype=np.int64)
    plus = 0
    for _ in range(reps):
        out, _ = ref.sample_replay(3, h)
        nz = out[0] != 0
        assert int(nz.sum()) == h and np.array_equal(nz, out[1] != 0)
        pos_hist += nz
        plus += int((out[0] == P[0] + 1).sum())          # the reference stores +1 as p+1 (core.hpp:387)
    arrays["hwt_64/pos_hist"], arrays["hwt_64/plus"], arrays["hwt_64/reps"] = pos_hist, np.int64(plus), np.int64(reps)
    path = os.path.join(ROOT, "tests", "golden", "samplers.npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s: %d arrays, %d bytes" % (path, len(arrays), os.path.getsize(path)))


if __name__ == "__main__":
    main()
