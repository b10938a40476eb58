#!/usr/bin/env python3
"""DEVELOPMENT AID (CPU only): the persistent one-launch plan on the gfx950 interpreter of tests/asm_emu.py under many
random workgroup interleavings and scheduler shapes -- more orders than tests/test_asm_emulated.py keeps in the suite.
Every run must equal the oracle and must not trip the interpreter's strict checks (waits, hazards, LDS races, cache
visibility) or its stuck detector.   usage: tools/xcd_emu_stress.py [runs] [first_seed]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import asm_emu                                 # noqa: E402
from nfllib_amd.params import params           # noqa: E402
from oracle import oracle as O                 # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prm = params(64)
n = 32768
bad = 0
for run in range(runs):
    rnd = random.Random(seed0 + run)
    pooled = rnd.random() < 0.5
    dlog = rnd.choice([0, 0, 1])
    rlog = rnd.choice([1, 2]) if pooled else rnd.choice([1, 2, 3])
    if pooled and rlog + dlog > 3:
        rlog = 1
    nm = rnd.choice([1, 1, 2, 3])
    rows_min = 8 << dlog
    batch = max(2, -(-rows_min // nm)) + rnd.randrange(0, 3)
    wgs = rnd.choice([w for w in (8, 9, 16, 24, 40, 64) if w >= (8 << dlog)])   # (every domain needs a workgroup)
    order = rnd.choice(["random", "bursty", "lowest", "highest"])
    o = O.Oracle(64, n, nm, prm)
    rng = np.random.default_rng(seed0 + run)
    P = np.asarray(prm.P[:nm], dtype=np.uint64)
    a = rng.integers(0, 1 << 62, size=(batch, nm, n), dtype=np.uint64) % P[None, :, None]
    b = rng.integers(0, 1 << 62, size=(batch, nm, n), dtype=np.uint64) % P[None, :, None]
    state = {"cur": None, "left": 0}

    def pick(live):
        if order == "random":
            return rnd.choice(live)
        if order == "lowest":
            return live[0] if rnd.random() < 0.9 else rnd.choice(live)
        if order == "highest":
            return live[-1] if rnd.random() < 0.9 else rnd.choice(live)
        if state["left"] > 0 and state["cur"] in live:      # bursty: one workgroup runs for a while, then another
            state["left"] -= 1
            return state["cur"]
        state["cur"], state["left"] = rnd.choice(live), rnd.randrange(1, 200)
        return state["cur"]

    stem = "polymul_xcd32768l" if pooled else "polymul_xcd32768"
    t = time.time()
    try:
        got = asm_emu.run_xcd_product(os.path.join(ROOT, "nfllib_amd", "csrc", stem + "_gfx950.s"), n, nm, prm, a, b, dlog, rlog,
                                      int(pooled), wgs, pick)
        ok = bool(np.array_equal(got, o.polymul(a, b)))
        msg = "ok" if ok else "WRONG RESULT"
    except RuntimeError as e:
        ok, msg = False, "%s: %s" % (type(e).__name__, e)
    bad += not ok
    print("seed %d: %s batch %d x %d moduli, D=2^%d R=2^%d, %d workgroups, order %s: %s (%.0f s)"
          % (seed0 + run, stem, batch, nm, dlog, rlog, wgs, order, msg, time.time() - t), flush=True)
sys.exit(1 if bad else 0)
