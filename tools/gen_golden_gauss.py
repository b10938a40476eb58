#!/usr/bin/env python3
"""Generate tests/golden/gauss_replay.npz from the REAL reference (build container only): for a few parameter sets, the
cumulative table FastGaussianNoise builds (oracle/ref_gauss_shim.cpp: nflref_gauss_barriers) and one fork-replay of
getNoise (nflref_gauss_replay): the uniform bytes of its fastrandombytes() calls and the samples it made of them.

    python tools/gen_golden_gauss.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SETS = [(3.19, 128, 0.0), (20.0, 64, 0.0), (4.0, 80, 2.5)]   # sigma, security, center (samples = 1024)
RLEN = 2048


def main():
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libnflref.so missing: run `make -C oracle` in the build container")
    arrays = {"sets": np.array(SETS, dtype=np.float64)}
    for k, (sigma, security, center) in enumerate(SETS):
        bp, rc, bar = O.ref_gauss_barriers(sigma, int(security), 1024, center)
        out, raw, call_words = O.ref_gauss_replay(sigma, int(security), 1024, center, RLEN)
        wp = bp // 8
        arrays["%d/barriers" % k] = np.frombuffer(b"".join(b.to_bytes(wp, "big") for b in bar), dtype=np.uint8).reshape(len(bar), wp)
        arrays["%d/meta" % k] = np.array([bp, rc, call_words], dtype=np.int64)
        arrays["%d/raw" % k] = raw
        arrays["%d/out" % k] = out
    path = os.path.join(ROOT, "tests", "golden", "gauss_replay.npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s: %d arrays, %d bytes" % (path, len(arrays), os.path.getsize(path)))


if __name__ == "__main__":
    main()
