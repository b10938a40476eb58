#!/usr/bin/env python3
"""Sampler rates (coefficients per second) per limb width and keystream rule, and the compact Gaussian sampler:
`python tools/probes/sampler_rates.py [quick]` -- one JSON line per measurement."""
import json
import sys

import torch
from nfllib_amd import Engine

KEY = bytes(range(32))
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def rate(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for lb, n, nm, batch in ((64, 4096, 4, 8192), (32, 1024, 2, 1 << 17), (16, 128, 1, 1 << 21)):
    e = Engine(lb, n, nm)
    d = e.empty(batch)
    words = batch * nm * n
    for narrow in ((False, True) if hasattr(e, "NARROW") else (False,)):
        kw = {"narrow": True} if narrow else {}
        t = rate(lambda: e.sample(d, 0, KEY, stream_id=1, **kw), 3 if quick else 10)
        print(json.dumps({"what": "sample_uniform", "limb_bits": lb, "n": n, "nm": nm, "batch": batch, "narrow": narrow,
                          "G_residue_words_per_s": round(words / t / 1e9, 1), "GBs": round(words * lb / 8 / t / 1e9, 1)}))
    for bits in ((64, 32) if hasattr(e, "NARROW") else (64,)):
        g = e.gauss_create(3.19, 128, n) if bits == 64 else e.gauss_create(3.19, 128, n, draw_bits=32)
        sm = e.empty_small(batch)
        t = rate(lambda: e.sample_gauss_small_seq(sm, g, KEY, 1), 3 if quick else 10)
        t2 = rate(lambda: e.sample_gauss(d, g, KEY, stream_id=2), 3 if quick else 10)
        print(json.dumps({"what": "gaussian", "limb_bits": lb, "n": n, "nm": nm, "batch": batch, "draw_bits": bits,
                          "compact_G_coefficients_per_s": round(batch * n / t / 1e9, 1), "words_G_coefficients_per_s": round(batch * n / t2 / 1e9, 1)}))
        e.gauss_destroy(g)
    e.close()
