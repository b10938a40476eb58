#!/usr/bin/env python3
"""Determinism soak: repeat the product many times per shape and require every result to equal the first one
(a missing wait or barrier in an exchange shows up as a rare mismatch).  python tools/soak.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfllib_amd import Engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for lb, n, m, batch in ((64, 4096, 4, 2048), (64, 8192, 2, 1024), (64, 16384, 8, 128), (64, 65536, 6, 16), (32, 1024, 2, 8192),
                        (64, 1024, 2, 4096), (64, 32768, 2, 64), (64, 32768, 2, 300), (64, 16384, 6, 257), (64, 8192, 3, 513), (64, 2048, 1, 4097),
                        (32, 2048, 1, 8191), (32, 4096, 2, 2049)):   # (32768 x 2 x 64: the one-launch plan; x 300: the register-resident rows)
    e = Engine(lb, n, m)
    a = e.fill_uniform(e.empty(batch), 11, 0)
    b = e.fill_uniform(e.empty(batch), 11, 1)
    ref = e.polymul(a, b)
    fa = e.ntt_(a.clone())
    torch.cuda.synchronize()
    mism = 0
    for it in range(iters):
        c = e.polymul(a, b)
        if it % 3 == 0:
            f2 = e.ntt_(a.clone())
            mism += int(e.any_neq(f2, fa))
            mism += int(e.any_neq(e.intt_(f2), a))
        mism += int(e.any_neq(c, ref))
    print("u%d/%d/%d batch %d: %d iterations, %d mismatches" % (lb, n, m, batch, iters, mism))
    bad += mism
    e.close()

# round 5: the wave-per-row fused forward kernels (LDS-staged compact rows) and the narrow-draw samplers, same question
KEY = bytes(range(32))
for lb, n, m, batch in ((32, 1024, 2, 4099), (64, 1024, 2, 2051), (64, 2048, 3, 1025), (32, 2048, 2, 2049), (32, 4096, 2, 513)):
    e = Engine(lb, n, m)
    g = e.gauss_create(3.19, 128, n, draw_bits=32)
    x, e0, e1 = (e.sample_gauss_small(e.empty_small(batch), g, KEY, stream_id=i + 1) for i in range(3))
    k0 = e.sample(e.empty(1), 0, KEY, stream_id=9, narrow=True)
    k1 = e.sample(e.empty(1), 0, KEY, stream_id=10, narrow=True)
    r0, r1 = e.fwd_fma2(x, k0, e0, k1, e1)
    r0, r1 = r0.clone(), r1.clone()
    torch.cuda.synchronize()
    mism = 0
    for it in range(iters):
        o0, o1 = e.fwd_fma2(x, k0, e0, k1, e1)
        mism += int(e.any_neq(o0, r0)) + int(e.any_neq(o1, r1))
        if it % 4 == 0:
            x2 = e.sample_gauss_small(e.empty_small(batch), g, KEY, stream_id=1)
            mism += int(not torch.equal(x2, x))
            u2 = e.sample(e.empty(1), 0, KEY, stream_id=9, narrow=True)
            mism += int(e.any_neq(u2, k0))
    print("u%d/%d/%d batch %d fused forward rows + narrow samplers: %d iterations, %d mismatches" % (lb, n, m, batch, iters, mism))
    bad += mism
    e.gauss_destroy(g)
    e.close()
# round 6, third session: the fused inverse pipeline on those rows (generated kernels at both limb widths) and the 32768-word composed
# product at a batch that takes the register-resident pair
for lb, n, m, batch in ((32, 1024, 2, 4099), (32, 4096, 2, 513), (64, 2048, 2, 1025)):
    e = Engine(lb, n, m)
    a = e.fill_uniform(e.empty(batch), 21, 0)
    b = e.fill_uniform(e.empty(batch), 21, 1)
    k = e.fill_uniform(e.empty(1), 22, 0)
    r = e.fma_inv(a, k, b, subtract=True).clone()
    torch.cuda.synchronize()
    mism = 0
    for it in range(iters):
        mism += int(e.any_neq(e.fma_inv(a, k, b, subtract=True), r))
    print("u%d/%d/%d batch %d fused inverse rows: %d iterations, %d mismatches" % (lb, n, m, batch, iters, mism))
    bad += mism
    e.close()
print("SOAK", "CLEAN" if bad == 0 else "MISMATCHES: %d" % bad)
sys.exit(1 if bad else 0)
