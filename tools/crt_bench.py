#!/usr/bin/env python3
"""Time CRT lift / project (GMP::poly2mpz / mpz2poly replacements) on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfllib_amd import Engine

for lb, n, nm, batch in [(64, 4096, 4, 4096), (64, 16384, 8, 256), (64, 65536, 30, 16)]:
    e = Engine(lb, n, nm)
    a = e.fill_uniform(e.empty(batch), 1, 0)
    limbs = e.crt_lift(a)
    back = e.crt_project(limbs)
    assert not e.any_neq(a, back)
    torch.cuda.synchronize()
    for name, fn in (("lift", lambda: e.crt_lift(a)), ("project", lambda: e.crt_project(limbs))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        L = e.crt_limbs
        alg = batch * (nm * n * 8 + n * L * 8)
        print("u%d/%d/%d batch %d  %-8s %8.3f ms  %10.1f polys/s  %6.1f GB/s algorithmic (L=%d)" % (lb, n, nm, batch, name, dt * 1e3, batch / dt, alg / dt / 1e9, L))
    e.close()
