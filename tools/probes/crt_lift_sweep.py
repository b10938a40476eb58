#!/usr/bin/env python3
"""CRT lift rate against the modulus count (u64, n = 4096, 4 Mi coefficients per launch) for the installed library:
where does the matrix-core kernel (kernels_crt_mfma.hip) overtake the VALU kernels (kernels_crt.hip)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nfllib_amd import Engine
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
for nm in (10, 12, 14, 16, 18, 20, 21, 22, 24, 30, 31, 32):
    n, batch = 4096, 1024
    e = Engine(64, n, nm)
    a = e.fill_uniform(e.empty(batch), 1, 0)
    e.crt_lift(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        e.crt_lift(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    L = e.crt_limbs
    limbs = e.crt_lift(a)
    e.crt_project(limbs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        e.crt_project(limbs)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t0) / 10
    print("%-8s nm %2d L %2d  lift %8.3f ms  %7.1f GB/s   project %8.3f ms  %7.1f GB/s (algorithmic)" % (
        tag, nm, L, dt * 1e3, batch * n * 8 * (nm + L) / dt / 1e9, dp * 1e3, batch * n * 8 * (nm + L) / dp / 1e9))
    e.close()
