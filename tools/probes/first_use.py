#!/usr/bin/env python3
"""First-use costs of the host-pointer entry points (what an unchanged nfl::poly caller pays the first time it touches an operation):
context creation, then first / second / tenth call of every operation family on ONE polynomial.
usage: python tools/probes/first_use.py [LIMB_BITS DEGREE NMODULI]     (GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

torch.cuda.init()
torch.zeros(1, device="cuda:0")
torch.cuda.synchronize()
from nfllib_amd import Engine

lb, n, nm = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 8192, 2)
t0 = time.perf_counter()
e = Engine(lb, n, nm)
t_ctx = time.perf_counter() - t0
rng = np.random.default_rng(1)
P = np.asarray(e.P[:nm], dtype=np.uint64)
a = (rng.integers(0, 1 << 62, size=(1, nm, n), dtype=np.uint64) % P[None, :, None]).astype(e.np_dtype)
b = (rng.integers(0, 1 << 62, size=(1, nm, n), dtype=np.uint64) % P[None, :, None]).astype(e.np_dtype)
print("u%d/%d/%d: context creation %.1f ms" % (lb, n, nm, t_ctx * 1e3))
ops = [("c = a + b", lambda: e.h_pointwise(0, a, b)), ("c = a - b", lambda: e.h_pointwise(1, a, b)), ("c = a * b", lambda: e.h_pointwise(2, a, b)),
       ("NTT", lambda: e.h_ntt(a)), ("inverse NTT", lambda: e.h_intt(a)), ("polymul", lambda: e.h_polymul(a, b)),
       ("a == b", lambda: e.h_any_eq(a, b)), ("CRT lift", lambda: e.h_crt_lift(a))]
for name, fn in ops:
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    print("%-14s first %9.1f us   second %8.1f us   median of 3..10 %8.1f us" % (name, ts[0], ts[1], sorted(ts[2:])[4]))
t0 = time.perf_counter()
e2 = Engine(lb, n, nm)
print("a second context of the same shape: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
