#!/usr/bin/env python3
"""Time nflhip_crt_lift_dev on u64/65536/30 for whatever library is installed as nfllib_amd/libnflhip.so (phase probes of
kernels_crt_mfma.hip are built with -DNFLHIP_CRT_MFMA_PROBE=bits: wrong results, honest time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nfllib_amd import Engine
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
for batch in (16, 64):
    e = Engine(64, 65536, 30)
    a = e.fill_uniform(e.empty(batch), 1, 0)
    e.crt_lift(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e.crt_lift(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-10s batch %3d  %8.3f ms  %9.1f polys/s" % (tag, batch, dt * 1e3, batch / dt))
    e.close()
