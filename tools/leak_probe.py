#!/usr/bin/env python3
"""Context life-cycle probe (GPU box): create / use / destroy contexts in a loop and watch free device memory."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from nfllib_amd import Engine

free0 = None
for it in range(60):
    for lb, n, m in ((64, 4096, 4), (64, 65536, 2), (32, 1024, 2), (16, 128, 1)):
        e = Engine(lb, n, m)
        a = e.fill_uniform(e.empty(4), 1, 0)
        b = e.fill_uniform(e.empty(4), 1, 1)
        c = e.polymul(a, b)
        e.h_polymul(e.to_host(a), e.to_host(b))
        g = e.gauss_create(3.2)
        e.sample_gauss(c, g, bytes(32))
        e.gauss_destroy(g)
        e.crt_project(e.crt_lift(a))
        del a, b, c
        e.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if it == 4:
        free0 = free
    if it % 10 == 9:
        print("iteration %d: free %.1f MiB" % (it + 1, free / 2**20), flush=True)
drift = (free0 - free) / 2**20
print("drift since iteration 5: %.1f MiB" % drift)
sys.exit(0 if drift < 64 else 1)
