import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from nfllib_amd import Engine
for nm, n, batch in ((30, 4096, 1024), (30, 65536, 16)):
    e = Engine(64, n, nm)
    for lin in (30, 32, 33, 60, 62, 64, 65):
        w = torch.randint(-2**62, 2**62, (batch, n, lin), dtype=torch.int64, device="cuda")
        e.crt_project(w); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): e.crt_project(w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("nm %d n %d batch %d L_in %2d: %.3f ms  %.1f GB/s (in + out)" % (nm, n, batch, lin, dt * 1e3, batch * n * 8 * (lin + nm) / dt / 1e9))
    e.close()
