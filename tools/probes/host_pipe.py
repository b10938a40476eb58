#!/usr/bin/env python3
"""Where the pipelined host-pointer path spends its time (nflhip_debug_host_pipe_seconds): python tools/probes/host_pipe.py [POLYS]
u64/4096/4, nflhip_polymul on pageable host arrays, five calls."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nfllib_amd import Engine

hb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e = Engine(64, 4096, 4)
d = e.fill_uniform(e.empty(hb), 3, 0)
ha = e.to_host(d)
hbb = e.to_host(e.fill_uniform(e.empty(hb), 3, 1))
hcc = e.h_polymul(ha, hbb)
e.lib.nflhip_debug_host_pipe_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
e.lib.nflhip_debug_host_pipe_seconds.restype = None
prev = (C.c_double * 4)()
e.lib.nflhip_debug_host_pipe_seconds(e.ctx, prev)
for i in range(5):
    t0 = time.perf_counter()
    e.h_polymul(ha, hbb, out=hcc)
    dt = time.perf_counter() - t0
    cur = (C.c_double * 4)()
    e.lib.nflhip_debug_host_pipe_seconds(e.ctx, cur)
    print("call %d: %.1f ms = %.0f products/s (%.1f GB/s over PCIe both ways); copy in %.1f ms, copy out %.1f ms, waiting for the device %.1f ms, inside %.1f ms"
          % (i, dt * 1e3, hb / dt, hb * 3 * 131072 / dt / 1e9, *[(cur[k] - prev[k]) * 1e3 for k in range(4)]))
    prev = cur
