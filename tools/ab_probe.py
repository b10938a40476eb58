#!/usr/bin/env python3
"""Same-box A/B of the metric kernel between two builds of the library, through the few C entry points both have
(raw ctypes, no version check): python tools/ab_probe.py LIB.so [seconds [level]]"""
import ctypes as C
import sys
import time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from nfllib_amd.params import params as limb_params

lib = C.CDLL(sys.argv[1])
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
level = int(sys.argv[3]) if len(sys.argv) > 3 else -1     # nflhip_debug_polymul_level: 0 complete transforms, 1 / 2 incomplete
if level >= 0:
    lib.nflhip_debug_polymul_level(level)
n, nm, batch = 4096, 4, 16384
pr = limb_params(64)
import numpy as np
keep = [np.ascontiguousarray(x[:nm]) for x in (pr.P, pr.primitive_roots, pr.invkmax)]
vp = lambda a: C.c_void_p(a.ctypes.data)
h = C.c_void_p()
lib.nflhip_ctx_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
assert lib.nflhip_ctx_create(C.byref(h), 0, 64, n, nm, *[vp(x) for x in keep], pr.kmax_log2) == 0
bufs = [torch.empty(batch * nm * n, dtype=torch.int64, device="cuda") for _ in range(3)]
lib.nflhip_fill_uniform_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int, C.c_void_p]
for i in (1, 2):
    assert lib.nflhip_fill_uniform_dev(h, C.c_void_p(bufs[i].data_ptr()), 0, batch, 1, i - 1, None) == 0
lib.nflhip_time_polymul_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
ms = C.c_float(0)
ptr = [C.c_void_p(b.data_ptr()) for b in bufs]
torch.cuda.synchronize()
t0, out = time.time(), []
while time.time() - t0 < seconds:
    assert lib.nflhip_time_polymul_dev(h, ptr[0], ptr[1], ptr[2], batch, 20, None, C.byref(ms)) == 0
    out.append(ms.value)
third = out[-len(out) // 3:]
print("%s%s: %.4f ms per launch of %d (last third of %.0f s; first %.4f), checksum %d" % (sys.argv[1], "" if level < 0 else " level %d" % level, sum(third) / len(third), batch, seconds, out[0], int(bufs[0][:4096].sum().item()) & 0xffffffff))
