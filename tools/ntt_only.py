#!/usr/bin/env python3
"""Stand-alone transform loop for profiling: python tools/ntt_only.py [fwd|inv|pntt] [batch]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from nfllib_amd import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
e = Engine(64, 4096, 4)
a = e.fill_uniform(e.empty(batch), 1, 0)
b = e.ntt_(e.fill_uniform(e.empty(batch), 1, 1))
c = e.empty(batch)
fn = {"fwd": lambda: e.ntt_(a), "inv": lambda: e.intt_(a), "pntt": lambda: e.polymul(a, b, out=c, b_is_ntt=True)}[kind]
for _ in range(3):
    fn()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    fn()
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 10
print("%s: %.3f ms per %d polys = %.2f M/s" % (kind, ms, batch, batch / ms / 1e3))
