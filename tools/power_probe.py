#!/usr/bin/env python3
"""Sustained product loop for power / clock sampling (GPU box): python tools/power_probe.py [B|A|C|F|G] [seconds]
Prints the product rate of every half second; sample `rocm-smi -P -c` beside it.  No parity check: the ablated libraries of
tools/sessions/build_ablations.sh compute wrong words on purpose."""
import sys
import time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch
from nfllib_amd import Engine

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
lb, n, nm, batch = {"B": (64, 4096, 4, 16384), "A": (32, 1024, 1, 1 << 19), "C": (64, 16384, 6, 1024), "F": (64, 32768, 2, 1024),
                    "G": (64, 8192, 2, 8192)}[wl]
e = Engine(lb, n, nm)
a = e.fill_uniform(e.empty(batch), 1, 0)
b = e.fill_uniform(e.empty(batch), 1, 1)
c = e.empty(batch)
for _ in range(3):
    e.polymul(a, b, out=c)
torch.cuda.synchronize()
t0 = time.time()
rates = []
while time.time() - t0 < seconds:
    ms = e.time_polymul(c, a, b, 20)
    rates.append(batch / ms * 1e3)
k = max(1, len(rates) // 12)
print("%s sustained %.1f s: products/s over time: %s" % (wl, seconds, " ".join("%.3g" % (sum(rates[i:i + k]) / len(rates[i:i + k])) for i in range(0, len(rates), k))))
print("%s last third mean: %.4g products/s, %.4f ms per launch of %d" % (wl, sum(rates[-len(rates) // 3:]) / len(rates[-len(rates) // 3:]), batch / (sum(rates[-len(rates) // 3:]) / len(rates[-len(rates) // 3:])) * 1e3, batch))
