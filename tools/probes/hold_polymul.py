#!/usr/bin/env python3
"""Hold one ring's batched product for a few seconds and report its rate, the package power and the shader clock beside it
(rocm-smi sampled while the launches are queued): `python tools/probes/hold_polymul.py DEGREE NMODULI BATCH [SECONDS]`.
PYTHONPATH selects the library (ablated builds live under build/abl_*)."""
import json
import os
import subprocess
import sys
import time

import torch
from nfllib_amd import Engine

n, nm, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
LB = int(os.environ.get("NFL_LIMB_BITS", "64"))     # probe knob: 64 (default) / 32 / 16-bit limbs
e = Engine(LB, n, nm)
if os.environ.get("NFL_POLYMUL_LEVEL"):      # probe knob (not read by the library): nflhip_debug_polymul_level, include/nflhip_debug.h
    e.lib.nflhip_debug_polymul_level(int(os.environ["NFL_POLYMUL_LEVEL"]))
a = e.fill_uniform(e.empty(batch), 1, 0)
b = e.fill_uniform(e.empty(batch), 1, 1)
c = e.empty(batch)
for _ in range(3):
    e.polymul(a, b, out=c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    e.polymul(a, b, out=c)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
iters = max(4, int(secs * 1e3 / ms))
e0.record()
for _ in range(iters):
    e.polymul(a, b, out=c)
e1.record()
samples = []
t_end = time.perf_counter() + secs * 0.9
while time.perf_counter() < t_end:
    txt = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
    js = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
    card = next(iter(js.values()))
    watts = next((float(v) for k, v in card.items() if "Power (W)" in k and "Max" not in k), None)
    sclk = next((v for k, v in card.items() if k.startswith("sclk clock speed")), "")
    mhz = int("".join(ch for ch in sclk if ch.isdigit()) or 0)
    if watts:
        samples.append((watts, mhz))
torch.cuda.synchronize()
ms_held = e0.elapsed_time(e1) / iters
busy = [s for s in samples[1:] if s[0] > 0.6 * max(x[0] for x in samples)] or samples
alg = 3 * nm * n * (LB // 8)
print(json.dumps({"n": n, "nm": nm, "batch": batch, "first_ms": round(ms, 4), "held_ms": round(ms_held, 4), "polymul_per_s": round(batch / ms_held * 1e3, 1),
                  "frac": round(batch / ms_held * 1e3 * alg / 8e12, 4), "package_W": round(sum(s[0] for s in busy) / len(busy), 1),
                  "sclk_MHz": round(sum(s[1] for s in busy) / len(busy)), "samples": len(busy)}))
