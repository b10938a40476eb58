#!/usr/bin/env python3
"""One CRT lift of u64/65536/30 x batch (stamp builds of kernels_crt_mfma.hip print their per-phase cycle counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nfllib_amd import Engine
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = Engine(64, 65536, 30)
a = e.fill_uniform(e.empty(batch), 1, 0)
for _ in range(2):
    e.crt_lift(a); torch.cuda.synchronize()
e.close()
