#!/usr/bin/env python3
"""Soak of the matrix-core CRT kernels against the CPU checker: every modulus count they serve (lift 21..31, projection 17..32),
every input width of the projection (5..32 words), random residues plus lifted values with all-ones / all-zero digit runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nfllib_amd import Engine
from nfllib_amd.params import params
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t0 = time.time()
checked = 0
for m in range(17, 33):
    n, batch = 256, 3
    o, e = O.Oracle(64, n, m, params(64)), Engine(64, n, m)
    Q = o.crt_modulus()
    a = o.fill_uniform(batch, int(rng.integers(1, 2**40)), 0)
    adv = [int(rng.integers(0, 2**62)) << int(rng.integers(0, 64 * o.crt_limbs)) for _ in range(60)]
    adv += [(1 << int(rng.integers(1, Q.bit_length()))) - 1 for _ in range(60)]
    adv += [Q - 1 - ((1 << int(rng.integers(1, Q.bit_length() - 1))) - 1) for _ in range(60)]
    adv += [((1 << 512) - 1) << (32 * int(rng.integers(0, 2 * o.crt_limbs - 16))) for _ in range(40)]
    adv = [x % Q for x in adv][:n]
    for idx, x in enumerate(adv):
        a[0, :, idx] = [x % int(p) for p in o.P[:m]]
    limbs = e.crt_lift(e.to_device(a))
    got = e.to_host(limbs).view(np.uint64)
    assert np.array_equal(got, o.crt_lift(a)), "lift, %d moduli" % m
    for idx, x in enumerate(adv):
        assert int.from_bytes(got[0, idx].tobytes(), "little") == x
    assert np.array_equal(e.to_host(e.crt_project(limbs)), a), "round trip, %d moduli" % m
    for lin in range(5, 33):
        wide = rng.integers(0, 2**63, size=(1, n, lin), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(1, n, lin), dtype=np.uint64)
        wide[0, 0, :] = np.uint64(0xFFFFFFFFFFFFFFFF)
        wide[0, 1, :] = 0
        wide[0, 2, :] = np.uint64(0x8080808080808080)
        wide[0, 3, :] = np.uint64(0x7F7F7F7F7F7F7F7F)
        dw = torch.from_numpy(wide.view(np.int64)).cuda()
        assert np.array_equal(e.to_host(e.crt_project(dw)), o.crt_project(wide)), "project, %d moduli, %d words" % (m, lin)
        checked += 1
    e.close()
print("crt soak ok: moduli 17..32, %d projection shapes, %.1f s" % (checked, time.time() - t0))
