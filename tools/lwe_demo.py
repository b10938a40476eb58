#!/usr/bin/env python3
"""The reference's LWE-like symmetric encryption demo (tests/nfllib_demo_main_op.cpp:26-58, 260-332) run end to end on
the GPU over a resident batch: key generation, `batch` encryptions of zero (three Gaussian polynomials each, three
forward NTTs, two fused multiply-adds), decryption (one fused a - b*s, one inverse NTT) and the demo's own
correctness check (every decrypted coefficient is even and small, i.e. the parity rule of decrypt() gives 0).

    python tools/lwe_demo.py [--degree 4096] [--nmoduli 4] [--batch 4096] [--sigma 3.19]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=4096)
    ap.add_argument("--nmoduli", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=3.19)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    from nfllib_amd import DIST_UNIFORM, Engine
    EXPR_ADD, EXPR_SUB, EXPR_MUL = 0x10, 0x11, 0x12                 # NFLHIP_EXPR_* (include/nflhip.h)

    e = Engine(64, args.degree, args.nmoduli)
    key = os.urandom(32)
    B = args.batch
    g = e.gauss_create(args.sigma, 128, 1 << 10)                    # FastGaussianNoise(SIGMA, 128, 1<<10), line 271
    sid = [0]

    def stream():
        sid[0] += 1
        return sid[0]

    # secret key s (NTT form), public key (pka uniform, pkb = 2e + pka*s), one key replicated over the batch
    s = e.ntt_(e.sample_gauss(e.empty(1), g, key, stream()))
    pka = e.sample(e.empty(1), DIST_UNIFORM, key, stream())
    pkb = e.ntt_(e.sample_gauss(e.empty(1), g, key, stream(), amplifier=2))
    pkb = e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_ADD]), [pkb, pka, s])                  # pkb + pka*s   (lines 281-283)
    S, PKA, PKB = (t.expand(B, -1, -1).contiguous() for t in (s, pka, pkb))
    u, e1, e2 = e.empty(B), e.empty(B), e.empty(B)
    resa, resb, dec = e.empty(B), e.empty(B), e.empty(B)

    def encrypt():
        e.ntt_(e.sample_gauss(u, g, key, stream()))                                    # u
        e.ntt_(e.sample_gauss(e1, g, key, stream(), amplifier=2))                      # 2*e_1
        e.ntt_(e.sample_gauss(e2, g, key, stream(), amplifier=2))                      # 2*e_2
        e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [u, PKA, e1], out=resa)           # resa = u*pka + 2e_1
        e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [u, PKB, e2], out=resb)           # resb = u*pkb + 2e_2

    def decrypt():
        e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_SUB]), [resb, resa, S], out=dec)         # resb - resa*s
        e.intt_(dec)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps

    t_enc, t_dec = timed(encrypt), timed(decrypt)
    # the demo's check: decrypt() maps v -> (v < p/2) ? v % 2 : 1 - v % 2 and the sum over ciphertexts must be 0
    h = e.to_host(dec[:64])
    from nfllib_amd.params import params
    P = [int(x) for x in params(64).P[:args.nmoduli]]
    v = h[:, 0, :].astype(object)
    bits = np.where(v < P[0] // 2, v % 2, 1 - v % 2)
    ok = bool((bits == 0).all())
    noise = np.where(v < P[0] // 2, v, v - P[0]).astype(np.float64)
    print(json.dumps({"demo": "LWE-like symmetric encryption of 0 (tests/nfllib_demo_main_op.cpp)", "degree": args.degree,
                      "nmoduli": args.nmoduli, "batch": B, "encrypt_us_per_ciphertext": round(t_enc / B * 1e6, 4),
                      "decrypt_us_per_ciphertext": round(t_dec / B * 1e6, 4), "encryptions_per_s": round(B / t_enc, 1),
                      "decryptions_per_s": round(B / t_dec, 1), "decrypts_to_zero": ok,
                      "noise_rms": round(float(np.sqrt((noise ** 2).mean())), 1)}))
    e.gauss_destroy(g)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
