#!/usr/bin/env python3
"""The reference's LWE-like symmetric encryption demo (tests/nfllib_demo_main_op.cpp:26-58, 260-332) run end to end on
the GPU over a resident batch: key generation, `batch` encryptions of zero (three Gaussian polynomials each, three
forward NTTs, two multiply-adds), decryption (one multiply-subtract, one inverse NTT) and the demo's own correctness
check (every decrypted coefficient is even and small, i.e. the parity rule of decrypt() gives 0).

Two plans behind the same arithmetic (identical ciphertexts for identical keystreams):
  --plan fused    (default) the transform-fused pipelines: three compact (int8) Gaussian sampler launches + ONE
                  nflhip_fwd_fma2_dev per encryption batch, ONE nflhip_fma_inv_dev per decryption batch; the
                  transformed polynomials never reach HBM
  --plan unfused  operator by operator, as the reference's code reads: 8 launches / ~17 polynomial passes per
                  encryption batch, 2 launches / 5 passes per decryption batch

    python tools/lwe_demo.py [--degree 4096] [--nmoduli 4] [--batch 4096] [--sigma 3.19] [--plan fused] [--traffic]

--traffic adds the HBM bytes per encryption / decryption measured by rocprofv3 counter passes of this same command
(FETCH_SIZE x 2 + WRITE_SIZE, separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) next to the bytes
the plan has to move (inputs read + outputs written).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import numpy as np
    import torch
    from nfllib_amd import DIST_UNIFORM, Engine
    from nfllib_amd._lib import FMT_I8
    EXPR_ADD, EXPR_SUB, EXPR_MUL = 0x10, 0x11, 0x12                 # NFLHIP_EXPR_* (include/nflhip.h)

    e = Engine(getattr(args, "limb_bits", 64), args.degree, args.nmoduli)
    if args.grid:
        from nfllib_amd import _lib
        _lib.lib.nflhip_debug_fused_grid(args.grid)
    key = bytes(range(1, 33)) if args.fixed_key else os.urandom(32)
    B = args.batch
    # FastGaussianNoise(SIGMA, 128, 1<<10), line 271; draw_bits 32 = the narrow draw (what include/nfl_hip/nfl.hpp's generator sets)
    draw_bits = getattr(args, "draw_bits", 32)
    g = e.gauss_create(args.sigma, 128, 1 << 10, draw_bits=draw_bits if args.degree >= 16 else 64)
    narrow = draw_bits == 32
    sid = [0]

    def stream():
        sid[0] += 1
        return sid[0]

    # secret key s (NTT form), public key (pka uniform, pkb = 2e + pka*s): ONE polynomial each, shared by the batch
    s = e.ntt_(e.sample_gauss(e.empty(1), g, key, stream()))
    pka = e.sample(e.empty(1), DIST_UNIFORM, key, stream(), narrow=narrow)
    pkb = e.ntt_(e.sample_gauss(e.empty(1), g, key, stream(), amplifier=2))
    pkb = e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_ADD]), [pkb, pka, s])                  # pkb + pka*s   (lines 281-283)
    resa, resb, dec = e.empty(B), e.empty(B), e.empty(B)
    fused = args.plan == "fused"
    if fused:
        u, e1, e2 = (e.empty_small(B, FMT_I8) for _ in range(3))

        def encrypt():
            if getattr(args, "multi_draw", True) and hasattr(e, "sample_gauss_small_multi"):
                # u, 2*e_1, 2*e_2: one launch, byte for byte the three single draws (nflhip_sample_gauss_small_multi_dev, ABI 6)
                e.sample_gauss_small_multi([u, e1, e2], g, key, [stream(), stream(), stream()], amplifiers=[1, 2, 2])
            else:
                e.sample_gauss_small(u, g, key, stream())                                  # u
                e.sample_gauss_small(e1, g, key, stream(), amplifier=2)                    # 2*e_1
                e.sample_gauss_small(e2, g, key, stream(), amplifier=2)                    # 2*e_2
            e.fwd_fma2(u, pka, e1, pkb, e2, out0=resa, out1=resb)                      # resa = u*pka + 2e_1, resb = u*pkb + 2e_2

        def decrypt():
            e.fma_inv(resa, s, resb, subtract=True, out=dec)                           # INTT(resb - resa*s)
        enc_bytes = 3 * args.degree * 2 + 2 * e.bytes_per_poly                          # compact noise written + read, two results
    else:
        S, PKA, PKB = (t.expand(B, -1, -1).contiguous() for t in (s, pka, pkb))
        u, e1, e2 = e.empty(B), e.empty(B), e.empty(B)

        def encrypt():
            e.ntt_(e.sample_gauss(u, g, key, stream()))
            e.ntt_(e.sample_gauss(e1, g, key, stream(), amplifier=2))
            e.ntt_(e.sample_gauss(e2, g, key, stream(), amplifier=2))
            e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [u, PKA, e1], out=resa)
            e.eval(bytes([0, 1, EXPR_MUL, 2, EXPR_ADD]), [u, PKB, e2], out=resb)

        def decrypt():
            e.eval(bytes([0, 1, 2, EXPR_MUL, EXPR_SUB]), [resb, resa, S], out=dec)
            e.intt_(dec)
        enc_bytes = 2 * e.bytes_per_poly                                                 # what MUST move: the two results
    dec_bytes = 3 * e.bytes_per_poly                                                     # resa, resb read; the plaintext written

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
    P = e.P
    v = h[:, 0, :].astype(object)
    bits = np.where(v < P[0] // 2, v % 2, 1 - v % 2)
    ok = bool((bits == 0).all())
    noise = np.where(v < P[0] // 2, v, v - P[0]).astype(np.float64)
    out = {"demo": "LWE-like symmetric encryption of 0 (tests/nfllib_demo_main_op.cpp)", "plan": args.plan, "grid": args.grid, "limb_bits": getattr(args, "limb_bits", 64), "degree": args.degree,
           "nmoduli": args.nmoduli, "batch": B, "draw_bits": draw_bits, "encrypt_us_per_ciphertext": round(t_enc / B * 1e6, 4),
           "decrypt_us_per_ciphertext": round(t_dec / B * 1e6, 4), "encryptions_per_s": round(B / t_enc, 1),
           "decryptions_per_s": round(B / t_dec, 1), "decrypts_to_zero": ok,
           "noise_rms": round(float(np.sqrt((noise ** 2).mean())), 1),
           "compulsory_bytes": {"encrypt": enc_bytes, "decrypt": dec_bytes},
           "digest": {"resa": e.digest(resa[:8].contiguous()), "dec": e.digest(dec[:8].contiguous())} if args.fixed_key else None}
    e.gauss_destroy(g)
    return out, ok


def measure_traffic(args, launches):
    """HBM bytes per ciphertext from two rocprofv3 counter passes over this command (child run, --reps as given)"""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tot = {}
    for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):   # KiB units; FETCH_SIZE reports half the bytes on gfx950
        d = tempfile.mkdtemp(prefix="lwe_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--degree", str(args.degree), "--nmoduli", str(args.nmoduli), "--batch", str(args.batch), "--plan", args.plan,
                   "--reps", str(args.reps), "--fixed-key"]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            if r.returncode != 0:
                return {"error": "rocprofv3 --pmc %s failed: %s" % (counter, r.stderr[-300:])}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    name = row["Kernel_Name"]
                    k = tot.setdefault(name.split("(")[0][:60], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
                    k[counter] += float(row["Counter_Value"]) * scale
                    if counter == "FETCH_SIZE":
                        k["n"] += 1
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--limb-bits", type=int, default=64, choices=(16, 32, 64), dest="limb_bits")
    ap.add_argument("--degree", type=int, default=4096)
    ap.add_argument("--nmoduli", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--sigma", type=float, default=3.19)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--plan", choices=("fused", "unfused"), default="fused")
    ap.add_argument("--draw-bits", type=int, default=32, choices=(32, 64), dest="draw_bits",
                    help="keystream bits a Gaussian sample consumes: 32 = the narrow draw (and narrow uniform lanes), 64 = one word per value")
    ap.add_argument("--traffic", action="store_true")
    ap.add_argument("--fixed-key", action="store_true", help="a fixed sampler key (reproducible digests)")
    ap.add_argument("--single-draws", action="store_false", dest="multi_draw", help="fused plan: three sampler launches per encryption instead of one")
    ap.add_argument("--grid", type=int, default=0, help="experiment: 1 / 2 force the 2-D / the XCD-dealt 1-D grid of the fused kernels (nflhip_debug_fused_grid)")
    args = ap.parse_args()
    out, ok = run(args)
    if args.traffic:
        tot = measure_traffic(args, args.reps + 1)
        if "error" in tot:
            out["traffic"] = tot
        else:
            per = args.batch * (args.reps + 1)     # the child runs every phase reps + 1 times
            enc = sum(v["FETCH_SIZE"] + v["WRITE_SIZE"] for k, v in tot.items()
                      if any(p in k for p in ("gauss", "fused_enc2", "fused_fma_fwd", "ntt_fwd", "k_eval")))
            dec = sum(v["FETCH_SIZE"] + v["WRITE_SIZE"] for k, v in tot.items() if any(p in k for p in ("fms_inv", "ntt_inv")))
            # (the unfused plan's two kinds of k_eval launches: 2 of 3 belong to encrypt)
            if args.plan == "unfused":
                ev = sum(v["FETCH_SIZE"] + v["WRITE_SIZE"] for k, v in tot.items() if "k_eval" in k)
                enc -= ev / 3.0
                dec += ev / 3.0
            cb = out["compulsory_bytes"]
            out["traffic"] = {"encrypt_bytes_per_ciphertext": round(enc / per, 1), "decrypt_bytes_per_ciphertext": round(dec / per, 1),
                              "encrypt_ratio": round(enc / per / cb["encrypt"], 3), "decrypt_ratio": round(dec / per / cb["decrypt"], 3),
                              "kernels": {k: {"launches": v["n"], "bytes_per_launch": round((v["FETCH_SIZE"] + v["WRITE_SIZE"]) / max(v["n"], 1))}
                                          for k, v in sorted(tot.items())},
                              "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / --pmc WRITE_SIZE, separate passes over this command"}
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
