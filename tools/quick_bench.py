#!/usr/bin/env python3
"""Experiment driver (GPU box): time kernel variants of the metric kernel and
check that every variant produces the same words as variant 2 (the plain
Harvey-range arithmetic).  One subprocess per variant because the variant is
read from the NFLHIP_VARIANT environment variable when a context is created.

  python tools/quick_bench.py asm hipcc [--batch 16384] [--iters 10]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(batch, iters, workload):
    import torch
    from nfllib_amd import Engine
    from nfllib_amd.sharding import digest_words
    lb, n, nm = {"B": (64, 4096, 4), "C": (64, 16384, 8), "E": (64, 65536, 30), "A": (32, 1024, 1), "A2": (32, 1024, 2), "A2K": (32, 2048, 1), "A4K": (32, 4096, 4), "A16K": (32, 16384, 4), "R8": (64, 8192, 2), "S1": (64, 1024, 2), "S2": (64, 2048, 1), "R32": (64, 32768, 2),
                  "H": (16, 128, 1), "T8": (32, 8, 2)}[workload]
    e = Engine(lb, n, nm)
    a = e.fill_uniform(e.empty(batch), 0x4E464C6C6962, 0)
    b = e.fill_uniform(e.empty(batch), 0x4E464C6C6962, 1)
    c = e.empty(batch)
    for _ in range(3):
        e.polymul(a, b, out=c)
    torch.cuda.synchronize()
    best = min(e.time_polymul(c, a, b, iters) for _ in range(3))
    small = e.to_host(c[:64])
    fa = e.ntt_(a[:64].clone())
    rt = e.intt_(fa.clone())
    ok = not e.any_neq(rt, a[:64].contiguous())
    def rate(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return batch / (e0.elapsed_time(e1) / reps * 1e-3)
    fwd_rate, inv_rate = rate(lambda: e.ntt_(c)), rate(lambda: e.intt_(c))
    print(json.dumps({"ms": best, "polymul_per_s": batch / best * 1e3, "ntt_per_s": fwd_rate, "intt_per_s": inv_rate, "digest": digest_words(small),
                      "ntt_digest": digest_words(e.to_host(fa)), "roundtrip_ok": ok}))


def main():
    if sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--batch=")), 16384))
    iters = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--iters=")), 10))
    workload = next((a.split("=")[1] for a in sys.argv if a.startswith("--workload=")), "B")
    ref = None
    for v in args:
        env = dict(os.environ, NFLHIP_VARIANT=v)
        out = subprocess.run([sys.executable, __file__, "--child", str(batch), str(iters), workload], env=env,
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("variant", v, "FAILED", out.stderr[-400:])
            continue
        r = json.loads(line[-1])
        if ref is None:
            ref = r
        same = r["digest"] == ref["digest"] and r["ntt_digest"] == ref["ntt_digest"]
        alg = {"B": 393216, "C": 3145728, "E": 47185920, "A": 12288, "A2": 24576, "A2K": 24576, "A4K": 196608, "A16K": 786432, "R8": 393216, "H": 768, "T8": 192, "S1": 49152, "S2": 49152, "R32": 1572864}[workload]
        print("variant %4s  %8.3f ms  %10.0f polymul/s  %5.1f%% of 8TB/s  fwd %.3g/s (%.2f TB/s)  inv %.3g/s (%.2f TB/s)  same_as_first=%s roundtrip=%s" % (
            v, r["ms"], r["polymul_per_s"], r["polymul_per_s"] * alg / 8e12 * 100, r["ntt_per_s"], r["ntt_per_s"] * alg / 1.5e12,
            r["intt_per_s"], r["intt_per_s"] * alg / 1.5e12, same, r["roundtrip_ok"]))


if __name__ == "__main__":
    main()
