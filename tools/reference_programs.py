#!/usr/bin/env python3
"""What an UNCHANGED caller of the reference gains: the reference's own timing programs (tests/nfllib_demo_main_op.cpp,
nfllib_demo_main_func.cpp per CONFIG, tests/ntt_perfs.cpp) built twice from the SAME sources --
  cpu : against the real reference (tests/reftests/Makefile `cpu` -> oracle/_ref/progs/, NFL_OPTIMIZED + NTT_AVX2), on this host's CPU
  gpu : against this repository's drop-in headers + libnflhip.so (tests/reftests/Makefile -> tests/_reftests/), on the MI355X
-- run side by side on the GPU box; their "Time per ...: V us" lines are paired into one table.
The programs time per-polynomial calls on nfl::poly (the reference's literal host-array type), so every device call carries
its PCIe round trip: this is the floor of the drop-in, not the resident rate (nfl::poly_p).

usage: tools/reference_programs.py [--json OUT.json] [--reps N] > table.txt      (test infrastructure: it runs oracle/_ref binaries)"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_DIR = os.path.join(ROOT, "tests", "_reftests")
CPU_DIR = os.path.join(ROOT, "oracle", "_ref", "progs")
LINE = re.compile(r"^Time per (.+?):\s*([0-9.eE+-]+)\s*us\s*$")


def programs():
    names = sorted(f for f in os.listdir(CPU_DIR) if not f.startswith("_") and os.access(os.path.join(CPU_DIR, f), os.X_OK)) if os.path.isdir(CPU_DIR) else []
    return [n for n in names if os.path.exists(os.path.join(GPU_DIR, n))]


def run(path, reps):
    """median over `reps` runs of every 'Time per X: V us' line -> ({op: us}, wall seconds of one run)"""
    vals, wall = {}, None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([path], capture_output=True, text=True, timeout=1800)
        w = time.perf_counter() - t0
        wall = w if wall is None else min(wall, w)
        if r.returncode != 0:
            raise RuntimeError("%s exited %d: %s" % (path, r.returncode, (r.stdout + r.stderr)[-300:]))
        seen = {}
        for ln in r.stdout.splitlines():
            m = LINE.match(ln.strip())
            if m:
                key = m.group(1)
                k, i = key, 2
                while k in seen:            # (a program may print one label twice)
                    k, i = "%s #%d" % (key, i), i + 1
                seen[k] = float(m.group(2))
        for k, v in seen.items():
            vals.setdefault(k, []).append(v)
    return {k: sorted(v)[len(v) // 2] for k, v in vals.items()}, wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="", help="substring filter on program names")
    args = ap.parse_args()
    progs = [p for p in programs() if args.only in p]
    if not progs:
        raise SystemExit("no program pairs found (build: make -C tests/reftests all cpu)")
    rows, out = [], {}
    for p in progs:
        cpu, wc = run(os.path.join(CPU_DIR, p), args.reps)
        gpu, wg = run(os.path.join(GPU_DIR, p), args.reps)
        out[p] = {"cpu_wall_s": round(wc, 3), "gpu_wall_s": round(wg, 3), "ops": {}}
        for op in cpu:
            if op in gpu:
                ratio = cpu[op] / gpu[op] if gpu[op] > 0 else float("inf")
                out[p]["ops"][op] = {"cpu_us": cpu[op], "gpu_us": gpu[op], "cpu_over_gpu": round(ratio, 3)}
                rows.append((p, op, cpu[op], gpu[op], ratio))
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except Exception:
        cpu_model = "unknown"
    print("# the reference's own timing programs, unchanged sources: real NFLlib on one core of %s (cpu) vs the drop-in header +" % cpu_model)
    print("# libnflhip.so on the MI355X (gpu); per-polynomial nfl::poly calls (host arrays: every device call includes its PCIe round trip);")
    print("# median of %d runs; ratio > 1: the unchanged program is faster on the device" % args.reps)
    print("%-44s %-58s %12s %12s %8s" % ("program", "operation (Time per ...)", "cpu [us]", "gpu [us]", "cpu/gpu"))
    for p, op, c, g, r in rows:
        print("%-44s %-58s %12.3f %12.3f %8.2f" % (p, op, c, g, r))
    slower = [(p, op) for p, op, c, g, r in rows if r < 1.0]
    print("# %d of %d operations are slower through nfl::poly on the device than on the CPU (small shapes / cheap element-wise"
          " operations: launch + PCIe per call); resident nfl::poly_p / device_batch is the fast path (INTEGRATION.md)" % (len(slower), len(rows)))
    if args.json:
        summary = {"programs": len(progs), "operations": len(rows), "slower_on_device": len(slower), "cpu": cpu_model,
                   "geomean_cpu_over_gpu": round(float(__import__("math").exp(sum(__import__("math").log(max(r, 1e-9)) for *_, r in rows) / len(rows))), 3)}
        with open(args.json, "w") as f:
            json.dump({"summary": summary, "programs": out}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
