#!/usr/bin/env python3
"""Round 5, VERDICT item 3: what would workload E (u64/65536/30) run at if NONE of the pipeline's passes reached HBM?
Needs the library built with -DNFLHIP_ABLATION_KNOBS (build/abl, tools/sessions/gpu_round5_a.sh): NFLHIP_PIPE_CHUNKS_RT sets
the chunk count, NFLHIP_ABLATE_PIPE_ALIAS lays every chunk over chunk 0's memory (wrong results by construction; the
window of a, b, c and the two scratch rows of one small chunk fits the 256 MiB Infinity Cache).
One process per configuration (the alias knob is read once)."""
import json
import os
import subprocess
import sys


def child(batch, iters):
    import torch
    from nfllib_amd import Engine
    e = Engine(64, 65536, 30)
    a = e.fill_uniform(e.empty(batch), 1, 0)
    b = e.fill_uniform(e.empty(batch), 1, 1)
    c = e.empty(batch)
    for _ in range(3):
        e.polymul(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        e.polymul(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"ms": ms, "polymul_per_s": batch / ms * 1e3, "frac": batch / ms * 1e3 * 47185920 / 8e12}))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    batch = int(sys.argv[1])
    for chunks in (4, 8, 16, 32, 64):
        if chunks * 2 > batch:
            continue
        for alias in (False, True):
            env = dict(os.environ, NFLHIP_PIPE_CHUNKS_RT=str(chunks), NFLHIP_XCD="0")
            if alias:
                env["NFLHIP_ABLATE_PIPE_ALIAS"] = "1"
            r = subprocess.run([sys.executable, __file__, "--child", str(batch), "8"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            window = batch // chunks * 5 * 30 * 65536 * 8 / 2**20
            print("batch %d chunks %2d (%d polys, window of one chunk's a b c s0 s1 = %4.0f MiB) %s: %s" % (
                batch, chunks, batch // chunks, window, "ALIASED " if alias else "shipped ", line[0] if line else r.stderr[-300:]), flush=True)
