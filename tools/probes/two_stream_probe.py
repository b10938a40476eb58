#!/usr/bin/env python3
"""Round 5: do two INDEPENDENT n = 65536 pipelines on two streams fill each other's launch tails / fill / drain phases?
Two contexts (each with its own scratch), half the batch each, launched alternately on two streams, against one context
with the whole batch.  `python tools/probes/two_stream_probe.py DEGREE NMODULI BATCH`"""
import json
import sys
import time

import torch
from nfllib_amd import Engine

n, nm, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40


def one():
    e = Engine(64, n, nm)
    a = e.fill_uniform(e.empty(batch), 1, 0); b = e.fill_uniform(e.empty(batch), 1, 1); c = e.empty(batch)
    for _ in range(3):
        e.polymul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        e.polymul(a, b, out=c)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e.close()
    return batch * iters / dt


def two(parts=2):
    es = [Engine(64, n, nm) for _ in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]
    h = batch // parts
    data = []
    for e in es:
        data.append((e.fill_uniform(e.empty(h), 1, 0), e.fill_uniform(e.empty(h), 1, 1), e.empty(h)))
    torch.cuda.synchronize()
    for _ in range(3):
        for e, s, (a, b, c) in zip(es, ss, data):
            e.polymul(a, b, out=c, stream=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for e, s, (a, b, c) in zip(es, ss, data):
            e.polymul(a, b, out=c, stream=s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in es:
        e.close()
    return h * parts * iters / dt


r1 = one()
r2 = two(2)
r3 = two(3) if batch % 3 == 0 else None
r1b = one()
print(json.dumps({"n": n, "nm": nm, "batch": batch, "one_stream": round(r1, 1), "two_streams": round(r2, 1), "three_streams": r3 and round(r3, 1),
                  "one_stream_again": round(r1b, 1), "gain": round(r2 / max(r1, r1b), 4)}))
