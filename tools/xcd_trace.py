#!/usr/bin/env python3
"""Where do the workgroups of the one-launch plan spend their time?  Runs ONE product of u64/DEGREE/NMODULI x BATCH on the persistent
kernel (NFLHIP_XCD=1) with the role trace switched on (include/nflhip_debug.h nflhip_debug_xcd_trace) and prints, per kind of role
(forward streaming / block product / inverse streaming): roles, time running (t2 - t1), time between a workgroup becoming free and its
next role's inputs being ready (t1 - t0: hand-out + waiting for dependencies), as shares of workgroup-time.
usage: NFLHIP_XCD=1 python tools/xcd_trace.py [DEGREE NMODULI BATCH [LEVEL]]       (s_memtime ticks; only ratios are used)"""
import os
import sys

os.environ.setdefault("NFLHIP_XCD", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nfllib_amd import Engine

n, nm, batch = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 30, 128)
level = int(sys.argv[4]) if len(sys.argv) > 4 else 2
e = Engine(64, n, nm)
e.lib.nflhip_debug_polymul_level(level)
a = e.fill_uniform(e.empty(batch), 1, 0)
b = e.fill_uniform(e.empty(batch), 1, 1)
c = e.empty(batch)
for _ in range(3):
    e.polymul(a, b, out=c)
torch.cuda.synchronize()
trace = torch.zeros(32 * 65536 * 4, dtype=torch.int32, device="cuda")
e.lib.nflhip_debug_xcd_trace.argtypes = [__import__("ctypes").c_void_p]
e.lib.nflhip_debug_xcd_trace(trace.data_ptr())
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
e.polymul(a, b, out=c)
ev1.record()
torch.cuda.synchronize()
e.lib.nflhip_debug_xcd_trace(None)
ms = ev0.elapsed_time(ev1)
raw = trace.cpu().numpy().view(np.uint32).reshape(-1, 4)
live = raw[:, 3] != 0
dom = (np.arange(raw.shape[0]) >> 16)[live]           # record (domain << 16 | sequence), domain = xcd + 8 sub
rec = raw[live]
kind = (rec[:, 0] >> 28).astype(np.int64)
t0, t1, t2 = (rec[:, k].astype(np.int64) for k in (1, 2, 3))
# s_memtime is NOT one clock per XCD (the first version of this tool assumed so and printed spans four times the launch): only
# differences inside one workgroup mean anything.  A workgroup's records chain: the stamp that closes a role (t2) is the very
# stamp that opens the wait for the next one (t0), so the records of a domain fall into one chain per workgroup.
wg = np.full(len(rec), -1, dtype=np.int64)
nwg = 0
start, life = {}, {}
for d in np.unique(dom):
    idx = np.nonzero(dom == d)[0]
    by_t0 = {}
    for i in idx:
        by_t0.setdefault(int(t0[i]), []).append(i)
    closes = set(int(t2[i]) for i in idx)
    for i in idx:
        if int(t0[i]) in closes or wg[i] >= 0:
            continue                                   # not the first role of a workgroup
        j = i
        start[nwg] = int(t0[i])
        while True:
            wg[j] = nwg
            life[nwg] = (int(t2[j]) - start[nwg]) & 0xFFFFFFFF
            nxt = [k for k in by_t0.get(int(t2[j]), []) if wg[k] < 0]
            if not nxt:
                break
            j = nxt[0]
        nwg += 1
if (wg < 0).any():                                     # two workgroups of a domain with the same stamp: rare; drop what could not be chained
    print("# %d records not chained (dropped)" % int((wg < 0).sum()))
    keep = wg >= 0
    rec, kind, t0, t1, t2, wg, dom = rec[keep], kind[keep], t0[keep], t1[keep], t2[keep], wg[keep], dom[keep]
run, wait = (t2 - t1) & 0xFFFFFFFF, (t1 - t0) & 0xFFFFFFFF
span = np.array([life[w] for w in range(nwg)], dtype=np.int64)
wrun = np.bincount(wg, weights=run, minlength=nwg)
wwait = np.bincount(wg, weights=wait, minlength=nwg)
roles_per = np.bincount(wg, minlength=nwg)
names = {0: "block product", 1: "forward streaming (a, b)", 2: "forward streaming", 3: "inverse streaming"}   # xcd.py take(kind, ...)
print("# one product of u64/%d/%d x %d on the one-launch plan, block products on %s transforms" % (n, nm, batch, "incomplete" if level == 2 else "complete"))
print("# launch %.3f ms by HIP events; %d roles traced, chained into %d workgroups (%.0f roles each, min %d max %d)"
      % (ms, len(rec), nwg, roles_per.mean(), roles_per.min(), roles_per.max()))
smax = float(span.max())
print("# a workgroup's life (first 'free' stamp -> last role done), ticks: min %.0f  median %.0f  max %.0f;  if the longest one spans the whole"
      " launch the clock is %.2f GHz" % (span.min(), np.median(span), smax, smax / ms / 1e6))
us = ms * 1e3 / smax                                   # microseconds per tick, upper bound (longest life <= launch)
print("%-28s %8s %14s %14s %12s %12s" % ("kind", "roles", "run mean [us]", "wait mean [us]", "run share", "wait share"))
total = smax * nwg                                     # workgroup-time of the launch, lower bound
for k in sorted(set(kind.tolist())):
    m = kind == k
    print("%-28s %8d %14.2f %14.2f %11.1f%% %11.1f%%" % (names.get(k, "kind %d" % k), int(m.sum()), run[m].mean() * us, wait[m].mean() * us,
                                                         100.0 * run[m].sum() / total, 100.0 * wait[m].sum() / total))
print("%-28s %8d %14s %14s %11.1f%% %11.1f%%" % ("all", len(rec), "", "", 100.0 * run.sum() / total, 100.0 * wait.sum() / total))
print("# shares are of (longest life x workgroups).  wait = workgroup free -> its next role's inputs ready (hand-out: two atomics + a poll; and"
      " dependencies).  The rest (%.1f%%) is life the other workgroups do not have: they end before the longest one does (drain)."
      % (100.0 * (1 - (run.sum() + wait.sum()) / total)))
print("# lives as a share of the longest: " + " ".join("p%d=%.2f" % (q, np.percentile(span, q) / smax) for q in (0, 5, 25, 50, 75, 95)))
# inside a life: how the wait share moves from its first to its last fifth
parts = np.zeros((5, 2))
for i in range(len(rec)):
    w = int(wg[i])
    pos = ((int(t1[i]) - start[w]) & 0xFFFFFFFF) / max(1.0, float(span[w]))
    q = min(4, int(pos * 5))
    parts[q, 0] += run[i]; parts[q, 1] += wait[i]
print("# wait / (run + wait) by fifth of a workgroup's life: " + " ".join("%.3f" % (parts[q, 1] / max(1.0, parts[q].sum())) for q in range(5)))
