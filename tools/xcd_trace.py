#!/usr/bin/env python3
"""Development aid (GPU box): per-role timeline of the one-launch, XCD-pinned kernel (NFLHIP_XCD=1).

  NFLHIP_XCD=1 python tools/xcd_trace.py E 32        # workload (quick_bench names) and batch

Every role of one launch leaves {ticket | role << 28, t0 draw, t1 dependencies met, t2 done} (s_memtime, 100 MHz) in a
device buffer (nflhip_debug_xcd_trace); this prints how long roles wait and run, and how busy the workgroups were."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from nfllib_amd import Engine
    workload, batch = sys.argv[1], int(sys.argv[2])
    lb, n, nm = {"E": (64, 65536, 30), "R32": (64, 32768, 2)}[workload]
    e = Engine(lb, n, nm)
    a = e.fill_uniform(e.empty(batch), 1, 0)
    b = e.fill_uniform(e.empty(batch), 1, 1)
    c = e.empty(batch)
    for _ in range(2):
        e.polymul(a, b, out=c)
    torch.cuda.synchronize()
    buf = torch.zeros(64 * 65536 * 4, dtype=torch.int32, device="cuda")
    e.lib.nflhip_debug_xcd_trace.argtypes = [C.c_void_p]
    e.lib.nflhip_debug_xcd_trace(C.c_void_p(buf.data_ptr()))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    e.polymul(a, b, out=c)
    ev1.record()
    torch.cuda.synchronize()
    host_us = ev0.elapsed_time(ev1) * 1e3
    e.lib.nflhip_debug_xcd_trace(C.c_void_p(0))
    r = buf.cpu().numpy().view(np.uint32).reshape(64, 65536, 4)
    # s_memtime's rate is not documented for gfx950: calibrate on the launch itself (span of the busiest XCD = host time)
    spans = []
    for x in range(64):
        rec = r[x][r[x][:, 0] != 0]
        if len(rec):
            base = rec[0, 1]
            t0 = (rec[:, 1] - base).astype(np.int32).astype(np.int64)
            t2 = (rec[:, 3] - base).astype(np.int32).astype(np.int64)
            ok = (t2 > t0) & (t2 - t0 < (1 << 28))
            if ok.any():
                spans.append(np.percentile(t2[ok], 99) - np.percentile(t0[ok], 1))
    tick = host_us / float(np.median(spans))
    print("host time %.1f us, median domain span %d ticks -> %.5f us per tick" % (host_us, np.median(spans), tick))
    names = {0: "V product", 1: "F fwd(a)", 2: "F fwd(b)", 3: "I inverse"}
    t_all0, t_all1 = None, None
    for x in range(64):
        rec = r[x][r[x][:, 0] != 0]
        if not len(rec):
            continue
        role = rec[:, 0] >> 28
        t0, t1, t2 = ((rec[:, k] - rec[0, 1]).astype(np.int32).astype(np.int64) for k in (1, 2, 3))
        lo, hi = t0.min(), t2.max()
        t_all0 = lo if t_all0 is None else min(t_all0, lo)
        t_all1 = hi if t_all1 is None else max(t_all1, hi)
        line = "xcd %d: %5d roles, span %8.1f us" % (x, len(rec), (hi - lo) * tick)
        for k in (1, 0, 3):
            m = role == k if k != 1 else (role == 1) | (role == 2)
            if m.any():
                line += " | %s n=%d wait %.1f run %.1f" % ("F" if k == 1 else names[k][0], m.sum(), ((t1 - t0)[m]).mean() * tick,
                                                          ((t2 - t1)[m]).mean() * tick)
        busy = (t2 - t1).sum() * tick
        wait = (t1 - t0).sum() * tick
        line += " | busy %.0f wait %.0f WG-us" % (busy, wait)
        print(line)
    print("%.0f products/s" % (batch / (host_us * 1e-6)))


if __name__ == "__main__":
    main()
