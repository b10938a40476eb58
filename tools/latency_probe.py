#!/usr/bin/env python3
"""Latency probe (GPU box): single-polynomial host-pointer calls vs resident calls, and the raw cost of small
pageable hipMemcpyAsync sequences on this runtime."""
import ctypes as C
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from nfllib_amd import Engine, OP_ADD


def raw():
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    hip.hipStreamCreateWithFlags(C.byref(st), 1)
    for size in (4096, 65536, 131072, 1 << 20):
        d = [C.c_void_p() for _ in range(3)]
        for x in d:
            hip.hipMalloc(C.byref(x), C.c_size_t(size))
        h = [np.zeros(size, dtype=np.uint8) for _ in range(3)]
        hp = [x.ctypes.data_as(C.c_void_p) for x in h]

        def seq(nin, out):
            for i in range(nin):
                hip.hipMemcpyAsync(d[i], hp[i], C.c_size_t(size), 1, st)
            if out:
                hip.hipMemcpyAsync(hp[2], d[2], C.c_size_t(size), 2, st)
            hip.hipStreamSynchronize(st)
        for nin, out in ((1, False), (2, False), (1, True), (2, True)):
            seq(nin, out)
            t0 = time.perf_counter()
            for _ in range(200):
                seq(nin, out)
            print("raw %8d B  %d x H2D %s: %.1f us" % (size, nin, "+ D2H" if out else "      ", (time.perf_counter() - t0) / 200 * 1e6), flush=True)


def main():
    raw()
    for lb, n, m in ((64, 4096, 4), (32, 1024, 1), (64, 16384, 8)):
        e = Engine(lb, n, m)
        a = e.to_host(e.fill_uniform(e.empty(1), 1, 0))
        b = e.to_host(e.fill_uniform(e.empty(1), 1, 1))
        for name, fn in (("h_ntt", lambda: e.h_ntt(a)), ("h_add", lambda: e.h_pointwise(OP_ADD, a, b)), ("h_polymul", lambda: e.h_polymul(a, b))):
            fn()
            t0 = time.perf_counter()
            for _ in range(300):
                fn()
            print(lb, n, m, name, "%.1f us per single-poly host call" % ((time.perf_counter() - t0) / 300 * 1e6), flush=True)
        da, db, c = e.to_device(a), e.to_device(b), e.empty(1)
        e.polymul(da, db, out=c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            e.polymul(da, db, out=c)
            torch.cuda.synchronize()
        print(lb, n, m, "resident polymul + sync %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6), flush=True)


if __name__ == "__main__":
    main()
