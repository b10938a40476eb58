#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric: poly-mults/sec (NTT + pointwise + INTT), n = 4096, 4 x 62-bit moduli.
One "step" = one pass of the hot path over one batch of synthetic input:
    c = INTT( NTT(a) (.) NTT(b) )   for every poly of the per-GPU batch
i.e. the reference sequence a.ntt_pow_phi(); b.ntt_pow_phi(); c = a*b;
c.invntt_pow_invphi() (poly.hpp:167-168, 350) as ONE fused HIP kernel.
Inputs are generated on the device (seeded counter stream) and are resident in
HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
N>1 is launched by torch.distributed.run, one rank per GPU: batch split with no
data-path collective (weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4E464C6C6962
WORKLOADS = {
    # name: (limb_bits, degree, nmoduli, default per-GPU batch)
    "B": (64, 4096, 4, 16384),   # BASELINE.json configs[1] -- the metric is quoted on this
    "D": (64, 4096, 4, 1 << 17), # configs[3]: B at 2^20 polys over 8 GPUs = 2^17 per GPU (3 x 16 GiB resident per GPU)
    "C": (64, 16384, 8, 1024),   # configs[2]
    "E": (64, 65536, 30, 128),   # configs[4] (3 x 2 GiB resident + 4 GiB of scratch; 32 polynomials leave the pipeline's fill and drain visible: 0.140 / 0.153 / 0.162 at 32 / 64 / 128, profiles/r04_E_chunks.txt)
    "A": (32, 1024, 1, 1 << 19), # configs[0]'s shape (30-bit moduli) on the device -- secondary
    "F": (64, 32768, 2, 512),    # the reference's own largest test config (32768, 124, uint64_t): tests/CMakeLists.txt:19-48
    "G": (64, 8192, 2, 8192),    # ... and (8192, 124, uint64_t)
    "H": (16, 128, 1, 1 << 22),  # ... (128, 14, uint16_t)
    "T": (32, 8, 2, 1 << 23),    # ... (8, 60, uint32_t)
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(limb_bits, degree, nmoduli, budget_s=12.0, check=None):
    """Time the CPU path on this host on a bounded sample of the same workload (median of 5 repetitions,
    BASELINE.md section 3).  `check` = (a, b, c) host copies of a few polynomials of the timed batch: the checker also
    recomputes those products and reports whether the GPU's words are identical (`parity_sample_ok`).
    kind "reference": the REAL NFLlib (oracle/_ref/libnflref.so, prebuilt in the
    build container from /root/reference's own sources) when it loads here;
    otherwise kind "port": oracle/nfl_oracle.c, rebuilt -march=native on this host."""
    import numpy as np
    from nfllib_amd.params import params
    from oracle import oracle as O
    out = {}
    native_dir = os.path.join(ROOT, "gpurun_out")
    libpath = None
    try:
        os.makedirs(native_dir, exist_ok=True)
        O.build(native_out=native_dir)
        libpath = os.path.join(native_dir, "libnfloracle_native.so")
    except Exception:
        libpath = None
    o = O.Oracle(limb_bits, degree, nmoduli, params(limb_bits), libpath=libpath)
    gen = o.fill_uniform

    REPS = 5

    def run(fn, label):
        a, b = gen(8, SEED, 0), gen(8, SEED, 1)
        fn(a, b)                                                    # warm-up (tables, page faults)
        t0 = time.perf_counter(); fn(a, b); dt = time.perf_counter() - t0
        chunk = max(1, min(512, int(8 * 0.5 / max(dt, 1e-6))))      # ~0.5 s of work per call
        a, b = gen(chunk, SEED, 0), gen(chunk, SEED, 1)
        rates, done, t_all = [], 0, 0.0
        for _ in range(REPS):
            d_r, t_r = 0, 0.0
            while t_r < budget_s / REPS:
                t0 = time.perf_counter(); fn(a, b); t_r += time.perf_counter() - t0
                d_r += chunk
            rates.append(d_r / t_r); done += d_r; t_all += t_r
        rates.sort()
        return {"value": rates[REPS // 2], "min": rates[0], "max": rates[-1],
                "sample": "median of %d repetitions, %d polymuls (%s) in %.1f s, 1 thread" % (REPS, done, label, t_all)}

    port = run(o.polymul, "oracle/nfl_oracle.c -O3 -march=native" if libpath else "oracle/nfl_oracle.c -O3 x86-64-v3")
    ref = None
    try:
        if O.ref_available():
            r = O.Reference(limb_bits, degree, nmoduli)
            ref = run(r.polymul, "real NFLlib via oracle/_ref, NFL_OPTIMIZED+NTT_AVX2 x86-64-v3")
    except Exception:
        ref = None
    best = ref or port
    out = {"value": round(best["value"], 2), "unit": "polymul/s", "cores": 1, "kind": "reference" if ref else "port",
           "sample": best["sample"], "spread": [round(best["min"], 2), round(best["max"], 2)],
           "port_value": round(port["value"], 2)}
    if check is not None:
        import numpy as np
        ha, hb, hc = check
        out["parity_sample_ok"] = bool(np.array_equal(o.polymul(ha, hb), hc))
        out["parity_sample"] = "%d polynomials of the timed batch recomputed by the CPU checker, bit-exact compare" % ha.shape[0]
    # socket-level figure for an honest comparison: the same port with the batch split
    # over every host thread (native pthreads inside the oracle library; polys are independent)
    try:
        nthreads = os.cpu_count() or 1
        nb = max(nthreads * 4, 256)
        a, b = gen(nb, SEED, 0), gen(nb, SEED, 1)
        o.polymul_mt(a, b, nthreads)
        done, t_used = 0, 0.0
        while t_used < max(2.0, budget_s * 0.4):
            t0 = time.perf_counter(); o.polymul_mt(a, b, nthreads); t_used += time.perf_counter() - t0
            done += nb
        out["all_cores"] = {"value": round(done / t_used, 1), "cores": nthreads, "kind": "port",
                            "sample": "%d polymuls on %d pthreads in %.1f s" % (done, nthreads, t_used)}
    except Exception as e:
        out["all_cores"] = {"value": None, "cores": 0, "sample": "failed: %r" % (e,)}
    try:
        with open("/proc/cpuinfo") as f:
            models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        out["cpu"] = models[0] if models else "unknown"
        out["host_cores"] = len(models)
    except Exception:
        pass
    return out


_ORACLES = {}


def oracle_for(limb_bits, degree, nmoduli):
    """the CPU checker (oracle/nfl_oracle.c rebuilt -march=native under gpurun_out/ when possible), one instance per shape.
    Test infrastructure: used OUTSIDE every timed region, to check sampled polynomials of the products the blocks time."""
    key = (limb_bits, degree, nmoduli)
    if key not in _ORACLES:
        from nfllib_amd.params import params
        from oracle import oracle as O
        libpath = None
        try:
            native_dir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(native_dir, exist_ok=True)
            O.build(native_out=native_dir)
            libpath = os.path.join(native_dir, "libnfloracle_native.so")
        except Exception:
            libpath = None
        _ORACLES[key] = O.Oracle(limb_bits, degree, nmoduli, params(limb_bits), libpath=libpath)
    return _ORACLES[key]


def oracle_samples(eng, a, b, c, idx, limbs=None):
    """Copy the polynomials `idx` of a timed product's operands and result to the host, recompute the products with the CPU
    oracle and compare bit for bit (and, given the device's lifted coefficients, the CRT lift of those results against
    oracle.crt_lift).  Returns the keys every bench block carries: parity_sample_ok, parity_sample[, crt_parity_sample_ok]."""
    import numpy as np
    o = oracle_for(eng.limb_bits, eng.degree, eng.nmoduli)
    idx = sorted(set(int(i) for i in idx))
    ha, hb, hc = (np.concatenate([eng.to_host(t[i:i + 1]) for i in idx]) for t in (a, b, c))
    t0 = time.perf_counter()
    out = {"parity_sample_ok": bool(np.array_equal(o.polymul(ha, hb), hc)),
           "parity_sample": "polynomials %s of the timed batch recomputed by the CPU oracle, bit-exact compare" % (idx,)}
    if limbs is not None:
        hl = np.concatenate([limbs[i:i + 1].detach().cpu().contiguous().numpy().view(np.uint64) for i in idx])
        out["crt_parity_sample_ok"] = bool(np.array_equal(np.asarray(o.crt_lift(hc)).reshape(hl.shape), hl))
    out["parity_sample_seconds"] = round(time.perf_counter() - t0, 2)
    return out


def measure_traffic(workload, batch):
    """HBM bytes one step moves, measured IN THIS RUN: two short re-runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace domains, as MI355X_MICROARCH.md's HBM
    section prescribes), summed over the product's kernels and corrected for gfx950 (FETCH_SIZE reports half of the
    streamed bytes, calibration in profiles/pmc_traffic.json).  Returns (bytes_per_step, source) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    steps, warmup = 3, 1
    products = steps + warmup + 2            # + the commutativity self-check product + the neighbour's shard (checksum of checksums)
    tot = {}
    for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(prefix="nflhip_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--batch", str(batch), "--steps", str(steps),
                   "--warmup", str(warmup), "--no-extras", "--no-cpu-baseline", "--no-traffic", "--no-rccl", "--no-side-configs", "--prewarm", "0"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            per_kernel = pmc_traffic.total(d, counter)
            if not per_kernel:
                return None, "no %s rows for the product's kernels" % counter
            tot[counter] = sum(v[1] for v in per_kernel.values()) * 1024.0 * factor / products
        except Exception as e:  # the profiler must never take the bench down
            return None, "in-run counter pass failed: %r" % (e,)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return tot["FETCH_SIZE"] + tot["WRITE_SIZE"], ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over "
                                                   "%d products, FETCH x2 (gfx950), per step; read %.4g B + written %.4g B"
                                                   % (products, tot["FETCH_SIZE"], tot["WRITE_SIZE"]))


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: re-run this command line as N ranks of one node
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...),
    one device each, RCCL backend; rank 0's JSON line passes through on stdout.  Exits non-zero when the node has fewer
    than N devices (NFLHIP_BENCH_ONE_DEVICE=1, the 1-GPU test knob, puts every rank on device 0 instead)."""
    import socket
    import subprocess
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    have = torch.cuda.device_count()
    if have < n and os.environ.get("NFLHIP_BENCH_ONE_DEVICE") != "1":
        raise SystemExit("--gpus %d: this node has %d GPU(s); refusing to report an n_gpus the run did not have" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver only supports dmabuf IPC (RCCL across processes)
    return subprocess.run(cmd, env=env).returncode


def side_config(torch, Engine, workload, batch, steps, dev, with_crt=False, round_trip=False):
    """One more BASELINE config timed inside the driver's run (extras.configs): `steps` products over a resident batch,
    HIP events on the launch stream, in-run counter traffic -- the same quantities as the headline, for the shapes the
    driver does not time itself."""
    lb, n, nm, _ = WORKLOADS[workload]
    eng = Engine(lb, n, nm, device=dev)
    try:
        a = eng.fill_uniform(eng.empty(batch), SEED, 0)
        b = eng.fill_uniform(eng.empty(batch), SEED, 1)
        c = eng.empty(batch)
        # warm-up by TIME: after an idle period the first launches of these long products run 5 - 20 % below the rate the part
        # then holds (profiles/r05_E_F_memory_plan_bound.txt item 4); half a second of products first, then the timed steps
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.5:
            for _ in range(4):
                eng.polymul(a, b, out=c)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            eng.polymul(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        ok = not eng.any_neq(c, eng.polymul(b, a))
        # ... and against the CPU oracle: the first polynomial, one inside, the first of the LAST chunk of the long-row plans
        # (n = 65536: the pipeline works through the batch in 4 chunks) and the last one
        try:
            parity = oracle_samples(eng, a, b, c, {0, batch // 3, batch - max(1, batch // 4), batch - 1} if n < 65536
                                    else {0, batch - batch // 4, batch - 1})
        except Exception as ex:   # reported (as a failed check), never fatal
            parity = {"parity_sample_ok": None, "parity_sample": "failed: %r" % (ex,)}
        alg = 3 * nm * n * (lb // 8)
        what = "BASELINE configs %s" % workload if workload != "F" else "the reference's largest test configuration, tests/CMakeLists.txt (32768, 124, uint64_t)"
        out = {"workload": "nfl::poly<uint%d_t,%d,%d> batched polymul (%s)" % (lb, n, nm, what), "batch": batch,
               "steps": steps, "value": round(batch / (ms * 1e-3), 1), "unit": "polymul/s", "ms_per_step": round(ms, 4),
               "achieved_GBs": round(alg * batch / (ms * 1e-3) / 1e9, 1), "frac": round(alg * batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "algorithmic_bytes_per_polymul": alg, "self_check": bool(ok)}
        out.update(parity)
        if round_trip:
            # BASELINE configs[0] is the NTT + INTT round trip of tests/ntt_perfs.cpp on this shape (there: CPU only)
            rt = a.clone()
            for _ in range(4):
                eng.intt_(eng.ntt_(rt))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                eng.intt_(eng.ntt_(rt))
            e1.record()
            torch.cuda.synchronize()
            msr = e0.elapsed_time(e1) / steps
            rt_bytes = 4 * nm * n * (lb // 8)          # two in-place transforms: 2 x (read + write)
            out["ntt_intt_round_trip"] = {"value": round(batch / (msr * 1e-3), 1), "unit": "round trips/s", "ms_per_step": round(msr, 4),
                                          "achieved_GBs": round(rt_bytes * batch / (msr * 1e-3) / 1e9, 1),
                                          "frac": round(rt_bytes * batch / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "returns_the_input": bool(not eng.any_neq(rt, a)),
                                          "reference": "tests/ntt_perfs.cpp: 50 000 round trips of poly<uint32_t,1024,1> on one core"}
            del rt
        if with_crt:
            # BASELINE configs[4] is "CRT lift + poly-mul": GMP::poly2mpz (gmp.hpp:183-209) of the product, all coefficients
            L = eng.crt_limbs
            limbs = eng.crt_lift(c)
            torch.cuda.synchronize()
            try:
                cp = oracle_samples(eng, a, b, c, {0, batch - 1}, limbs=limbs)
                out["crt_parity_sample_ok"] = cp["crt_parity_sample_ok"] and cp["parity_sample_ok"]
                out["crt_parity_sample"] = "GMP::poly2mpz of polynomials 0 and %d of the product against oracle.crt_lift, bit-exact limbs" % (batch - 1)
            except Exception as ex:
                out["crt_parity_sample_ok"] = None
                out["crt_parity_sample"] = "failed: %r" % (ex,)
            e0.record()
            for _ in range(steps):
                eng.crt_lift(c)
            e1.record()
            torch.cuda.synchronize()
            msl = e0.elapsed_time(e1) / steps
            crt_bytes = nm * n * (lb // 8) + n * L * 8      # residues read + limbs written
            out["crt_lift"] = {"value": round(batch / (msl * 1e-3), 1), "unit": "polys/s", "ms_per_step": round(msl, 4),
                               "achieved_GBs": round(crt_bytes * batch / (msl * 1e-3) / 1e9, 1),
                               "frac": round(crt_bytes * batch / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_poly": crt_bytes, "limbs_per_coefficient": L}
            # BASELINE configs[4] states ONE workload, "CRT lift + poly-mul": the product AND the lift of the product, back to
            # back on the stream, as one rate in polynomials per second (bytes: the two operations' algorithmic bytes together)
            e0.record()
            for _ in range(steps):
                eng.polymul(a, b, out=c)
                eng.crt_lift(c)
            e1.record()
            torch.cuda.synchronize()
            msb = e0.elapsed_time(e1) / steps
            out["polymul_plus_crt_lift"] = {"value": round(batch / (msb * 1e-3), 1), "unit": "polys/s", "ms_per_step": round(msb, 4),
                                            "achieved_GBs": round((alg + crt_bytes) * batch / (msb * 1e-3) / 1e9, 1),
                                            "frac": round((alg + crt_bytes) * batch / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "algorithmic_bytes_per_poly": alg + crt_bytes,
                                            "what": "c = a * b (coefficient form in and out), then GMP::poly2mpz of c: BASELINE configs[4] as one figure"}
            # and the way back, GMP::mpz2poly (gmp.hpp:211-219): the lifted coefficients projected onto the moduli again
            back = eng.crt_project(limbs)
            ok_rt = not eng.any_neq(back, c)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                eng.crt_project(limbs)
            e1.record()
            torch.cuda.synchronize()
            msp = e0.elapsed_time(e1) / steps
            out["crt_project"] = {"value": round(batch / (msp * 1e-3), 1), "unit": "polys/s", "ms_per_step": round(msp, 4),
                                  "achieved_GBs": round(crt_bytes * batch / (msp * 1e-3) / 1e9, 1),
                                  "frac": round(crt_bytes * batch / (msp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "round_trip_check": bool(ok_rt)}
            del limbs, back
        del a, b, c
    finally:
        eng.close()
    tr, src = measure_traffic(workload, batch)
    out["traffic_bytes_per_polymul"] = None if tr is None else round(tr / batch, 1)
    out["traffic_ratio"] = None if tr is None else round(tr / batch / alg, 4)
    out["traffic_source"] = src
    return out


D_SHARD = 1 << 17   # BASELINE configs[3]: 2^20 polynomials over 8 GPUs


def shard_config_d(torch, eng, rank, world, steps, barrier, gather_u64, max_over_ranks):
    """extras.configs.D -- one shard of BASELINE configs[3] per GPU on the headline's engine (same ring, same kernel): `steps`
    products over 2^17 resident polynomials per rank, barrier + synchronize on both sides, MAX over ranks, and a checksum of
    checksums over the logical global batch of world x 2^17.  NFLHIP_BENCH_D_SHARD shrinks the shard (tests on a shared device);
    the block then says that it is not config 4."""
    shard = int(os.environ.get("NFLHIP_BENCH_D_SHARD", D_SHARD))
    first = rank * shard
    # pre-flight: three resident tensors of the shard (48 GiB at 2^17) + the commutativity check's fourth + head room must fit in
    # the device's FREE memory; every rank takes the same decision (the smallest free figure over the ranks decides)
    need = 4 * shard * eng.nmoduli * eng.degree * 8 + (1 << 30)
    free = -max_over_ranks(-float(torch.cuda.mem_get_info()[0]))
    preflight = {"ok": bool(need <= free), "needed_GiB": round(need / 2.0**30, 2), "free_GiB": round(free / 2.0**30, 2)}
    if not preflight["ok"]:
        raise RuntimeError("pre-flight: a shard of %d polynomials needs %.1f GiB resident, %.1f GiB are free on the fullest device"
                           % (shard, preflight["needed_GiB"], preflight["free_GiB"]))
    a = eng.fill_uniform(eng.empty(shard), SEED, 0, first_poly=first)
    b = eng.fill_uniform(eng.empty(shard), SEED, 1, first_poly=first)
    c = eng.empty(shard)
    steps = max(1, min(steps, 10))
    for _ in range(2):
        eng.polymul(a, b, out=c)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        eng.polymul(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    kernel_ms = max_over_ranks(e0.elapsed_time(e1) / steps)
    commutes = not eng.any_neq(c, eng.polymul(b, a))
    try:
        parity = oracle_samples(eng, a, b, c, {0, shard // 2, shard - 1}) if rank == 0 else {}
    except Exception as ex:
        parity = {"parity_sample_ok": None, "parity_sample": "failed: %r" % (ex,)}
    own = gather_u64(eng.digest(c, first_poly=first))
    nxt = (rank + 1) % world
    eng.fill_uniform(a, SEED, 0, first_poly=nxt * shard)
    eng.fill_uniform(b, SEED, 1, first_poly=nxt * shard)
    eng.polymul(a, b, out=c)
    cross = gather_u64(eng.digest(c, first_poly=nxt * shard))
    cross = [cross[(r - 1) % world] for r in range(world)]
    del a, b, c
    torch.cuda.empty_cache()
    from nfllib_amd import sharding
    alg = 3 * eng.nmoduli * eng.degree * 8
    ach = alg * shard / (kernel_ms * 1e-3) / 1e9
    return {"workload": "nfl::poly<uint64_t,4096,4> batch=2^20 sharded across 8 GPUs (BASELINE configs D): 2^17 polynomials per GPU",
            "batch_per_gpu": shard, "n_gpus": world, "global_batch": shard * world,
            "is_baseline_config_4": bool(shard == D_SHARD and world == 8), "shard_is_baseline_shard": bool(shard == D_SHARD),
            "steps": steps, "value": round(world * shard * steps / dt, 1), "unit": "polymul/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "kernel_ms": round(kernel_ms, 4), "achieved_GBs_per_gpu": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
            "resident_GiB_per_gpu": round(3 * shard * eng.nmoduli * eng.degree * 8 / 2.0**30, 2), "scaling": "weak",
            "self_check": bool(commutes and own == cross), **parity, "preflight": preflight,
            "checksum_of_checksums": {"ok": bool(own == cross), "sum_of_shard_digests": "%016x" % sharding.combine_digests(own),
                                      "recomputed_on_the_neighbouring_gpu": "%016x" % sharding.combine_digests(cross), "shards": world},
            "how": "value = world x shard x steps / max-over-ranks wall time between barriers; no data-path collective"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="polys per GPU (default: workload default)")
    ap.add_argument("--workload", default="B", choices=sorted(WORKLOADS))
    ap.add_argument("--prewarm", type=float, default=1.0, help="seconds of untimed launches before the W warm-up steps (clock ramp after idle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel secondary rates")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 counter passes (roofline.traffic)")
    ap.add_argument("--no-rccl", action="store_true", help="N = 1 only: skip the world-size-1 RCCL initialisation")
    ap.add_argument("--no-side-configs", action="store_true", help="workload B only: skip extras.configs (configs A, C, D, E and the reference's largest test shape F timed in the same run)")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="N > 1 only: also time one step whose operands start on rank 0 and whose product returns there "
                         "(grouped RCCL send/recv of contiguous shards, SURVEY.md 8(e)); reported beside `value`, never in it")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus N`): become the driver's launch line -- N ranks, one per GPU, under
        # torch.distributed.run on 127.0.0.1 -- instead of quietly running one rank.  Refuses when the node has fewer devices.
        sys.exit(self_launch(args.gpus))

    # stdout carries exactly ONE line, the result: libraries that print banners there (RCCL announces its version on
    # communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from nfllib_amd import Engine
    from nfllib_amd import sharding

    rank, world, local_rank = sharding.env_rank_world()
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # test knobs (tests/test_bench_contract.py runs the N = 2 launch line on a 1-GPU box): every rank on device 0 and the
    # three tiny control collectives (barrier, max, max) over gloo -- RCCL refuses two ranks on one device
    backend = os.environ.get("NFLHIP_BENCH_BACKEND", "nccl")
    dev = local_rank if world > 1 and os.environ.get("NFLHIP_BENCH_ONE_DEVICE") != "1" else 0
    torch.cuda.set_device(dev)
    # RCCL is initialised at EVERY world size, 1 included (a 1-GPU box then exercises the same init / barrier / all-reduce
    # calls the N-GPU launch line makes; --no-rccl keeps the profiler's inner re-runs of this script light)
    rccl = {"initialised": False, "world_size": world, "backend": backend}
    use_dist = world > 1 or not args.no_rccl
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:
            if world > 1:
                raise
            use_dist = False
            rccl["error"] = "torch.distributed init failed: %r" % (e,)
    red_dev = torch.device("cuda", dev) if backend == "nccl" else None

    lb, n, nm, dflt = WORKLOADS[args.workload]
    batch = args.batch or dflt
    eng = Engine(lb, n, nm, device=dev)
    first_poly = rank * batch  # shard of the logical global batch; no data-path collective
    a = eng.fill_uniform(eng.empty(batch), SEED, 0, first_poly=first_poly)
    b = eng.fill_uniform(eng.empty(batch), SEED, 1, first_poly=first_poly)
    c = eng.empty(batch)
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    # the engine's own communicator (include/nflhip.h nflhip_comm_*: ncclCommInitRank through the C ABI; the id travels
    # over torch.distributed) -- what a C++ caller with one process per GPU uses; here it carries the digests
    comm = None
    if use_dist and backend == "nccl":
        try:
            from nfllib_amd import Comm
            box = [Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = Comm(eng, world, rank, box[0])
            comm.barrier()
            probe = torch.ones(1, dtype=torch.int64, device=red_dev)
            dist.all_reduce(probe)
            rccl = {"initialised": True, "world_size": world, "backend": backend, "all_reduce_of_ones": int(probe.item()),
                    "calls": "torch.distributed nccl init + barrier + all_reduce; nflhip_comm_create (ncclCommInitRank) + "
                             "nflhip_comm_barrier + nflhip_comm_allgather_u64"}
        except Exception as e:   # the digests then travel over torch.distributed (same RCCL); the line says so
            comm = None
            rccl = {"initialised": world > 1, "world_size": world, "backend": backend, "error": "nflhip_comm: " + repr(e)}

    # device warm-up BY TIME before the W warm-up steps: after an idle period the part needs a few hundred milliseconds of work to
    # reach the clock it then holds under its power limit -- the first launches of the metric kernel run ~6 % slower than the held
    # rate (profiles/r03_power_clock.txt: 3.24 ms first, 3.04 ms held per 16 384 products), and W = 5 steps are 15 ms.  Untimed,
    # like the W steps; `sustained` below (the same launch held for 2.5 s) is the figure this makes the headline agree with.
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.prewarm:
        for _ in range(8):
            eng.polymul(a, b, out=c)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.polymul(a, b, out=c)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # on the current stream == the stream the kernels are launched on
    for _ in range(args.steps):
        eng.polymul(a, b, out=c)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    if use_dist:
        dt = sharding.allreduce_max(dt, dist, device=red_dev)
        kernel_ms = sharding.allreduce_max(kernel_ms, dist, device=red_dev)

    # cheap in-run sanity: one sampled poly against nothing but itself commuting (parity lives in tests/)
    ok = not eng.any_neq(c, eng.polymul(b, a))
    # ... and host copies of a few polynomials of the timed product for the CPU checker (cpu_baseline leg below)
    check = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        idx = sorted({0, batch // 3, batch // 2, batch - 1})
        check = tuple(np.concatenate([eng.to_host(t[i:i + 1]) for i in idx]) for t in (a, b, c))

    # checksum of checksums: every rank digests ITS shard of the product (positions counted in the
    # logical global batch, so the digests add up to the digest of the whole batch) and then recomputes its NEIGHBOUR's
    # shard from the shared counter stream on its own GPU; the two sets of digests -- shard r computed on GPU r, and on
    # GPU r - 1 -- must be identical, rank by rank and in the sum.
    def gather_u64(v):
        if comm is not None:
            return comm.allgather_u64(v)
        if use_dist:
            return sharding.allgather_digests(v, dist, world, device=red_dev)
        return [v]
    own = gather_u64(eng.digest(c, first_poly=first_poly))
    nxt = (rank + 1) % world
    eng.fill_uniform(a, SEED, 0, first_poly=nxt * batch)
    eng.fill_uniform(b, SEED, 1, first_poly=nxt * batch)
    eng.polymul(a, b, out=c)
    cross = gather_u64(eng.digest(c, first_poly=nxt * batch))      # entry r = shard (r + 1) mod world
    cross = [cross[(r - 1) % world] for r in range(world)]          # ... reordered: entry r = shard r
    sums_ok = own == cross
    checksum = {"ok": bool(sums_ok), "sum_of_shard_digests": "%016x" % sharding.combine_digests(own),
                "recomputed_on_the_neighbouring_gpu": "%016x" % sharding.combine_digests(cross), "shards": world,
                "via": "nflhip_comm_allgather_u64 (RCCL)" if comm is not None else ("torch.distributed" if use_dist else "local")}
    ok = ok and sums_ok

    alg_bytes_per_poly = 3 * nm * n * (lb // 8)       # read a, read b, write c
    launch_bytes = alg_bytes_per_poly * batch
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    value = world * batch * args.steps / dt

    kwl = "B" if args.workload == "D" else args.workload   # D runs B's kernel on a larger batch
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and not args.no_traffic:
        # (bytes per polynomial do not depend on the batch: the counter passes use at most the kernel's default batch)
        tb = min(batch, WORKLOADS[kwl][3])
        traffic, traffic_src = measure_traffic(kwl, tb)
        if traffic is not None:
            traffic = round(traffic / tb * batch, 1)
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if traffic is None and os.path.exists(tpath):
        why = traffic_src
        try:
            tj = json.load(open(tpath)).get("workloads", {}).get(kwl)
            if tj:
                traffic = tj["hbm_bytes_per_poly"] * batch
                traffic_src = "NOT measured in this run (%s): profiles/pmc_traffic.json, rocprofv3 FETCH_SIZE/WRITE_SIZE passes of round %s, kernel %s" % (
                    why or "skipped", tj.get("round", "?"), tj.get("kernel") or ",".join(tj.get("kernels", {})))
        except Exception:
            traffic = None

    scatter = None
    if world > 1 and args.scatter_gather and comm is not None:
        # data originating on one device: root -> shards -> polymul -> root, through the C ABI's RCCL scatter / gather
        # (grouped ncclSend / ncclRecv of contiguous shards).  Outside the timed region of `value`.
        fa = fb = fc = None
        if rank == 0:
            fa = eng.fill_uniform(eng.empty(batch * world), SEED, 0)
            fb = eng.fill_uniform(eng.empty(batch * world), SEED, 1)
            fc = eng.empty(batch * world)
        torch.cuda.synchronize(); barrier()
        ts = time.perf_counter()
        comm.scatter(a, fa, batch * world)
        comm.scatter(b, fb, batch * world)
        eng.polymul(a, b, out=c)
        comm.gather(fc, c, batch * world)
        torch.cuda.synchronize(); barrier()
        tsg = sharding.allreduce_max(time.perf_counter() - ts, dist, device=red_dev)
        scatter = {"polymul_per_s_incl_scatter_gather": round(world * batch / tsg, 1), "seconds": round(tsg, 4),
                   "bytes_moved": 3 * (world - 1) * batch * nm * n * (lb // 8), "via": "nflhip_scatter_dev / nflhip_gather_dev (RCCL)",
                   "note": "one step; both operands scattered from rank 0, product gathered back (grouped send/recv over xGMI)"}
        del fa, fb, fc
    elif world > 1 and args.scatter_gather:
        # data originating on one device: root -> shards -> polymul -> root.  Outside the timed region of `value`.
        fa = fb = fc = None
        if rank == 0:
            fa = eng.fill_uniform(eng.empty(batch * world), SEED, 0)
            fb = eng.fill_uniform(eng.empty(batch * world), SEED, 1)
            fc = eng.empty(batch * world)
        torch.cuda.synchronize(); barrier()
        ts = time.perf_counter()
        sharding.scatter_batch(fa, a, dist, rank, world)
        sharding.scatter_batch(fb, b, dist, rank, world)
        eng.polymul(a, b, out=c)
        sharding.gather_batch(c, fc, dist, rank, world)
        torch.cuda.synchronize(); barrier()
        tsg = sharding.allreduce_max(time.perf_counter() - ts, dist, device=red_dev)
        scatter = {"polymul_per_s_incl_scatter_gather": round(world * batch / tsg, 1), "seconds": round(tsg, 4),
                   "bytes_moved": 3 * (world - 1) * batch * nm * n * (lb // 8),
                   "note": "one step; both operands scattered from rank 0, product gathered back (grouped send/recv over xGMI)"}
        del fa, fb, fc

    extras = None
    power = None
    sustained = None
    if rank == 0 and world == 1 and not args.no_extras:
        # package power and shader clock UNDER the product kernel (the product kernels run at the package
        # power limit, so their clock -- not their schedule -- is what separates them from the issue bound): ~2.5 s of
        # products queued on the stream, rocm-smi sampled beside them.  Outside the timed region.
        try:
            import subprocess
            t_step = dt / args.steps
            n_sus = max(8, min(20000, int(2.5 / max(t_step, 1e-6))))
            sus0, sus1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sus0.record()
            for _ in range(n_sus):
                eng.polymul(a, b, out=c)
            sus1.record()
            samples = []
            t_end = time.perf_counter() + 2.2
            while time.perf_counter() < t_end and len(samples) < 8:
                txt = subprocess.run(["rocm-smi", "-P", "-c", "-M", "--json"], capture_output=True, text=True, timeout=10).stdout
                js = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
                card = js.get("card%d" % dev) or next(iter(js.values()))
                watts = next((float(v) for k, v in card.items() if "Power (W)" in k and "Max" not in k), None)
                cap = next((float(v) for k, v in card.items() if "Max" in k and "Power" in k), None)
                sclk = next((v for k, v in card.items() if k.startswith("sclk clock speed")), "")
                mhz = int("".join(ch for ch in sclk if ch.isdigit()) or 0)
                if watts is not None:
                    samples.append((watts, mhz, cap))
            torch.cuda.synchronize()
            sus_s = sus0.elapsed_time(sus1) * 1e-3
            sustained = {"value": round(n_sus * batch / sus_s, 1), "unit": "polymul/s", "steps": n_sus, "seconds": round(sus_s, 3),
                         "frac": round(n_sus * batch * 3 * nm * n * (lb // 8) / sus_s / 1e9 / HBM_PEAK_GBS, 4),
                         "how": "the same launch held for ~2.5 s (HIP events on the launch stream): the package settles at its power "
                                "limit after the first ~0.1 s, which the K-step headline does not reach"}
            busy = [x for x in samples[1:] if x[0] > 0.5 * max(y[0] for y in samples)] or samples
            if busy:
                power = {"package_W": round(sum(x[0] for x in busy) / len(busy), 1), "sclk_MHz": round(sum(x[1] for x in busy) / len(busy)),
                         "package_limit_W": busy[0][2], "samples": len(busy),
                         "how": "rocm-smi -P -c -M sampled while ~2.5 s of products run, outside the timed region"}
        except Exception as ex:  # reported, never fatal
            torch.cuda.synchronize()
            power = {"error": repr(ex)}
    if rank == 0 and world == 1 and not args.no_extras:
        # secondary rates of SURVEY.md 8(d): per-kernel transforms, point-wise ops, the
        # "one operand pre-transformed" product and CRT lift/project (same batch, same stream)
        from nfllib_amd import OP_ADD, OP_MUL

        def rate(fn, reps=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3
        w = lb // 8
        tr_bytes, pw_bytes = 2 * nm * n * w * batch, 3 * nm * n * w * batch
        bn = eng.ntt_(b.clone())
        t_f = rate(lambda: eng.ntt_(c)); t_i = rate(lambda: eng.intt_(c))
        t_add = rate(lambda: eng.pointwise(OP_ADD, a, b, out=c)); t_mul = rate(lambda: eng.pointwise(OP_MUL, a, b, out=c))
        t_pn = rate(lambda: eng.polymul(a, bn, out=c, b_is_ntt=True))
        sub = min(batch, max(2048, (256 << 20) // (nm * n * w)))   # (enough bytes for the launch not to dominate)
        limbs = eng.crt_lift(a[:sub])
        t_l = rate(lambda: eng.crt_lift(a[:sub]), 5); t_p = rate(lambda: eng.crt_project(limbs), 5)
        crt_bytes = sub * (nm * n * w + n * eng.crt_limbs * 8)
        # samplers: polys per second written (bytes = one write of the batch)
        skey = bytes(range(32))
        gs = eng.gauss_create(3.19, 128, n)
        t_su = rate(lambda: eng.sample(c, 0, skey, stream_id=1), 5)
        t_sg = rate(lambda: eng.sample_gauss(c, gs, skey, stream_id=2), 5)
        eng.gauss_destroy(gs)
        # the narrow draws (include/nflhip.h NFLHIP_DIST_NARROW, nflhip_gauss_set_draw_bits): keystream lanes instead of 64-bit words
        gs32 = eng.gauss_create(3.19, 128, n, draw_bits=32 if n >= 16 else 64)
        t_sun = rate(lambda: eng.sample(c, 0, skey, stream_id=1, narrow=True), 5)
        t_sgn = rate(lambda: eng.sample_gauss(c, gs32, skey, stream_id=2), 5)
        small = eng.empty_small(batch)
        t_sgc = rate(lambda: eng.sample_gauss_small(small, gs32, skey, stream_id=3), 5)
        del small
        eng.gauss_destroy(gs32)
        # the host-pointer entry point (nflhip_polymul: pageable host buffers in and out, PCIe included) -- never `value`
        hb = min(batch, max(1, (256 << 20) // (nm * n * w)))
        ha, hbb = eng.to_host(a[:hb]), eng.to_host(b[:hb])
        hcc = eng.h_polymul(ha, hbb)          # (also touches the result array's pages: the caller's arrays exist before the call)
        t_hosts = []
        for _ in range(5):                    # host threads copy: the median of five calls (single calls vary 2-3x with page placement)
            t0h = time.perf_counter()
            eng.h_polymul(ha, hbb, out=hcc)
            t_hosts.append(time.perf_counter() - t0h)
        t_host = sorted(t_hosts)[2]
        # core::ntt (core.hpp:455-532), the cyclic row transform tests/ntt_perfs.cpp:155-171 times on
        # poly<uint64_t,1024,2> rows (BASELINE configs[0]'s path): nflhip_ntt_row_dev over resident rows of one modulus
        row_extra = {}
        try:
            er = Engine(64, 1024, 2, device=dev)
            nrows = 1 << 17
            rows = er.fill_uniform(er.empty(nrows // 2), SEED, 0).view(-1, 1024)
            t_row = rate(lambda: er.ntt_row_(rows, 0))
            row_extra = {"core_ntt_rows_per_s_u64_1024": round(nrows / t_row, 1),
                         "core_ntt_GBs_u64_1024": round(nrows * 1024 * 8 * 2 / t_row / 1e9, 1),
                         "core_ntt_reference_us_per_row": "6.86 us on one core (tests/ntt_perfs.cpp, SURVEY.md section 6)"}
            del rows
            er.close()
        except Exception as ex:  # secondary figure: never takes the bench down
            row_extra = {"core_ntt_rows_per_s_u64_1024": None, "core_ntt_error": repr(ex)}
        extras = {
            "ntt_fwd_per_s": round(batch / t_f, 1), "ntt_fwd_GBs": round(tr_bytes / t_f / 1e9, 1),
            "ntt_inv_per_s": round(batch / t_i, 1), "ntt_inv_GBs": round(tr_bytes / t_i / 1e9, 1),
            "pointwise_add_GBs": round(pw_bytes / t_add / 1e9, 1), "pointwise_mul_GBs": round(pw_bytes / t_mul / 1e9, 1),
            "polymul_b_pretransformed_per_s": round(batch / t_pn, 1),
            "crt_lift_per_s": round(sub / t_l, 1), "crt_lift_GBs": round(crt_bytes / t_l / 1e9, 1),
            "crt_project_per_s": round(sub / t_p, 1), "crt_project_GBs": round(crt_bytes / t_p / 1e9, 1),
            "sample_uniform_per_s": round(batch / t_su, 1), "sample_uniform_GBs": round(batch * nm * n * w / t_su / 1e9, 1),
            "sample_gaussian_per_s": round(batch / t_sg, 1),
            "narrow_draws": {"sample_uniform_per_s": round(batch / t_sun, 1), "sample_uniform_GBs": round(batch * nm * n * w / t_sun / 1e9, 1),
                             "sample_gaussian_per_s": round(batch / t_sgn, 1), "sample_gaussian_compact_per_s": round(batch / t_sgc, 1),
                             "sample_gaussian_compact_G_coefficients_per_s": round(batch * n / t_sgc / 1e9, 1),
                             "what": "limb-width / 32-bit keystream lanes per value instead of one 64-bit word (profiles/r05_sampler_rates.txt)"},
            "host_pointer_polymul_per_s": round(hb / t_host, 1),
            **row_extra,
            "note": "GB/s are algorithmic bytes (SURVEY.md 8(d)) / event time; polys per second over the same batch",
        }
        del bn, limbs
        # what callers run AROUND the transforms: the reference's LWE demo (tests/nfllib_demo_main_op.cpp:26-58) on this
        # workload's ring, operator by operator and through the transform-fused pipelines -- same
        # keystreams, so the two plans must produce the same ciphertexts (digests compared)
        if lb == 64 and n in (4096, 8192, 16384, 32768):
            try:
                import types
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import lwe_demo
                lw = {}
                for plan in ("unfused", "fused"):
                    la = types.SimpleNamespace(degree=n, nmoduli=nm, batch=min(batch, 8192), sigma=3.19, reps=5, plan=plan, grid=0, fixed_key=True)
                    r, okl = lwe_demo.run(la)
                    lw[plan] = {"encryptions_per_s": r["encryptions_per_s"], "decryptions_per_s": r["decryptions_per_s"],
                                "decrypts_to_zero": okl, "digest": r["digest"]}
                lw["same_ciphertexts"] = lw["fused"]["digest"] == lw["unfused"]["digest"]
                lw["batch"] = min(batch, 8192)
                lw["what"] = ("encrypt: 3 Gaussian polynomials, 3 forward transforms, 2 multiply-adds against the public key; decrypt: "
                              "multiply-subtract against the secret key + inverse transform; fused = nflhip_fwd_fma2_dev / nflhip_fma_inv_dev")
                extras["lwe"] = lw
            except Exception as ex:   # secondary figure: never takes the bench down
                extras["lwe"] = {"error": repr(ex)}
            # ... and the same demo as the reference's callers write it: plain nfl::poly_p operators through the drop-in header
            # (deferred queue, transform fusion) and nfl::device_batch -- tests/cpp/resident_main.cpp, u64/4096/4, 16 384 iterations
            if (n, nm) == (4096, 4) and isinstance(extras.get("lwe"), dict) and "error" not in extras["lwe"]:
                try:
                    exe = os.path.join(ROOT, "tests", "cpp", "resident_test")
                    if not os.path.exists(exe):
                        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "resident_test"], check=True, timeout=300,
                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    torch.cuda.synchronize()
                    r = subprocess.run([exe], capture_output=True, text=True, timeout=240, env=dict(os.environ, NFL_LWE_REPS="16384"))
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    if r.returncode != 0 or not line:
                        raise RuntimeError("resident_test rc %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:]))
                    cpp = json.loads(line[0])["lwe_u64_4096_4"]
                    extras["lwe"]["cpp_header"] = {k: cpp[k] for k in (
                        "poly_p_encryptions_per_s", "poly_p_decryptions_per_s", "device_batch_encryptions_per_s", "device_batch_decryptions_per_s",
                        "device_batch_fused_encryptions_per_s", "device_batch_fused_decryptions_per_s", "poly_p_eager_encryptions_per_s",
                        "poly_p_eager_decryptions_per_s", "deferred_operations", "launches_they_became") if k in cpp}
                    extras["lwe"]["cpp_header"]["what"] = ("tests/cpp/resident_test, 16 384 iterations: the demo written with nfl::poly_p operators (deferred queue) "
                                                           "and on nfl::device_batch (operator by operator / its fused methods); all checks passed")
                except Exception as ex:
                    extras["lwe"]["cpp_header"] = {"error": repr(ex)}
        # what an UNCHANGED caller of the reference gains (north_star: "drops into existing callers"): one of the reference's own
        # timing programs, same source, real NFLlib on this host's CPU against the drop-in header on this GPU
        # (tools/reference_programs.py; the full table of all 11 programs: profiles/r06_reference_programs.txt)
        if args.workload == "B" and not args.no_side_configs:
            try:
                tool = os.path.join(ROOT, "tools", "reference_programs.py")
                rp_json = os.path.join(ROOT, "gpurun_out", "reference_programs_bench.json")
                os.makedirs(os.path.dirname(rp_json), exist_ok=True)
                torch.cuda.synchronize()
                r = subprocess.run([sys.executable, tool, "--reps", "1", "--only", "nfllib_demo_main_op__8192_124_uint64_t", "--json", rp_json],
                                   capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    raise RuntimeError((r.stdout + r.stderr)[-300:])
                rp = json.load(open(rp_json))
                prog = rp["programs"]["nfllib_demo_main_op__8192_124_uint64_t"]
                extras["reference_programs"] = {
                    "program": "tests/nfllib_demo_main_op.cpp, CONFIG 8192, 124, uint64_t -- unchanged source, built twice",
                    "cpu": rp["summary"]["cpu"], "ops_us": {k: [v["cpu_us"], v["gpu_us"]] for k, v in prog["ops"].items()},
                    "columns": "[real NFLlib on one host core, drop-in header on the MI355X] microseconds per polynomial",
                    "note": "nfl::poly is a host array: every device call carries its PCIe round trip; resident nfl::poly_p is the fast path",
                    "full_table": "profiles/r06_reference_programs.txt"}
            except Exception as ex:   # secondary figure: never takes the bench down
                extras["reference_programs"] = {"error": repr(ex)}
        # the other single-GPU BASELINE configs, timed inside this same run (driver-visible, not builder-only): configs[2]
        # (C) and configs[4] (E: "CRT lift + poly-mul"); the headline's own tensors are released first
        if args.workload == "B" and not args.no_side_configs:
            a = b = c = None
            torch.cuda.empty_cache()
            side = {}
            for wl, sb, st_, crt in (("A", 1 << 19, 20, False), ("G", 8192, 20, False), ("C", 2048, 20, False), ("F", 2048, 20, False), ("E", 128, 20, True)):
                try:
                    side[wl] = side_config(torch, Engine, wl, sb, st_, dev, with_crt=crt, round_trip=wl == "A")
                except Exception as ex:   # reported, never fatal
                    side[wl] = {"error": repr(ex)}
            extras["configs"] = side

    # BASELINE configs[3] (D): "nfl::poly<uint64_t, 4096, 4> batch=2^20 sharded across 8 GPUs" -- 2^17 polynomials per GPU, so the
    # driver's `--gpus 8` line carries config 4 itself (global batch N x 2^17 = 2^20 at N = 8) beside the weak-scaling `value` on
    # workload B's per-GPU batch; the N = 1 line carries the same block for ONE shard (3 x 16 GiB resident).  Collective: every
    # rank runs it, same barriers and max-over-ranks clock as the headline, its own checksum of checksums.
    config_d = None
    if args.workload == "B" and not args.no_side_configs:
        a = b = c = None
        torch.cuda.empty_cache()
        try:
            config_d = shard_config_d(torch, eng, rank, world, args.steps, barrier, gather_u64,
                                      (lambda v: sharding.allreduce_max(v, dist, device=red_dev)) if use_dist else (lambda v: v))
        except Exception as ex:   # reported, never fatal (every rank fails or succeeds alike: allocation sizes are the same)
            config_d = {"error": repr(ex)}
        if rank == 0:
            if extras is None:
                extras = {}
            extras.setdefault("configs", {})["D"] = config_d

    result = {
        "metric": "poly-mults/sec (NTT+pointwise+INTT), n=4096, 4x62-bit moduli" if args.workload == "B"
                  else "poly-mults/sec (NTT+pointwise+INTT), n=%d, %dx%d-bit moduli" % (n, nm, lb - 2),
        "value": round(value, 1), "unit": "polymul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "prewarm_s": args.prewarm, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u%d" % lb, "data": "synthetic",
        "config": {"workload": "nfl::poly<uint%d_t,%d,%d> batched polymul (BASELINE configs %s)" % (lb, n, nm, args.workload),
                   "degree": n, "nmoduli": nm, "limb_bits": lb, "batch_per_gpu": batch, "global_batch": batch * world,
                   "parallelism": "batch-split x%d, no data-path collective" % world, "self_check": bool(ok),
                   "checksum_of_checksums": checksum, "rccl": rccl},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": {"A": "nflhip_row1024_i2_u32_asm", "B": "nflhip_polymul4096i2_asm (2 stages dropped each way, base multiplication mod X^4 -+ zeta)",
                                "C": "nflhip_polymul16384i2_asm",
                                "E": "nflhip_polymul_pipe65536nti2_asm (block products on incomplete transforms + streaming passes, 6 launches per step)",
                                "F": "nflhip_ntt_fwd32768si2_asm (b -> scratch two stages short, layout [block][pair][thread]) + nflhip_polymul_ntt32768si2_asm "
                                     "(a, b' streamed, base multiplication mod X^4 -+ zeta): register-resident 32768-word rows, 2 launches per step" if batch * nm >= 256 else
                                     "nflhip_polymul_xcd32768_asm (one launch of persistent workgroups; fewer than 256 rows)",
                                "G": "nflhip_polymul8192i2_asm", "H": "nflhip_row128_u16_asm", "T": "nflhip_row8_u32_asm"}[kwl],
                     "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": launch_bytes},
    }
    # what actually binds the metric kernel: not HBM -- `bound` / `frac` above stay as SURVEY.md 8(d)
    # defines them -- but wave64 VALU issue at the package power limit.  `ceiling_frac_no_memory` is the measured rate of the same
    # kernel with every HBM, twiddle and LDS access removed (only the butterflies' arithmetic left), as a fraction of the same
    # 8 TB/s: the most this instruction stream can reach on this part whatever the memory system does.
    if kwl == "B":
        ceil = 0.351
        result["roofline"].update({
            "binding": "valu-issue at the package power limit",
            "ceiling_frac_no_memory": ceil,
            "ceiling_measured": "round 6, session v, on the shipped incomplete-transform kernel (not re-measured in this run)",
            "ceiling_source": "profiles/r06_power_ablation.txt: nflhip_polymul4096i2_asm with HBM, twiddle and LDS traffic removed runs 7.15 M polymul/s "
                              "(x 393 216 B = 2.81 TB/s = 0.351 of 8 TB/s) at 1 293 W, sclk 2.39 GHz; shipped, same box, same 6 s hold: 6.05 M/s at "
                              "1 333-1 374 W, sclk 2.11-2.17 GHz (operands in the L2 alone: 7.07 M/s at 2.35 GHz -- what HBM traffic costs is clock)",
            "frac_of_ceiling": round(achieved / HBM_PEAK_GBS / ceil, 4)})
    # secondary ceiling (BASELINE.md section 4 asks for it beside the HBM roofline): VALU issue.  `peak` / `frac` are the
    # HARDWARE ceiling and do not depend on this kernel or this run: one wave64 VALU instruction per SIMD every 2 cycles
    # (MI355X_MICROARCH.md) at the nominal 2.4 GHz, 256 CUs x 4 SIMDs = 1 228.8 G wave-instructions/s.  Two tighter, still
    # kernel-independent prices ride along: `opcode_grid` = the isolated issue costs of tools/ubench_issue.hip (multiply /
    # carry / VOP3 opcodes 4.2 cycles, plain VOP2 2.5: the 62-bit butterfly's 10 + 8 mix = 62 cycles per 18 instructions) at
    # 2.4 GHz; and `fitted` = the FITTED figure of earlier rounds (3.86 cycles per instruction, the butterfly stream's own
    # measured cost, at the clock this run got under the 1 400 W package limit) -- by construction close to 1, kept only as
    # a consistency check of the instruction counts.
    # Instructions per product: dynamic counts of the generated kernels on the interpreter of tests/asm_emu.py
    # (tools/asm_cost.py -> profiles/r03_valu_issue_model.txt; B and A agree with their SQ counter passes).
    model = "profiles/r06_valu_issue_model.txt"
    valu = {"B": (90064, 1), "A": (1846, nm), "G": (97312, 1), "C": (835840, 1), "F": (447008, 1),
            "E": (14347680, 1), "H": (227, 1), "T": (13, 1)}.get(kwl)
    if valu:
        inst_per_poly = valu[0] * valu[1]
        nominal_ghz = 2.4
        peak_hw = 256 * 4 * nominal_ghz / 2.0            # G wave64-instructions/s
        ach_gi = inst_per_poly * batch / (kernel_ms * 1e-3) / 1e9
        sec = {"bound": "valu-issue", "achieved": round(ach_gi, 1), "peak": round(peak_hw, 1), "unit": "G wave64-inst/s",
               "frac": round(ach_gi / peak_hw, 4), "peak_is": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction",
               "wave_instructions_per_polymul": inst_per_poly}
        if lb == 64:
            grid_cpi = (10 * 4.2 + 8 * 2.5) / 18.0
            peak_grid = 256 * 4 * nominal_ghz / grid_cpi
            sec["opcode_grid"] = {"peak": round(peak_grid, 1), "frac": round(ach_gi / peak_grid, 4), "cycles_per_instruction": round(grid_cpi, 3),
                                  "clock_GHz": nominal_ghz, "source": "profiles/r03_ubench_issue.txt (isolated streams, all-VGPR operands)"}
        clock_ghz, cyc_per_inst = 2.0, 3.86
        if power and power.get("sclk_MHz"):
            clock_ghz = round(power["sclk_MHz"] / 1000.0, 3)    # the clock THIS run's kernel got (sampled above)
        peak_fit = 256 * 4 * clock_ghz / cyc_per_inst
        sec["fitted"] = {"kind": "fitted", "peak_at_measured_clock": round(peak_fit, 1), "frac": round(ach_gi / peak_fit, 4),
                        "clock_GHz": clock_ghz, "cycles_per_instruction": cyc_per_inst,
                        "source": model + ", profiles/r03_ubench_issue.txt, profiles/r03_power_clock.txt, profiles/r03_operand_ab.txt"}
        sec["power"] = power
        result["roofline"]["secondary"] = sec
    if sustained is not None:
        result["sustained"] = sustained
    if extras is not None:
        result["extras"] = extras
    if scatter is not None:
        result["scatter_gather"] = scatter
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(lb, n, nm, args.cpu_budget, check=check)
        except Exception as e:  # the checker must never take the bench down
            result["cpu_baseline"] = {"value": None, "unit": "polymul/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    # the headline number carries its own correctness evidence: a product that does not commute, or whose sampled
    # polynomials differ from the CPU checker's, is not a measurement
    bad = (not ok) or result.get("cpu_baseline", {}).get("parity_sample_ok") is False
    if bad:
        result["invalid_value"] = result["value"]
        result["value"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
