#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter-collection CSVs into HBM bytes per polymul.

Run on the GPU box, separate passes per counter as MI355X_MICROARCH.md prescribes (no trace domains with --pmc):

    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_<wl>_$c -- \
          python bench.py --workload <wl> --steps 3 --warmup 1 --no-extras --no-cpu-baseline
    done
    python tools/pmc_traffic.py <wl> <batch> gpurun_out/pmc_<wl>_FETCH_SIZE gpurun_out/pmc_<wl>_WRITE_SIZE [products]

Counter units are KiB; on gfx950 FETCH_SIZE reports exactly half of the streamed bytes (calibrated with a copy
kernel: profiles/r01_pmc_*_calibration_ubench.csv), WRITE_SIZE is exact.  Only dispatches of the product's own
kernels are summed (per launch of the whole product = one step of bench.py).
"""
import csv
import glob
import json
import os
import sys

KERNELS = ("nflhip_polymul", "nflhip_row", "nflhip_ntt", "k_ntt_fwd_outer", "k_ntt_inv_outer", "k_row1024_u32", "k_row<", "k_row_block", "k_polymul4096")


def total(d, counter):
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not any(k in r["Kernel_Name"] for k in KERNELS):
                continue
            name = r["Kernel_Name"].split("(")[0].split("<")[0]
            e = per_kernel.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return per_kernel


def main():
    wl, batch, fdir, wdir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    alg = {"A": 3 * 1024 * 4, "B": 393216, "C": 3145728, "E": 47185920}[wl]
    f, w = total(fdir, "FETCH_SIZE"), total(wdir, "WRITE_SIZE")
    # launches of the product: the dominant (fused) kernel's dispatch count
    fused = max(f, key=lambda k: ("polymul" in k or "row1024" in k, f[k][0]))
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else f[fused][0]   # whole products profiled (a product may be several dispatches)
    fetch = sum(v[1] for v in f.values()) * 1024 * 2.0 / launches
    write = sum(v[1] for v in w.values()) * 1024 * 1.0 / launches
    out = {"workload": wl, "batch": batch, "launches": launches, "kernels": {k: v[0] for k, v in f.items()},
           "fetch_bytes_per_poly": round(fetch / batch, 1), "write_bytes_per_poly": round(write / batch, 1),
           "hbm_bytes_per_poly": round((fetch + write) / batch, 1), "algorithmic_bytes_per_poly": alg,
           "ratio": round((fetch + write) / batch / alg, 4),
           "calibration": "FETCH_SIZE x2 (gfx950 reports half), WRITE_SIZE x1; see profiles/pmc_traffic.json"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
