#!/usr/bin/env python3
"""Rewrites profiles/pmc_traffic.json (bench.py's fallback when rocprofv3 is missing) from the in-run traffic figures of
the committed bench lines: python tools/refresh_pmc_traffic.py r03_final"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
doc["collected_with"] = ("bench.py's in-run rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate), taken over from "
                         "profiles/%s_bench_<workload>.json by tools/refresh_pmc_traffic.py" % tag)
for wl in "BACEFGHTD":
    f = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, wl))
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).readline())
    r = d["roofline"]
    if not r.get("traffic") or "measured in this run" not in r.get("traffic_source", ""):
        continue
    batch = d["config"]["batch_per_gpu"]
    doc["workloads"][wl] = {
        "workload": wl, "batch": batch, "kernel": r.get("kernel"),
        "hbm_bytes_per_poly": round(r["traffic"] / batch, 1),
        "algorithmic_bytes_per_poly": r["algorithmic_bytes_per_launch"] // batch,
        "ratio": round(r["traffic"] / r["algorithmic_bytes_per_launch"], 4),
        "round": tag.split("_")[0], "source": "bench.py in-run counter passes (profiles/%s_bench_%s.json)" % (tag, wl)}
json.dump(doc, open(path, "w"), indent=1)
print("wrote", path, {k: v["ratio"] for k, v in doc["workloads"].items()})
