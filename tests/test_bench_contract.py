"""The bench.py / __graft_entry__ contract: one JSON line with the agreed keys (metric of BASELINE.json, whole-job
value, roofline, cpu_baseline), CLI defaults, and -- without a GPU -- a loud failure instead of a CPU fallback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _gpu():
    import torch
    return torch.cuda.is_available()


def test_metric_string_is_the_baselines():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    src = open(os.path.join(ROOT, "bench.py")).read()
    # "poly-mults/sec (NTT+pointwise+INTT), n=4096, 4x62-bit moduli" (ASCII 'x' for the multiplication sign)
    want = base["metric"].split(", 1/2/4/8")[0].replace("×", "x")
    assert want in src


@pytest.mark.skipif(_gpu(), reason="CPU-only behaviour")
def test_bench_refuses_to_run_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,batch", [("B", 512), ("C", 16), ("A", 4096)])
def test_bench_line_contract(workload, batch):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload",
                        workload, "--batch", str(batch), "--cpu-budget", "1.0"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "polymul/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] in ("u64", "u32") and "workload" in d["config"] and d["config"]["self_check"] is True
    assert d["value"] > 0 and abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.2
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    if workload == "B":
        assert d["metric"].startswith("poly-mults/sec (NTT+pointwise+INTT), n=4096, 4x62-bit moduli")
