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
    assert len(r.stdout.strip().splitlines()) == 1, "stdout is exactly one line: " + r.stdout[:300]
    d = json.loads(r.stdout)
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "polymul/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] in ("u64", "u32") and "workload" in d["config"] and d["config"]["self_check"] is True
    assert d["value"] > 0 and abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.2
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # HBM traffic measured in the same run (rocprofv3 counter passes), within a few % of the algorithmic bytes for the
    # single-launch kernels
    assert rf["traffic"] is not None and "measured in this run" in rf["traffic_source"], rf["traffic_source"]
    # (at these tiny test batches the twiddle tables are a visible share: 2 MiB per 16 polynomials of workload C)
    assert 0.98 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.5
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert "median of 5" in cb["sample"] and cb["spread"][0] <= cb["value"] <= cb["spread"][1]
    assert cb["parity_sample_ok"] is True
    assert workload != "A" or "uint32_t" in d["config"]["workload"]
    if workload == "B":
        assert d["metric"].startswith("poly-mults/sec (NTT+pointwise+INTT), n=4096, 4x62-bit moduli")


@pytest.mark.gpu
def test_two_rank_launch_line_of_the_driver():
    """The driver's N > 1 command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...) on a 1-GPU box: both ranks share device 0 and the control
    collectives go over gloo (test knobs of bench.py); rank 0 alone prints the line, value is the whole-job rate, the
    scatter/gather leg moves the shards and reports beside it."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NFLHIP_BENCH_BACKEND="gloo", NFLHIP_BENCH_ONE_DEVICE="1", NFLHIP_BENCH_D_SHARD="1024")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--batch", "512", "--scatter-gather"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line"
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 1024
    assert abs(d["value"] - 2 * 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.2
    assert "cpu_baseline" not in d and set(d["extras"]) == {"configs"} and set(d["extras"]["configs"]) == {"D"}   # the rest: rank 0 at N = 1 only
    # BASELINE configs[3] in the N > 1 line: global batch = N x shard (2^17 per GPU on a real node; shrunk here, two ranks share a device)
    dd = d["extras"]["configs"]["D"]
    assert "error" not in dd, dd
    assert dd["n_gpus"] == 2 and dd["batch_per_gpu"] == 1024 and dd["global_batch"] == 2 * dd["batch_per_gpu"]
    assert dd["shard_is_baseline_shard"] is False and dd["is_baseline_config_4"] is False
    assert dd["self_check"] is True and dd["checksum_of_checksums"]["ok"] is True and dd["checksum_of_checksums"]["shards"] == 2
    assert abs(dd["value"] - dd["global_batch"] / (dd["ms_per_step"] * 1e-3)) / dd["value"] < 0.05
    sg = d["scatter_gather"]
    assert sg["polymul_per_s_incl_scatter_gather"] > 0 and sg["bytes_moved"] == 3 * 512 * 4 * 4096 * 8
    assert d["config"]["self_check"] is True


@pytest.mark.gpu
def test_eight_rank_rehearsal_of_the_scale_line():
    """The SCALE line the driver will launch on an 8-GPU node (--gpus 8 under torch.distributed.run), rehearsed with eight
    ranks on ONE device (gloo control plane, reduced shards): n_gpus 8, world size 8, configs[3]'s block with
    global batch = 8 x shard and a checksum of checksums over 8 shards -- so the first real 8-GPU run is not spent on a
    launch-line bug.  No rate is asserted: eight ranks share one GPU here."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NFLHIP_BENCH_BACKEND="gloo", NFLHIP_BENCH_ONE_DEVICE="1", NFLHIP_BENCH_D_SHARD="256")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3",
                        "--warmup", "1", "--batch", "256", "--prewarm", "0.1"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line"
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 * 256
    assert d["config"]["rccl"]["world_size"] == 8
    assert d["config"]["self_check"] is True and d["config"]["checksum_of_checksums"]["ok"] is True
    assert d["config"]["checksum_of_checksums"]["shards"] == 8
    assert abs(d["value"] - 8 * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.2
    dd = d["extras"]["configs"]["D"]
    assert "error" not in dd, dd
    assert dd["n_gpus"] == 8 and dd["batch_per_gpu"] == 256 and dd["global_batch"] == 8 * dd["batch_per_gpu"]
    assert dd["shard_is_baseline_shard"] is False and dd["is_baseline_config_4"] is False   # (the real shard is 2^17: 48 GiB per GPU)
    assert dd["self_check"] is True and dd["checksum_of_checksums"]["ok"] is True and dd["checksum_of_checksums"]["shards"] == 8
    assert dd["parity_sample_ok"] is True
    assert dd["preflight"]["ok"] is True and dd["preflight"]["needed_GiB"] < dd["preflight"]["free_GiB"]


@pytest.mark.gpu
def test_config_d_preflight_refuses_a_shard_that_does_not_fit():
    """the pre-flight of extras.configs.D: a shard whose three resident tensors exceed the device's free memory is refused
    with a reason in the block (the headline still prints), instead of an out-of-memory fault mid-run"""
    env = dict(os.environ, NFLHIP_BENCH_D_SHARD=str(1 << 22))      # 3 x 512 GiB
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "256", "--no-extras",
                        "--no-cpu-baseline", "--no-traffic", "--prewarm", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    dd = d["extras"]["configs"]["D"]
    assert "error" in dd and "pre-flight" in dd["error"] and d["value"] > 0


@pytest.mark.gpu
def test_plain_gpus_n_launches_n_ranks_itself_or_refuses():
    """`python bench.py --gpus 2` with NO launcher around it spawns the two ranks itself (torch.distributed.run on
    127.0.0.1) and reports n_gpus 2 -- here on one device through the test knobs; without the knob, on a box with fewer
    than N devices, it exits non-zero instead of printing an n_gpus the run did not have"""
    import torch
    env = dict(os.environ, NFLHIP_BENCH_BACKEND="gloo", NFLHIP_BENCH_ONE_DEVICE="1", NFLHIP_BENCH_D_SHARD="1024")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl"]["world_size"] == 2 and d["config"]["global_batch"] == 1024
    assert d["config"]["checksum_of_checksums"]["ok"] is True and d["config"]["checksum_of_checksums"]["shards"] == 2
    if torch.cuda.device_count() < 2:
        env.pop("NFLHIP_BENCH_ONE_DEVICE")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert "refusing" in r.stderr


@pytest.mark.gpu
def test_headline_line_carries_sustained_independent_secondary_and_side_configs():
    """workload B's line (what the driver records): `sustained` (the same launch held for seconds), `roofline.secondary`
    priced at the hardware issue rate with the fitted figure labelled as such, and extras.configs with configs C and E
    (polymul and CRT lift) timed and counter-measured inside the same run"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-budget", "1.0"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    sus = d["sustained"]
    assert sus["unit"] == "polymul/s" and sus["seconds"] > 1.5 and 0 < sus["frac"] < 1 and sus["value"] > 0
    sec = d["roofline"]["secondary"]
    assert sec["peak"] == 1228.8 and abs(sec["frac"] - sec["achieved"] / sec["peak"]) < 1e-3
    assert "model" not in sec and sec["fitted"]["kind"] == "fitted" and "peak_at_measured_clock" in sec["fitted"] and sec["opcode_grid"]["clock_GHz"] == 2.4
    # what binds, said in the line: HBM stays the declared roofline; the kernel is VALU-issue / power bound
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and "valu-issue" in rf["binding"] and rf["ceiling_frac_no_memory"] == 0.351 and "profiles/" in rf["ceiling_source"]
    assert abs(rf["frac_of_ceiling"] - rf["frac"] / rf["ceiling_frac_no_memory"]) < 2e-3 and rf["frac_of_ceiling"] < 1
    # BASELINE configs[3] for ONE shard: 2^17 polynomials resident on this GPU, its own clock and checksum of checksums
    dd = d["extras"]["configs"]["D"]
    assert "error" not in dd, dd
    assert dd["batch_per_gpu"] == 1 << 17 and dd["n_gpus"] == 1 and dd["global_batch"] == 1 << 17 and dd["shard_is_baseline_shard"] is True
    assert dd["is_baseline_config_4"] is False and dd["self_check"] is True and dd["checksum_of_checksums"]["ok"] is True
    assert dd["value"] > 0 and abs(dd["value"] - dd["global_batch"] / (dd["ms_per_step"] * 1e-3)) / dd["value"] < 0.05
    lwe = d["extras"]["lwe"]
    assert "error" not in lwe and lwe["same_ciphertexts"] is True and lwe["fused"]["decrypts_to_zero"] is True
    assert lwe["fused"]["encryptions_per_s"] > 1.3 * lwe["unfused"]["encryptions_per_s"]
    assert lwe["fused"]["decryptions_per_s"] > 1.3 * lwe["unfused"]["decryptions_per_s"]
    hdr = lwe["cpp_header"]      # the same demo through the drop-in header: plain poly_p operators and device_batch's fused methods
    assert "error" not in hdr and hdr["poly_p_encryptions_per_s"] > 20 * hdr["poly_p_eager_encryptions_per_s"]
    assert hdr["device_batch_fused_encryptions_per_s"] > 1.3 * hdr["device_batch_encryptions_per_s"]
    cfg = d["extras"]["configs"]
    rt = cfg["A"]["ntt_intt_round_trip"]        # BASELINE configs[0]'s operation on the device
    assert rt["value"] > 0 and rt["returns_the_input"] is True and "uint32_t,1024,1" in cfg["A"]["workload"]
    for wl in ("A", "G", "C", "F", "E"):   # (G = the reference's own test configuration (8192, 124, uint64_t))
        c = cfg[wl]
        assert "error" not in c, c
        assert c["self_check"] is True and c["value"] > 0 and 0 < c["frac"] < 1 and c["traffic_ratio"] is not None and c["traffic_ratio"] > 0.98
        # every block carries polynomials of ITS timed product recomputed by the CPU oracle (not only the headline)
        assert c["parity_sample_ok"] is True and "CPU oracle" in c["parity_sample"], c.get("parity_sample")
    assert dd["parity_sample_ok"] is True
    assert cfg["E"]["crt_parity_sample_ok"] is True and "oracle.crt_lift" in cfg["E"]["crt_parity_sample"]
    assert cfg["E"]["crt_lift"]["value"] > 0 and cfg["E"]["crt_lift"]["limbs_per_coefficient"] == 30
    one = cfg["E"]["polymul_plus_crt_lift"]     # BASELINE configs[4] as ONE figure: slower than either part, faster than their serial sum allows
    assert one["unit"] == "polys/s" and 0 < one["value"] < min(cfg["E"]["value"], cfg["E"]["crt_lift"]["value"])
    assert one["ms_per_step"] < 1.15 * (cfg["E"]["ms_per_step"] + cfg["E"]["crt_lift"]["ms_per_step"])
