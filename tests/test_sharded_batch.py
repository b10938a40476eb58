"""The batch split in the PRODUCT: nfl::sharded_batch<P> of include/nfl_hip/nfl.hpp cuts a dense array
of polynomials (tests/tools.h:6-17 of the reference) into contiguous shards over the GPUs of one node from one C++
process, over the C ABI's per-device contexts, peer copies and shard-composable digests (include/nflhip.h "multi-GPU").

CPU: tests/cpp/sharded_main.cpp against tests/cpp/mock with EIGHT virtual devices -- toy arithmetic, but every buffer
belongs to one device and every operation checks that its pointers belong to its own context's device, so a shard
enqueued on the wrong context, a slice cut at the wrong polynomial or a keystream read at the wrong position fails.
GPU: the same program against the real library (several shards on the box's one device, and every visible device)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
MOCK = os.path.join(CPP, "_mock")


def _build_against_mock(out, include=os.path.join(ROOT, "include")):
    os.makedirs(MOCK, exist_ok=True)
    c = os.path.join(MOCK, "mock_backend.c")
    subprocess.check_call([sys.executable, os.path.join(CPP, "mock", "make_mock_backend.py"), c], stdout=subprocess.DEVNULL)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-o",
                           os.path.join(MOCK, "libnflhip.so"), c, "-lpthread"])
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + include, "-DNFL_HIP_NO_GMP", "-o", out,
                           os.path.join(CPP, "sharded_main.cpp"), "-L" + MOCK, "-lnflhip", "-Wl,-rpath," + MOCK])
    return out


@pytest.fixture(scope="module")
def mock_exe():
    return _build_against_mock(os.path.join(MOCK, "sharded_test"))


def _run(exe, *args, devices=8):
    env = dict(os.environ, NFLHIP_MOCK_DEVICES=str(devices))
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=600)


@pytest.mark.parametrize("devs,batch,ndev", [
    ("0,1,2,3,4,5,6,7", 37, 8),      # 8 virtual GPUs, a batch that does not divide
    ("0,1,2,3,4,5,6,7", 64, 8),      # ... one that does
    ("0,1,2,3,4,5,6,7", 3, 8),       # more devices than polynomials: empty shards
    ("5,0,3", 10, 8),                # any subset, any order; the whole batch lives on device 5
    ("0,0,0", 10, 1),                # several shards on one device (what a 1-GPU box can run)
    ("0", 9, 1),                     # the degenerate split
])
def test_sharded_batch_equals_one_device_batch_on_virtual_devices(mock_exe, devs, batch, ndev):
    r = _run(mock_exe, devs, batch, devices=ndev)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_a_device_that_does_not_exist_is_an_exception(mock_exe):
    r = _run(mock_exe, "0,1,2", 5, devices=2)
    assert r.returncode == 2 and "device index out of range" in r.stdout


@pytest.mark.parametrize("good,bad", [
    # a fan-out that hands shard r the neighbour's operand
    ("shards_[r].assign_polymul(a.shards_[r], b.shards_[r]);", "shards_[r].assign_polymul(a.shards_[r], b.shards_[(r + 1) % shards()]);"),
    # in-place generation that forgets the shard's offset in the logical batch
    ("if (u.seeded) shards_[r].set(u, first_[r]);", "if (u.seeded) shards_[r].set(u, 0);"),
    # digests that count positions per shard instead of per batch
    ("d[r] = count(r) ? shards_[r].digest(first_[r]) : 0;", "d[r] = count(r) ? shards_[r].digest(0) : 0;"),
])
def test_the_virtual_devices_notice_a_broken_split(tmp_path, good, bad):
    """mutants of the header must fail: proof that the CPU stand-in keeps what the split's correctness depends on"""
    inc = tmp_path / "include"
    shutil.copytree(os.path.join(ROOT, "include"), inc)
    hdr = inc / "nfl_hip" / "batch.hpp"      # (nfl::sharded_batch lives in the batch part of the split header)
    text = hdr.read_text()
    assert good in text
    hdr.write_text(text.replace(good, bad))
    exe = _build_against_mock(str(tmp_path / "sharded_mutant"), include=str(inc))
    r = _run(exe, "0,1,2,3,4,5,6,7", 37)
    assert r.returncode != 0 and "all checks passed" not in r.stdout, r.stdout[-2000:]


def test_builds_against_the_real_library():
    subprocess.check_call(["make", "-s", "-C", CPP, "sharded_test"])


@pytest.mark.gpu
@pytest.mark.parametrize("devs,batch", [("0,0,0", 37), ("0", 16), ("all", 41)])
def test_sharded_batch_equals_one_device_batch_on_the_gpu(devs, batch):
    """the same program, real arithmetic: three shards on device 0, the degenerate split, and every visible device"""
    subprocess.check_call(["make", "-s", "-C", CPP, "sharded_test"])
    if devs == "all":
        import torch
        devs = ",".join(str(i) for i in range(torch.cuda.device_count()))
    r = subprocess.run([os.path.join(CPP, "sharded_test"), devs, str(batch), "real"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
