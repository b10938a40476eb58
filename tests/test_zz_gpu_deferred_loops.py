"""The loop shapes of tests/cpp/deferred_loops.cpp (LWE loop, strided / reversed / chained loops, handles dying queued)
against the REAL library: deferred == immediate, word for word.  The same program runs on the CPU against the toy
arithmetic of tests/cpp/mock (tests/test_host_logic.py).  (Sorted last on purpose: it is the newest GPU test.)"""
import os
import subprocess

import pytest

CPP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [None, 97])
def test_loop_shapes_deferred_equals_immediate_on_the_gpu(limit):
    subprocess.check_call(["make", "-s", "-C", CPP, "deferred_loops"])
    env = dict(os.environ)
    if limit is not None:
        env["NFL_HIP_QUEUE_LIMIT"] = str(limit)
    r = subprocess.run([os.path.join(CPP, "deferred_loops"), "200"], capture_output=True, text=True, timeout=1800, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_threads_with_their_own_handles_on_the_gpu():
    """tests/cpp/deferred_threads.cpp against the real library: four host threads on their own poly_p handles (shared
    queue, stream and buffer pool); every thread's results equal the same program run alone"""
    subprocess.check_call(["make", "-s", "-C", CPP, "deferred_threads"])
    env = dict(os.environ)
    env["NFL_HIP_QUEUE_LIMIT"] = "61"
    r = subprocess.run([os.path.join(CPP, "deferred_threads"), "4", "200"], capture_output=True, text=True, timeout=1800, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_generators_that_die_early_and_key_changes_on_the_gpu():
    """tests/cpp/deferred_edges.cpp against the real library: a FastGaussianNoise that dies before its polynomials are
    used, set_sampler_key between a constructor and the queue run -- deferred == immediate"""
    subprocess.check_call(["make", "-s", "-C", CPP, "deferred_edges"])
    r = subprocess.run([os.path.join(CPP, "deferred_edges")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_deferred_poly_p_products_against_the_cpu_checker(tmp_path, oracle_factory):
    """the reference's product sequence on resident handles in a loop (queued, levelled, coalesced into batched launches)
    against the ORACLE, not against another HIP path: operand i is nfl::uniform(seed + i), which the checker regenerates"""
    import numpy as np
    from conftest import SEED
    subprocess.check_call(["make", "-s", "-C", CPP, "deferred_product"])
    K, out = 40, str(tmp_path / "products.bin")
    r = subprocess.run([os.path.join(CPP, "deferred_product"), out, str(K), str(SEED)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = open(out, "rb").read()
    off = 0
    for lb, n, m in ((64, 4096, 4), (32, 1024, 2)):
        o = oracle_factory(lb, n, m)
        dt = {64: np.uint64, 32: np.uint32}[lb]
        words = K * n * m
        got = np.frombuffer(raw, dtype=dt, count=words, offset=off).reshape(K, m, n)
        off += words * (lb // 8)
        a = np.concatenate([o.fill_uniform(1, SEED + 2 * i, 0) for i in range(K)])
        b = np.concatenate([o.fill_uniform(1, SEED + 2 * i + 1, 0) for i in range(K)])
        assert np.array_equal(got, o.polymul(a, b)), (lb, n, m)
    assert off == len(raw)
