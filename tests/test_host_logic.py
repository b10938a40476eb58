"""Host logic of the header-only layer WITHOUT a GPU: the resident poly_p handles and the deferred queue of
include/nfl_hip/nfl.hpp (levelling by data dependencies, grouping by signature, key operands, stride runs, consecutive
buffers, the queue that runs by itself) executed against tests/cpp/mock -- the C ABI over host memory with toy arithmetic
that keeps the contracts between entry points (batch = loop of singles, strides, sequence stream ids) and in which no
two operations commute.  The programs compare deferred with immediate execution word for word; on the GPU the same
programs run against the real library (tests/test_cpp_surface.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "cpp", "_mock")


def build_program(src, out, include=os.path.join(ROOT, "include")):
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-I" + include, "-DNFL_HIP_NO_GMP", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", src), "-L" + MOCK, "-lnflhip", "-Wl,-rpath," + MOCK])


@pytest.fixture(scope="module")
def mock():
    os.makedirs(MOCK, exist_ok=True)
    c = os.path.join(MOCK, "mock_backend.c")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "cpp", "mock", "make_mock_backend.py"), c],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-o",
                           os.path.join(MOCK, "libnflhip.so"), c, "-lpthread"])
    from concurrent.futures import ThreadPoolExecutor
    asan = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    tsan = ["-O1", "-g", "-fsanitize=thread", "-pthread"]
    jobs = [("deferred_fuzz", "deferred_fuzz", None), ("deferred_loops", "deferred_loops", None), ("serialize_archive", "serialize_archive", None),
            ("deferred_fuzz", "fuzz_asan", asan), ("deferred_edges", "edges_asan", asan), ("deferred_fuzz", "deferred_fuzz_tsan", tsan),
            ("deferred_loops", "deferred_loops_tsan", tsan), ("deferred_threads", "threads_tsan", tsan)]

    def build(job):     # (the header is the slow part of each: the programs and their sanitizer builds compile side by side)
        src, out, flags = job
        if flags is None:
            return build_program(src + ".cpp", os.path.join(MOCK, out))
        subprocess.check_call(["g++", "-std=c++11"] + flags + ["-I" + os.path.join(ROOT, "include"), "-DNFL_HIP_NO_GMP", "-o", os.path.join(MOCK, out),
                               os.path.join(ROOT, "tests", "cpp", src + ".cpp"), "-L" + MOCK, "-lnflhip", "-Wl,-rpath," + MOCK])
    with ThreadPoolExecutor(6) as pool:
        list(pool.map(build, jobs))
    return MOCK


def run(exe, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=600)


def test_random_programs_deferred_equals_immediate(mock):
    from conftest import FUZZ_SEED
    for seed in (2024, FUZZ_SEED):      # the historical seed as a regression run + the tree's own (conftest.fuzz_seed)
        r = run(os.path.join(mock, "deferred_fuzz"), 60, seed)
        assert r.returncode == 0 and "all checks passed" in r.stdout, "NFL_FUZZ_SEED=%d\n" % seed + r.stdout + r.stderr


def test_the_archive_hook_writes_the_manual_image(mock):
    """poly::serialize(Archive &) / poly_p::serialize(Archive &) (poly.hpp:189-191; the reference's
    tests/poly_serialize_cereal.cpp) instantiated with a binary archive of cereal's calling convention: the bytes are
    serialize_manually's, and they read back (host logic only: the toy device serves the handles)"""
    r = run(os.path.join(mock, "serialize_archive"), "toy")
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("limit,thread,min_run", [(None, 1, None), (97, 1, None), (1, 1, None), (None, 0, None), (97, 0, None), (None, 1, 16), (50, 1, 7)])
def test_loop_shapes_deferred_equals_immediate(mock, limit, thread, min_run):
    """the LWE loop, strided / reversed / chained loops, handles dying queued; with the default queue length, with a queue
    that runs by itself every 97 records (in the middle of groups) and with one that holds a single record; runs executed by
    the queue's own thread (the default: a run that starts by itself is handed over and the program goes on recording) or by
    the recording thread (NFL_HIP_QUEUE_THREAD=0); hand-overs from 16 / 7 records on (NFL_HIP_QUEUE_MIN: the run lengths of a
    loop grow geometrically from there)"""
    env = {"NFL_HIP_QUEUE_THREAD": str(thread)}
    if limit is not None:
        env["NFL_HIP_QUEUE_LIMIT"] = str(limit)
    if min_run is not None:
        env["NFL_HIP_QUEUE_MIN"] = str(min_run)
    r = run(os.path.join(mock, "deferred_loops"), 300 if limit != 1 else 40, env=env)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    if thread and limit == 97:      # the run was really handed over (NFL_HIP_TRACE_DEFERRED names the executing thread)
        r = run(os.path.join(mock, "deferred_loops"), 60, env=dict(env, NFL_HIP_TRACE_DEFERRED="1"))
        assert "(the queue's thread)" in r.stderr and "all checks passed" in r.stdout


def test_runs_that_start_inside_an_iteration_leave_it_whole(mock):
    """lazy::clean_cut: a run that starts by itself while an iteration's Gaussian temporaries still have their handles leaves
    that iteration's records in the queue -- so every sample / transform / multiply-add sequence the one-run queue fuses is
    still fused when runs start every 97 or 61 records, on either thread"""
    def fused(env):
        r = run(os.path.join(mock, "deferred_loops"), 300, env=dict(env, NFL_HIP_TRACE_DEFERRED="1"))
        assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
        return sum(int(l.split(":")[2].split()[0]) for l in r.stderr.splitlines() if l.startswith("nfl(hip) deferred:") and " kind 6:" in l)
    whole = fused({"NFL_HIP_QUEUE_THREAD": "0", "NFL_HIP_QUEUE_LIMIT": "1000000"})
    assert whole >= 300
    for env in ({"NFL_HIP_QUEUE_THREAD": "0", "NFL_HIP_QUEUE_LIMIT": "97"}, {"NFL_HIP_QUEUE_THREAD": "1", "NFL_HIP_QUEUE_LIMIT": "97", "NFL_HIP_QUEUE_MIN": "16"},
                {"NFL_HIP_QUEUE_THREAD": "0", "NFL_HIP_QUEUE_LIMIT": "61"}):
        assert fused(env) == whole, env


def test_early_queue_runs_do_not_change_results(mock):
    """NFL_HIP_EARLY_RUN=1: from 1 024 records on the queue runs whenever the stream is idle (always, on the toy device):
    runs start in the middle of loop iterations and of groups"""
    for exe, args in (("deferred_loops", (700,)), ("deferred_fuzz", (30, 77))):
        r = run(os.path.join(mock, exe), *args, env={"NFL_HIP_EARLY_RUN": "1"})
        assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


def test_the_loops_are_coalesced(mock):
    """what deferral is for: the LWE loop's 300 x 10 operations leave as a handful of launches -- with the transform fusion
    (the default where the context has the fused kernels: the toy device says it does) as 300 forward multiply-adds in
    four launches (three compact sampler launches + the fused one) and 300 multiply-subtract-inverse operations in one;
    without it (NFL_HIP_NO_FUSION=1) operator by operator, as recorded.  (One queue run for the whole loop: the queue's own thread,
    which would start on the loop's first 2 048 records while the rest is being recorded, is switched off.)"""
    r = run(os.path.join(mock, "deferred_loops"), 300, env={"NFL_HIP_TRACE_DEFERRED": "1", "NFL_HIP_NO_FUSION": "1", "NFL_HIP_QUEUE_THREAD": "0"})
    assert r.returncode == 0
    lines = [l for l in r.stderr.splitlines() if l.startswith("nfl(hip) deferred:")]
    first_ring = lines[:7]      # level 0: gaussians (3 groups), level 1: transforms, 2: products, 3: decryption, 4: inverse
    ops = sum(int(l.split(":")[2].split()[0]) for l in first_ring)
    launches = sum(int(l.split("->")[1].split()[0]) for l in first_ring)
    assert ops >= 300 * 10 and launches <= 12, first_ring
    r = run(os.path.join(mock, "deferred_loops"), 300, env={"NFL_HIP_TRACE_DEFERRED": "1", "NFL_HIP_QUEUE_THREAD": "0"})
    assert r.returncode == 0 and "all checks passed" in r.stdout
    lines = [l for l in r.stderr.splitlines() if l.startswith("nfl(hip) deferred:")]
    fused = [l for l in lines[:6] if " kind 6:" in l or " kind 7:" in l]       # K_FWD_FMA, K_FMA_INV
    assert len(fused) == 2, lines[:8]
    assert all(int(l.split(":")[2].split()[0]) == 300 for l in fused), fused
    assert sum(int(l.split("->")[1].split()[0]) for l in fused) <= 5, fused
    assert sum(int(l.split("->")[1].split()[0]) for l in lines[:6]) <= 11, lines[:6]


def test_the_harness_notices_a_broken_queue(mock, tmp_path):
    """a header whose levelling forgets write-after-read hazards must fail the comparison (the toy operations do not
    commute): proof that the mock keeps what the queue's correctness depends on"""
    inc = tmp_path / "include"
    shutil.copytree(os.path.join(ROOT, "include"), inc)
    hdr = inc / "nfl_hip" / "queue.hpp"      # (the deferred queue's part of the split header)
    text = hdr.read_text()
    good = "L = std::max(L, std::max(wlev[o.out_pin], rlev[o.out_pin]) + 1);"
    assert good in text
    hdr.write_text(text.replace(good, "L = std::max(L, wlev[o.out_pin] + 1);"))
    exe = str(tmp_path / "fuzz_mutant")
    build_program("deferred_fuzz.cpp", exe, include=str(inc))
    r = run(exe, 40, 1)
    assert r.returncode != 0 and "all checks passed" not in r.stdout


def _mutants_fail(tmp_path, edits, tag, also=None):
    """builds deferred_loops.cpp against copies of include/ with ONE line of queue.hpp changed each (side by side) and expects
    every one of them to fail the loop comparison"""
    from concurrent.futures import ThreadPoolExecutor

    def build(k):
        old, new = edits[k]
        inc = tmp_path / ("include%d" % k)
        shutil.copytree(os.path.join(ROOT, "include"), inc)
        hdr = inc / "nfl_hip" / "queue.hpp"      # (the deferred queue's part of the split header)
        text = hdr.read_text()
        assert old in text
        text = text.replace(old, new)
        if also and k in also:
            assert also[k][0] in text
            text = text.replace(*also[k])
        hdr.write_text(text)
        exe = str(tmp_path / ("%s_mutant%d" % (tag, k)))
        build_program("deferred_loops.cpp", exe, include=str(inc))
        return exe
    with ThreadPoolExecutor(2) as pool:
        exes = list(pool.map(build, range(len(edits))))
    for exe in exes:
        r = run(exe, 60)
        assert r.returncode != 0 and "all checks passed" not in r.stdout, exe


def test_the_harness_notices_a_broken_fusion(mock, tmp_path):
    """a header whose transform fusion forgets that a sampled-and-transformed temporary may still have a handle (or another
    reader) must fail the loop comparison: proof that the look-alike shapes of deferred_loops.cpp bite"""
    _mutants_fail(tmp_path, [("      return fw[k] != d || r.dead[k] != 0;", "      return true;"), ("uses[size_t(dn)] != want_uses || ", "")], "loops",
                  also={1: ("if (ux != 1 && ux != 2) continue;", "if (ux < 1) continue;")})


def test_the_harness_notices_a_transform_joined_too_eagerly(mock, tmp_path):
    """a header whose transforms join the producing record although the value was read in between, or although that record
    already carries a transform, must fail the loop comparison (section 5 of deferred_loops.cpp)"""
    _mutants_fail(tmp_path, [("p->rec_r > p->rec_w || ", ""), ("if (t.post || t.out != p || ", "if (t.out != p || ")], "join")


def test_random_programs_under_address_and_undefined_behaviour_sanitizers(mock, tmp_path):
    """the same random programs with ASan + UBSan + leak detection: handles dying while queued, the queue's raw payload
    pointers and its one-reference-per-run pins, the buffer pool's free lists"""
    exe = os.path.join(mock, "fuzz_asan")       # (built by the fixture, beside the others)
    r = run(exe, 12, 99, env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "NFL_HIP_QUEUE_LIMIT": "61"})
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_the_queue_thread_and_one_recording_thread_under_thread_sanitizer(mock, tmp_path):
    """the hand-over protocol of the queue's own thread (lazy::take / post / execute / collect / retire) under ThreadSanitizer:
    one recording thread, runs handed over every few records (NFL_HIP_QUEUE_MIN=5, limit 40), random programs and the loop
    shapes: no data race -- a run works on the per-run arrays take() filled and never touches a payload -- and deferred ==
    immediate.  (NFL_HIP_NO_BIASED_LOCK: the buffer pool's lock is taken by both threads; its membarrier-based bias is not
    something ThreadSanitizer can see through.)"""
    programs = (("deferred_fuzz", (25, 31)), ("deferred_loops", (120,)))
    for name, args in programs:
        exe = os.path.join(mock, name + "_tsan")
        r = run(exe, *args, env={"NFL_HIP_QUEUE_THREAD": "1", "NFL_HIP_QUEUE_MIN": "5", "NFL_HIP_QUEUE_LIMIT": "40", "NFL_HIP_NO_BIASED_LOCK": "1",
                                 "TSAN_OPTIONS": "halt_on_error=0"})
        assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]


def test_threads_with_their_own_handles_share_the_queue_safely(mock, tmp_path):
    """several host threads, each on its own handles: the queue, the stream and the buffer pool are shared, and a queue
    run started by one thread retires the others' operations.  Under ThreadSanitizer, with a queue that runs by itself
    every 37 records: no data race (the copy-on-write test reads the queue's reference and its flag under the queue's
    lock) and every thread's results equal those of the same program run alone."""
    exe = os.path.join(mock, "threads_tsan")
    for _ in range(3):
        r = run(exe, 6, 400, env={"NFL_HIP_QUEUE_LIMIT": "37", "TSAN_OPTIONS": "halt_on_error=0"})
        assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]


def test_generators_that_die_early_key_changes_and_failing_launches(mock, tmp_path):
    """tests/cpp/deferred_edges.cpp under ASan + UBSan: a FastGaussianNoise destroyed before the polynomials built from it
    are used (its device table must outlive the recorded draws), nfl::set_sampler_key between a random constructor and the
    queue run (deferred == immediate), and a launch that fails in the middle of a queue run (injected by the CPU stand-in):
    what ran keeps its value, what never ran throws on access, an overwritten handle is usable again; the same inside a run
    that started by itself on the queue's own thread (what was recorded while it was in flight is poisoned too, the queue works
    again afterwards), and fork() with that thread alive (the child runs its queue itself)"""
    exe = os.path.join(mock, "edges_asan")
    for limit in (None, "64"):
        r = run(exe, env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", **({"NFL_HIP_QUEUE_LIMIT": limit} if limit else {})})
        assert r.returncode == 0 and "with failure injection" in r.stdout and "all checks passed" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_pipeline_xcd_remap_is_a_bijection_and_its_one_multiply_division_is_exact():
    """n = 65536 pipeline kernel (tools/gen_polymul_asm.py build_pipe, kernels_fast.hip launch_polymul_pipe64k_u64): the workgroup
    with linear index L = wgx + gx * cm takes unit u = (L mod 8) * U/8 + L div 8 of the modulus-major order and recovers
    (cm, wgx) = (u div gx, u mod gx) with ONE 32-bit multiply-high by ceil(2^32 / gx).  The launcher switches the remap on
    only if U = gx * nm is a multiple of 8 and U * gx < 2^32: under exactly that condition the division must be exact and the
    map a bijection onto the (wgx, cm) grid, and XCD slot k must get a contiguous range of units."""
    import random
    rnd = random.Random(5)
    cases = [(28 * c, nm) for c in (1, 2, 8, 32, 256, 1024) for nm in (1, 2, 3, 6, 8, 30, 64)]
    cases += [(28 * rnd.randrange(1, 600), rnd.randrange(1, 64)) for _ in range(300)]
    checked = 0
    for gx, nm in cases:
        units = gx * nm
        if units % 8 or units * gx >= 1 << 32:
            continue                      # (the launcher leaves the plain grid in place)
        per, magic = units // 8, (1 << 32) // gx + 1
        sample = range(units) if units <= 20000 else sorted({rnd.randrange(units) for _ in range(4000)} | {0, units - 1, per - 1, per})
        seen = set()
        for L in sample:
            u = (L & 7) * per + (L >> 3)
            cm = (u * magic) >> 32
            wgx = u - cm * gx
            assert cm == u // gx and 0 <= wgx < gx and cm < nm, (gx, nm, L)
            seen.add((wgx, cm))
        assert len(seen) == len(sample)
        if units <= 20000:
            assert seen == {(x, c) for x in range(gx) for c in range(nm)}
            for k in range(8):            # XCD slot k walks through a contiguous, modulus-major range
                us = sorted((L & 7) * per + (L >> 3) for L in range(k, units, 8))
                assert us == list(range(k * per, (k + 1) * per))
        checked += 1
    assert checked > 60
