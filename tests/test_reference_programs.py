"""The drop-in gate: the REFERENCE's own test programs, unchanged, against this repository's headers and library.

north_star: "keeps the nfl::poly<T, Degree, NbModuli> template surface and expression-template operators so it drops
into existing callers".  tests/reftests/Makefile generates the `#define CONFIG ...` + `#include "<test>.cpp"` wrappers
of /root/reference/tests/CMakeLists.txt:1-77 for every test source x the reference's five configs (plus prng_demo,
ntt_perfs and the two-TU ntt_multi), compiles them with `-I include` only and links them against libnflhip.so.

  * build container (has /root/reference): all 63 programs must compile and link   -- `-m "not gpu"`
  * GPU box (no /root/reference; the binaries travel under tests/_reftests/, git-ignored): every program must
    exit 0 -- the reference's own pass criterion (tests/CMakeLists.txt add_test run_*)        -- `-m gpu`
Nothing of the reference is copied into the repository: the wrappers only #include it where it lies.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MK = os.path.join(ROOT, "tests", "reftests")
OUT = os.path.join(ROOT, "tests", "_reftests")
REF = "/root/reference/tests"
CONFIGS = ["8_60_uint32_t", "128_14_uint16_t", "1024_60_uint32_t", "8192_124_uint64_t", "32768_124_uint64_t"]
PERCFG = ["nfllib_demo_main_op", "nfllib_demo_main_func", "nfl_add", "nfl_sub", "nfl_mul", "nfl_eq", "nfl_neq", "nfl_stream",
          "poly_p", "poly_set", "poly_mpz", "poly_serialize_manually"]
PROGRAMS = ["%s__%s" % (t, c) for c in CONFIGS for t in PERCFG] + ["prng_demo", "ntt_perfs", "ntt_multi"]


def _gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(not os.path.isdir(REF), reason="build-container only: needs the reference's test sources")
def test_every_reference_test_program_compiles_and_links_unchanged():
    # 16 sources (poly_serialize_cereal is skipped exactly as the reference's CMake skips it without cereal), 5 configs
    sources = {f for f in os.listdir(REF) if f.endswith(".cpp")}
    covered = {t + ".cpp" for t in PERCFG} | {"prng_demo_main.cpp", "ntt_perfs.cpp", "multi0.cpp", "multi1.cpp"}
    assert sources - covered == {"poly_serialize_cereal.cpp"}, sources - covered
    r = subprocess.run(["make", "-s", "-j8", "-C", MK], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    missing = [p for p in PROGRAMS if not os.path.exists(os.path.join(OUT, p))]
    assert not missing, missing
    assert len(PROGRAMS) == 63


@pytest.mark.skipif(_gpu(), reason="CPU-only behaviour")
@pytest.mark.skipif(not os.path.exists(os.path.join(OUT, "nfl_add__8_60_uint32_t")), reason="programs not built")
def test_reference_program_fails_loudly_without_a_gpu():
    r = subprocess.run([os.path.join(OUT, "nfl_add__8_60_uint32_t")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGRAMS)
def test_reference_program_passes_on_the_device(prog):
    exe = os.path.join(OUT, prog)
    if not os.path.exists(exe):
        pytest.fail("tests/_reftests/%s is missing: run __graft_entry__.build() in the build container" % prog)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1200, cwd=OUT)
    assert r.returncode == 0, "%s exited %d\n%s\n%s" % (prog, r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    if prog.startswith("nfllib_demo"):
        assert "ERROR" not in r.stdout
