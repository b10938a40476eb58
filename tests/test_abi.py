"""CPU: the C-ABI library loads and exports every symbol include/nflhip.h declares;
without a GPU it fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "nflhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nflhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound():
    from nfllib_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libnflhip.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))
    assert lib.nflhip_abi_version() == _lib.ABI_VERSION == 6


def test_the_library_exports_exactly_what_its_two_headers_declare():
    """every exported C symbol is declared in include/nflhip.h (the boundary) or include/nflhip_debug.h (test hooks), and
    nothing else leaks out of the shared object"""
    import subprocess
    from nfllib_amd import _lib
    txt = open(os.path.join(ROOT, "include", "nflhip_debug.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    debug = set(re.findall(r"\b(nflhip_[a-z0-9_]+)\s*\(", txt))
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_Z")}
    exported = {e for e in exported if e not in ("_init", "_fini")}
    assert exported == set(_declared()) | debug, sorted(exported ^ (set(_declared()) | debug))
    assert len(debug) <= 6     # (round 6: + nflhip_debug_polymul_level, nflhip_debug_xcd_trace)


def test_the_library_reads_three_environment_variables():
    import subprocess
    from nfllib_amd import _lib
    strings = subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    knobs = sorted(set(re.findall(r"\bNFLHIP_[A-Z0-9_]+\b", strings)))
    assert knobs == ["NFLHIP_COMM_PIECE_BYTES", "NFLHIP_VARIANT", "NFLHIP_XCD"], knobs
    srcs = ""
    for f in os.listdir(os.path.join(ROOT, "nfllib_amd", "csrc")):
        if f.endswith((".hip", ".cpp", ".h")):
            srcs += open(os.path.join(ROOT, "nfllib_amd", "csrc", f)).read()
    assert srcs.count("getenv(") == 3


def test_no_torch_types_in_the_boundary():
    txt = open(os.path.join(ROOT, "include", "nflhip.h")).read()
    assert "torch" not in txt.lower() and "at::" not in txt and "#include <hip" not in txt


def _have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_have_gpu(), reason="CPU-only behaviour")
def test_fails_loudly_without_gpu():
    from nfllib_amd import Engine, NflHipError
    with pytest.raises(NflHipError) as ei:
        Engine(64, 4096, 4)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)


def test_argument_validation_precedes_device_use():
    from nfllib_amd import _lib
    from nfllib_amd.params import params
    lib = _lib.lib
    pr = params(64)
    h = C.c_void_p()
    P, R, K = [np.ascontiguousarray(x[:4]) for x in (pr.P, pr.primitive_roots, pr.invkmax)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    # degree not a power of two, bad limb width, NULL tables, degree > kMaxPolyDegree (core.hpp:59-60)
    assert lib.nflhip_ctx_create(C.byref(h), 0, 64, 4095, 4, vp(P), vp(R), vp(K), pr.kmax_log2) == 1
    assert lib.nflhip_ctx_create(C.byref(h), 0, 48, 4096, 4, vp(P), vp(R), vp(K), pr.kmax_log2) == 1
    assert lib.nflhip_ctx_create(C.byref(h), 0, 64, 4096, 4, None, vp(R), vp(K), pr.kmax_log2) == 1
    assert lib.nflhip_ctx_create(C.byref(h), 0, 64, 1 << 21, 4, vp(P), vp(R), vp(K), pr.kmax_log2) == 1
    assert b"kMaxPolyDegree" in lib.nflhip_last_error(None)
    assert lib.nflhip_ntt_fwd_dev(None, None, 1, None) == 1  # NULL ctx


def test_params_mirror_is_consistent():
    from nfllib_amd.params import params
    for lb in (16, 32, 64):
        pr = params(lb)
        for j in range(pr.max_moduli):
            p, r = int(pr.P[j]), int(pr.primitive_roots[j])
            assert p % (2 * pr.kmax) == 1 and p.bit_length() == pr.modulus_bits
            assert pow(r, pr.kmax, p) == p - 1
            assert int(pr.invkmax[j]) * pr.kmax % p == 1
            assert int(pr.Pn[j]) == ((1 << (2 * lb)) // p) % (1 << lb)
    hdr = open(os.path.join(ROOT, "include", "nflhip_params.h")).read()
    assert str(int(params(64).P[0])) in hdr and str(int(params(32).primitive_roots[3])) in hdr


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under nfllib_amd/ or include/ may reference it."""
    bad = []
    for base in ("nfllib_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"nfl_oracle|from oracle|import oracle|oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_product_path_never_touches_the_test_doubles():
    """tests/cpp/mock (the C ABI over host memory with toy arithmetic) and tools/hostprof (the do-nothing stand-in) exist for
    CPU tests / host profiling of the header layer only: nothing the product, the bench or smoke() loads may name them, and
    the library path the Python layer opens is the in-tree HIP build"""
    from nfllib_amd import _lib
    assert os.path.realpath(_lib.LIB_PATH) == os.path.realpath(os.path.join(ROOT, "nfllib_amd", "libnflhip.so"))
    bad = []
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for base in ("nfllib_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            files += [os.path.join(dp, f) for f in fs if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile"))]
    for f in files:
        txt = open(f, errors="ignore").read()
        if re.search(r"_mock|mock_backend|null_backend|hostprof/_build", txt):
            bad.append(f)
    assert not bad, bad


def test_every_generated_kernel_the_launchers_name_is_in_the_code_object():
    """kernels_fast.hip looks the generated assembly kernels up BY NAME in the embedded code object and treats a miss as
    "not supported" (the compiled kernels then serve the call): a typo would silently cost the tuned path.  Every name in
    kAsmNames must be a kernel symbol of the code object the Makefile links, and every kernel in it must be named."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "nfllib_amd", "csrc")
    src = open(os.path.join(csrc, "kernels_fast.hip")).read()
    block = src[src.index("kAsmNames[kAsmCount] = {"):]
    block = block[:block.index("};")]
    names = re.findall(r'"(nflhip_[a-z0-9_]+_asm)"', block)
    assert len(names) >= 30 and len(set(names)) == len(names)
    enum = src[src.index("enum AsmKind {"):]
    enum = enum[:enum.index("kAsmCount")]
    kinds = re.findall(r"\bkAsm[A-Za-z0-9]+\b", re.sub(r"//[^\n]*", "", enum))
    assert len(kinds) == len(names), "enum AsmKind and kAsmNames must list the same kernels in the same order"
    hsaco = os.path.join(csrc, "polymul4096_gfx950.hsaco")
    if not os.path.exists(hsaco):
        pytest.skip("code object not built here")
    tool = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(tool):
        pytest.skip("llvm-readelf not here")
    syms = subprocess.run([tool, "--symbols", "--wide", hsaco], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r"\bFUNC\s+GLOBAL\s+\S+\s+\d+\s+(nflhip_[a-z0-9_]+_asm)\b", syms))
    assert set(names) <= defined, sorted(set(names) - defined)
    assert defined <= set(names), sorted(defined - set(names))
