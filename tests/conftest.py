import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x4E464C6C6962  # "NFLlib": the primary seed of SURVEY.md 8(d)


def fuzz_seed():
    """the seed of every randomised test (tests/cpp/deferred_fuzz, tests/test_gpu_fuzz.py): NFL_FUZZ_SEED when set, otherwise
    derived from the tree under test (the commit hash where there is a .git, else a digest of the product sources: the GPU
    box gets a snapshot without history) -- a new tree explores new programs, one tree always the same ones.  Failing tests
    print it; NFL_FUZZ_SEED=<that> reproduces them."""
    v = os.environ.get("NFL_FUZZ_SEED")
    if v:
        return int(v, 0) & 0x7FFFFFFF
    import hashlib
    import subprocess
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        if head.returncode == 0 and head.stdout.strip():
            return int(head.stdout.strip()[:8], 16) & 0x7FFFFFFF
    except Exception:
        pass
    h = hashlib.sha256()
    for rel in ("include/nflhip.h", "nfllib_amd/csrc/api.hip", "nfllib_amd/csrc/kernels_fast.hip", "nfllib_amd/csrc/kernels_generic.hip"):
        try:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        except OSError:
            pass
    for d in ("include/nfl_hip",):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            h.update(open(os.path.join(ROOT, d, f), "rb").read())
    return int(h.hexdigest()[:8], 16) & 0x7FFFFFFF


FUZZ_SEED = fuzz_seed()


def pytest_report_header(config):
    return "NFL_FUZZ_SEED=%d (randomised tests; set the variable to reproduce a run)" % FUZZ_SEED


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_factory():
    """TEST INFRASTRUCTURE: the CPU oracle (oracle/), keyed by shape."""
    from nfllib_amd.params import params
    from oracle import oracle as O
    cache = {}

    def get(limb_bits, degree, nmoduli):
        key = (limb_bits, degree, nmoduli)
        if key not in cache:
            cache[key] = O.Oracle(limb_bits, degree, nmoduli, params(limb_bits))
        return cache[key]

    return get


@pytest.fixture(scope="session")
def engine_factory():
    from nfllib_amd import Engine
    cache = {}

    def get(limb_bits, degree, nmoduli):
        key = (limb_bits, degree, nmoduli)
        if key not in cache:
            cache[key] = Engine(limb_bits, degree, nmoduli, device=0)
        return cache[key]

    yield get
    for e in cache.values():
        e.close()


@pytest.fixture(scope="session")
def compiled_engine_factory():
    """contexts that were created under NFLHIP_VARIANT=hipcc: every call is served by the compiled (hipcc) kernels -- the
    independent cross-check of the generated assembly kernels (the variable is read once, at context creation)"""
    from nfllib_amd import Engine
    cache = {}

    def get(limb_bits, degree, nmoduli):
        key = (limb_bits, degree, nmoduli)
        if key not in cache:
            saved = os.environ.get("NFLHIP_VARIANT")
            os.environ["NFLHIP_VARIANT"] = "hipcc"
            try:
                cache[key] = Engine(limb_bits, degree, nmoduli, device=0)
            finally:
                if saved is None:
                    os.environ.pop("NFLHIP_VARIANT", None)
                else:
                    os.environ["NFLHIP_VARIANT"] = saved
        return cache[key]

    yield get
    for e in cache.values():
        e.close()
