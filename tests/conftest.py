import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x4E464C6C6962  # "NFLlib": the primary seed of SURVEY.md 8(d)


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
