"""nfllib_amd -- MI355X-native NTT polynomial-ring engine (NFLlib-compatible).

Only what the hot path needs: `csrc/` (hand-written HIP kernels + the C ABI of
include/nflhip.h, built into libnflhip.so) and a thin host driver.  Importing
this package loads libnflhip.so and fails loudly when it is missing.
"""
from ._lib import NflHipError, lib  # noqa: F401
from .engine import (Engine, Comm, shard_range, OP_ADD, OP_COMPUTE_SHOUP, OP_MUL, OP_MUL_SHOUP, OP_SUB,  # noqa: F401
                     TAB_INVDEGREE, TAB_MODULUS, TAB_PSI, DIST_UNIFORM, DIST_BOUNDED, DIST_ZO, DIST_HWT)
from .params import params  # noqa: F401

__all__ = ["Engine", "Comm", "shard_range", "NflHipError", "params", "OP_ADD", "OP_SUB", "OP_MUL", "OP_MUL_SHOUP", "OP_COMPUTE_SHOUP",
           "DIST_UNIFORM", "DIST_BOUNDED", "DIST_ZO", "DIST_HWT"]
