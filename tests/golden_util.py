"""Shared loader for tests/golden (fixtures generated from the REAL reference by tools/gen_golden.py)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = ["ntt", "intt", "add", "sub", "mul", "compute_shoup", "mul_shoup", "polymul"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        index = json.load(f)
    arrays = np.load(os.path.join(HERE, "golden", "golden.npz"))
    return index, arrays


def shape_keys(index, mode=None, max_words=None):
    out = []
    for key, ent in index["shapes"].items():
        if mode and ent["mode"] != mode:
            continue
        if max_words and ent["degree"] * ent["nmoduli"] > max_words:
            continue
        out.append(key)
    return out


def edge_inputs(P, dtype, n, m):
    zero = np.zeros((1, m, n), dtype=dtype)
    pm1 = np.stack([np.full(n, P[cm] - 1, dtype=dtype) for cm in range(m)])[None]
    imp0 = zero.copy(); imp0[0, :, 0] = 1
    impl = zero.copy(); impl[0, :, n - 1] = 1
    x1 = zero.copy(); x1[0, :, 1 % n] = 1
    return {"zero": zero, "pm1": pm1, "imp0": imp0, "implast": impl, "x1": x1}


def compute_all(impl, a, b, OP):
    """impl: object with ntt/intt/pointwise/polymul taking & returning numpy arrays."""
    bp = impl.pointwise(OP["COMPUTE_SHOUP"], b)
    return {
        "ntt": impl.ntt(a), "intt": impl.intt(a),
        "add": impl.pointwise(OP["ADD"], a, b), "sub": impl.pointwise(OP["SUB"], a, b),
        "mul": impl.pointwise(OP["MUL"], a, b), "compute_shoup": bp,
        "mul_shoup": impl.pointwise(OP["MUL_SHOUP"], a, b, bp), "polymul": impl.polymul(a, b),
    }


def wide_integers(n, Lw):
    return np.random.default_rng(7).integers(0, 2**63, size=(1, n, Lw), dtype=np.uint64) * np.uint64(2) + \
        np.random.default_rng(8).integers(0, 2, size=(1, n, Lw), dtype=np.uint64)
