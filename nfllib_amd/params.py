"""Modulus parameter tables mirrored from the reference's ``params<T>``
(/root/reference include/nfl/params.hpp:11-119): the first K moduli per limb
type, produced (and re-derived / asserted) by tools/extract_params.py.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DTYPES = {16: np.uint16, 32: np.uint32, 64: np.uint64}
_NAMES = {16: "uint16_t", 32: "uint32_t", 64: "uint64_t"}

with open(os.path.join(_HERE, "data", "params.json")) as _f:
    _RAW = json.load(_f)


class LimbParams:
    """params<T> for one limb width (params.hpp:11-40 / 43-79 / 82-119)."""

    def __init__(self, limb_bits):
        d = _RAW[_NAMES[limb_bits]]
        self.limb_bits = limb_bits
        self.dtype = np.dtype(_DTYPES[limb_bits])
        self.modulus_bits = d["modulus_bits"]          # kModulusBitsize
        self.kmax = d["kmax"]                          # kMaxPolyDegree
        self.kmax_log2 = d["kmax"].bit_length() - 1
        self.P = np.array(d["P"], dtype=self.dtype)
        self.Pn = np.array(d["Pn"], dtype=self.dtype)
        self.primitive_roots = np.array(d["primitive_roots"], dtype=self.dtype)
        self.invkmax = np.array(d["invkmax"], dtype=self.dtype)
        self.max_moduli = len(d["P"])


_CACHE = {}


def params(limb_bits):
    if limb_bits not in _DTYPES:
        raise ValueError("limb_bits must be 16, 32 or 64")
    if limb_bits not in _CACHE:
        _CACHE[limb_bits] = LimbParams(limb_bits)
    return _CACHE[limb_bits]
