"""ctypes bindings of the CPU oracle (oracle/libnfloracle.so) and, when it was
built, of the REAL reference (oracle/_ref/libnflref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never from nfllib_amd/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DT = {16: np.uint16, 32: np.uint32, 64: np.uint64}

OP_ADD, OP_SUB, OP_MUL, OP_MUL_SHOUP, OP_COMPUTE_SHOUP = range(5)
(TAB_PHIS, TAB_SHOUPPHIS, TAB_INVPOLY_INVPHIS, TAB_SHOUPINVPOLY_INVPHIS, TAB_OMEGAS, TAB_INVOMEGAS,
 TAB_INVPOLYDEGREE) = range(7)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _load(path):
    lib = C.CDLL(path)
    return lib


def build(native_out=None):
    """(Re)build libnfloracle.so (and _ref when /root/reference exists)."""
    import subprocess
    import sys
    # (anything make prints goes to stderr: bench.py's stdout is exactly one JSON line)
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"], stdout=sys.stderr)
    if native_out:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", "OUT=" + native_out], stdout=sys.stderr)


class Oracle:
    """One (limb_bits, degree, nmoduli) instance of the CPU oracle."""

    def __init__(self, limb_bits, degree, nmoduli, params, libpath=None):
        path = libpath or os.path.join(_HERE, "libnfloracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = _load(path)
        L.nfl_oracle_create.restype = C.c_void_p
        L.nfl_oracle_create.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int]
        L.nfl_oracle_destroy.argtypes = [C.c_void_p]
        L.nfl_oracle_table.restype = C.c_void_p
        L.nfl_oracle_table.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        for name in ("nfl_oracle_ntt_pow_phi", "nfl_oracle_invntt_pow_invphi"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            getattr(L, name).restype = None
        L.nfl_oracle_ntt_row.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.nfl_oracle_ntt_row.restype = None
        L.nfl_oracle_pointwise.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_size_t]
        L.nfl_oracle_pointwise.restype = None
        L.nfl_oracle_polymul.argtypes = [C.c_void_p] * 4 + [C.c_size_t]
        L.nfl_oracle_polymul.restype = None
        L.nfl_oracle_polymul_mt.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_int]
        L.nfl_oracle_polymul_mt.restype = None
        for name in ("nfl_oracle_any_eq", "nfl_oracle_any_neq"):
            getattr(L, name).argtypes = [C.c_void_p] * 3 + [C.c_size_t]
            getattr(L, name).restype = C.c_int
        for name in ("nfl_oracle_crt_limbs", "nfl_oracle_crt_bits", "nfl_oracle_crt_shift"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        for name in ("nfl_oracle_crt_modulus", "nfl_oracle_crt_modulus_shoup"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            getattr(L, name).restype = C.c_size_t
        L.nfl_oracle_crt_lifting.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.nfl_oracle_crt_lifting.restype = C.c_size_t
        L.nfl_oracle_crt_lift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.nfl_oracle_crt_lift.restype = None
        L.nfl_oracle_crt_project.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.nfl_oracle_crt_project.restype = None
        L.nfl_oracle_fill_uniform.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int]
        L.nfl_oracle_fill_uniform.restype = None

        self.limb_bits, self.degree, self.nmoduli = limb_bits, degree, nmoduli
        self.dtype = np.dtype(_DT[limb_bits])
        self.params = params
        if nmoduli > params.max_moduli:
            raise ValueError("not enough moduli in the mirrored table")
        self._keep = [np.ascontiguousarray(x[:nmoduli]) for x in
                      (params.P, params.Pn, params.primitive_roots, params.invkmax)]
        self.ctx = L.nfl_oracle_create(limb_bits, degree, nmoduli, *[_vp(x) for x in self._keep],
                                       params.kmax_log2)
        if not self.ctx:
            raise ValueError("nfl_oracle_create rejected the shape")
        self.P = [int(v) for v in self._keep[0]]

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.nfl_oracle_destroy(self.ctx)
            self.ctx = None

    # -- helpers
    def _chk(self, a):
        assert a.dtype == self.dtype and a.flags.c_contiguous
        assert a.size % (self.degree * self.nmoduli) == 0
        return a.size // (self.degree * self.nmoduli)

    def table(self, which, cm):
        n = {TAB_OMEGAS: 2 * self.degree, TAB_INVOMEGAS: 2 * self.degree, TAB_INVPOLYDEGREE: 1}.get(which, self.degree)
        ptr = self.lib.nfl_oracle_table(self.ctx, which, cm)
        buf = (C.c_char * (n * self.dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=self.dtype).copy()

    def ntt(self, a):
        out = a.copy()
        self.lib.nfl_oracle_ntt_pow_phi(self.ctx, _vp(out), self._chk(out))
        return out

    def intt(self, a):
        out = a.copy()
        self.lib.nfl_oracle_invntt_pow_invphi(self.ctx, _vp(out), self._chk(out))
        return out

    def ntt_row(self, row, cm, inverse_tables=False):
        out = np.ascontiguousarray(row).copy()
        self.lib.nfl_oracle_ntt_row(self.ctx, _vp(out), cm, int(inverse_tables))
        return out

    def pointwise(self, op, a, b=None, bprime=None):
        out = np.empty_like(a)
        for other in (b, bprime):
            assert other is None or (other.flags.c_contiguous and other.shape == a.shape and other.dtype == a.dtype)
        self.lib.nfl_oracle_pointwise(self.ctx, op, _vp(out), _vp(a), _vp(b), _vp(bprime), self._chk(a))
        return out

    def polymul(self, a, b):
        out = np.empty_like(a)
        self.lib.nfl_oracle_polymul(self.ctx, _vp(out), _vp(a), _vp(b), self._chk(a))
        return out

    def polymul_mt(self, a, b, nthreads):
        out = np.empty_like(a)
        self.lib.nfl_oracle_polymul_mt(self.ctx, _vp(out), _vp(a), _vp(b), self._chk(a), nthreads)
        return out

    def any_eq(self, a, b):
        return bool(self.lib.nfl_oracle_any_eq(self.ctx, _vp(a), _vp(b), self._chk(a)))

    def any_neq(self, a, b):
        return bool(self.lib.nfl_oracle_any_neq(self.ctx, _vp(a), _vp(b), self._chk(a)))

    @property
    def crt_limbs(self):
        return self.lib.nfl_oracle_crt_limbs(self.ctx)

    @property
    def crt_bits(self):
        return self.lib.nfl_oracle_crt_bits(self.ctx)

    @property
    def crt_shift(self):
        return self.lib.nfl_oracle_crt_shift(self.ctx)

    def _big(self, fn, *pre):
        buf = np.zeros(2048, dtype=np.uint64)
        n = fn(self.ctx, *pre, _vp(buf), buf.size)
        return int.from_bytes(buf[:n].tobytes(), "little")

    def crt_modulus(self):
        return self._big(self.lib.nfl_oracle_crt_modulus)

    def crt_modulus_shoup(self):
        return self._big(self.lib.nfl_oracle_crt_modulus_shoup)

    def crt_lifting(self, cm):
        return self._big(self.lib.nfl_oracle_crt_lifting, cm)

    def crt_lift(self, a):
        batch = self._chk(a)
        out = np.zeros((batch, self.degree, self.crt_limbs), dtype=np.uint64)
        self.lib.nfl_oracle_crt_lift(self.ctx, _vp(out), _vp(a), batch)
        return out

    def crt_project(self, limbs):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        batch, deg, L = limbs.shape
        assert deg == self.degree
        out = np.empty((batch, self.nmoduli, self.degree), dtype=self.dtype)
        self.lib.nfl_oracle_crt_project(self.ctx, _vp(out), _vp(limbs), L, batch)
        return out

    def fill_uniform(self, batch, seed, operand=0, first_poly=0):
        out = np.empty((batch, self.nmoduli, self.degree), dtype=self.dtype)
        self.lib.nfl_oracle_fill_uniform(self.ctx, _vp(out), first_poly, batch, seed, operand)
        return out


REF_PATH = os.path.join(_HERE, "_ref", "libnflref.so")


def ref_available():
    return os.path.exists(REF_PATH)


def ref_gauss_barriers(sigma, security, samples, center=0.0):
    """The cumulative table the REAL reference builds for FastGaussianNoise<uint8_t, uint64_t, 2>(sigma, security, samples,
    center) (oracle/ref_gauss_shim.cpp): (bit_precision, rounded_center, [barrier_0, ...] as Python ints of that many
    bits); None when this build of the reference library lacks the entry point."""
    L = _load(REF_PATH)
    if not hasattr(L, "nflref_gauss_barriers"):
        return None
    L.nflref_gauss_barriers.restype = C.c_long
    L.nflref_gauss_barriers.argtypes = [C.c_double, C.c_uint, C.c_uint, C.c_double, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    bp, wp, nb, rc = C.c_uint(), C.c_uint(), C.c_uint(), C.c_int()
    need = -L.nflref_gauss_barriers(sigma, security, samples, center, None, 0, C.byref(bp), C.byref(wp), C.byref(nb), C.byref(rc))
    buf = np.zeros(need, dtype=np.uint8)
    got = L.nflref_gauss_barriers(sigma, security, samples, center, buf.ctypes.data, buf.size, C.byref(bp), C.byref(wp),
                                  C.byref(nb), C.byref(rc))
    assert got == need
    rows = buf.reshape(nb.value, wp.value)
    return bp.value, rc.value, [int.from_bytes(bytes(r), "big") for r in rows]


def ref_gauss_replay(sigma, security, samples, center, rlen):
    """Fork-replay of the REAL FastGaussianNoise<uint8_t, uint64_t, 2>::getNoise: (samples int64[rlen], raw uint8
    [3, call_bytes] -- the buffers its fastrandombytes() calls would return, in order --, call_words); None when the
    prebuilt reference library lacks the entry point."""
    L = _load(REF_PATH)
    if not hasattr(L, "nflref_gauss_replay"):
        return None
    L.nflref_gauss_replay.restype = C.c_long
    L.nflref_gauss_replay.argtypes = [C.c_double, C.c_uint, C.c_uint, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    cb, cw = C.c_uint64(), C.c_uint64()
    out = np.zeros(rlen, dtype=np.int64)
    need = -L.nflref_gauss_replay(sigma, security, samples, center, rlen, out.ctypes.data, None, 0, C.byref(cb), C.byref(cw))
    raw = np.zeros(need, dtype=np.uint8)
    got = L.nflref_gauss_replay(sigma, security, samples, center, rlen, out.ctypes.data, raw.ctypes.data, raw.size,
                                C.byref(cb), C.byref(cw))
    if got != need:
        raise RuntimeError("nflref_gauss_replay failed")
    return out, raw.reshape(3, cb.value), cw.value


class Reference:
    """The REAL reference (one poly at a time), through oracle/ref_shim.cpp."""

    _lib = None

    def __init__(self, limb_bits, degree, nmoduli):
        if Reference._lib is None:
            L = Reference._lib = _load(REF_PATH)
            L.nflref_find.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
            L.nflref_table.restype = C.c_void_p
            L.nflref_table.argtypes = [C.c_int, C.c_int, C.c_size_t]
            for name in ("nflref_crt_bits", "nflref_crt_shift"):
                getattr(L, name).restype = C.c_size_t
                getattr(L, name).argtypes = [C.c_int]
            for name in ("nflref_crt_modulus", "nflref_crt_modulus_shoup"):
                getattr(L, name).restype = C.c_size_t
                getattr(L, name).argtypes = [C.c_int, C.c_void_p, C.c_size_t]
            L.nflref_crt_lifting.restype = C.c_size_t
            L.nflref_crt_lifting.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
            L.nflref_ntt_pow_phi.argtypes = [C.c_int, C.c_void_p]
            L.nflref_invntt_pow_invphi.argtypes = [C.c_int, C.c_void_p]
            L.nflref_pointwise.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4
            L.nflref_any_eq.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
            L.nflref_any_neq.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
            L.nflref_ntt_row.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int]
            L.nflref_crt_lift.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
            L.nflref_crt_project.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
            if hasattr(L, "nflref_sample_replay"):
                L.nflref_sample_replay.restype = C.c_long
                L.nflref_sample_replay.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                                   C.c_void_p, C.c_size_t]
        self.lib = Reference._lib
        self.id = self.lib.nflref_find(limb_bits, degree, nmoduli)
        if self.id < 0:
            raise KeyError("shape not instantiated in ref_shim.cpp")
        self.limb_bits, self.degree, self.nmoduli = limb_bits, degree, nmoduli
        self.dtype = np.dtype(_DT[limb_bits])

    def sample_replay(self, kind, p0=0, p1=1, sigma=0.0):
        """One random constructor of the real reference (core.hpp:146-391): kind 0 uniform | 1 non_uniform(p0 = ub,
        p1 = amplifier) | 2 ZO_dist(p0 = rho) | 3 hwt_dist(p0 = h) | 4 gaussian(sigma, p0 = security, p1 = amplifier).
        Returns (poly [nm, n], raw bytes the constructor consumed -- empty for kinds 3 and 4)."""
        out = np.zeros((self.nmoduli, self.degree), dtype=self.dtype)
        raw = np.zeros(self.nmoduli * self.degree * self.dtype.itemsize, dtype=np.uint8)
        got = self.lib.nflref_sample_replay(self.id, kind, p0, p1, float(sigma), out.ctypes.data, raw.ctypes.data, raw.size)
        if got < 0:
            raise RuntimeError("nflref_sample_replay failed")
        return out, raw[:got].copy()

    def _each(self, a, fn):
        out = np.ascontiguousarray(a).copy().reshape(-1, self.nmoduli, self.degree)
        for k in range(out.shape[0]):
            fn(out[k])
        return out.reshape(a.shape)

    def ntt(self, a):
        return self._each(a, lambda p: self.lib.nflref_ntt_pow_phi(self.id, _vp(p)))

    def intt(self, a):
        return self._each(a, lambda p: self.lib.nflref_invntt_pow_invphi(self.id, _vp(p)))

    def ntt_row(self, row, cm, inverse_tables=False):
        out = np.ascontiguousarray(row).copy()
        self.lib.nflref_ntt_row(self.id, _vp(out), cm, int(inverse_tables))
        return out

    def pointwise(self, op, a, b=None, bprime=None):
        A = np.ascontiguousarray(a).reshape(-1, self.nmoduli, self.degree)
        B = A if b is None else np.ascontiguousarray(b).reshape(A.shape)
        BP = A if bprime is None else np.ascontiguousarray(bprime).reshape(A.shape)
        out = np.empty_like(A)
        for k in range(A.shape[0]):
            self.lib.nflref_pointwise(self.id, op, _vp(out[k]), _vp(A[k]), _vp(B[k]), _vp(BP[k]))
        return out.reshape(a.shape)

    def polymul(self, a, b):
        return self.intt(self.pointwise(OP_MUL, self.ntt(a), self.ntt(b)))

    def any_eq(self, a, b):
        return bool(self.lib.nflref_any_eq(self.id, _vp(np.ascontiguousarray(a)), _vp(np.ascontiguousarray(b))))

    def any_neq(self, a, b):
        return bool(self.lib.nflref_any_neq(self.id, _vp(np.ascontiguousarray(a)), _vp(np.ascontiguousarray(b))))

    def table(self, which, cm):
        n = {TAB_OMEGAS: 2 * self.degree, TAB_INVOMEGAS: 2 * self.degree, TAB_INVPOLYDEGREE: 1}.get(which, self.degree)
        ptr = self.lib.nflref_table(self.id, which, cm)
        buf = (C.c_char * (n * self.dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=self.dtype).copy()

    @property
    def crt_bits(self):
        return self.lib.nflref_crt_bits(self.id)

    @property
    def crt_shift(self):
        return self.lib.nflref_crt_shift(self.id)

    def _big(self, fn, *pre):
        buf = np.zeros(2048, dtype=np.uint64)
        n = fn(self.id, *pre, _vp(buf), buf.size)
        return int.from_bytes(buf[:n].tobytes(), "little")

    def crt_modulus(self):
        return self._big(self.lib.nflref_crt_modulus)

    def crt_modulus_shoup(self):
        return self._big(self.lib.nflref_crt_modulus_shoup)

    def crt_lifting(self, cm):
        return self._big(self.lib.nflref_crt_lifting, cm)

    def crt_lift(self, a, L):
        A = np.ascontiguousarray(a).reshape(-1, self.nmoduli, self.degree)
        out = np.zeros((A.shape[0], self.degree, L), dtype=np.uint64)
        for k in range(A.shape[0]):
            self.lib.nflref_crt_lift(self.id, _vp(A[k]), _vp(out[k]), L)
        return out

    def crt_project(self, limbs):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        batch, deg, L = limbs.shape
        out = np.empty((batch, self.nmoduli, self.degree), dtype=self.dtype)
        for k in range(batch):
            self.lib.nflref_crt_project(self.id, _vp(out[k]), _vp(limbs[k]), L)
        return out
