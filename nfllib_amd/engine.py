"""Host-side driver of the HIP engine for tests, bench.py and the multi-GPU
batch split.  This is plumbing over the C ABI (include/nflhip.h): device
memory and streams come from PyTorch, all arithmetic happens in libnflhip.so.

Data layout is the reference's: a batch of nfl::poly<T,Degree,NbModuli> is a
dense [batch][NbModuli][Degree] tensor of T (poly.hpp:82-88, tests/tools.h:6-17).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FMT_I8, FMT_I16, FMT_I32, FMT_WORDS, NflHipError, OP_ADD, OP_COMPUTE_SHOUP, OP_MUL, OP_MUL_SHOUP, OP_SUB,  # noqa: F401
                   DIST_REFERENCE_WORDS, ROW_BITREV_IO, ROW_INVERSE_TABLES, TAB_INVDEGREE, TAB_INVOMEGAS,
                   TAB_INVPOLY_INVPHIS, TAB_MODULUS, TAB_OMEGAS, TAB_PHIS, TAB_PSI, TAB_SHOUPINVPOLY_INVPHIS,
                   TAB_SHOUPPHIS)
from .params import params as limb_params

_NP = {16: np.uint16, 32: np.uint32, 64: np.uint64}


def _torch():
    import torch
    return torch


def _vp(x):
    """void* of a numpy array, a torch tensor, an int or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


DIST_UNIFORM, DIST_BOUNDED, DIST_ZO, DIST_HWT = 0, 1, 2, 3   # NFLHIP_DIST_* (include/nflhip.h)


class Engine:
    """One (T, Degree, NbModuli) context on one GPU -- the device-side
    counterpart of poly<T,Degree,NbModuli>::base / ::gmp (poly.hpp:247, 275)."""

    def __init__(self, limb_bits, degree, nmoduli, device=0):
        self.lib = _lib.lib
        self.limb_bits, self.degree, self.nmoduli, self.device = limb_bits, degree, nmoduli, device
        self.np_dtype = np.dtype(_NP[limb_bits])
        pr = self.params = limb_params(limb_bits)
        if nmoduli > pr.max_moduli:
            raise ValueError("only %d moduli are mirrored for %d-bit limbs" % (pr.max_moduli, limb_bits))
        self._keep = [np.ascontiguousarray(x[:nmoduli]) for x in (pr.P, pr.primitive_roots, pr.invkmax)]
        h = C.c_void_p()
        rc = self.lib.nflhip_ctx_create(C.byref(h), device, limb_bits, degree, nmoduli,
                                        *[_vp(x) for x in self._keep], pr.kmax_log2)
        if rc != 0:
            raise NflHipError(rc, self.lib.nflhip_last_error(None).decode())
        self.ctx = h
        self.P = [int(v) for v in self._keep[0]]
        self.crt_limbs = self.lib.nflhip_crt_limbs(self.ctx)
        self.words_per_poly = degree * nmoduli
        self._next_stream = 1 << 32   # ids handed out to sampler calls that do not name one (see _sid)
        self.bytes_per_poly = self.words_per_poly * self.np_dtype.itemsize

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nflhip_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise NflHipError(rc, self.lib.nflhip_last_error(self.ctx).decode())

    # ---- device tensors (torch is only the allocator / stream provider) ----
    @property
    def torch_dtype(self):
        t = _torch()
        return {16: t.int16, 32: t.int32, 64: t.int64}[self.limb_bits]

    def empty(self, batch):
        t = _torch()
        return t.empty((batch, self.nmoduli, self.degree), dtype=self.torch_dtype, device="cuda:%d" % self.device)

    def to_device(self, arr):
        t = _torch()
        arr = np.ascontiguousarray(arr, dtype=self.np_dtype)
        signed = arr.view({16: np.int16, 32: np.int32, 64: np.int64}[self.limb_bits])
        return t.from_numpy(signed.copy()).to("cuda:%d" % self.device)

    def to_host(self, ten):
        return ten.detach().cpu().contiguous().numpy().view(self.np_dtype)

    def _stream(self, stream=None):
        t = _torch()
        s = stream if stream is not None else t.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def _batch(self, ten):
        n = ten.numel()
        assert n % self.words_per_poly == 0 and ten.is_contiguous()
        return n // self.words_per_poly

    def ntt_(self, d, stream=None):
        self._chk(self.lib.nflhip_ntt_fwd_dev(self.ctx, _vp(d), self._batch(d), self._stream(stream)))
        return d

    def intt_(self, d, stream=None):
        self._chk(self.lib.nflhip_ntt_inv_dev(self.ctx, _vp(d), self._batch(d), self._stream(stream)))
        return d

    def ntt_row_(self, rows, cm, inverse_tables=False, bitrev_io=False, stream=None):
        """core::ntt (core.hpp:455-532) on contiguous rows of modulus cm, in place: cyclic, natural in, bit-reversed
        out; inverse_tables selects the invomegas tables, bitrev_io wraps it in the two permutations of core::inv_ntt"""
        assert rows.is_contiguous() and rows.numel() % self.degree == 0
        mode = (ROW_INVERSE_TABLES if inverse_tables else 0) | (ROW_BITREV_IO if bitrev_io else 0)
        self._chk(self.lib.nflhip_ntt_row_dev(self.ctx, _vp(rows), cm, mode, rows.numel() // self.degree,
                                              self._stream(stream)))
        return rows

    def h_ntt_row(self, rows, cm, inverse_tables=False, bitrev_io=False):
        out = np.ascontiguousarray(rows, dtype=self.np_dtype).copy()
        mode = (ROW_INVERSE_TABLES if inverse_tables else 0) | (ROW_BITREV_IO if bitrev_io else 0)
        self._chk(self.lib.nflhip_ntt_row(self.ctx, _vp(out), cm, mode, out.size // self.degree))
        return out

    def pointwise(self, op, a, b=None, bprime=None, out=None, stream=None):
        out = out if out is not None else _torch().empty_like(a)
        self._chk(self.lib.nflhip_pointwise_dev(self.ctx, op, _vp(out), _vp(a), _vp(b), _vp(bprime), self._batch(a),
                                                self._stream(stream)))
        return out

    def eval(self, program, operands, out=None, stream=None):
        """Fused expression tree: `program` = postfix bytes (k<8 push operand k, 0x10 add, 0x11 sub,
        0x12 mul, 0x13 mul_shoup, 0x14 compute_shoup), one device pass (nflhip_eval_dev)."""
        out = out if out is not None else _torch().empty_like(operands[0])
        ptrs = (C.c_void_p * len(operands))(*[o.data_ptr() for o in operands])
        prog = (C.c_ubyte * len(program))(*program)
        self._chk(self.lib.nflhip_eval_dev(self.ctx, _vp(out), C.cast(ptrs, C.c_void_p), len(operands),
                                           C.cast(prog, C.c_void_p), len(program), self._batch(operands[0]),
                                           self._stream(stream)))
        return out

    def eval_strided(self, program, operands, strides, out, out_stride=1, batch=None, stream=None):
        """nflhip_eval_strided_dev: element i reads operand j at operands[j] + i*strides[j] polynomials (0 = one shared
        polynomial) and writes out + i*out_stride polynomials"""
        if len(strides) != len(operands):
            raise ValueError("one stride per operand")
        if not out.is_contiguous() or not all(o.is_contiguous() for o in operands):
            raise ValueError("operands and result must be contiguous tensors")
        if batch is None:   # the elements the result buffer holds at this stride
            batch = (self._batch(out) + out_stride - 1) // out_stride
        ptrs = (C.c_void_p * len(operands))(*[o.data_ptr() for o in operands])
        sd = (C.c_size_t * len(operands))(*strides)
        prog = (C.c_ubyte * len(program))(*program)
        self._chk(self.lib.nflhip_eval_strided_dev(self.ctx, _vp(out), out_stride, C.cast(ptrs, C.c_void_p),
                                                   C.cast(sd, C.c_void_p), len(operands), C.cast(prog, C.c_void_p),
                                                   len(program), batch, self._stream(stream)))
        return out

    def h_eval(self, program, operands):
        out = np.empty_like(operands[0])
        ptrs = (C.c_void_p * len(operands))(*[o.ctypes.data for o in operands])
        prog = (C.c_ubyte * len(program))(*program)
        self._chk(self.lib.nflhip_eval(self.ctx, _vp(out), C.cast(ptrs, C.c_void_p), len(operands),
                                       C.cast(prog, C.c_void_p), len(program), self._hb(operands[0])))
        return out

    def polymul(self, a, b, out=None, b_is_ntt=False, stream=None):
        out = out if out is not None else _torch().empty_like(a)
        fn = self.lib.nflhip_polymul_ntt_dev if b_is_ntt else self.lib.nflhip_polymul_dev
        self._chk(fn(self.ctx, _vp(out), _vp(a), _vp(b), self._batch(a), self._stream(stream)))
        return out

    # ---- transform-fused pipelines (include/nflhip.h "transform-fused pipelines") ----
    def _operand(self, ten, words_only=False):
        """nflhip_operand of a tensor: [count][nmoduli][degree] limb words, or -- forward inputs -- [count][degree] int8 /
        int16 / int32 (one signed integer per coefficient); count 1 = one polynomial for the whole batch (stride 0)"""
        t = _torch()
        if not ten.is_contiguous():
            raise ValueError("operands must be contiguous tensors")
        # the shape names the format (torch has no unsigned 16 / 32-bit types, so a u16 / u32 ring's words share int16 / int32
        # with the compact formats): limb words are 3-D [count][nmoduli][degree], compact polynomials 2-D [count][degree];
        # anything else is ambiguous and refused
        compact = {t.int8: FMT_I8, t.int16: FMT_I16, t.int32: FMT_I32}.get(ten.dtype)
        if ten.dim() == 3 and ten.dtype == self.torch_dtype and tuple(ten.shape[1:]) == (self.nmoduli, self.degree):
            fmt = FMT_WORDS
        elif ten.dim() == 2 and compact is not None and ten.shape[1] == self.degree and not words_only:
            fmt = compact
        else:
            raise ValueError("operand shape / dtype: limb words are [count][nmoduli][degree] of the ring's word type, compact "
                             "polynomials [count][degree] int8 / int16 / int32")
        per = self.degree if fmt != FMT_WORDS else self.words_per_poly
        count = ten.numel() // per
        return _lib.Operand(ten.data_ptr(), 0 if count == 1 else 1, fmt), count

    def fwd_fma(self, x, k, e, out=None, batch=None, stream=None):
        """out = NTT(x) * k + NTT(e) in one pass (nflhip_fwd_fma_dev)"""
        (ox, nx), (ok, nk), (oe, ne) = self._operand(x), self._operand(k, True), self._operand(e)
        batch = batch if batch is not None else max(nx, nk, ne)
        out = out if out is not None else self.empty(batch)
        self._chk(self.lib.nflhip_fwd_fma_dev(self.ctx, _vp(out), C.byref(ox), C.byref(ok), C.byref(oe), batch, self._stream(stream)))
        return out

    def fwd_fma2(self, x, k0, e0, k1, e1, out0=None, out1=None, batch=None, stream=None):
        """out0 = NTT(x) * k0 + NTT(e0), out1 = NTT(x) * k1 + NTT(e1) in one pass: the LWE demo's encrypt()
        (tests/nfllib_demo_main_op.cpp:26-46) for the whole batch (nflhip_fwd_fma2_dev)"""
        ops = [self._operand(x), self._operand(k0, True), self._operand(e0), self._operand(k1, True), self._operand(e1)]
        batch = batch if batch is not None else max(n for _, n in ops)
        out0 = out0 if out0 is not None else self.empty(batch)
        out1 = out1 if out1 is not None else self.empty(batch)
        self._chk(self.lib.nflhip_fwd_fma2_dev(self.ctx, _vp(out0), _vp(out1), *[C.byref(o) for o, _ in ops], batch, self._stream(stream)))
        return out0, out1

    def fma_inv(self, a, k, b, subtract=False, out=None, batch=None, stream=None):
        """out = INTT(b + a * k) or INTT(b - a * k): the demo's decrypt() (tests/nfllib_demo_main_op.cpp:49-51)"""
        (oa, na), (ok, nk), (ob, nb) = self._operand(a, True), self._operand(k, True), self._operand(b, True)
        batch = batch if batch is not None else max(na, nk, nb)
        out = out if out is not None else self.empty(batch)
        self._chk(self.lib.nflhip_fma_inv_dev(self.ctx, _vp(out), C.byref(oa), C.byref(ok), C.byref(ob), int(bool(subtract)), batch,
                                              self._stream(stream)))
        return out

    def expand_small(self, src, batch=None, out=None, stream=None):
        """compact polynomials (one signed integer per coefficient) -> residue words (nflhip_expand_small_dev)"""
        o, n = self._operand(src)
        batch = batch if batch is not None else n
        out = out if out is not None else self.empty(batch)
        self._chk(self.lib.nflhip_expand_small_dev(self.ctx, _vp(out), C.byref(o), batch, self._stream(stream)))
        return out

    def empty_small(self, batch, fmt=FMT_I8):
        t = _torch()
        return t.empty((batch, self.degree), dtype={FMT_I8: t.int8, FMT_I16: t.int16, FMT_I32: t.int32}[fmt], device="cuda:%d" % self.device)

    def sample_gauss_small(self, d, g, key, stream_id=None, amplifier=1, first_poly=0, stream=None):
        """compact Gaussian polynomials: d[b][i] = sample * amplifier as int8 / int16 / int32 (the tensor's dtype), the same
        samples sample_gauss spreads over the moduli"""
        o, n = self._operand(d)
        self._chk(self.lib.nflhip_sample_gauss_small_dev(self.ctx, _vp(d), o.format, first_poly, n, g, amplifier, self._key(key),
                                                         self._sid(stream_id), self._stream(stream)))
        return d

    def sample_gauss_small_seq(self, d, g, key, first_stream_id, stream_id_stride=1, amplifier=1, stream=None):
        o, n = self._operand(d)
        self._chk(self.lib.nflhip_sample_gauss_small_seq_dev(self.ctx, _vp(d), o.format, n, g, amplifier, self._key(key),
                                                             first_stream_id, stream_id_stride, self._stream(stream)))
        return d

    def sample_gauss_small_multi(self, ds, g, key, stream_ids, stream_id_strides=None, amplifiers=None, stream=None):
        """up to four compact draws of one table in ONE launch (nflhip_sample_gauss_small_multi_dev): draw j fills ds[j] exactly as
        sample_gauss_small_seq(ds[j], g, key, stream_ids[j], stream_id_strides[j], amplifiers[j]) would -- or, without strides, as
        sample_gauss_small(ds[j], g, key, stream_ids[j], amplifiers[j])"""
        cnt = len(ds)
        o, n = self._operand(ds[0])
        amplifiers = list(amplifiers) if amplifiers is not None else [1] * cnt
        ptrs = (C.c_void_p * cnt)(*[d.data_ptr() for d in ds])
        amps = (C.c_uint64 * cnt)(*amplifiers)
        sids = (C.c_uint64 * cnt)(*stream_ids)
        strides = (C.c_uint64 * cnt)(*stream_id_strides) if stream_id_strides is not None else None
        self._chk(self.lib.nflhip_sample_gauss_small_multi_dev(self.ctx, ptrs, cnt, o.format, n, g, amps, self._key(key), sids, strides,
                                                               self._stream(stream)))
        return ds

    def check_range(self, d, stream=None):
        """CHECK_STRICTMOD's assertion over a resident batch: True iff some word is >= its row's modulus"""
        r = C.c_int(0)
        self._chk(self.lib.nflhip_check_range_dev(self.ctx, _vp(d), self._batch(d), C.byref(r), self._stream(stream)))
        return bool(r.value)

    def h_check_range(self, a):
        r = C.c_int(0)
        self._chk(self.lib.nflhip_check_range(self.ctx, _vp(a), self._hb(a), C.byref(r)))
        return bool(r.value)

    def any_eq(self, a, b, stream=None):
        r = C.c_int(0)
        self._chk(self.lib.nflhip_any_eq_dev(self.ctx, _vp(a), _vp(b), self._batch(a), C.byref(r), self._stream(stream)))
        return bool(r.value)

    def any_neq(self, a, b, stream=None):
        r = C.c_int(0)
        self._chk(self.lib.nflhip_any_neq_dev(self.ctx, _vp(a), _vp(b), self._batch(a), C.byref(r), self._stream(stream)))
        return bool(r.value)

    def broadcast(self, one, count, stream=None):
        """`count` copies of one polynomial (nflhip_broadcast_dev)"""
        out = self.empty(count)
        self._chk(self.lib.nflhip_broadcast_dev(self.ctx, _vp(out), _vp(one), count, self._stream(stream)))
        return out

    def fill_uniform(self, d, seed, operand=0, first_poly=0, stream=None):
        self._chk(self.lib.nflhip_fill_uniform_dev(self.ctx, _vp(d), first_poly, self._batch(d), seed, operand,
                                                   self._stream(stream)))
        return d

    # ---- samplers (include/nflhip.h "samplers"): key = 32 bytes, stream_id selects the keystream ----
    @staticmethod
    def _key(key):
        key = bytes(key)
        assert len(key) == 32, "the sampler key is 32 bytes"
        return C.create_string_buffer(key, 32)

    def _sid(self, stream_id):
        """A (key, stream_id, distribution) triple must never be used twice for values that must be independent (a
        public polynomial and the noise next to it): without an explicit id every call takes a fresh one."""
        if stream_id is None:
            stream_id = self._next_stream
            self._next_stream += 1
        return stream_id

    NARROW = True   # this engine knows the narrow draws (DIST_NARROW, gauss_create(draw_bits=32))

    def sample(self, d, dist, key, stream_id=None, param0=0, param1=1, first_poly=0, stream=None, narrow=False):
        """dist: DIST_UNIFORM | DIST_BOUNDED (param0 = upper bound, param1 = amplifier) | DIST_ZO (param0 = rho)
        | DIST_HWT (param0 = hamming weight); narrow (uniform only): keystream lanes of the limb width (NFLHIP_DIST_NARROW)"""
        dist |= _lib.DIST_NARROW if narrow else 0
        self._chk(self.lib.nflhip_sample_dev(self.ctx, _vp(d), first_poly, self._batch(d), dist, param0, param1,
                                             self._key(key), self._sid(stream_id), self._stream(stream)))
        return d

    def sample_seq(self, d, dist, key, first_stream_id, stream_id_stride=1, param0=0, param1=1, stream=None, narrow=False):
        """polynomial b = sample(one polynomial, stream id first_stream_id + b*stride) (nflhip_sample_seq_dev)"""
        dist |= _lib.DIST_NARROW if narrow else 0
        self._chk(self.lib.nflhip_sample_seq_dev(self.ctx, _vp(d), self._batch(d), dist, param0, param1, self._key(key),
                                                 first_stream_id, stream_id_stride, self._stream(stream)))
        return d

    def sample_gauss_seq(self, d, g, key, first_stream_id, stream_id_stride=1, amplifier=1, stream=None):
        self._chk(self.lib.nflhip_sample_gauss_seq_dev(self.ctx, _vp(d), self._batch(d), g, amplifier, self._key(key),
                                                       first_stream_id, stream_id_stride, self._stream(stream)))
        return d

    def random_words(self, nwords, key, stream_id=0, first_word=0, stream=None):
        t = _torch()
        out = t.empty((nwords,), dtype=t.int64, device="cuda:%d" % self.device)
        self._chk(self.lib.nflhip_random_words_dev(self.ctx, _vp(out), first_word, nwords, self._key(key), stream_id,
                                                   self._stream(stream)))
        return out

    def gauss_create(self, sigma, security=128, samples=None, center=0.0, draw_bits=64):
        """FastGaussianNoise(sigma, security, samples, center): returns a handle for sample_gauss / gauss_info;
        draw_bits = 32: the narrow draw (nflhip_gauss_set_draw_bits)"""
        h = C.c_void_p()
        self._chk(self.lib.nflhip_gauss_create(self.ctx, C.byref(h), float(sigma), int(security),
                                               int(samples if samples is not None else self.degree), float(center)))
        if draw_bits != 64:
            if self.lib.nflhip_gauss_set_draw_bits(h, int(draw_bits)) != 0:
                self.lib.nflhip_gauss_destroy(self.ctx, h)
                raise ValueError("draw_bits must be 64 or 32")
        return h

    def gauss_destroy(self, g):
        self._chk(self.lib.nflhip_gauss_destroy(self.ctx, g))

    def gauss_info(self, g):
        import numpy as np
        x_min, entries, words, bits, tail = C.c_longlong(), C.c_size_t(), C.c_int(), C.c_uint(), C.c_double()
        self._chk(self.lib.nflhip_gauss_info(g, C.byref(x_min), C.byref(entries), C.byref(words), C.byref(bits),
                                             C.byref(tail), None))
        tab = np.zeros((entries.value, words.value), dtype=np.uint64)
        self._chk(self.lib.nflhip_gauss_info(g, None, None, None, None, None, tab.ctypes.data_as(C.c_void_p)))
        return {"x_min": x_min.value, "entries": entries.value, "words": words.value, "bit_precision": bits.value,
                "tail": tail.value, "table": tab}

    def sample_gauss(self, d, g, key, stream_id=None, amplifier=1, first_poly=0, stream=None):
        self._chk(self.lib.nflhip_sample_gauss_dev(self.ctx, _vp(d), first_poly, self._batch(d), g, amplifier,
                                                   self._key(key), self._sid(stream_id), self._stream(stream)))
        return d

    def gauss_noise(self, g, count, key, stream_id=None, first_sample=0, stream=None):
        """raw signed samples (FastGaussianNoise::getNoise) as an int64 device tensor"""
        t = _torch()
        out = t.empty((count,), dtype=t.int64, device="cuda:%d" % self.device)
        self._chk(self.lib.nflhip_gauss_noise_dev(self.ctx, _vp(out), first_sample, count, g, self._key(key),
                                                  self._sid(stream_id), self._stream(stream)))
        return out

    def h_gauss_noise(self, g, count, key, stream_id=None):
        out = np.empty((count,), dtype=np.int64)
        self._chk(self.lib.nflhip_gauss_noise(self.ctx, out.ctypes.data_as(C.c_void_p), count, g, self._key(key),
                                              self._sid(stream_id)))
        return out

    def crt_lift(self, d, stream=None):
        t = _torch()
        batch = self._batch(d)
        out = t.empty((batch, self.degree, self.crt_limbs), dtype=t.int64, device=d.device)
        self._chk(self.lib.nflhip_crt_lift_dev(self.ctx, _vp(out), _vp(d), batch, self._stream(stream)))
        return out

    def crt_project(self, limbs, stream=None):
        batch, deg, L = limbs.shape
        assert deg == self.degree and limbs.is_contiguous()
        out = self.empty(batch)
        self._chk(self.lib.nflhip_crt_project_dev(self.ctx, _vp(out), _vp(limbs), L, batch, self._stream(stream)))
        return out

    # ---- the batch split (include/nflhip.h "multi-GPU") ----
    def digest(self, d, first_poly=0, stream=None):
        """shard-composable 64-bit digest of a resident batch (sharding.digest_words is its numpy statement)"""
        out = C.c_uint64(0)
        self._chk(self.lib.nflhip_digest_dev(self.ctx, _vp(d), first_poly, self._batch(d), C.byref(out), self._stream(stream)))
        return int(out.value)

    def time_polymul(self, c, a, b, iters, stream=None):
        ms = C.c_float(0)
        self._chk(self.lib.nflhip_time_polymul_dev(self.ctx, _vp(c), _vp(a), _vp(b), self._batch(a), iters,
                                                   self._stream(stream), C.byref(ms)))
        return ms.value

    # ---- host-pointer path (numpy in, numpy out): what the per-poly C++ surface calls ----
    def _hb(self, a):
        assert isinstance(a, np.ndarray) and a.dtype == self.np_dtype and a.flags.c_contiguous
        assert a.size % self.words_per_poly == 0
        return a.size // self.words_per_poly

    def h_ntt(self, a):
        out = a.copy()
        self._chk(self.lib.nflhip_ntt_fwd(self.ctx, _vp(out), self._hb(out)))
        return out

    def h_intt(self, a):
        out = a.copy()
        self._chk(self.lib.nflhip_ntt_inv(self.ctx, _vp(out), self._hb(out)))
        return out

    def h_pointwise(self, op, a, b=None, bprime=None):
        out = np.empty_like(a)
        self._chk(self.lib.nflhip_pointwise(self.ctx, op, _vp(out), _vp(a), _vp(b), _vp(bprime), self._hb(a)))
        return out

    def h_polymul(self, a, b, out=None):
        out = np.empty_like(a) if out is None else out
        self._chk(self.lib.nflhip_polymul(self.ctx, _vp(out), _vp(a), _vp(b), self._hb(a)))
        return out

    def h_any_eq(self, a, b):
        r = C.c_int(0)
        self._chk(self.lib.nflhip_any_eq(self.ctx, _vp(a), _vp(b), self._hb(a), C.byref(r)))
        return bool(r.value)

    def h_any_neq(self, a, b):
        r = C.c_int(0)
        self._chk(self.lib.nflhip_any_neq(self.ctx, _vp(a), _vp(b), self._hb(a), C.byref(r)))
        return bool(r.value)

    def h_crt_lift(self, a):
        batch = self._hb(a)
        out = np.zeros((batch, self.degree, self.crt_limbs), dtype=np.uint64)
        self._chk(self.lib.nflhip_crt_lift(self.ctx, _vp(out), _vp(a), batch))
        return out

    def h_crt_project(self, limbs):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        batch, deg, L = limbs.shape
        out = np.empty((batch, self.nmoduli, self.degree), dtype=self.np_dtype)
        self._chk(self.lib.nflhip_crt_project(self.ctx, _vp(out), _vp(limbs), L, batch))
        return out

    def table(self, which, cm):
        n = {TAB_PSI: 2 * self.degree, TAB_OMEGAS: 2 * self.degree, TAB_INVOMEGAS: 2 * self.degree, TAB_MODULUS: 1,
             TAB_INVDEGREE: 1}.get(which, self.degree)
        out = np.zeros(n, dtype=self.np_dtype)
        self._chk(self.lib.nflhip_get_table(self.ctx, which, cm, _vp(out), out.nbytes))
        return out

    def crt_constant(self, what, cm=0):
        buf = np.zeros(self.crt_limbs + 2, dtype=np.uint64)
        n = C.c_size_t(0)
        self._chk(self.lib.nflhip_get_crt_constant(self.ctx, what, cm, _vp(buf), buf.size, C.byref(n)))
        return int.from_bytes(buf[:n.value].tobytes(), "little")


def shard_range(total, nranks, rank):
    """(first, count) of rank's contiguous shard: nflhip_shard_range (no device needed)"""
    f, c = C.c_size_t(0), C.c_size_t(0)
    rc = _lib.lib.nflhip_shard_range(total, nranks, rank, C.byref(f), C.byref(c))
    if rc != 0:
        raise NflHipError(rc, _lib.lib.nflhip_last_error(None).decode())
    return int(f.value), int(c.value)


class Comm:
    """One process per GPU: the RCCL communicator of include/nflhip.h (nflhip_comm_*).  Rank 0 draws `Comm.unique_id()`
    and hands the 128 bytes to the other ranks out of band (bench.py: torch.distributed's broadcast)."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        rc = _lib.lib.nflhip_comm_unique_id(buf)
        if rc != 0:
            raise NflHipError(rc, _lib.lib.nflhip_last_error(None).decode())
        return bytes(buf)

    def __init__(self, engine, nranks, rank, unique_id):
        self.eng, self.lib = engine, _lib.lib
        h = C.c_void_p()
        idb = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        rc = self.lib.nflhip_comm_create(C.byref(h), engine.ctx, nranks, rank, idb)
        if rc != 0:
            raise NflHipError(rc, self.lib.nflhip_last_error(None).decode())
        self.h, self.nranks, self.rank = h, nranks, rank

    def close(self):
        if getattr(self, "h", None):
            self.lib.nflhip_comm_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise NflHipError(rc, self.lib.nflhip_last_error(None).decode())

    def scatter(self, shard, full, total, root=0, stream=None):
        self._chk(self.lib.nflhip_scatter_dev(self.h, _vp(shard), _vp(full), total, root, self.eng._stream(stream)))
        return shard

    def gather(self, full, shard, total, root=0, stream=None):
        self._chk(self.lib.nflhip_gather_dev(self.h, _vp(full), _vp(shard), total, root, self.eng._stream(stream)))
        return full

    def barrier(self, stream=None):
        self._chk(self.lib.nflhip_comm_barrier(self.h, self.eng._stream(stream)))

    def allgather_u64(self, value, stream=None):
        out = (C.c_uint64 * self.nranks)()
        self._chk(self.lib.nflhip_comm_allgather_u64(self.h, value & ((1 << 64) - 1), out, self.eng._stream(stream)))
        return [int(v) for v in out]


def gauss_table(sigma, security=128, samples=1024, center=0.0):
    """The cumulative table of nflhip_gauss_create, built on the host (no device needed): dict like Engine.gauss_info."""
    from ._lib import load
    lib = load()
    x_min, entries, words, bits, tail = C.c_longlong(), C.c_size_t(), C.c_int(), C.c_uint(), C.c_double()
    rc = lib.nflhip_gauss_table(sigma, security, samples, center, C.byref(x_min), C.byref(entries), C.byref(words), C.byref(bits),
                                C.byref(tail), None, 0)
    if rc:
        raise NflHipError(rc, lib.nflhip_last_error(None).decode())
    tab = np.zeros((entries.value, words.value), dtype=np.uint64)
    rc = lib.nflhip_gauss_table(sigma, security, samples, center, None, None, None, None, None, tab.ctypes.data_as(C.c_void_p), tab.size)
    if rc:
        raise NflHipError(rc, lib.nflhip_last_error(None).decode())
    return {"x_min": x_min.value, "entries": entries.value, "words": words.value, "bit_precision": bits.value,
            "tail": tail.value, "table": tab}
