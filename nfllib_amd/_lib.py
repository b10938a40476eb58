"""ctypes binding of libnflhip.so (the C ABI declared in include/nflhip.h).

The library is hand-written HIP for gfx950 and is the ONLY compute path of this
package: if it cannot be loaded, importing fails loudly -- there is no CPU or
PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnflhip.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = range(6)
OP_ADD, OP_SUB, OP_MUL, OP_MUL_SHOUP, OP_COMPUTE_SHOUP = range(5)
TAB_PSI, TAB_MODULUS, TAB_INVDEGREE = range(3)
TAB_PHIS, TAB_SHOUPPHIS, TAB_INVPOLY_INVPHIS, TAB_SHOUPINVPOLY_INVPHIS, TAB_OMEGAS, TAB_INVOMEGAS = range(3, 9)
ROW_INVERSE_TABLES, ROW_BITREV_IO = 1, 2
DIST_REFERENCE_WORDS = 0x100
DIST_NARROW = 0x200
ABI_VERSION = 6
FMT_WORDS, FMT_I8, FMT_I16, FMT_I32 = range(4)

# every symbol include/nflhip.h declares: (name, restype, argtypes)
_vp, _sz, _i, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
SYMBOLS = [
    ("nflhip_abi_version", _i, []),
    ("nflhip_last_error", C.c_char_p, [_vp]),
    ("nflhip_device_count", _i, [C.POINTER(_i)]),
    ("nflhip_ctx_create", _i, [C.POINTER(_vp), _i, _i, _sz, _sz, _vp, _vp, _vp, _i]),
    ("nflhip_ctx_destroy", _i, [_vp]),
    ("nflhip_degree", _sz, [_vp]),
    ("nflhip_nmoduli", _sz, [_vp]),
    ("nflhip_limb_bits", _i, [_vp]),
    ("nflhip_crt_limbs", _sz, [_vp]),
    ("nflhip_get_table", _i, [_vp, _i, _sz, _vp, _sz]),
    ("nflhip_get_crt_constant", _i, [_vp, _i, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("nflhip_ntt_fwd_dev", _i, [_vp, _vp, _sz, _vp]),
    ("nflhip_ntt_inv_dev", _i, [_vp, _vp, _sz, _vp]),
    ("nflhip_ntt_fwd", _i, [_vp, _vp, _sz]),
    ("nflhip_ntt_inv", _i, [_vp, _vp, _sz]),
    ("nflhip_ntt_row_dev", _i, [_vp, _vp, _sz, _i, _sz, _vp]),
    ("nflhip_ntt_row", _i, [_vp, _vp, _sz, _i, _sz]),
    ("nflhip_pointwise_dev", _i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_pointwise", _i, [_vp, _i, _vp, _vp, _vp, _vp, _sz]),
    ("nflhip_eval_dev", _i, [_vp, _vp, _vp, _sz, _vp, _sz, _sz, _vp]),
    ("nflhip_eval", _i, [_vp, _vp, _vp, _sz, _vp, _sz, _sz]),
    ("nflhip_eval_strided_dev", _i, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _sz, _vp]),
    ("nflhip_polymul_dev", _i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_polymul", _i, [_vp, _vp, _vp, _vp, _sz]),
    ("nflhip_polymul_ntt_dev", _i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_fwd_fma_dev", _i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_fwd_fma2_dev", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_fma_inv_dev", _i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    ("nflhip_expand_small_dev", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_has_fused_kernels", _i, [_vp]),
    ("nflhip_any_eq_dev", _i, [_vp, _vp, _vp, _sz, C.POINTER(_i), _vp]),
    ("nflhip_any_neq_dev", _i, [_vp, _vp, _vp, _sz, C.POINTER(_i), _vp]),
    ("nflhip_any_eq", _i, [_vp, _vp, _vp, _sz, C.POINTER(_i)]),
    ("nflhip_any_neq", _i, [_vp, _vp, _vp, _sz, C.POINTER(_i)]),
    ("nflhip_check_range_dev", _i, [_vp, _vp, _sz, C.POINTER(_i), _vp]),
    ("nflhip_check_range", _i, [_vp, _vp, _sz, C.POINTER(_i)]),
    ("nflhip_crt_lift_dev", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_crt_project_dev", _i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    ("nflhip_crt_lift", _i, [_vp, _vp, _vp, _sz]),
    ("nflhip_crt_project", _i, [_vp, _vp, _vp, _sz, _sz]),
    ("nflhip_fill_uniform_dev", _i, [_vp, _vp, _sz, _sz, _u64, _i, _vp]),
    ("nflhip_sample_dev", _i, [_vp, _vp, _sz, _sz, _i, _u64, _u64, _vp, _u64, _vp]),
    ("nflhip_sample", _i, [_vp, _vp, _sz, _i, _u64, _u64, _vp, _u64]),
    ("nflhip_sample_seq_dev", _i, [_vp, _vp, _sz, _i, _u64, _u64, _vp, _u64, _u64, _vp]),
    ("nflhip_sample_gauss_seq_dev", _i, [_vp, _vp, _sz, _vp, _u64, _vp, _u64, _u64, _vp]),
    ("nflhip_random_words_dev", _i, [_vp, _vp, _u64, _sz, _vp, _u64, _vp]),
    ("nflhip_gauss_create", _i, [_vp, C.POINTER(_vp), C.c_double, C.c_uint, C.c_uint, C.c_double]),
    ("nflhip_gauss_destroy", _i, [_vp, _vp]),
    ("nflhip_gauss_set_draw_bits", _i, [_vp, _i]),
    ("nflhip_gauss_draw_bits", _i, [_vp]),
    ("nflhip_gauss_table", _i, [C.c_double, C.c_uint, C.c_uint, C.c_double, C.POINTER(C.c_longlong), C.POINTER(_sz), C.POINTER(_i),
                               C.POINTER(C.c_uint), C.POINTER(C.c_double), _vp, _sz]),
    ("nflhip_gauss_info", _i, [_vp, C.POINTER(C.c_longlong), C.POINTER(_sz), C.POINTER(_i), C.POINTER(C.c_uint),
                               C.POINTER(C.c_double), _vp]),
    ("nflhip_sample_gauss_dev", _i, [_vp, _vp, _sz, _sz, _vp, _u64, _vp, _u64, _vp]),
    ("nflhip_sample_gauss", _i, [_vp, _vp, _sz, _vp, _u64, _vp, _u64]),
    ("nflhip_sample_gauss_small_dev", _i, [_vp, _vp, _i, _sz, _sz, _vp, _u64, _vp, _u64, _vp]),
    ("nflhip_sample_gauss_small_seq_dev", _i, [_vp, _vp, _i, _sz, _vp, _u64, _vp, _u64, _u64, _vp]),
    ("nflhip_sample_gauss_small_multi_dev", _i, [_vp, _vp, _sz, _i, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("nflhip_gauss_noise_dev", _i, [_vp, _vp, _u64, _sz, _vp, _vp, _u64, _vp]),
    ("nflhip_gauss_noise", _i, [_vp, _vp, _sz, _vp, _vp, _u64]),
    ("nflhip_malloc", _i, [_vp, C.POINTER(_vp), _sz]),
    ("nflhip_free", _i, [_vp, _vp]),
    ("nflhip_memcpy_h2d", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_memcpy_d2h", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_memcpy_d2d", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_memset_dev", _i, [_vp, _vp, _i, _sz, _vp]),
    ("nflhip_stream_sync", _i, [_vp, _vp]),
    ("nflhip_stream_idle", _i, [_vp, _vp, C.POINTER(_i)]),
    ("nflhip_stream_create", _i, [_vp, C.POINTER(_vp)]),
    ("nflhip_stream_destroy", _i, [_vp, _vp]),
    ("nflhip_broadcast_dev", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("nflhip_random_bytes", _i, [_i, _vp, _sz, _vp, _u64]),
    ("nflhip_time_polymul_dev", _i, [_vp, _vp, _vp, _vp, _sz, _i, _vp, C.POINTER(C.c_float)]),
    # multi-GPU: the batch split
    ("nflhip_ctx_device", _i, [_vp]),
    ("nflhip_shard_range", _i, [_sz, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    ("nflhip_digest_dev", _i, [_vp, _vp, _sz, _sz, C.POINTER(_u64), _vp]),
    ("nflhip_memcpy_peer_dev", _i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    ("nflhip_scatter_local_dev", _i, [_vp, _i, _vp, _i, _vp, _sz, _vp]),
    ("nflhip_gather_local_dev", _i, [_vp, _i, _vp, _i, _vp, _sz, _vp]),
    ("nflhip_comm_unique_id", _i, [_vp]),
    ("nflhip_comm_create", _i, [C.POINTER(_vp), _vp, _i, _i, _vp]),
    ("nflhip_comm_destroy", _i, [_vp]),
    ("nflhip_comm_rank", _i, [_vp]),
    ("nflhip_comm_size", _i, [_vp]),
    ("nflhip_scatter_dev", _i, [_vp, _vp, _vp, _sz, _i, _vp]),
    ("nflhip_gather_dev", _i, [_vp, _vp, _vp, _sz, _i, _vp]),
    ("nflhip_comm_barrier", _i, [_vp, _vp]),
    ("nflhip_comm_allgather_u64", _i, [_vp, _u64, C.POINTER(_u64), _vp]),
]
COMM_ID_BYTES = 128


class Operand(C.Structure):
    """nflhip_operand: device pointer, stride in polynomials (0 = one polynomial for the whole batch), NFLHIP_FMT_*"""
    _fields_ = [("ptr", C.c_void_p), ("stride", C.c_size_t), ("format", C.c_int)]


class NflHipError(RuntimeError):
    """Non-zero status from the C ABI (the header-only C++ surface throws
    std::runtime_error in the same situations: core.hpp:111-115, gmp.hpp:83-87)."""

    def __init__(self, code, msg):
        super().__init__("nflhip status %d: %s" % (code, msg))
        self.code = code


def load():
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 pair; the dynamic
    # loader shares ONE HIP runtime per process by soname, so when torch is going to be
    # used (it is our device allocator / stream provider) it must get there first --
    # otherwise torch would run on the system runtime with its bundled HSA and see no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nfllib_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if lib.nflhip_abi_version() != ABI_VERSION:
        raise ImportError("nfllib_amd: libnflhip.so ABI version mismatch")
    return lib


lib = load()
