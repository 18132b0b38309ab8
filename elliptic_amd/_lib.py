"""ctypes binding of the C ABI declared in include/ellgpu.h.

`load()` opens the in-tree libellgpu.so (built by __graft_entry__.build() /
elliptic_amd/build.py with hipcc for gfx950).  There is no fallback of any
kind: if the library is missing this raises, and if it loads on a machine
without a usable MI355X, `Context()` raises from ellgpu_ctx_create.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "lib", "libellgpu.so")

c_u8p = ctypes.c_void_p      # raw addresses (numpy .ctypes.data / torch .data_ptr())

_SIGS = {
    "ellgpu_version": (ctypes.c_int, []),
    "ellgpu_source_digest": (ctypes.c_char_p, []),
    "ellgpu_last_error": (ctypes.c_char_p, []),
    "ellgpu_curve_id": (ctypes.c_int, [ctypes.c_char_p]),
    "ellgpu_curve_field_bytes": (ctypes.c_int, [ctypes.c_int]),
    "ellgpu_curve_order_bytes": (ctypes.c_int, [ctypes.c_int]),
    "ellgpu_device_count": (ctypes.c_int, []),
    "ellgpu_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "ellgpu_group_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "ellgpu_group_size": (ctypes.c_int, [ctypes.c_void_p]),
    "ellgpu_ctx_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "ellgpu_ctx_defer": (ctypes.c_int, [ctypes.c_void_p]),
    "ellgpu_ctx_collect": (ctypes.c_int, [ctypes.c_void_p]),
    "ellgpu_curve_define_short": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                                 ctypes.POINTER(ctypes.c_int)]),
    "ellgpu_curve_define_edwards": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                                   ctypes.POINTER(ctypes.c_int)]),
    "ellgpu_ctx_destroy": (None, [ctypes.c_void_p]),
    "ellgpu_ctx_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "ellgpu_ctx_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]),
    "ellgpu_ctx_comb_bits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "ellgpu_mul_fixed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p]),
    "ellgpu_mul_var": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_mul_add2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_ecdsa_verify": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_x25519_ladder": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_decompress": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_decode_points": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_size_t, c_u8p, c_u8p]),
    "ellgpu_decode_points_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_encode_points": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, c_u8p]),
    "ellgpu_encode_points_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, c_u8p, ctypes.c_void_p]),
    "ellgpu_validate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_int, c_u8p]),
    "ellgpu_validate_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_int, c_u8p, ctypes.c_void_p]),
    "ellgpu_sig_from_der": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_sig_from_der_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_sig_to_der": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, ctypes.c_size_t, c_u8p]),
    "ellgpu_sig_to_der_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, ctypes.c_void_p]),
    "ellgpu_ecdsa_verify_wire": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p]),
    "ellgpu_ecdsa_verify_wire_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_point_add": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_point_add_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_decompress_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_ecdsa_sign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_ecdsa_sign_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_eddsa_verify": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_eddsa_verify_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_ecdsa_sign_det": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_ecdsa_sign_det_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_ecdsa_recover": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_ecdsa_recover_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_eddsa_sign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p]),
    "ellgpu_eddsa_sign_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, ctypes.c_size_t, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_mul_fixed_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_mul_var_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_mul_add2_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_ecdsa_verify_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, c_u8p, ctypes.c_int, ctypes.c_int, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_x25519_derive": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p]),
    "ellgpu_x25519_ladder_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_u8p, c_u8p, c_u8p, c_u8p, ctypes.c_void_p]),
    "ellgpu_ctx_set_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "ellgpu_ctx_get_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    "ellgpu_debug_field_op": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, c_u8p, c_u8p, c_u8p]),
    "ellgpu_probe_valu": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

# every symbol include/ellgpu.h declares (tests check the .so exports them all)
SYMBOLS = sorted(_SIGS)

# ELLGPU_VERSION of the include/ellgpu.h this table was written against: load() refuses any other
# library (a prototype changed -- e.g. ellgpu_ecdsa_verify's out_status -- and a stale table would
# pass garbage for the added arguments)
ABI_VERSION = 0x000200


class EllgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ellgpu error %d: %s" % (code, msg))
        self.code = code


def load(path=None, optional=()):
    path = path or os.environ.get("ELLGPU_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise ImportError(
            "libellgpu.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    lib.ellgpu_version.restype = ctypes.c_int
    got = lib.ellgpu_version()
    if got != ABI_VERSION:
        raise ImportError("%s reports ABI version 0x%06x, this binding was written for 0x%06x (include/ellgpu.h "
                          "ELLGPU_VERSION): rebuild the library from this tree" % (path, got, ABI_VERSION))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in optional:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    lib._path = path
    return lib
