"""ctypes binding of ``libsdr_amd.so`` (declared in ``include/sdr_amd.h``).

There is no CPU fallback: if the shared library is missing the import fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDR_AMD_LIB selects another build of the engine (kernel tuning experiments)
LIB_PATH = os.environ.get("SDR_AMD_LIB") or os.path.join(_HERE, "libsdr_amd.so")

OK = 0
OP_AND, OP_OR, OP_XOR, OP_RM = 0, 1, 2, 3
DATA_EMPTY, DATA_UNIT, DATA_ARITH = 0, 1, 2
FLAG_SWAP, FLAG_FORCE_MERGE, FLAG_FORCE_BITSLICE, FLAG_FORCE_POPCOUNT, FLAG_FORCE_MFMA, FLAG_ROWWISE, FLAG_FORCE_ROWTABLE = 1, 2, 4, 8, 16, 32, 64
FLAG_FORCE_RANK = 128
FLAG_FORCE_BITMAP, FLAG_FORCE_LANEMERGE, FLAG_FORCE_LANERANK = 256, 512, 1024
FLAG_FORCE_SPARSE = 2048
FLAG_FORCE_MASKTABLE = 4096
FLAG_FORCE_BITSLICE_LONG = 8192
FLAG_FORCE_SEARCH = 16384

ERROR_NAMES = {
    -1: "SDRB_ERR_INVALID_ARG",
    -2: "SDRB_ERR_NOT_ASCENDING",
    -3: "SDRB_ERR_ID_RANGE",
    -4: "SDRB_ERR_HIP",
    -5: "SDRB_ERR_UNSUPPORTED",
    -6: "SDRB_ERR_OVERFLOW",
    -7: "SDRB_ERR_PAYLOAD_RANGE",
    -8: "SDRB_ERR_IO",
    -9: "SDRB_ERR_FORMAT",
    -10: "SDRB_ERR_NO_MEMORY",
}


class SdrbError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


# name -> (restype, argtypes); every symbol include/sdr_amd.h declares
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p
PROTOTYPES = {
    "sdrb_last_error": (C.c_char_p, []),
    "sdrb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sdrb_batch_create": (C.c_int, [C.c_int, C.c_int32, C.c_int, C.POINTER(_vp)]),
    "sdrb_batch_create64": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "sdrb_batch_upload_csr64": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "sdrb_elem_op_fill64": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _vp, _vp, _vp]),
    "sdrb_batch_upload_csr": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "sdrb_csr_check": (C.c_int, [C.c_uint64, _vp, _vp, _vp, C.c_int32, C.c_int]),
    "sdrb_batch_upload_device_csr": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "sdrb_batch_adopt_device_csr": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp, C.c_uint32, C.c_int32, C.c_int32]),
    "sdrb_batch_id_span": (C.c_int, [_vp, _i32p, _i32p]),
    "sdrb_batch_table_width": (C.c_int, [_vp, _i32p]),
    "sdrb_batch_shift": (C.c_int, [_vp, C.c_int32, _vp]),
    "sdrb_batch_info": (C.c_int, [_vp, C.POINTER(C.c_int), _u64p, _u64p, _i32p, C.POINTER(C.c_int), _u32p]),
    "sdrb_batch_device_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "sdrb_batch_destroy": (C.c_int, [_vp]),
    "sdrb_batch_export_device_csr": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sdrb_file_writer_open": (C.c_int, [C.c_char_p, C.c_int32, C.c_int, C.POINTER(_vp)]),
    "sdrb_file_writer_append": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp]),
    "sdrb_file_writer_close": (C.c_int, [_vp]),
    "sdrb_file_writer_abort": (C.c_int, [_vp]),
    "sdrb_file_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int)]),
    "sdrb_file_row_span": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "sdrb_file_read_rows": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    "sdrb_batch_upload_file": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.c_uint64, _vp]),
    "sdrb_size_op": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, C.c_uint64, _vp]),
    "sdrb_size_op_narrow": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, C.c_uint64, C.c_int, _vp]),
    "sdrb_size_op_static": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, C.c_uint64, _vp]),
    "sdrb_range_size": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, _vp]),
    "sdrb_range_elems_count": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _u64p, _vp]),
    "sdrb_range_elems_fill": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdrb_point_query": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp]),
    "sdrb_batch_transpose": (C.c_int, [_vp, _vp, _vp]),
    "sdrb_inner": (C.c_int, [_vp, _vp, C.c_uint, _vp, C.c_uint64, _vp]),
    "sdrb_allpairs_topk": (C.c_int, [_vp, _vp, C.c_uint32, C.c_int64, C.c_uint, _vp, _vp, _vp]),
    "sdrb_elem_op_count": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _u64p, _vp]),
    "sdrb_elem_op_bound": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _u64p]),
    "sdrb_elem_op": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "sdrb_ande_few": (C.c_int, [_vp, _vp, C.c_uint, _vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "sdrb_equal": (C.c_int, [_vp, _vp, C.c_uint, _vp, C.c_uint64, _vp]),
    "sdrb_visit_count": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _u64p, _vp]),
    "sdrb_visit_fill": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _vp, _vp, _vp]),
    "sdrb_elem_op_fill": (C.c_int, [_vp, _vp, C.c_int, C.c_uint, _vp, _vp, _vp, _vp]),
    "sdrb_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "sdrb_free": (C.c_int, [C.c_int, _vp]),
    "sdrb_memcpy_h2d": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "sdrb_memcpy_d2h": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "sdrb_stream_sync": (C.c_int, [C.c_int, _vp]),
    "sdrb_stream_release": (C.c_int, [C.c_int, _vp]),
    "sdrb_synth_csr": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _vp, _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the engine; raises if it has not been built (``make -C sdr_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C sdr_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        # A process must hold ONE HIP runtime.  torch wheels bundle their own libamdhip64
        # (same soname as /opt/rocm's); importing torch first makes the loader resolve this
        # library's libamdhip64.so.7 to the copy torch already mapped, so stream handles and
        # device pointers are shared.  Without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise SdrbError(rc, load().sdrb_last_error().decode("utf-8", "replace"))
