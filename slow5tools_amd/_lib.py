"""ctypes binding of libslow5gpu.so (the C ABI in include/slow5gpu.h).

There is no fallback: if the library is missing it is built (hipcc); if that fails, or a call is
made without a gfx950 GPU, the error is raised to the caller.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None

REC_NONE, REC_ZLIB, REC_ZSTD = 0, 1, 2
SIG_NONE, SIG_SVB_ZD, SIG_EX_ZD = 0, 1, 2

# numpy mirrors of the C structs (same field order / padding as include/slow5gpu.h)
READ_DESC = np.dtype([("sig_off", "<u8"), ("hdr_off", "<u8"), ("aux_off", "<u8"), ("out_off", "<u8"),
                      ("n_samples", "<u4"), ("hdr_len", "<u4"), ("aux_len", "<u4"), ("slot_cap", "<u4")])
REC_DESC = np.dtype([("in_off", "<u8"), ("pay_off", "<u8"), ("sig_off", "<u8"),
                     ("in_len", "<u4"), ("pay_cap", "<u4"), ("sig_cap", "<u4"), ("reserved", "<u4")])
REC_FIELDS = np.dtype([("status", "<i4"), ("payload_len", "<u4"), ("n_samples", "<u4"), ("read_id_len", "<u4"),
                       ("read_group", "<u4"), ("aux_off", "<u4"), ("aux_len", "<u4"), ("reserved", "<u4"),
                       ("digitisation", "<f8"), ("offset", "<f8"), ("range", "<f8"), ("sampling_rate", "<f8")])
assert READ_DESC.itemsize == 48 and REC_DESC.itemsize == 40 and REC_FIELDS.itemsize == 64


class EncodeArgs(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("rec_method", C.c_int32), ("sig_method", C.c_int32),
                ("desc", C.c_void_p), ("sig", C.c_void_p), ("hdr", C.c_void_p), ("aux", C.c_void_p),
                ("slots", C.c_void_p), ("out_len", C.c_void_p), ("max_payload", C.c_uint32),
                ("lds_payload_cap", C.c_uint32), ("ovf", C.c_void_p)]


DEC_NO_PAYLOAD = 1


class DecodeArgs(C.Structure):
    _fields_ = [("n_recs", C.c_uint32), ("rec_method", C.c_int32), ("sig_method", C.c_int32), ("flags", C.c_uint32),
                ("desc", C.c_void_p), ("in_", C.c_void_p), ("payload", C.c_void_p), ("sig_out", C.c_void_p),
                ("fields", C.c_void_p), ("payload_bytes", C.c_uint64), ("max_pay_cap", C.c_uint32), ("max_in_len", C.c_uint32)]


assert C.sizeof(DecodeArgs) == 72


class S5GpuError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("S5GPU_LIB") or _build.LIB   # S5GPU_LIB: an experimental build made by tools/variant.sh
    if not os.path.exists(path):
        _build.build()
    # One HIP runtime per process: the torch wheel carries its own libamdhip64 / libhsa-runtime64.  If libslow5gpu.so pulled in
    # /opt/rocm's copies first, a later `import torch` would find "No HIP GPUs".  Python callers use torch for HBM buffers and
    # streams anyway, so let it load its runtime first; the library then binds to the copies already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.s5gpu_last_error.restype = C.c_char_p
    L.s5gpu_init.argtypes = [i32]
    L.s5gpu_init_mask.argtypes = [u64]
    L.s5gpu_devices_in_use.restype = i32
    L.s5gpu_recompress_batch.argtypes = [u32, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp]
    L.s5gpu_device_count.restype = i32
    L.s5gpu_slot_bound.restype = u64
    L.s5gpu_slot_bound.argtypes = [u32, u32, u32, i32, i32]
    L.s5gpu_payload_bound.restype = u64
    L.s5gpu_payload_bound.argtypes = [u32, u32, u32, i32]
    L.s5gpu_encode_dev.argtypes = [C.POINTER(EncodeArgs), vp]
    L.s5gpu_svbzd_encode_dev.argtypes = [C.POINTER(EncodeArgs), vp]
    L.s5gpu_decode_dev.argtypes = [C.POINTER(DecodeArgs), vp]
    L.s5gpu_compact_dev.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp]
    L.s5gpu_synth_dev.argtypes = [vp, u64, u64, u64, u64, u64, vp]
    L.s5gpu_synth_hdr_dev.argtypes = [vp, u64, u64, vp]
    L.s5gpu_event_create.argtypes = [C.POINTER(vp)]
    L.s5gpu_event_record.argtypes = [vp, vp]
    L.s5gpu_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.s5gpu_event_destroy.argtypes = [vp]
    L.s5gpu_encode_batch.argtypes = [u32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    L.s5gpu_decode_batch.argtypes = [u32, vp, vp, i32, i32, vp, vp, vp]
    L.s5gpu_encode_stream_dev.argtypes = [C.POINTER(EncodeArgs), vp, vp, vp, vp, vp]
    L.s5gpu_svbzd_encode_stream_dev.argtypes = [C.POINTER(EncodeArgs), vp, vp, vp, vp, vp]
    L.s5gpu_set_option.argtypes = [C.c_char_p, C.c_long]
    L.s5gpu_solo_batch.argtypes = [i32, u32, vp, vp, vp, vp, vp]
    L.s5gpu_deflate_parked_dev.argtypes = [C.POINTER(EncodeArgs), vp]
    L.s5gpu_pack_parked_dev.argtypes = [C.POINTER(EncodeArgs), vp]
    L.s5gpu_inflate_dev.argtypes = [C.POINTER(DecodeArgs), vp]
    L.s5gpu_svbzd_decode_dev.argtypes = [C.POINTER(DecodeArgs), vp]
    L.s5gpu_ascii_parse_dev.argtypes = [u32, vp, vp, vp, vp, vp]
    L.s5gpu_ascii_format_dev.argtypes = [u32, vp, vp, vp, vp, vp, vp]
    L.s5gpu_gather_dev.argtypes = [u32, vp, vp, vp, vp, vp, vp]
    L.s5gpu_aux_types_parse.argtypes = [C.c_char_p, C.c_size_t, vp, u32]
    L.s5gpu_ascii_to_blow5_batch.argtypes = [u32, vp, vp, u32, vp, i32, i32, vp, i32, vp, vp, vp]
    L.s5gpu_blow5_to_ascii_batch.argtypes = [u32, vp, vp, i32, i32, u32, vp, vp, i32, vp, vp, vp]
    _LIB = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().s5gpu_last_error().decode(errors="replace")
        raise S5GpuError("%s failed (rc=%d): %s" % (what or "libslow5gpu call", rc, msg))


EXPORTS = [
    "s5gpu_init", "s5gpu_init_mask", "s5gpu_devices_in_use", "s5gpu_shutdown", "s5gpu_last_error", "s5gpu_device_count", "s5gpu_slot_bound", "s5gpu_payload_bound",
    "s5gpu_encode_dev", "s5gpu_decode_dev", "s5gpu_svbzd_encode_dev", "s5gpu_compact_dev", "s5gpu_synth_dev",
    "s5gpu_synth_hdr_dev", "s5gpu_event_create", "s5gpu_event_record", "s5gpu_event_elapsed_ms", "s5gpu_event_destroy",
    "s5gpu_encode_batch", "s5gpu_decode_batch", "s5gpu_solo_batch", "s5gpu_deflate_parked_dev", "s5gpu_inflate_dev",
    "s5gpu_svbzd_decode_dev", "s5gpu_set_option", "s5gpu_recompress_batch", "s5gpu_patch_u32_dev", "s5gpu_encode_stream_dev",
    "s5gpu_svbzd_encode_stream_dev", "s5gpu_pack_parked_dev", "s5gpu_warmup",
]
