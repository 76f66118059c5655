"""SLOW5 ASCII <-> BLOW5 for whole batches of records — thin ctypes wrappers.

The reference does this inside slow5_rec_depress_parse / slow5_rec_to_mem when one side of `view` is a .slow5 file
(/root/reference/src/view.c:35-57).  Here the raw_signal column is parsed / printed on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

REC_NONE, REC_ZLIB = 0, 1
SIG_NONE, SIG_SVB_ZD, SIG_EX_ZD = 0, 1, 2


def aux_types(types_line):
    """'#char*\\tuint32_t\\t...' (bytes or str) -> bytes of S5GPU_AUX_* codes for the columns after raw_signal."""
    L = _lib.lib()
    if isinstance(types_line, str):
        types_line = types_line.encode()
    buf = (C.c_uint8 * 1024)()
    n = L.s5gpu_aux_types_parse(types_line, len(types_line), buf, 1024)
    if n < 0:
        check(n, "s5gpu_aux_types_parse")
    return bytes(buf[:n])


def _ptrs(items):
    n = len(items)
    bufs = [C.create_string_buffer(bytes(b), max(len(b), 1)) for b in items]
    ptr = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(b) for b in items])
    return bufs, ptr, lens


def _collect(out, out_len, n):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    res = []
    for i in range(n):
        res.append(C.string_at(out[i], out_len[i]) if out[i] else None)
        libc.free(out[i])
    return res


def ascii_to_blow5(lines, types=b"", rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD, new_read_group=None, drop_aux=False, status=None):
    """Record lines -> BLOW5 records, each [u64 size][press(payload)]."""
    L = _lib.lib()
    n = len(lines)
    if n == 0:
        return []
    keep, ptr, lens = _ptrs(lines)
    tb = (C.c_uint8 * max(len(types), 1))(*types)
    rg = None if new_read_group is None else np.ascontiguousarray(new_read_group, dtype=np.uint32)
    out = (C.c_void_p * n)()
    out_len = (C.c_size_t * n)()
    st = (C.c_int32 * n)()
    rc = L.s5gpu_ascii_to_blow5_batch(n, ptr, lens, len(types), tb, rec_method, sig_method, rg.ctypes.data if rg is not None else None,
                                      int(drop_aux), out, out_len, st)
    if status is not None:
        status[:] = list(st)
    check(rc, "s5gpu_ascii_to_blow5_batch")
    return _collect(out, out_len, n)


def blow5_to_ascii(records, types=b"", rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD, new_read_group=None, drop_aux=False, status=None):
    """BLOW5 records (bytes without the u64 prefix) -> record lines ending in a newline."""
    L = _lib.lib()
    n = len(records)
    if n == 0:
        return []
    keep, ptr, lens = _ptrs(records)
    tb = (C.c_uint8 * max(len(types), 1))(*types)
    rg = None if new_read_group is None else np.ascontiguousarray(new_read_group, dtype=np.uint32)
    out = (C.c_void_p * n)()
    out_len = (C.c_size_t * n)()
    st = (C.c_int32 * n)()
    rc = L.s5gpu_blow5_to_ascii_batch(n, ptr, lens, rec_method, sig_method, len(types), tb, rg.ctypes.data if rg is not None else None,
                                      int(drop_aux), out, out_len, st)
    if status is not None:
        status[:] = list(st)
    check(rc, "s5gpu_blow5_to_ascii_batch")
    return _collect(out, out_len, n)
