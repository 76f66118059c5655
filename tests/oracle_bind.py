"""ctypes binding of the CPU oracle (oracle/libs5oracle.so) — test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

REC_NONE, REC_ZLIB = 0, 1
SIG_NONE, SIG_SVB_ZD = 0, 1


class Rec(C.Structure):
    _fields_ = [
        ("read_id_len", C.c_uint16),
        ("read_id", C.c_char_p),
        ("read_group", C.c_uint32),
        ("digitisation", C.c_double),
        ("offset", C.c_double),
        ("range", C.c_double),
        ("sampling_rate", C.c_double),
        ("len_raw_signal", C.c_uint64),
        ("raw_signal", C.c_void_p),
        ("aux", C.c_void_p),
        ("aux_len", C.c_size_t),
    ]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "libs5oracle.so")
    try:   # cheap when up to date; keeps the .so in step with the sources
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(so):
            raise
    L = C.CDLL(so)
    L.s5o_svbzd_bound.restype = C.c_size_t
    L.s5o_svbzd_bound.argtypes = [C.c_uint64]
    L.s5o_svbzd_encode.restype = C.c_size_t
    L.s5o_svbzd_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.s5o_svbzd_decode.restype = C.c_int
    L.s5o_svbzd_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]
    L.s5o_zlib_bound.restype = C.c_size_t
    L.s5o_zlib_bound.argtypes = [C.c_size_t]
    L.s5o_zlib_compress.restype = C.c_int
    L.s5o_zlib_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.s5o_zlib_decompress.restype = C.c_int
    L.s5o_zlib_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.s5o_adler32.restype = C.c_uint32
    L.s5o_adler32.argtypes = [C.c_void_p, C.c_size_t]
    L.s5o_payload_bound.restype = C.c_size_t
    L.s5o_payload_bound.argtypes = [C.POINTER(Rec), C.c_int]
    L.s5o_rec_pack.restype = C.c_size_t
    L.s5o_rec_pack.argtypes = [C.POINTER(Rec), C.c_int, C.c_void_p]
    L.s5o_rec_to_mem_bound.restype = C.c_size_t
    L.s5o_rec_to_mem_bound.argtypes = [C.POINTER(Rec), C.c_int]
    L.s5o_rec_to_mem.restype = C.c_size_t
    L.s5o_rec_to_mem.argtypes = [C.POINTER(Rec), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.s5o_rec_parse.restype = C.c_int
    L.s5o_rec_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Rec), C.c_void_p]
    L.s5o_synth_read.restype = None
    L.s5o_synth_read.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.s5o_synth_read_id.restype = None
    L.s5o_synth_read_id.argtypes = [C.c_uint64, C.c_char_p]
    L.s5o_encode_batch_mt.restype = C.c_uint64
    L.s5o_encode_batch_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.s5o_decode_batch_mt.restype = C.c_uint64
    L.s5o_decode_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.s5o_convert_ascii_batch_mt.restype = C.c_uint64
    L.s5o_convert_ascii_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.s5o_exzd_bound.restype = C.c_size_t
    L.s5o_exzd_bound.argtypes = [C.c_uint64]
    L.s5o_exzd_encode.restype = C.c_size_t
    L.s5o_exzd_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.s5o_exzd_decode.restype = C.c_int
    L.s5o_exzd_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]
    L.s5o_aux_types.restype = C.c_int
    L.s5o_aux_types.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint]
    L.s5o_signal_to_text.restype = C.c_size_t
    L.s5o_signal_to_text.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.s5o_text_to_signal.restype = C.c_int64
    L.s5o_text_to_signal.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64]
    L.s5o_ascii_line_to_payload.restype = C.c_size_t
    L.s5o_ascii_line_to_payload.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint, C.c_void_p]
    L.s5o_payload_to_ascii_line.restype = C.c_size_t
    L.s5o_payload_to_ascii_line.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint, C.c_void_p]
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def svbzd_encode(x):
    x = np.ascontiguousarray(x, dtype=np.int16)
    out = np.empty(lib().s5o_svbzd_bound(len(x)), dtype=np.uint8)
    n = lib().s5o_svbzd_encode(_ptr(x), len(x), _ptr(out))
    return out[:n].tobytes()


def svbzd_decode(blob):
    b = np.frombuffer(blob, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().s5o_svbzd_decode(_ptr(b), len(b), None, C.byref(n))
    if rc != 0:
        raise ValueError("svbzd header rc=%d" % rc)
    out = np.empty(n.value, dtype=np.int16)
    rc = lib().s5o_svbzd_decode(_ptr(b), len(b), _ptr(out), C.byref(n))
    if rc != 0:
        raise ValueError("svbzd decode rc=%d" % rc)
    return out


def zlib_compress(data):
    b = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(lib().s5o_zlib_bound(len(b)), dtype=np.uint8)
    n = C.c_size_t(len(out))
    rc = lib().s5o_zlib_compress(_ptr(b), len(b), _ptr(out), C.byref(n))
    assert rc == 0, rc
    return out[: n.value].tobytes()


def zlib_decompress(data, cap):
    b = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_size_t(cap)
    rc = lib().s5o_zlib_decompress(_ptr(b), len(b), _ptr(out), C.byref(n))
    if rc != 0:
        raise ValueError("inflate rc=%d" % rc)
    return out[: n.value].tobytes()


def adler32(data):
    b = np.frombuffer(data, dtype=np.uint8)
    return lib().s5o_adler32(_ptr(b), len(b))


def make_rec(read_id, read_group, digitisation, offset, rng, sampling_rate, signal, aux=b""):
    """returns (Rec, keepalive) — keepalive holds the buffers the struct points to."""
    rid = read_id if isinstance(read_id, bytes) else read_id.encode()
    sig = np.ascontiguousarray(signal, dtype=np.int16)
    auxb = np.frombuffer(aux, dtype=np.uint8) if aux else None
    r = Rec()
    r.read_id_len = len(rid)
    r.read_id = rid
    r.read_group = read_group
    r.digitisation, r.offset, r.range, r.sampling_rate = digitisation, offset, rng, sampling_rate
    r.len_raw_signal = len(sig)
    r.raw_signal = sig.ctypes.data
    r.aux = auxb.ctypes.data if auxb is not None else None
    r.aux_len = len(aux)
    return r, (rid, sig, auxb)


def rec_pack(rec, sig_method):
    out = np.empty(lib().s5o_payload_bound(C.byref(rec), sig_method), dtype=np.uint8)
    n = lib().s5o_rec_pack(C.byref(rec), sig_method, _ptr(out))
    return out[:n].tobytes()


def rec_to_mem(rec, rec_method, sig_method):
    scratch = np.empty(lib().s5o_payload_bound(C.byref(rec), sig_method), dtype=np.uint8)
    out = np.empty(lib().s5o_rec_to_mem_bound(C.byref(rec), sig_method), dtype=np.uint8)
    n = lib().s5o_rec_to_mem(C.byref(rec), rec_method, sig_method, _ptr(scratch), _ptr(out))
    assert n > 0
    return out[:n].tobytes()


def rec_parse(payload, sig_method):
    """returns dict of primary fields + signal + aux bytes"""
    b = np.frombuffer(payload, dtype=np.uint8)
    r = Rec()
    rc = lib().s5o_rec_parse(_ptr(b), len(b), sig_method, C.byref(r), None)
    if rc != 0:
        raise ValueError("rec_parse rc=%d" % rc)
    sig = np.empty(r.len_raw_signal, dtype=np.int16)
    rc = lib().s5o_rec_parse(_ptr(b), len(b), sig_method, C.byref(r), _ptr(sig))
    if rc != 0:
        raise ValueError("rec_parse rc=%d" % rc)
    base = b.ctypes.data
    id_off = 2
    aux_off = r.aux - base
    return dict(
        read_id=payload[id_off : id_off + r.read_id_len],
        read_group=r.read_group,
        digitisation=r.digitisation,
        offset=r.offset,
        range=r.range,
        sampling_rate=r.sampling_rate,
        signal=sig,
        aux=payload[aux_off : aux_off + r.aux_len],
    )


def synth_read(seed, read_idx, n):
    out = np.empty(n, dtype=np.int16)
    lib().s5o_synth_read(seed, read_idx, n, _ptr(out))
    return out


def synth_reads(seed, first, count, n):
    out = np.empty((count, n), dtype=np.int16)
    for i in range(count):
        lib().s5o_synth_read(seed, first + i, n, out[i].ctypes.data_as(C.c_void_p))
    return out


def synth_read_id(read_idx):
    buf = C.create_string_buffer(37)
    lib().s5o_synth_read_id(read_idx, buf)
    return buf.value


def encode_batch_mt(sig2d, first_idx, n_threads, batch_size=4096, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD, pooled_zstream=False):
    """pooled_zstream: NOT the reference's shape — one deflate state per worker thread, deflateReset per record (bench: how much of the CPU
    figure is the reference's per-record slow5_press_init)"""
    sig2d = np.ascontiguousarray(sig2d, dtype=np.int16)
    C.c_int.in_dll(lib(), "s5o_pool_zstream").value = 1 if pooled_zstream else 0
    secs = C.c_double()
    ck = C.c_uint64()
    total = lib().s5o_encode_batch_mt(_ptr(sig2d), sig2d.shape[0], sig2d.shape[1], first_idx, rec_method, sig_method,
                                      n_threads, batch_size, C.byref(secs), C.byref(ck))
    return total, secs.value, ck.value


def decode_batch_mt(stream, rec_off, ids, n_threads, batch_size=4096, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD):
    """compute phase of `get --benchmark` on the CPU: returns (samples decoded, seconds, checksum)"""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    secs = C.c_double()
    ck = C.c_uint64()
    total = lib().s5o_decode_batch_mt(_ptr(stream), _ptr(rec_off), _ptr(ids), ids.size, rec_method, sig_method, n_threads,
                                      batch_size, C.byref(secs), C.byref(ck))
    return total, secs.value, ck.value


def convert_ascii_batch_mt(text, line_off, n_threads, batch_size=4096, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD):
    """the whole view worker on SLOW5 text (line parse + svb-zd + zlib) on the CPU: returns (output bytes, seconds, checksum)"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    line_off = np.ascontiguousarray(line_off, dtype=np.uint64)
    secs = C.c_double()
    ck = C.c_uint64()
    total = lib().s5o_convert_ascii_batch_mt(_ptr(text), _ptr(line_off), line_off.size - 1, rec_method, sig_method, n_threads, batch_size,
                                             C.byref(secs), C.byref(ck))
    return total, secs.value, ck.value


def view_file(in_path, out_path, n_threads, batch_size=4096, max_reads=0):
    """the CPU twin of the whole `view` loop on files (oracle/batch.c s5o_view_file): returns (records, dict of phase seconds)"""
    ph = (C.c_double * 4)()
    L = lib()
    L.s5o_view_file.restype = C.c_uint64
    L.s5o_view_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    n = L.s5o_view_file(str(in_path).encode(), str(out_path).encode(), n_threads, batch_size, max_reads, ph)
    return int(n), dict(read=ph[0], compute=ph[1], write=ph[2], first_read_to_last_write=ph[3])


def get_file(path, pos, length, n_threads, batch_size=4096):
    """the CPU twin of `get --benchmark` on a file: returns (samples decoded, seconds)"""
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    secs = C.c_double()
    L = lib()
    L.s5o_get_file.restype = C.c_uint64
    L.s5o_get_file.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    n = L.s5o_get_file(str(path).encode(), pos.ctypes.data, length.ctypes.data, pos.size, n_threads, batch_size, C.byref(secs))
    return int(n), secs.value


# ---- §8f row 2: SLOW5 ASCII ----
def aux_types(types_line):
    buf = (C.c_uint8 * 1024)()
    n = lib().s5o_aux_types(types_line, len(types_line), buf, 1024)
    assert n >= 0, "bad types line"
    return bytes(buf[:n])


def signal_to_text(sig):
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    out = C.create_string_buffer(7 * sig.size + 16)
    n = lib().s5o_signal_to_text(_ptr(sig), sig.size, out)
    return out.raw[:n]


def text_to_signal(txt, cap=None):
    cap = cap if cap is not None else len(txt) // 2 + 2
    out = np.empty(cap, dtype=np.int16)
    n = lib().s5o_text_to_signal(txt, len(txt), _ptr(out), cap)
    return None if n < 0 else out[:n].copy()


def line_to_payload(line, types=b""):
    out = C.create_string_buffer(2 * len(line) + 64)
    n = lib().s5o_ascii_line_to_payload(line, len(line), types, len(types), out)
    return out.raw[:n] if n else None


def payload_to_line(payload, types=b""):
    out = C.create_string_buffer(8 * len(payload) + 512)
    n = lib().s5o_payload_to_ascii_line(payload, len(payload), types, len(types), out)
    return out.raw[:n] if n else None


# ---- §8f row 4: ex-zd ----
SIG_EX_ZD = 2


def exzd_encode(x):
    x = np.ascontiguousarray(x, dtype=np.int16)
    out = np.empty(lib().s5o_exzd_bound(x.size), dtype=np.uint8)
    n = lib().s5o_exzd_encode(_ptr(x), x.size, _ptr(out))
    return out[:n].tobytes()


def exzd_decode(blob):
    b = np.frombuffer(blob, dtype=np.uint8)
    n = C.c_uint64(0)
    if lib().s5o_exzd_decode(_ptr(b), len(b), None, C.byref(n)) != 0:
        return None
    out = np.empty(n.value, dtype=np.int16)
    if lib().s5o_exzd_decode(_ptr(b), len(b), _ptr(out), C.byref(n)) != 0:
        return None
    return out


# ---- §8f row 4: zstd record press ----
# The reference calls libzstd (a dependency, not vendored).  The real library is the oracle where the image has it
# (libzstd.so.1, no headers needed); oracle/zstd_dec.c is a restatement of the frame decoder that is pinned against it.
REC_ZSTD = 2
_ZSTD = None


def zstd_ref():
    """the real libzstd through ctypes, or None when the image has none"""
    global _ZSTD
    if _ZSTD is None:
        try:
            Z = C.CDLL("libzstd.so.1")
        except OSError:
            _ZSTD = False
            return None
        Z.ZSTD_compressBound.restype = C.c_size_t
        Z.ZSTD_compressBound.argtypes = [C.c_size_t]
        Z.ZSTD_compress.restype = C.c_size_t
        Z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        Z.ZSTD_decompress.restype = C.c_size_t
        Z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        Z.ZSTD_isError.restype = C.c_uint
        Z.ZSTD_isError.argtypes = [C.c_size_t]
        Z.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        Z.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        _ZSTD = Z
    return _ZSTD or None


def zstd_compress(data, level=1):
    Z = zstd_ref()
    cap = Z.ZSTD_compressBound(len(data))
    out = C.create_string_buffer(cap)
    n = Z.ZSTD_compress(out, cap, data, len(data), level)
    assert not Z.ZSTD_isError(n)
    return out.raw[:n]


def zstd_decompress(frame, cap=None):
    """libzstd's answer; None when it rejects the frame"""
    Z = zstd_ref()
    if cap is None:
        cap = Z.ZSTD_getFrameContentSize(frame, len(frame))
        if cap >= 2**63:
            return None
    out = C.create_string_buffer(cap + 1)
    n = Z.ZSTD_decompress(out, cap, frame, len(frame))
    return None if Z.ZSTD_isError(n) else out.raw[:n]


def zstd_restated_decompress(frame, cap):
    L = lib()
    L.s5o_zstd_restated_decompress.restype = C.c_size_t
    L.s5o_zstd_restated_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    out = C.create_string_buffer(cap + 1)
    n = L.s5o_zstd_restated_decompress(frame, len(frame), out, cap)
    return None if n == 2**64 - 1 else out.raw[:n]


def zstd_literals_compress(data):
    """oracle/zstd_enc.c: the frame layout of the device encoder, stated on the CPU"""
    L = lib()
    L.s5o_zstd_literals_bound.restype = C.c_size_t
    L.s5o_zstd_literals_bound.argtypes = [C.c_size_t]
    L.s5o_zstd_literals_compress.restype = C.c_size_t
    L.s5o_zstd_literals_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    out = C.create_string_buffer(L.s5o_zstd_literals_bound(len(data)))
    n = L.s5o_zstd_literals_compress(data, len(data), out)
    return out.raw[:n]
