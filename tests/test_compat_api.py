"""slow5lib-compatible C entry points (include/slow5_compat.h) exercised the way slow5tools' workers use
them: src/view.c:35-57 (depress_parse -> press_init -> rec_to_mem), src/merge.c:43-70 (read_group rewrite),
src/get.c:37-66.  GPU tests; the results are checked against the oracle / golden fixtures."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, golden

pytestmark = pytest.mark.gpu

NONE, ZLIB, SVB = 0, 1, 2


class PressMethod(C.Structure):
    _fields_ = [("record_method", C.c_int), ("signal_method", C.c_int)]


class InnerPress(C.Structure):
    _fields_ = [("method", C.c_int), ("stream", C.c_void_p)]


class Press(C.Structure):
    _fields_ = [("record_press", C.POINTER(InnerPress)), ("signal_press", C.POINTER(InnerPress))]


class Rec(C.Structure):
    _fields_ = [("read_id_len", C.c_uint16), ("read_id", C.c_void_p), ("read_group", C.c_uint32),
                ("digitisation", C.c_double), ("offset", C.c_double), ("range", C.c_double),
                ("sampling_rate", C.c_double), ("len_raw_signal", C.c_uint64), ("raw_signal", C.c_void_p),
                ("aux_blob", C.c_void_p), ("aux_len", C.c_uint64)]


class File(C.Structure):
    _fields_ = [("fp", C.c_void_p), ("format", C.c_int), ("compress", C.POINTER(Press)), ("header", C.c_void_p),
                ("index", C.c_void_p), ("pathname", C.c_char_p)]


@pytest.fixture(scope="module")
def L():
    from slow5tools_amd import _lib

    lib = _lib.lib()
    _lib.check(lib.s5gpu_init(0), "s5gpu_init")
    lib.slow5_press_init.restype = C.POINTER(Press)
    lib.slow5_press_init.argtypes = [PressMethod]
    lib.slow5_press_free.argtypes = [C.POINTER(Press)]
    lib.slow5_rec_to_mem.restype = C.c_void_p
    lib.slow5_rec_to_mem.argtypes = [C.POINTER(Rec), C.c_void_p, C.c_int, C.POINTER(Press), C.POINTER(C.c_size_t)]
    lib.slow5_rec_depress_parse.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p,
                                            C.POINTER(C.POINTER(Rec)), C.POINTER(File)]
    lib.slow5_rec_free.argtypes = [C.POINTER(Rec)]
    lib.slow5_ptr_compress_solo.restype = C.c_void_p
    lib.slow5_ptr_compress_solo.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.slow5_ptr_depress_solo.restype = C.c_void_p
    lib.slow5_ptr_depress_solo.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.slow5_gpu_recompress_batch.argtypes = [C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), PressMethod,
                                               PressMethod, C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_size_t)]
    return lib


libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]


def _malloc_copy(b):
    p = libc.malloc(max(len(b), 1))
    C.memmove(p, b, len(b))
    return p


def test_press_init_rejects_unimplemented_codecs(L):
    assert not L.slow5_press_init(PressMethod(7, 0))      # no such method
    z = L.slow5_press_init(PressMethod(3, SVB))           # zstd record press
    assert z and z.contents.record_press.contents.method == 3
    L.slow5_press_free(z)
    assert not L.slow5_press_init(PressMethod(4, 0))      # ex-zd is a signal press, not a record press
    q = L.slow5_press_init(PressMethod(ZLIB, 4))          # zlib + ex-zd: the `degrade` default
    assert q and q.contents.signal_press.contents.method == 4
    L.slow5_press_free(q)
    p = L.slow5_press_init(PressMethod(ZLIB, SVB))
    assert p and p.contents.record_press.contents.method == ZLIB and p.contents.signal_press.contents.method == SVB
    L.slow5_press_free(p)


def test_view_worker_sequence_on_fixture(L):
    """depress_parse(zlib+svb fixture record) -> rec_to_mem(none,none) == the reference's uncompressed golden"""
    src = Blow5(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"))
    want = Blow5(golden("exp_1_lossless.blow5"))
    press_in = L.slow5_press_init(PressMethod(ZLIB, SVB))
    f = File(None, 2, press_in, None, None, None)
    mem = C.c_void_p(_malloc_copy(src.records[0]))
    nbytes = C.c_size_t(len(src.records[0]))
    rec = C.POINTER(Rec)()
    assert L.slow5_rec_depress_parse(C.byref(mem), C.byref(nbytes), None, C.byref(rec), C.byref(f)) == 0
    libc.free(mem)                                            # src/view.c:41
    r = rec.contents
    assert r.len_raw_signal == 59676 and C.string_at(r.read_id) == b"a649a4ae-c43d-492a-b6a1-a5b8b8076be4"
    out_press = L.slow5_press_init(PressMethod(NONE, NONE))
    n = C.c_size_t()
    buf = L.slow5_rec_to_mem(rec, C.c_void_p(1), 2, out_press, C.byref(n))
    assert buf
    got = C.string_at(buf, n.value)
    libc.free(buf)
    assert got == struct.pack("<Q", len(want.records[0])) + want.records[0]     # byte-identical to the golden file
    # and back to zlib+svb: stock zlib must inflate it to the golden's payload
    zs = L.slow5_press_init(PressMethod(ZLIB, SVB))
    buf = L.slow5_rec_to_mem(rec, C.c_void_p(1), 2, zs, C.byref(n))
    got = C.string_at(buf, n.value)
    libc.free(buf)
    assert zlib.decompress(got[8:]) == zlib.decompress(src.records[0])
    # aux_meta == NULL drops the aux fields (lossy, src/merge.c:58-62)
    buf = L.slow5_rec_to_mem(rec, None, 2, out_press, C.byref(n))
    lossy = C.string_at(buf, n.value)
    libc.free(buf)
    assert lossy == got_prefix(want.records[0], r.aux_len)
    for p in (press_in, out_press, zs):
        L.slow5_press_free(p)
    L.slow5_rec_free(rec)


def got_prefix(full_record, aux_len):
    body = full_record[: len(full_record) - aux_len]
    return struct.pack("<Q", len(body)) + body


def test_solo_press_calls(L):
    rng = np.random.default_rng(2)
    sig = (500 + 40 * rng.standard_normal(30000)).astype(np.int16)
    n = C.c_size_t()
    p = L.slow5_ptr_compress_solo(SVB, sig.ctypes.data, sig.nbytes, C.byref(n))
    blob = C.string_at(p, n.value)
    libc.free(p)
    assert blob == ob.svbzd_encode(sig)                       # bit-exact
    p = L.slow5_ptr_depress_solo(SVB, blob, len(blob), C.byref(n))
    back = np.frombuffer(C.string_at(p, n.value), dtype=np.int16)
    libc.free(p)
    assert np.array_equal(back, sig)
    for data in (b"", b"a", bytes(1000), blob, rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()):
        p = L.slow5_ptr_compress_solo(ZLIB, data, len(data), C.byref(n))
        z = C.string_at(p, n.value)
        libc.free(p)
        assert zlib.decompress(z) == data
        ref = zlib.compress(data, 6)
        p = L.slow5_ptr_depress_solo(ZLIB, ref, len(ref), C.byref(n))
        assert C.string_at(p, n.value) == data
        libc.free(p)
    assert not L.slow5_ptr_depress_solo(ZLIB, b"\x78\x9c\x01\x02", 4, C.byref(n))      # truncated stream -> NULL
    for data in (b"", b"a", bytes(1000), blob, rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()):   # zstd (method 3) both ways
        p = L.slow5_ptr_compress_solo(3, data, len(data), C.byref(n))
        z = C.string_at(p, n.value)
        libc.free(p)
        assert ob.zstd_restated_decompress(z, len(data)) == data
        p = L.slow5_ptr_depress_solo(3, z, len(z), C.byref(n))
        assert C.string_at(p, n.value) == data
        libc.free(p)
    assert not L.slow5_ptr_depress_solo(3, b"\x28\xb5\x2f\xfd\x20", 5, C.byref(n))


def test_merge_batch_rewrites_read_group(L):
    """slow5_gpu_recompress_batch = the merge worker over a whole batch (src/merge.c:43-70)"""
    src = Blow5(golden("merged_expected_zlib_svb.blow5"))
    n = len(src.records)
    mem = (C.c_void_p * n)(*[_malloc_copy(r) for r in src.records])
    nb = (C.c_size_t * n)(*[len(r) for r in src.records])
    rg = (C.c_uint32 * n)(*[10 + i for i in range(n)])
    out = (C.c_void_p * n)()
    ol = (C.c_size_t * n)()
    assert L.slow5_gpu_recompress_batch(n, mem, nb, PressMethod(ZLIB, SVB), PressMethod(ZLIB, SVB), rg, 0, out, ol) == 0
    for i in range(n):
        got = C.string_at(out[i], ol[i])
        libc.free(out[i])
        pl = zlib.decompress(got[8:])
        ref = zlib.decompress(src.records[i])
        idl = struct.unpack_from("<H", ref, 0)[0]
        want = ref[: 2 + idl] + struct.pack("<I", 10 + i) + ref[2 + idl + 4:]
        assert pl == want


def test_arena_form_of_the_view_hook_hands_out_the_same_bytes_with_one_release(L):
    """slow5_gpu_hook_recompress_arena / slow5_gpu_hook_release (include/slow5gpu_hooks.h): the view worker of src/view.c:292 with the
    free loop of src/view.c:296-299 replaced by ONE release — records identical to the malloc form's, the pointers lie inside a few
    library-owned buffers, a second batch reuses them, and a failing call leaves no batch behind"""
    L.slow5_gpu_hook_recompress.argtypes = [C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.slow5_gpu_hook_recompress_arena.argtypes = L.slow5_gpu_hook_recompress.argtypes + [C.POINTER(C.c_void_p)]
    L.slow5_gpu_hook_release.argtypes = [C.c_void_p]
    src = Blow5(golden("merged_expected_zlib_svb.blow5"))
    recs = src.records * 40                     # 240 records, 15 k .. 243 k samples
    n = len(recs)

    def batch():
        return (C.c_void_p * n)(*[_malloc_copy(r) for r in recs]), (C.c_size_t * n)(*[len(r) for r in recs])

    mem, nb = batch()
    out = (C.c_void_p * n)()
    ol = (C.c_size_t * n)()
    assert L.slow5_gpu_hook_recompress(n, mem, nb, ZLIB, SVB, ZLIB, SVB, None, 0, out, ol) == 0
    want = [C.string_at(out[i], ol[i]) for i in range(n)]
    for i in range(n):
        libc.free(out[i])
    seen = []
    for rep in range(3):
        mem, nb = batch()
        h = C.c_void_p()
        assert L.slow5_gpu_hook_recompress_arena(n, mem, nb, ZLIB, SVB, ZLIB, SVB, None, 0, out, ol, C.byref(h)) == 0
        assert h.value and all(mem[i] is None for i in range(n))          # inputs freed like the reference's worker does
        assert [C.string_at(out[i], ol[i]) for i in range(n)] == want
        addr = sorted(out[i] for i in range(n))
        assert addr[-1] - addr[0] < 2 * sum(ol) + (16 << 20)               # one arena buffer, not n allocations
        seen.append(addr[0])
        L.slow5_gpu_hook_release(h)
    assert seen[1] == seen[2]                                              # the pool hands the same buffer out again
    for i, r in enumerate(want[:6]):
        assert zlib.decompress(r[8:]) == zlib.decompress(src.records[i])
    # a corrupt record fails the whole call, *batch stays NULL, nothing to release
    mem, nb = batch()
    bad = bytearray(recs[3]); bad[len(bad) // 2] ^= 0x55
    libc.free(mem[3]); mem[3] = _malloc_copy(bytes(bad))
    h = C.c_void_p(1)
    assert L.slow5_gpu_hook_recompress_arena(n, mem, nb, ZLIB, SVB, ZLIB, SVB, None, 0, out, ol, C.byref(h)) == -1
    assert h.value is None
    for i in range(n):
        if mem[i]:
            libc.free(mem[i])
    L.slow5_gpu_hook_release(None)


def test_submit_wait_pair_keeps_two_batches_in_flight_and_equals_the_synchronous_hook(L):
    """slow5_gpu_hook_recompress_submit / _wait (round 6): the loop of /root/reference/src/view.c:254-300 with batch k + 1 read while batch k
    is on the device.  Six batches, two tickets in flight at any time, two alternating sets of arrays: every batch's records equal the
    synchronous hook's byte for byte, inputs are freed by the time _wait returns, a corrupt record fails ITS ticket only (message in the
    waiting thread), a NULL ticket is an error"""
    hk = [C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.slow5_gpu_hook_recompress.argtypes = hk
    L.slow5_gpu_hook_recompress_submit.argtypes = hk
    L.slow5_gpu_hook_recompress_submit.restype = C.c_void_p
    L.slow5_gpu_hook_recompress_wait.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.slow5_gpu_hook_release.argtypes = [C.c_void_p]
    L.slow5_gpu_hook_error.restype = C.c_char_p
    src = Blow5(golden("merged_expected_zlib_svb.blow5"))
    base = src.records * 20                      # 120 records
    n = len(base)
    batches = [base[k:] + base[:k] for k in range(6)]          # six different orders

    def arrays(recs):
        return (C.c_void_p * n)(*[_malloc_copy(r) for r in recs]), (C.c_size_t * n)(*[len(r) for r in recs]), (C.c_void_p * n)(), (C.c_size_t * n)()

    want = []
    for recs in batches[:2]:
        mem, nb, out, ol = arrays(recs)
        assert L.slow5_gpu_hook_recompress(n, mem, nb, ZLIB, SVB, ZLIB, SVB, None, 0, out, ol) == 0
        want.append([C.string_at(out[i], ol[i]) for i in range(n)])
        for i in range(n):
            libc.free(out[i])
    by_first = {w[0]: w for w in want}

    flight = []                                  # (ticket, arrays): at most two
    done = 0
    for k, recs in enumerate(batches):
        a = arrays(recs)
        t = L.slow5_gpu_hook_recompress_submit(n, a[0], a[1], ZLIB, SVB, ZLIB, SVB, None, 0, a[2], a[3])
        assert t
        flight.append((t, a, k))
        if len(flight) == 2:
            t0, a0, k0 = flight.pop(0)
            h = C.c_void_p()
            assert L.slow5_gpu_hook_recompress_wait(t0, C.byref(h)) == 0 and h.value
            assert all(a0[0][i] is None for i in range(n))
            got = [C.string_at(a0[2][i], a0[3][i]) for i in range(n)]
            if k0 < 2:
                assert got == want[k0]
            else:                                # a rotation of batch 0: the same records in another order
                assert sorted(got) == sorted(want[0])
            L.slow5_gpu_hook_release(h)
            done += 1
    t0, a0, k0 = flight.pop(0)
    h = C.c_void_p()
    assert L.slow5_gpu_hook_recompress_wait(t0, C.byref(h)) == 0
    assert sorted(C.string_at(a0[2][i], a0[3][i]) for i in range(n)) == sorted(want[0])
    L.slow5_gpu_hook_release(h)
    # one good and one corrupt batch in flight together
    good = arrays(batches[0])
    badrecs = list(batches[1])
    b = bytearray(badrecs[7]); b[len(b) // 2] ^= 0x55; badrecs[7] = bytes(b)
    bad = arrays(badrecs)
    tg = L.slow5_gpu_hook_recompress_submit(n, good[0], good[1], ZLIB, SVB, ZLIB, SVB, None, 0, good[2], good[3])
    tb = L.slow5_gpu_hook_recompress_submit(n, bad[0], bad[1], ZLIB, SVB, ZLIB, SVB, None, 0, bad[2], bad[3])
    hb = C.c_void_p(1)
    assert L.slow5_gpu_hook_recompress_wait(tb, C.byref(hb)) == -1 and hb.value is None
    assert L.slow5_gpu_hook_error()              # the failing batch's message reached the waiting thread
    hg = C.c_void_p()
    assert L.slow5_gpu_hook_recompress_wait(tg, C.byref(hg)) == 0
    assert [C.string_at(good[2][i], good[3][i]) for i in range(n)] == want[0]
    L.slow5_gpu_hook_release(hg)
    for i in range(n):
        if bad[0][i]:
            libc.free(bad[0][i])
    assert L.slow5_gpu_hook_recompress_wait(None, None) == -1
    assert L.slow5_gpu_hook_recompress_submit(n, good[0], good[1], 99, SVB, ZLIB, SVB, None, 0, good[2], good[3]) is None


def test_arena_form_of_encode_batch_at_the_reference_batch_sizes(L):
    """s5gpu_encode_batch_arena at K = 4096 (/root/reference/src/cmd.h:8) and K = 10 000 (test/test_view_integrity.sh:62-66): the records
    of the malloc form, byte for byte"""
    from slow5tools_amd import press

    L.s5gpu_encode_batch_arena.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.s5gpu_arena_release.argtypes = [C.c_void_p]
    for n in (4096, 10000):
        ns_ = [1000 + (i * 37) % 5000 for i in range(n)]
        sig = [ob.synth_read(0x5105, i % 64, ns_[i]) for i in range(n)]
        hdrs = [press.pack_hdr(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0) for i in range(n)]
        want = press.encode_records(sig, hdrs)
        vp = C.c_void_p
        sig_p = (vp * n)(*[s.ctypes.data for s in sig])
        nsa = (C.c_uint64 * n)(*ns_)
        hb = [C.create_string_buffer(h, len(h)) for h in hdrs]
        hdr_p = (vp * n)(*[C.addressof(b) for b in hb])
        hl = (C.c_uint32 * n)(*[len(h) for h in hdrs])
        out = (vp * n)()
        ol = (C.c_size_t * n)()
        h = vp()
        assert L.s5gpu_encode_batch_arena(n, sig_p, nsa, hdr_p, hl, None, None, 1, 1, out, ol, C.byref(h)) == 0
        assert [C.string_at(out[i], ol[i]) for i in range(n)] == want
        L.s5gpu_arena_release(h)
