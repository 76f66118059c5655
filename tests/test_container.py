"""BLOW5 container framing + index through include/slow5_compat.h, and the end-to-end view loop
(examples/s5view.c).  Mirrors the reference's golden-file diffs: test/test_view.sh:142-149 (zlib+svb -> uncompressed),
test/test_index.sh cases 3 and 4 (.idx byte-identical), test/test_get.sh."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S5VIEW = os.path.join(ROOT, "slow5tools_amd", "s5view")


class Version(C.Structure):
    _fields_ = [("major", C.c_uint8), ("minor", C.c_uint8), ("patch", C.c_uint8)]


class Hdr(C.Structure):
    _fields_ = [("version", Version), ("num_read_groups", C.c_uint32), ("data", C.c_void_p), ("data_len", C.c_uint32),
                ("aux_meta", C.c_void_p)]


class InnerPress(C.Structure):
    _fields_ = [("method", C.c_int), ("stream", C.c_void_p)]


class Press(C.Structure):
    _fields_ = [("record_press", C.POINTER(InnerPress)), ("signal_press", C.POINTER(InnerPress))]


class File(C.Structure):
    _fields_ = [("fp", C.c_void_p), ("format", C.c_int), ("compress", C.POINTER(Press)), ("header", C.POINTER(Hdr)),
                ("index", C.c_void_p), ("pathname", C.c_char_p), ("start_rec_offset", C.c_uint64)]


class PressMethod(C.Structure):
    _fields_ = [("record_method", C.c_int), ("signal_method", C.c_int)]


@pytest.fixture(scope="module")
def L():
    from slow5tools_amd import _lib

    lib = _lib.lib()
    lib.slow5_open.restype = C.POINTER(File)
    lib.slow5_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.slow5_close.argtypes = [C.POINTER(File)]
    lib.slow5_get_next_mem.restype = C.c_void_p
    lib.slow5_get_next_mem.argtypes = [C.POINTER(C.c_size_t), C.POINTER(File)]
    lib.slow5_hdr_fwrite.argtypes = [C.c_void_p, C.POINTER(Hdr), C.c_int, PressMethod]
    lib.slow5_eof_fwrite.restype = C.c_long
    lib.slow5_eof_fwrite.argtypes = [C.c_void_p]
    lib.slow5_idx_load.argtypes = [C.POINTER(File)]
    lib.slow5_get_mem.restype = C.c_void_p
    lib.slow5_get_mem.argtypes = [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(File)]
    return lib


libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
libc.free.argtypes = [C.c_void_p]


def _errno(L):
    return C.c_int.in_dll(L, "slow5_errno").value if False else None   # thread-local: not readable via in_dll


@pytest.mark.parametrize("name", ["exp_1_lossless_zlib_svb_v0.2.0.blow5", "example_multi_rg_v0.2.0.blow5", "exp_1_lossless.blow5",
                                  "merged_expected_zlib_svb.blow5"])
def test_open_and_sequential_framing(L, name):
    ref = Blow5(golden(name))
    f = L.slow5_open(golden(name).encode(), b"r")
    assert f
    h = f.contents.header.contents
    assert (h.version.major, h.version.minor, h.version.patch) == ref.version
    assert h.num_read_groups == ref.num_read_groups
    assert C.string_at(h.data, h.data_len) == ref.header_text
    # API enum: record none/zlib = 0/1, signal none/svb-zd = 0/2
    assert f.contents.compress.contents.record_press.contents.method == ref.rec_method
    assert f.contents.compress.contents.signal_press.contents.method == (2 if ref.sig_method == 1 else 0)
    got = []
    n = C.c_size_t()
    while True:
        p = L.slow5_get_next_mem(C.byref(n), f)
        if not p:
            break
        got.append(C.string_at(p, n.value))
        libc.free(p)
    assert got == ref.records
    L.slow5_close(f)


def test_open_rejects_bad_files(L, tmp_path):
    bad = tmp_path / "bad.blow5"
    bad.write_bytes(b"SLOW5\x01" + bytes(100))
    assert not L.slow5_open(str(bad).encode(), b"r")
    trunc = tmp_path / "trunc.blow5"
    trunc.write_bytes(open(golden("sp1_dna.blow5"), "rb").read()[:40])
    assert not L.slow5_open(str(trunc).encode(), b"r")
    assert not L.slow5_open(str(tmp_path / "missing.blow5").encode(), b"r")
    # a record cut short is reported by the sequential reader, not silently dropped (quickcheck-style)
    cut = tmp_path / "cut.blow5"
    raw = open(golden("sp1_dna.blow5"), "rb").read()
    cut.write_bytes(raw[:-2000])
    f = L.slow5_open(str(cut).encode(), b"r")
    assert f
    n = C.c_size_t()
    k = 0
    while True:
        p = L.slow5_get_next_mem(C.byref(n), f)
        if not p:
            break
        libc.free(p)
        k += 1
    assert k < 5
    L.slow5_close(f)


def test_header_writer_reproduces_golden_headers(L, tmp_path):
    """header of the v0.1.0 uncompressed file written for zlib+svb-zd == header bytes of the reference's
    zlib+svb golden (version raised to 0.2.0, press codes 1/1); same-method rewrite == original"""
    src = L.slow5_open(golden("exp_1_lossless.blow5").encode(), b"r")
    out = tmp_path / "h.bin"
    fp = libc.fopen(str(out).encode(), b"wb")
    nb = L.slow5_hdr_fwrite(fp, src.contents.header, 2, PressMethod(1, 2))
    assert L.slow5_eof_fwrite(fp) == 5
    libc.fclose(fp)
    want = Blow5(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"))
    wrote = out.read_bytes()
    assert nb == len(wrote) - 5 and wrote[-5:] == b"5WOLB"
    assert wrote[:-5] == want.raw[: 68 + len(want.header_text)]
    fp = libc.fopen(str(out).encode(), b"wb")
    L.slow5_hdr_fwrite(fp, src.contents.header, 2, PressMethod(0, 0))
    libc.fclose(fp)
    orig = Blow5(golden("exp_1_lossless.blow5"))
    assert out.read_bytes() == orig.raw[: 68 + len(orig.header_text)]
    L.slow5_close(src)


def test_index_load_and_get_mem(L, tmp_path):
    shutil.copy(golden("example_multi_rg_v0.2.0.blow5"), tmp_path / "f.blow5")
    shutil.copy(golden("example_multi_rg_v0.2.0.blow5.idx.exp"), tmp_path / "f.blow5.idx")
    ref = Blow5(golden("example_multi_rg_v0.2.0.blow5"))
    f = L.slow5_open(str(tmp_path / "f.blow5").encode(), b"r")
    assert L.slow5_idx_load(f) == 0
    n = C.c_size_t()
    import zlib

    for rec in ref.records:
        rid = zlib.decompress(rec)[2:38]
        p = L.slow5_get_mem(rid, C.byref(n), f)
        assert p and C.string_at(p, n.value) == rec
        libc.free(p)
    assert not L.slow5_get_mem(b"no-such-read", C.byref(n), f)
    L.slow5_close(f)


def test_index_older_than_the_file_is_still_used_and_a_hostile_index_is_refused(L, tmp_path):
    """an index whose mtime lies before the BLOW5's (cp / rsync reorder mtimes) is warned about and used, as slow5lib does — never
    rebuilt in place; an entry whose offset + size wraps 64 bits must not pass the extent check"""
    import struct, zlib

    blow = tmp_path / "f.blow5"
    shutil.copy(golden("example_multi_rg_v0.2.0.blow5"), blow)
    shutil.copy(golden("example_multi_rg_v0.2.0.blow5.idx.exp"), str(blow) + ".idx")
    os.utime(str(blow) + ".idx", (1_000_000_000, 1_000_000_000))      # years older than the file
    before = open(str(blow) + ".idx", "rb").read()
    L.slow5_set_log_level(0)
    f = L.slow5_open(str(blow).encode(), b"r")
    assert L.slow5_idx_load(f) == 0
    ref = Blow5(str(blow))
    n = C.c_size_t()
    rid = zlib.decompress(ref.records[3])[2:38]
    p_ = L.slow5_get_mem(rid, C.byref(n), f)
    assert p_ and C.string_at(p_, n.value) == ref.records[3]
    libc.free(p_)
    L.slow5_close(f)
    assert open(str(blow) + ".idx", "rb").read() == before               # the user's index was not touched
    # hostile entry: offset near 2^64 so that offset + size wraps to a small number
    raw = bytearray(before)
    first = 64 + 2 + struct.unpack_from("<H", raw, 64)[0]
    struct.pack_into("<QQ", raw, first, 0xFFFFFFFFFFFFFF00, 0x200)
    open(str(blow) + ".idx", "wb").write(raw)
    f = L.slow5_open(str(blow).encode(), b"r")
    assert L.slow5_idx_load(f) != 0
    L.slow5_close(f)
    L.slow5_set_log_level(1)


# ---------------------------------------------------------------- end to end on the GPU
def _run(*args):
    r = subprocess.run([S5VIEW] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r


@pytest.mark.gpu
def test_view_zlib_svb_to_uncompressed_is_byte_identical_to_golden(tmp_path):
    out = tmp_path / "o.blow5"
    _run(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"), out, "none", "none")
    assert out.read_bytes() == open(golden("exp_1_lossless_v0.2.0.blow5"), "rb").read()


@pytest.mark.gpu
def test_index_cases_3_and_4_of_the_reference(tmp_path):
    src = tmp_path / "example_multi_rg_v0.2.0.blow5"
    shutil.copy(golden("example_multi_rg_v0.2.0.blow5"), src)
    _run("--index", src)                                                     # test/test_index.sh case 3
    assert open(str(src) + ".idx", "rb").read() == open(golden("example_multi_rg_v0.2.0.blow5.idx.exp"), "rb").read()
    nn = tmp_path / "example_multi_rg_v0.2.0_none_none.blow5"
    _run(src, nn, "none", "none")                                            # case 4: view -c none -s none, then index
    _run("--index", nn)
    assert open(str(nn) + ".idx", "rb").read() == open(golden("example_multi_rg_v0.2.0_none_none.blow5.idx.exp"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_kb", [0, 16, 200])
def test_index_builder_reads_record_heads_only_and_agrees_with_the_files(tmp_path, chunk_kb):
    """slow5_idx_create: the file in chunks, records framed in place, ids from the head of each record (k_inflate_head); the reference's
    .idx files must come out byte for byte whatever the chunk size cuts (a 16 KiB chunk is smaller than most records: the chunk grows),
    also for zstd files and 300-character ids (both take the general decode), and an index of 3000 own records finds every read"""
    import struct, zlib
    from slow5tools_amd import press

    env = dict(os.environ)
    if chunk_kb:
        env["SLOW5_IDX_CHUNK_KB"] = str(chunk_kb)

    def index(path):
        r = subprocess.run([S5VIEW, "--index", str(path)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        return open(str(path) + ".idx", "rb").read()

    for name in ("example_multi_rg_v0.2.0.blow5", "example_multi_rg_v0.2.0_none_none.blow5"):
        if not os.path.exists(golden(name)):
            continue
        src = tmp_path / name
        shutil.copy(golden(name), src)
        assert index(src) == open(golden(name + ".idx.exp"), "rb").read(), name
    z = tmp_path / "example_multi_rg_v0.2.0_zstd_svb-zd.blow5"
    shutil.copy(golden("example_multi_rg_v0.2.0_zstd_svb-zd.blow5"), z)
    zi, ref = index(z), open(golden("example_multi_rg_v0.2.0.blow5.idx.exp"), "rb").read()
    ids = lambda raw: [raw[o + 2:o + 2 + struct.unpack_from("<H", raw, o)[0]] for o in _idx_entries(raw)]
    assert ids(zi) == ids(ref)
    # own records: short and 300-character ids, empty reads, a long read
    rng = np.random.default_rng(4)
    n = 3000
    rid = [(b"r%05d" % i) if i % 7 else (b"long%05d_" % i) + b"x" * 290 for i in range(n)]
    sigs = [ob.synth_read(0x5105, i, int(k)) for i, k in enumerate(rng.integers(0, 6000, n))]
    sigs[11] = ob.synth_read(0x5105, 11, 150000)
    recs = press.encode_records(sigs, [press.pack_hdr(r, 0, 8192.0, 23.0, 1467.61, 4000.0) for r in rid])
    text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 1; head[10:14] = struct.pack("<I", 1); head[14] = 1
    own = tmp_path / "own.blow5"
    own.write_bytes(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")
    raw = index(own)
    offs = _idx_entries(raw)
    assert len(offs) == n
    at = 68 + len(text)
    for i, o in enumerate(offs):
        l = struct.unpack_from("<H", raw, o)[0]
        off, size = struct.unpack_from("<QQ", raw, o + 2 + l)
        assert raw[o + 2:o + 2 + l] == rid[i] and off == at and size == len(recs[i]), i
        at += len(recs[i])
    r = subprocess.run([S5VIEW, "--get", str(own), rid[7].decode()], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.split("\t")[0] == rid[7].decode() and int(r.stdout.split("\t")[2]) == len(sigs[7]), r.stderr


def _idx_entries(raw):
    """offsets of the entries of a .idx file"""
    import struct
    assert raw[:9] == b"SLOW5IDX\x01" and raw[-8:] == b"XDI5WOLS"
    o, out = 64, []
    while o < len(raw) - 8:
        out.append(o)
        o += 2 + struct.unpack_from("<H", raw, o)[0] + 16
    assert o == len(raw) - 8
    return out


@pytest.mark.gpu
def test_view_roundtrip_and_get(tmp_path):
    a, b, c = tmp_path / "a.blow5", tmp_path / "b.blow5", tmp_path / "c.blow5"
    _run(golden("merged_expected_zlib_svb.blow5"), a, "none", "none", 4)     # batches of 4: crosses a batch boundary
    _run(a, b, "zlib", "svb-zd", 3)
    _run(b, c, "none", "none")
    assert a.read_bytes() == c.read_bytes()
    fb = Blow5(str(b))
    ref = Blow5(golden("merged_expected_zlib_svb.blow5"))
    assert fb.version == (0, 2, 0) and fb.rec_method == 1 and fb.sig_method == 1 and fb.header_text == ref.header_text
    import zlib

    assert [zlib.decompress(r) for r in fb.records] == [zlib.decompress(r) for r in ref.records]
    assert sum(map(len, fb.records)) <= 1.02 * sum(map(len, ref.records))
    rid = zlib.decompress(ref.records[2])[2:38].decode()
    r = _run("--get", b, rid)
    d = ob.rec_parse(zlib.decompress(ref.records[2]), 1)
    f = r.stdout.strip().split("\t")
    assert f[0] == rid and int(f[2]) == len(d["signal"]) and f[3] == ",".join(str(int(x)) for x in d["signal"][:8])


@pytest.mark.gpu
@pytest.mark.parametrize("K,workers", [(1, 3), (2, 1), (3, 2), (4096, 3)])
def test_pipelined_view_writes_the_same_bytes_as_the_serial_phases(tmp_path, K, workers):
    """read || GPU || write must not change a byte or the record order (src/view.c:296-299)"""
    src = golden("merged_expected_zlib_svb.blow5")
    serial, piped = tmp_path / "s.blow5", tmp_path / "p.blow5"
    _run(src, serial, "zlib", "svb-zd", K, 0)
    _run(src, piped, "zlib", "svb-zd", K, workers)
    assert serial.read_bytes() == piped.read_bytes()
    back = tmp_path / "b.slow5"
    _run(piped, back, "none", "none", K, workers)                # and through the ASCII printer
    serial_txt = tmp_path / "st.slow5"
    _run(serial, serial_txt, "none", "none", 4096, 0)
    assert back.read_bytes() == serial_txt.read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_kb,readers", [(300, 1), (517, 3), (4096, 4)])
def test_chunked_view_pipeline_equals_the_per_record_pipeline(tmp_path, chunk_kb, readers):
    """BLOW5 -> BLOW5 through s5gpu_recompress_stream (file chunks into pinned memory, records framed in place, one contiguous
    stream back, one write per chunk): byte-identical to the per-record pipeline, whatever the chunk size cuts in two"""
    import struct

    from slow5tools_amd import press

    rng = np.random.default_rng(21)
    n = 700
    sigs = [(480 + 35 * rng.standard_normal(int(k))).astype(np.int16) for k in rng.integers(50, 9000, n)]
    sigs[5] = (480 + 35 * rng.standard_normal(120000)).astype(np.int16)        # one record near the smallest chunk's size
    hdrs = [press.pack_hdr(b"r%06d" % i, i % 3, 8192.0, 23.0, 1467.61, 4000.0) for i in range(n)]
    recs = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_NONE)
    text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 0; head[10:14] = struct.pack("<I", 3); head[14] = 0
    src = tmp_path / "in.blow5"
    src.write_bytes(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")
    env = dict(os.environ, S5VIEW_CHUNK_KB=str(chunk_kb), S5VIEW_READERS=str(readers))
    a, b = tmp_path / "chunked.blow5", tmp_path / "per_record.blow5"
    r = subprocess.run([S5VIEW, str(src), str(a), "zlib", "svb-zd", "4096", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "chunked pipeline" in r.stderr, r.stderr
    r = subprocess.run([S5VIEW, str(src), str(b), "zlib", "svb-zd", "64", "2"], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5VIEW_PER_RECORD="1"))
    assert r.returncode == 0 and "chunked pipeline" not in r.stderr, r.stderr
    assert a.read_bytes() == b.read_bytes()
    if chunk_kb == 517:                                               # the chunks over three (aliased) devices: same bytes
        c = tmp_path / "three.blow5"
        env3 = dict(env, S5GPU_ALIAS_DEVICES="1", S5VIEW_DEV_MASK="7", S5GPU_MULTI_MIN="8")
        r = subprocess.run([S5VIEW, str(src), str(c), "zlib", "svb-zd", "4096", "2"], capture_output=True, text=True, timeout=300, env=env3)
        assert r.returncode == 0 and "chunked pipeline" in r.stderr, r.stderr
        assert c.read_bytes() == a.read_bytes()
    if chunk_kb == 517:                                               # round 4: the pipeline's knobs change no byte — slot count, parallel chunk writer (pwrite / mmap), full exit
        for k, extra in enumerate(({"S5VIEW_SLOTS": "2"}, {"S5VIEW_SLOTS": "16", "S5VIEW_WRITERS": "4", "S5VIEW_CHUNK_KB": "8192"},      # (a chunk's output must exceed 1 MiB for several writers)
                                   {"S5VIEW_WRITERS": "3", "S5VIEW_WRITE_MODE": "mmap", "S5VIEW_CHUNK_KB": "8192"},
                                   {"S5_FULL_EXIT": "1", "S5VIEW_TIMING": "1"})):
            d = tmp_path / ("knob%d.blow5" % k)
            r = subprocess.run([S5VIEW, str(src), str(d), "zlib", "svb-zd", "4096", "3"], capture_output=True, text=True, timeout=300, env=dict(env, **extra))
            assert r.returncode == 0 and "chunked pipeline" in r.stderr, r.stderr
            assert d.read_bytes() == a.read_bytes(), extra
            if "S5VIEW_TIMING" in extra:
                assert "s5view[t]" in r.stderr and "stages (seconds, summed)" in r.stderr
    back = tmp_path / "back.blow5"                                   # and back: zlib + svb-zd -> none, through the chunks again
    r = subprocess.run([S5VIEW, str(a), str(back), "none", "none", "4096", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == src.read_bytes()


@pytest.mark.gpu
def test_chunked_view_takes_tiny_records_and_falls_back_on_a_record_larger_than_a_chunk(tmp_path):
    """(a) a chunk of records smaller than the slot's descriptor pitch (empty reads: ~54 bytes framed) used to end in
    'bad record framing' on the last chunk; (b) a record larger than the chunk aborted the tool: it now redoes the file record by record"""
    import struct
    from slow5tools_amd import press

    text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 0; head[10:14] = struct.pack("<I", 1); head[14] = 0

    def blow5(path, sigs, ids):
        hdrs = [press.pack_hdr(i, 0, 8192.0, 23.0, 1467.61, 4000.0) for i in ids]
        recs = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_NONE)
        path.write_bytes(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")

    def both(src, chunk_kb, expect_fallback, slot_recs=0):
        a, b = tmp_path / "chunked.blow5", tmp_path / "per_record.blow5"
        env = dict(os.environ, S5VIEW_CHUNK_KB=str(chunk_kb), S5VIEW_SLOT_RECS=str(slot_recs))
        r = subprocess.run([S5VIEW, str(src), str(a), "zlib", "svb-zd", "4096", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert ("per-record pipeline" in r.stderr) == expect_fallback, r.stderr
        r = subprocess.run([S5VIEW, str(src), str(b), "zlib", "svb-zd", "64", "2"], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5VIEW_PER_RECORD="1"))
        assert r.returncode == 0, r.stderr
        assert a.read_bytes() == b.read_bytes()

    tiny = tmp_path / "tiny.blow5"                                     # 30 000 empty reads with one-character ids: 55 bytes framed
    blow5(tiny, [np.zeros(0, np.int16)] * 30000, [b"%c" % (65 + i % 26) for i in range(30000)])
    both(tiny, 300, False)
    both(tiny, 64, False, slot_recs=1000)                              # 64 KiB hold 1191 of them: every chunk is framed in two rounds, the last one too
    rng = np.random.default_rng(3)
    big = tmp_path / "big.blow5"
    sigs = [(480 + 35 * rng.standard_normal(int(k))).astype(np.int16) for k in (300, 5000, 90000, 20, 7000)]
    blow5(big, sigs, [b"big%d" % i for i in range(5)])
    both(big, 64, True)                                                # the 180 KB record does not fit a 64 KiB chunk


S5GET = os.path.join(ROOT, "slow5tools_amd", "s5get")
S5MERGE = os.path.join(ROOT, "slow5tools_amd", "s5merge")


@pytest.mark.gpu
def test_merge_harness_reproduces_the_reference_merged_file(tmp_path):
    """test/test_merge.sh case 1.6: merge rg0..rg3.slow5 -c zlib -s svb-zd against merged_expected_zlib_svb.blow5 (tests/golden/merge_rg*.slow5 are
    the reference's raw/merge inputs).  The four files have different aux fields (rg2 carries an enum the others lack, in another column
    order): the output header — read groups appended per run_id, attributes as a sorted union with '.', aux fields enums first then sorted
    (src/merge.c:217-350) — must equal the reference's byte for byte, and every record must inflate to the reference's payload (0xFF for the
    missing enum).  Then the same merge from BLOW5 inputs, and lossy."""
    import zlib

    ins = [golden("merge_rg%d.slow5" % i) for i in range(4)]
    ref = Blow5(golden("merged_expected_zlib_svb.blow5"))
    out = tmp_path / "m.blow5"
    r = subprocess.run([S5MERGE, str(out), "-c", "zlib", "-s", "svb-zd"] + ins, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = Blow5(str(out))
    assert (got.version, got.rec_method, got.sig_method, got.num_read_groups) == (ref.version, 1, 1, 4)
    assert got.header_text == ref.header_text
    assert [zlib.decompress(x) for x in got.records] == [zlib.decompress(x) for x in ref.records]
    assert sum(map(len, got.records)) <= 1.02 * sum(map(len, ref.records))
    # BLOW5 inputs (each file converted first): one-read-group files with the output's columns would go straight through the device;
    # these need the detour (other columns), a second merge of the merged file does not
    bl = []
    for i, f in enumerate(ins):
        b = tmp_path / ("rg%d.blow5" % i)
        _run(f, b, "zlib", "svb-zd")
        bl.append(str(b))
    out2 = tmp_path / "m2.blow5"
    r = subprocess.run([S5MERGE, str(out2), "-K", "2"] + bl, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got2 = Blow5(str(out2))
    assert got2.header_text == ref.header_text and [zlib.decompress(x) for x in got2.records] == [zlib.decompress(x) for x in ref.records]
    out3 = tmp_path / "m3.blow5"                                        # the merged file merged again with one of its parts: same run_ids map back
    r = subprocess.run([S5MERGE, str(out3), str(out), bl[2]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got3 = Blow5(str(out3))
    pay = [zlib.decompress(x) for x in ref.records]
    assert got3.header_text == ref.header_text and [zlib.decompress(x) for x in got3.records] == pay + [pay[3], pay[4]]
    # lossy (-l): no aux columns in the header, no aux bytes in the records
    out4 = tmp_path / "m4.blow5"
    r = subprocess.run([S5MERGE, str(out4), "-l"] + ins, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got4 = Blow5(str(out4))
    assert got4.header_text.endswith(b"len_raw_signal\traw_signal\n")
    for x, y in zip(got4.records, ref.records):
        d, e = ob.rec_parse(zlib.decompress(x), 1), ob.rec_parse(zlib.decompress(y), 1)
        assert d["aux"] == b"" and d["read_id"] == e["read_id"] and d["read_group"] == e["read_group"] and np.array_equal(d["signal"], e["signal"])


@pytest.mark.gpu
def test_get_harness_fetches_decodes_and_rewrites_the_reads_of_an_id_list(tmp_path):
    """the loop of src/get.c:321-386 (test/test_get.sh): index, id list, batches of K — preads, ONE call per batch, ordered output.  500 ids drawn
    with replacement from a 3000-read file: the output holds their records in list order, byte-identical payloads; --benchmark (get.c:52)
    decodes the same reads and reports the sample count and checksum Python computes; an unknown id fails, or is skipped with S5GET_SKIP=1"""
    import struct, zlib
    from slow5tools_amd import press

    rng = np.random.default_rng(8)
    n = 3000
    rid = [b"read-%06d" % i for i in range(n)]
    sigs = [ob.synth_read(0x5105, i, int(k)) for i, k in enumerate(rng.integers(0, 9000, n))]
    sigs[5] = ob.synth_read(0x5105, 5, 200000)
    recs = press.encode_records(sigs, [press.pack_hdr(r, 0, 8192.0, 23.0, 1467.61, 4000.0) for r in rid])
    text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 1; head[10:14] = struct.pack("<I", 1); head[14] = 1
    src = tmp_path / "in.blow5"
    src.write_bytes(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")
    ids = tmp_path / "ids.txt"
    r = subprocess.run([S5GET, "--random", str(src), "500", "1", str(ids)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = [int(l.split("-")[1]) for l in ids.read_text().split()]
    assert len(want) == 500 and len(set(want)) > 400
    want[3] = 5                                                        # the 200 000-sample read: its record outgrows the first batch buffers
    ids.write_text("".join("read-%06d\n" % i for i in want))
    for K, readers in ((64, 3), (4096, 8)):
        out = tmp_path / ("o%d.blow5" % K)
        r = subprocess.run([S5GET, str(src), str(ids), str(out), "zlib", "svb-zd", str(K), str(readers)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = Blow5(str(out))
        assert got.header_text == text and len(got.records) == 500
        assert [zlib.decompress(x) for x in got.records] == [zlib.decompress(recs[i][8:]) for i in want]
    out = tmp_path / "raw.blow5"
    r = subprocess.run([S5GET, str(src), str(ids), str(out), "none", "none", "100"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert [x for x in Blow5(str(out)).records] == [ob.rec_pack(ob.make_rec(rid[i], 0, 8192.0, 23.0, 1467.61, 4000.0, sigs[i])[0], ob.SIG_NONE) for i in want]
    r = subprocess.run([S5GET, "--benchmark", str(src), str(ids), "128", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "p50" in r.stderr, r.stderr
    tot, samples, ck = r.stdout.split()
    exp = 0
    for i in want:
        s_ = sigs[i]
        if len(s_):
            exp = (exp * 1000003 + (int(s_[0]) & 0xFFFF) + ((int(s_[len(s_) // 2]) & 0xFFFF) << 16) + ((int(s_[-1]) & 0xFFFF) << 32)) & 0xFFFFFFFFFFFFFFFF
    assert int(tot) == 500 and int(samples) == sum(len(sigs[i]) for i in want) and int(ck, 16) == exp
    ids.write_text("read-000001\nno-such-read\nread-000002\n")
    r = subprocess.run([S5GET, str(src), str(ids), str(tmp_path / "x.blow5")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not in the index" in r.stderr
    r = subprocess.run([S5GET, str(src), str(ids), str(tmp_path / "y.blow5")], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5GET_SKIP="1"))
    assert r.returncode == 0 and len(Blow5(str(tmp_path / "y.blow5")).records) == 2


@pytest.mark.gpu
def test_concurrent_host_batches_from_threads():
    """two host threads inside the batch API at once (each owns one context): results equal the single-threaded ones"""
    import threading
    from slow5tools_amd import press
    rng = np.random.default_rng(9)
    jobs = []
    for t in range(4):
        sigs = [(500 + rng.integers(-30, 30, int(rng.integers(1, 9000)))).astype(np.int16) for _ in range(300)]
        hdrs = [press.pack_hdr(b"t%d_%d" % (t, i), t, 8192.0, 1.0, 1400.0, 4000.0) for i in range(300)]
        jobs.append((sigs, hdrs))
    want = [press.encode_records(s, h) for s, h in jobs]
    got = [None] * 4
    def run(k):
        for _ in range(3):
            got[k] = press.encode_records(*jobs[k])
    th = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == want
    for k in range(4):
        dec = press.decode_records([r[8:] for r in got[k]])
        assert all(d["status"] == 0 and np.array_equal(d["signal"], s) for d, s in zip(dec, jobs[k][0]))


@pytest.mark.gpu
def test_two_workers_first_batches_repeat(tmp_path):
    """Round 5: twenty fresh processes with two workers each write the same bytes.  (A small pinned buffer allocated by one worker
    during the other's first batch was, one process in a hundred, a buffer the stream's D2H copies never reached — `read 0 of 64: device
    produced an impossible record extent`; pinned allocations are 2 MiB or more since: slow5tools_amd/csrc/host_ctx.h, tools/view_flake.sh.)"""
    import struct

    from slow5tools_amd import press

    rng = np.random.default_rng(33)
    n = 300
    sigs = [(480 + 35 * rng.standard_normal(int(k))).astype(np.int16) for k in rng.integers(50, 9000, n)]
    hdrs = [press.pack_hdr(b"r%06d" % i, 0, 8192.0, 23.0, 1467.61, 4000.0) for i in range(n)]
    recs = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_NONE)
    text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 0; head[10:14] = struct.pack("<I", 1); head[14] = 0
    src = tmp_path / "in.blow5"
    src.write_bytes(bytes(head) + struct.pack("<I", len(text)) + text + b"".join(recs) + b"5WOLB")
    first = None
    for k in range(20):
        out = tmp_path / ("o%d.blow5" % k)
        r = subprocess.run([S5VIEW, str(src), str(out), "zlib", "svb-zd", "64", "2"], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, S5VIEW_PER_RECORD="1"))
        assert r.returncode == 0, "run %d: %s" % (k, r.stderr)
        b = out.read_bytes()
        if first is None:
            first = b
        assert b == first, "run %d wrote other bytes" % k
        out.unlink()
