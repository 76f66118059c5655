"""SLOW5 ASCII <-> BLOW5: the GPU parse / format of the raw_signal column and the host conversion of the
other columns, against the oracle (oracle/ascii.c, itself pinned on the reference's fixture pairs in test_oracle_golden.py) and
directly against the reference's ASCII / binary twins — the same comparisons test/test_view.sh makes with `diff`."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, golden
from test_container import File, Hdr, L, PressMethod, libc, S5VIEW   # noqa: F401  (L is a fixture)

PAIRS = [
    ("exp_1_lossless.slow5", "exp_1_lossless.blow5"),
    ("aux_array_exp_lossless.slow5", "aux_array_exp_lossless.blow5"),
    ("example_multi_rg_v0.1.0.slow5", "example_multi_rg_v0.1.0.blow5"),
]


class AuxMeta(C.Structure):
    _fields_ = [("num", C.c_uint32), ("types", C.POINTER(C.c_uint8))]


def ascii_file(path):
    lines = open(path, "rb").read().split(b"\n")
    assert lines[-1] == b""
    lines = [l + b"\n" for l in lines[:-1]]
    k = next(i for i, l in enumerate(lines) if l.startswith(b"#read_id"))
    return lines[:k + 1], lines[k + 1:]


def unsvb(payload):
    """svb-zd payload -> the same record with the raw int16 signal (what `-s none` stores)"""
    d = ob.rec_parse(payload, 1)
    rec, keep = ob.make_rec(d["read_id"], d["read_group"], d["digitisation"], d["offset"], d["range"], d["sampling_rate"], d["signal"], d["aux"])
    return ob.rec_pack(rec, 0)


# ------------------------------------------------------------------ host logic, no GPU needed
TYPE_LINES = [
    b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*",
    b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tchar*\tdouble\tint32_t\tuint8_t\tuint64_t\n",
    b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tenum{unknown,partial,mux_change}\tint8_t\tint16_t\tint64_t\tuint16_t\tuint32_t\tfloat\tchar\tfloat*\tdouble*\tuint8_t*\tenum{a,b}*\tint64_t*",
]


@pytest.mark.parametrize("line", TYPE_LINES)
def test_aux_types_equal_the_oracle(line):
    from slow5tools_amd import ascii as s5a
    assert s5a.aux_types(line) == ob.aux_types(line)


def test_aux_types_reject_unknown():
    from slow5tools_amd import ascii as s5a
    with pytest.raises(RuntimeError):
        s5a.aux_types(b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tint128_t")
    with pytest.raises(RuntimeError):
        s5a.aux_types(b"#char*\tuint32_t\tdouble")


@pytest.mark.parametrize("slow5", [p[0] for p in PAIRS])
def test_open_ascii_and_line_framing(L, slow5):
    hdr, recs = ascii_file(golden(slow5))
    f = L.slow5_open(golden(slow5).encode(), b"r")
    assert f
    assert f.contents.format == 1                                # SLOW5_FORMAT_ASCII
    h = f.contents.header.contents
    assert (h.version.major, h.version.minor, h.version.patch) == tuple(int(x) for x in hdr[0].split(b"\t")[1].split(b"."))
    assert h.num_read_groups == int(hdr[1].split(b"\t")[1])
    assert C.string_at(h.data, h.data_len) == b"".join(hdr[2:])
    types = ob.aux_types(hdr[-2])
    if types:
        am = C.cast(h.aux_meta, C.POINTER(AuxMeta)).contents
        assert bytes(am.types[:am.num]) == types
    else:
        assert not h.aux_meta
    got = []
    n = C.c_size_t()
    while True:
        p = L.slow5_get_next_mem(C.byref(n), f)
        if not p:
            break
        got.append(C.string_at(p, n.value))
        libc.free(p)
    assert got == [r[:-1] for r in recs]                        # lines come back without the newline
    L.slow5_close(f)


def test_hdr_fwrite_ascii_reproduces_the_fixture_header(L, tmp_path):
    hdr, _ = ascii_file(golden("exp_1_lossless.slow5"))
    f = L.slow5_open(golden("exp_1_lossless.blow5").encode(), b"r")   # header text comes from the binary twin
    assert f
    out = tmp_path / "h.slow5"
    fp = libc.fopen(str(out).encode(), b"wb")
    assert L.slow5_hdr_fwrite(fp, f.contents.header, 1, PressMethod(0, 0)) == len(b"".join(hdr))
    libc.fclose(fp)
    L.slow5_close(f)
    assert out.read_bytes() == b"".join(hdr)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("slow5,blow5", PAIRS)
def test_ascii_to_blow5_on_fixture_pairs(slow5, blow5):
    from slow5tools_amd import ascii as s5a
    hdr, recs = ascii_file(golden(slow5))
    types = s5a.aux_types(hdr[-2])
    want = [ob.line_to_payload(l, types) for l in recs]
    got = s5a.ascii_to_blow5(recs, types, s5a.REC_NONE, s5a.SIG_NONE)
    assert [g[8:] for g in got] == want
    assert all(struct.unpack("<Q", g[:8])[0] == len(g) - 8 for g in got)
    if "multi_rg" not in slow5:                                  # <= 6 decimals: byte-identical to the reference's own .blow5
        assert [g[8:] for g in got] == Blow5(golden(blow5)).records
    # default press: stock zlib inflates it, the oracle's svb decoder recovers the same record
    got = s5a.ascii_to_blow5(recs, types)
    for g, w in zip(got, want):
        assert unsvb(zlib.decompress(g[8:])) == w


@pytest.mark.gpu
@pytest.mark.parametrize("slow5,blow5", PAIRS + [("exp_1_lossless.slow5", "exp_1_lossless_zlib_svb_v0.2.0.blow5"),
                                                 ("exp_1_lossless.slow5", "exp_1_lossless_zlib.blow5")])
def test_blow5_to_ascii_reproduces_the_reference_slow5(slow5, blow5):
    from slow5tools_amd import ascii as s5a
    hdr, recs = ascii_file(golden(slow5))
    b5 = Blow5(golden(blow5))
    got = s5a.blow5_to_ascii(b5.records, s5a.aux_types(hdr[-2]), b5.rec_method, b5.sig_method)
    assert got == recs


def _random_lines(rng, n, types, max_len=20000):
    """payloads with every aux kind, printed by the oracle"""
    lines, pays = [], []
    for i in range(n):
        ns = int(rng.integers(0, max_len)) if i % 50 else [0, 1, 2, 15, 16, 17, 4095, 4096, 4097][i // 50 % 9]
        style = i % 4
        if style == 0:
            sig = rng.integers(-32768, 32768, ns)
        elif style == 1:
            sig = 500 + rng.integers(-40, 40, ns)
        elif style == 2:
            sig = rng.integers(-9, 10, ns)
        else:
            sig = np.where(rng.random(ns) < 0.5, -32768, 32767)
        sig = sig.astype(np.int16)
        rid = ("read_%d_%s" % (i, "x" * int(rng.integers(0, 40)))).encode()
        head = struct.pack("<H", len(rid)) + rid + struct.pack("<I4d", int(rng.integers(0, 5)), 8192.0, float(rng.integers(-50, 50)),
                                                               round(float(rng.random() * 2000), int(rng.integers(0, 7))), 4000.0)
        aux = b""
        for t in types:
            kind, arr = t & 15, t & 0x80
            cnt = int(rng.integers(0, 6)) if arr else 1
            if arr:
                aux += struct.pack("<Q", cnt)
            for _ in range(cnt):
                if kind <= 3:
                    bits = 8 << kind
                    aux += int(rng.integers(-(1 << (bits - 1)), (1 << (bits - 1)) - 1, dtype=np.int64)).to_bytes(bits // 8, "little", signed=True)
                elif kind <= 7:
                    bits = 8 << (kind - 4)
                    aux += int(rng.integers(0, min((1 << bits) - 1, (1 << 63) - 1), dtype=np.int64)).to_bytes(bits // 8, "little")
                elif kind == 8:
                    aux += struct.pack("<f", float(rng.integers(-4000, 4000)) / 8.0)
                elif kind == 9:
                    aux += struct.pack("<d", float(rng.integers(-4000000, 4000000)) / 64.0)
                elif kind == 10:
                    aux += bytes([int(rng.integers(65, 91))])
                else:
                    aux += bytes([int(rng.integers(0, 3))])
        pay = head + struct.pack("<Q", ns) + sig.tobytes() + aux
        line = ob.payload_to_line(pay, types)
        assert line is not None
        lines.append(line)
        pays.append(pay)
    return lines, pays


@pytest.mark.gpu
def test_random_batch_both_directions_equal_the_oracle():
    from slow5tools_amd import ascii as s5a
    types = s5a.aux_types(TYPE_LINES[2])
    rng = np.random.default_rng(11)
    lines, pays = _random_lines(rng, 1200, types)
    got = s5a.ascii_to_blow5(lines, types, s5a.REC_NONE, s5a.SIG_NONE)
    assert [g[8:] for g in got] == pays
    back = s5a.blow5_to_ascii(pays, types, s5a.REC_NONE, s5a.SIG_NONE)
    assert back == lines
    # through the default press and back
    z = s5a.ascii_to_blow5(lines, types)
    assert s5a.blow5_to_ascii([r[8:] for r in z], types) == lines
    # merge-style options: read_group rewrite, lossy
    rg = np.arange(len(lines), dtype=np.uint32) % 7
    lossy = s5a.ascii_to_blow5(lines[:64], types, s5a.REC_NONE, s5a.SIG_NONE, new_read_group=rg[:64], drop_aux=True)
    for g, p, r in zip(lossy, pays, rg):
        rec = ob.rec_parse(p, 0)
        rec_g = ob.rec_parse(g[8:], 0)
        assert rec_g["read_group"] == r and rec_g["aux"] == b"" and np.array_equal(rec_g["signal"], rec["signal"])


@pytest.mark.gpu
def test_one_long_read_and_line_ending_variants():
    from slow5tools_amd import ascii as s5a
    rng = np.random.default_rng(3)
    sig = rng.integers(-32768, 32768, 1_500_000).astype(np.int16)
    head = struct.pack("<H", 2) + b"r0" + struct.pack("<I4d", 0, 8192.0, 3.0, 1400.5, 4000.0)
    pay = head + struct.pack("<Q", sig.size) + sig.tobytes()
    line = ob.payload_to_line(pay)
    for variant in (line, line[:-1], line[:-1] + b"\r\n"):
        got = s5a.ascii_to_blow5([variant], b"", s5a.REC_NONE, s5a.SIG_NONE)
        assert got[0][8:] == pay
    assert s5a.blow5_to_ascii([pay], b"", s5a.REC_NONE, s5a.SIG_NONE) == [line]


BAD_SIGNAL = [(b"1,2,x", 3, 1), (b"1,,3", 3, 3), (b"1,2,3", 4, 4), (b"1,2,3", 2, 4), (b"1,2,32768", 3, 2), (b"-32769,2,3", 3, 2),
              (b"1,2,", 3, 3), (b",1,2", 3, 3), (b"1,2 ,3", 3, 1), (b"1,-,3", 3, 3), (b"1,2-2,3", 3, 1), (b"123456,2,3", 3, 2)]


@pytest.mark.gpu
def test_malformed_lines_fail_loudly():
    from slow5tools_amd import ascii as s5a
    good = b"r\t0\t8192\t1\t1400\t4000\t3\t1,2,3\n"
    for txt, n, code in BAD_SIGNAL:
        assert ob.text_to_signal(txt, 8) is None or len(ob.text_to_signal(txt, 8)) != n     # the oracle rejects it too
        st = [0, 0]
        with pytest.raises(RuntimeError):
            s5a.ascii_to_blow5([good, b"r\t0\t8192\t1\t1400\t4000\t%d\t%s\n" % (n, txt)], b"", status=st)
        assert st == [0, code], (txt, st)
    for bad in (b"r\t0\t8192\t1\t1400\t4000\t3\n", b"r\tx\t8192\t1\t1400\t4000\t3\t1,2,3\n", b"\t0\t8192\t1\t1400\t4000\t3\t1,2,3\n",
                b"r\t0\t8192\t1\t1400\t4000\t3\t1,2,3\textra\n", b"r\t0\t81x92\t1\t1400\t4000\t3\t1,2,3\n"):
        st = [0]
        with pytest.raises(RuntimeError):
            s5a.ascii_to_blow5([bad], b"", status=st)
        assert st == [16]
    # a declared aux column that is missing, or of the wrong type
    types = s5a.aux_types(TYPE_LINES[1])
    with pytest.raises(RuntimeError):
        s5a.ascii_to_blow5([good], types)
    with pytest.raises(RuntimeError):
        s5a.ascii_to_blow5([good[:-1] + b"\tch\t1.5\tnotint\t1\t2\n"], types)
    # aux bytes that do not match the declared types on the way out
    with pytest.raises(RuntimeError):
        s5a.blow5_to_ascii([ob.line_to_payload(good)], types, s5a.REC_NONE, s5a.SIG_NONE)


def _run(*args):
    r = subprocess.run([S5VIEW] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r


@pytest.mark.gpu
def test_view_between_slow5_and_blow5_matches_the_reference_twins(tmp_path):
    """test/test_view.sh: SLOW5 -> BLOW5 and BLOW5 -> SLOW5 are diffed against the committed twins"""
    out = tmp_path / "a.blow5"
    _run(golden("exp_1_lossless.slow5"), out, "none", "none")
    assert out.read_bytes() == open(golden("exp_1_lossless.blow5"), "rb").read()
    txt = tmp_path / "a.slow5"
    _run(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"), txt)
    want = open(golden("exp_1_lossless.slow5"), "rb").read().replace(b"#slow5_version\t0.1.0", b"#slow5_version\t0.2.0")
    assert txt.read_bytes() == want
    z = tmp_path / "z.blow5"
    _run(golden("exp_1_lossless.slow5"), z)                      # defaults zlib + svb-zd: BASELINE configs[0]
    mine, ref = Blow5(str(z)), Blow5(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"))
    assert (mine.version, mine.rec_method, mine.sig_method, mine.header_text) == (ref.version, 1, 1, ref.header_text)
    assert [zlib.decompress(r) for r in mine.records] == [zlib.decompress(r) for r in ref.records]
    t2 = tmp_path / "b.slow5"
    _run(golden("aux_array_exp_lossless.slow5"), t2)             # ASCII -> ASCII
    assert t2.read_bytes() == open(golden("aux_array_exp_lossless.slow5"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_kb,env_extra", [(200, {}), (517, {}), (4096, {}), (300, {"S5GPU_ALIAS_DEVICES": "1", "S5VIEW_DEV_MASK": "7", "S5GPU_MULTI_MIN": "8"})])
def test_chunked_slow5_to_blow5_equals_the_per_record_pipeline(tmp_path, chunk_kb, env_extra):
    """SURVEY 8f row 3 for the conversion BASELINE configs[0] names (SLOW5 -> BLOW5, /root/reference/src/view.c:35-57 with a .slow5
    input): the file is read in chunks, lines are framed in place (a line the chunk's end cuts is carried), each chunk goes through
    ONE s5gpu_ascii_to_blow5_stream call (the raw_signal columns parsed where they lie) and comes back as one contiguous record
    stream.  Byte-identical to the per-record pipeline (slow5_get_next_mem + slow5_gpu_convert_batch) whatever the chunk size cuts in
    two, with every aux kind, CR LF line ends, a last line without its newline; also over three (aliased) devices."""
    from slow5tools_amd import ascii as s5a
    types = s5a.aux_types(TYPE_LINES[2])
    rng = np.random.default_rng(23)
    lines, pays = _random_lines(rng, 900, types)
    lines[17] = lines[17][:-1] + b"\r\n"
    lines[-1] = lines[-1][:-1]                                    # no newline at the end of the file
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t5\n@asic_id\ta\tb\tc\td\te\n" + TYPE_LINES[2] + b"\n" + \
          b"#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\t" + b"\t".join(b"a%d" % i for i in range(len(types))) + b"\n"
    src = tmp_path / "in.slow5"
    src.write_bytes(hdr + b"".join(lines))
    env = dict(os.environ, S5VIEW_CHUNK_KB=str(chunk_kb), **env_extra)
    a, b = tmp_path / "chunked.blow5", tmp_path / "per_record.blow5"
    r = subprocess.run([S5VIEW, str(src), str(a), "zlib", "svb-zd", "4096", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "chunked pipeline (SLOW5 text in)" in r.stderr, r.stderr
    r = subprocess.run([S5VIEW, str(src), str(b), "zlib", "svb-zd", "64", "2"], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5VIEW_PER_RECORD="1"))
    assert r.returncode == 0 and "chunked pipeline" not in r.stderr, r.stderr
    assert a.read_bytes() == b.read_bytes()
    got = Blow5(str(a))
    assert len(got.records) == len(pays)
    for rec, pay in zip(got.records, pays):                       # and the records are the oracle's payloads (svb-zd inside)
        assert unsvb(zlib.decompress(rec)) == pay
    if chunk_kb == 200 and not env_extra:                         # a line longer than the chunk: the file is redone record by record
        big = tmp_path / "big.slow5"
        sig = rng.integers(-3000, 3000, 40000).astype(np.int16)
        pay = struct.pack("<H", 2) + b"r0" + struct.pack("<I4d", 0, 8192.0, 3.0, 1400.5, 4000.0) + struct.pack("<Q", sig.size) + sig.tobytes()
        big.write_bytes(b"#slow5_version\t0.2.0\n#num_read_groups\t1\n" + TYPE_LINES[0] + b"\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
                        + lines_plain(3) + ob.payload_to_line(pay) + lines_plain(2))
        o1, o2 = tmp_path / "big1.blow5", tmp_path / "big2.blow5"
        r = subprocess.run([S5VIEW, str(big), str(o1), "none", "none", "4096", "1"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "per-record pipeline" in r.stderr, r.stderr
        r = subprocess.run([S5VIEW, str(big), str(o2), "none", "none", "4096", "1"], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5VIEW_PER_RECORD="1"))
        assert r.returncode == 0, r.stderr
        assert o1.read_bytes() == o2.read_bytes() and len(Blow5(str(o1)).records) == 6


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_kb", [150, 1024])
def test_chunked_blow5_to_slow5_equals_the_per_record_pipeline_and_the_source_text(tmp_path, chunk_kb):
    """the other direction through chunks (s5gpu_blow5_to_ascii_stream): a BLOW5 file made from 700 random lines with every aux kind is
    printed back chunk by chunk — the signal columns on the device, prefix | signal | suffix of every line put in place there — and must
    give the per-record pipeline's bytes and the source lines"""
    from slow5tools_amd import ascii as s5a
    types = s5a.aux_types(TYPE_LINES[2])
    rng = np.random.default_rng(29)
    lines, pays = _random_lines(rng, 700, types, max_len=9000)
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t5\n@asic_id\ta\tb\tc\td\te\n" + TYPE_LINES[2] + b"\n" + \
          b"#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\t" + b"\t".join(b"a%d" % i for i in range(len(types))) + b"\n"
    src = tmp_path / "in.slow5"
    src.write_bytes(hdr + b"".join(lines))
    z = tmp_path / "z.blow5"
    _run(src, z, "zlib", "svb-zd")
    env = dict(os.environ, S5VIEW_CHUNK_KB=str(chunk_kb))
    a, b = tmp_path / "chunked.slow5", tmp_path / "per_record.slow5"
    r = subprocess.run([S5VIEW, str(z), str(a), "none", "none", "4096", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "(SLOW5 text out)" in r.stderr, r.stderr
    r = subprocess.run([S5VIEW, str(z), str(b), "none", "none", "64", "2"], capture_output=True, text=True, timeout=300, env=dict(os.environ, S5VIEW_PER_RECORD="1"))
    assert r.returncode == 0 and "chunked pipeline" not in r.stderr, r.stderr
    assert a.read_bytes() == b.read_bytes() == hdr + b"".join(lines)


def lines_plain(k):
    out = b""
    for i in range(k):
        out += b"p%d\t0\t8192\t1\t1400\t4000\t4\t1,-2,3,%d\n" % (i, i)
    return out
