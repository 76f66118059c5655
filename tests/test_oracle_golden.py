"""Pins the CPU oracle against the reference's own golden fixtures.

Mirrors the byte-identical diffs of the reference's shell tests:
  test/test_view.sh:90-165 (all codec combos, both directions), test/test_merge.sh:131-140,
  test/test_index.sh cases 3-4 (record offsets/sizes).
"""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import (Blow5, NONE_NONE_FIXTURES, ZLIB_NONE_FIXTURES, ZLIB_SVB_FIXTURES, golden,
                           read_slow5_ascii)


def _payload(f, rec):
    if f.rec_method == 1:
        return ob.zlib_decompress(rec, 8 * len(rec) + 4096)
    return rec


@pytest.mark.parametrize("name", ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES)
def test_zlib_records_recompress_byte_identical(name):
    f = Blow5(golden(name))
    assert f.rec_method == 1
    for rec in f.records:
        pl = _payload(f, rec)
        assert zlib.decompress(rec) == pl
        assert ob.zlib_compress(pl) == rec  # oracle zlib == reference bytes
        assert ob.adler32(pl) == struct.unpack(">I", rec[-4:])[0]


@pytest.mark.parametrize("name", ZLIB_SVB_FIXTURES)
def test_svbzd_blobs_reencode_bit_identical(name):
    f = Blow5(golden(name))
    assert f.sig_method == 1
    for rec in f.records:
        pl = _payload(f, rec)
        d = ob.rec_parse(pl, ob.SIG_SVB_ZD)
        idl = struct.unpack_from("<H", pl, 0)[0]
        L = struct.unpack_from("<Q", pl, 2 + idl + 4 + 32)[0]
        blob = pl[2 + idl + 4 + 32 + 8 :][:L]
        assert struct.unpack_from("<I", blob, 0)[0] == len(d["signal"])
        assert ob.svbzd_encode(d["signal"]) == blob
        assert np.array_equal(ob.svbzd_decode(blob), d["signal"])


@pytest.mark.parametrize("name", ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES + NONE_NONE_FIXTURES)
def test_rec_to_mem_reproduces_fixture_records(name):
    f = Blow5(golden(name))
    for rec in f.records:
        pl = _payload(f, rec)
        d = ob.rec_parse(pl, f.sig_method)
        r, keep = ob.make_rec(d["read_id"], d["read_group"], d["digitisation"], d["offset"], d["range"],
                              d["sampling_rate"], d["signal"], d["aux"])
        assert ob.rec_pack(r, f.sig_method) == pl
        mem = ob.rec_to_mem(r, f.rec_method, f.sig_method)
        assert mem == struct.pack("<Q", len(rec)) + rec


def test_decode_matches_ascii_twin():
    truth = read_slow5_ascii(golden("exp_1_lossless.slow5"))
    assert len(truth) == 1
    t = truth[0]
    for name in ["exp_1_lossless.blow5", "exp_1_lossless_zlib.blow5", "exp_1_lossless_zlib_svb_v0.2.0.blow5"]:
        f = Blow5(golden(name))
        d = ob.rec_parse(_payload(f, f.records[0]), f.sig_method)
        assert d["read_id"] == t["read_id"]
        assert d["read_group"] == t["read_group"]
        for k in ("digitisation", "offset", "range", "sampling_rate"):
            assert d[k] == t[k]
        assert np.array_equal(d["signal"], t["signal"])
        assert len(d["signal"]) == 59676


def test_first_key_byte_known_answer():
    # N = 59676 -> 4 + 14919 + 60333 = 75256 B; first key byte 0x05
    f = Blow5(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"))
    pl = _payload(f, f.records[0])
    d = ob.rec_parse(pl, 1)
    blob = ob.svbzd_encode(d["signal"])
    assert len(blob) == 75256
    assert blob[4] == 0x05
    assert list(d["signal"][:3]) == [1039, 588, 588]


def test_index_offsets_match_idx_fixture():
    f = Blow5(golden("example_multi_rg_v0.2.0.blow5"))
    idx = open(golden("example_multi_rg_v0.2.0.blow5.idx.exp"), "rb").read()
    assert idx[:9] == b"SLOW5IDX\x01"
    off = 64
    got = []
    while idx[off : off + 8] != b"XDI5WOLS":
        (l,) = struct.unpack_from("<H", idx, off)
        rid = idx[off + 2 : off + 2 + l]
        o, s = struct.unpack_from("<QQ", idx, off + 2 + l)
        got.append((rid, o, s))
        off += 2 + l + 16
    assert len(got) == len(f.records)
    for (rid, o, s), fo, rec in zip(got, f.offsets, f.records):
        assert o == fo and s == 8 + len(rec)
        assert _payload(f, rec)[2:38] == rid


EDGE_SIGNALS = {
    "empty": np.zeros(0, np.int16),
    "one": np.array([-5], np.int16),
    "zeros": np.zeros(1000, np.int16),
    "const": np.full(777, 1234, np.int16),
    "alt_extremes": np.tile(np.array([32767, -32768], np.int16), 500),
    "ramp": np.arange(-3000, 3000, dtype=np.int16),
}


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 4000, 4095, 4096, 4097, 65535, 65536])
def test_svbzd_roundtrip_lengths(n):
    rng = np.random.default_rng(n)
    x = rng.integers(-32768, 32768, n, dtype=np.int16)
    blob = ob.svbzd_encode(x)
    assert struct.unpack_from("<I", blob, 0)[0] == n
    assert np.array_equal(ob.svbzd_decode(blob), x)
    y = (rng.normal(500, 30, n)).astype(np.int16)
    assert np.array_equal(ob.svbzd_decode(ob.svbzd_encode(y)), y)


@pytest.mark.parametrize("name", sorted(EDGE_SIGNALS))
def test_svbzd_edge_signals(name):
    x = EDGE_SIGNALS[name]
    blob = ob.svbzd_encode(x)
    assert np.array_equal(ob.svbzd_decode(blob), x)
    if name == "alt_extremes":
        # |delta| = 65535 -> zigzag 131069/131070 -> 3-byte codes (code 2) except the first (65534: 2 bytes)
        keys = np.frombuffer(blob[4 : 4 + 250], dtype=np.uint8)
        assert keys[0] == 0b10101001 and (keys[1:] == 0b10101010).all()
    if name == "zeros":
        assert len(blob) == 4 + 250 + 1000 and set(blob[4:]) == {0}


def test_svbzd_decoder_accepts_4byte_codes_and_rejects_truncation():
    # code 3 never occurs on encode with int16 input but a decoder must accept it
    blob = struct.pack("<I", 1) + bytes([3]) + struct.pack("<I", 10)
    assert list(ob.svbzd_decode(blob)) == [5]
    good = ob.svbzd_encode(np.arange(100, dtype=np.int16))
    with pytest.raises(ValueError):
        ob.svbzd_decode(good[:-1])
    with pytest.raises(ValueError):
        ob.svbzd_decode(good + b"\x00")


def test_synth_generator_statistics():
    # svb ~1.265 B/sample, zlib-L6 ~0.87 B/sample at N=4000, P(2-byte code) ~1.4 %
    sig = ob.synth_reads(0x5105, 0, 64, 4000)
    svb = sum(len(ob.svbzd_encode(s)) for s in sig) / sig.size
    d = np.diff(np.concatenate([np.zeros((64, 1), np.int32), sig.astype(np.int32)], axis=1), axis=1)
    p2 = float((np.abs(d[:, 1:]) >= 128).mean())
    r, keep = ob.make_rec(ob.synth_read_id(0), 0, 8192.0, 23.0, 1467.61, 4000.0, sig[0])
    z = len(ob.rec_to_mem(r, 1, 1)) / 4000
    assert 1.25 <= svb <= 1.28, svb
    assert 0.008 <= p2 <= 0.02, p2
    assert 0.80 <= z <= 0.92, z
    assert sig.min() > 0 and sig.max() < 1400
    assert ob.synth_read_id(0x1234abcd) == b"1234abcd-0000-4000-8000-00001234abcd"


def test_batch_mt_matches_single_thread():
    sig = ob.synth_reads(0x5105, 100, 96, 1000)
    t1 = ob.encode_batch_mt(sig, 100, 1, batch_size=32)
    t4 = ob.encode_batch_mt(sig, 100, 4, batch_size=32)
    assert t1[0] == t4[0] and t1[2] == t4[2]
    r, keep = ob.make_rec(ob.synth_read_id(100), 0, 8192.0, 23.0, 1467.61, 4000.0, sig[0])
    total = sum(len(ob.rec_to_mem(ob.make_rec(ob.synth_read_id(100 + i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig[i])[0],
                                  1, 1)) for i in range(96))
    assert total == t1[0]


# ---- §8f row 2: SLOW5 ASCII <-> BLOW5, pinned on the reference's ASCII / binary fixture pairs ----
ASCII_PAIRS = [
    ("exp_1_lossless.slow5", "exp_1_lossless.blow5"),
    ("aux_array_exp_lossless.slow5", "aux_array_exp_lossless.blow5"),
    ("example_multi_rg_v0.1.0.slow5", "example_multi_rg_v0.1.0.blow5"),   # zlib records; doubles with > 6 decimals
]


def _ascii_file(path):
    raw = open(path, "rb").read()
    lines = raw.split(b"\n")
    assert lines[-1] == b""
    lines = [l + b"\n" for l in lines[:-1]]
    k = next(i for i, l in enumerate(lines) if l.startswith(b"#read_id"))
    return lines[:k + 1], lines[k + 1:]


def _payloads(b5):
    import zlib
    return [zlib.decompress(r) if b5.rec_method == 1 else r for r in b5.records]


@pytest.mark.parametrize("slow5,blow5", ASCII_PAIRS)
def test_ascii_header_text_is_the_blow5_header_text(slow5, blow5):
    hdr, _ = _ascii_file(golden(slow5))
    b5 = Blow5(golden(blow5))
    if "multi_rg" in slow5:
        # this older pair writes a read group's missing attribute as "." in the .slow5 and as "" in the .blow5; the
        # header attribute table is outside the path — everything else is the same text
        assert b"".join(hdr[2:]).replace(b"\t.", b"\t") == b5.header_text.replace(b"\t.", b"\t")
    else:
        assert b"".join(hdr[2:]) == b5.header_text
    assert hdr[1] == b"#num_read_groups\t%d\n" % b5.num_read_groups


@pytest.mark.parametrize("slow5,blow5", ASCII_PAIRS)
def test_ascii_line_to_payload_matches_golden_blow5(slow5, blow5):
    hdr, recs = _ascii_file(golden(slow5))
    types = ob.aux_types(hdr[-2])
    b5 = Blow5(golden(blow5))
    assert b5.sig_method == 0
    pays = _payloads(b5)
    assert len(pays) == len(recs) > 0
    exact = 0
    for line, pay in zip(recs, pays):
        mine = ob.line_to_payload(line, types)
        assert mine is not None
        # the text keeps 6 decimals of a double, so a payload made from text equals the golden one except in those doubles:
        # compare through the text form, and count byte-identical payloads
        assert ob.payload_to_line(mine, types) == line
        exact += mine == pay
    if "multi_rg" not in slow5:
        assert exact == len(recs)            # fixtures whose doubles have <= 6 decimals convert byte for byte


@pytest.mark.parametrize("slow5,blow5", ASCII_PAIRS)
def test_ascii_payload_to_line_matches_golden_slow5(slow5, blow5):
    hdr, recs = _ascii_file(golden(slow5))
    types = ob.aux_types(hdr[-2])
    pays = _payloads(Blow5(golden(blow5)))
    for line, pay in zip(recs, pays):
        assert ob.payload_to_line(pay, types) == line


def test_ascii_signal_text_round_trip_and_rejects():
    rng = np.random.default_rng(5)
    sig = rng.integers(-32768, 32768, 5000).astype(np.int16)
    sig[:4] = [-32768, 32767, 0, -1]
    txt = ob.signal_to_text(sig)
    assert txt == ",".join(str(int(v)) for v in sig).encode()
    assert np.array_equal(ob.text_to_signal(txt), sig)
    for bad in (b"1,,2", b",1", b"1,", b"32768", b"-32769", b"1 ,2", b"1,2x", b"--1", b"1-2"):
        assert ob.text_to_signal(bad) is None, bad


def test_ascii_missing_values_and_types():
    types = ob.aux_types(b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\tenum{a,b}\tchar*\tdouble\tint32_t\tuint8_t\tuint64_t\tint16_t*\tfloat\tchar\tenum{x,y}*")
    assert list(types) == [11, 0x8A, 9, 2, 4, 7, 0x81, 8, 10, 0x8B]
    line = b"r1\t3\t8192\t-4\t1467.61\t4000\t3\t5,-6,7\t.\t.\t.\t.\t.\t.\t.\t.\tq\t.\n"
    pay = ob.line_to_payload(line, types)
    assert pay is not None
    assert ob.payload_to_line(pay, types) == line
    full = b"r1\t3\t8192\t-4\t1467.61\t4000\t3\t5,-6,7\t1\tch12\t0.5\t-7\t2\t99\t1,-2,3\t1.25\tq\t0,1,1\n"
    pay = ob.line_to_payload(full, types)
    assert ob.payload_to_line(pay, types) == full
    assert ob.line_to_payload(b"r1\t3\t8192\t-4\t1467.61\t4000\t3\t5,-6\n", b"") is None          # count mismatch
    assert ob.line_to_payload(b"r1\t3\t8192\t-4\t1467.61\t4000\t2\t5,-6\textra\n", b"") is None   # undeclared column


# ---- §8f row 4: ex-zd signal codec, pinned on the reference's ex-zd fixtures ----
EX_ZD_FIXTURES = ["exp_1_lossless_zlib_ex_zd.blow5",                  # 59676 samples, 656 exceptions, q = 0
                  "PRPN119035_read1_b2.blow5",                         # `degrade -b 2`: q = 2
                  "na12878_prom_merged_r9.4.1_chr22_read1_b2.blow5",
                  "gridr10dna_b3.blow5"]                               # 8 reads, q = 3, no exceptions


def _exzd_blobs(name):
    b5 = Blow5(golden(name))
    assert b5.sig_method == 2 and b5.rec_method == 1
    out = []
    for r in b5.records:
        p = zlib.decompress(r)
        idl = struct.unpack_from("<H", p, 0)[0]
        at = 2 + idl + 4 + 32
        (L,) = struct.unpack_from("<Q", p, at)
        out.append((p, p[at + 8: at + 8 + L]))
    return out


@pytest.mark.parametrize("name", EX_ZD_FIXTURES)
def test_exzd_decode_then_encode_reproduces_every_fixture_blob(name):
    for payload, blob in _exzd_blobs(name):
        sig = ob.exzd_decode(blob)
        assert sig is not None and len(sig) == struct.unpack_from("<Q", blob, 1)[0]
        assert ob.exzd_encode(sig) == blob                     # bit for bit, header fields and both exception sections included
        rec = ob.rec_parse(payload, ob.SIG_EX_ZD)              # and through the record layer
        assert np.array_equal(rec["signal"], sig)
        r, keep = ob.make_rec(rec["read_id"], rec["read_group"], rec["digitisation"], rec["offset"], rec["range"], rec["sampling_rate"], sig, rec["aux"])
        assert ob.rec_pack(r, ob.SIG_EX_ZD) == payload


def test_exzd_lossless_fixture_equals_the_ascii_twin():
    ascii_sig = read_slow5_ascii(golden("exp_1_lossless.slow5"))[0]["signal"]
    (payload, blob), = _exzd_blobs("exp_1_lossless_zlib_ex_zd.blow5")
    assert np.array_equal(ob.exzd_decode(blob), ascii_sig)
    assert blob[9] == 0 and struct.unpack_from("<I", blob, 12)[0] == 656


def test_exzd_degraded_fixture_is_the_source_with_low_bits_rounded():
    """exp/degrade/*_b2: q = 2 in the blob, every decoded sample a multiple of 4"""
    for name in ("PRPN119035_read1_b2.blow5", "na12878_prom_merged_r9.4.1_chr22_read1_b2.blow5"):
        for payload, blob in _exzd_blobs(name):
            assert blob[9] == 2
            assert (ob.exzd_decode(blob) & 3 == 0).all()


def test_exzd_round_trip_edge_cases():
    rng = np.random.default_rng(8)
    cases = [np.zeros(0, np.int16), np.array([5], np.int16), np.array([-32768], np.int16), np.zeros(100, np.int16),
             np.array([32767, -32768] * 50, np.int16), (rng.integers(-32768, 32768, 5000)).astype(np.int16),
             (500 + rng.integers(-40, 40, 5000)).astype(np.int16), (8 * rng.integers(-4000, 4000, 3000)).astype(np.int16),
             np.full(1000, -32768, np.int16), np.arange(-300, 300, dtype=np.int16)]
    for x in cases:
        blob = ob.exzd_encode(x)
        y = ob.exzd_decode(blob)
        assert y is not None and np.array_equal(x, y), x[:8]
    for bad in (b"", b"\x01" + bytes(9), ob.exzd_encode(cases[5])[:-1], ob.exzd_encode(cases[5]) + b"\x00"):
        assert ob.exzd_decode(bad) is None


# ---- §8f row 4: zstd record press — libzstd is the oracle; oracle/zstd_dec.c restates the decoder and is pinned on it ----
ZSTD_FIXTURES = [("exp_1_lossless_zstd_v0.2.0.blow5", 0, "exp_1_lossless_v0.2.0.blow5"),
                 ("exp_1_lossless_zstd_svb_v0.2.0.blow5", 1, "exp_1_lossless_zlib_svb_v0.2.0.blow5"),
                 ("example_multi_rg_v0.2.0_zstd_svb-zd.blow5", 1, "example_multi_rg_v0.2.0.blow5")]
needs_zstd = pytest.mark.skipif(ob.zstd_ref() is None, reason="no libzstd.so.1 in this image")


def zstd_test_inputs(rng, sizes=(10, 100, 1000, 5000, 40000, 150000, 300000)):
    yield b""
    yield b"a"
    yield b"a" * 1000
    yield bytes(rng.integers(0, 256, 100000, dtype=np.uint8))
    for n in sizes:
        yield np.cumsum(rng.integers(-20, 21, n)).astype(np.int16).tobytes()
        yield bytes(rng.integers(0, 4, n, dtype=np.uint8))
        yield bytes(rng.choice(np.array([0, 1, 2, 3, 50, 200], dtype=np.uint8), n, p=[.5, .2, .1, .1, .05, .05]))
        t = (b"the quick brown fox jumps over the lazy dog " * (n // 40 + 1))[:n]
        yield t
        a = bytearray(t)
        for k in rng.integers(0, max(1, n), n // 20):
            a[k] = rng.integers(0, 256)
        yield bytes(a)


@needs_zstd
@pytest.mark.parametrize("name,sig,twin", ZSTD_FIXTURES)
def test_zstd_fixture_records_decode_to_the_twin_payloads(name, sig, twin):
    """the reference's zstd files hold the same payloads as their zlib / uncompressed twins; both decoders agree on them"""
    b5, tw = Blow5(golden(name)), Blow5(golden(twin))
    assert (b5.rec_method, b5.sig_method) == (2, sig) and len(b5.records) == len(tw.records)
    for r, t in zip(b5.records, tw.records):
        want = zlib.decompress(t) if tw.rec_method == 1 else t
        got = ob.zstd_decompress(r)
        assert got is not None
        assert ob.zstd_restated_decompress(r, len(got)) == got
        assert ob.rec_parse(got, sig)["signal"].tobytes() == ob.rec_parse(want, tw.sig_method)["signal"].tobytes()
        if tw.sig_method == sig:
            assert got == want


@needs_zstd
def test_zstd_level_1_reproduces_the_reference_frames():
    """slow5lib compresses records with ZSTD_compress at level 1: same bytes out of libzstd 1.4.8 for the plain-signal file"""
    b5 = Blow5(golden("exp_1_lossless_zstd_v0.2.0.blow5"))
    for r in b5.records:
        assert ob.zstd_compress(ob.zstd_decompress(r), 1) == r


@needs_zstd
def test_zstd_restated_decoder_matches_libzstd():
    rng = np.random.default_rng(11)
    n = 0
    for d in zstd_test_inputs(rng):
        for level in (1, 3, 5, 9, 15, 19, -5):
            f = ob.zstd_compress(d, level)
            assert ob.zstd_restated_decompress(f, len(d)) == d
            n += 1
    assert n > 200


@needs_zstd
def test_zstd_restated_decoder_rejects_or_agrees_on_damaged_frames():
    rng = np.random.default_rng(12)
    for d in list(zstd_test_inputs(rng, sizes=(1000, 40000)))[:12]:
        f = ob.zstd_compress(d, 1)
        for _ in range(60):
            g = bytearray(f)
            for k in rng.integers(0, len(g), 3):
                g[k] = rng.integers(0, 256)
            mine, ref = ob.zstd_restated_decompress(bytes(g), len(d)), ob.zstd_decompress(bytes(g), len(d))
            if ref is not None and mine is not None:
                assert mine == ref                       # no checksum in these frames: some damage still decodes
            # (libzstd's double-symbol Huffman decoder clamps an overrun on the last literal of a stream, so it accepts a few
            #  damaged frames that the restatement rejects; never the other way round on valid data)
        for cut in range(0, len(f), max(1, len(f) // 25)):
            assert ob.zstd_restated_decompress(f[:cut], len(d)) is None


def zstd_literal_inputs(rng):
    """byte blocks that steer the literals-only encoder through each of its paths"""
    yield from zstd_test_inputs(rng, sizes=(10, 100, 1000, 5000, 40000))
    for n in (64, 100, 1000, 5000, 16383, 16384, 16385, 20000, 70000):
        yield bytes(np.clip(rng.normal(128, 30, n), 0, 255).astype(np.uint8))                       # > 128 symbols: FSE-compressed weights
        yield bytes(np.clip(rng.normal(10, 3, n), 0, 255).astype(np.uint8)) + bytes(range(256))     # few heavy, many rare
        yield bytes(rng.integers(0, 256, n, dtype=np.uint8))                                        # incompressible: raw blocks
        yield bytes((rng.integers(0, 256, n) * (rng.random(n) < 0.02)).astype(np.uint8)) + bytes(range(256))
        yield bytes(rng.integers(0, 100, n, dtype=np.uint8))                                        # <= 128 symbols: direct weights
        yield bytes([7]) * n                                                                        # RLE blocks
        yield bytes(rng.choice(np.array([3, 250], dtype=np.uint8), n, p=[.97, .03]))               # two symbols


@needs_zstd
def test_zstd_literals_only_frames_are_valid():
    """oracle/zstd_enc.c (the layout the device encoder writes): libzstd and the restated decoder both take its frames"""
    rng = np.random.default_rng(13)
    n = 0
    for d in zstd_literal_inputs(rng):
        f = ob.zstd_literals_compress(d)
        assert ob.zstd_decompress(f) == d and ob.zstd_restated_decompress(f, len(d)) == d
        assert len(f) <= len(d) + 3 * (len(d) // 16384 + 1) + 9
        n += 1
    assert n > 80
    b5 = Blow5(golden("exp_1_lossless_zstd_svb_v0.2.0.blow5"))
    p = ob.zstd_decompress(b5.records[0])
    assert len(ob.zstd_literals_compress(p)) < 1.04 * len(b5.records[0])     # literals only: within 4 % of the reference's frame


def test_zstd_sequence_tables_of_the_device_are_the_predefined_ones():
    """csrc/zstd_seq_tables.h is generated (tools/gen_zstd_seq_tables.py); the generator, the oracle's run-time construction
    (oracle/zstd_enc.c, whose frames libzstd reads) and the committed header must agree"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_zstd_seq_tables as g
    assert open(os.path.join(root, "slow5tools_amd", "csrc", "zstd_seq_tables.h")).read() == g.text()
    L = ob.lib()
    for which, norm in ((0, g.LL), (1, g.ML)):
        nx, dnb, dfs = (C.c_uint16 * 64)(), (C.c_int32 * 53)(), (C.c_int32 * 53)()
        n = L.s5o_zstd_seq_ctable(which, nx, dnb, dfs)
        a, b, c = g.ctable(norm)
        assert n == len(norm) and list(nx) == a and list(dnb)[:n] == b and list(dfs)[:n] == c
    # ... and the decode cells the device decoder copies for a block in predefined mode: the oracle's fse_build over the same distributions,
    # and the first rows of RFC 8878 appendix A (state: symbol, bits, baseline)
    for which, norm, log in ((0, g.LL, 6), (1, g.OF, 5), (2, g.ML, 6)):
        cells = (C.c_uint32 * 64)()
        assert L.s5o_zstd_seq_dtable(which, cells) == 1 << log
        assert list(cells)[: 1 << log] == g.dtable(norm, log)
    def rows(v):
        return [(x & 255, (x >> 8) & 15, x >> 12) for x in v]
    assert rows(g.dtable(g.LL, 6))[:6] == [(0, 4, 0), (0, 4, 16), (1, 5, 32), (3, 5, 0), (4, 5, 0), (6, 5, 0)]
    assert rows(g.dtable(g.OF, 5))[:6] == [(0, 5, 0), (6, 4, 0), (9, 5, 0), (15, 5, 0), (21, 5, 0), (3, 5, 0)]
    assert rows(g.dtable(g.ML, 6))[:6] == [(0, 6, 0), (1, 4, 0), (2, 5, 32), (3, 5, 0), (5, 5, 0), (6, 5, 0)]


@needs_zstd
def test_zstd_twin_run_sequences():
    """the run sequences of oracle/zstd_enc.c: runs of every length around the threshold, > 127 sequences, long literal and
    match lengths; and the gain on a real record (the key bytes)"""
    rng = np.random.default_rng(14)
    for n in (64, 300, 5000, 16384, 50000):
        for R in (4, 5, 6, 9, 40, 300):
            pat = bytes([5]) * R + bytes(rng.integers(6, 256, 3, dtype=np.uint8))
            d = (pat * (n // len(pat) + 1))[:n]
            f = ob.zstd_literals_compress(d)
            assert ob.zstd_decompress(f) == d and ob.zstd_restated_decompress(f, len(d)) == d
        d = bytes(np.repeat(rng.integers(0, 256, n, dtype=np.uint8), rng.integers(1, 30, n))[:n])
        f = ob.zstd_literals_compress(d)
        assert ob.zstd_decompress(f) == d and ob.zstd_restated_decompress(f, len(d)) == d
    tot = ref = 0
    for i in range(10):
        sig = ob.synth_read(0x5105, i, 4000)
        rec, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 3.0, 1400.0, 4000.0, sig)
        p = ob.rec_pack(rec, ob.SIG_SVB_ZD)
        tot += len(ob.zstd_literals_compress(p)); ref += len(ob.zstd_compress(p, 1))
    assert tot < 1.003 * ref


def test_file_level_view_and_get_twins_reproduce_the_reference_files(tmp_path):
    """bench.py's e2e CPU baselines (oracle/batch.c: s5o_view_file, s5o_get_file — the whole loop of src/view.c:241-323 and of
    `get --benchmark`, src/get.c:52) are the oracle's worker behind a serial read and an ordered write: BLOW5 -> BLOW5 must give the
    reference's own zlib + svb-zd records back byte for byte, SLOW5 text -> BLOW5 the records of the reference's binary twin, and the get
    twin every sample of the file"""
    src = golden("exp_1_lossless_zlib_svb_v0.2.0.blow5")
    out = tmp_path / "o.blow5"
    n, ph = ob.view_file(src, out, 3)
    a, b = Blow5(src), Blow5(out)
    assert n == len(a.records) and a.records == b.records and a.header_text == b.header_text and a.raw[:64] == b.raw[:64]
    assert ph["first_read_to_last_write"] >= ph["compute"] > 0
    for name in ("exp_1_lossless", "aux_array_exp_lossless", "example_multi_rg_v0.1.0"):
        t = tmp_path / (name + ".blow5")
        n, _ = ob.view_file(golden(name + ".slow5"), t, 2, batch_size=3)
        ref = Blow5(golden(name + ".blow5"))                       # the reference's binary twin of the same reads
        got = Blow5(t)
        assert n == len(ref.records) == len(got.records) and (got.rec_method, got.sig_method) == (1, 1)
        assert got.version == ref.version and got.num_read_groups == ref.num_read_groups
        if got.num_read_groups == 1:           # (several read groups: a missing attribute is "." in the text and empty in the binary header — the
            assert got.header_text == ref.header_text      #  timing twin copies the lines; the product's header writer is tested in test_ascii.py)
        for g, r in zip(got.records, ref.records):
            pay = zlib.decompress(g)
            rec = ob.rec_parse(pay, ob.SIG_SVB_ZD)
            want = ob.rec_parse(zlib.decompress(r) if ref.rec_method == 1 else r, ref.sig_method)
            # (aux floats lose digits in the text, "%f": the text files are not the binary twins' equals there)
            assert rec["read_id"] == want["read_id"] and np.array_equal(rec["signal"], want["signal"]) and len(rec["aux"]) == len(want["aux"])
    pos = np.array(b.offsets, dtype=np.uint64)
    ln = np.array([len(r) + 8 for r in b.records], dtype=np.uint32)
    samples, secs = ob.get_file(out, np.tile(pos, 5), np.tile(ln, 5), 2, batch_size=2)
    want = sum(ob.rec_parse(zlib.decompress(r), ob.SIG_SVB_ZD)["signal"].size for r in b.records)
    assert samples == 5 * want and secs > 0
