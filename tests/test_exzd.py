"""ex-zd signal codec on the GPU against the oracle (oracle/exzd.c, itself pinned bit for bit on the
reference's ex-zd fixtures in test_oracle_golden.py) and directly against those fixtures."""
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, golden
from test_container import S5VIEW, _run   # noqa: F401

pytestmark = pytest.mark.gpu

FIXTURES = ["exp_1_lossless_zlib_ex_zd.blow5", "PRPN119035_read1_b2.blow5", "na12878_prom_merged_r9.4.1_chr22_read1_b2.blow5",
            "gridr10dna_b3.blow5"]
HDR_ARGS = (0, 8192.0, 23.0, 1467.61, 4000.0)


@pytest.fixture(scope="module")
def press():
    from slow5tools_amd import _lib, press as p
    _lib.check(_lib.lib().s5gpu_init(0), "s5gpu_init")
    return p


def _payload(press, hdr, sig, aux=b""):
    idl = struct.unpack_from("<H", hdr, 0)[0]
    rg, dg, of, rn, sr = struct.unpack_from("<Idddd", hdr, 2 + idl)
    r, keep = ob.make_rec(hdr[2:2 + idl], rg, dg, of, rn, sr, sig, aux)
    return ob.rec_pack(r, ob.SIG_EX_ZD)


@pytest.mark.parametrize("name", FIXTURES)
def test_decode_reference_exzd_files(press, name):
    b5 = Blow5(golden(name))
    assert (b5.rec_method, b5.sig_method) == (1, 2)
    got = press.decode_records(b5.records, press.REC_ZLIB, press.SIG_EX_ZD)
    for g, r in zip(got, b5.records):
        want = ob.rec_parse(zlib.decompress(r), ob.SIG_EX_ZD)
        assert g["status"] == 0 and np.array_equal(g["signal"], want["signal"]) and g["read_id"] == want["read_id"]


@pytest.mark.parametrize("name", FIXTURES)
def test_reencode_reference_exzd_records_payload_identical(press, name):
    """decode a reference ex-zd record, encode it again with (zlib, ex-zd): stock zlib inflates it to the reference's payload"""
    b5 = Blow5(golden(name))
    dec = press.decode_records(b5.records, press.REC_ZLIB, press.SIG_EX_ZD)
    sigs = [d["signal"] for d in dec]
    hdrs, auxs = [], []
    for r in b5.records:
        p = zlib.decompress(r)
        w = ob.rec_parse(p, ob.SIG_EX_ZD)
        idl = len(w["read_id"])
        hdrs.append(p[:2 + idl + 4 + 32])
        auxs.append(w["aux"])
    out = press.encode_records(sigs, hdrs, auxs, press.REC_ZLIB, press.SIG_EX_ZD)
    for o, r in zip(out, b5.records):
        assert zlib.decompress(o[8:]) == zlib.decompress(r)          # the ex-zd blob inside is bit-identical to the reference's
    raw = press.encode_records(sigs, hdrs, auxs, press.REC_NONE, press.SIG_EX_ZD)
    for o, r in zip(raw, b5.records):
        assert o[8:] == zlib.decompress(r)


def _signals(rng):
    out = [np.zeros(0, np.int16), np.array([7], np.int16), np.array([-32768, 32767], np.int16), np.zeros(300, np.int16)]
    for n in (2, 3, 15, 16, 17, 18, 4095, 4096, 4097, 4098, 8193, 12289, 70000):
        out.append((500 + rng.integers(-40, 40, n)).astype(np.int16))            # few exceptions
    for n in (100, 4097, 9000, 30000):
        out.append(rng.integers(-32768, 32768, n).astype(np.int16))              # nearly every position an exception
    out.append((16 * rng.integers(-1500, 1500, 20000)).astype(np.int16))         # q = 4, as `degrade` leaves it
    out.append(np.where(np.arange(20000) % 977 == 0, 30000, 512).astype(np.int16))   # isolated exceptions, long gaps
    out.append(np.concatenate([np.full(5000, 100, np.int16), rng.integers(-32768, 32768, 5000).astype(np.int16), np.full(5000, -3, np.int16)]))
    s = (400 + rng.integers(-30, 30, 300000)).astype(np.int16)                  # exceptions > 4096 spread over many tiles
    s[::50] = 20000
    out.append(s)
    return out


def test_encode_matches_oracle_bit_for_bit_and_round_trips(press):
    rng = np.random.default_rng(21)
    sigs = _signals(rng)
    hdrs = [press.pack_hdr(ob.synth_read_id(i), *HDR_ARGS) for i in range(len(sigs))]
    auxs = [b"" if i % 3 else bytes([i % 251] * (i % 40)) for i in range(len(sigs))]
    raw = press.encode_records(sigs, hdrs, auxs, press.REC_NONE, press.SIG_EX_ZD)
    for i, (o, s) in enumerate(zip(raw, sigs)):
        assert struct.unpack_from("<Q", o, 0)[0] == len(o) - 8
        assert o[8:] == _payload(press, hdrs[i], s, auxs[i]), "read %d (%d samples)" % (i, len(s))
    z = press.encode_records(sigs, hdrs, auxs, press.REC_ZLIB, press.SIG_EX_ZD)
    for i, (o, s) in enumerate(zip(z, sigs)):
        assert zlib.decompress(o[8:]) == _payload(press, hdrs[i], s, auxs[i])
    for recs, rm in ((raw, press.REC_NONE), (z, press.REC_ZLIB)):
        got = press.decode_records([r[8:] for r in recs], rm, press.SIG_EX_ZD)
        for g, s, a in zip(got, sigs, auxs):
            assert g["status"] == 0 and np.array_equal(g["signal"], s) and g["aux"] == a


def test_decode_rejects_malformed_exzd_blobs(press):
    rng = np.random.default_rng(4)
    sig = rng.integers(-2000, 2000, 9000).astype(np.int16)
    hdr = press.pack_hdr(b"r", *HDR_ARGS)
    good = _payload(press, hdr, sig)
    at = len(hdr) + 8                                     # first byte of the blob
    bad = []
    for off, val in ((0, 1), (9, 16), (12, 0xFF), (13, 0xFF), (16, 0), (1, 0x7F)):   # version, q, nex, section length, N
        b = bytearray(good)
        b[at + off] = val
        bad.append(bytes(b))
    b = bytearray(good); b[at + 12:at + 16] = struct.pack("<I", 5); bad.append(bytes(b))          # fewer exceptions than there are
    b = bytearray(good[:-7]); struct.pack_into("<Q", b, len(hdr), len(b) - at); bad.append(bytes(b))   # truncated byte array
    got = press.decode_records(bad + [good], press.REC_NONE, press.SIG_EX_ZD, raise_on_error=False)
    assert [g["status"] for g in got[:-1]] == [7] * len(bad)
    assert got[-1]["status"] == 0 and np.array_equal(got[-1]["signal"], sig)
    # 400 random damages: the call returns, nothing reports success with different data
    variants = []
    for v in range(400):
        b = bytearray(good)
        p = int(rng.integers(at, len(b)))
        if v % 2:
            b[p] ^= 1 << int(rng.integers(0, 8))
        else:
            b[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        variants.append(bytes(b[:len(good)]))
    for g in press.decode_records(variants, press.REC_NONE, press.SIG_EX_ZD, raise_on_error=False):
        assert g["status"] in (0, 6, 7)
        if g["status"] == 0:
            assert len(g["signal"]) == len(sig)


def test_wave_decoder_behind_the_inflate_equals_the_workgroup_decoder(press):
    """zlib records with ex-zd signals are unpacked by the wave that inflated them (exzd_decode_wave: chunks of 512 exceptions, tiles of
    1024 positions) — every shape of _signals, damaged blobs included, must come out as from the two-kernel form (unpack_fused = 0:
    exzd_decode_wg) and as from the call without payload output (S5GPU_DEC_NO_PAYLOAD), status for status and sample for sample."""
    from slow5tools_amd import _lib
    rng = np.random.default_rng(33)
    sigs = _signals(rng)
    hdrs = [press.pack_hdr(ob.synth_read_id(i), *HDR_ARGS) for i in range(len(sigs))]
    auxs = [b"" if i % 3 else bytes([i % 251] * (i % 40)) for i in range(len(sigs))]
    pays = [_payload(press, h, s, a) for h, s, a in zip(hdrs, sigs, auxs)]
    # damaged blobs inside intact zlib streams: header fields, section lengths, truncation, 120 random hits
    sig = rng.integers(-2000, 2000, 9000).astype(np.int16)
    hdr = press.pack_hdr(b"r", *HDR_ARGS)
    good = _payload(press, hdr, sig)
    at = len(hdr) + 8
    for off, val in ((0, 1), (9, 16), (12, 0xFF), (13, 0xFF), (16, 0), (1, 0x7F)):
        b = bytearray(good); b[at + off] = val; pays.append(bytes(b))
    b = bytearray(good); b[at + 12:at + 16] = struct.pack("<I", 5); pays.append(bytes(b))
    b = bytearray(good[:-7]); struct.pack_into("<Q", b, len(hdr), len(b) - at); pays.append(bytes(b))
    for v in range(120):
        b = bytearray(good)
        p = int(rng.integers(at, len(b)))
        if v % 2:
            b[p] ^= 1 << int(rng.integers(0, 8))
        else:
            b[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        pays.append(bytes(b))
    recs = [zlib.compress(p, 6 if i % 2 else 1) for i, p in enumerate(pays)]
    L = _lib.lib()
    fused = press.decode_records(recs, press.REC_ZLIB, press.SIG_EX_ZD, raise_on_error=False)
    _lib.check(L.s5gpu_set_option(b"unpack_fused", 0))
    try:
        split = press.decode_records(recs, press.REC_ZLIB, press.SIG_EX_ZD, raise_on_error=False)
    finally:
        _lib.check(L.s5gpu_set_option(b"unpack_fused", 1))
    f_np, s_np = press.decode_signals_dev(recs, press.REC_ZLIB, max_pay_cap=max(len(p) for p in pays) + 64,
                                          sig_caps=[max(len(s) for s in sigs) + 16] * len(recs), sig_method=press.SIG_EX_ZD)
    assert [g["status"] for g in fused[:len(sigs)]] == [0] * len(sigs)
    for i, (a, b) in enumerate(zip(fused, split)):
        assert a["status"] == b["status"] == int(f_np["status"][i]), (i, a["status"], b["status"], int(f_np["status"][i]))
        if a["status"] == 0:
            assert np.array_equal(a["signal"], b["signal"]) and np.array_equal(a["signal"], s_np[i]), i
            assert a["aux"] == b["aux"] and a["read_id"] == b["read_id"]
            if i < len(sigs):
                assert np.array_equal(a["signal"], sigs[i]), i
    assert sum(1 for g in fused[len(sigs):] if g["status"] == 7) >= 8


def test_view_to_and_from_exzd_files(tmp_path):
    """s5view: the reference's zlib+svb-zd file -> zlib+ex-zd reproduces the reference's ex-zd payloads; and back"""
    out = tmp_path / "x.blow5"
    _run(golden("exp_1_lossless_zlib_svb_v0.2.0.blow5"), out, "zlib", "ex-zd")
    mine, ref = Blow5(str(out)), Blow5(golden("exp_1_lossless_zlib_ex_zd.blow5"))
    assert (mine.rec_method, mine.sig_method, mine.header_text) == (1, 2, ref.header_text)
    assert [zlib.decompress(r) for r in mine.records] == [zlib.decompress(r) for r in ref.records]
    back = tmp_path / "n.blow5"
    _run(golden("exp_1_lossless_zlib_ex_zd.blow5"), back, "none", "none")
    assert back.read_bytes()[68:] == open(golden("exp_1_lossless_v0.2.0.blow5"), "rb").read()[68:]
    txt = tmp_path / "d.slow5"
    _run(golden("gridr10dna_b3.blow5"), txt)                       # degraded ex-zd file straight to SLOW5 text
    lines = [l for l in txt.read_text().split("\n") if l and l[0] not in "#@"]
    assert len(lines) == 8 and all(int(v) % 8 == 0 for v in lines[0].split("\t")[7].split(",")[:200])
