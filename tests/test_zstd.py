"""zstd record press on the GPU (decode side: csrc/zstd_dev.h) against libzstd itself, against the
restated decoder (oracle/zstd_dec.c) and against the reference's zstd fixtures."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, golden
from test_oracle_golden import ZSTD_FIXTURES, zstd_test_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def press():
    from slow5tools_amd import _lib, press as p
    _lib.check(_lib.lib().s5gpu_init(0), "s5gpu_init")
    return p


@pytest.fixture(autouse=True, params=["weights-pass", "in-wave"])
def first_tree(request, press):
    """every test of this file twice: with the pass that decodes the first tree description of every frame a frame per lane
    (k_zstd_weights, by default from 256 frames on) forced on for any batch, and without it (the frame's wave walks the chain itself)"""
    from slow5tools_amd import _lib
    L = _lib.lib()
    _lib.check(L.s5gpu_set_option(b"zstd_pre_min", 1 if request.param == "weights-pass" else 0))
    yield request.param
    _lib.check(L.s5gpu_set_option(b"zstd_pre_min", 256))


def zstd_solo(frames):
    """s5gpu_solo_batch(stage 4): whole frames in, payloads (or None) and per-frame status out"""
    from slow5tools_amd import _lib
    L = _lib.lib()
    n = len(frames)
    bufs = [C.create_string_buffer(f, len(f)) if len(f) else C.create_string_buffer(1) for f in frames]
    inp = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(f) for f in frames])
    out, olen, st = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_int32 * n)()
    rc = L.s5gpu_solo_batch(4, n, inp, lens, out, olen, st)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    res = []
    for i in range(n):
        res.append(C.string_at(out[i], olen[i]) if out[i] else None)
        if out[i]:
            libc.free(out[i])
    return rc, res, list(st)


@pytest.mark.parametrize("name,sig,twin", ZSTD_FIXTURES)
def test_decode_reference_zstd_files(press, name, sig, twin):
    b5, tw = Blow5(golden(name)), Blow5(golden(twin))
    got = press.decode_records(b5.records, press.REC_ZSTD, sig)
    for g, r, t in zip(got, b5.records, tw.records):
        want = ob.rec_parse(zlib.decompress(t) if tw.rec_method == 1 else t, tw.sig_method)
        assert g["status"] == 0 and g["read_id"] == want["read_id"] and np.array_equal(g["signal"], want["signal"])
        if ob.zstd_ref() is not None:
            assert g["payload"] == ob.zstd_decompress(r)


def test_frames_from_every_level_match_libzstd(press):
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    rng = np.random.default_rng(21)
    data, frames = [], []
    for d in zstd_test_inputs(rng):
        for level in (1, 3, 5, 9, 15, 19, -5):
            data.append(d)
            frames.append(ob.zstd_compress(d, level))
    rc, res, st = zstd_solo(frames)
    assert rc == 0 and all(s == 0 for s in st)
    for d, r in zip(data, res):
        assert r == d


def test_damaged_frames_never_hang_and_agree_when_accepted(press):
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    rng = np.random.default_rng(22)
    frames, caps = [], []
    for d in list(zstd_test_inputs(rng, sizes=(1000, 40000)))[:14]:
        f = ob.zstd_compress(d, 1)
        for _ in range(40):
            g = bytearray(f)
            for k in rng.integers(0, len(g), 3):
                g[k] = rng.integers(0, 256)
            frames.append(bytes(g))
            caps.append(len(d))
        for cut in range(1, len(f), max(1, len(f) // 10)):
            frames.append(f[:cut])
            caps.append(len(d))
    rc, res, st = zstd_solo(frames)
    n_ok = 0
    for f, cap, r, s in zip(frames, caps, res, st):
        if s == 0:
            n_ok += 1
            assert r == ob.zstd_restated_decompress(f, len(r))   # the device decoder and its CPU twin decode the same bytes
    assert 0 < n_ok < len(frames)


def test_payload_slot_too_small_is_retried_with_the_frame_size(press):
    """a frame says its content size: the host sizes the slot from it, the kernel reports 5 when a caller's slot is smaller"""
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    import torch
    from slow5tools_amd import _lib
    L = _lib.lib()
    d = bytes(np.random.default_rng(3).integers(0, 7, 50000, dtype=np.uint8))
    f = ob.zstd_compress(d, 1)
    dev = torch.device("cuda:0")
    t_in = torch.zeros(len(f) + 64, dtype=torch.uint8, device=dev)
    t_in[:len(f)] = torch.frombuffer(bytearray(f), dtype=torch.uint8).to(dev)
    pay = torch.zeros(60000, dtype=torch.uint8, device=dev)
    for cap, want in ((1000, 5), (49999, 5), (50000, 0), (60000 - 16, 0)):
        desc = np.zeros(1, dtype=_lib.REC_DESC)
        desc["in_len"], desc["pay_cap"] = len(f), cap
        t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        fields = torch.zeros(_lib.REC_FIELDS.itemsize, dtype=torch.uint8, device=dev)
        a = _lib.DecodeArgs()
        a.n_recs, a.rec_method, a.sig_method = 1, press.REC_ZSTD, 0
        a.desc, a.in_, a.payload, a.fields = t_desc.data_ptr(), t_in.data_ptr(), pay.data_ptr(), fields.data_ptr()
        _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "s5gpu_inflate_dev")
        torch.cuda.synchronize()
        got = fields.cpu().numpy().view(_lib.REC_FIELDS)[0]
        assert got["status"] == want and got["payload_len"] == 50000
        if want == 0:
            assert bytes(pay[:50000].cpu().numpy()) == d


# ---- encode side (csrc/zstd_enc_dev.h): valid frames that libzstd decompresses to the identical payload ----
def zstd_solo_compress(datas):
    from slow5tools_amd import _lib
    L = _lib.lib()
    n = len(datas)
    bufs = [C.create_string_buffer(d, len(d)) if len(d) else C.create_string_buffer(1) for d in datas]
    inp = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * n)(*[len(d) for d in datas])
    out, olen, st = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_int32 * n)()
    _lib.check(L.s5gpu_solo_batch(5, n, inp, lens, out, olen, st), "s5gpu_solo_batch(5)")
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    res = []
    for i in range(n):
        res.append(C.string_at(out[i], olen[i]))
        libc.free(out[i])
    return res


def test_compressed_buffers_are_valid_frames(press):
    """every block type (raw / RLE / Huffman with direct and FSE-compressed weights), block boundaries, tiny inputs"""
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    from test_oracle_golden import zstd_literal_inputs
    datas = list(zstd_literal_inputs(np.random.default_rng(31)))
    frames = zstd_solo_compress(datas)
    worst = 0.0
    for d, f in zip(datas, frames):
        assert ob.zstd_decompress(f) == d
        assert len(f) <= len(d) + 3 * (len(d) // 16384 + 1) + 9
        twin = ob.zstd_literals_compress(d)
        if len(d) >= 1000:
            worst = max(worst, len(f) / len(twin))
    assert worst < 1.025                                    # same layout as the CPU statement; the device's tree-free code lengths cost up to ~1.6 % on
                                                            # 'few heavy, many rare' byte distributions (0.13 % on signal payloads)
    rc, back, st = zstd_solo(frames)                        # and the device decoder reads its own frames
    assert rc == 0 and back == datas


def _hdr(press, i):
    return press.pack_hdr(b"read_%06d" % i, i % 3, 8192.0, 23.0, 1467.61, 4000.0)


@pytest.mark.parametrize("sig_name", ["svb-zd", "none", "ex-zd"])
def test_records_encode_to_frames_libzstd_reads(press, sig_name):
    sm = {"svb-zd": press.SIG_SVB_ZD, "none": press.SIG_NONE, "ex-zd": press.SIG_EX_ZD}[sig_name]
    rng = np.random.default_rng(32)
    sigs = [(500 + np.cumsum(rng.integers(-15, 16, n)) % 300).astype(np.int16)
            for n in (0, 1, 5, 50, 100, 400, 1000, 4000, 4000, 9000, 20000, 70000, 150000)]   # fused, overflow list and multi-block staged reads
    sigs += [np.zeros(5000, np.int16), np.full(40000, 77, np.int16), rng.integers(-32768, 32768, 6000).astype(np.int16)]
    sigs += [ob.synth_read(0x5105, i, 4000) for i in range(40)]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    auxs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 40, len(sigs))]
    out = press.encode_records(sigs, hdrs, auxs, press.REC_ZSTD, sm)
    raw = press.encode_records(sigs, hdrs, auxs, press.REC_NONE, sm)
    for o, r in zip(out, raw):
        assert int.from_bytes(o[:8], "little") == len(o) - 8
        if ob.zstd_ref() is not None:
            assert ob.zstd_decompress(o[8:]) == r[8:]
        assert ob.zstd_restated_decompress(o[8:], len(r)) == r[8:]
    dec = press.decode_records([o[8:] for o in out], press.REC_ZSTD, sm)
    for d, s, a in zip(dec, sigs, auxs):
        assert d["status"] == 0 and np.array_equal(d["signal"], s) and d["aux"] == a


def test_all_staged_batch_and_tiny_lds_budget(press):
    """long reads only (every read staged) and a batch forced onto the overflow list"""
    rng = np.random.default_rng(33)
    sigs = [(600 + rng.integers(-50, 50, n)).astype(np.int16) for n in (120000, 90000, 100001)]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    out = press.encode_records(sigs, hdrs, None, press.REC_ZSTD, press.SIG_SVB_ZD)
    raw = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_SVB_ZD)
    for o, r in zip(out, raw):
        assert ob.zstd_restated_decompress(o[8:], len(r)) == r[8:]
    b = press.DeviceBatch([len(s) for s in sigs[:1]] + [4000] * 8, hdr_len=len(_hdr(press, 0)), rec_method=press.REC_ZSTD, lds_payload_cap=1024, with_stream_out=False)
    some = [sigs[0]] + [ob.synth_read(1, i, 4000) for i in range(8)]
    b.upload(some, [_hdr(press, i) for i in range(9)])
    b.encode()
    recs = b.records()
    want = press.encode_records(some, [_hdr(press, i) for i in range(9)], None, press.REC_NONE, press.SIG_SVB_ZD)
    for o, r in zip(recs, want):
        assert ob.zstd_restated_decompress(o[8:], len(r)) == r[8:]


def test_view_to_zstd_and_back_through_the_compat_api(press, tmp_path):
    """s5view zlib -> zstd -> zlib: the payloads survive (the container path behind `view -c zstd`)"""
    from test_container import _run
    src = golden("exp_1_lossless_zlib_svb_v0.2.0.blow5")
    mid, back = str(tmp_path / "z.blow5"), str(tmp_path / "back.blow5")
    _run(src, mid, "zstd", "svb-zd")
    b5 = Blow5(mid)
    assert b5.rec_method == 2 and b5.sig_method == 1
    want = [zlib.decompress(r) for r in Blow5(src).records]
    assert [ob.zstd_restated_decompress(r, 10 ** 6) for r in b5.records] == want
    _run(mid, back, "zlib", "svb-zd")
    assert [zlib.decompress(r) for r in Blow5(back).records] == want
    # and the reference's own zstd file converts to the bytes of its zlib twin's payloads
    out2 = str(tmp_path / "ref.blow5")
    _run(golden("exp_1_lossless_zstd_svb_v0.2.0.blow5"), out2, "none", "svb-zd")
    assert Blow5(out2).records == want


def test_frames_without_a_content_size_field(press):
    """streaming writers may leave the content size out of the frame header (window descriptor instead): the slot is then a
    guess, a frame that does not fit reports 5 and is retried with a larger one"""
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    rng = np.random.default_rng(41)
    datas, frames = [], []
    for n in (0, 10, 3000, 50000, 300000):
        d = bytes(rng.integers(0, 6, n, dtype=np.uint8))                  # compresses ~3x: beyond the first slot guess for the big ones
        f = ob.zstd_compress(d, 3)
        fhd = f[4]
        flag, single = fhd >> 6, (fhd >> 5) & 1
        nb = {0: single, 1: 2, 2: 4, 3: 8}[flag]
        body = f[5 + (0 if single else 1) + nb:]
        g = f[:4] + bytes([0x00, 0x70]) + body                             # no content size, window log 24: same blocks
        assert ob.zstd_decompress(g, len(d)) == d
        datas.append(d)
        frames.append(g)
    highly = bytes(1 << 20)                                               # 1 MiB of zeros in a ~50-byte frame: far beyond any guess
    f = ob.zstd_compress(highly, 1)
    nb = {0: (f[4] >> 5) & 1, 1: 2, 2: 4, 3: 8}[f[4] >> 6]
    frames.append(f[:4] + bytes([0x00, 0x70]) + f[5 + (0 if (f[4] >> 5) & 1 else 1) + nb:])
    datas.append(highly)
    rc, res, st = zstd_solo(frames)
    assert rc == 0 and all(s == 0 for s in st)
    assert res == datas


def test_random_buffers_round_trip_through_both_directions(press):
    """300 random buffers (random length 0..40000, random alphabet size and skew, runs, boundaries around 16 KiB): the device
    encoder's frames are read back by libzstd (when present), by the restated decoder and by the device decoder"""
    rng = np.random.default_rng(51)
    datas = []
    for _ in range(300):
        n = int(rng.choice([rng.integers(0, 300), rng.integers(300, 5000), rng.integers(5000, 40000), 16384 + rng.integers(-3, 4)]))
        k = int(rng.choice([1, 2, 3, 8, 40, 129, 200, 256]))
        alphabet = rng.choice(256, size=k, replace=False).astype(np.uint8)
        p = rng.dirichlet(np.full(k, rng.choice([0.05, 0.3, 1.0, 5.0])))
        d = alphabet[rng.choice(k, size=n, p=p)]
        if n > 50 and rng.random() < 0.3:                       # a run in the middle
            a = int(rng.integers(0, n - 20))
            d[a:a + int(rng.integers(5, min(3000, n - a)))] = alphabet[0]
        datas.append(d.tobytes())
    frames = zstd_solo_compress(datas)
    for d, f in zip(datas, frames):
        assert ob.zstd_restated_decompress(f, len(d)) == d
        if ob.zstd_ref() is not None:
            assert ob.zstd_decompress(f, len(d)) == d
    rc, back, st = zstd_solo(frames)
    assert rc == 0 and back == datas


def test_random_length_records_through_the_split_encoder(press):
    """200 reads of random length (0 .. 12000 samples: with and without a key | data split, one- and two-tile payloads) with
    random aux tails: every frame holds the uncompressed record"""
    rng = np.random.default_rng(52)
    ns = rng.integers(0, 12000, 200)
    sigs = [(rng.integers(200, 900) + np.cumsum(rng.integers(-25, 26, int(n))) % 400).astype(np.int16) for n in ns]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    auxs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 300, len(sigs))]
    out = press.encode_records(sigs, hdrs, auxs, press.REC_ZSTD, press.SIG_SVB_ZD)
    raw = press.encode_records(sigs, hdrs, auxs, press.REC_NONE, press.SIG_SVB_ZD)
    for o, r in zip(out, raw):
        assert ob.zstd_restated_decompress(o[8:], len(r)) == r[8:]
    dec = press.decode_records([o[8:] for o in out], press.REC_ZSTD, press.SIG_SVB_ZD)
    assert all(d["status"] == 0 and np.array_equal(d["signal"], s) and d["aux"] == a for d, s, a in zip(dec, sigs, auxs))


def _set_zseq(v):
    from slow5tools_amd import _lib
    _lib.check(_lib.lib().s5gpu_set_option(b"zstd_sequences", v), "zstd_sequences")


def test_runs_go_out_as_sequences_and_close_the_gap_to_libzstd(press):
    """Verdict r01 item 9: runs of >= 5 equal bytes are one literal + one match at the repeat offset (predefined FSE tables).
    Same reads with the option off (round 1's literals-only frames): both decode, the new records are smaller and within
    0.5 % of libzstd level 1 on the bench reads"""
    if ob.zstd_ref() is None:
        pytest.skip("no libzstd.so.1 in this image")
    sigs = [ob.synth_read(0x5105, i, 4000) for i in range(64)]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    raw = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_SVB_ZD)
    try:
        _set_zseq(0)
        old = press.encode_records(sigs, hdrs, None, press.REC_ZSTD, press.SIG_SVB_ZD)
    finally:
        _set_zseq(1)
    new = press.encode_records(sigs, hdrs, None, press.REC_ZSTD, press.SIG_SVB_ZD)
    for o, n, r in zip(old, new, raw):
        assert ob.zstd_decompress(o[8:]) == r[8:] and ob.zstd_decompress(n[8:]) == r[8:]
    so, sn = sum(map(len, old)), sum(map(len, new))
    ref = sum(len(ob.zstd_compress(r[8:], 1)) + 8 for r in raw)
    assert sn < 0.985 * so, (so, sn)
    assert sn < 1.005 * ref, (sn, ref)
    dec = press.decode_records([n[8:] for n in new], press.REC_ZSTD, press.SIG_SVB_ZD)
    assert all(d["status"] == 0 and np.array_equal(d["signal"], s) for d, s in zip(dec, sigs))


def test_run_patterns_that_corner_the_tokeniser(press):
    """runs that start / end on lane-chunk and block boundaries, runs of exactly 4 / 5 / 6 bytes, a run longer than a block,
    more than 127 sequences in a block (two-byte count), literal and match lengths with extra bits, blocks where sequences
    are refused (no room / no certain gain) — every frame must hold the input, for libzstd and for the device decoder"""
    rng = np.random.default_rng(77)
    datas = []
    for n in (64, 65, 255, 256, 1024, 4096, 16383, 16384, 16385, 40000):
        datas.append(bytes(n))                                                        # one run
        datas.append(bytes([1]) + bytes(n - 1))                                       # literal + run to the end
        datas.append(bytes(n - 1) + bytes([9]))                                       # run + literal
        for R in (4, 5, 6, 7, 64, 257, 300):
            pat = (bytes([3]) * R + bytes([200]))
            datas.append((pat * (n // len(pat) + 1))[:n])                             # runs of exactly R, one literal between them
        x = rng.integers(0, 256, n, dtype=np.uint8)
        x[n // 3: n // 3 + 40] = 7; x[n // 2:] = np.where(rng.random(n - n // 2) < 0.9, 0, x[n // 2:])
        datas.append(bytes(x))                                                        # random literals, then mostly zeros
        y = np.repeat(rng.integers(0, 256, n // 16 + 1, dtype=np.uint8), rng.integers(1, 40, n // 16 + 1))[:n]
        datas.append(bytes(y))                                                        # runs of random length 1..39
        datas.append(bytes(rng.integers(0, 3, n, dtype=np.uint8)))                    # short runs only, few symbols
    frames = zstd_solo_compress(datas)
    for d, f in zip(datas, frames):
        if ob.zstd_ref() is not None:
            assert ob.zstd_decompress(f, len(d)) == d
        assert ob.zstd_restated_decompress(f, len(d)) == d
        assert len(f) <= len(d) + 3 * (len(d) // 16384 + 1) + 9
    rc, back, st = zstd_solo(frames)
    assert rc == 0 and back == datas
    # the twin states the same tokeniser: sizes agree closely wherever the Huffman tables do
    for d, f in zip(datas, frames):
        if len(d) >= 4096:
            assert len(f) <= 1.03 * len(ob.zstd_literals_compress(d)) + 16


def _xxh64_py(b):
    """XXH64, seed 0 (the published algorithm) in plain Python: the test's own statement of the checksum"""
    M = (1 << 64) - 1
    P1, P2, P3, P4, P5 = 0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x85EBCA77C2B2AE63, 0x27D4EB2F165667C5
    rotl = lambda x, r: ((x << r) | (x >> (64 - r))) & M
    rnd = lambda acc, v: (rotl((acc + v * P2) & M, 31) * P1) & M
    n, i = len(b), 0
    if n >= 32:
        v = [(P1 + P2) & M, P2, 0, (-P1) & M]
        while i + 32 <= n:
            for k in range(4):
                v[k] = rnd(v[k], int.from_bytes(b[i + 8 * k:i + 8 * k + 8], "little"))
            i += 32
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
        for k in range(4):
            h = ((h ^ rnd(0, v[k])) * P1 + P4) & M
    else:
        h = P5
    h = (h + n) & M
    while i + 8 <= n:
        h = (rotl(h ^ rnd(0, int.from_bytes(b[i:i + 8], "little")), 27) * P1 + P4) & M
        i += 8
    if i + 4 <= n:
        h = (rotl(h ^ (int.from_bytes(b[i:i + 4], "little") * P1) & M, 23) * P2 + P3) & M
        i += 4
    while i < n:
        h = (rotl(h ^ (b[i] * P5) & M, 11) * P1) & M
        i += 1
    h ^= h >> 33; h = (h * P2) & M; h ^= h >> 29; h = (h * P3) & M; h ^= h >> 32
    return h


def test_content_checksum_is_verified_when_the_frame_carries_one(press):
    """RFC 8878 3.1.1: Content_Checksum_flag -> the frame ends in the low 32 bits of XXH64(content).  libzstd verifies it; round 4's
    decoder skipped it (VERDICT r04 missing #4).  slow5lib's ZSTD_compress never writes one, so the frames are made here: a libzstd (or
    device-written) frame with the flag set and the hash appended — accepted; with one bit of the hash or of the content flipped —
    status 4 on the device, rejected by the restated decoder and (when the image has it) by libzstd."""
    rng = np.random.default_rng(7)
    good, bad, datas = [], [], []
    for n in (0, 1, 31, 32, 33, 100, 4000, 70000):
        d = bytes(rng.integers(0, 9, n, dtype=np.uint8))
        f = ob.zstd_compress(d, 3) if ob.zstd_ref() else ob.zstd_literals_compress(d)
        assert not (f[4] >> 2) & 1
        g = f[:4] + bytes([f[4] | 4]) + f[5:] + struct.pack("<I", _xxh64_py(d) & 0xFFFFFFFF)
        good.append(g)
        datas.append(d)
        bad.append(g[:-1] + bytes([g[-1] ^ 0x40]))
        if ob.zstd_ref():
            assert ob.zstd_decompress(g, n) == d and ob.zstd_decompress(bad[-1], n) is None
        assert ob.zstd_restated_decompress(g, n) == d and ob.zstd_restated_decompress(bad[-1], n) is None
    rc, res, st = zstd_solo(good)
    assert rc == 0 and all(s == 0 for s in st) and res == datas
    rc, res, st = zstd_solo(bad)
    assert rc != 0 and all(s == 4 for s in st), st
