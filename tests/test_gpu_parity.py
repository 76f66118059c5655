"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (north_star): svb-zd stage bit-exact vs the oracle; the DEFLATE stage must be a valid zlib stream
that STOCK zlib inflates to the byte-identical record payload; decode returns the identical signal.
Mirrors the reference's byte-identical diffs in test/test_view.sh:90-165 (encode and decode, all
codec combinations) and test/test_view_integrity.sh:62-68 (round trip).
"""
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import Blow5, NONE_NONE_FIXTURES, ZLIB_NONE_FIXTURES, ZLIB_SVB_FIXTURES, golden

pytestmark = pytest.mark.gpu

HDR_ARGS = (0, 8192.0, 23.0, 1467.61, 4000.0)


@pytest.fixture(scope="module")
def press():
    from slow5tools_amd import _lib, press as p

    _lib.check(_lib.lib().s5gpu_init(0), "s5gpu_init")
    return p


def _hdr(press, i):
    return press.pack_hdr(ob.synth_read_id(i), *HDR_ARGS)


def _oracle_payload(hdr, signal, aux, sig_method):
    idl = struct.unpack_from("<H", hdr, 0)[0]
    rg, dg, of, rn, sr = struct.unpack_from("<Idddd", hdr, 2 + idl)
    r, keep = ob.make_rec(hdr[2:2 + idl], rg, dg, of, rn, sr, signal, aux)
    return ob.rec_pack(r, sig_method), ob.rec_to_mem(r, ob.REC_ZLIB, sig_method)


def _check_records(press, signals, hdrs, auxs, recs, rec_method, sig_method):
    """every record: [u64 size] prefix right, stock zlib inflates to the oracle's payload"""
    tot_gpu = tot_ref = 0
    for i, rec in enumerate(recs):
        aux = auxs[i] if auxs is not None else b""
        payload, ref_mem = _oracle_payload(hdrs[i], signals[i], aux, sig_method)
        (sz,) = struct.unpack_from("<Q", rec, 0)
        assert sz == len(rec) - 8, (i, sz, len(rec))
        body = rec[8:]
        if rec_method == press.REC_ZLIB:
            assert body[:2] == b"\x78\x9c"
            got = zlib.decompress(body)                      # Python's stock zlib
            assert got == payload, "record %d: inflate mismatch" % i
            assert ob.zlib_decompress(body, len(payload) + 16) == payload   # system libz via the oracle
            tot_gpu += len(rec)
            tot_ref += len(ref_mem)
        else:
            assert body == payload
    return tot_gpu, tot_ref


# ---------------------------------------------------------------- svb-zd stage: bit-exact
@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 4000, 4095, 4096, 4097, 12289, 65536])
def test_svbzd_encode_bit_exact_lengths(press, n):
    rng = np.random.default_rng(n + 1)
    sigs = [rng.integers(-32768, 32768, n, dtype=np.int16), (rng.normal(500, 40, n)).astype(np.int16),
            np.zeros(n, np.int16), np.full(n, -7, np.int16)]
    b = press.DeviceBatch([n] * len(sigs), with_stream_out=False)
    b.upload(sigs, [_hdr(press, i) for i in range(len(sigs))])
    b.svbzd_encode()
    for s, blob in zip(sigs, b.records()):
        assert blob == ob.svbzd_encode(s)


def test_svbzd_encode_synthetic_4000_bit_exact(press):
    n_reads, n = 512, 4000
    b = press.DeviceBatch([n] * n_reads, with_stream_out=False)
    b.synth(seed=0x5105, first=1000)
    sig_dev = b.sig[: n_reads * n].cpu().numpy().reshape(n_reads, n)
    sig_cpu = ob.synth_reads(0x5105, 1000, n_reads, n)
    assert np.array_equal(sig_dev, sig_cpu)                 # device generator == oracle generator
    hdr_dev = b.hdr[: 74 * n_reads].cpu().numpy().tobytes()
    assert hdr_dev[:74] == _hdr(press, 1000) and hdr_dev[-74:] == _hdr(press, 1000 + n_reads - 1)
    b.svbzd_encode()
    for s, blob in zip(sig_cpu, b.records()):
        assert blob == ob.svbzd_encode(s)


def test_svbzd_encode_adversarial(press):
    sigs = [np.tile(np.array([32767, -32768], np.int16), 2500), np.arange(-3000, 3000, dtype=np.int16),
            np.random.default_rng(7).integers(-32768, 32768, 100000, dtype=np.int16)]
    b = press.DeviceBatch([len(s) for s in sigs], with_stream_out=False)
    b.upload(sigs, [_hdr(press, i) for i in range(len(sigs))])
    b.svbzd_encode()
    for s, blob in zip(sigs, b.records()):
        assert blob == ob.svbzd_encode(s)


def test_svbzd_encode_packed_path_and_its_edges(press):
    """Round 5: svb_tile_classify does two samples per instruction when every sample a wave holds lies in [-16384, 16384): the largest
    deltas that path can meet, its limits on either side, one sample outside it at the seams (first / last of a lane, of a wave, of a tile,
    the sample in front of a tile), and through the fused record encoder as well — all against the oracle, bit for bit."""
    rng = np.random.default_rng(55)
    n = 9000
    base = rng.integers(-16384, 16384, n).astype(np.int16)
    sigs = [base,
            np.tile(np.array([16383, -16384], np.int16), n // 2),          # deltas of +-32767: two-byte values throughout
            np.tile(np.array([-16384, -16384, 16383, 16383], np.int16), n // 4),
            np.full(n, 16383, np.int16), np.full(n, -16384, np.int16),
            (rng.normal(500, 40, n)).astype(np.int16)]
    for pos in (0, 1, 15, 16, 17, 1023, 1024, 4095, 4096, 4097, 8191, 8192, 8999):
        for v in (16384, -16385, 32767, -32768):
            t = base.copy(); t[pos] = v
            sigs.append(t)
    b = press.DeviceBatch([len(x) for x in sigs], with_stream_out=False)
    b.upload(sigs, [_hdr(press, i) for i in range(len(sigs))])
    b.svbzd_encode()
    for x, blob in zip(sigs, b.records()):
        assert blob == ob.svbzd_encode(x)
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    for x, h, rec in zip(sigs, hdrs, press.encode_records(sigs, hdrs, None, press.REC_ZLIB, press.SIG_SVB_ZD)):
        assert zlib.decompress(rec[8:]) == _oracle_payload(h, x, b"", press.SIG_SVB_ZD)[0]


def test_svbzd_stream_one_pass_equals_slots_plus_compaction(press):
    """s5gpu_svbzd_encode_stream_dev: the blob stream and the offsets of svbzd_encode + compact, in one launch; ragged lengths,
    empty reads and a read past the LDS budget (ctl[0] tells the caller to take the two-pass route)"""
    rng = np.random.default_rng(11)
    lens = [4000, 0, 1, 3, 5, 4096, 4097, 777, 12289, 2, 4000, 16, 9000] + [int(x) for x in rng.integers(0, 6000, 300)]
    sigs = [(rng.normal(500, 60, n)).astype(np.int16) if i % 3 else rng.integers(-32768, 32768, n, dtype=np.int16) for i, n in enumerate(lens)]
    b = press.DeviceBatch(lens)
    b.upload(sigs, [_hdr(press, i) for i in range(len(lens))])
    b.svbzd_encode()
    b.compact()
    want, want_off = b.stream_bytes()
    b.stream_out.zero_()
    b.rec_off.zero_()
    b.svbzd_encode_stream()
    assert b.stream_ok()
    got, got_off = b.stream_bytes()
    assert np.array_equal(got_off, want_off) and got == want
    blobs = [got[int(got_off[i]):int(got_off[i + 1])] for i in range(len(lens))]
    for s, blob in zip(sigs[:40], blobs[:40]):
        assert blob == ob.svbzd_encode(s)
    for k in (1, 2, 3, 5):                          # fewer reads than a group of four, and a last group that is not full
        bk = press.DeviceBatch(lens[:k] if k < 5 else [4000, 4000, 17, 4000, 3999])
        sk = sigs[:k] if k < 5 else [rng.integers(-500, 500, n).astype(np.int16) for n in (4000, 4000, 17, 4000, 3999)]
        bk.upload(sk, [_hdr(press, i) for i in range(k)])
        bk.svbzd_encode_stream()
        assert bk.stream_ok()
        gk, ok_ = bk.stream_bytes()
        assert [gk[int(ok_[i]):int(ok_[i + 1])] for i in range(k)] == [ob.svbzd_encode(x) for x in sk]
    # one read that cannot be staged in LDS: the call reports it, nothing hangs
    big = [4000, 200000, 4000]
    b2 = press.DeviceBatch(big)
    b2.upload([rng.integers(-32768, 32768, n, dtype=np.int16) for n in big], [_hdr(press, i) for i in range(3)])
    b2.svbzd_encode_stream()
    assert not b2.stream_ok()


# ---------------------------------------------------------------- full encode: valid zlib, identical payload
def test_full_encode_synthetic_4000(press):
    n_reads, n = 1024, 4000
    b = press.DeviceBatch([n] * n_reads)
    b.synth(seed=0x5105, first=0)
    b.encode()
    b.compact()
    recs = b.records()
    sig = ob.synth_reads(0x5105, 0, n_reads, n)
    hdrs = [_hdr(press, i) for i in range(n_reads)]
    tg, tr = _check_records(press, sig, hdrs, None, recs, press.REC_ZLIB, press.SIG_SVB_ZD)
    assert tg <= 1.02 * tr, "GPU records %.4f x zlib-L6 size" % (tg / tr)
    stream, off = b.stream_bytes()
    assert stream == b"".join(recs)
    assert list(off[: n_reads + 1]) == list(np.concatenate([[0], np.cumsum([len(r) for r in recs])]))


@pytest.mark.parametrize("n_reads", [1, 7, 300, 20000])
def test_single_pass_stream_equals_slots_plus_compaction(press, n_reads):
    """ordered single-pass output (decoupled look-back) == encode into slots + compaction, byte for byte"""
    n = 4000
    b = press.DeviceBatch([n] * n_reads)
    b.synth(seed=0x5105, first=31)
    b.encode()
    b.compact()
    want, off_want = b.stream_bytes()
    want_len = b.out_len[:n_reads].cpu().numpy().copy()
    b.stream_out.zero_()
    b.rec_off.zero_()
    for _ in range(3):   # repeated launches reuse the look-back state: it must be reset every call
        b.encode_stream()
    got, off_got = b.stream_bytes()
    assert b.stream_ok()
    assert list(off_got[: n_reads + 1]) == list(off_want[: n_reads + 1])
    assert got == want
    assert (b.out_len[:n_reads].cpu().numpy() == want_len).all()


def test_single_pass_stream_reports_lds_overflow(press):
    rng = np.random.default_rng(3)
    sig = ob.synth_reads(0x5105, 0, 64, 4000)
    sig[11] = rng.integers(-32768, 32768, 4000, dtype=np.int16)     # does not fit the LDS budget
    b = press.DeviceBatch([4000] * 64)
    b.upload(list(sig), [_hdr(press, i) for i in range(64)])
    b.encode_stream()
    b.torch.cuda.synchronize()
    assert not b.stream_ok() and int(b.lb_ctl[0].item()) == 1        # the caller falls back to encode + compact
    b.encode()
    b.compact()
    _check_records(press, sig, [_hdr(press, i) for i in range(64)], None, b.records(), 1, 1)


@pytest.mark.parametrize("rec_method,sig_method", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_encode_all_method_combinations_ragged(press, rec_method, sig_method):
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 3, 5, 16, 100, 1000, 4000, 4096, 4097, 9999, 20000]
    sigs = []
    for k, n in enumerate(lens):
        base = (500 + 30 * rng.standard_normal(n)).astype(np.int16)
        if k % 3 == 1:
            base[: n // 2] = 77        # long constant run
        sigs.append(base)
    hdrs = [press.pack_hdr(("read-%d" % k) * (1 + k % 3), k, 8192.0, 3.0 + k, 1467.61, 4000.0) for k in range(len(lens))]
    auxs = [bytes(rng.integers(0, 256, (7 * k) % 53, dtype=np.uint8)) for k in range(len(lens))]
    recs = press.encode_records(sigs, hdrs, auxs, rec_method, sig_method)
    _check_records(press, sigs, hdrs, auxs, recs, rec_method, sig_method)


def test_encode_incompressible_and_constant(press):
    rng = np.random.default_rng(5)
    sigs = [rng.integers(-32768, 32768, 4000, dtype=np.int16),      # ~all 3-byte codes, stored-block territory
            np.zeros(4000, np.int16), np.full(50000, 1234, np.int16),  # pure runs
            np.tile(np.array([32767, -32768], np.int16), 3000)]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    recs = press.encode_records(sigs, hdrs)
    tg, tr = _check_records(press, sigs, hdrs, None, recs, 1, 1)
    # random data must not blow up: stored fallback keeps it within a few bytes of the payload
    assert len(recs[0]) <= len(_oracle_payload(hdrs[0], sigs[0], b"", 1)[0]) + 8 + 2 + 4 + 6


def test_encode_long_reads_staged_path(press):
    sig = ob.synth_reads(0x5105, 5, 3, 100000)
    sigs = [sig[0], sig[1][:65537], sig[2][:30000]]
    hdrs = [_hdr(press, i) for i in range(3)]
    recs = press.encode_records(sigs, hdrs)
    tg, tr = _check_records(press, sigs, hdrs, None, recs, 1, 1)
    assert tg <= 1.02 * tr


def test_staged_encode_as_two_calls_equals_the_one_call(press):
    """s5gpu_pack_parked_dev + s5gpu_deflate_parked_dev (the two steps of the staged path, as a chunked job puts them on different
    streams) leave the same records in the slots as s5gpu_encode_dev on a batch of long reads"""
    rng = np.random.default_rng(29)
    lens = [70000, 100000, 65537, 131072, 90001]
    sigs = [ob.synth_read(0x5105, 9000 + i, n) if i % 2 == 0 else (500 + rng.integers(-300, 300, n)).astype(np.int16) for i, n in enumerate(lens)]
    hdrs = [_hdr(press, 9000 + i) for i in range(len(lens))]
    auxs = [b"", bytes(range(40)), b"", b"\x01" * 7, b""]
    b = press.DeviceBatch(lens, aux_len=[len(a) for a in auxs])
    b.upload(sigs, hdrs, auxs)
    b.encode()
    one = b.records()
    b.slots.zero_()
    b.out_len.zero_()
    b.pack_parked()
    b.deflate_parked()
    two = b.records()
    assert one == two
    _check_records(press, sigs, hdrs, auxs, two, press.REC_ZLIB, press.SIG_SVB_ZD)


@pytest.mark.parametrize("cap", [0, 2048, 5000, 5400])
def test_lds_overflow_reads_take_the_staged_path(press, cap):
    """reads whose payload exceeds the fused kernel's LDS budget (incompressible signal, or a forced
    small budget) are redone by the HBM-staged kernels: same bytes out of stock zlib either way"""
    rng = np.random.default_rng(17)
    n_reads, n = 96, 4000
    sig = ob.synth_reads(0x5105, 400, n_reads, n)
    sig[5] = rng.integers(-32768, 32768, n, dtype=np.int16)       # ~3 bytes/sample: never fits the default budget
    sig[40] = rng.integers(-2000, 2000, n, dtype=np.int16)        # ~2 bytes/sample
    sig[41][:] = 0
    hdrs = [_hdr(press, 400 + i) for i in range(n_reads)]
    from slow5tools_amd import _lib
    L = _lib.lib()
    _lib.check(L.s5gpu_set_option(b"fused_tier2", 0))            # one fused launch: whatever misses the named budget is staged
    try:
        b = press.DeviceBatch([n] * n_reads, lds_payload_cap=cap)
        b.upload(list(sig), hdrs)
        b.encode()
        b.compact()
        recs = b.records()
        n_ovf = int(b.ovf[0].item())
    finally:
        _lib.check(L.s5gpu_set_option(b"fused_tier2", 0))
    if cap == 2048:
        assert n_ovf == n_reads
    elif cap == 0:
        assert 2 <= n_ovf <= 4
    _check_records(press, sig, hdrs, None, recs, press.REC_ZLIB, press.SIG_SVB_ZD)
    stream, off = b.stream_bytes()
    assert stream == b"".join(recs)


@pytest.mark.parametrize("name", ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES + NONE_NONE_FIXTURES)
def test_reencode_fixture_records(press, name):
    """decode with the oracle, re-encode on the GPU, inflate with stock zlib -> the fixture's payload"""
    f = Blow5(golden(name))
    sigs, hdrs, auxs, pays = [], [], [], []
    for rec in f.records:
        pl = zlib.decompress(rec) if f.rec_method == 1 else rec
        d = ob.rec_parse(pl, f.sig_method)
        sigs.append(d["signal"])
        hdrs.append(press.pack_hdr(d["read_id"], d["read_group"], d["digitisation"], d["offset"], d["range"], d["sampling_rate"]))
        auxs.append(d["aux"])
        pays.append(pl)
    recs = press.encode_records(sigs, hdrs, auxs, f.rec_method, f.sig_method)
    tg = tr = 0
    for rec, pl, ref in zip(recs, pays, f.records):
        body = rec[8:]
        assert struct.unpack_from("<Q", rec, 0)[0] == len(body)
        assert (zlib.decompress(body) if f.rec_method == 1 else body) == pl
        tg += len(body)
        tr += len(ref)
    if f.rec_method == 1 and f.sig_method == 1:
        assert tg <= 1.02 * tr, (tg, tr)
    if f.rec_method == 1 and f.sig_method == 0:
        # raw int16 samples under zlib (test/test_view.sh:89-91, every v0.1.0 file): the LZ77 matcher (csrc/lz_dev.h) has to find
        # what zlib level 6 finds — within 3 % of the reference's record (run-length + Huffman alone: 12.7 % over)
        assert tg <= 1.03 * tr, (tg, tr)


# ---------------------------------------------------------------- decode
@pytest.fixture(params=["wave-per-record", "lane-per-record", "parallel-in-record"])
def inflate_kernel(request, press):
    """all three inflate kernels must pass every decode test: force each through the tuning knobs (the default is the decoder
    that is parallel inside a record, with the wave-per-record one behind it for what it declines)"""
    from slow5tools_amd import _lib

    L = _lib.lib()
    _lib.check(L.s5gpu_set_option(b"inflate_par", 1 if request.param == "parallel-in-record" else 0))
    _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 1 if request.param == "lane-per-record" else 1 << 30))
    yield request.param
    _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 24576))
    _lib.check(L.s5gpu_set_option(b"inflate_par", 1))


@pytest.mark.parametrize("name", ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES + NONE_NONE_FIXTURES)
def test_decode_fixture_records(press, inflate_kernel, name):
    """records written by the reference (stock zlib, arbitrary LZ77 distances) decode to the oracle's answer"""
    f = Blow5(golden(name))
    got = press.decode_records(f.records, f.rec_method, f.sig_method)
    for g, rec in zip(got, f.records):
        pl = zlib.decompress(rec) if f.rec_method == 1 else rec
        d = ob.rec_parse(pl, f.sig_method)
        assert g["status"] == 0
        assert g["payload"] == pl
        assert np.array_equal(g["signal"], d["signal"])
        for k in ("read_id", "read_group", "digitisation", "offset", "range", "sampling_rate", "aux"):
            assert g[k] == d[k], k


def test_roundtrip_own_streams(press, inflate_kernel):
    n_reads, n = 300, 4000
    sig = ob.synth_reads(0x5105, 77, n_reads, n)
    hdrs = [_hdr(press, 77 + i) for i in range(n_reads)]
    recs = press.encode_records(list(sig), hdrs)
    got = press.decode_records([r[8:] for r in recs])
    for i, g in enumerate(got):
        assert g["status"] == 0 and np.array_equal(g["signal"], sig[i]) and g["read_id"] == ob.synth_read_id(77 + i)


def test_decode_highly_compressible_records_retry_with_exact_slots(press, inflate_kernel):
    """a constant signal deflates > 100x: the first payload-slot guess (4x + 4 KiB) overflows, the decoder reports
    the size it needs and the batch call retries (status 5 -> exact slot), transparently"""
    sigs = [np.full(50000, 321, np.int16), np.zeros(200000, np.int16), ob.synth_read(0x5105, 3, 4000)]
    hdrs = [_hdr(press, i) for i in range(3)]
    recs = [r[8:] for r in press.encode_records(sigs, hdrs)]
    assert len(recs[0]) < 1000 and len(recs[1]) < 2000
    for g, s in zip(press.decode_records(recs), sigs):
        assert g["status"] == 0 and np.array_equal(g["signal"], s)


def test_staged_stored_blocks_behind_a_token_list_in_the_bit_buffer(press):
    """Incompressible payloads with a few planted 4-byte runs through k_deflate_staged: the runs make general slabs whose token list (> 640
    entries) continues inside the bit buffer, the random bytes make the block a STORED one — whose plain byte stores end inside the words the
    list stood in.  The Adler-32 trailer (and a following block's carry) are ORed in behind them: those words have to be clean (round-5
    advisor finding, deflate2_dev.h stored branch).  Every record must come out of stock zlib as the oracle's payload."""
    rng = np.random.default_rng(606)
    from slow5tools_amd import _lib
    L = _lib.lib()
    sigs, hdrs, auxs = [], [], []
    for i, alen in enumerate(list(range(3000, 12500, 250)) + [15800, 16300, 16500, 20000, 33000]):
        a = rng.integers(0, 256, alen, dtype=np.uint8)
        for k in range(3 + i % 37):                         # planted runs, spread over the block: three or more general slabs
            at = int(rng.integers(0, alen - 8))
            a[at:at + 4 + k % 3] = a[at]
        sigs.append(np.asarray([3, 4, 5, 4], dtype=np.int16))
        hdrs.append(_hdr(press, 7000 + i))
        auxs.append(a.tobytes())
    _lib.check(L.s5gpu_set_option(b"fused_tier2", 0))            # one fused launch: whatever misses the named budget is staged
    try:
        b = press.DeviceBatch([4] * len(sigs), aux_len=[len(a) for a in auxs], lds_payload_cap=2048)
        b.upload(sigs, hdrs, auxs)
        b.encode()
        recs = b.records()
        assert int(b.ovf[0].item()) == len(sigs)                 # every read went through the staged kernel
    finally:
        _lib.check(L.s5gpu_set_option(b"fused_tier2", 0))
    for i, rec in enumerate(recs):
        payload, _ = _oracle_payload(hdrs[i], sigs[i], auxs[i], 1)
        assert zlib.decompress(rec[8:]) == payload, "record %d (aux %d bytes)" % (i, len(auxs[i]))
        assert len(rec) <= len(payload) + 8 + 2 + 4 + 5 * (len(payload) // 16384 + 1) + 64


def test_very_long_read_roundtrip(press):
    """2 M samples (the reference's longest fixture read, p2solo_ulk114_dna, is 2 050 027): ~160 DEFLATE blocks
    through the staged kernels, in place in the slot"""
    sig = ob.synth_read(0x5105, 9, 2_050_027)
    hdr = _hdr(press, 9)
    rec = press.encode_records([sig], [hdr])[0]
    payload, ref = _oracle_payload(hdr, sig, b"", 1)
    assert zlib.decompress(rec[8:]) == payload
    assert len(rec) <= 1.01 * len(ref)
    g = press.decode_records([rec[8:]])[0]
    assert g["status"] == 0 and np.array_equal(g["signal"], sig)


def test_decode_rejects_corrupt_records(press, inflate_kernel):
    sig = ob.synth_reads(0x5105, 0, 4, 4000)
    hdrs = [_hdr(press, i) for i in range(4)]
    recs = [r[8:] for r in press.encode_records(list(sig), hdrs)]
    bad_adler = recs[0][:-1] + bytes([recs[0][-1] ^ 1])
    trunc = recs[1][: len(recs[1]) // 2]
    bad_hdr = b"\x00\x00" + recs[2][2:]
    got = press.decode_records([bad_adler, trunc, bad_hdr, recs[3]], raise_on_error=False)
    assert got[0]["status"] == 4 and got[1]["status"] in (2, 3) and got[2]["status"] == 1
    assert got[3]["status"] == 0 and np.array_equal(got[3]["signal"], sig[3])
    with pytest.raises(press.S5GpuError):
        press.decode_records([trunc])


def test_decode_stored_and_fixed_blocks(press, inflate_kernel):
    """streams from other encoders: zlib level 0 (stored), Z_FIXED, and a 32 KiB-distance match"""
    rng = np.random.default_rng(3)
    sig = (500 + 30 * rng.standard_normal(70000)).astype(np.int16)
    hdr = _hdr(press, 1)
    payload, _ = _oracle_payload(hdr, sig, b"", 1)
    streams = [zlib.compress(payload, 0), zlib.compress(payload, 9)]
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    streams.append(c.compress(payload) + c.flush())
    for g in press.decode_records(streams):
        assert g["status"] == 0 and np.array_equal(g["signal"], sig)
    # raw int16 record with a far repeat (distance ~32 KiB) exercises window copies
    sig2 = np.concatenate([sig[:16000], sig[:16000]])
    payload2, _ = _oracle_payload(hdr, sig2, b"", 0)
    g = press.decode_records([zlib.compress(payload2, 9)], 1, 0)[0]
    assert g["status"] == 0 and np.array_equal(g["signal"], sig2)


def test_decode_fuzzed_streams_terminate_and_never_pass_wrong_data(press, inflate_kernel):
    """2000 damaged variants of valid records (bit flips, truncations, garbage tails, zero fill) through both inflate
    kernels: the call returns (no hang — the whole test is bounded by the suite's timeout), every record gets a status, and
    a record that still reports status 0 carries exactly the original signal (the Adler-32 + svb checks hold the line)."""
    rng = np.random.default_rng(2024)
    n0 = 40
    sig = ob.synth_reads(0x5105, 100, n0, 4000)
    hdrs = [_hdr(press, 100 + i) for i in range(n0)]
    good = [r[8:] for r in press.encode_records(list(sig), hdrs)]
    # also zlib-written streams (matches with real distances) of the same payloads
    for i in range(8):
        payload, _ = _oracle_payload(hdrs[i], sig[i], b"", 1)
        good.append(zlib.compress(payload, 6))
    src_of = []
    bad = []
    for v in range(2000):
        k = int(rng.integers(0, len(good)))
        b = bytearray(good[k])
        mode = v % 5
        if mode == 0:      # one bit flip somewhere
            p = int(rng.integers(0, len(b)))
            b[p] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:    # burst of random bytes
            p = int(rng.integers(0, len(b) - 8))
            b[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
        elif mode == 2:    # truncation
            b = b[: int(rng.integers(1, len(b)))]
        elif mode == 3:    # zero fill from a random point
            p = int(rng.integers(2, len(b)))
            b[p:] = bytes(len(b) - p)
        else:              # garbage appended after a cut
            p = int(rng.integers(2, len(b)))
            b = b[:p] + rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
        bad.append(bytes(b))
        src_of.append(k % n0 if k < n0 else k - n0)
    got = press.decode_records(bad, raise_on_error=False)
    assert len(got) == len(bad)
    n_ok = 0
    for g, k in zip(got, src_of):
        assert 0 <= g["status"] <= 7
        if g["status"] == 0:
            n_ok += 1
            assert np.array_equal(g["signal"], sig[k]), "a damaged record decoded to different data with status 0"
    assert n_ok < len(bad) // 10          # almost every damage is caught (a flip in the unused tail of a byte may survive)


def test_dynamic_headers_of_every_shape(press, inflate_kernel):
    """hand-made dynamic blocks (tests/deflate_craft.py; stock zlib inflates every valid one first): what varies is the block HEADER — the
    parallel decoder takes the code-length sequence 1024 bits per round, sixteen bit offsets per lane and at most eight tokens per lane
    (infl_cl_sequence_wave2), so: sequences of one, two and three rounds, tokens of one bit (sixteen to a lane: the 64-offset parser
    takes over), repeats that reach across lanes and rounds, a repeat of a ZERO length, and the headers that are wrong — a repeat with
    nothing in front, a run over the end, seven bits that start no code in the middle of the sequence"""
    import deflate_craft as dc

    rng = np.random.default_rng(77)
    streams, sigs, shapes = [], [], []

    def add(sig, litlens, **kw):
        payload, _ = _oracle_payload(_hdr(press, len(streams)), sig, b"", 0)
        s, nbits = dc.zlib_stream(payload, litlens, **kw)
        streams.append(s); sigs.append(sig); shapes.append(nbits)

    # skewed bytes: lit/len lengths 5 .. 13
    sig = np.minimum(255, rng.geometric(0.03, 6000)).astype(np.int16)
    pay, _ = _oracle_payload(_hdr(press, 0), sig, b"", 0)
    freq = [1] * 257
    for x in pay:
        freq[x] += 1
    skew = dc.huff_lengths(freq, 15)
    for mode in ("plain", "rle", "rep0"):
        add(sig, skew, mode=mode)                                           # ~550 .. 770 bits: one round
    add(sig, skew, mode="plain", cl_weights={k: 1 for k in range(19)})      # every token five bits: two rounds
    # 286 + 30 lengths, one seven-bit token each: 2212 bits, three rounds
    ll = [8, 9] * 60 + [8] * 166                                             # 226 x 8 + 60 x 9 bits: complete
    assert len(ll) == 286 and sum(2 ** (15 - l) for l in ll) == 32768
    dl = [4, 4] + [5] * 28
    w7 = {8: 1, 9: 1, 4: 2, 5: 4, 0: 8, 1: 16, 2: 32, 3: 64}               # lengths 7 7 6 5 4 3 2 1: the two lit/len lengths cost seven bits
    sig8 = rng.integers(-3000, 3000, 5000).astype(np.int16)
    add(sig8, ll, mode="plain", cl_weights=w7, dlens=dl)
    assert shapes[-1] > 2048
    # one-bit tokens: all lit/len lengths 8 but two, "8" is the 1-bit code — sixteen tokens in a lane's sixteen bits
    flat = [8] * 255 + [9, 9]
    add(sig8, flat, mode="plain", cl_weights={8: 100, 9: 3, 1: 2})
    add(sig8, flat, mode="rle", cl_weights={8: 100, 9: 3, 1: 2, 16: 50})    # 42 repeats of six in a row: two-bit codes + two extra bits
    add(sig8, flat + [0] * 29, mode="rep0")                                   # a zero length repeated by symbol 16
    add(sig8, flat + [0] * 29, mode="rle")
    got = press.decode_records(streams, 1, 0)
    for g, s_, nb in zip(got, sigs, shapes):
        assert g["status"] == 0 and np.array_equal(g["signal"], s_), "code-length sequence of %d bits" % nb
    # wrong headers: a status, never a hang, never status 0
    cll = [0] * 19
    for s_, l_ in ((0, 2), (8, 2), (16, 2), (17, 3), (18, 3)):
        cll[s_] = l_
    bad = [dc.raw_stream_with_sequence([(16, 0), (8, 0)] + [(8, 0)] * 300, cll, 257, 1),                       # nothing to repeat
           dc.raw_stream_with_sequence([(18, 127), (18, 127)], cll, 257, 1),                                   # 276 lengths for 258
           dc.raw_stream_with_sequence([(8, 0)] * 256 + [(16, 3)], cll, 257, 1)]                               # 262 for 258
    inc = [0] * 19
    for s_, l_ in ((0, 2), (8, 2), (18, 3)):
        inc[s_] = l_                                                        # an incomplete code-length code: 101, 110, 111, ... start no code
    b = dc.Bits()
    b.put(1, 1); b.put(2, 2); b.put(0, 5); b.put(0, 5); b.put(15, 4)
    for k in range(19):
        b.put(inc[dc.ORDER[k]], 3)
    clc = dc.canonical(inc)
    for _ in range(40):
        b.put_code(clc[8], 2)
    b.put(0b111, 3)                                                         # no code, 40 lengths into the sequence
    b.put(0, 700)
    bad.append(b"\x78\x9c" + b.done() + b"\x00\x00\x00\x01")
    for g in press.decode_records(bad, 1, 0, raise_on_error=False):
        assert g["status"] in (2, 3)


def test_big_host_batch_split_over_two_contexts_equals_one_context(press, monkeypatch):
    """s5gpu_encode_batch cuts batches of >= 16384 reads in two halves that run concurrently: same records, same order"""
    rng = np.random.default_rng(77)
    n = 20001
    sigs = [(500 + rng.integers(-30, 30, int(rng.integers(1, 120)))).astype(np.int16) for _ in range(n)]
    hdrs = [press.pack_hdr(b"r%d" % i, i % 5, 8192.0, 1.0, 1400.0, 4000.0) for i in range(n)]
    monkeypatch.setenv("S5GPU_SPLIT", "0")
    one = press.encode_records(sigs, hdrs)
    monkeypatch.setenv("S5GPU_SPLIT", "1")
    two = press.encode_records(sigs, hdrs)
    assert one == two
    for i in (0, n // 2 - 1, n // 2, n - 1):
        payload, _ = _oracle_payload(hdrs[i], sigs[i], b"", 1)
        assert zlib.decompress(two[i][8:]) == payload


def test_big_mixed_length_batch_is_routed_by_length(press):
    """a batch of >= 1024 zlib records with very different lengths: the lane kernel takes them counting-sorted by compressed length
    and hands records of >= 32 KiB to the wave kernel; the result is the one the unrouted kernels give"""
    from slow5tools_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(17)
    ns = np.clip(np.exp(rng.normal(np.log(1500), 1.0, 1400)), 1, 90000).astype(int)
    ns[:4] = (0, 1, 90000, 60000)
    sigs = [ob.synth_read(0x77, i, int(n)) for i, n in enumerate(ns)]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    recs = [r[8:] for r in press.encode_records(sigs, hdrs)]
    assert sum(len(r) >= 32768 for r in recs) >= 2 and sum(len(r) < 32768 for r in recs) > 1024
    _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 1), "set_option")
    _lib.check(L.s5gpu_set_option(b"inflate_par", 0), "set_option")
    try:
        outs = []
        for route in (1, 0):
            _lib.check(L.s5gpu_set_option(b"inflate_route", route), "set_option")
            outs.append(press.decode_records(recs))
    finally:
        _lib.check(L.s5gpu_set_option(b"inflate_route", 1), "set_option")
        _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 24576), "set_option")
        _lib.check(L.s5gpu_set_option(b"inflate_par", 1), "set_option")
    outs.append(press.decode_records(recs))            # and the default: parallel inside the record, many rounds for the long ones
    for g, h, q, s in zip(outs[0], outs[1], outs[2], sigs):
        assert g["status"] == 0 and h["status"] == 0 and q["status"] == 0 and np.array_equal(g["signal"], s) and g["payload"] == h["payload"] == q["payload"]


def test_host_batch_with_one_very_long_read_keeps_the_short_reads_fused(press):
    """a batch the device entry point would stage as a whole (longest read >> the LDS budget): the host call names an 8 KiB fused
    budget from the lengths it sees — short reads fused, the long ones through the overflow list; same records either way"""
    rng = np.random.default_rng(23)
    sigs = [ob.synth_read(0x99, 0, 230000)] + [ob.synth_read(0x99, 1 + i, int(n)) for i, n in enumerate(rng.integers(1, 9000, 40))]
    hdrs = [_hdr(press, i) for i in range(len(sigs))]
    recs = press.encode_records(sigs, hdrs)
    for r, h, s in zip(recs, hdrs, sigs):
        payload, _ = _oracle_payload(h, s, b"", 1)
        assert zlib.decompress(r[8:]) == payload


# ---------------------------------------------------------------- LZ77 matcher (signal press none / solo zlib press)
def _solo_zlib(bufs):
    import ctypes as C

    from slow5tools_amd import _lib

    L = _lib.lib()
    n = len(bufs)
    vp = C.c_void_p
    keep = [C.create_string_buffer(b, max(len(b), 1)) for b in bufs]
    inp = (vp * n)(*[C.addressof(k) for k in keep])
    il = (C.c_size_t * n)(*[len(b) for b in bufs])
    out = (vp * n)()
    ol = (C.c_size_t * n)()
    st = (C.c_int32 * n)()
    _lib.check(L.s5gpu_solo_batch(0, n, inp, il, out, ol, st), "solo zlib")
    libc = C.CDLL(None)
    libc.free.argtypes = [vp]
    res = [C.string_at(out[i], ol[i]) for i in range(n)]
    for i in range(n):
        libc.free(out[i])
    return res


def test_lz77_matcher_streams_inflate_with_stock_zlib_and_find_the_redundancy(press):
    """every shape the matcher's stages can get wrong: matches at the four short distances and through the hash table, at block
    boundaries (16 KiB blocks, history = the previous block), runs longer than a lane's 64 positions (the parse hands overhangs
    from lane to lane), a whole buffer of one value (the longest hand-over chain), incompressible bytes (stored blocks), text"""
    rng = np.random.default_rng(12)
    word = lambda: bytes(rng.integers(97, 123, int(rng.integers(3, 9)), dtype=np.uint8))
    vocab = [word() for _ in range(300)]
    text = b" ".join(vocab[int(i)] for i in rng.integers(0, 300, 30000))
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    sig = (520 + 30 * rng.standard_normal(60000)).astype(np.int16).tobytes()
    period = lambda p, n: (bytes(rng.integers(0, 256, p, dtype=np.uint8)) * (n // p + 1))[:n]
    bufs = [b"", b"a", b"abc", b"abcd" * 2, bytes(5), bytes(100000), b"\xff" * 16384, b"\x01" * 16385, text, text[:16383], text[:16384], text[:16385],
            text[:32769], noise, noise[:1000] * 40, sig, period(1, 5000), period(2, 5000), period(3, 50000), period(4, 33000), period(5, 20000),
            period(255, 40000), period(257, 40000), period(4000, 50000), period(16384, 49152 + 7), period(20000, 70000),
            bytes(300) + noise[:300] + bytes(300), sig[:16384] + sig[:16384] + sig[:100]]
    outs = _solo_zlib(bufs)
    for i, (b, z) in enumerate(zip(bufs, outs)):
        assert zlib.decompress(z) == b, i
        ref = len(zlib.compress(b, 6))
        far = b[:20000] * 3 == b[:60000] and len(b) == 70000     # period 20000: beyond the window of some positions (previous block + current)
        if len(b) >= 5000 and not far:
            assert len(z) <= 1.10 * ref + 64 + 48 * (len(b) // 16384 + 1), (i, len(b), len(z), ref)   # never far from zlib (+ a block header per 16 KiB)
    # the same streams come back through both GPU inflate kernels
    from slow5tools_amd import _lib

    for par, thr in ((1, 24576), (0, 1), (0, 1 << 30)):
        _lib.check(_lib.lib().s5gpu_set_option(b"inflate_par", par))
        _lib.check(_lib.lib().s5gpu_set_option(b"inflate_simt_min", thr))
        import ctypes as C
        L = _lib.lib(); n = len(outs); vp = C.c_void_p
        keep = [C.create_string_buffer(b, max(len(b), 1)) for b in outs]
        inp = (vp * n)(*[C.addressof(k) for k in keep]); il = (C.c_size_t * n)(*[len(b) for b in outs])
        out = (vp * n)(); ol = (C.c_size_t * n)(); st = (C.c_int32 * n)()
        _lib.check(L.s5gpu_solo_batch(1, n, inp, il, out, ol, st), "solo inflate")
        libc = C.CDLL(None); libc.free.argtypes = [vp]
        for i in range(n):
            assert C.string_at(out[i], ol[i]) == bufs[i], i
            libc.free(out[i])
    _lib.check(_lib.lib().s5gpu_set_option(b"inflate_simt_min", 24576))
    _lib.check(_lib.lib().s5gpu_set_option(b"inflate_par", 1))


def test_lz77_short_record_shape_builds_its_payloads_and_shares_batches_with_the_long_shape(press):
    """round 3: raw-signal records whose payload fits one 8 KiB block take LzShort (38 KiB of LDS, the payload put together inside the kernel);
    a batch that also holds longer records runs both shapes, each on its share.  Ids of odd and even length (the samples then sit 1- or
    2-byte aligned in the payload), aux tails, empty and one-sample reads, a record of exactly 8192 payload bytes and one of 8193: stock
    zlib must give back the oracle's payloads, and a short record must come out the same whether or not long ones share its batch"""
    rng = np.random.default_rng(77)
    def mk(n, k, aux=b""):
        sig = ob.synth_read(0x5105, 3000 + k, n)
        rid = b"r" * (5 + k % 7)
        return sig, press.pack_hdr(rid, k % 3, 8192.0, 23.0, 1467.61, 4000.0), aux
    short = [mk(int(rng.integers(0, 4000)), k, bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))) for k in range(120)]
    short += [mk(0, 200), mk(1, 201), mk(2, 202)]
    hl = len(press.pack_hdr(b"r" * 6, 0, 1.0, 1.0, 1.0, 1.0))                     # 2 + 6 + 36
    short.append(mk((8192 - hl - 8) // 2, 1))                                   # payload of exactly 8192 bytes (id of 6 characters)
    longer = [mk((8192 - hl - 8) // 2 + 1, 1), mk(30000, 300), mk(9000, 301)]
    def run(items):
        recs = press.encode_records([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], press.REC_ZLIB, press.SIG_NONE)
        for (sig, hdr, aux), rec in zip(items, recs):
            assert int.from_bytes(rec[:8], "little") == len(rec) - 8
            assert zlib.decompress(rec[8:]) == hdr + struct.pack("<Q", sig.size) + sig.tobytes() + aux
        return recs
    a = run(short)
    b = run(short[:60] + longer + short[60:])
    assert a == b[:60] + b[63:]
    z6 = sum(len(zlib.compress(i[1] + struct.pack("<Q", i[0].size) + i[0].tobytes() + i[2], 6)) + 8 for i in short)
    assert sum(map(len, a)) <= 1.04 * z6


def test_lz77_matcher_output_is_deterministic_and_independent_of_the_batch(press):
    rng = np.random.default_rng(13)
    sigs = [(500 + 25 * rng.standard_normal(int(n))).astype(np.int16) for n in rng.integers(100, 30000, 40)]
    hdrs = [press.pack_hdr(b"read-%d" % i, 0, 8192.0, 23.0, 1467.61, 4000.0) for i in range(len(sigs))]
    a = press.encode_records(sigs, hdrs, None, press.REC_ZLIB, press.SIG_NONE)
    b = press.encode_records(sigs, hdrs, None, press.REC_ZLIB, press.SIG_NONE)
    c = press.encode_records(sigs[::-1], hdrs[::-1], None, press.REC_ZLIB, press.SIG_NONE)[::-1]
    assert a == b == c
    for rec, s, h in zip(a, sigs, hdrs):
        pay = zlib.decompress(rec[8:])
        assert pay == h + struct.pack("<Q", s.size) + s.tobytes()


def test_lz77_long_shape_is_deterministic_under_bucket_collisions(press):
    """round 4: the long shape runs 16 waves; four of them share each way of the matcher's table and write in turn, and the current round's
    earliest position per hash comes from an LDS atomic min — neither may let the waves' timing show.  Signals of a handful of distinct values
    (nearly every four bytes hash to one of a few buckets: dozens of writers per bucket and round), plain periodic ones and noise, 40 k - 150 k
    samples each, encoded four times in two orders: byte-identical records every time, each inflated by stock zlib to its payload"""
    rng = np.random.default_rng(131)
    sigs = []
    for i, n in enumerate(rng.integers(40000, 150000, 24)):
        n = int(n)
        if i % 4 == 0: s = 500 + 2 * rng.integers(0, 3, n)                                  # three values
        elif i % 4 == 1: s = np.tile(400 + rng.integers(-50, 50, 509), n // 509 + 1)[:n]    # period 1018 bytes: inside one round of 1024 positions
        elif i % 4 == 2: s = 500 + (np.arange(n) // 7) % 5                                  # runs of seven, period 35
        else: s = 500 + 25 * rng.standard_normal(n)
        sigs.append(np.asarray(s).astype(np.int16))
    hdrs = [press.pack_hdr(b"read-%d" % i, 0, 8192.0, 23.0, 1467.61, 4000.0) for i in range(len(sigs))]
    runs = [press.encode_records(sigs, hdrs, None, press.REC_ZLIB, press.SIG_NONE) for _ in range(3)]
    runs.append(press.encode_records(sigs[::-1], hdrs[::-1], None, press.REC_ZLIB, press.SIG_NONE)[::-1])
    assert runs[0] == runs[1] == runs[2] == runs[3]
    for i, (rec, s, h) in enumerate(zip(runs[0], sigs, hdrs)):
        pay = h + struct.pack("<Q", s.size) + s.tobytes()
        assert zlib.decompress(rec[8:]) == pay
        # size: the periodic and the noisy ones stay near zlib level 6 (+ a block header per 16 KiB); the few-valued ones are where a greedy matcher
        # with four recent positions per bucket loses to zlib's 128-deep chains (1.5 x on three random values): determinism and content only
        if i % 4 in (1, 3):
            ref = len(zlib.compress(pay, 6))
            assert len(rec) <= 1.10 * ref + 64 + 48 * (len(pay) // 16384 + 1), (i, len(rec), ref)


def test_no_payload_decode_keeps_short_records_in_lds_and_hands_the_rest_to_the_slot_decoder(press):
    """S5GPU_DEC_NO_PAYLOAD with max_in_len naming records of one inflate window (round 6: k_inflate_par_np_lp — the record is inflated into
    LDS, the output pass reads its bits out of global memory, the unpack reads LDS; the uncompressed record never reaches HBM).  One batch
    with everything that kernel takes (our own records and stock zlib's at levels 1 / 6 / 9, Z_FIXED, tiny reads) and everything it has
    to DECLINE before writing a byte — a stored block, a stream of two blocks, a record longer than a window, a payload over the cap —
    plus damaged records: statuses and signals equal the slot form's (option np_lds_payload = 0), and every record that fits decodes."""
    from slow5tools_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(606)
    recs, sigs, want_ok = [], [], []

    def add(sig, stream, ok=True):
        sigs.append(sig); recs.append(stream); want_ok.append(ok)

    own_n = [4000, 4000, 3999, 4001, 2000, 513, 64, 5, 1, 4100]
    own_sig = [ob.synth_read(0x5105, 40 + i, n) for i, n in enumerate(own_n)]
    own = press.encode_records(own_sig, [_hdr(press, 40 + i) for i in range(len(own_n))])
    for sg, r in zip(own_sig, own):
        add(sg, r[8:])
    for i, level in enumerate((1, 6, 9, 6, 6)):
        sg = ob.synth_read(0x5105, 80 + i, 4000 - 7 * i)
        pay, _ = _oracle_payload(_hdr(press, 80 + i), sg, b"", 1)
        add(sg, zlib.compress(pay, level))
    sg = ob.synth_read(0x5105, 90, 3000)
    pay, _ = _oracle_payload(_hdr(press, 90), sg, b"", 1)
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    add(sg, c.compress(pay) + c.flush())                                     # fixed codes: taken
    add(sg, zlib.compress(pay, 0))                                           # stored block: declined -> slot decoder
    c = zlib.compressobj(6)
    add(sg, c.compress(pay[:1500]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(pay[1500:]) + c.flush())   # several blocks: declined
    sg = rng.integers(-32768, 32768, 2050).astype(np.int16)                   # ~2.5 bytes per sample: 5.2 KB of payload in ~4.9 KB of stream -> two windows: declined
    pay, _ = _oracle_payload(_hdr(press, 91), sg, b"", 1)
    add(sg, zlib.compress(pay, 6))
    sg = ob.synth_read(0x5105, 92, 6000)                                      # payload over the cap: status 5 (too small a slot) on both forms
    pay, _ = _oracle_payload(_hdr(press, 92), sg, b"", 1)
    add(sg, zlib.compress(pay, 6), ok=False)
    bad = bytearray(own[0][8:]); bad[len(bad) // 2] ^= 0x10
    add(own_sig[0], bytes(bad), ok=False)                                     # damaged in the middle
    add(own_sig[1], own[1][8:-3], ok=False)                                   # truncated
    add(own_sig[2], own[2][8:-1] + bytes([own[2][-1] ^ 1]), ok=False)         # wrong Adler-32
    cap = 5344
    res = {}
    for lds in (1, 0):
        _lib.check(L.s5gpu_set_option(b"np_lds_payload", lds))
        try:
            # (max_in_len is a hint: 4000 names the LDS kernel although two records of the batch are longer — they must be declined, not mangled)
            res[lds] = press.decode_signals_dev(recs, max_pay_cap=cap, sig_caps=[max(len(s_), 8) for s_ in sigs], max_in_len=4000)
        finally:
            _lib.check(L.s5gpu_set_option(b"np_lds_payload", 1))
    f1, s1 = res[1]
    f0, s0 = res[0]
    assert list(f1["status"]) == list(f0["status"])
    for i, ok in enumerate(want_ok):
        if ok:
            assert f1["status"][i] == 0, (i, int(f1["status"][i]))
            assert np.array_equal(s1[i], sigs[i]) and np.array_equal(s0[i], sigs[i]), i
            assert f1["n_samples"][i] == len(sigs[i]) and f1["read_group"][i] == f0["read_group"][i] and f1["aux_len"][i] == f0["aux_len"][i]
        else:
            assert f1["status"][i] != 0, i
    # a batch of a thousand copies (several records per persistent workgroup: the window's storage is reused record after record)
    many = [recs[i % 15] for i in range(3000)]
    f, sg_out = press.decode_signals_dev(many, max_pay_cap=cap, sig_caps=[max(len(sigs[i % 15]), 8) for i in range(3000)])
    assert (f["status"] == 0).all()
    for i in range(0, 3000, 7):
        assert np.array_equal(sg_out[i], sigs[i % 15])


def test_fused_unpack_and_the_records_it_leaves_to_the_second_kernel(press):
    """s5gpu_decode_dev on zlib + svb-zd records: the wave that inflates a record also unpacks it (k_inflate_par<true>); records the
    parallel decoder declines (periodic signals: their svb bytes are far matches for stock zlib) are inflated by the fallback
    decoder and unpacked by k_unpack_rest.  One batch with both kinds + malformed payloads + a signal slot that is too small;
    identical answers with the fusion switched off"""
    from slow5tools_amd import _lib
    rng = np.random.default_rng(91)
    sigs, streams = [], []
    for i in range(96):
        n = int(rng.integers(1, 9000))
        if i % 3 == 0:
            sig = np.tile((400 + rng.integers(-300, 300, 53)).astype(np.int16), n // 53 + 1)[:n]     # periodic: far matches
        elif i % 3 == 1:
            sig = ob.synth_read(0x5105, i, n)
        else:
            sig = (rng.integers(-2000, 2000, n)).astype(np.int16)
        payload, _ = _oracle_payload(_hdr(press, i), sig, bytes(rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8)), 1)
        sigs.append(sig)
        streams.append(zlib.compress(payload, 9 if i % 2 else 6))
    # a payload whose svb-zd length field lies, and one that is cut short: malformed records (status 7) on either path
    good, _ = _oracle_payload(_hdr(press, 500), ob.synth_read(1, 1, 500), b"", 1)
    hl = 2 + int.from_bytes(good[:2], "little") + 4 + 32
    lying = good[:hl] + (int.from_bytes(good[hl:hl + 8], "little") + 9).to_bytes(8, "little") + good[hl + 8:]
    streams += [zlib.compress(lying), zlib.compress(good[:hl + 4])]
    res = {}
    for fused in (1, 0):
        _lib.check(_lib.lib().s5gpu_set_option(b"unpack_fused", fused))
        try:
            res[fused] = press.decode_records(streams, raise_on_error=False)
        finally:
            _lib.check(_lib.lib().s5gpu_set_option(b"unpack_fused", 1))
    for a, b in zip(res[1], res[0]):
        assert a["status"] == b["status"]
        if a["status"] == 0:
            assert np.array_equal(a["signal"], b["signal"]) and a["aux"] == b["aux"] and a["read_id"] == b["read_id"] and a["payload"] == b["payload"]
    for g, s in zip(res[1], sigs):
        assert g["status"] == 0 and np.array_equal(g["signal"], s)
    assert res[1][-2]["status"] == 7 and res[1][-1]["status"] == 7
    # some of the periodic records really took the second kernel: the parallel decoder alone declines them
    import ctypes as C
    L = _lib.lib()
    _lib.check(L.s5gpu_set_option(b"inflate_par", 2))       # tools mode: no fallback pass, declined records keep status 8
    try:
        alone = press.decode_records(streams[:96], raise_on_error=False)
    finally:
        _lib.check(L.s5gpu_set_option(b"inflate_par", 1))
    assert any(g["status"] == 8 for g in alone)


def test_big_mixed_batch_is_launched_longest_first_and_decodes_the_same(press):
    """batches of >= 8192 zlib records are counting-sorted by compressed length on the device and launched longest first (one wave per
    record: the batch ends with its longest one); the order must not show in the results — full form, no-payload form and file order
    (option order_min = 0) agree record for record on 9000 reads of 1 .. 60000 samples"""
    from slow5tools_amd import _lib
    rng = np.random.default_rng(41)
    n_rec = 9000
    lens = np.clip(np.exp(rng.normal(np.log(1500), 1.1, n_rec)), 1, 60000).astype(np.int64)
    lens[:4] = (60000, 1, 2, 45000)
    sigs = [(500 + rng.integers(-60, 60, int(n))).astype(np.int16) for n in lens]
    hdrs = [_hdr(press, i) for i in range(n_rec)]
    recs = [r[8:] for r in press.encode_records(sigs, hdrs)]
    f_np, s_np = press.decode_signals_dev(recs, press.REC_ZLIB, max_pay_cap=int(lens.max()) * 13 // 4 + 512, sig_caps=[int(n) + 8 for n in lens])
    assert (f_np["status"] == 0).all()
    L = _lib.lib()
    full = press.decode_records(recs)
    _lib.check(L.s5gpu_set_option(b"order_min", 0))
    try:
        f_fo, s_fo = press.decode_signals_dev(recs, press.REC_ZLIB, max_pay_cap=int(lens.max()) * 13 // 4 + 512, sig_caps=[int(n) + 8 for n in lens])
    finally:
        _lib.check(L.s5gpu_set_option(b"order_min", 8192))
    for i in range(n_rec):
        assert np.array_equal(s_np[i], sigs[i]) and np.array_equal(s_fo[i], sigs[i]) and np.array_equal(full[i]["signal"], sigs[i]), i
        assert int(f_np["read_group"][i]) == int(f_fo["read_group"][i]) == full[i]["read_group"]


@pytest.mark.parametrize("scratch", ["default", "three-slots"])
def test_decode_without_payload_output_equals_the_full_decode(press, scratch):
    """S5GPU_DEC_NO_PAYLOAD (fields + signals only; persistent workgroups inflate into a reused scratch slot and unpack out of it):
    the same mixed batch as above — records the parallel decoder declines (its fallback has its own slots), malformed payloads, stock
    zlib levels — must give the fields and signals of the full decode; with three slots of scratch as with thousands.  A record larger
    than max_pay_cap reports status 5 and the size it needs; a signal slot that is too small status 6."""
    rng = np.random.default_rng(92)
    sigs, streams = [], []
    for i in range(200):
        n = int(rng.integers(1, 9000))
        if i % 3 == 0:
            sig = np.tile((400 + rng.integers(-300, 300, 53)).astype(np.int16), n // 53 + 1)[:n]
        elif i % 3 == 1:
            sig = ob.synth_read(0x5105, i, n)
        else:
            sig = (rng.integers(-2000, 2000, n)).astype(np.int16)
        payload, _ = _oracle_payload(_hdr(press, i), sig, bytes(rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8)), 1)
        sigs.append(sig)
        streams.append(zlib.compress(payload, 9 if i % 2 else 6))
    good, _ = _oracle_payload(_hdr(press, 500), ob.synth_read(1, 1, 500), b"", 1)
    hl = 2 + int.from_bytes(good[:2], "little") + 4 + 32
    lying = good[:hl] + (int.from_bytes(good[hl:hl + 8], "little") + 9).to_bytes(8, "little") + good[hl + 8:]
    big, _ = _oracle_payload(_hdr(press, 501), ob.synth_read(1, 2, 30000), b"", 1)          # larger than max_pay_cap below
    streams += [zlib.compress(lying), zlib.compress(good[:hl + 4]), zlib.compress(big), streams[4][:30] + b"\x55" + streams[4][31:]]
    own = press.encode_records([ob.synth_read(0x5105, 900 + i, 4000) for i in range(64)], [_hdr(press, 900 + i) for i in range(64)])
    streams += [r[8:] for r in own]
    sigs_all = sigs + [None] * 4 + [ob.synth_read(0x5105, 900 + i, 4000) for i in range(64)]
    cap = 9000 * 3 + 300
    caps = np.full(len(streams), 9000)
    caps[7] = 10                                                                             # signal slot too small
    f, got = press.decode_signals_dev(streams, 1, max_pay_cap=cap, sig_caps=caps, scratch_bytes=None if scratch == "default" else 64 + 3 * (cap + 32))
    ref = press.decode_records(streams, raise_on_error=False)
    for i, (g, r) in enumerate(zip(got, ref)):
        if i == 7:
            assert f["status"][i] == 6 and f["n_samples"][i] == len(sigs[7])
        elif i == 202:
            assert f["status"][i] == 5 and f["payload_len"][i] == len(big)
        else:
            assert f["status"][i] == r["status"], (i, f["status"][i], r["status"])
            if r["status"] == 0:
                assert np.array_equal(g, r["signal"]) and np.array_equal(g, sigs_all[i])
                assert f["read_group"][i] == r["read_group"] and f["payload_len"][i] == len(r["payload"]) and f["aux_len"][i] == len(r["aux"]) and f["digitisation"][i] == r["digitisation"]
    assert f["status"][200] == 7 and f["status"][201] == 7 and f["status"][203] != 0
    # zstd records through the same mode; and the reference's own files (stock zlib)
    zrec = press.encode_records(sigs[:40], [_hdr(press, i) for i in range(40)], None, 2, 1)
    f, got = press.decode_signals_dev([r[8:] for r in zrec], 2, max_pay_cap=cap)
    assert (f["status"] == 0).all() and all(np.array_equal(g, s) for g, s in zip(got, sigs[:40]))
    for name in ZLIB_SVB_FIXTURES:
        fx = Blow5(golden(name))
        want = press.decode_records(fx.records, 1, 1)
        f, got = press.decode_signals_dev(fx.records, 1, max_pay_cap=max(len(w["payload"]) for w in want) + 8)
        assert (f["status"] == 0).all() and all(np.array_equal(g, w["signal"]) for g, w in zip(got, want)), name


def test_parallel_inflate_takes_stock_zlib_streams_without_the_fallback(press):
    """Records written by stock zlib (the reference's encoder) hold ~110 matches per 4000-sample svb-zd record whose source is not
    the decoding lane's own output, several hundred per window in raw-signal records: with inflate_par = 2 (no fallback pass: a
    declined record keeps status 8) every golden zlib fixture and zlib-1/6/9 streams of synthetic records must still decode,
    bit-exactly"""
    from slow5tools_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(17)
    streams, want, sms = [], [], []
    for i in range(48):
        n = int(rng.integers(50, 30000))
        sig = ob.synth_read(0x5105, 1000 + i, n)
        sm = 1 if i % 4 else 0
        payload, _ = _oracle_payload(_hdr(press, i), sig, b"", sm)
        streams.append(zlib.compress(payload, (1, 6, 9)[i % 3])); want.append(sig); sms.append(sm)
    _lib.check(L.s5gpu_set_option(b"inflate_par", 2))
    try:
        for sm in (0, 1):
            idx = [i for i in range(len(streams)) if sms[i] == sm]
            got = press.decode_records([streams[i] for i in idx], 1, sm, raise_on_error=False)
            for g, i in zip(got, idx):
                assert g["status"] == 0 and np.array_equal(g["signal"], want[i])
        for name in ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES:
            f = Blow5(golden(name))
            got = press.decode_records(f.records, f.rec_method, f.sig_method, raise_on_error=False)
            for g, rec in zip(got, f.records):
                assert g["status"] == 0 and g["payload"] == zlib.decompress(rec), name
    finally:
        _lib.check(L.s5gpu_set_option(b"inflate_par", 1))


def test_parallel_inflate_on_awkward_zlib_streams(press, inflate_kernel):
    """stock-zlib streams that stress the block / round / waiting-match machinery: a flush every few hundred bytes (dozens of
    blocks and empty stored blocks per window), text-like data (long matches at every distance, dense in matches), runs of every
    length, matches that reach back 32 KiB, levels 1 / 9, Z_RLE / Z_HUFFMAN_ONLY / Z_FIXED; the payload is a raw-signal record so
    that any byte pattern is a valid signal"""
    rng = np.random.default_rng(404)
    def words(n):
        vocab = [bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(200)]
        out = bytearray()
        while len(out) < n:
            out += vocab[int(rng.integers(0, len(vocab)))] + b" "
        return bytes(out[:n])
    blobs = [words(60000), words(3000),
             bytes(np.repeat(rng.integers(0, 256, 4000, dtype=np.uint8), rng.integers(1, 40, 4000))),              # runs of every length
             bytes(rng.integers(0, 256, 40000, dtype=np.uint8)) * 2,                                              # 40 000-byte period: > 32 KiB, no matches; then ...
             bytes(rng.integers(0, 256, 30000, dtype=np.uint8)) * 3,                                              # ... 30 000: matches at distance 30 000
             bytes(200000), bytes([1, 2, 3]) * 30000, words(500) * 100]
    streams, sigs = [], []
    for i, blob in enumerate(blobs):
        blob = blob[: len(blob) // 2 * 2]
        sig = np.frombuffer(blob, dtype=np.int16).copy()
        payload, _ = _oracle_payload(_hdr(press, i), sig, b"", 0)
        variants = [zlib.compress(payload, 1), zlib.compress(payload, 9)]
        for strat in (zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED, zlib.Z_FILTERED):
            c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
            variants.append(c.compress(payload) + c.flush())
        c = zlib.compressobj(6)
        parts, step = [], int(rng.integers(150, 900))
        for k in range(0, len(payload), step):
            parts.append(c.compress(payload[k:k + step]))
            parts.append(c.flush(zlib.Z_SYNC_FLUSH if (k // step) % 3 else zlib.Z_FULL_FLUSH))
        parts.append(c.flush())
        variants.append(b"".join(parts))
        for v in variants:
            assert zlib.decompress(v) == payload
            streams.append(v); sigs.append(sig)
    got = press.decode_records(streams, 1, 0)
    for g, s in zip(got, sigs):
        assert g["status"] == 0 and np.array_equal(g["signal"], s)


def test_waiting_matches_copy_in_dependency_order(press, inflate_kernel):
    """matches whose source is another waiting match's output, in chains of every shape: each waits for exactly the entries that reach into its
    source (two binary searches over the list, inflate_par_dev.h), not for everything in front of it — phrases that quote the phrase before
    (chains as long as the list), quotes of quotes at shrinking lengths (one source spanning several earlier destinations), a quote that
    overlaps its own output, far quotes between near ones (entries that may overtake their neighbours), at zlib levels 1 - 9; the payload is
    a raw-signal record so that any byte pattern is a valid signal (reference decode: /root/reference/src/view.c + zlib inflate)"""
    rng = np.random.default_rng(2024)
    def rnd(n):
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    def chain(total, lo, hi):                          # every phrase = the tail of what precedes it + a few fresh bytes
        out = bytearray(rnd(64))
        while len(out) < total:
            k = min(int(rng.integers(lo, hi)), len(out))
            back = int(rng.integers(k, min(len(out), 4 * hi) + 1))
            out += out[len(out) - back: len(out) - back + k] + rnd(int(rng.integers(1, 4)))
        return bytes(out[:total])
    def nested(total):                                 # quotes of quotes: a source that spans several earlier quotes and the literals between
        out = bytearray(rnd(200))
        while len(out) < total:
            for k in (37, 23, 11, 7, 5, 3):
                out += out[-(k + 9): -9] + rnd(1)
            out += out[-150:-20]                       # one long quote across all of them
        return bytes(out[:total])
    def far_and_near(total):
        base = rnd(3000)
        out = bytearray(base)
        while len(out) < total:
            p = int(rng.integers(0, 2900))
            out += base[p: p + int(rng.integers(4, 60))]          # far: literals of the first window
            out += out[-int(rng.integers(3, 12)):] * 2            # near: overlaps what was just written
            out += out[-40:-30] + rnd(2)
        return bytes(out[:total])
    blobs = [chain(9000, 3, 9), chain(9000, 3, 40), chain(60000, 5, 258), nested(9000), nested(70000), far_and_near(9000), far_and_near(40000),
             (b"ab" * 3 + b"c") * 2000, chain(3000, 3, 5) * 20]
    streams, sigs = [], []
    for i, blob in enumerate(blobs):
        blob = blob[: len(blob) // 2 * 2]
        sig = np.frombuffer(blob, dtype=np.int16).copy()
        payload, _ = _oracle_payload(_hdr(press, i), sig, b"", 0)
        for level in range(1, 10):
            v = zlib.compress(payload, level)
            streams.append(v); sigs.append(sig)
        c = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FILTERED)
        streams.append(c.compress(payload) + c.flush()); sigs.append(sig)
    got = press.decode_records(streams, 1, 0)
    for k, (g, s) in enumerate(zip(got, sigs)):
        assert g["status"] == 0 and np.array_equal(g["signal"], s), (k // 10, k % 10)
    if inflate_kernel == "parallel-in-record":
        # and without the fallback pass behind it: what the parallel decoder does not decline (status 8: a segment over the list's capacity)
        # it must decode right by itself — and it takes the ones whose matches are not too dense for the list (37 of 90 when this was written)
        from slow5tools_amd import _lib
        L = _lib.lib()
        _lib.check(L.s5gpu_set_option(b"inflate_par", 2))
        try:
            got = press.decode_records(streams, 1, 0, raise_on_error=False)
        finally:
            _lib.check(L.s5gpu_set_option(b"inflate_par", 1))
        taken = 0
        for k, (g, s) in enumerate(zip(got, sigs)):
            assert g["status"] in (0, 8), (k, g["status"])
            if g["status"] == 0:
                taken += 1
                assert np.array_equal(g["signal"], s), (k // 10, k % 10)
        assert taken >= len(streams) // 4, taken


def test_mixed_batch_overflow_list_is_launched_longest_first_and_encodes_the_same(press):
    """encode side of the launch order (round 4): a mixed batch with an LDS budget sends its long reads through the overflow list, which is
    counting-sorted by read length on the device (longest first) for batches of >= 8192 reads; the order must not show in the output —
    every record byte-identical to the file-order run (option order_min = 0), every sampled record inflated by stock zlib to the oracle's
    payload"""
    from slow5tools_amd import _lib
    import torch
    rng = np.random.default_rng(77)
    n_rec = 9000
    ns = np.clip(np.exp(rng.normal(np.log(5000), 1.0, n_rec)), 1, 90000).astype(np.uint64)
    ns[:3] = (90000, 1, 70000)
    sigs = [(520 + rng.integers(-70, 70, int(n))).astype(np.int16) for n in ns]
    hdrs = [_hdr(press, i) for i in range(n_rec)]
    L = _lib.lib()
    out = {}
    for order_min in (8192, 0):
        _lib.check(L.s5gpu_set_option(b"order_min", order_min))
        try:
            b = press.DeviceBatch(ns, hdr_len=[len(h) for h in hdrs], lds_payload_cap=8192)
            b.upload(sigs, hdrs)
            b.encode()
            b.compact()
            stream, off = b.stream_bytes()
            assert int(b.ovf[0].item()) > 1000          # the list really carries the long reads
            out[order_min] = (stream, off.copy())
            del b
            torch.cuda.empty_cache()
        finally:
            _lib.check(L.s5gpu_set_option(b"order_min", 8192))
    assert out[8192][0] == out[0][0] and np.array_equal(out[8192][1], out[0][1])
    stream, off = out[8192]
    for i in list(range(0, n_rec, 450)) + [0, 1, 2]:
        rec = stream[int(off[i]):int(off[i + 1])]
        want, _ = _oracle_payload(hdrs[i], sigs[i], b"", 1)
        assert zlib.decompress(rec[8:]) == want, i


def test_mixed_batch_second_fused_launch_takes_the_reads_in_between(press):
    """round 4, option fused_tier2 (off by default: it bought nothing on the mixed leg): a batch with a named LDS budget gets TWO fused launches
    — the named budget (8 KiB, eight workgroups per CU) and then up to one 16 KiB DEFLATE block for the reads in between; only what fits neither is staged.  Every record must inflate to the
    oracle's payload whichever launch made it, the overflow list must shrink to the long reads, and the records of reads neither launch
    touches differently (short ones, long ones) must be byte-identical to the one-launch run"""
    from slow5tools_amd import _lib
    import torch
    rng = np.random.default_rng(91)
    n_rec = 3000
    ns = np.clip(np.exp(rng.normal(np.log(6000), 0.9, n_rec)), 1, 120000).astype(np.uint64)
    ns[:6] = (120000, 1, 6300, 6500, 12400, 12900)               # around both budgets' edges
    sigs = [(520 + np.cumsum(rng.integers(-9, 10, int(n))) % 300).astype(np.int16) for n in ns]
    sigs[7] = rng.integers(-32768, 32768, int(ns[7]), dtype=np.int16)   # ~3 bytes per sample: overflows wherever its length would fit
    hdrs = [_hdr(press, i) for i in range(n_rec)]
    L = _lib.lib()
    out = {}
    for tier2 in (16384, 12288, 0):
        _lib.check(L.s5gpu_set_option(b"fused_tier2", tier2))
        try:
            b = press.DeviceBatch(ns, hdr_len=[len(h) for h in hdrs], lds_payload_cap=8192)
            b.upload(sigs, hdrs)
            b.encode()
            b.compact()
            stream, off = b.stream_bytes()
            out[tier2] = (stream, off.copy(), int(b.ovf[0].item()))
            del b
            torch.cuda.empty_cache()
        finally:
            _lib.check(L.s5gpu_set_option(b"fused_tier2", 0))
    assert out[16384][2] < out[12288][2] < out[0][2], [out[k][2] for k in out]
    n_long = int((ns > 13500).sum())
    assert n_long <= out[16384][2] <= n_long + int(((ns > 9000) & (ns <= 13500)).sum())
    for tier2 in (16384, 12288):
        stream, off, _ = out[tier2]
        for i in range(n_rec):
            rec = stream[int(off[i]):int(off[i + 1])]
            one = out[0][0][int(out[0][1][i]):int(out[0][1][i + 1])]
            if (ns[i] < 5000 or ns[i] > 17000) and i != 7:
                assert rec == one, i
            if i < 64 or i % 37 == 0 or rec != one:
                want, _ = _oracle_payload(hdrs[i], sigs[i], b"", 1)
                assert zlib.decompress(rec[8:]) == want, (tier2, i)
        assert len(stream) <= 1.002 * len(out[0][0])
