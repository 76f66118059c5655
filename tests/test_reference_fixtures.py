"""Every `.blow5` the reference's tests hold (74 files, tests/golden/ref/, copied by tests/golden/make_manifest.py) against
tests/golden/manifest.json — the oracle's reading of each record, written in the build container.

What the reference checks with these files: test/test_view.sh:90-165 (every codec combination, both directions, byte
identical), test/test_degrade.sh:95-96 (the 2 050 027-sample stock-zlib record of raw/degrade/p2solo_ulk114_dna.blow5),
test/test_split.sh:117-119, test/test_merge.sh:131-140, test/test_get.sh, test/test_quickcheck.sh (the files a reader has to
refuse).  Here:

  not gpu : the committed copies are the manifest's files; the oracle re-reads every record to the manifest's hashes; stock
            zlib level 6 re-compresses every zlib record to the reference's bytes and oracle/svbzd.c / oracle/exzd.c
            re-encode every signal to the reference's blob (the pin of SURVEY §8c on ALL files, not the tier-1 set only).
  gpu     : every record of every file through each of the three inflate kernels (+ the zstd decoder, + the uncompressed
            form), through the no-payload form of s5gpu_decode_dev, re-encoded on the device and inflated by the stock
            library, and the container reader of csrc/blow5_file.c over every file incl. the negative ones.
"""
import ctypes as C
import hashlib
import json
import os
import struct
import sys
import zlib

import numpy as np
import pytest

import oracle_bind as ob
from blow5_fixture import GOLDEN, Blow5

sys.path.insert(0, GOLDEN)
import make_manifest  # noqa: E402  (tests/golden/make_manifest.py: its walk() is the manifest's definition)

REF = os.path.join(GOLDEN, "ref")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
GOOD = sorted(k for k, v in MANIFEST.items() if not v.get("negative"))
NEGATIVE = sorted(k for k, v in MANIFEST.items() if v.get("negative"))
ZLIB = [k for k in GOOD if MANIFEST[k]["rec_method"] == 1]
NOT_ZLIB = [k for k in GOOD if MANIFEST[k]["rec_method"] != 1]
# the no-payload form serves zlib / zstd records with svb-zd signals and zlib records with ex-zd signals (include/slow5gpu.h)
NO_PAYLOAD = [k for k in GOOD if (MANIFEST[k]["rec_method"], MANIFEST[k]["sig_method"]) in ((1, 1), (2, 1), (1, 2))]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_the_sweep_covers_the_reference_inventory():
    """74 files, the long-read stress record, all three header versions, every codec pair the reference writes"""
    assert len(MANIFEST) == 74 and len(NEGATIVE) == 3
    assert max(r["n_samples"] for k in GOOD for r in MANIFEST[k]["records"]) == 2_050_027
    pairs = {(tuple(MANIFEST[k]["version"]), MANIFEST[k]["rec_method"], MANIFEST[k]["sig_method"]) for k in GOOD}
    assert {((1, 0, 0), 1, 1), ((1, 1, 0), 1, 1), ((0, 1, 0), 0, 0), ((0, 1, 0), 1, 0), ((0, 2, 0), 2, 1), ((0, 2, 0), 1, 2)} <= pairs


@pytest.mark.parametrize("rel", sorted(MANIFEST))
def test_committed_copy_is_the_manifest_file_and_the_oracle_rereads_it(rel):
    path = os.path.join(REF, rel)
    assert sha(open(path, "rb").read()) == MANIFEST[rel]["sha256"]
    assert make_manifest.walk(path) == MANIFEST[rel]


@pytest.mark.parametrize("rel", GOOD)
def test_oracle_reencodes_every_record_to_the_reference_bytes(rel):
    """zlib level 6 / windowBits 15 / memLevel 8 one-shot, svb-zd and ex-zd bit layouts: pinned by every record of every file"""
    m = MANIFEST[rel]
    if m["rec_method"] == 2:
        pytest.skip("libzstd's compressor is not restated (its decoder is: tests/test_zstd.py); the signal blob is checked through zlib twins")
    f = Blow5(os.path.join(REF, rel))
    for body, r in zip(f.records, m["records"]):
        pl = zlib.decompress(body) if m["rec_method"] == 1 else body
        d = ob.rec_parse(pl, m["sig_method"])
        rec, keep = ob.make_rec(d["read_id"], d["read_group"], d["digitisation"], d["offset"], d["range"], d["sampling_rate"], d["signal"], d["aux"])
        assert ob.rec_pack(rec, m["sig_method"]) == pl
        if m["rec_method"] == 1:
            assert ob.zlib_compress(pl) == body


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def press():
    from slow5tools_amd import _lib, press as p

    _lib.check(_lib.lib().s5gpu_init(0), "s5gpu_init")
    return p


@pytest.fixture
def force_kernel(press):
    from slow5tools_amd import _lib

    L = _lib.lib()

    def force(kind):
        _lib.check(L.s5gpu_set_option(b"inflate_par", 1 if kind == "parallel-in-record" else 0))
        _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 1 if kind == "lane-per-record" else 1 << 30))

    yield force
    _lib.check(L.s5gpu_set_option(b"inflate_simt_min", 24576))
    _lib.check(L.s5gpu_set_option(b"inflate_par", 1))


def _check_decoded(got, m, with_payload=True):
    assert len(got) == len(m["records"])
    for g, r in zip(got, m["records"]):
        assert g["status"] == 0, (r["read_id"], g["status"])
        assert g["signal"].size == r["n_samples"] and sha(g["signal"].tobytes()) == r["signal_sha256"], r["read_id"]
        assert g["read_id"].decode("latin-1") == r["read_id"] and g["read_group"] == r["read_group"]
        assert struct.pack("<dddd", g["digitisation"], g["offset"], g["range"], g["sampling_rate"]).hex() == r["doubles"]
        if with_payload:
            assert len(g["payload"]) == r["payload_len"] and sha(g["payload"]) == r["payload_sha256"]
            assert len(g["aux"]) == r["aux_len"] and sha(g["aux"]) == r["aux_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["wave-per-record", "lane-per-record", "parallel-in-record"])
@pytest.mark.parametrize("rel", ZLIB)
def test_every_zlib_record_of_the_reference_through_each_inflate_kernel(press, force_kernel, rel, kind):
    """records written by stock zlib (real LZ77 distances, 32 KiB window), up to 1.77 MB compressed / 2 050 027 samples"""
    force_kernel(kind)
    m = MANIFEST[rel]
    f = Blow5(os.path.join(REF, rel))
    _check_decoded(press.decode_records(f.records, m["rec_method"], m["sig_method"]), m)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", NOT_ZLIB)
def test_every_zstd_and_uncompressed_record_of_the_reference(press, rel):
    m = MANIFEST[rel]
    f = Blow5(os.path.join(REF, rel))
    _check_decoded(press.decode_records(f.records, m["rec_method"], m["sig_method"]), m)


@pytest.mark.gpu
@pytest.mark.parametrize("rel", NO_PAYLOAD)
def test_every_record_through_the_no_payload_form(press, rel):
    """fields + signals only (what `get` takes, /root/reference/src/get.c:37-66): scratch slots sized by the manifest's payloads"""
    m = MANIFEST[rel]
    f = Blow5(os.path.join(REF, rel))
    cap = max(r["payload_len"] for r in m["records"]) + 64
    fields, sigs = press.decode_signals_dev(f.records, m["rec_method"], max_pay_cap=cap, sig_caps=[r["n_samples"] + 8 for r in m["records"]],
                                            sig_method=m["sig_method"])
    for i, r in enumerate(m["records"]):
        assert fields["status"][i] == 0, (r["read_id"], fields["status"][i])
        assert int(fields["n_samples"][i]) == r["n_samples"] and sha(sigs[i].tobytes()) == r["signal_sha256"], r["read_id"]
        assert int(fields["read_group"][i]) == r["read_group"]
        assert struct.pack("<dddd", *(float(fields[k][i]) for k in ("digitisation", "offset", "range", "sampling_rate"))).hex() == r["doubles"]


@pytest.mark.gpu
@pytest.mark.parametrize("rel", GOOD)
def test_every_record_reencoded_on_the_device_inflates_to_the_reference_payload(press, rel):
    """decode -> slow5_rec_to_mem's batch form with the file's own codecs -> the STOCK library reads the record back to the
    manifest's payload (test/test_view.sh's both-directions diff, with 'valid stream + identical payload' where the
    reference has 'identical bytes': BASELINE north_star)"""
    m = MANIFEST[rel]
    f = Blow5(os.path.join(REF, rel))
    got = press.decode_records(f.records, m["rec_method"], m["sig_method"])
    hdrs = [press.pack_hdr(g["read_id"], g["read_group"], g["digitisation"], g["offset"], g["range"], g["sampling_rate"]) for g in got]
    recs = press.encode_records([g["signal"] for g in got], hdrs, [g["aux"] for g in got], m["rec_method"], m["sig_method"])
    tot = 0
    for rec, r in zip(recs, m["records"]):
        body = rec[8:]
        assert struct.unpack_from("<Q", rec, 0)[0] == len(body)
        if m["rec_method"] == 1:
            pl = zlib.decompress(body)
        elif m["rec_method"] == 2:
            pl = ob.zstd_decompress(body) if ob.zstd_ref() else ob.zstd_restated_decompress(body, r["payload_len"])
        else:
            pl = body
        assert pl is not None and sha(pl) == r["payload_sha256"], r["read_id"]
        tot += len(body)
    if m["rec_method"] == 1:
        # size against the reference's own records: svb-zd / ex-zd payloads within 2 %, raw int16 payloads (LZ77 matcher) within 3 %
        assert tot <= (1.03 if m["sig_method"] == 0 else 1.02) * sum(r["zlen"] for r in m["records"]), rel


class _File(C.Structure):
    pass


@pytest.mark.gpu
@pytest.mark.parametrize("rel", sorted(MANIFEST))
def test_container_reader_walks_every_file_and_refuses_the_negative_ones(press, rel):
    """slow5_open + slow5_get_next_bytes (csrc/blow5_file.c; /root/reference/src/view.c:266, src/skim.c:385): record count, offsets
    and sizes of the manifest, SLOW5_ERR_EOF at the marker; the files test/test_quickcheck.sh expects to fail do not reach it"""
    from slow5tools_amd import _lib

    L = _lib.lib()
    L.slow5_open.restype = C.c_void_p
    L.slow5_open.argtypes = [C.c_char_p, C.c_char_p]
    L.slow5_close.argtypes = [C.c_void_p]
    L.slow5_get_next_bytes.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    m = MANIFEST[rel]
    s = L.slow5_open(os.path.join(REF, rel).encode(), b"r")
    if not s:
        assert m.get("negative"), rel
        return
    sizes = []
    while True:
        mem, n = C.c_void_p(), C.c_size_t()
        rc = L.slow5_get_next_bytes(C.byref(mem), C.byref(n), s)
        if rc != 0:
            break
        sizes.append(n.value)
        libc.free(mem)
    L.slow5_close(s)
    if m.get("negative"):
        assert rc != -1 and len(sizes) == m.get("n_records_before", 0), (rel, rc, len(sizes))   # anything but SLOW5_ERR_EOF
    else:
        assert rc == -1 and sizes == [r["zlen"] for r in m["records"]], (rel, rc)
