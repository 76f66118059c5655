#!/usr/bin/env python3
"""Builds tests/golden/ref/ + tests/golden/manifest.json.  Run in the BUILD container (where /root/reference exists):

    python tests/golden/make_manifest.py

1. copies every `.blow5` the reference's tests hold (/root/reference/test/data/**, 74 files, 12 MB; data files,
   not source) to tests/golden/ref/<same relative path>;
2. walks each with the CPU oracle only (stock zlib / libzstd through oracle/libs5oracle.so + oracle/rec.c) and
   records, per file: sha256, version, press codes, header sha256; per record: compressed length, sha256 of the
   uncompressed payload, of the int16 signal, of the aux bytes, the primary fields (doubles as their 8 bytes in
   hex) — what `slow5_rec_depress_parse` must hand back (/root/reference/src/view.c:38, test/test_view.sh:90-165);
3. files the reference's own tests expect readers to REJECT (test/test_quickcheck.sh) are listed with
   `"negative": true` and the reason this walker found.

Nothing here is read on the GPU box: the `-m gpu` sweep (tests/test_reference_fixtures.py) compares the HIP
decoders with this manifest and the committed copies.
"""
import hashlib
import json
import os
import shutil
import struct
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_bind as ob  # noqa: E402

REF = "/root/reference/test/data"
OUT = os.path.join(HERE, "ref")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def walk(path):
    b = open(path, "rb").read()
    ent = dict(bytes=len(b), sha256=sha(b))
    if b[:6] != b"BLOW5\x01":
        return dict(ent, negative=True, why="bad magic")
    ent.update(version=list(b[6:9]), rec_method=b[9], num_read_groups=struct.unpack_from("<I", b, 10)[0], sig_method=b[14])
    (hl,) = struct.unpack_from("<I", b, 64)
    if 68 + hl > len(b):
        return dict(ent, negative=True, why="header length beyond the file")
    ent["header_sha256"] = sha(b[68:68 + hl])
    off = 68 + hl
    recs = []
    while True:
        if b[off:off + 5] == b"5WOLB" and off + 5 == len(b):
            break
        if off + 8 > len(b):
            return dict(ent, negative=True, why="no end-of-file marker", n_records_before=len(recs))
        (sz,) = struct.unpack_from("<Q", b, off)
        if off + 8 + sz > len(b):
            return dict(ent, negative=True, why="record beyond the file", n_records_before=len(recs))
        body = b[off + 8:off + 8 + sz]
        try:
            if ent["rec_method"] == 1:
                pl = zlib.decompress(body)
            elif ent["rec_method"] == 2:
                pl = ob.zstd_decompress(body)
                assert pl is not None
            else:
                pl = body
            d = ob.rec_parse(pl, ent["sig_method"])
        except Exception as e:  # noqa: BLE001 — a negative fixture can fail anywhere
            return dict(ent, negative=True, why="record %d: %s" % (len(recs), e), n_records_before=len(recs))
        recs.append(dict(
            off=off, zlen=sz, payload_len=len(pl), payload_sha256=sha(pl),
            n_samples=int(d["signal"].size), signal_sha256=sha(d["signal"].tobytes()),
            read_id=d["read_id"].decode("latin-1"), read_group=int(d["read_group"]),
            doubles=struct.pack("<dddd", d["digitisation"], d["offset"], d["range"], d["sampling_rate"]).hex(),
            aux_len=len(d["aux"]), aux_sha256=sha(d["aux"])))
        off += 8 + sz
    ent["records"] = recs
    return ent


def main():
    files = []
    for root, _, names in os.walk(REF):
        for n in names:
            if n.endswith(".blow5"):
                files.append(os.path.relpath(os.path.join(root, n), REF))
    files.sort()
    man = {}
    for rel in files:
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
        man[rel] = walk(dst)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
        f.write("\n")
    neg = [k for k, v in man.items() if v.get("negative")]
    nrec = sum(len(v.get("records", ())) for v in man.values())
    big = max((r["n_samples"], k) for k, v in man.items() for r in v.get("records", ()))
    print("%d files (%d negative: %s), %d records, longest %d samples in %s" % (len(man), len(neg), neg, nrec, big[0], big[1]))


if __name__ == "__main__":
    main()
