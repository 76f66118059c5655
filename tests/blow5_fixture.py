"""Minimal BLOW5 / SLOW5 fixture readers for the tests (layout: SURVEY.md Appendix A;
the reference states the same layout in test/misc/make_blow5.c:11-101)."""
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return os.path.join(GOLDEN, name)


class Blow5:
    def __init__(self, path):
        b = open(path, "rb").read()
        assert b[:6] == b"BLOW5\x01", "bad magic"
        self.version = tuple(b[6:9])
        self.rec_method = b[9]
        self.num_read_groups = struct.unpack_from("<I", b, 10)[0]
        self.sig_method = b[14]
        (hl,) = struct.unpack_from("<I", b, 64)
        self.header_text = b[68 : 68 + hl]
        off = 68 + hl
        self.records = []  # raw (possibly compressed) record bytes, without the u64 prefix
        self.offsets = []  # file offset of each record's u64 size prefix
        while not (b[off : off + 5] == b"5WOLB" and off + 5 == len(b)):
            (sz,) = struct.unpack_from("<Q", b, off)
            self.offsets.append(off)
            self.records.append(b[off + 8 : off + 8 + sz])
            off += 8 + sz
        self.raw = b


def read_slow5_ascii(path):
    """returns list of dicts (read_id, read_group, digitisation, offset, range, sampling_rate, signal)"""
    out = []
    for line in open(path, "r"):
        if line[0] in "#@":
            continue
        f = line.rstrip("\n").split("\t")
        out.append(
            dict(
                read_id=f[0].encode(),
                read_group=int(f[1]),
                digitisation=float(f[2]),
                offset=float(f[3]),
                range=float(f[4]),
                sampling_rate=float(f[5]),
                signal=np.array(f[7].split(","), dtype=np.int64).astype(np.int16),
                aux_text=f[8:],
            )
        )
    return out


ZLIB_SVB_FIXTURES = [
    "exp_1_lossless_zlib_svb_v0.2.0.blow5",
    "sp1_dna.blow5",
    "example_multi_rg_v0.2.0.blow5",
    "merged_expected_zlib_svb.blow5",
]
ZLIB_NONE_FIXTURES = ["exp_1_lossless_zlib.blow5", "exp_lossless_gzip.blow5"]
NONE_NONE_FIXTURES = ["exp_1_lossless.blow5", "aux_array_exp_lossless.blow5"]
