#!/usr/bin/env python3
"""Run in the build container only (needs /root/reference): every ex-zd record the reference ships (signal-press code 2)
must decode with the ORACLE and re-encode to the identical blob.  The committed tests pin four of these files; this walks all."""
import os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind as ob

base = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/data"
files = recs = exact = with_ex = 0
for dp, _, fs in os.walk(base):
    for f in sorted(fs):
        if not f.endswith(".blow5"):
            continue
        b = open(os.path.join(dp, f), "rb").read()
        if len(b) < 70 or b[:6] != b"BLOW5\x01" or b[14] != 2 or b[9] not in (0, 1):
            continue
        files += 1
        (hl,) = struct.unpack_from("<I", b, 64)
        off = 68 + hl
        while b[off:off + 5] != b"5WOLB":
            (sz,) = struct.unpack_from("<Q", b, off)
            r = b[off + 8: off + 8 + sz]
            off += 8 + sz
            if b[9] == 1:
                r = zlib.decompress(r)
            idl = struct.unpack_from("<H", r, 0)[0]
            p = 2 + idl + 4 + 32
            (L,) = struct.unpack_from("<Q", r, p)
            blob = r[p + 8: p + 8 + L]
            sig = ob.exzd_decode(blob)
            recs += 1
            with_ex += struct.unpack_from("<I", blob, 12)[0] > 0
            exact += sig is not None and ob.exzd_encode(sig) == blob
print("ex-zd files %d, records %d (with exceptions: %d), decode + re-encode bit-exact: %d" % (files, recs, with_ex, exact))
