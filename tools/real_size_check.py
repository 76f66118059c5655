"""Record sizes on the reference's own reads (tests/golden/ref, every distinct record): the HIP zlib + svb-zd encoder against zlib level 6 (the oracle)
on the same payloads.  python tools/real_size_check.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
import oracle_bind as ob
from shared_code_study import real_records
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
seen, sigs, hdrs = set(), [], []
for path, rid, sig in real_records():
    key = (bytes(rid), sig.size)
    if key in seen or sig.size == 0: continue
    seen.add(key); sigs.append(np.ascontiguousarray(sig)); hdrs.append(press.pack_hdr(bytes(rid), 0, 8192.0, 23.0, 1467.61, 4000.0))
out = press.encode_records(sigs, hdrs, None, press.REC_ZLIB, press.SIG_SVB_ZD)
raw = press.encode_records(sigs, hdrs, None, press.REC_NONE, press.SIG_SVB_ZD)
g = sum(len(o) - 8 for o in out)
z = sum(len(ob.zlib_compress(r[8:])) for r in raw)
ns = sum(s.size for s in sigs)
worse = sum(1 for o, r in zip(out, raw) if len(o) - 8 > len(ob.zlib_compress(r[8:])))
print("real reads: %d records, %d samples: GPU %.5f B/sample, zlib-6 %.5f B/sample, ratio %.5f; records larger than zlib-6: %d" % (len(sigs), ns, g / ns, z / ns, g / z, worse))
