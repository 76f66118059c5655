"""Soak test of the encode paths (not part of the suite): thousands of records of random length and content through the zlib and zstd
record presses with every signal press; every record is decompressed by the stock library on the CPU (zlib / libzstd) and compared
with the oracle's payload.  python tools/encode_soak.py [records] [seed]"""
import os, sys, zlib, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_bind as ob
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
sigs = []
for i in range(n_rec):
    kind = i % 6
    n = int(np.exp(rng.uniform(np.log(1), np.log(int(os.environ.get('S5_SOAK_MAXLEN', 40000))))))
    if kind == 0: sig = ob.synth_read(0x5105, 9000 + i, n)
    elif kind == 1: sig = (500 + np.cumsum(rng.integers(-12, 13, n)) % 400).astype(np.int16)
    elif kind == 2: sig = rng.integers(-32768, 32768, n).astype(np.int16)
    elif kind == 3: sig = np.repeat(rng.integers(300, 900, n // 7 + 1), rng.integers(1, 14, n // 7 + 1))[:n].astype(np.int16)
    elif kind == 4: sig = np.tile((400 + rng.integers(-200, 200, int(rng.integers(3, 90)))).astype(np.int16), n)[:n]
    else: sig = np.where(rng.random(n) < 0.97, 512, rng.integers(0, 1024, n)).astype(np.int16)
    sigs.append(sig)
hdrs = [press.pack_hdr(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0) for i in range(n_rec)]
auxs = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(0, 60, n_rec)]
bad = 0
have_zstd = ob.zstd_ref() is not None
for rec_name, rm in (("zlib", press.REC_ZLIB), ("zstd", press.REC_ZSTD)):
    if rm == press.REC_ZSTD and not have_zstd: continue
    for sig_name, sm in (("svb-zd", press.SIG_SVB_ZD), ("none", press.SIG_NONE), ("ex-zd", press.SIG_EX_ZD)):
        t0 = time.time()
        out = press.encode_records(sigs, hdrs, auxs, rm, sm)
        raw = press.encode_records(sigs, hdrs, auxs, press.REC_NONE, sm)
        wrong = 0
        for o, r in zip(out, raw):
            try:
                p = zlib.decompress(o[8:]) if rm == press.REC_ZLIB else ob.zstd_decompress(o[8:], len(r))
            except Exception:
                p = None
            wrong += p != r[8:] or int.from_bytes(o[:8], "little") != len(o) - 8
        bad += wrong
        print("%-5s + %-6s %6d records  %.4f B/sample  wrong %d  (%.1f s)" % (rec_name, sig_name, n_rec, sum(map(len, out)) / max(1, sum(map(len, sigs))), wrong, time.time() - t0))
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
