"""Soak test of the decode path (not part of the suite: minutes, not seconds): thousands of records of random length and content,
compressed by stock zlib at every level / strategy and by this library's own encoder, decoded by the default kernels and compared
with the source signals; also through the fused-unpack switch and the option that forbids the fallback decoder.
python tools/decode_soak.py [records] [seed]"""
import os, sys, zlib, time, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_bind as ob
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
t0 = time.time()
sigs, streams = [], []
for i in range(n_rec):
    kind = i % 6
    n = int(np.exp(rng.uniform(np.log(1), np.log(60000))))
    if kind == 0: sig = ob.synth_read(0x5105, 5000 + i, n)
    elif kind == 1: sig = (500 + np.cumsum(rng.integers(-12, 13, n)) % 400).astype(np.int16)
    elif kind == 2: sig = rng.integers(-32768, 32768, n).astype(np.int16)
    elif kind == 3: sig = np.repeat(rng.integers(300, 900, n // 7 + 1), rng.integers(1, 14, n // 7 + 1))[:n].astype(np.int16)   # stalls
    elif kind == 4: sig = np.tile((400 + rng.integers(-200, 200, int(rng.integers(3, 90)))).astype(np.int16), n)[:n]       # periodic
    else: sig = np.where(rng.random(n) < 0.97, 512, rng.integers(0, 1024, n)).astype(np.int16)
    sigs.append(sig)
hdrs = [press.pack_hdr(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0) for i in range(n_rec)]
own = [r[8:] for r in press.encode_records(sigs, hdrs)]
for i, sig in enumerate(sigs):
    rec, keep = ob.make_rec(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0, sig)
    p = ob.rec_pack(rec, ob.SIG_SVB_ZD)
    lvl = (1, 6, 9, 6)[i % 4]
    if i % 7 == 3:
        c = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, (zlib.Z_RLE, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED)[i % 4])
        streams.append(c.compress(p) + c.flush())
    else:
        streams.append(zlib.compress(p, lvl))
print("built %d records (%.1f M samples) in %.0f s" % (n_rec, sum(map(len, sigs)) / 1e6, time.time() - t0))
bad = 0
for label, recs, opts in (("own encoder, default", own, {}), ("stock zlib, default", streams, {}), ("stock zlib, no fused unpack", streams, {b"unpack_fused": 0}),
                          ("stock zlib, round-1 kernels", streams, {b"inflate_par": 0}), ("own encoder, no fallback pass", own, {b"inflate_par": 2}),
                          ("stock zlib, no fallback pass", streams, {b"inflate_par": 2})):
    for k, v in opts.items(): _lib.check(L.s5gpu_set_option(k, v))
    t1 = time.time()
    got = press.decode_records(recs, raise_on_error=False)
    for k in opts: _lib.check(L.s5gpu_set_option(k, 1))
    st = collections.Counter(g["status"] for g in got)
    wrong = sum(1 for g, s in zip(got, sigs) if g["status"] == 0 and not np.array_equal(g["signal"], s))
    fail = sum(1 for g in got if g["status"] not in (0, 8))
    bad += wrong + fail
    print("%-34s statuses %s  wrong signals %d  (%.1f s)" % (label, dict(st), wrong, time.time() - t1))
if ob.zstd_ref() is not None:      # zstd records: this library's frames and libzstd's own (levels 1, 3, 9)
    sub = list(range(0, n_rec, max(1, n_rec // 4000)))
    own_z = [r[8:] for r in press.encode_records([sigs[i] for i in sub], [hdrs[i] for i in sub], None, press.REC_ZSTD, press.SIG_SVB_ZD)]
    lib_z = []
    for k, i in enumerate(sub):
        rec, keep = ob.make_rec(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0, sigs[i])
        lib_z.append(ob.zstd_compress(ob.rec_pack(rec, ob.SIG_SVB_ZD), (1, 3, 9)[k % 3]))
    for label, recs in (("zstd, own frames", own_z), ("zstd, libzstd frames", lib_z)):
        got = press.decode_records(recs, press.REC_ZSTD, press.SIG_SVB_ZD, raise_on_error=False)
        wrong = sum(1 for g, i in zip(got, sub) if g["status"] != 0 or not np.array_equal(g["signal"], sigs[i]))
        bad += wrong
        print("%-34s %d records  wrong %d" % (label, len(recs), wrong))
sub = list(range(1, n_rec, max(1, n_rec // 4000)))     # ex-zd signal press under zlib and zstd: device round trip (the oracle checks the blobs in the suite)
for rm_name, rm in (("zlib", press.REC_ZLIB), ("zstd", press.REC_ZSTD)):
    recs = [r[8:] for r in press.encode_records([sigs[i] for i in sub], [hdrs[i] for i in sub], None, rm, press.SIG_EX_ZD)]
    got = press.decode_records(recs, rm, press.SIG_EX_ZD, raise_on_error=False)
    wrong = sum(1 for g, i in zip(got, sub) if g["status"] != 0 or not np.array_equal(g["signal"], sigs[i]))
    bad += wrong
    print("%-34s %d records  wrong %d" % ("ex-zd under " + rm_name, len(recs), wrong))
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
