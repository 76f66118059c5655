"""Soak of the no-payload decode's LDS form (k_inflate_par_np_lp, round 6; not part of the suite): tens of thousands of records of random
length (1 .. 5200 samples: most fit one inflate window and the LDS payload, some do not and must be declined and redone by the slot
decoder) and content, written by this library's encoder and by stock zlib at every level / strategy, decoded with max_in_len naming the
LDS kernel, compared with the source signals and with the slot form's statuses.   python tools/np_lds_soak.py [records] [seed]"""
import os, sys, zlib, time, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_bind as ob
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
t0 = time.time()
sigs = []
for i in range(n_rec):
    kind = i % 6
    n = int(rng.integers(1, 5200)) if i % 5 else int(rng.integers(3800, 4200))
    if kind == 0: sig = ob.synth_read(0x5105, 7000 + i, n)
    elif kind == 1: sig = (500 + np.cumsum(rng.integers(-12, 13, n)) % 400).astype(np.int16)
    elif kind == 2: sig = rng.integers(-32768, 32768, n).astype(np.int16)
    elif kind == 3: sig = np.repeat(rng.integers(300, 900, n // 7 + 1), rng.integers(1, 14, n // 7 + 1))[:n].astype(np.int16)
    elif kind == 4: sig = np.tile((400 + rng.integers(-200, 200, int(rng.integers(3, 90)))).astype(np.int16), n)[:n]
    else: sig = np.where(rng.random(n) < 0.97, 512, rng.integers(0, 1024, n)).astype(np.int16)
    sigs.append(sig)
hdrs = [press.pack_hdr(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0) for i in range(n_rec)]
own = [r[8:] for r in press.encode_records(sigs, hdrs)]
streams = []
for i, sig in enumerate(sigs):
    rec, keep = ob.make_rec(b"read_%07d" % i, i % 5, 8192.0, 3.0, 1400.0, 4000.0, sig)
    p = ob.rec_pack(rec, ob.SIG_SVB_ZD)
    lvl = (1, 6, 9, 0)[i % 4] if i % 11 else 6
    if i % 7 == 3:
        c = zlib.compressobj(max(lvl, 1), zlib.DEFLATED, 15, 8, (zlib.Z_RLE, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED)[i % 4])
        streams.append(c.compress(p) + c.flush())
    else:
        streams.append(zlib.compress(p, lvl))
print("built %d records (%.1f M samples) in %.0f s" % (n_rec, sum(map(len, sigs)) / 1e6, time.time() - t0), flush=True)
caps = [max(len(s), 8) for s in sigs]
bad = 0
for label, recs in (("own encoder", own), ("stock zlib", streams)):
    res = {}
    for lds in (1, 0):
        _lib.check(L.s5gpu_set_option(b"np_lds_payload", lds))
        t1 = time.time()
        res[lds] = press.decode_signals_dev(recs, max_pay_cap=3 * 5200 + 2048, sig_caps=caps, max_in_len=4000)
        dt = time.time() - t1
    _lib.check(L.s5gpu_set_option(b"np_lds_payload", 1))
    f1, s1 = res[1]; f0, s0 = res[0]
    wrong = sum(1 for i in range(n_rec) if f1["status"][i] != 0 or not np.array_equal(s1[i], sigs[i]))
    diff = int((f1["status"] != f0["status"]).sum())
    big = sum(1 for r in recs if len(r) > 4000)
    bad += wrong + diff
    print("%-12s LDS form: statuses %s, wrong or failed %d, statuses differing from the slot form %d; %d records longer than the hint (declined -> slot decoder)" %
          (label, dict(collections.Counter(f1["status"].tolist())), wrong, diff, big), flush=True)
print("np_lds_soak:", "OK" if bad == 0 else "%d BAD" % bad)
sys.exit(1 if bad else 0)
