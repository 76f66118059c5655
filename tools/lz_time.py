"""The LZ77 path (record press zlib over signal press none, csrc/lz_dev.h): record sizes on the reference's fixtures next to the
reference's own records and zlib levels, and the encode rate on synthetic reads.  python tools/lz_time.py [reads] [samples]"""
import sys, zlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_bind as ob
from blow5_fixture import Blow5, golden
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
for name in ("exp_1_lossless_zlib.blow5", "exp_lossless_gzip.blow5", "example_multi_rg_v0.1.0.blow5"):
    f = Blow5(golden(name))
    sigs, hdrs, auxs, refs, z1 = [], [], [], 0, 0
    for rec in f.records:
        pl = zlib.decompress(rec) if f.rec_method == 1 else rec
        d = ob.rec_parse(pl, f.sig_method)
        sigs.append(d["signal"]); hdrs.append(press.pack_hdr(d["read_id"], d["read_group"], d["digitisation"], d["offset"], d["range"], d["sampling_rate"])); auxs.append(d["aux"])
        refs += len(zlib.compress(pl, 6)); z1 += len(zlib.compress(pl, 1))
    recs = press.encode_records(sigs, hdrs, auxs, press.REC_ZLIB, press.SIG_NONE)
    got = sum(len(r) - 8 for r in recs)
    print("%-34s GPU %7d B   zlib-6 %7d B (x %.4f)   zlib-1 %7d B" % (name, got, refs, got / refs, z1))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), rec_method=press.REC_ZLIB, sig_method=press.SIG_NONE)
b.synth()
for _ in range(2): b.encode()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.encode(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = min(ts)
z = int(b.out_len[:n_reads].sum().item())
idx = list(range(0, n_reads, max(1, n_reads // 64)))
ref = 0
for i, rec in zip(idx, b.records(idx)):
    sig = ob.synth_read(0x5105, i, n)
    r, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
    pay = ob.rec_pack(r, ob.SIG_NONE)
    assert zlib.decompress(rec[8:]) == pay
    ref += len(zlib.compress(pay, 6)) + 8
got = int(b.out_len[torch.tensor(idx)].sum().item())
print("none + zlib, %d reads x %d samples: (k_pack +) k_deflate_lz %.2f ms = %.1f GB/s of raw signal, %.2f M reads/s; %.4f B/sample (zlib-6 on the sampled reads: x %.4f)" % (
    n_reads, n, ms, n_reads * 2 * n / ms / 1e6, n_reads / ms / 1e3, z / (n_reads * n), got / ref))
