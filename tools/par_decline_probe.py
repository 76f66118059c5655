"""Records WRITTEN BY STOCK ZLIB (the reference's encoder: level 6, arbitrary LZ77 distances) through the inflate kernels: how many does
the parallel-inside-the-record decoder decline, and how fast is each kernel on them?  svb-zd records of synthetic reads, compressed on
the CPU (distinct records, tiled to the batch size).  python tools/par_decline_probe.py [distinct] [samples] [batch]"""
import ctypes as C, os, sys, zlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_bind as ob
from slow5tools_amd import _lib
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
distinct = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
streams, plen = [], 0
for i in range(distinct):
    sig = ob.synth_read(0x5105, i, n)
    rec, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 3.0, 1400.0, 4000.0, sig)
    p = ob.rec_pack(rec, ob.SIG_SVB_ZD); plen = max(plen, len(p))
    streams.append(zlib.compress(p, 6))
lens = np.array([len(s) for s in streams], dtype=np.int64)
off1 = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])
blob = np.zeros(off1[-1] + 64, dtype=np.uint8)
for s, o in zip(streams, off1[:-1]):
    blob[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
idx = np.arange(batch) % distinct
pay_cap = 16 * ((plen + 31) // 16)
d = np.zeros(batch, dtype=_lib.REC_DESC)
d["in_off"] = off1[idx]; d["in_len"] = lens[idx]
d["pay_off"] = np.arange(batch, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
desc = torch.from_numpy(d.view(np.uint8).copy()).cuda()
inp = torch.from_numpy(blob).cuda()
pay = torch.empty(batch * pay_cap + 64, dtype=torch.uint8, device="cuda")
fields = torch.zeros(batch * 64, dtype=torch.uint8, device="cuda")
a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = batch, 1, 1
a.desc, a.in_, a.payload, a.fields = desc.data_ptr(), inp.data_ptr(), pay.data_ptr(), fields.data_ptr()
print("stock zlib (level 6) svb-zd records: %d distinct x %d samples, %.0f B each, batch of %d" % (distinct, n, lens.mean(), batch))
for mode, what in ((2, "parallel inside the record alone (declined records keep status 8)"), (1, "default: parallel + fallback pass"), (0, "round-1 kernels (lane per record)")):
    _lib.check(L.s5gpu_set_option(b"inflate_par", mode), "opt")
    ts = []
    for _ in range(3):
        fields.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    st = fields.cpu().numpy().view(_lib.REC_FIELDS)["status"]
    print("  inflate_par=%d  %8.3f ms  %6.2f M records/s  statuses %s   %s" % (mode, min(ts), batch / min(ts) / 1e3, dict(collections.Counter(st.tolist())), what))
if "probe" not in os.environ.get("S5GPU_LIB", ""):      # the cut-offs and counters need the probe build (tools/variant.sh probe -DS5_PAR_PROBE)
    sys.exit(0)
_lib.check(L.s5gpu_set_option(b"inflate_par", 2), "opt")
a.sig_method = 99
fields.zero_(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); torch.cuda.synchronize()
f = fields.cpu().numpy().view(_lib.REC_FIELDS)
print("  sync passes per record: mean %.1f max %d; rounds mean %.2f; decline reasons %s" % (f["n_samples"].mean(), f["n_samples"].max(), f["read_id_len"].mean(), dict(collections.Counter(f["read_group"].tolist()))))
for cut, what in ((91, "block header + tables"), (92, "+ window, sync passes"), (93, "+ output pass, runs, waiting matches"), (1, "+ Adler-32 (whole kernel)"), (94, "whole kernel without the waiting matches")):
    a.sig_method = cut
    tt = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
    print("  cut-off %-38s %.3f ms" % (what, min(tt)))
_lib.check(L.s5gpu_set_option(b"inflate_par", 1), "opt")
