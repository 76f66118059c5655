"""Which records does the parallel-inside-the-record inflate decline?  Status histogram of k_inflate_par alone (no fallback pass)
on synthetic reads.  python tools/par_probe.py [reads] [samples]
The cut-offs and counters need the probe build:  tools/variant.sh probe -DS5_PAR_PROBE && S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so python tools/par_probe.py
(the product kernel carries no probe hooks; without the variant only the status histogram and the whole-kernel times are printed)"""
import ctypes as C, sys, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode(); b.compact(); torch.cuda.synchronize()
off = b.rec_off.cpu().numpy().astype(np.int64)
pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
d = np.zeros(n_reads, dtype=_lib.REC_DESC)
d["in_off"] = off[:-1] + 8; d["in_len"] = np.diff(off) - 8
d["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
desc = torch.from_numpy(d.view(np.uint8).copy()).cuda()
pay = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device="cuda")
fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device="cuda")
a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, 1, 1
a.desc, a.in_, a.payload, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), pay.data_ptr(), fields.data_ptr()
for mode in (2, 1, 0):
    _lib.check(L.s5gpu_set_option(b"inflate_par", mode), "opt")
    ts = []
    for _ in range(3):
        fields.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    f = fields.cpu().numpy().view(_lib.REC_FIELDS)
    st = f["status"]
    print("inflate_par=%d: %.3f ms  statuses %s" % (mode, min(ts), dict(collections.Counter(st.tolist()))))
    if mode == 2 and "probe" in os.environ.get("S5GPU_LIB", ""):
        a.sig_method = 99
        fields.zero_(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); torch.cuda.synchronize()
        f = fields.cpu().numpy().view(_lib.REC_FIELDS)
        print("   sync passes per record: mean %.1f max %d; rounds mean %.2f; decline reasons %s" % (f["n_samples"].mean(), f["n_samples"].max(), f["read_id_len"].mean(), dict(collections.Counter(f["read_group"].tolist()))))
        for cut, what in ((81, "block header, 3-bit lengths"), (82, "+ code-length code tables"), (83, "+ code-length sequence"), (84, "+ lit/len symbols in canonical order"), (91, "+ distance tables (block header + tables)"), (95, "+ limits, lit/len lookup table"), (96, "+ window, first (tail) pass"), (92, "+ sync passes"), (97, "+ output pass"), (93, "+ runs, waiting matches"), (1, "+ Adler-32 (whole kernel)")):
            a.sig_method = cut
            tt = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
            print("   cut-off %-34s %.3f ms" % (what, min(tt)))
        a.sig_method = 1
