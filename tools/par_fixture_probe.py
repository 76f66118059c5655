"""The reference's own zlib records (tests/golden: real signals, written by slow5tools with stock zlib) through the inflate kernels:
which does the parallel-inside-the-record decoder take without the fallback pass, and how fast is a batch of them (the file's
records tiled to `batch` records)?  python tools/par_fixture_probe.py [batch]"""
import ctypes as C, os, sys, zlib, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from blow5_fixture import Blow5, golden, ZLIB_SVB_FIXTURES, ZLIB_NONE_FIXTURES
from slow5tools_amd import _lib
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for name in ZLIB_SVB_FIXTURES + ZLIB_NONE_FIXTURES:
    f = Blow5(golden(name))
    streams = list(f.records)
    plen = max(len(zlib.decompress(s)) for s in streams)
    lens = np.array([len(s) for s in streams], dtype=np.int64)
    off1 = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)])
    blob = np.zeros(off1[-1] + 64, dtype=np.uint8)
    for s, o in zip(streams, off1[:-1]):
        blob[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    idx = np.arange(batch) % len(streams)
    pay_cap = 16 * ((plen + 31) // 16)
    d = np.zeros(batch, dtype=_lib.REC_DESC)
    d["in_off"] = off1[idx]; d["in_len"] = lens[idx]
    d["pay_off"] = np.arange(batch, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
    desc = torch.from_numpy(d.view(np.uint8).copy()).cuda()
    inp = torch.from_numpy(blob).cuda()
    pay = torch.empty(batch * pay_cap + 64, dtype=torch.uint8, device="cuda")
    fields = torch.zeros(batch * 64, dtype=torch.uint8, device="cuda")
    a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = batch, 1, f.sig_method
    a.desc, a.in_, a.payload, a.fields = desc.data_ptr(), inp.data_ptr(), pay.data_ptr(), fields.data_ptr()
    out = []
    zbytes = float(lens[idx].sum())
    for mode in (2, 1, 0):
        _lib.check(L.s5gpu_set_option(b"inflate_par", mode), "opt")
        ts = []
        for _ in range(3):
            fields.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        st = fields.cpu().numpy().view(_lib.REC_FIELDS)["status"][:len(streams)]
        out.append((mode, min(ts), dict(collections.Counter(st.tolist()))))
    if ("merged" in name or "exp_1_lossless_zlib_svb" in name) and "probe" in os.environ.get("S5GPU_LIB", ""):     # where the time goes (tools/variant.sh probe -DS5_PAR_PROBE) (cut-offs of the parallel decoder)
        _lib.check(L.s5gpu_set_option(b"inflate_par", 2), "opt")
        keep = a.sig_method
        a.sig_method = 99
        fields.zero_(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); torch.cuda.synchronize()
        ff = fields.cpu().numpy().view(_lib.REC_FIELDS)[:len(streams)]
        cuts = []
        for cut in (91, 92, 94, 93, 1):
            a.sig_method = cut
            tt = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
            cuts.append(min(tt))
        a.sig_method = keep
        print("    waiting matches: batches of 64 per record %s, dependency steps %s" % ([int(x) & 0xFFFF for x in ff["read_group"]], [int(x) >> 16 for x in ff["read_group"]]))
        print("    sync passes per record %s, rounds %s; cumulative ms: header %.2f, + sync %.2f, + output pass / runs %.2f, + waiting matches %.2f, + Adler %.2f" % (ff["n_samples"].tolist(), ff["read_id_len"].tolist(), *cuts))
    if f.sig_method == 1:   # svb-zd records: the whole decode as s5gpu_decode_dev runs it (the inflating wave unpacks; its waiting list is the small one)
        _lib.check(L.s5gpu_set_option(b"inflate_par", 1), "opt")
        sig_cap = 8 * ((plen + 7) // 8)          # (samples <= payload bytes)
        d2 = d.copy()
        d2["sig_off"] = np.arange(batch, dtype=np.uint64) * sig_cap; d2["sig_cap"] = sig_cap
        desc2 = torch.from_numpy(d2.view(np.uint8).copy()).cuda()
        sig = torch.empty(batch * sig_cap + 64, dtype=torch.int16, device="cuda")
        a2 = _lib.DecodeArgs(); a2.n_recs, a2.rec_method, a2.sig_method = batch, 1, 1
        a2.desc, a2.in_, a2.payload, a2.fields, a2.sig_out = desc2.data_ptr(), inp.data_ptr(), pay.data_ptr(), fields.data_ptr(), sig.data_ptr()
        a2.max_pay_cap = pay_cap          # what the host calls pass: short records get the 24-wave shape of the kernel
        ts = []
        for _ in range(4):
            fields.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _lib.check(L.s5gpu_decode_dev(C.byref(a2), None), "decode"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ff = fields.cpu().numpy().view(_lib.REC_FIELDS)[:len(streams)]
        ok = all(int(x) == 0 for x in ff["status"])
        nsamp = int(fields.cpu().numpy().view(_lib.REC_FIELDS)["n_samples"][:batch].astype(np.int64).sum())
        print("    whole decode (s5gpu_decode_dev: inflate + parse + svb-zd unpack by the same wave): %.2f ms = %.2f G samples/s, statuses ok %s" % (min(ts[1:]), nsamp / min(ts[1:]) / 1e6, ok))
        del sig
    print("%-44s %2d records (%d..%d B) x %d:  " % (name, len(streams), lens.min(), lens.max(), batch) +
          "   ".join("par=%d %.2f ms (%.1f GB/s of zlib stream) %s" % (m, t, zbytes / t / 1e6, s) for m, t, s in out))
_lib.check(L.s5gpu_set_option(b"inflate_par", 1), "opt")
