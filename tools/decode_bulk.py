"""One bulk decode call (the whole index in one s5gpu_decode_dev) a few times, nothing else: the driver of tools/pmc_decode_traffic.sh and of
quick A/B timings.   python tools/decode_bulk.py [reads] [samples] [np|full] [reps]
np   = S5GPU_DEC_NO_PAYLOAD (k_inflate_par_np: persistent workgroups, scratch slots, fields + signals out)
full = payload slots written out as well (k_inflate_par<1> + k_inflate_fallback + k_unpack_rest)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
for kv in filter(None, os.environ.get("S5_OPTS", "").split(",")):      # library options for A/B runs, e.g. S5_OPTS=np_lds_payload=0
    k, v = kv.split("="); _lib.check(L.s5gpu_set_option(k.encode(), int(v)), "option " + kv)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
mode = sys.argv[3] if len(sys.argv) > 3 else "np"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode_stream(); torch.cuda.synchronize()
assert b.stream_ok()
off = b.rec_off.cpu().numpy().astype(np.int64)
pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
sig_cap = (n + 7) // 8 * 8
d = np.zeros(n_reads, dtype=_lib.REC_DESC)
d["in_off"] = off[:-1] + 8; d["in_len"] = np.diff(off) - 8
d["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
d["sig_off"] = np.arange(n_reads, dtype=np.uint64) * sig_cap; d["sig_cap"] = sig_cap
desc = torch.from_numpy(d.view(np.uint8).copy()).cuda()
sig = torch.empty(n_reads * sig_cap + 64, dtype=torch.int16, device="cuda")
fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device="cuda")
a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, 1, 1
a.desc, a.in_, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), sig.data_ptr(), fields.data_ptr()
if mode == "np":
    L.s5gpu_decode_scratch_bytes.restype = C.c_uint64; L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
    sb = int(os.environ.get("S5_SCRATCH_BYTES", 0)) or int(L.s5gpu_decode_scratch_bytes(pay_cap))
    scr = torch.empty(sb, dtype=torch.uint8, device="cuda")
    a.flags, a.payload, a.payload_bytes, a.max_pay_cap = _lib.DEC_NO_PAYLOAD, scr.data_ptr(), sb, pay_cap
    a.max_in_len = int(d["in_len"].max())
    a.flags |= int(os.environ.get("S5_CUT", 0)) << 24      # probe build only (tools/np_probe_pmc.sh): stop every record at a cut-off of the inflate
else:
    pay = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device="cuda")
    a.payload, a.max_pay_cap = pay.data_ptr(), pay_cap
ts = []
soak = os.environ.get("S5_SOAK", "") not in ("", "0")     # check every repetition (signals cleared in between): a soak for rare wrong decodes
soak_bad = 0
want = b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]
for rep in range(reps):
    if soak:
        sig.zero_(); fields.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.s5gpu_decode_dev(C.byref(a), None), "decode"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    if soak:
        st_ = fields.view(torch.int32).view(n_reads, 16)[:, 0]
        good = bool((st_ == 0).all().item()) and bool(torch.equal(sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n], want))
        if not good:
            soak_bad += 1
            neq = (sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n] != want).any(dim=1)
            bad = torch.nonzero(neq).flatten().cpu().numpy()
            print("   rep %d WRONG: %d reads differ, first %s; nonzero statuses %d" % (rep, len(bad), bad[:12], int((st_ != 0).sum().item())))
if soak:
    print("soak %s: %d repetitions, %d wrong" % (mode, reps, soak_bad))
    ts = ts[:8]
st = fields.view(torch.int32).view(n_reads, 16)[:, 0]
ok = bool((st == 0).all().item()) and bool(torch.equal(sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n], b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]))
if not ok:
    import collections
    print("   statuses", dict(collections.Counter(st.cpu().tolist())))
    neq = (sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n] != b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]).any(dim=1)
    bad = torch.nonzero(neq).flatten().cpu().numpy()
    print("   reads with a wrong signal: %d, first %s" % (len(bad), bad[:20]))
z = int(off[-1])
print("decode_bulk %s: %d reads x %d samples: %s ms (min %.3f)  %.1f M reads/s  Z+2N = %.0f B/read  identical %s" % (
    mode, n_reads, n, " ".join("%.3f" % t for t in ts), min(ts[1:] or ts), n_reads / min(ts[1:] or ts) / 1e3, (z + 2 * n * n_reads) / n_reads, ok))
