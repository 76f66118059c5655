"""Where a record's wave spends its time in the parallel inflate (k_inflate_par_np): clock ticks per phase, averaged over the batch (variant build:
tools/variant.sh iprobe -DS5_IPROBE; S5GPU_LIB=slow5tools_amd/_variants/libs5_iprobe.so python tools/inflate_phases.py [reads] [samples]).
The probe's own atomics slow the kernel: read the shares."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode_stream(); torch.cuda.synchronize()
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
L.s5gpu_decode_scratch_bytes.restype = C.c_uint64; L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
sb = int(L.s5gpu_decode_scratch_bytes(pay_cap))
scr = torch.empty(sb, dtype=torch.uint8, device="cuda")
a.flags, a.payload, a.payload_bytes, a.max_pay_cap = _lib.DEC_NO_PAYLOAD, scr.data_ptr(), sb, pay_cap
a.max_in_len = int(d["in_len"].max())
z = (C.c_ulonglong * 20)()
ts = []
for i in range(3):
    L.s5gpu_iprobe_read(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.s5gpu_decode_dev(C.byref(a), None), "decode"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
L.s5gpu_iprobe_read(z)
st = fields.view(torch.int32).view(n_reads, 16)[:, 0]
names = ["zlib header check, window load", "block header, 3-bit lengths of the code-length code", "tables of the code-length code (infl_build, 19 symbols)",
         "code-length sequence (infl_cl_sequence_wave)", "lit/len symbols in canonical order (infl_build_syms)", "distance tables (infl_build)",
         "canonical limits, length-step choice", "segments, synchronisation passes", "prefix sums, round bookkeeping", "output pass",
         "runs, fences", "waiting matches", "Adler-32", "parse + svb-zd unpack"]
tot = sum(z[:14]); nrec = max(z[19], 1)
print("%d records x %d samples: %.2f ms (%.2f M records/s), ok %s; ticks per record %.0f" % (n_reads, n, min(ts), n_reads / min(ts) / 1e3, bool((st == 0).all().item()), tot / nrec))
for i, nm in enumerate(names):
    print("  %-62s %8.0f ticks  %5.1f %%" % (nm, z[i] / nrec, 100.0 * z[i] / max(tot, 1)))
