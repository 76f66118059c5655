"""Where does the lane-per-record inflate kernel (csrc/inflate_simt_dev.h) lose time?  The same 262 144 decodes with
  - every lane on its own record (the real case),
  - the same 64 records in every wave (input hot in L2),
  - one record in every lane (no divergence at all),
  - one record per wave, a different one in each wave,
  - 4096 copies of one record at distinct addresses (control flow convergent, addresses scattered).
Result that shaped the kernel: scattered addresses and HBM latency cost nothing; lanes disagreeing on the path of a symbol cost
2.6x (table form) / 1.9x (compare form).  python tools/simt_probe.py"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads, n_samp = 262144, 4000
b = press.DeviceBatch([n_samp] * n_reads, rec_method=press.REC_ZLIB, with_stream_out=True)
b.synth(); b.encode(); b.compact(); torch.cuda.synchronize()
off = b.rec_off.cpu().numpy(); dev = b.dev
pcap = (b.tot["max_payload"] + 31) // 16 * 16
def run(tag, in_off, in_len):
    desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
    desc["in_off"] = in_off; desc["in_len"] = in_len
    desc["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pcap; desc["pay_cap"] = pcap - 16
    t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    pay = torch.empty(n_reads * pcap + 64, dtype=torch.uint8, device=dev)
    fields = torch.zeros(n_reads * _lib.REC_FIELDS.itemsize, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, press.REC_ZLIB, press.SIG_SVB_ZD
    a.desc, a.in_, a.payload, a.fields = t_desc.data_ptr(), b.stream_out.data_ptr(), pay.data_ptr(), fields.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), st), "inflate"); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    f = fields.cpu().numpy().view(_lib.REC_FIELDS)
    print("%-40s %.2f ms  %.2f M reads/s ok=%s" % (tag, min(ts[1:]), n_reads / min(ts[1:]) / 1e3, bool((f["status"] == 0).all())))
io = (off[:-1] + 8).astype(np.uint64); il = (off[1:] - off[:-1] - 8).astype(np.uint32)
run("distinct records (real)", io, il)
idx = np.arange(n_reads) % 64
run("64 distinct records, repeated (L2-hot)", io[idx], il[idx])
idx = np.zeros(n_reads, dtype=np.int64)
run("one record for every lane (no divergence)", io[idx], il[idx])
idx = (np.arange(n_reads) // 64) % 4096
run("same record within a wave, distinct waves", io[idx], il[idx])
# 4096 copies of record 0 at distinct addresses: control flow convergent, input addresses divergent
r0 = b.stream_out[int(io[0]):int(io[0]) + int(il[0])].clone()
stride = (int(il[0]) + 64 + 15) // 16 * 16 + 48      # odd multiple of 16 to spread over channels
ncopy = 4096
buf = torch.zeros(ncopy * stride + 64, dtype=torch.uint8, device=dev)
for c in range(ncopy):
    buf[c * stride: c * stride + int(il[0])] = r0
old = b.stream_out
b.stream_out = buf
idx = np.arange(n_reads) % ncopy
run("copies of one record (control convergent)", (idx * stride).astype(np.uint64), np.full(n_reads, il[0], np.uint32))
