"""Decode rate of k_zstd_inflate on frames written by libzstd itself (level 1, what a reference-written zstd BLOW5 holds):
131 072 frames of 4000-sample svb-zd records (256 distinct ones, repeated).  Needs libzstd.so.1.  python tools/zstd_ref_frames.py"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, oracle_bind as ob
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads, n = 131072, 4000
# libzstd level-1 frames of real svb payloads (the frames a reference-written file holds)
b = press.DeviceBatch([n] * 256, rec_method=press.REC_NONE, with_stream_out=False)
b.synth(); b.encode(); raw = b.records()
frames = [ob.zstd_compress(r[8:], 1) for r in raw]
blob = bytearray(); offs = []; lens = []
for i in range(n_reads):
    f = frames[i % 256]; offs.append(len(blob)); lens.append(len(f)); blob += f; blob += bytes((-len(blob)) % 16)
dev = torch.device("cuda:0")
t_in = torch.frombuffer(blob + bytes(64), dtype=torch.uint8).to(dev)
pcap = 6144
desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
desc["in_off"] = offs; desc["in_len"] = lens
desc["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pcap; desc["pay_cap"] = pcap - 16
t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
pay = torch.empty(n_reads * pcap + 64, dtype=torch.uint8, device=dev)
fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
for dbg in (0,):
    a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, 2, 1
    a.desc, a.in_, a.payload, a.fields = t_desc.data_ptr(), t_in.data_ptr(), pay.data_ptr(), fields.data_ptr()
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    st = fields.cpu().numpy().view(_lib.REC_FIELDS)["status"]
    print("k_zstd_inflate on libzstd level-1 frames: %.2f ms  (%.2f M frames/s)  ok %s" % (min(ts[1:]), n_reads / min(ts[1:]) / 1e3, bool((st == 0).all())))
print("frame bytes avg", np.mean(lens))
