"""Where a frame's wave spends its time in the zstd decoder: clock ticks per phase, averaged over the batch (variant build:
tools/variant.sh zprobe -DS5_ZPROBE; S5GPU_LIB=slow5tools_amd/_variants/libs5_zprobe.so python tools/zstd_phases.py [own|libzstd] [n_frames])."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
kind = sys.argv[1] if len(sys.argv) > 1 else "own"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
n = 4000
dev = torch.device("cuda:0")
if kind == "own":
    b = press.DeviceBatch([n] * n_reads, rec_method=press.REC_ZSTD, with_stream_out=True)
    b.synth(); b.encode(); b.compact(); torch.cuda.synchronize()
    off = b.rec_off.cpu().numpy()
    offs, lens, t_in = off[:-1] + 8, (off[1:] - off[:-1] - 8).astype(np.uint32), b.stream_out
else:
    import oracle_bind as ob
    b = press.DeviceBatch([n] * 256, rec_method=press.REC_NONE, with_stream_out=False)
    b.synth(); b.encode(); raw = b.records()
    frames = [ob.zstd_compress(r[8:], 1) for r in raw]
    blob = bytearray(); offs = []; lens = []
    for i in range(n_reads):
        f = frames[i % 256]; offs.append(len(blob)); lens.append(len(f)); blob += f; blob += bytes((-len(blob)) % 16)
    t_in = torch.frombuffer(blob + bytes(64), dtype=torch.uint8).to(dev)
pcap = 6144
desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
desc["in_off"] = offs; desc["in_len"] = lens
desc["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pcap; desc["pay_cap"] = pcap - 16
t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
pay = torch.empty(n_reads * pcap + 64, dtype=torch.uint8, device=dev)
fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, 2, 1
a.desc, a.in_, a.payload, a.fields = t_desc.data_ptr(), t_in.data_ptr(), pay.data_ptr(), fields.data_ptr()
z = (C.c_ulonglong * 16)()
ts = []
for i in range(3):
    L.s5gpu_zprobe_read(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), None), "inflate"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
L.s5gpu_zprobe_read(z)
st = fields.cpu().numpy().view(_lib.REC_FIELDS)["status"]
names = ["headers", "tree description, Huffman table", "literal streams", "sequence tables", "(between)", "FSE chains", "batch bookkeeping, checks",
         "batch literals", "batch matches", "last literals, end"]
tot = sum(z[:10])
print("%s frames: %d x %d samples, %.2f ms (%.2f M frames/s), ok %s; ticks per frame %.0f" % (kind, n_reads, n, min(ts), n_reads / min(ts) / 1e3, bool((st == 0).all()), tot / max(z[15], 1)))
print("  full passes over the literal streams per frame: %.2f" % (z[10] / max(z[15], 1)))
for i, nm in enumerate(names):
    print("  %-36s %8.0f ticks  %5.1f %%" % (nm, z[i] / max(z[15], 1), 100.0 * z[i] / max(tot, 1)))
