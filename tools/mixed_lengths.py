"""Big zlib batches with the read lengths of a real run (log-normal, median 6000 samples, a tail of 50x the median): the
lane-per-record inflate kernel alone, the wave-per-record kernel alone, and the routed form (records counting-sorted by
compressed length, >= 32 KiB to the wave kernel on a second stream).  python tools/mixed_lengths.py"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
rng = np.random.default_rng(5)
n_reads = 262144
# log-normal read lengths like a nanopore run: median ~6000 samples, long tail
ns = np.clip(np.exp(rng.normal(np.log(6000), 0.9, n_reads)), 200, 400000).astype(np.uint64)
print("samples: median %d mean %d max %d total %.2f G" % (np.median(ns), ns.mean(), ns.max(), ns.sum() / 1e9))
b = press.DeviceBatch(ns, rec_method=press.REC_ZLIB, with_stream_out=True)
tot = b.sig.numel()
# one long synthetic trace cut into the reads: k_synth as a single read of `tot` samples (the event model is position-keyed)
_lib.check(L.s5gpu_synth_dev(b.sig.data_ptr(), 1, tot - 64, tot, 0x5105, 0, b._stream()), "synth")
_lib.check(L.s5gpu_synth_hdr_dev(b.hdr.data_ptr(), n_reads, 0, b._stream()), "hdr")
b.encode(); b.compact(); torch.cuda.synchronize()
off = b.rec_off.cpu().numpy(); dev = b.dev
io = (off[:-1] + 8).astype(np.uint64); il = (off[1:] - off[:-1] - 8).astype(np.uint32)
pay_off = np.zeros(n_reads, dtype=np.uint64)
pcaps = ((ns * 13 // 4 + 200 + 31) // 16 * 16).astype(np.uint64)
pay_off[1:] = np.cumsum(pcaps)[:-1]
pay = torch.empty(int(pcaps.sum()) + 64, dtype=torch.uint8, device=dev)
def run(tag, order, thr):
    desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
    desc["in_off"] = io[order]; desc["in_len"] = il[order]
    desc["pay_off"] = pay_off[order]; desc["pay_cap"] = (pcaps[order] - 16).astype(np.uint32)
    t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    fields = torch.zeros(n_reads * _lib.REC_FIELDS.itemsize, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs(); a.n_recs, a.rec_method, a.sig_method = n_reads, press.REC_ZLIB, press.SIG_SVB_ZD
    a.desc, a.in_, a.payload, a.fields = t_desc.data_ptr(), b.stream_out.data_ptr(), pay.data_ptr(), fields.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(L.s5gpu_set_option(b"inflate_simt_min", thr), "opt")
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.s5gpu_inflate_dev(C.byref(a), st), "inflate"); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    f = fields.cpu().numpy().view(_lib.REC_FIELDS)
    ms = min(ts[1:])
    print("%-34s %.2f ms  %.2f M reads/s  %.1f G samples/s ok=%s" % (tag, ms, n_reads / ms / 1e3, ns.sum() / ms / 1e6, bool((f["status"] == 0).all())))
_lib.check(L.s5gpu_set_option(b"inflate_par", 0), "opt")
ident = np.arange(n_reads)
_lib.check(L.s5gpu_set_option(b"inflate_route", 0), "opt")
run("lane kernel alone, file order", ident, 0)
run("wave kernel alone, file order", ident, 1 << 30)
_lib.check(L.s5gpu_set_option(b"inflate_route", 1), "opt")
run("routed: sorted lanes + long on waves", ident, 0)
_lib.check(L.s5gpu_set_option(b"inflate_par", 1), "opt")
run("parallel inside the record (default)", ident, 0)
# the same with the descriptors sorted by compressed length, longest first, on the host: what launch order is worth (one wave per record:
# the batch ends when its longest record does, and that one should not start last)
run("parallel, longest first (host-sorted)", np.argsort(-il.astype(np.int64), kind="stable"), 0)
run("parallel, shortest first (host-sorted)", np.argsort(il.astype(np.int64), kind="stable"), 0)
# ... and what the library does by itself (batches of >= 8192 records are counting-sorted by compressed length on the device): the default
# line above IS that; with the option off, file order
_lib.check(L.s5gpu_set_option(b"order_min", 0), "opt")
run("parallel, file order (order_min = 0)", ident, 0)
_lib.check(L.s5gpu_set_option(b"order_min", 8192), "opt")
