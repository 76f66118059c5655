"""Device time of the zstd record press on the bench workload: encode (fused), decode (wave per frame).
python tools/zstd_time.py [n_reads] [n_samples]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
from slow5tools_amd import _lib, press  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
L = _lib.lib()
_lib.check(L.s5gpu_init(0), "init")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for rec, name in ((press.REC_ZSTD, "zstd"), (press.REC_ZLIB, "zlib")):
    b = press.DeviceBatch([n_samp] * n_reads, rec_method=rec, with_stream_out=True)
    b.synth()
    ms = timed(b.encode)
    lens = b.out_len[: b.n].cpu().numpy().astype(np.int64)
    print("%s encode: %.2f ms  %.1f GB/s of signal  %.2f M reads/s  %.4f B/sample" %
          (name, ms, n_reads * n_samp * 2 / ms / 1e6, n_reads / ms / 1e3, (lens.sum() - 8 * n_reads) / (n_reads * n_samp)))
    # decode the same records
    b.compact()
    torch.cuda.synchronize()
    off = b.rec_off.cpu().numpy()
    dev = b.dev
    desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
    desc["in_off"] = off[:-1] + 8
    desc["in_len"] = (off[1:] - off[:-1] - 8).astype(np.uint32)
    pcap = (b.tot["max_payload"] + 31) // 16 * 16
    desc["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pcap
    desc["pay_cap"] = pcap - 16
    desc["sig_off"] = np.arange(n_reads, dtype=np.uint64) * ((n_samp + 7) // 8 * 8)
    desc["sig_cap"] = n_samp
    t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    pay = torch.empty(n_reads * pcap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(n_reads * ((n_samp + 7) // 8 * 8) + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_reads * _lib.REC_FIELDS.itemsize, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method, a.max_pay_cap = n_reads, rec, press.SIG_SVB_ZD, pcap - 16
    a.desc, a.in_, a.payload, a.sig_out, a.fields = t_desc.data_ptr(), b.stream_out.data_ptr(), pay.data_ptr(), sig.data_ptr(), fields.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ms_i = timed(lambda: _lib.check(L.s5gpu_inflate_dev(C.byref(a), st), "inflate"))
    ms_d = timed(lambda: _lib.check(L.s5gpu_decode_dev(C.byref(a), st), "decode"))
    f = fields.cpu().numpy().view(_lib.REC_FIELDS)
    ok = bool((f["status"] == 0).all() and (f["n_samples"] == n_samp).all())
    ref = b.sig[: n_samp].cpu().numpy()
    ok = ok and np.array_equal(sig[:n_samp].cpu().numpy(), ref)
    print("%s decode: record stage %.2f ms (%.2f M reads/s), whole decode %.2f ms (%.2f M reads/s)  ok=%s" %
          (name, ms_i, n_reads / ms_i / 1e3, ms_d, n_reads / ms_d / 1e3, ok))
    del b, pay, sig
    torch.cuda.empty_cache()
