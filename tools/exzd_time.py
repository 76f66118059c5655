#!/usr/bin/env python3
"""Throughput of the ex-zd paths on the synthetic workload: encode (k_pack + k_deflate_staged) and decode (inflate + k_unpack)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

L = _lib.lib()
_lib.check(L.s5gpu_init(0))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = "cuda:0"
for sm, name in ((press.SIG_EX_ZD, "ex-zd"), (press.SIG_SVB_ZD, "svb-zd (two-pass path, for comparison)")):
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), rec_method=press.REC_ZLIB, sig_method=sm, device=dev)
    b.synth()

    def enc():
        b.encode(); b.compact()

    for _ in range(2):
        enc()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); enc(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    off = b.rec_off.cpu().numpy().astype(np.int64)
    total = int(off[n_reads])
    print("%s: encode %.3f ms for %d reads x %d = %.1f GB/s raw signal, %.2f M reads/s, %.4f B/sample" %
          (name, ms, n_reads, n, n_reads * n * 2 / ms / 1e6, n_reads / ms / 1e3, total / (n_reads * n)))
    # decode
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    sig_cap = (n + 7) // 8 * 8
    payload = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(n_reads * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
    d = np.zeros(n_reads, dtype=_lib.REC_DESC)
    d["in_off"] = off[:-1] + 8; d["in_len"] = np.diff(off) - 8
    d["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
    d["sig_off"] = np.arange(n_reads, dtype=np.uint64) * sig_cap; d["sig_cap"] = sig_cap
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method, a.max_pay_cap = n_reads, 1, sm, pay_cap     # (max_pay_cap: short records, the kernel's 24-wave shape)
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()

    def dec():
        _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()))

    dec(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dec(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[1]
    ok = int(fields.view(torch.int32).view(n_reads, 16)[:, 0].abs().sum().item()) == 0
    stride = sig_cap
    same = torch.equal(sig[: n_reads * stride].view(n_reads, stride)[:, :n], b.sig[: n_reads * stride].view(n_reads, stride)[:, :n])
    print("%s: decode %.3f ms = %.2f M reads/s  status ok %s  round trip %s" % (name, ms, n_reads / ms / 1e3, ok, same))
    del b, payload, sig
    torch.cuda.empty_cache()
