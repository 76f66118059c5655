#!/usr/bin/env python3
"""Quick timings of the non-headline configs: long reads (staged path) and decode."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

L = _lib.lib()
_lib.check(L.s5gpu_init(0))


def t_ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def decode_batch_dev(b, n_recs):
    """decode records [0, n_recs) of b's compacted stream on device; returns (fn, tensors)"""
    stream, off = b.stream_bytes()
    off = off.astype(np.int64)
    n = int(b.desc_np["n_samples"][0])
    d = np.zeros(n_recs, dtype=_lib.REC_DESC)
    d["in_off"] = off[:n_recs] + 8
    d["in_len"] = (off[1:n_recs + 1] - off[:n_recs] - 8)
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    d["pay_off"] = np.arange(n_recs, dtype=np.uint64) * pay_cap
    d["pay_cap"] = pay_cap
    sig_cap = (n + 7) // 8 * 8
    d["sig_off"] = np.arange(n_recs, dtype=np.uint64) * sig_cap
    d["sig_cap"] = sig_cap
    dev = b.dev
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    payload = torch.empty(n_recs * pay_cap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(n_recs * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_recs * 64, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method = n_recs, 1, 1
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
    def fn():
        _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()))
    return fn, (desc, payload, sig, fields, sig_cap)


if "long" in sys.argv:
    n_reads, n = 8192, 100000
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), with_stream_out=True)
    b.synth()
    ms = t_ms(b.encode)
    b.compact(); torch.cuda.synchronize()
    z = int(b.out_len[:n_reads].sum().item())
    print("long reads: %d x %d samples: encode %.2f ms -> %.1f GB/s raw, %.4f B/sample" % (n_reads, n, ms, n_reads * n * 2 / ms / 1e6, z / (n_reads * n)))
if "decode" in sys.argv:
    n_reads, n = 200000, 4000
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64))
    b.synth(); b.encode(); b.compact()
    for k in (4096, 65536, 200000):
        fn, keep = decode_batch_dev(b, k)
        ms = t_ms(fn)
        sig_cap = keep[4]
        got = keep[2][: k * sig_cap].view(k, sig_cap)[:, :n]
        ok = bool((got == b.sig[: k * n].view(k, n)).all().item())
        st = keep[3].view(torch.int32).view(k, 16)[:, 0]
        print("decode %6d recs: %.3f ms -> %.2f M reads/s, %.1f GB/s raw out; status ok=%s signals ok=%s" % (k, ms, k / ms / 1e3, k * n * 2 / ms / 1e6, bool((st == 0).all().item()), ok))
