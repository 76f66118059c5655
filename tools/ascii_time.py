#!/usr/bin/env python3
"""Time the raw_signal text kernels (k_ascii_format / k_ascii_parse) on the synthetic workload and check the round trip."""
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
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
b = press.DeviceBatch(np.full(n_reads, ns, dtype=np.uint64), with_stream_out=False)
b.synth()
dev = b.dev
cap = (7 * ns + 16 + 15) // 16 * 16
td = np.zeros(n_reads, dtype=[("txt_off", "<u8"), ("sig_off", "<u8"), ("txt_len", "<u4"), ("n_samples", "<u4"), ("r0", "<u4"), ("r1", "<u4")])
td["txt_off"] = np.arange(n_reads, dtype=np.uint64) * cap
td["sig_off"] = b.desc_np["sig_off"]
td["txt_len"] = cap
td["n_samples"] = ns
d_td = torch.from_numpy(td.view(np.uint8).copy()).to(dev)
text = torch.zeros(n_reads * cap + 64, dtype=torch.uint8, device=dev)
tl = torch.zeros(n_reads, dtype=torch.int32, device=dev)
st = torch.zeros(n_reads, dtype=torch.int32, device=dev)
sig2 = torch.zeros_like(b.sig)
stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def fmt():
    _lib.check(L.s5gpu_ascii_format_dev(n_reads, d_td.data_ptr(), b.sig.data_ptr(), text.data_ptr(), tl.data_ptr(), st.data_ptr(), stream()))


def t(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]


ms_f = t(fmt)
assert int(st.abs().sum()) == 0
lens = tl.cpu().numpy().astype(np.int64)
chars = int(lens.sum())
# parse descriptors: same slots, actual lengths
td2 = td.copy()
td2["txt_len"] = lens
d_td2 = torch.from_numpy(td2.view(np.uint8).copy()).to(dev)


def parse():
    _lib.check(L.s5gpu_ascii_parse_dev(n_reads, d_td2.data_ptr(), text.data_ptr(), sig2.data_ptr(), st.data_ptr(), stream()))


ms_p = t(parse)
if not os.environ.get("S5GPU_ASCII_DBG"):
    assert int(st.abs().sum()) == 0
# round trip (stride padding is never written: compare sample ranges)
stride = (ns + 7) // 8 * 8
a = b.sig[: n_reads * stride].view(n_reads, stride)[:, :ns]
c = sig2[: n_reads * stride].view(n_reads, stride)[:, :ns]
assert os.environ.get("S5GPU_ASCII_DBG") or torch.equal(a, c)
byt = chars + 2 * n_reads * ns
print("reads %d x %d samples, %.2f chars/sample" % (n_reads, ns, chars / (n_reads * ns)))
print("format: %.3f ms  %.1f GB/s (text+signal bytes)  %.2f Mreads/s" % (ms_f, byt / ms_f / 1e6, n_reads / ms_f / 1e3))
print("parse : %.3f ms  %.1f GB/s (text+signal bytes)  %.2f Mreads/s" % (ms_p, byt / ms_p / 1e6, n_reads / ms_p / 1e3))
