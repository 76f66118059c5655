#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer batch call (what a patched slow5tools view would see per batch):
host int16 signals in, one malloc'd record per read out.  Reported in DESIGN.md, never as bench `value`."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slow5tools_amd import _lib, press

L = _lib.lib()
_lib.check(L.s5gpu_init(0))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = 4000
rng = np.random.default_rng(0)
base = (500 + 30 * rng.standard_normal((1024, n))).astype(np.int16)
sig = np.ascontiguousarray(np.tile(base, (n_reads // 1024 + 1, 1))[:n_reads])
hdr = np.frombuffer(press.pack_hdr("0" * 36, 0, 8192.0, 23.0, 1467.61, 4000.0), dtype=np.uint8)
vp = C.c_void_p
sig_p = (vp * n_reads)(*[sig.ctypes.data + 2 * n * i for i in range(n_reads)])
ns = (C.c_uint64 * n_reads)(*([n] * n_reads))
hdr_p = (vp * n_reads)(*([hdr.ctypes.data] * n_reads))
hl = (C.c_uint32 * n_reads)(*([len(hdr)] * n_reads))
out = (vp * n_reads)()
ol = (C.c_size_t * n_reads)()
libc = C.CDLL(None)
libc.free.argtypes = [vp]
for it in range(3):
    t0 = time.perf_counter()
    _lib.check(L.s5gpu_encode_batch(n_reads, sig_p, ns, hdr_p, hl, None, None, 1, 1, out, ol))
    dt = time.perf_counter() - t0
    tot = sum(ol)
    for i in range(n_reads):
        libc.free(out[i])
    print("s5gpu_encode_batch: %d reads x %d samples in %.1f ms -> %.2f GB/s raw signal, %.2f M reads/s (host in, per-record malloc out; %.3f B/sample)"
          % (n_reads, n, dt * 1e3, n_reads * n * 2 / dt / 1e9, n_reads / dt / 1e6, tot / (n_reads * n)))
