#!/usr/bin/env python3
"""svb-zd stage alone on 1 M resident reads: slots + compaction (k_svbzd_encode + k_compact) against the one-pass blob stream
(k_svbzd_stream).  S5GPU_LIB selects a variant build (tools/variant.sh NAME -DS5_SVS_G=4 ...)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
L = _lib.lib()
_lib.check(L.s5gpu_init(0))
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64))
b.synth()


def run(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_enc = run(b.svbzd_encode)
t_two = run(lambda: (b.svbzd_encode(), b.compact()))
torch.cuda.synchronize()
want_off = b.rec_off.clone()
tot = int(want_off[n_reads].item())
want = b.stream_out[:tot].clone()
b.stream_out.zero_()
t_one = run(b.svbzd_encode_stream)
same = b.stream_ok() and torch.equal(b.rec_off, want_off) and torch.equal(b.stream_out[:tot], want)
alg = 2 * n * n_reads + tot
print("%s: %d reads x %d: k_svbzd_encode %.3f ms; + k_compact %.3f ms (%.0f GB/s raw); k_svbzd_stream %.3f ms (%.0f GB/s raw, %.2f TB/s of 2N+S) identical %s"
      % (os.path.basename(os.environ.get("S5GPU_LIB", "default")), n_reads, n, t_enc, t_two, 2 * n * n_reads / t_two / 1e6, t_one, 2 * n * n_reads / t_one / 1e6, alg / t_one / 1e9, same))
