#!/usr/bin/env python3
"""k_encode_stream on 1 M resident reads, event-timed (S5GPU_LIB selects a variant build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
_lib.check(_lib.lib().s5gpu_init(0))
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64))
b.synth()
for _ in range(3):
    b.encode_stream()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.encode_stream()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print("%s: k_encode_stream %d x %d: %s ms  (min %.3f = %.1f GB/s raw)  ok %s  %.5f B/sample"
      % (os.path.basename(os.environ.get("S5GPU_LIB", "default")), n_reads, n, " ".join("%.3f" % t for t in ts), min(ts), 2 * n * n_reads / min(ts) / 1e6, b.stream_ok(),
         int(b.rec_off[n_reads].item()) / (n * n_reads)))
