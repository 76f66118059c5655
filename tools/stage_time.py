#!/usr/bin/env python3
"""Time k_encode_fused (and, with S5GPU_DEBUG_STAGE=1, only its svb-zd + pack stage) with torch events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

_lib.check(_lib.lib().s5gpu_init(0))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = press.DeviceBatch(np.full(n_reads, 4000, dtype=np.uint64), with_stream_out=False)
b.synth()
for _ in range(2):
    b.encode()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.encode(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("S5GPU_DEBUG_STAGE=%s  encode launch: min %.3f ms  median %.3f ms  (%d reads)" % (os.environ.get("S5GPU_DEBUG_STAGE", "0"), min(ts), sorted(ts)[2], n_reads))
