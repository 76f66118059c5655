#!/usr/bin/env python3
"""Time the ordered single-pass encode (k_encode_stream) against encode + compaction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

_lib.check(_lib.lib().s5gpu_init(0))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = press.DeviceBatch(np.full(n_reads, 4000, dtype=np.uint64))
b.synth()


def t(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[2]


def two_pass():
    b.encode(); b.compact()


print("encode + compact : min %.3f ms median %.3f ms" % t(two_pass))
print("single-pass stream: min %.3f ms median %.3f ms  ok=%s" % (t(b.encode_stream) + (b.stream_ok(),)))
