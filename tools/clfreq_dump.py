#!/usr/bin/env python3
"""Dump the code-length-alphabet histograms (S.clfreq) of the fused kernel for the synthetic workload: S5GPU_DEBUG_STAGE=41.
Used once to derive / evaluate a static code-length code; not part of the product."""
import os, sys
os.environ["S5GPU_DEBUG_STAGE"] = "41"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

_lib.check(_lib.lib().s5gpu_init(0))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
b = press.DeviceBatch(np.full(n_reads, ns, dtype=np.uint64), with_stream_out=False)
b.synth()
b.encode()
torch.cuda.synchronize()
slots = b.slots.cpu().numpy()
off = b.desc_np["out_off"].astype(np.int64)
fr = np.stack([slots[o + 16:o + 16 + 80].view(np.uint32) for o in off])
tot = fr[:, :19].sum(0).astype(np.float64)
print("mean clfreq per read:", np.round(tot / n_reads, 2).tolist())
print("mean dynamic CL bits per read (code + extra):", fr[:, 19].mean())
# optimal static code for the aggregate (Huffman limited to 7 bits via simple heuristic: package-merge-free, use heapq then clamp)
import heapq
items = [(f, [s]) for s, f in enumerate(tot) if f > 0]
lens = [0] * 19
h = [(f, i, syms) for i, (f, syms) in enumerate(items)]
heapq.heapify(h)
cnt = len(h)
while len(h) > 1:
    a = heapq.heappop(h); c = heapq.heappop(h)
    for s in a[2] + c[2]:
        lens[s] += 1
    cnt += 1
    heapq.heappush(h, (a[0] + c[0], cnt, a[2] + c[2]))
print("huffman lens of the aggregate:", lens, "max", max(lens))
extra = np.array([0] * 16 + [2, 3, 7])
for name, L in (("aggregate-huffman", lens),):
    L = np.array(L)
    bits = (fr[:, :19] * (L + extra)).sum(1)
    print(name, "mean bits/read:", bits.mean(), "vs dynamic", fr[:, 19].mean(), "loss bytes/read", (bits.mean() - fr[:, 19].mean()) / 8)
np.save(os.path.join(ROOT, "gpurun_out", "clfreq.npy"), fr)
