#!/usr/bin/env python3
"""End-to-end timing of examples/s5view.c (file -> GPU batch hook -> file) on a synthetic BLOW5 file.
Reported in DESIGN.md beside the kernel-only numbers; the serial read/write phases are the reference's
(src/view.c:265-278,296-299) and bound this figure, not the GPU."""
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n = 4000
_lib.check(_lib.lib().s5gpu_init(0))
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64))
b.synth(); b.encode(); b.compact()
stream, off = b.stream_bytes()
hdr_text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
head = bytearray(64)
head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 1; head[10:14] = struct.pack("<I", 1); head[14] = 1
src = "/tmp/e2e_in.blow5"
with open(src, "wb") as f:
    f.write(head); f.write(struct.pack("<I", len(hdr_text))); f.write(hdr_text); f.write(stream); f.write(b"5WOLB")
del b
torch.cuda.empty_cache()
exe = os.path.join(ROOT, "slow5tools_amd", "s5view")
raw_gb = n_reads * n * 2 / 1e9
runs = [("none", "none", "/tmp/e2e_raw.blow5", 4096, 2)]
for K in (4096, 65536):
    for W in (0, 1, 2, 3):
        runs.append(("zlib", "svb-zd", "/tmp/e2e_z_%d_%d.blow5" % (K, W), K, W))
ref_out = None
for (rm, sm, dst, K, W) in runs:
    inp = src if rm == "none" else "/tmp/e2e_raw.blow5"
    t0 = time.perf_counter()
    r = subprocess.run([exe, inp, dst, rm, sm, str(K), str(W)], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    print("s5view %s -> (%s,%s) K=%d workers=%d%s: %d reads in %.2f s = %.2f GB/s raw signal, %.0f k reads/s  [in %.0f MB, out %.0f MB]"
          % (os.path.basename(inp), rm, sm, K, W, " (serial phases)" if W == 0 else "", n_reads, dt, raw_gb / dt, n_reads / dt / 1e3,
             os.path.getsize(inp) / 1e6, os.path.getsize(dst) / 1e6))
    if rm == "zlib":   # the pipeline must not change a byte
        data = open(dst, "rb").read()
        if ref_out is None:
            ref_out = data
        assert data == ref_out, "output differs between pipeline settings"
        os.remove(dst)
