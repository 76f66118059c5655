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
# (record press, signal press, output, K, workers, env): the per-record pipeline of round 1 next to the chunked one
runs = [("none", "none", "/tmp/e2e_raw.blow5", 4096, 2, {})]
for K, W in ((4096, 0), (65536, 2)):
    runs.append(("zlib", "svb-zd", "/tmp/e2e_z_%d_%d.blow5" % (K, W), K, W, {"S5VIEW_PER_RECORD": "1"}))
for W, R, C in ((1, 1, 64), (2, 4, 64), (3, 4, 64), (3, 8, 64), (3, 4, 128), (4, 8, 32)):
    runs.append(("zlib", "svb-zd", "/tmp/e2e_c_%d_%d_%d.blow5" % (W, R, C), 4096, W, {"S5VIEW_READERS": str(R), "S5VIEW_CHUNK_MB": str(C)}))
# SLOW5 text in (the conversion BASELINE configs[0] names): the .slow5 twin of the same reads is printed first (BLOW5 -> SLOW5, per-record
# pipeline), then converted by the per-record pipeline and by the chunked one
runs.append(("txt", "txt", "/tmp/e2e_txt_pr.slow5", 4096, 2, {"S5VIEW_PER_RECORD": "1"}))
runs.append(("txt", "txt", "/tmp/e2e_txt.slow5", 4096, 2, {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "32"}))
runs.append(("txt", "txt", "/tmp/e2e_txt3.slow5", 4096, 3, {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "16"}))
runs.append(("zlib", "svb-zd", "/tmp/e2e_t_pr.blow5", 4096, 2, {"S5VIEW_PER_RECORD": "1", "IN": "/tmp/e2e_txt.slow5"}))
for W, R, C in ((1, 4, 32), (2, 4, 32), (2, 8, 32), (3, 8, 64), (3, 8, 128)):
    runs.append(("zlib", "svb-zd", "/tmp/e2e_t_%d_%d_%d.blow5" % (W, R, C), 4096, W, {"S5VIEW_READERS": str(R), "S5VIEW_CHUNK_MB": str(C), "IN": "/tmp/e2e_txt.slow5"}))
ref_out = None
import re
for (rm, sm, dst, K, W, env) in runs:
    inp = env.pop("IN", None) or (src if rm == "none" else "/tmp/e2e_raw.blow5")
    if rm == "txt":
        rm, sm = "none", "none"
    t0 = time.perf_counter()
    r = subprocess.run([exe, inp, dst, rm, sm, str(K), str(W)], capture_output=True, text=True, env=dict(os.environ, **env))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    m = re.search(r"chunked pipeline(?: \(SLOW5 text (?:in|out)\))?: ([0-9.]+) s", r.stderr)
    inner = float(m.group(1)) if m else None
    print("s5view %s -> (%s,%s) %s workers=%d: %d reads, whole process %.2f s = %.2f GB/s raw signal%s  [in %.0f MB, out %.0f MB]"
          % (os.path.basename(inp), rm, sm, ("per-record pipeline K=%d%s" % (K, " (serial phases)" if W == 0 else "")) if "S5VIEW_PER_RECORD" in env or "S5VIEW_CHUNK_MB" not in env else
             "chunked pipeline (%s MB chunks, %s pread threads)" % (env["S5VIEW_CHUNK_MB"], env["S5VIEW_READERS"]), W, n_reads, dt, raw_gb / dt,
             "; first read to last write %.3f s = %.2f GB/s" % (inner, raw_gb / inner) if inner else "", os.path.getsize(inp) / 1e6, os.path.getsize(dst) / 1e6))
    sys.stdout.flush()
    if dst.endswith(".slow5"):   # text out: the same bytes whichever pipeline printed them
        import hashlib
        hsh = hashlib.sha256(open(dst, "rb").read()).hexdigest()
        ref_txt = globals().setdefault("ref_txt", hsh)
        assert hsh == ref_txt, "text output differs between pipeline settings"
        if dst != "/tmp/e2e_txt.slow5":
            os.remove(dst)
    if rm == "zlib":   # the pipeline must not change a byte
        data = open(dst, "rb").read()
        if ref_out is None:
            ref_out = data
        assert data == ref_out, "output differs between pipeline settings"
        os.remove(dst)
