#!/usr/bin/env python3
"""File-backed `get` (BASELINE configs[4] as SURVEY 8(d) states it): a 1 M-read BLOW5 file (zlib + svb-zd) + its .idx, 100 k uniformly random
read ids (seed 1), batches of K = 4096 — index build, index load, preads, ONE GPU call per batch (examples/s5get.c).  Also the CPU twin of
the decode (oracle, get --benchmark shape) on the same ids for scale.   python tools/get_bench.py [reads] [ids]"""
import os, re, struct, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from slow5tools_amd import _lib, press
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n = 4000
_lib.check(_lib.lib().s5gpu_init(0))
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode_stream(); torch.cuda.synchronize(); assert b.stream_ok()
stream, off = b.stream_bytes()
hdr_text = b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
head = bytearray(64)
head[:6] = b"BLOW5\x01"; head[6:9] = bytes([0, 2, 0]); head[9] = 1; head[10:14] = struct.pack("<I", 1); head[14] = 1
src = "/tmp/get_in.blow5"
with open(src, "wb") as f:
    f.write(head); f.write(struct.pack("<I", len(hdr_text))); f.write(hdr_text); f.write(stream); f.write(b"5WOLB")
del b, stream
torch.cuda.empty_cache()
view, get = os.path.join(ROOT, "slow5tools_amd", "s5view"), os.path.join(ROOT, "slow5tools_amd", "s5get")
if os.path.exists(src + ".idx"):
    os.remove(src + ".idx")
print("file: %d reads x %d samples, %.2f GB (zlib + svb-zd)" % (n_reads, n, os.path.getsize(src) / 1e9))
t0 = time.perf_counter(); r = subprocess.run([view, "--index", src], capture_output=True, text=True); dt = time.perf_counter() - t0
assert r.returncode == 0, r.stderr
print("index build (slow5_idx_create: chunks of the file, record heads inflated on the GPU): whole process %.2f s = %.2f M records/s, .idx %.1f MB" % (dt, n_reads / dt / 1e6, os.path.getsize(src + ".idx") / 1e6))
ids = "/tmp/get_ids.txt"
r = subprocess.run([get, "--random", src, str(n_ids), "1", ids], capture_output=True, text=True); assert r.returncode == 0, r.stderr
for label, args in (("get --benchmark (fetch + decode)", ["--benchmark", src, ids, "4096", "8"]),
                    ("get --benchmark, 16 pread threads", ["--benchmark", src, ids, "4096", "16"]),
                    ("get -> zlib + svb-zd file", [src, ids, "/tmp/get_out.blow5", "zlib", "svb-zd", "4096", "8"]),
                    ("get -> uncompressed file", [src, ids, "/tmp/get_out2.blow5", "none", "none", "4096", "8"])):
    t0 = time.perf_counter(); r = subprocess.run([get] + args, capture_output=True, text=True); dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    print("%-36s whole process %.2f s | %s" % (label, dt, " | ".join(l.replace("s5get: ", "") for l in r.stderr.strip().splitlines())))
    sys.stdout.flush()
for f in ("/tmp/get_out.blow5", "/tmp/get_out2.blow5"):
    if os.path.exists(f): os.remove(f)
