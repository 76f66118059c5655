#!/usr/bin/env python3
"""Where the time of `s5view in.slow5 out.blow5` goes: the stage summary of S5VIEW_TIMING and the library's trace of the first chunk calls
(S5GPU_TRACE), over slot counts and chunk sizes; 1 M reads in /dev/shm.   python tools/e2e_probe.py [reads]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench_e2e as E
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = 4000
raw_gb = n_reads * n * 2 / 1e9
work = "/dev/shm/s5probe_%d" % os.getpid()
os.makedirs(work, exist_ok=True)
exe = os.path.join(ROOT, "slow5tools_amd", "s5view")
def run(inp, out, workers, env, trace=False):
    e = dict(os.environ, S5VIEW_TIMING="1", **env)
    if trace: e["S5GPU_TRACE"] = "1"
    t0 = time.perf_counter()
    r = subprocess.run([exe, inp, out, "zlib", "svb-zd", "4096", str(workers)], capture_output=True, text=True, env=e)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-500:]
    return dt, r.stderr
try:
    blow5, slow5, out = work + "/in.blow5", work + "/in.slow5", work + "/out.blow5"
    E.write_blow5(blow5, L, _lib, press, torch, "cuda:0", n_reads, n)
    E.view_run(blow5, slow5, "none", "none", 3, {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "32"}, raw_gb)
    del L
    for inp, label, readers in ((slow5, ".slow5 -> BLOW5", 8), (blow5, "BLOW5 -> BLOW5", 4)):
        dt, err = run(inp, out, 3, {"S5VIEW_READERS": str(readers), "S5VIEW_CHUNK_MB": "32"}, trace=True)
        print("==== %s, 3 workers, 32 MB chunks, 6 slots, with the library trace: whole process %.3f s" % (label, dt))
        lines = err.splitlines()
        print("\n".join(l for l in lines if "s5view" in l))
        print("\n".join([l for l in lines if "s5gpu[trace]" in l][:40]))
        for workers, chunk, slots, rd in ((3, 32, 4, readers), (3, 32, 6, readers), (3, 32, 8, readers), (3, 32, 12, readers), (3, 16, 8, readers), (3, 16, 12, readers), (3, 8, 16, readers), (4, 16, 12, readers),
                                           (2, 32, 6, readers), (3, 32, 8, 2 * readers), (3, 32, 8, 2)):
            dt, err = run(inp, out, workers, {"S5VIEW_READERS": str(rd), "S5VIEW_CHUNK_MB": str(chunk), "S5VIEW_SLOTS": str(slots)})
            m = re.search(r"chunked pipeline[^:]*: ([0-9.]+) s", err)
            st = [l for l in err.splitlines() if "stages" in l]
            print("%-16s workers %d chunk %2d MB slots %2d readers %2d: whole %.3f s = %5.2f GB/s, first read to last write %s s | %s" % (label, workers, chunk, slots, rd, dt, raw_gb / dt, m.group(1) if m else "?", st[0].split("stages")[1] if st else ""), flush=True)
finally:
    import shutil
    shutil.rmtree(work, ignore_errors=True)
