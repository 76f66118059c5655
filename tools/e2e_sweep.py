#!/usr/bin/env python3
"""s5view end to end on 1 M-read files in /dev/shm over the pipeline's knobs (GPU workers, pread threads, chunk size, writer threads and mode):
the table bench_e2e.py's fixed choices come from.   python tools/e2e_sweep.py [reads]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench_e2e as E
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = 4000
raw_gb = n_reads * n * 2 / 1e9
work = "/dev/shm/s5sweep_%d" % os.getpid()
os.makedirs(work, exist_ok=True)
try:
    blow5, slow5, out = work + "/in.blow5", work + "/in.slow5", work + "/out.blow5"
    E.write_blow5(blow5, L, _lib, press, torch, "cuda:0", n_reads, n)
    r = E.view_run(blow5, slow5, "none", "none", 3, {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "32"}, raw_gb)
    print("BLOW5 -> .slow5 (3 workers, writers default): whole %.3f s = %.2f GB/s, inner %.3f s" % (r["whole_process_s"], r["GB_per_s_whole_process"], r["first_read_to_last_write_s"]), flush=True)
    for inp, label, readers in ((slow5, ".slow5 -> BLOW5", 8), (blow5, "BLOW5 -> BLOW5", 4)):
        for workers, chunk, writers, mode in ((2, 32, 1, "pwrite"), (2, 32, 4, "pwrite"), (3, 32, 4, "pwrite"), (3, 64, 4, "pwrite"), (3, 32, 8, "pwrite"), (3, 32, 4, "mmap"), (3, 32, 8, "mmap"),
                                               (4, 32, 8, "pwrite"), (3, 16, 4, "pwrite")):
            for rd in ((readers, 16) if workers == 3 and chunk == 32 and writers == 4 and mode == "pwrite" else (readers,)):
                env = {"S5VIEW_READERS": str(rd), "S5VIEW_CHUNK_MB": str(chunk), "S5VIEW_WRITERS": str(writers), "S5VIEW_WRITE_MODE": mode}
                r = E.view_run(inp, out, "zlib", "svb-zd", workers, env, raw_gb)
                t = r["timeline_s"]
                print("%-16s workers %d chunk %2d MB readers %2d writers %d %-6s: whole %.3f s = %5.2f GB/s | first read to last write %.3f s = %5.2f GB/s | device ready %.3f, first chunk through GPU %.3f, last write %.3f, shut down %.3f"
                      % (label, workers, chunk, rd, writers, mode, r["whole_process_s"], r["GB_per_s_whole_process"], r["first_read_to_last_write_s"], r["GB_per_s_first_read_to_last_write"],
                         t.get("device ready", 0), t.get("first chunk through the GPU call", 0), t.get("last write", 0), t.get("library shut down", 0)), flush=True)
finally:
    import shutil
    shutil.rmtree(work, ignore_errors=True)
