#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_encode_fused (lane 0's timeline, summed over workgroups).
Builds the -DS5_PROFILE variant of the library; never used by the product path."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "slow5tools_amd", "csrc")
PROF_LIB = os.path.join(ROOT, "slow5tools_amd", "libslow5gpu_prof.so")

NAMES = {14: "code-length code: merge rounds (wave 0)", 15: "code-length code: depths + lengths (wave 0)",
         0: "svb-zd encode + pack (HBM->LDS payload)", 1: "break mask + 2 scans + adler partials", 2: "tokenise + histogram",
         3: "lit/len rank sort", 4: "lit/len huffman merge (1 lane)", 5: "depths + limit + lengths", 6: "code-length canonical codes + header cost (wave 0)",
         7: "hlit + code-length RLE (wave 0)", 8: "code-length code: sort (wave 0)", 9: "cost compare", 10: "header emit",
         11: "token bit totals + scan", 12: "token pack", 13: "whole zlib_compress_fused (phases 1-12 + flush)"}


def main():
    if "--build" in sys.argv or not os.path.exists(PROF_LIB):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DS5_PROFILE", "-shared",
                               os.path.join(CSRC, "kernels.hip"), os.path.join(CSRC, "host_api.hip"), "-o", PROF_LIB])
    os.environ["S5GPU_LIB"] = PROF_LIB
    import numpy as np
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0))
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    n_reads = int(nums[0]) if nums else 200000
    n = 4000
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), with_stream_out=False)
    b.synth()
    b.encode()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    L.s5gpu_prof_read(buf, 1)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    b.encode()
    t1.record()
    torch.cuda.synchronize()
    L.s5gpu_prof_read(buf, 0)
    wgs = buf[31]
    tot = sum(buf[k] for k in list(range(13)) + [14, 15])
    print("k_encode_fused (profiled build): %d reads x %d samples, %.2f ms" % (n_reads, n, t0.elapsed_time(t1)))
    print("%-45s %12s %7s" % ("phase", "cycles/WG", "share"))
    for k in list(range(13)) + [14, 15, 13]:
        print("%-45s %12.0f %6.1f%%" % (NAMES[k], buf[k] / max(wgs, 1), 100.0 * buf[k] / max(tot, 1)))
    print("%-45s %12.0f" % ("sum of phases 0-12 (lane-0 timeline, shader cycles)", tot / max(wgs, 1)))


if __name__ == "__main__":
    main()
