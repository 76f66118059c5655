"""Where a workgroup's time goes in the LZ77 block encoder (deflate_block_lz, csrc/lz_dev.h): clock ticks of thread 0 per phase, averaged over the blocks
(variant build: tools/variant.sh lzprobe -DS5_LZPROBE [-DS5_LZ_TN=...]; S5GPU_LIB=slow5tools_amd/_variants/libs5_lzprobe.so python tools/lz_phases.py [reads] [samples])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), rec_method=press.REC_ZLIB, sig_method=press.SIG_NONE)
b.synth(); b.encode(); torch.cuda.synchronize()
z = (C.c_ulonglong * 16)()
L.s5gpu_lzprobe_read(z)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); b.encode(); e1.record(); torch.cuda.synchronize()
L.s5gpu_lzprobe_read(z)
names = ["match rounds", "parse iterations", "settled parse: lengths parked", "histograms, Adler", "code lengths + codes", "header, costs", "bit totals", "scan, pack"]
nb = max(z[15], 1); tot = sum(z[:8])
print("%d reads x %d samples: %.2f ms = %.1f GB/s; %d blocks, %.0f ticks per block; parse iterations per block %.2f" % (
    n_reads, n, e0.elapsed_time(e1), n_reads * 2 * n / e0.elapsed_time(e1) / 1e6, nb, tot / nb, z[9] / nb))
for i, nm in enumerate(names):
    print("  %-34s %9.0f ticks  %5.1f %%" % (nm, z[i] / nb, 100.0 * z[i] / max(tot, 1)))
