"""The host batch call at the reference's batch sizes under the library's two host-side options (round 6):
host_turns (concurrent batches take turns at packing and uploading) x d2h_kernel_copy (results return by a kernel's stores instead of the
copy engines): the synchronous arena call and two batches in flight (s5gpu_encode_batch_submit / s5gpu_batch_wait), GB/s of raw signal.
python tools/hook_modes.py [K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_e2e
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
Ks = [int(x) for x in sys.argv[1:]] or [4096, 10000, 65536]
for turns in (0, 1):
    for kc in (0, 1):
        _lib.check(L.s5gpu_set_option(b"host_turns", turns), "opt"); _lib.check(L.s5gpu_set_option(b"d2h_kernel_copy", kc), "opt")
        for K in Ks:
            a = bench_e2e._pcie_one(L, _lib, press, K, 4000, 6, arena=True)
            t = bench_e2e._pcie_two_in_flight(L, _lib, press, K, 4000, 48 if K <= 10000 else 12)
            t2 = bench_e2e._pcie_two_in_flight(L, _lib, press, K, 4000, 48 if K <= 10000 else 12)
            print("host_turns=%d d2h_kernel_copy=%d K=%6d: synchronous arena call %6.2f GB/s; two in flight %6.2f / %6.2f GB/s (%.3f ms per batch)" %
                  (turns, kc, K, a["GB_per_s"], t["GB_per_s"], t2["GB_per_s"], t2["ms_per_batch"]), flush=True)
