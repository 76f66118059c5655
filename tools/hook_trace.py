"""Where a host batch call's time goes at the reference's batch size (K = 4096): S5GPU_TRACE=1 python tools/hook_trace.py [K]
prints the library's trace points (ms since the calling thread's previous point) for a few synchronous arena calls and for a short
two-in-flight run (s5gpu_encode_batch_submit / s5gpu_batch_wait)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("S5GPU_TRACE", "1")
import bench_e2e
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
print("== synchronous arena calls", flush=True)
r = bench_e2e._pcie_one(L, _lib, press, K, 4000, 5, arena=True)
print(r, flush=True)
print("== two in flight", flush=True)
r = bench_e2e._pcie_two_in_flight(L, _lib, press, K, 4000, 8)
print(r, flush=True)
