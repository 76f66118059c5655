"""slow5tools_amd — MI355X-native BLOW5 record press path (svb-zd + zlib), see DESIGN.md.

Only what the hot path needs lives here:
  csrc/      HIP kernels (gfx950) + the C ABI of include/slow5gpu.h + the slow5lib-compatible C layer
  _lib.py    ctypes binding (fails loudly without the built library / a GPU; no CPU fallback)
  press.py   host-side mirror: batch encode / decode, device-resident batches for bench.py
"""
from ._lib import REC_NONE, REC_ZLIB, REC_ZSTD, SIG_NONE, SIG_SVB_ZD, S5GpuError  # noqa: F401
