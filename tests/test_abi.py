"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports every symbol that
include/*.h declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:s5gpu|slow5)_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from slow5tools_amd import _lib

    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    headers = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert "slow5gpu.h" in headers
    missing = []
    for h in headers:
        for sym in _declared_symbols(h):
            if not hasattr(lib, sym):
                missing.append((h, sym))
    assert not missing, missing


def test_python_binding_lists_the_same_symbols():
    from slow5tools_amd import _lib

    assert set(_lib.EXPORTS) <= set(_declared_symbols("slow5gpu.h"))


def test_bounds_are_consistent(lib):
    from slow5tools_amd import press

    for n in (0, 1, 4000, 100000):
        for rec in (0, 1):
            for sig in (0, 1):
                p = lib.s5gpu_payload_bound(n, 74, 10, sig)
                s = lib.s5gpu_slot_bound(n, 74, 10, rec, sig)
                assert s % 16 == 0 and s >= p + 8 + 16
    d, tot = press.make_read_desc([4000, 5, 0], 74, [0, 3, 9], 1, 1)
    assert list(d["sig_off"]) == [0, 4000, 4008] and tot["samples"] == 4008
    assert all(int(o) % 16 == 0 for o in d["out_off"])


def test_no_cpu_fallback_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from slow5tools_amd import press

    rc = lib.s5gpu_init(0)
    assert rc != 0
    assert b"no HIP device" in lib.s5gpu_last_error() or b"failed" in lib.s5gpu_last_error()
    with pytest.raises(press.S5GpuError):
        press.encode_records([[1, 2, 3]], [press.pack_hdr("r", 0, 1.0, 2.0, 3.0, 4.0)])


def test_struct_layouts_match_header():
    from slow5tools_amd import _lib

    assert C.sizeof(_lib.EncodeArgs) == 80
    assert C.sizeof(_lib.DecodeArgs) == 56
