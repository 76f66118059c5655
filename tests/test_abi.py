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


def test_struct_layouts_match_header(tmp_path):
    """the ctypes mirrors against what gcc makes of include/slow5gpu.h: sizes, and the offsets of the fields added in round 3"""
    import subprocess

    from slow5tools_amd import _lib

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "slow5gpu.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(s5gpu_encode_args_t), sizeof(s5gpu_decode_args_t), offsetof(s5gpu_decode_args_t, flags), offsetof(s5gpu_decode_args_t, desc), '
                   'offsetof(s5gpu_decode_args_t, payload_bytes), offsetof(s5gpu_decode_args_t, max_pay_cap), sizeof(s5gpu_rec_desc_t));return 0;}\n')
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "sz")])
    got = [int(x) for x in subprocess.check_output([str(tmp_path / "sz")], text=True).split()]
    D = _lib.DecodeArgs
    assert got == [C.sizeof(_lib.EncodeArgs), C.sizeof(D), D.flags.offset, D.desc.offset, D.payload_bytes.offset, D.max_pay_cap.offset, _lib.REC_DESC.itemsize]
    assert C.sizeof(_lib.EncodeArgs) == 80 and C.sizeof(D) == 72 and D.flags.offset == 12 and D.desc.offset == 16


SURVEY_8B_SYMBOLS = ["slow5_press_init", "slow5_press_free", "slow5_rec_to_mem", "slow5_rec_fwrite", "slow5_rec_depress_parse", "slow5_decode",
                     "slow5_get_next_mem", "slow5_get_next_bytes", "slow5_get", "slow5_rec_free", "slow5_hdr_fwrite", "slow5_eof_fwrite",
                     "slow5_set_log_level", "slow5_set_exit_condition", "slow5_set_skip_rid", "slow5_ptr_compress", "slow5_ptr_compress_solo",
                     "slow5_ptr_depress", "slow5_ptr_depress_solo", "slow5_open", "slow5_open_with", "slow5_close", "slow5_idx_load", "slow5_idx_unload",
                     "slow5_idx_create", "s5gpu_init", "s5gpu_init_mask", "s5gpu_encode_batch", "s5gpu_decode_batch", "s5gpu_shutdown"]


def test_every_boundary_symbol_of_survey_8b_is_exported():
    import subprocess

    from slow5tools_amd import _lib

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = [s for s in SURVEY_8B_SYMBOLS if s not in exported]
    assert not missing, missing
    assert "slow5_errno" in exported                      # the thread-local error code (TLS symbol)


@pytest.mark.parametrize("compiler", [["gcc", "-std=c11"], ["g++", "-std=c++11", "-x", "c++"]])
def test_integration_hunk_compiles_next_to_slow5lib_names(compiler, tmp_path):
    """the patched src/view.c:292 hunk includes include/slow5gpu_hooks.h beside a header that defines
    slow5lib's own names (tests/compile_check/slow5/slow5.h, a stand-in: the submodule is absent).  slow5_compat.h could not
    be included there — both would define struct slow5_rec, enum slow5_press_method, ... — the hooks header declares none."""
    import subprocess

    cc = os.path.join(ROOT, "tests", "compile_check")
    cmd = compiler + ["-Wall", "-Wextra", "-Werror", "-c", "-I", cc, "-I", os.path.join(ROOT, "include"), os.path.join(cc, "view_patch.c"),
                      "-o", str(tmp_path / "view_patch.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and the two headers really do clash, which is why the hooks header exists
    clash = tmp_path / "clash.c"
    clash.write_text("#include <slow5/slow5.h>\n#include <slow5_compat.h>\n")
    r = subprocess.run(["gcc", "-std=c11", "-c", "-I", cc, "-I", os.path.join(ROOT, "include"), str(clash), "-o", str(tmp_path / "clash.o")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "redefinition" in r.stderr


@pytest.mark.parametrize("compiler", [["gcc", "-std=c11"], ["g++", "-std=c++11", "-x", "c++"]])
def test_integration_hunk_refuses_a_slow5lib_with_other_enum_values(compiler, tmp_path):
    """the hooks take slow5lib's enum values as plain ints, and the values are recalled, not read (the submodule is absent): the
    hunk carries SLOW5_GPU_HOOK_CHECK_ENUMS, so a slow5lib whose enums are ordered differently stops the build"""
    import subprocess

    cc = os.path.join(ROOT, "tests", "compile_check")
    alt = tmp_path / "slow5"
    alt.mkdir()
    txt = open(os.path.join(cc, "slow5", "slow5.h")).read()
    swapped = txt.replace("SLOW5_COMPRESS_SVB_ZD, SLOW5_COMPRESS_ZSTD", "SLOW5_COMPRESS_ZSTD, SLOW5_COMPRESS_SVB_ZD")
    assert swapped != txt
    (alt / "slow5.h").write_text(swapped)
    cmd = compiler + ["-c", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"), os.path.join(cc, "view_patch.c"), "-o", str(tmp_path / "v.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "mis-map codecs" in r.stderr, r.stderr


def test_hooks_header_declares_no_slow5lib_name():
    txt = open(os.path.join(ROOT, "include", "slow5gpu_hooks.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name in ("struct slow5_rec", "enum slow5_press_method", "slow5_press_method_t", "struct slow5_file", "slow5_aux_meta", "slow5_fmt"):
        assert name not in code, name


def test_pinned_host_memory_has_one_allocator():
    """Round 5 found D2H copies that never reached a SMALL pinned buffer allocated while another host thread ran its first kernels; the
    library's cure is that every pinned byte comes from s5_pinned_alloc (host_api.hip), which never pins less than 2 MiB at a time
    (tools/hw_probe/pinned_small_d2h.hip is the stand-alone reproducer, profiles/r06_pinned_small_d2h.txt what it showed).  A
    hipHostMalloc anywhere else in the product sources re-opens the fault, silently: this test is the guard."""
    pat = re.compile(r"\b(hipHostMalloc|hipHostAlloc|hipMallocHost|hipHostRegister)\s*\(")
    hits = []
    for d in ("slow5tools_amd/csrc", "examples", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if not f.endswith((".hip", ".h", ".c", ".cpp")):
                    continue
                for ln, line in enumerate(open(os.path.join(root, f), errors="replace"), 1):
                    code = line.split("//")[0]
                    if pat.search(code):
                        hits.append("%s:%d" % (os.path.relpath(os.path.join(root, f), ROOT), ln))
    assert len(hits) == 1 and hits[0].startswith("slow5tools_amd/csrc/host_api.hip:"), hits
    src = open(os.path.join(ROOT, "slow5tools_amd/csrc/host_api.hip")).read()
    body = src[src.index("hipError_t s5_pinned_alloc("):]
    body = body[: body.index("\n}") + 2]
    assert "hipHostMalloc" in body and "S5_PIN_MIN" in body
