"""bench.py / bench_e2e.py host logic that needs no GPU: the modules import, the traffic stamp only answers for the sources it was collected on, the
e2e helpers pick a directory with room and compare files."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_modules_import_and_traffic_stamp_follows_the_sources(tmp_path, monkeypatch):
    import bench
    import bench_e2e  # noqa: F401

    sha = bench.csrc_sha256()
    assert len(sha) == 16
    entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    kernels = {e["kernel"] for e in entries}
    assert {"k_encode_stream", "k_inflate_par_np", "k_pack+k_deflate_staged"} <= kernels
    for e in entries:                      # a figure is quoted only for the sources it was measured on
        got, src = bench.pmc_traffic(e["kernel"], e["samples_per_read"], 1000)
        if e["csrc_sha256"] == sha:
            assert got == int(e["hbm_bytes_per_read"] * 1000) and src["csrc_sha256"] == sha
        else:
            assert got is None and src is None
    assert bench.pmc_traffic("no such kernel", 4000, 1) == (None, None)


def test_e2e_helpers(tmp_path):
    import bench_e2e as E

    assert E.pick_dir(1 << 20) in ("/dev/shm", "/tmp")
    assert E.pick_dir(1 << 62) is None
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    a.write_bytes(b"x" * 100000)
    b.write_bytes(b"x" * 100000)
    c.write_bytes(b"x" * 99999 + b"y")
    assert E._same_file(a, b, block=4096) and not E._same_file(a, c, block=4096) and E._same_file(a, a)
    st = E._stamps("s5view[t]    0.232  device ready\nnoise\ns5view[t]    1.674  last write\n")
    assert st == {"device ready": 0.232, "last write": 1.674}
