"""several GPUs behind ONE host batch call.  s5gpu_init_mask names the devices; the host batch calls cut a batch
into one contiguous index range per device (as work_db cuts it per thread, /root/reference/src/thread.c:76-90) and every
result lands in the caller's own slot, so the ordered write loop (/root/reference/src/view.c:296-299) sees the same bytes as with
one device.  The GPU box has one MI355X: S5GPU_ALIAS_DEVICES=1 lets device ordinals past the last one wrap around, so the split,
the per-device contexts and the concurrent host threads run exactly as on a node (each in a fresh process: the device set is
process state)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CODE = r"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import oracle_bind as ob
from slow5tools_amd import _lib, press
L = _lib.lib()
mask = %(mask)d
_lib.check(L.s5gpu_init_mask(mask), "init_mask")
assert L.s5gpu_devices_in_use() == bin(mask).count("1")
_lib.check(L.s5gpu_set_option(b"multi_min_per_device", 8), "opt")
rng = np.random.default_rng(11)
n = %(n)d
lens = rng.integers(1, 9000, n); lens[::97] = 0; lens[5] = 70000
sigs = [ob.synth_read(0x5105, i, int(l)) if l else np.zeros(0, np.int16) for i, l in enumerate(lens)]
hdrs = [press.pack_hdr(ob.synth_read_id(i), i %% 3, 8192.0, 23.0, 1467.61, 4000.0) for i in range(n)]
recs = press.encode_records(sigs, hdrs)
for i, r in enumerate(recs):
    rec, keep = ob.make_rec(ob.synth_read_id(i), i %% 3, 8192.0, 23.0, 1467.61, 4000.0, sigs[i])
    assert int.from_bytes(r[:8], "little") == len(r) - 8
    assert zlib.decompress(r[8:]) == ob.rec_pack(rec, ob.SIG_SVB_ZD), i
dec = press.decode_records([r[8:] for r in recs])
assert all(d["status"] == 0 and np.array_equal(d["signal"], sigs[i]) and d["read_group"] == i %% 3 for i, d in enumerate(dec))
# a corrupt record in the second device's range is reported in its own slot, the others still decode
bad = [r[8:] for r in recs]; k = n - 3; bad[k] = bad[k][:20] + bytes([bad[k][20] ^ 0x55]) + bad[k][21:]
dec2 = press.decode_records(bad, raise_on_error=False)
assert dec2[k]["status"] != 0 and all(d["status"] == 0 for i, d in enumerate(dec2) if i != k)
# the view worker (decode + re-encode, device-resident) through the same split: zlib+svb -> none+none -> back
import ctypes as C
vp = C.c_void_p
def recompress(rs, f, t, rg=None):
    m = len(rs)
    bufs = [C.create_string_buffer(r, len(r)) for r in rs]
    ptr = (vp * m)(*[C.addressof(b) for b in bufs]); ln = (C.c_size_t * m)(*[len(r) for r in rs])
    out = (vp * m)(); ol = (C.c_size_t * m)(); st = (C.c_int32 * m)()
    rgp = (C.c_uint32 * m)(*rg) if rg is not None else None
    L.s5gpu_recompress_batch.argtypes = [C.c_uint32, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]
    _lib.check(L.s5gpu_recompress_batch(m, ptr, ln, f[0], f[1], t[0], t[1], rgp, 0, out, ol, st), "recompress")
    libc = C.CDLL(None); libc.free.argtypes = [vp]
    res = [C.string_at(out[i], ol[i]) for i in range(m)]
    for i in range(m): libc.free(out[i])
    return res
plain = recompress([r[8:] for r in recs], (1, 1), (0, 0), rg=[7] * n)
for i, p in enumerate(plain):
    rec, keep = ob.make_rec(ob.synth_read_id(i), 7, 8192.0, 23.0, 1467.61, 4000.0, sigs[i])
    assert p[8:] == ob.rec_pack(rec, ob.SIG_NONE), i
# the chunk form of the worker (s5gpu_recompress_stream): one framed host buffer in, one contiguous stream out.  With too little
# room it must say how much the WHOLE output needs, however many devices shared the work (ADVICE round 2: the first overflowing
# share used to report only the shares up to itself, so the caller's retry could overflow again)
chunk = b"".join(recs) + bytes(64)
pos = np.cumsum([0] + [len(r) for r in recs])[:-1].astype(np.uint64) + 8
ln32 = np.array([len(r) - 8 for r in recs], dtype=np.uint32)
L.s5gpu_recompress_stream.argtypes = [C.c_uint32, vp, C.c_size_t, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_size_t, vp, vp]
def stream(cap):
    ob_ = C.create_string_buffer(max(cap, 1)); off = (C.c_uint64 * (n + 1))(); st = (C.c_int32 * n)()
    rc = L.s5gpu_recompress_stream(n, chunk, len(chunk) - 64, pos.ctypes.data, ln32.ctypes.data, 1, 1, 0, 0, None, 0, ob_, cap, off, st)
    return rc, list(off), ob_.raw
full = sum(len(p) for p in plain)
rc, off, raw = stream(full + 64)
assert rc == 0 and off[n] == full and raw[:full] == b"".join(plain[i][:8] + ob.rec_pack(ob.make_rec(ob.synth_read_id(i), i %% 3, 8192.0, 23.0, 1467.61, 4000.0, sigs[i])[0], ob.SIG_NONE) for i in range(n))
for cap in (full // 5, full // 2, full - 1):          # overflow in the first, the second and the last share
    rc, off, raw = stream(cap)
    assert rc == -3 and off[0] == full, (cap, rc, off[0], full)
# a corrupt record in the LAST share (ADVICE round 3): the call must come back with the data error of the share that holds it — not with
# a secondary code of a share in front that merely noticed — and status[] must name the record
bad_i = n - 2
broken = bytearray(chunk)
broken[int(pos[bad_i]) + 20] ^= 0x55
ob_ = C.create_string_buffer(full + 64); off = (C.c_uint64 * (n + 1))(); st = (C.c_int32 * n)()
rc = L.s5gpu_recompress_stream(n, bytes(broken), len(chunk) - 64, pos.ctypes.data, ln32.ctypes.data, 1, 1, 0, 0, None, 0, ob_, full + 64, off, st)
msg = L.s5gpu_last_error()
assert rc == -5, (rc, msg)                               # S5GPU_ERR_DATA
assert st[bad_i] != 0 and sum(1 for x in st if x) == 1, list(st)[-6:]
ndev = L.s5gpu_devices_in_use()
assert ndev == 1 or (b"device slot %%d" %% (ndev - 1)) in msg, msg
# the arena form of the worker (round 5): the same records, pointers into pinned buffers (one or more per device share), one release;
# a batch still held across s5gpu_shutdown is released without touching the drained pool
def recompress_arena(rs, f, t):
    m = len(rs)
    bufs = [C.create_string_buffer(r, len(r)) for r in rs]
    ptr = (vp * m)(*[C.addressof(b) for b in bufs]); ln = (C.c_size_t * m)(*[len(r) for r in rs])
    out = (vp * m)(); ol = (C.c_size_t * m)(); st = (C.c_int32 * m)(); h = vp()
    L.s5gpu_recompress_batch_arena.argtypes = [C.c_uint32, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.POINTER(vp)]
    _lib.check(L.s5gpu_recompress_batch_arena(m, ptr, ln, f[0], f[1], t[0], t[1], None, 0, out, ol, st, C.byref(h)), "recompress_arena")
    return [C.string_at(out[i], ol[i]) for i in range(m)], h
L.s5gpu_arena_release.argtypes = [vp]
again, h = recompress_arena([p[8:] for p in plain], (0, 0), (1, 1))
for i, r in enumerate(again):
    rec, keep = ob.make_rec(ob.synth_read_id(i), 7, 8192.0, 23.0, 1467.61, 4000.0, sigs[i])
    assert zlib.decompress(r[8:]) == ob.rec_pack(rec, ob.SIG_SVB_ZD), i
L.s5gpu_arena_release(h)
again2, h2 = recompress_arena([p[8:] for p in plain], (0, 0), (1, 1))
assert again2 == again
L.s5gpu_shutdown()
assert L.s5gpu_devices_in_use() == 0
L.s5gpu_arena_release(h2)
print("multi ok", len(b"".join(recs)))
"""


def _run(mask, n, alias):
    env = dict(os.environ)
    if alias:
        env["S5GPU_ALIAS_DEVICES"] = "1"
    r = subprocess.run([sys.executable, "-c", CODE % dict(root=ROOT, mask=mask, n=n)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "multi ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout.strip().splitlines()[-1]


def test_batch_split_over_three_logical_devices_matches_one_device():
    one = _run(0b1, 301, alias=False)
    three = _run(0b111, 301, alias=True)
    assert one == three          # same total bytes: the records are the same whichever device made them


def test_mask_naming_a_missing_device_fails_without_alias():
    import torch

    if torch.cuda.device_count() > 3:
        pytest.skip("box has the devices")
    code = ("import sys; sys.path.insert(0, %r)\nfrom slow5tools_amd import _lib\nL = _lib.lib()\n"
            "rc = L.s5gpu_init_mask(0b1001)\nassert rc != 0 and b'out of range' in L.s5gpu_last_error(), (rc, L.s5gpu_last_error())\n"
            "assert L.s5gpu_init_mask(0) != 0\nprint('ok')\n" % ROOT)
    env = dict(os.environ)
    env.pop("S5GPU_ALIAS_DEVICES", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


# ---- one process per GPU (the bench / torchrun shape): two gloo ranks, each runs the HIP path on its shard ----
WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from slow5tools_amd import _lib, press, shard
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
L = _lib.lib()
_lib.check(L.s5gpu_init(rank %% torch.cuda.device_count()), "init")
n_total, n = int(sys.argv[4]), int(sys.argv[5])
lo, hi = shard.shard_range(n_total, rank, world)
b = press.DeviceBatch(np.full(hi - lo, n, dtype=np.uint64), device="cuda:%%d" %% (rank %% torch.cuda.device_count()))
b.synth(seed=0x5105, first=lo)
shard.barrier()
b.encode_stream(); torch.cuda.synchronize()
assert b.stream_ok()
blob, off = b.stream_bytes()
shard.barrier()
open(os.path.join(sys.argv[6], "shard%%d.bin" %% rank), "wb").write(blob)
dist.destroy_process_group()
"""


def test_two_gloo_ranks_run_the_hip_path_on_their_shards(tmp_path):
    import zlib

    import numpy as np

    import oracle_bind as ob

    n_total, n, world = 1001, 4000, 2
    port = str(29700 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER % dict(root=ROOT), str(r), str(world), port, str(n_total), str(n), str(tmp_path)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    stream = b"".join(open(tmp_path / ("shard%d.bin" % r), "rb").read() for r in range(world))
    # the concatenation in rank order is the record stream of the whole index space: walk it, inflate every record with stock
    # zlib, compare a sample with the oracle's payload and every read id with its index
    pos, i = 0, 0
    while pos < len(stream):
        size = int.from_bytes(stream[pos:pos + 8], "little")
        pay = zlib.decompress(stream[pos + 8:pos + 8 + size])
        assert pay[2:38] == ob.synth_read_id(i)
        if i % 97 == 0 or i == n_total - 1:
            rec, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0, ob.synth_read(0x5105, i, n))
            assert pay == ob.rec_pack(rec, ob.SIG_SVB_ZD)
        pos += 8 + size
        i += 1
    assert i == n_total and pos == len(stream)


def test_bench_two_ranks_preflight_on_one_gpu(tmp_path):
    """The driver's N > 1 run is `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (bench.py re-executes itself
    under exactly that module when it is started bare): rehearse that launch
    with two ranks on this box's one GPU (S5BENCH_ALIAS_DEVICES=1 maps both ranks onto device 0 and swaps RCCL — which refuses two
    ranks on one device — for gloo; everything else is the production path: rendezvous, per-rank shards, barriers, MAX / SUM over
    ranks, the legs of every config).  One JSON line, n_gpus 2, both ranks seen by the all-reduce, parity true in every leg."""
    import json

    env = dict(os.environ, S5BENCH_ALIAS_DEVICES="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # round 6: PLAIN `python bench.py --gpus 2` — no launcher around it: bench.py makes its own ranks (torch.distributed.run, the driver's
    # module) when WORLD_SIZE is unset, so `--gpus N` can never end as N copies' worth of one rank with `n_gpus: 1`
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "20000", "--long-reads", "512", "--cpu-seconds", "0",
           "--get-reads", "20000", "--min-leg-seconds", "0.2", "--min-leg-steps-svb", "4", "--min-leg-steps-long", "2", "--mixed-reads", "8192"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["parity_spot_check"] is True and j["config"]["reads_per_gpu"] == 20000
    assert j["configs1"]["parity_spot_check"] is True and j["configs1"]["n_gpus"] == 2
    assert j["configs3"]["parity_spot_check"] is True and j["configs3"]["reads_rank0"] == 256 and j["configs3"]["scaling"] == "strong"
    assert j["configs4"]["roundtrip_identical"] is True and j["configs4"]["bulk_decode_one_call"]["stock_zlib_records"]["roundtrip_identical"] is True
    assert j["cpu_baseline"] is None                       # an N = 1 figure
    assert j["mixed"]["parity_spot_check"] is True and j["mixed"]["scaling"] == "weak"
    assert j["e2e"] is None and j["pcie_inclusive"] is None                   # file-to-file and host-buffer figures are N = 1 figures too
    # ... but the HOST-FED curve comes out of the same command (round 5): one fresh process, s5gpu_init_mask over the N devices (aliases of
    # device 0 here), the arena batch call at 20000 reads per call — what a patched slow5tools would do on the node
    hf = j["host_fed"]
    assert hf["devices"] == 2 and hf["dev_mask"] == 3, hf
    assert hf["batch_20000"]["arena"]["GB_per_s"] > 0 and hf["batch_20000"]["arena"]["bytes_per_sample"] < 1.1
    assert j["roofline"]["legs"]["host_fed_all_devices_GB_per_s"]["batch_20000"] == hf["batch_20000"]["arena"]["GB_per_s"]
