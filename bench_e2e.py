"""bench.py's `e2e` object: the conversions `north_star` sets its target on, FILE TO FILE, under the driver's clock, each next to its CPU twin.

    GPU side   examples/s5view.c / examples/s5get.c (the loops of /root/reference/src/view.c:241-323 and src/get.c:321-386 with work_db()
               replaced by the library's chunk calls), run as separate processes on files in /dev/shm:
                 .slow5 -> zlib + svb-zd BLOW5      (BASELINE configs[0]'s conversion, the one north_star's ">= 10x view" names)
                 BLOW5  -> BLOW5                    (zlib + svb-zd both sides: decode + re-encode, what `merge` / `view` do to binary input)
                 BLOW5  -> .slow5                   (made on the way: the text twin is printed by s5view itself)
                 get --benchmark / get -> file      (100 k random ids, K = 4096)
               each with the WHOLE PROCESS wall time (fork to exit: HIP start-up, pinned buffers, shutdown included) and the time from the
               first read to the last write that the program reports itself.
    CPU side   the oracle's worker behind a serial read and an ordered write (oracle/batch.c s5o_view_file / s5o_get_file: the reference's
               three phases per batch, threads created and joined per batch) on the same files, at -t <all cores> (the reference's shape) and
               at the best -t of a short sweep; on a bounded sample of the file (stated).  A reported baseline, not the target.
    PCIe       the host-buffer batch call a patched `view` would make (s5gpu_encode_batch: host int16 in, one malloc'd record per read out)
               at the full batch size — the PCIe-inclusive figure beside the device-resident `value`.
Nothing here is `value`.  The oracle is touched only for the CPU twins."""
import ctypes as C
import os
import re
import shutil
import struct
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HDR_TEXT = (b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
            b"#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")


def _exe(name):
    return os.path.join(ROOT, "slow5tools_amd", name)


def _run(cmd, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (" ".join(cmd[:3]), r.stderr[-400:]))
    return dt, r


def _stamps(stderr):
    return {m.group(2).strip(): float(m.group(1)) for m in re.finditer(r"s5view\[t\]\s+([0-9.]+)\s+(.+)", stderr)}


def pick_dir(need_bytes):
    for d in ("/dev/shm", "/tmp"):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > need_bytes * 1.15:
                return d
        except OSError:
            pass
    return None


def write_blow5(path, L, _lib, press, torch, dev, n_reads, n, chunk_reads=250_000):
    """zlib + svb-zd BLOW5 file of the synthetic reads (the GPU encoder writes it; checked against the oracle elsewhere)"""
    head = bytearray(64)
    head[:6] = b"BLOW5\x01"
    head[6:9] = bytes([0, 2, 0])
    head[9] = 1
    head[10:14] = struct.pack("<I", 1)
    head[14] = 1
    pos = np.zeros(n_reads, dtype=np.uint64)
    ln = np.zeros(n_reads, dtype=np.uint32)
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack("<I", len(HDR_TEXT)))
        f.write(HDR_TEXT)
        at = f.tell()
        for lo in range(0, n_reads, chunk_reads):
            m = min(chunk_reads, n_reads - lo)
            b = press.DeviceBatch(np.full(m, n, dtype=np.uint64), device=dev)
            b.synth(seed=0x5105, first=lo)
            b.encode_stream()
            torch.cuda.synchronize()
            assert b.stream_ok()
            off = b.rec_off.cpu().numpy().astype(np.uint64)
            f.write(b.stream_out[: int(off[m])].cpu().numpy().tobytes())
            pos[lo:lo + m] = at + off[:m]
            ln[lo:lo + m] = np.diff(off).astype(np.uint32)
            at += int(off[m])
            del b
        f.write(b"5WOLB")
    torch.cuda.empty_cache()
    return pos, ln      # file extent of every record, size prefix included


def view_run(inp, out, rec, sig, workers, env, raw_gb):
    dt, r = _run([_exe("s5view"), inp, out, rec, sig, "4096", str(workers)], dict(env, S5VIEW_TIMING="1"))
    m = re.search(r"chunked pipeline(?: \(SLOW5 text (?:in|out)\))?: ([0-9.]+) s", r.stderr)
    inner = float(m.group(1)) if m else None
    res = {"whole_process_s": round(dt, 3), "GB_per_s_whole_process": round(raw_gb / dt, 3),
           "first_read_to_last_write_s": inner, "GB_per_s_first_read_to_last_write": round(raw_gb / inner, 3) if inner else None,
           "gpu_workers": workers, "pread_threads": int(env.get("S5VIEW_READERS", 0)) or None, "chunk_MB": int(env.get("S5VIEW_CHUNK_MB", 32)),
           "in_MB": round(os.path.getsize(inp) / 1e6, 1), "out_MB": round(os.path.getsize(out) / 1e6, 1), "timeline_s": _stamps(r.stderr)}
    return res


def cpu_view(ob, inp, out, cores, sample_reads, n, sweep):
    """the oracle's view loop on the first sample_reads records of the file: the reference's shape (-t all cores) and the best -t found"""
    raw_gb = sample_reads * n * 2 / 1e9
    pts = []
    for t in ([cores] + [t for t in sweep if t < cores]):
        t0 = time.perf_counter()
        got, ph = ob.view_file(inp, out, t, 4096, sample_reads)
        dt = time.perf_counter() - t0
        assert got == sample_reads, "CPU view twin failed (%d of %d records)" % (got, sample_reads)
        pts.append({"t": t, "K": 4096, "whole_call_s": round(dt, 3), "GB_per_s": round(raw_gb / dt, 3),
                    "phases_s": {k: round(v, 3) for k, v in ph.items()}})
    best = max(pts, key=lambda p: p["GB_per_s"])
    return {"kind": "port", "cores": cores, "sample": "first %d records of the same file (%.2f GB of raw signal)" % (sample_reads, raw_gb),
            "shape": "read K = 4096 records (one getline / fread + malloc each), work_db over -t pthreads created per batch, ordered fwrite + free per record; the three phases in turn",
            "t_all": pts[0], "best": best, "points": pts}


def e2e(args, L, _lib, press, torch, dev, ob, want_cpu=True):
    n, n_reads = 4000, args.e2e_reads
    raw_gb = n_reads * n * 2 / 1e9
    need = n_reads * n * (4.4 + 2 * 0.9 + 0.9)          # text + two BLOW5 files + slack
    d = pick_dir(need)
    while d is None and n_reads > 65536:                 # a small /dev/shm: a smaller file (said in the line)
        n_reads //= 2
        raw_gb = n_reads * n * 2 / 1e9
        d = pick_dir(n_reads * n * (4.4 + 2 * 0.9 + 0.9))
    if d is None:
        return {"error": "no room for the files in /dev/shm or /tmp"}
    work = os.path.join(d, "s5bench_e2e_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)
    res = {"dir": d, "reads": n_reads, "samples_per_read": n, "raw_signal_GB": round(raw_gb, 3)}
    cores = os.cpu_count() or 1
    try:
        blow5, slow5 = os.path.join(work, "in.blow5"), os.path.join(work, "in.slow5")
        out_b, out_c = os.path.join(work, "out.blow5"), os.path.join(work, "cpu.blow5")
        t0 = time.perf_counter()
        pos, ln = write_blow5(blow5, L, _lib, press, torch, dev, n_reads, n)
        res["setup_s"] = {"blow5_written": round(time.perf_counter() - t0, 2)}
        # a tiny warm-up run: the executable, the library and the runtime's files are in the page cache afterwards (every run below
        # still pays its own HIP start-up)
        tiny = os.path.join(work, "tiny.blow5")
        with open(blow5, "rb") as f:
            head = f.read(64 + 4 + len(HDR_TEXT) + 64 * 3600)
        cut = 64 + 4 + len(HDR_TEXT)
        p = cut
        while p + 8 <= len(head):
            (sz,) = struct.unpack_from("<Q", head, p)
            if p + 8 + sz > len(head):
                break
            p += 8 + sz
        with open(tiny, "wb") as f:
            f.write(head[:p] + b"5WOLB")
        _run([_exe("s5view"), tiny, os.path.join(work, "tiny_out.blow5"), "zlib", "svb-zd", "4096", "1"])
        # BLOW5 -> .slow5 (the text twin every later run reads)
        env_t = {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "32"}
        res["blow5_to_slow5"] = {"gpu": view_run(blow5, slow5, "none", "none", 3, env_t, raw_gb)}
        # .slow5 -> BLOW5 (zlib + svb-zd): the headline conversion
        best = None
        for workers, readers, chunk in ((3, 8, 64), (2, 8, 32)):
            r = view_run(slow5, out_b, "zlib", "svb-zd", workers, {"S5VIEW_READERS": str(readers), "S5VIEW_CHUNK_MB": str(chunk)}, raw_gb)
            if best is None or r["whole_process_s"] < best["whole_process_s"]:
                best = r
        same = os.path.getsize(out_b) == os.path.getsize(blow5)          # same encoder, same reads: the same bytes (checked below)
        res["slow5_to_blow5"] = {"gpu": best, "output_equals_the_device_encoders_file": bool(same and _same_file(out_b, blow5))}
        # BLOW5 -> BLOW5
        best = None
        for workers, readers, chunk in ((3, 4, 64), (2, 4, 32)):
            r = view_run(blow5, out_b, "zlib", "svb-zd", workers, {"S5VIEW_READERS": str(readers), "S5VIEW_CHUNK_MB": str(chunk)}, raw_gb)
            if best is None or r["whole_process_s"] < best["whole_process_s"]:
                best = r
        res["blow5_to_blow5"] = {"gpu": best, "output_equals_input": _same_file(out_b, blow5)}
        # get: index, 100 k random ids
        dt, _ = _run([_exe("s5view"), "--index", blow5])
        res["index"] = {"whole_process_s": round(dt, 3), "records_per_s": round(n_reads / dt, 1)}
        ids = os.path.join(work, "ids.txt")
        _run([_exe("s5get"), "--random", blow5, str(args.get_reads), "1", ids])
        g = {}
        for label, a in (("benchmark", ["--benchmark", blow5, ids, "4096", "8"]), ("to_file", [blow5, ids, out_b, "zlib", "svb-zd", "4096", "8"])):
            dt, r = _run([_exe("s5get")] + a)
            m = re.search(r"in ([0-9.]+) s = ([0-9.]+) reads/s", r.stderr)
            lat = re.search(r"p50 ([0-9.]+) ms, p99 ([0-9.]+) ms", r.stderr)
            il = re.search(r"index load ([0-9.]+) s", r.stderr)
            g[label] = {"whole_process_s": round(dt, 3), "reads_per_s_whole_process": round(args.get_reads / dt, 1),
                        "first_read_to_last_write_s": float(m.group(1)) if m else None, "reads_per_s_first_read_to_last_write": float(m.group(2)) if m else None,
                        "gpu_call_p50_ms": float(lat.group(1)) if lat else None, "gpu_call_p99_ms": float(lat.group(2)) if lat else None,
                        "index_load_s": float(il.group(1)) if il else None}
        res["get_100k"] = {"ids": args.get_reads, "K": 4096, "gpu": g}
        if want_cpu:
            sample = min(n_reads, args.e2e_cpu_reads)
            sweep = [t for t in (32, 64, 128) if t < cores]
            c = cpu_view(ob, slow5, out_c, cores, sample, n, sweep)
            res["slow5_to_blow5"]["cpu"] = c
            gw = res["slow5_to_blow5"]["gpu"]
            res["slow5_to_blow5"]["ratio"] = {"whole_process_vs_cpu_t_all": round(gw["GB_per_s_whole_process"] / c["t_all"]["GB_per_s"], 2),
                                              "whole_process_vs_cpu_best": round(gw["GB_per_s_whole_process"] / c["best"]["GB_per_s"], 2),
                                              "first_to_last_vs_cpu_t_all": round((gw["GB_per_s_first_read_to_last_write"] or 0) / c["t_all"]["GB_per_s"], 2),
                                              "first_to_last_vs_cpu_best": round((gw["GB_per_s_first_read_to_last_write"] or 0) / c["best"]["GB_per_s"], 2)}
            c = cpu_view(ob, blow5, out_c, cores, sample, n, [c["best"]["t"]] if c["best"]["t"] != cores else [])
            res["blow5_to_blow5"]["cpu"] = c
            gw = res["blow5_to_blow5"]["gpu"]
            res["blow5_to_blow5"]["ratio"] = {"whole_process_vs_cpu_t_all": round(gw["GB_per_s_whole_process"] / c["t_all"]["GB_per_s"], 2),
                                              "whole_process_vs_cpu_best": round(gw["GB_per_s_whole_process"] / c["best"]["GB_per_s"], 2)}
            # get twin: as many ids drawn the same way (uniform, with replacement), the records' extents as the writer noted them
            rng = np.random.default_rng(1)
            pick = rng.integers(0, n_reads, args.get_reads)
            pts = []
            for t in [cores] + [t for t in (32, 64) if t < cores]:
                t0 = time.perf_counter()
                samples, secs = ob.get_file(blow5, pos[pick], ln[pick], t, 4096)
                dt = time.perf_counter() - t0
                assert samples == args.get_reads * n
                pts.append({"t": t, "K": 4096, "whole_call_s": round(dt, 3), "reads_per_s": round(args.get_reads / dt, 1)})
            res["get_100k"]["cpu"] = {"kind": "port", "cores": cores, "shape": "get --benchmark: per id pread + inflate + parse + svb-zd decode in the worker threads (created per batch), nothing written",
                                      "t_all": pts[0], "best": max(pts, key=lambda p: p["reads_per_s"]), "points": pts}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return res


def _same_file(a, b, block=1 << 24):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(block), fb.read(block)
            if x != y:
                return False
            if not x:
                return True


def pcie_inclusive(L, _lib, press, n_reads, n, reps=2):
    """s5gpu_encode_batch on host buffers: what a patched view.c sees per batch (host int16 signals in, one malloc'd record per read out: the
    ownership contract of slow5_rec_to_mem, /root/reference/src/view.c:49,298) — at the reference's own batch sizes (K = 4096, /root/reference/src/cmd.h:8;
    K = 10 000, test/test_view_integrity.sh:62-66: one synchronous call per batch, as view.c:292 would issue it), at 65536 reads per call and at the
    full 1 M.  The first call of a size also allocates the library's pinned and device workspaces; best and first are both reported.  The 1 M call of
    the malloc form hands out 3.5 GB in a million buffers the process has never touched: page faults, not PCIe, bound it.  `arena` = the same call
    through s5gpu_encode_batch_arena (round 5): records are pointers into pooled pinned buffers, one release per batch."""
    out = {"call": "s5gpu_encode_batch (host int16 signals in, one malloc'd record per read out) and its arena form (pointers into pooled pinned buffers, one release)",
           "note": "PCIe-inclusive: never `value`"}
    for m in (4096, 10000, 65536, n_reads):
        if m > n_reads:
            continue
        r = 8 if m <= 10000 else 6 if m <= 65536 else reps      # (the allocator needs a few calls to settle: 1.4 / 5.4 / 21.5 GB/s on calls 1 / 2 / 3 of 65536)
        out["batch_%d" % m] = _pcie_one(L, _lib, press, m, n, r, arena=False)
        out["batch_%d" % m]["arena"] = _pcie_one(L, _lib, press, m, n, r + 1, arena=True)
        if m <= 65536:   # the reference's loop with batch k + 1 submitted while batch k is on the device (round 6: s5gpu_encode_batch_submit / s5gpu_batch_wait)
            try:
                out["batch_%d" % m]["two_in_flight"] = _pcie_two_in_flight(L, _lib, press, m, n, 48 if m <= 10000 else 12)
            except Exception as e:      # (never fatal for the line)
                out["batch_%d" % m]["two_in_flight"] = {"error": repr(e)}
    big = out["batch_%d" % n_reads]
    out.update({"reads": n_reads, "samples_per_read": n, "GB_per_s": big["GB_per_s"], "reads_per_s": big["reads_per_s"], "arena_GB_per_s": big["arena"]["GB_per_s"]})
    return out


def _pcie_one(L, _lib, press, n_reads, n, reps, arena=False):
    rng = np.random.default_rng(0)
    base = (500 + 30 * rng.standard_normal((1024, n))).astype(np.int16)
    sig = np.ascontiguousarray(np.tile(base, (n_reads // 1024 + 1, 1))[:n_reads])
    hdr = np.frombuffer(press.pack_hdr("0" * 36, 0, 8192.0, 23.0, 1467.61, 4000.0), dtype=np.uint8)
    vp = C.c_void_p
    addr = sig.ctypes.data + 2 * n * np.arange(n_reads, dtype=np.uint64)
    sig_p = (vp * n_reads).from_buffer_copy(addr.tobytes())
    ns = (C.c_uint64 * n_reads).from_buffer_copy(np.full(n_reads, n, dtype=np.uint64).tobytes())
    hdr_p = (vp * n_reads).from_buffer_copy(np.full(n_reads, hdr.ctypes.data, dtype=np.uint64).tobytes())
    hl = (C.c_uint32 * n_reads).from_buffer_copy(np.full(n_reads, len(hdr), dtype=np.uint32).tobytes())
    out = (vp * n_reads)()
    ol = (C.c_size_t * n_reads)()
    libc = C.CDLL(None)
    libc.free.argtypes = [vp]
    L.s5gpu_encode_batch_arena.argtypes = [C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
    L.s5gpu_arena_release.argtypes = [vp]
    times = []
    tot = 0
    for it in range(reps):
        if arena:
            h = vp()
            t0 = time.perf_counter()
            _lib.check(L.s5gpu_encode_batch_arena(n_reads, sig_p, ns, hdr_p, hl, None, None, 1, 1, out, ol, C.byref(h)))
            times.append(time.perf_counter() - t0)
        else:
            t0 = time.perf_counter()
            _lib.check(L.s5gpu_encode_batch(n_reads, sig_p, ns, hdr_p, hl, None, None, 1, 1, out, ol))
            times.append(time.perf_counter() - t0)
        lens = np.frombuffer(ol, dtype=np.uint64 if C.sizeof(C.c_size_t) == 8 else np.uint32)
        tot = int(lens.sum())
        if arena:
            t1 = time.perf_counter()
            L.s5gpu_arena_release(h)
            times[-1] += time.perf_counter() - t1          # the release belongs to the call's cost, as the free loop does to the malloc form's user
        else:
            for p in np.frombuffer(out, dtype=np.uint64).tolist():
                libc.free(p)
    best = min(times)
    return {"reads": n_reads, "seconds": [round(t, 4) for t in times], "GB_per_s": round(n_reads * n * 2 / best / 1e9, 3), "reads_per_s": round(n_reads / best, 1),
            "first_call_GB_per_s": round(n_reads * n * 2 / times[0] / 1e9, 3), "bytes_per_sample": round(tot / (n_reads * n), 4)}


def _pcie_two_in_flight(L, _lib, press, m, n, batches):
    """`batches` host batches of m reads through s5gpu_encode_batch_submit (arena form), two tickets in flight at any time, two alternating sets
    of output arrays: submit k, wait for k - 1, release it — what the loop of /root/reference/src/view.c:254-300 does once it reads batch k + 1
    while batch k is on the device.  Rate = all reads / (first submit .. last release); the first two batches (pool
    and workspace warm-up) are run once before the clock starts."""
    rng = np.random.default_rng(0)
    base = (500 + 30 * rng.standard_normal((1024, n))).astype(np.int16)
    sig = np.ascontiguousarray(np.tile(base, (m // 1024 + 1, 1))[:m])
    hdr = np.frombuffer(press.pack_hdr("0" * 36, 0, 8192.0, 23.0, 1467.61, 4000.0), dtype=np.uint8)
    vp = C.c_void_p
    addr = sig.ctypes.data + 2 * n * np.arange(m, dtype=np.uint64)
    sig_p = (vp * m).from_buffer_copy(addr.tobytes())
    ns = (C.c_uint64 * m).from_buffer_copy(np.full(m, n, dtype=np.uint64).tobytes())
    hdr_p = (vp * m).from_buffer_copy(np.full(m, hdr.ctypes.data, dtype=np.uint64).tobytes())
    hl = (C.c_uint32 * m).from_buffer_copy(np.full(m, len(hdr), dtype=np.uint32).tobytes())
    sets = [((vp * m)(), (C.c_size_t * m)()) for _ in range(2)]
    L.s5gpu_encode_batch_submit.argtypes = [C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.s5gpu_encode_batch_submit.restype = vp
    L.s5gpu_batch_wait.argtypes = [vp, C.POINTER(vp)]
    L.s5gpu_arena_release.argtypes = [vp]

    def run(nb):
        tickets = [None, None]
        tot = 0
        t0 = time.perf_counter()
        for k in range(nb + 1):
            cur = k & 1
            if k < nb:
                t = L.s5gpu_encode_batch_submit(m, sig_p, ns, hdr_p, hl, None, None, 1, 1, sets[cur][0], sets[cur][1], 1)
                if not t:
                    raise RuntimeError("s5gpu_encode_batch_submit failed")
                tickets[cur] = t
            prev = cur ^ 1
            if tickets[prev]:
                h = vp()
                _lib.check(L.s5gpu_batch_wait(tickets[prev], C.byref(h)), "s5gpu_batch_wait")
                tickets[prev] = None
                tot += int(np.frombuffer(sets[prev][1], dtype=np.uint64).sum())
                L.s5gpu_arena_release(h)
        return time.perf_counter() - t0, tot

    run(2)
    dt, tot = run(batches)
    return {"batches": batches, "reads_per_batch": m, "seconds": round(dt, 4), "GB_per_s": round(batches * m * n * 2 / dt / 1e9, 3),
            "reads_per_s": round(batches * m / dt, 1), "ms_per_batch": round(dt / batches * 1e3, 3), "bytes_per_sample": round(tot / (batches * m * n), 4)}


HOST_FED_CODE = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import bench_e2e
from slow5tools_amd import _lib, press
L = _lib.lib()
mask = %(mask)d
_lib.check(L.s5gpu_init_mask(mask), "s5gpu_init_mask")
out = {"devices": L.s5gpu_devices_in_use(), "dev_mask": mask, "call": "s5gpu_encode_batch_arena / s5gpu_encode_batch: ONE host batch call cut into one contiguous index range per device (SURVEY 8e), host int16 signals in, records out"}
for m in %(sizes)r:
    out["batch_%%d" %% m] = {"arena": bench_e2e._pcie_one(L, _lib, press, m, %(n)d, 4, arena=True), "malloc": bench_e2e._pcie_one(L, _lib, press, m, %(n)d, 2, arena=False) if m <= 65536 else None}
print("HOSTFED " + json.dumps(out))
"""


def host_fed_multi(n_gpus, n_reads, n, alias):
    """The host-buffer path over ALL the node's GPUs from one process (what a patched slow5tools does after slow5_gpu_hook_init(mask)): a fresh
    process — the bench's own ranks each hold one device — initialises the library on devices 0 .. n_gpus - 1 and times the arena batch call at
    65536 reads and at n_reads per call.  With S5BENCH_ALIAS_DEVICES (one-GPU rehearsal) the devices are aliases of device 0."""
    import subprocess
    import sys
    env = dict(os.environ)
    if alias:
        env["S5GPU_ALIAS_DEVICES"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    sizes = [m for m in (65536, n_reads) if m <= n_reads] or [n_reads]
    code = HOST_FED_CODE % dict(root=ROOT, mask=(1 << n_gpus) - 1, sizes=sorted(set(sizes)), n=n)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("HOSTFED "):
            import json
            return json.loads(line[8:])
    return {"error": (r.stdout + r.stderr)[-600:]}
