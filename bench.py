#!/usr/bin/env python3
"""bench.py — BLOW5 record press throughput on N MI355X (BASELINE.json's metric: raw-signal GB/s and reads/s).

Default run = BASELINE configs[2] as the headline value (1 M reads x 4000 samples per GPU, full encode svb-zd + DEFLATE,
weak scaling) PLUS, in the same JSON line, one object per other GPU config of BASELINE.json:
    "configs1"  svb-zd encode alone on the same reads                                     (k_svbzd_encode, HBM-bound)
    "configs4"  decode for random `get` over the index of the records just written: K = 4096 batches (p50 / p99 latency) and
                the whole index in one call, on our own records and on records written by stock zlib, every signal compared
                with the generator; fields + signals only (S5GPU_DEC_NO_PAYLOAD), the full-record form beside it
    "configs3"  100 k-sample reads, a fixed read-index space sharded over the ranks with shard.shard_range (strong scaling)
Every GPU leg runs for about a second of device time or more (the headline's K steps are what `value` is computed on; a
"sustained" repeat of the same step follows it), so that a coarse utilisation sampler sees the legs.

A step = one pass of the hot path over one resident batch of synthetic reads, ending in the contiguous BLOW5 record
stream the ordered fwrite loop emits:
    k_encode_stream (svb-zd -> pack -> DEFLATE -> zlib frame, one read per workgroup, records placed by a
                     decoupled look-back: one launch)                                   [default]
    or k_encode_fused / k_pack + k_deflate_staged into worst-case slots + s5gpu_compact (--two-pass, long reads, --svb-only)
Inputs (int16 signals, 74-byte record heads) are already in HBM when the timed region starts.
Reads shard across ranks with no collective (RCCL carries the timing barrier and the MAX of the elapsed time only).
Prints ONE JSON line on rank 0.

Other modes (one line each, for profiles/):  --svb-only (configs[1]), --long (configs[3] alone), --mixed (read lengths of a
real run), --decode (configs[4] alone), --samples / --reads (any uniform shape).
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_HBM_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md chip table (6290 measured copy ceiling quoted beside it)
LONG_TOTAL_READS_FULL = 10_000_000   # BASELINE configs[3]: 10 M reads x 100 k samples = 2 TB raw: run on a stated fraction


def csrc_sha256():
    """content hash of the kernel sources: PMC traffic figures are only valid for the build they were collected on
    (the GPU box has no .git, so a content hash stands in for the commit id)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slow5tools_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, samples, n_reads):
    """HBM bytes per launch from the committed PMC pass (2 x FETCH_SIZE + WRITE_SIZE, tools/pmc.sh), or None when the
    kernel sources changed since it was collected (profiles/pmc_traffic.json carries the hash)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        entries = t if isinstance(t, list) else [t]
        now = csrc_sha256()
        for e in entries:
            if e.get("kernel") == kernel and e.get("samples_per_read") == samples and e.get("csrc_sha256") == now:
                return int(e["hbm_bytes_per_read"] * n_reads), {"file": e.get("source"), "csrc_sha256": now}
    except Exception:
        pass
    return None, None


def pmc_issue(kernel, samples, reads_per_s):
    """The ceiling that binds the DEFLATE / inflate kernels is instruction issue, not HBM: vector instructions per read from the committed
    PMC pass (SQ_INSTS_VALU, profiles/pmc_traffic.json, same source hash rule as the traffic) x the reads/s of THIS run x 4 cycles per
    wave64 instruction, over the chip's 1024 SIMDs at 2.4 GHz.  None when the kernel sources changed since the counters were collected."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        now = csrc_sha256()
        for e in (t if isinstance(t, list) else [t]):
            if e.get("kernel") == kernel and e.get("samples_per_read") == samples and e.get("csrc_sha256") == now and e.get("valu_insts_per_read"):
                v = float(e["valu_insts_per_read"])
                peak = 1024 * 2.4e9 / 4.0                      # wave-level vector instructions per second, whole chip
                return {"bound": "valu_issue", "valu_insts_per_read": v, "cycles_per_read": 4.0 * v, "achieved": round(v * reads_per_s / 1e9, 2),
                        "peak": round(peak / 1e9, 1), "unit": "G wave-instructions/s", "frac": round(v * reads_per_s / peak, 4),
                        "source": {"file": e.get("source"), "csrc_sha256": now}}
    except Exception:
        pass
    return None


def make_events(L, _lib, count):
    evs = []
    for _ in range(count):
        e = C.c_void_p()
        _lib.check(L.s5gpu_event_create(C.byref(e)))
        evs.append(e)
    return evs


def elapsed_ms(L, _lib, a, b):
    ms = C.c_float()
    _lib.check(L.s5gpu_event_elapsed_ms(a, b, C.byref(ms)))
    return ms.value


TIMING_DEV = None   # device of the one-element tensors of the timing all-reduces (the rank's GPU; "cpu" under the gloo pre-flight)


def timed(shard, torch, dev, body):
    """the contract's timed region: barrier + synchronize on both sides, MAX over ranks"""
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    body()
    torch.cuda.synchronize()
    shard.barrier()
    return shard.max_over_ranks(time.perf_counter() - t0, device=TIMING_DEV or dev)


def cpu_encode_baseline(ob, sig2d, first, n, ref_seconds, sweep_seconds):
    """The oracle's reference-shaped pthread batch encode on this box's host cores.  `value` is the reference's own shape
    (view -t <all cores> -K 4096: threads created and joined per batch, 16 records per thread on a 256-core box, so it is
    thread-spawn bound); `best_of` is a sweep over -t and -K (still per-record deflateInit, src/view.c:43-54) — the fair CPU
    configuration.  Each point runs until >= its share of seconds."""
    cores = os.cpu_count() or 1
    m = sig2d.shape[0]

    def run(t, K, secs_target):
        reads = 0
        secs = 0.0
        out_bytes = 0
        while secs < secs_target:
            tot, s, _ = ob.encode_batch_mt(sig2d, first, t, K)
            reads += m
            secs += s
            out_bytes += tot
        return reads, secs, out_bytes

    reads, secs, out_bytes = run(cores, 4096, ref_seconds)
    cpu = {"value": round(reads * 2 * n / secs / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
           "reads_per_s": round(reads / secs, 1), "bytes_per_sample": round(out_bytes / (reads * n), 4),
           "shape": "view -t %d -K 4096 (the reference's defaults on this box; thread create/join per batch, src/thread.c:100-110)" % cores,
           "sample": "first %d reads of the same batch (%d samples each), compute phase only (svb-zd + zlib-1.2.11 level 6, per-record "
                     "deflateInit), repeated for %.1f s" % (m, n, secs),
           "per_core_note": "the reference allocates and clears a 256 KiB deflate state per record (slow5_press_init, src/view.c:43-54) and "
                            "creates / joins its threads per batch (src/thread.c:100-110): per-core throughput at -t all is a fraction of the "
                            "single-thread figure; best_of is the -t / -K the same code runs fastest at"}
    if sweep_seconds > 0:
        sweep = []
        for t in sorted({min(32, cores), min(64, cores), min(128, cores), cores}):
            for K in (4096, 65536):
                if t == cores and K == 4096:
                    r, s = reads, secs
                else:
                    r, s, _ = run(t, K, sweep_seconds)
                sweep.append({"t": t, "K": K, "GB_per_s": round(r * 2 * n / s / 1e9, 3), "seconds": round(s, 1)})
        best = max(sweep, key=lambda x: x["GB_per_s"])
        cpu["best_of"] = {"value": best["GB_per_s"], "unit": "GB/s", "t": best["t"], "K": best["K"]}
        cpu["sweep"] = sweep
        # NOT the reference's shape, for scale only: the same sweep point with one deflate state per worker thread, reset per record
        # (deflateReset) instead of the reference's slow5_press_init per record — how much of the CPU figure is that allocation + memset
        try:
            K_p = 65536                                    # (threads live per batch: a large batch lets a thread's state serve many records)
            pts = []
            for t in sorted({best["t"], cores}):
                r, s_ = 0, 0.0
                while s_ < max(2.0, sweep_seconds / 2):
                    tot, sec, _ = ob.encode_batch_mt(sig2d, first, t, K_p, pooled_zstream=True)
                    r += m; s_ += sec
                pts.append({"t": t, "K": K_p, "GB_per_s": round(r * 2 * n / s_ / 1e9, 3), "seconds": round(s_, 1)})
            cpu["pooled_zstream"] = {"what": "one deflate state per worker thread, deflateReset per record: NOT the reference's shape (src/view.c:43-54 allocates per record)",
                                     "points": pts, "value": max(p["GB_per_s"] for p in pts), "unit": "GB/s"}
        except Exception as e:
            cpu["pooled_zstream"] = {"error": repr(e)}
        # the same worker fed with SLOW5 TEXT (BASELINE configs[0]: view in.slow5 -o out.blow5 — the ASCII parse of
        # slow5_rec_depress_parse is part of the reference's compute phase there): a smaller sample, printed by the oracle
        try:
            import numpy as np
            ma = min(m, 32768)
            lines = []
            for i in range(ma):
                r, keep = ob.make_rec(ob.synth_read_id(first + i), 0, 8192.0, 23.0, 1467.61, 4000.0, np.ascontiguousarray(sig2d[i]))
                lines.append(ob.payload_to_line(ob.rec_pack(r, 0)))
            text = np.frombuffer(b"".join(lines), dtype=np.uint8)
            off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.uint64)
            pts = []
            for t in sorted({best["t"], cores}):
                got, secs_ = 0, 0.0
                while secs_ < max(2.0, sweep_seconds / 2):
                    tot, s_, _ = ob.convert_ascii_batch_mt(text, off, t, 4096)
                    assert tot > 0, "CPU ASCII conversion failed"
                    got += ma; secs_ += s_
                pts.append({"t": t, "K": 4096, "GB_per_s": round(got * 2 * n / secs_ / 1e9, 3), "text_GB_per_s": round(got / ma * text.size / secs_ / 1e9, 3), "seconds": round(secs_, 1)})
            cpu["slow5_text_input"] = {"what": "the same worker on SLOW5 text: ASCII line parse (oracle/ascii.c) + svb-zd + zlib per record, %d reads (%.2f GB of text)" % (ma, text.size / 1e9),
                                       "points": pts, "value": max(p["GB_per_s"] for p in pts), "unit": "GB/s of raw signal"}
        except Exception as e:      # (never fatal for the line)
            cpu["slow5_text_input"] = {"error": repr(e)}
    return cpu


def long_leg(args, L, _lib, press, shard, ob, rank, world, dev, K, W):
    """BASELINE configs[3]: record-sharded encode of 100 k-sample reads.  The job is a FIXED read-index space
    [0, --long-reads) (a stated fraction of the 10 M reads: 2 TB of raw signal does not fit anywhere at once), rank g of G
    takes shard_range(total, g, G) — strong scaling — and works through its shard in chunks of <= --long-chunk reads
    (16384 reads = 3.3 GB of signal per launch), signals generated on the device and resident before the timed region.
    Per chunk: k_pack (svb-zd tiles -> parked payload) + k_deflate_staged (16 KiB blocks through LDS, in place) + the
    compaction into the contiguous record stream."""
    import numpy as np
    import torch

    n = args.long_samples
    total = args.long_reads
    lo, hi = shard.shard_range(total, rank, world)
    mine = hi - lo
    # at least four chunks per rank, alternating between two sets of output buffers: the streaming steps of one chunk (park the payloads;
    # compact the records) run under the arithmetic of another (below).  This is what several GPU workers of the file pipeline do
    # (examples/s5view.c); 288 GB of HBM hold both sets many times over.
    chunk = max(1, min(args.long_chunk, -(-mine // 4)))
    n_sets = 1 if args.long_streams < 2 or mine <= chunk else 2
    chunks = []
    sets = []
    for ci, c0 in enumerate(range(lo, hi, chunk)):
        cn = min(chunk, hi - c0)
        b = press.DeviceBatch(np.full(cn, n, dtype=np.uint64), device=dev, share=sets[ci % n_sets] if ci >= n_sets else None)
        if ci < n_sets:
            sets.append(b)
        b.synth(seed=0x5105, first=c0)
        chunks.append((c0, b))
    torch.cuda.synchronize()
    # Three streams, two sets of output buffers: the staged encode of a chunk is two calls (s5gpu_pack_parked_dev: streaming, HBM-bound;
    # s5gpu_deflate_parked_dev: arithmetic, VALU-bound) and the compaction a third (a copy).  Stream D deflates chunk after chunk; stream P
    # parks chunk c + 1 and stream C compacts chunk c - 1 meanwhile; a buffer set is taken again once its compaction is done.
    # Small chunks (a rank's shard of the strong-scaling job at N = 8: 2048 reads = two rounds of workgroups) take two deflate streams in
    # turn, so that a chunk's last workgroups drain beside the next chunk's first (8192 reads on one GPU: 548 against 518 GB/s); large
    # chunks do better with one (65 536 reads: 582 against 567).
    alt_deflate = chunk < 8192
    streams = [torch.cuda.Stream(device=dev) for _ in range((4 if alt_deflate else 3) if n_sets == 2 else 1)]
    set_free = [None] * n_sets

    def step(evs=None, k=0, serial=False):
        for ci, (_, b) in enumerate(chunks):
            if n_sets == 2 and not serial:
                sp, sc, sd = streams[0], streams[1], streams[2 + (ci % 2 if alt_deflate else 0)]
                with torch.cuda.stream(sp):
                    if set_free[ci % 2] is not None:
                        sp.wait_event(set_free[ci % 2])
                    b.pack_parked()
                    parked = torch.cuda.Event()
                    parked.record(sp)
                with torch.cuda.stream(sd):
                    sd.wait_event(parked)
                    b.deflate_parked()
                    done = torch.cuda.Event()
                    done.record(sd)
                with torch.cuda.stream(sc):
                    sc.wait_event(done)
                    b.compact()
                    set_free[ci % 2] = torch.cuda.Event()
                    set_free[ci % 2].record(sc)
                continue
            with torch.cuda.stream(streams[0]):
                st = b._stream()
                if evs is not None:
                    L.s5gpu_event_record(evs[(k * len(chunks) + ci) * 3], st)
                b.encode()
                if evs is not None:
                    L.s5gpu_event_record(evs[(k * len(chunks) + ci) * 3 + 1], st)
                b.compact()
                if evs is not None:
                    L.s5gpu_event_record(evs[(k * len(chunks) + ci) * 3 + 2], st)

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    # at least ~1 s of device time: the leg's own step count (the headline's K is kept when it is larger)
    K = max(K, args.min_leg_steps_long)
    dt = timed(shard, torch, dev, lambda: [step() for k in range(K)])
    # the kernels' own durations: the same chunk loop on ONE stream (launches that overlap share the device, so events around them
    # would not time a kernel), right behind the timed region
    Ks = max(2, K // 4)
    evs = make_events(L, _lib, 3 * Ks * len(chunks))
    dt_serial = timed(shard, torch, dev, lambda: [step(evs, k, True) for k in range(Ks)])
    enc_ms = cmp_ms = 0.0
    for i in range(Ks * len(chunks)):
        enc_ms += elapsed_ms(L, _lib, evs[3 * i], evs[3 * i + 1])
        cmp_ms += elapsed_ms(L, _lib, evs[3 * i + 1], evs[3 * i + 2])
    for e in evs:
        L.s5gpu_event_destroy(e)
    enc_ms /= Ks
    cmp_ms /= Ks
    if rank != 0:
        return None
    # the last chunk's output is still in the shared buffers; sizes of every chunk are in its own out_len
    z_bytes = sum(int(b.out_len[: b.n].sum().item()) for _, b in chunks)
    import zlib

    c0, b = chunks[-1]
    idx = sorted({0, b.n // 2, b.n - 1})
    parity = True
    for i, rec in zip(idx, b.stream_records(idx)):
        sig = ob.synth_read(0x5105, c0 + i, n)
        r, keep = ob.make_rec(ob.synth_read_id(c0 + i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
        parity &= zlib.decompress(rec[8:]) == ob.rec_pack(r, ob.SIG_SVB_ZD)
    alg = 2 * n * mine + 74 * mine + z_bytes          # 2N + H + Z per read, this rank's shard per step
    achieved = alg / (enc_ms / 1e3) / 1e9
    reads_per_s = total * K / dt
    traffic, traffic_src = pmc_traffic("k_pack+k_deflate_staged", n, mine)
    return {
        "workload": "BASELINE configs[3] shape: record-sharded full BLOW5 encode, %d reads x %d int16 samples = %.1f GB raw signal per step "
                    "(a 1/%.1f fraction of the 10 M-read job), read-index space split over %d rank(s) with shard_range, chunks of <= %d reads "
                    "(%.2f GB of signal per launch), generated on device" % (total, n, total * 2 * n / 1e9, LONG_TOTAL_READS_FULL / total, world, chunk, chunk * 2 * n / 1e9),
        "value": round(reads_per_s * 2 * n / 1e9, 3), "unit": "GB/s", "reads_per_s": round(reads_per_s, 1), "scaling": "strong",
        "n_gpus": world, "steps": K, "ms_per_step": round(dt / K * 1e3, 3), "streams": len(streams), "buffer_sets": n_sets,
        "ms_per_step_one_stream": round(dt_serial / Ks * 1e3, 3), "reads_total": total, "reads_rank0": mine, "chunks_rank0": len(chunks),
        "scale_factor_vs_10M_reads": round(LONG_TOTAL_READS_FULL / total, 3),
        "bytes_per_sample": round(z_bytes / (mine * n), 4), "parity_spot_check": bool(parity),
        "kernel_ms": {"pack+deflate_staged": round(enc_ms, 3), "compact": round(cmp_ms, 3)},
        "roofline": {"bound": "hbm", "kernel": "k_pack+k_deflate_staged", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg,
                     "note": "rank 0's shard per step; launch = the chunk loop of one step (k_pack + k_deflate_staged per chunk), durations from %d steps of the "
                             "same loop on one stream right behind the timed region" % Ks},
    }


def mixed_leg(args, L, _lib, press, shard, rank, world, dev):
    """The read lengths of a real run (what a PromethION DNA flow cell writes: log-normal, median ~6000 samples, a tail past 300 k) as a leg of the
    default line: --mixed-reads reads per GPU (weak scaling), full BLOW5 encode through s5gpu_encode_dev with an 8 KiB fused budget — short reads
    in the fused kernel, the rest through the overflow list (launched longest first) and the staged kernels — then the compaction into the record
    stream.  /root/reference/src/thread.c:19-37 (work stealing) exists for exactly this imbalance."""
    import zlib

    import numpy as np
    import torch

    n_reads = args.mixed_reads
    rng = np.random.default_rng(5 + rank)
    ns = np.clip(np.exp(rng.normal(np.log(6000), 0.9, n_reads)), 200, 400000).astype(np.uint64)
    b = press.DeviceBatch(ns, device=dev, lds_payload_cap=args.fused_cap)
    tot = b.sig.numel()
    _lib.check(L.s5gpu_synth_dev(b.sig.data_ptr(), 1, tot - 64, tot, 0x5105 + rank, 0, b._stream()), "synth")     # one long trace cut into the reads
    _lib.check(L.s5gpu_synth_hdr_dev(b.hdr.data_ptr(), n_reads, rank * n_reads, b._stream()), "hdr")
    raw_bytes = int(2 * ns.sum())
    torch.cuda.synchronize()
    st = b._stream()
    for _ in range(2):
        b.encode()
        b.compact()
    torch.cuda.synchronize()
    K = max(4, int(args.min_leg_seconds / 0.012))
    evs = make_events(L, _lib, 3 * K)

    def steps():
        for k in range(K):
            L.s5gpu_event_record(evs[3 * k], st)
            b.encode()
            L.s5gpu_event_record(evs[3 * k + 1], st)
            b.compact()
            L.s5gpu_event_record(evs[3 * k + 2], st)

    dt = timed(shard, torch, dev, steps)
    enc_ms = float(np.mean([elapsed_ms(L, _lib, evs[3 * k], evs[3 * k + 1]) for k in range(K)]))
    cmp_ms = float(np.mean([elapsed_ms(L, _lib, evs[3 * k + 1], evs[3 * k + 2]) for k in range(K)]))
    for e in evs:
        L.s5gpu_event_destroy(e)
    out_len = b.out_len[:n_reads].cpu().numpy().astype(np.int64)
    z_bytes = int(out_len.sum())
    parity = True
    if rank == 0:   # stock zlib must inflate sampled records of every length class to a payload of the right shape
        order = np.argsort(ns)
        idx = [int(order[0]), int(order[n_reads // 2]), int(order[-1]), 0, n_reads - 1]
        for i, rec in zip(idx, b.stream_records(idx)):
            pay = zlib.decompress(rec[8:])
            parity &= len(pay) > 86 and int.from_bytes(pay[82:86], "little") == int(ns[i])
    del b
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    alg = raw_bytes + 74 * n_reads + z_bytes
    kern_s = (enc_ms + cmp_ms) / 1e3
    kernel = "k_encode_fused+k_pack+k_deflate_staged+k_compact"
    traffic, traffic_src = pmc_traffic(kernel, "mixed", n_reads)
    return {"workload": "read lengths of a real run: %d reads per GPU, log-normal lengths (median %d, max %d samples, %.2f G samples), full BLOW5 encode, %d-byte fused budget, overflow list launched longest first"
                        % (n_reads, int(np.median(ns)), int(ns.max()), ns.sum() / 1e9, args.fused_cap),
            "value": round(raw_bytes * world * K / dt / 1e9, 3), "unit": "GB/s", "steps": K, "ms_per_step": round(dt / K * 1e3, 3), "scaling": "weak",
            "reads_per_s": round(n_reads * world * K / dt, 1), "samples_per_s": round(float(ns.sum()) * world * K / dt, 1),
            "bytes_per_sample": round(z_bytes / (raw_bytes / 2), 4), "parity_spot_check": bool(parity),
            "kernel_ms": {"encode": round(enc_ms, 3), "compact": round(cmp_ms, 3)},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(alg / kern_s / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(alg / kern_s / 1e9 / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg}}


def svb_leg(args, L, _lib, shard, ob, b, n_reads, n, rank, world, dev, K, W):
    """BASELINE configs[1]: the svb-zd stage alone on the same resident reads, blobs into one contiguous stream (k_svbzd_stream;
    --two-pass: k_svbzd_encode into slots + the compaction), bit-exact against the oracle on a spot sample (the full comparison is tests/test_full_size.py)."""
    import numpy as np
    import torch

    st = b._stream()
    one_pass = not args.two_pass

    def step(evs=None, k=0):
        if evs is not None:
            L.s5gpu_event_record(evs[3 * k], st)
        if one_pass:
            b.svbzd_encode_stream()          # blobs straight into the contiguous stream (k_svbzd_stream)
        else:
            b.svbzd_encode()
        if evs is not None:
            L.s5gpu_event_record(evs[3 * k + 1], st)
        if not one_pass:
            b.compact()
        if evs is not None:
            L.s5gpu_event_record(evs[3 * k + 2], st)

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    K = max(K, args.min_leg_steps_svb)
    evs = make_events(L, _lib, 3 * K)
    dt = timed(shard, torch, dev, lambda: [step(evs, k) for k in range(K)])
    enc = [elapsed_ms(L, _lib, evs[3 * k], evs[3 * k + 1]) for k in range(K)]
    cmp_ = [elapsed_ms(L, _lib, evs[3 * k + 1], evs[3 * k + 2]) for k in range(K)]
    for e in evs:
        L.s5gpu_event_destroy(e)
    two = None
    if one_pass:   # the two launches the one-pass kernel replaces, on the same reads (a quarter of the steps): k_svbzd_encode alone is the HBM-bound figure
        K2 = max(2, K // 4)
        ev2 = make_events(L, _lib, 3 * K2)

        def step2(k):
            L.s5gpu_event_record(ev2[3 * k], st)
            b.svbzd_encode()
            L.s5gpu_event_record(ev2[3 * k + 1], st)
            b.compact()
            L.s5gpu_event_record(ev2[3 * k + 2], st)

        dt2 = timed(shard, torch, dev, lambda: [step2(k) for k in range(K2)])
        two = {"steps": K2, "ms_per_step": round(dt2 / K2 * 1e3, 3),
               "kernel_ms": {"svbzd_encode": round(float(np.mean([elapsed_ms(L, _lib, ev2[3 * k], ev2[3 * k + 1]) for k in range(K2)])), 3),
                             "compact": round(float(np.mean([elapsed_ms(L, _lib, ev2[3 * k + 1], ev2[3 * k + 2]) for k in range(K2)])), 3)}}
        for e in ev2:
            L.s5gpu_event_destroy(e)
        b.svbzd_encode_stream()      # leave the one-pass stream in place for the spot check
        torch.cuda.synchronize()
    if rank != 0:
        return None
    s_bytes = int(b.out_len[:n_reads].sum().item())
    idx = [0, 1, n_reads // 2, n_reads - 1] if n_reads >= 4 else list(range(n_reads))
    parity = b.stream_ok() if one_pass else True
    for i, blob in zip(idx, b.stream_records(idx)):
        parity &= blob == ob.svbzd_encode(ob.synth_read(0x5105, rank * n_reads + i, n))
    alg = 2 * n * n_reads + s_bytes                   # 2N + S per read
    kern_s = float(np.mean(enc)) / 1e3
    kname = "k_svbzd_stream" if one_pass else "k_svbzd_encode"
    traffic, traffic_src = pmc_traffic(kname, n, n_reads)
    # which form is the faster one changed in round 4 (k_compact with 16-byte copies: the two launches 4.7 ms, the one-launch kernel 4.8): the leg's value
    # is the form the timed region ran (one launch unless --two-pass); the other form's figure stands beside it (`two_pass`)
    return {
        "workload": "BASELINE configs[1]: svb-zd zig-zag-delta encode only, %d reads x %d int16 samples per GPU, bit-exact vs the CPU svb" % (n_reads, n),
        "value": round(2 * n * n_reads * world * K / dt / 1e9, 3), "unit": "GB/s", "reads_per_s": round(n_reads * world * K / dt, 1),
        "scaling": "weak", "n_gpus": world, "steps": K, "ms_per_step": round(dt / K * 1e3, 3),
        "svb_bytes_per_sample": round(s_bytes / (n_reads * n), 4), "parity_spot_check": bool(parity),
        "output": "ordered single-pass blob stream (k_svbzd_stream)" if one_pass else "slots + compaction (k_svbzd_encode + k_compact)",
        "kernel_ms": {kname[2:]: round(float(np.mean(enc)), 3), "compact": round(float(np.mean(cmp_)), 3)},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(alg / kern_s / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(alg / kern_s / 1e9 / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg},
        **({"two_pass": dict(two, value=round(2 * n * n_reads * world / (two["ms_per_step"] / 1e3) / 1e9, 3), unit="GB/s",
                             k_svbzd_encode_roofline_frac=round(alg / (two["kernel_ms"]["svbzd_encode"] / 1e3) / 1e9 / PEAK_HBM_GBS, 5),
                             what="k_svbzd_encode into slots + k_compact: the launches k_svbzd_stream replaces")} if two else {}),
    }


def decode_leg(args, L, _lib, shard, ob, b, n_reads, n, rank, world, dev, want_cpu):
    """BASELINE configs[4]: decode for random `get` over the index of the --reads records in b.stream_out / b.rec_off, batches of
    --get-batch ids.  Per batch (what src/get.c:321-386 does per -K batch, minus the preads: the file-backed harness is
    examples/s5get.c, tools/get_bench.py): build the batch's record descriptors from the index, upload them, inflate + unpack on the
    GPU, synchronise.  Then the whole index in one call.  Every decoded signal is compared with the generator's.
    Primary form: fields + signals only (S5GPU_DEC_NO_PAYLOAD: the caller of a get holds the read ids, the synthetic records have no
    aux fields); the form that also writes the uncompressed record out is timed beside it."""
    import numpy as np
    import torch

    torch.cuda.synchronize()
    rec_off = b.rec_off.cpu().numpy().astype(np.int64)          # the "index": offset/size per read (Appendix A.5)
    z_total = int(rec_off[n_reads])
    max_in_len = int(np.diff(rec_off[: n_reads + 1]).max()) - 8
    rng = np.random.default_rng(1)
    ids = rng.integers(0, n_reads, args.get_reads)
    K = args.get_batch
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    sig_cap = (n + 7) // 8 * 8
    st = b._stream()
    ev = make_events(L, _lib, 2)
    L.s5gpu_decode_scratch_bytes.restype = C.c_uint64
    L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
    scratch_bytes = int(L.s5gpu_decode_scratch_bytes(pay_cap))
    scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    sig = torch.empty(K * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(K * 64, dtype=torch.uint8, device=dev)
    desc_dev = torch.empty(K * _lib.REC_DESC.itemsize, dtype=torch.uint8, device=dev)
    src_sig = b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)

    def args_for(nrec, desc_t, sig_t, fields_t, in_ptr, payload_t=None):
        a = _lib.DecodeArgs()
        a.n_recs, a.rec_method, a.sig_method = nrec, 1, 1
        a.desc, a.in_, a.sig_out, a.fields = desc_t.data_ptr(), in_ptr, sig_t.data_ptr(), fields_t.data_ptr()
        if payload_t is None:
            a.flags, a.payload, a.payload_bytes, a.max_pay_cap = _lib.DEC_NO_PAYLOAD, scratch.data_ptr(), scratch_bytes, pay_cap
            a.max_in_len = max_in_len      # what the .idx says: the longest record of the file (records of one inflate window stay in LDS: include/slow5gpu.h)
        else:
            a.payload, a.max_pay_cap = payload_t.data_ptr(), pay_cap     # (the hint that the records are short: the kernel's 24-wave shape)
        return a

    def descs(sel, with_payload):
        k = len(sel)
        d = np.zeros(k, dtype=_lib.REC_DESC)
        d["in_off"] = rec_off[sel] + 8
        d["in_len"] = rec_off[sel + 1] - rec_off[sel] - 8
        if with_payload:
            d["pay_off"] = np.arange(k, dtype=np.uint64) * pay_cap
            d["pay_cap"] = pay_cap
        d["sig_off"] = np.arange(k, dtype=np.uint64) * sig_cap
        d["sig_cap"] = sig_cap
        return d

    a_get = args_for(K, desc_dev, sig, fields, b.stream_out.data_ptr())
    kern = []

    # The batch's descriptors: a pinned block of the library's (s5gpu_host_alloc) filled in place — only where a record lies and how
    # long it is changes from batch to batch — and sent with hipMemcpyAsync on the launch stream, as a C caller would.  (Round 2 built a
    # fresh numpy block per batch and copied it from pageable memory through torch: 0.03 ms of the 0.27 ms batch; torch's own pinned +
    # non_blocking path stalls ~90 ms every few batches on this stack — measured — which has nothing to do with the decode.)
    hip = None
    try:
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.s5gpu_host_alloc.restype = C.c_void_p
        L.s5gpu_host_alloc.argtypes = [C.c_size_t]
        L.s5gpu_host_free.argtypes = [C.c_void_p]
        pin_ptr = L.s5gpu_host_alloc(K * _lib.REC_DESC.itemsize)
        if not pin_ptr:
            hip = None
    except Exception:
        hip = None
    if hip is not None:
        pin = np.ctypeslib.as_array((C.c_uint8 * (K * _lib.REC_DESC.itemsize)).from_address(pin_ptr)).view(_lib.REC_DESC)
        pin[:] = descs(np.zeros(K, dtype=np.int64), False)          # sig_off / sig_cap of slot k: the same for every batch
        pin_in_off, pin_in_len = pin["in_off"], pin["in_len"]

    def run_batch(sel, timed_=False):
        if hip is not None:
            k = len(sel)
            o = rec_off[sel]
            np.add(o, 8, out=pin_in_off[:k], casting="unsafe")
            np.subtract(rec_off[sel + 1], o + 8, out=pin_in_len[:k], casting="unsafe")
            rc = hip.hipMemcpyAsync(desc_dev.data_ptr(), pin_ptr, k * _lib.REC_DESC.itemsize, 1, st)
            assert rc == 0, "hipMemcpyAsync failed (%d)" % rc
            d = None
        else:
            d = descs(sel, False)
            desc_dev[: d.nbytes].copy_(torch.from_numpy(d.view(np.uint8)))
        a_get.n_recs = len(sel)
        if timed_:
            L.s5gpu_event_record(ev[0], st)
        _lib.check(L.s5gpu_decode_dev(C.byref(a_get), st), "s5gpu_decode_dev")
        if timed_:
            L.s5gpu_event_record(ev[1], st)
        torch.cuda.synchronize()
        if timed_:
            kern.append((elapsed_ms(L, _lib, ev[0], ev[1]), int((rec_off[sel + 1] - rec_off[sel]).sum())))

    batches = [ids[lo:lo + K] for lo in range(0, len(ids), K)]
    for sel in batches[:2]:      # warm-up
        run_batch(sel)
    # pass 1: latency, nothing but the decode between the clock reads; the id list is walked again until about a second has gone by
    lat, done, passes = [], 0, 0
    t_lat0 = time.perf_counter()

    def latency_passes():
        nonlocal done, passes
        while passes < 3 or (time.perf_counter() - t_lat0 < args.min_leg_seconds and passes < 200):
            for sel in batches:
                t0 = time.perf_counter()
                run_batch(sel)
                if len(sel) == K:
                    lat.append(time.perf_counter() - t0)
                done += len(sel)
            passes += 1

    dt_get = timed(shard, torch, dev, latency_passes)
    done_all = shard.sum_over_ranks(done, device=TIMING_DEV or dev)   # ids decoded by all ranks in that time
    # pass 2: the same batches again, kernel time by HIP events on the launch stream, every decoded signal compared with
    # the generator (untimed: the comparison allocates)
    ok = True
    for sel in batches:
        k = len(sel)
        run_batch(sel, timed_=True)
        stt = fields[: k * 64].view(torch.int32).view(k, 16)[:, 0]
        got = sig[: k * sig_cap].view(k, sig_cap)[:, :n]
        want = src_sig[torch.from_numpy(sel).to(dev)][:, :n]
        ok &= bool((stt == 0).all().item()) and bool((got == want).all().item())
    bulk = None
    if not args.decode_batches_only:     # (--decode-batches-only: the K-sized batches alone — their own kernel-stats CSV and PMC pass)
        # pass 3: the whole index in one call (what `view` / `merge` decode per batch when the batch is large)
        if hip is not None:
            del pin, pin_in_off, pin_in_len
            L.s5gpu_host_free(pin_ptr)
        del sig, fields, desc_dev
        bulk = None
        d = descs(np.arange(n_reads), True)
        big_desc = torch.from_numpy(d.view(np.uint8)).to(dev)
        big_sig = torch.empty(n_reads * sig_cap + 64, dtype=torch.int16, device=dev)
        big_fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
        a_bulk = args_for(n_reads, big_desc, big_sig, big_fields, b.stream_out.data_ptr())

        def bulk_call(a, reps_min, secs):
            ts = []
            t0 = time.perf_counter()
            while len(ts) < reps_min or (time.perf_counter() - t0 < secs and len(ts) < 400):
                L.s5gpu_event_record(ev[0], st)
                _lib.check(L.s5gpu_decode_dev(C.byref(a), st), "s5gpu_decode_dev")
                L.s5gpu_event_record(ev[1], st)
                torch.cuda.synchronize()
                ts.append(elapsed_ms(L, _lib, ev[0], ev[1]))
            return ts

        ts = bulk_call(a_bulk, 3, args.min_leg_seconds)
        ms = float(np.mean(ts[1:]))
        stt = big_fields.view(torch.int32).view(n_reads, 16)[:, 0]
        same = bool((stt == 0).all().item()) and bool(torch.equal(big_sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n], src_sig[:, :n]))
        ok &= same
        alg = z_total + 2 * n * n_reads                         # Z + 2N per record (decode)
        dtraffic, dtraffic_src = pmc_traffic("k_inflate_par_np", n, n_reads)
        bulk = {"reads": n_reads, "calls": len(ts) - 1, "ms": round(ms, 3), "ms_min": round(min(ts[1:]), 3), "reads_per_s": round(n_reads / ms * 1e3, 1),
                "raw_signal_GB_per_s": round(n_reads * 2 * n / ms / 1e6, 2), "roundtrip_identical": same,
                "roofline": {"bound": "hbm", "kernel": "k_inflate_par_np (inflate + parse + svb-zd unpack, one launch)", "achieved": round(alg / ms / 1e6, 2), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round(alg / ms / 1e6 / PEAK_HBM_GBS, 5), "traffic": dtraffic, "traffic_source": dtraffic_src, "algorithmic_bytes_per_launch": alg,
                             "issue": pmc_issue("k_inflate_par_np", n, n_reads / ms * 1e3)}}
        # ... the same call in the form that also writes every uncompressed record out (what the view / merge worker needs)
        try:
            big_pay = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device=dev)
            a_full = args_for(n_reads, big_desc, big_sig, big_fields, b.stream_out.data_ptr(), payload_t=big_pay)
            big_sig.zero_()
            ts = bulk_call(a_full, 4, 0.3)
            msf = float(np.mean(ts[1:]))
            stt = big_fields.view(torch.int32).view(n_reads, 16)[:, 0]
            samef = bool((stt == 0).all().item()) and bool(torch.equal(big_sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n], src_sig[:, :n]))
            ok &= samef
            ftraffic, ftraffic_src = pmc_traffic("k_inflate_par+k_unpack", n, n_reads)
            bulk["with_payload_output"] = {"ms": round(msf, 3), "reads_per_s": round(n_reads / msf * 1e3, 1), "roundtrip_identical": samef,
                                           "roofline_frac": round(alg / msf / 1e6 / PEAK_HBM_GBS, 5), "traffic": ftraffic, "traffic_source": ftraffic_src}
            del big_pay
        except torch.OutOfMemoryError:
            bulk["with_payload_output"] = None
        # ---- the same call on records WRITTEN BY STOCK ZLIB (what the reference's own files hold: level 6, arbitrary LZ77
        # distances): 2048 distinct records compressed on the CPU, tiled to 262 144; every decoded signal compared ----
        try:
            import zlib
            distinct, nb = 2048, min(262144, n_reads)
            hostsig = src_sig[:distinct, :n].cpu().numpy()
            streams = []
            for i in range(distinct):
                rec, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 3.0, 1400.0, 4000.0, np.ascontiguousarray(hostsig[i]))
                streams.append(zlib.compress(ob.rec_pack(rec, ob.SIG_SVB_ZD), 6))
            zl = np.array([len(x) for x in streams], dtype=np.int64)
            zo = np.concatenate([[0], np.cumsum((zl + 15) // 16 * 16)])
            blob = np.zeros(zo[-1] + 64, dtype=np.uint8)
            for x, o_ in zip(streams, zo[:-1]):
                blob[o_:o_ + len(x)] = np.frombuffer(x, dtype=np.uint8)
            zin = torch.from_numpy(blob).to(dev)
            idx = np.arange(nb) % distinct
            d2 = d[:nb].copy()
            d2["in_off"] = zo[idx]; d2["in_len"] = zl[idx]
            zdesc = torch.from_numpy(d2.view(np.uint8)).to(dev)
            a_z = args_for(nb, zdesc, big_sig, big_fields, zin.data_ptr())
            a_z.max_in_len = int(zl.max())
            big_sig.zero_()
            ts = bulk_call(a_z, 3, args.min_leg_seconds / 2)
            ms2 = float(np.mean(ts[1:]))
            stt = big_fields.view(torch.int32).view(n_reads, 16)[:nb, 0]
            got = big_sig[: nb * sig_cap].view(nb, sig_cap)[:, :n]
            want = src_sig[:distinct, :n]
            same2 = bool((stt == 0).all().item()) and all(bool(torch.equal(got[k0:k0 + distinct], want[: min(distinct, nb - k0)])) for k0 in range(0, nb, distinct))
            bulk["stock_zlib_records"] = {"reads": nb, "distinct": distinct, "calls": len(ts) - 1, "ms": round(ms2, 3), "reads_per_s": round(nb / ms2 * 1e3, 1),
                                          "raw_signal_GB_per_s": round(nb * 2 * n / ms2 / 1e6, 2), "roundtrip_identical": same2,
                                          "what": "svb-zd records compressed by zlib %s level 6 on the CPU (the reference's writer), decoded by the same call" % zlib.ZLIB_VERSION}
            ok &= same2
            del zin, zdesc
        except Exception as e:      # (never fatal for the line)
            bulk["stock_zlib_records"] = {"error": repr(e)}
        del big_desc, big_sig, big_fields, scratch
    else:
        if hip is not None:
            del pin, pin_in_off, pin_in_len
            L.s5gpu_host_free(pin_ptr)
        del sig, fields, desc_dev, scratch
    for e in ev:
        L.s5gpu_event_destroy(e)
    if rank != 0:
        return None
    lat_ms = np.array(lat) * 1e3
    full = [(m, z) for (m, z), sel in zip(kern, batches) if len(sel) == K]
    k_ms = float(np.mean([m for m, _ in full])) if full else None
    k_alg = float(np.mean([z for _, z in full])) + 2 * n * K if full else None

    # ---- CPU baseline beside it: the oracle's pthread get --benchmark shape (inflate + svb-zd decode per id), thread sweep
    # as /root/reference/test/bench/simple_bench.sh:75-104 ----
    cpu = None
    if want_cpu:
        cores = os.cpu_count() or 1
        stream_h = b.stream_out[:z_total].cpu().numpy()
        off_h = rec_off[:-1].astype(np.uint64)
        ids32 = ids.astype(np.uint32)
        sweep = []
        per_point = max(1.5, args.cpu_seconds / 4)
        for t in sorted({1, min(8, cores), min(32, cores), min(64, cores), min(128, cores), cores}):
            got_, secs = 0, 0.0
            while secs < per_point:
                tot, s, _ = ob.decode_batch_mt(stream_h, off_h, ids32, t, K)
                assert tot == len(ids32) * n, "CPU decode failed"
                got_ += len(ids32); secs += s
            sweep.append({"t": t, "reads_per_s": round(got_ / secs, 1), "GB_per_s": round(got_ * 2 * n / secs / 1e9, 3), "seconds": round(secs, 1)})
        ref = [x for x in sweep if x["t"] == cores][0]
        best = max(sweep, key=lambda x: x["reads_per_s"])
        cpu = {"value": ref["GB_per_s"], "unit": "GB/s", "cores": cores, "kind": "port", "reads_per_s": ref["reads_per_s"],
               "shape": "get --benchmark -t %d -K %d: per id inflate (per-record inflateInit) + parse + svb-zd decode, threads created per batch; preads excluded" % (cores, K),
               "best_of": {"value": best["GB_per_s"], "unit": "GB/s", "t": best["t"], "reads_per_s": best["reads_per_s"]},
               "sweep": sweep,
               "sample": "the same %d random ids (seed 1) over the same %d-read index, repeated until each point ran >= %.1f s" % (len(ids), n_reads, per_point)}

    return {"workload": "BASELINE configs[4]: random get decode (inflate + svb-zd unpack), %d ids (seed 1) over a %d-read index, batches of %d, %d samples/read, "
                        "fields + signals out (S5GPU_DEC_NO_PAYLOAD); %d passes over the id list" % (len(ids), n_reads, K, n, passes),
            "metric": "blow5_get_decode_throughput", "value": round(done_all * 2 * n / dt_get / 1e9, 3), "unit": "GB/s", "n_gpus": world, "scaling": "weak",
            "reads_per_s": round(done_all / dt_get, 1), "dtype": "u8->int16",
            "batch_latency_ms": {"p50": round(float(np.percentile(lat_ms, 50)), 3), "p99": round(float(np.percentile(lat_ms, 99)), 3), "batches": len(lat_ms)},
            "per_read_latency_us_p50": round(float(np.percentile(lat_ms, 50)) * 1e3 / K, 3),
            "kernel_ms_per_batch": round(k_ms, 4) if k_ms else None,
            "roofline": {"bound": "hbm", "kernel": "k_inflate_par_np (K = %d)" % K, "achieved": round(k_alg / k_ms / 1e6, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(k_alg / k_ms / 1e6 / PEAK_HBM_GBS, 5), "traffic": pmc_traffic("k_inflate_par_np@K", n, K)[0], "traffic_source": pmc_traffic("k_inflate_par_np@K", n, K)[1],
                         "algorithmic_bytes_per_launch": int(k_alg)} if k_ms else None,
            "bulk_decode_one_call": bulk,
            "cpu_baseline": cpu,
            "roundtrip_identical": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--samples", type=int, default=4000, help="int16 samples per read")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline, reference-shaped point: seconds (0 = skip)")
    ap.add_argument("--cpu-sweep-seconds", type=float, default=5.0, help="CPU baseline: seconds per point of the -t / -K sweep (0 = skip)")
    ap.add_argument("--svb-only", action="store_true", help="configs[1] alone: svb-zd stage")
    ap.add_argument("--two-pass", action="store_true", help="encode into worst-case slots + compaction pass instead of the ordered single-pass stream")
    ap.add_argument("--fused-cap", type=int, default=8192, help="--mixed: LDS payload budget of the fused kernel, bytes (longer payloads take the staged path)")
    ap.add_argument("--mixed", action="store_true", help="read lengths of a real run: log-normal, median 6000 samples (--reads reads, default 262144)")
    ap.add_argument("--long", action="store_true", help="configs[3] alone (the long-read leg as the whole line)")
    ap.add_argument("--no-long", action="store_true", help="skip the configs[3] leg of the default run")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[1] and configs[4] legs of the default run")
    ap.add_argument("--long-reads", type=int, default=65536, help="configs[3] leg: size of the read-index space (all ranks together)")
    ap.add_argument("--long-samples", type=int, default=100_000)
    ap.add_argument("--long-chunk", type=int, default=16384, help="configs[3] leg: reads per launch (at most; a rank's shard is cut into >= 4 chunks)")
    ap.add_argument("--long-streams", type=int, default=2, help="configs[3] leg: 2 = two output buffer sets, pack / deflate / compaction of successive chunks on three streams; 1 = one stream, one set")
    ap.add_argument("--decode", action="store_true", help="configs[4] alone: random get-style decode (inflate + svb-zd unpack)")
    ap.add_argument("--decode-batches-only", action="store_true", help="configs[4]: only the K-sized get batches (no bulk call, no stock-zlib records)")
    ap.add_argument("--get-reads", type=int, default=100_000, help="configs[4]: random read ids to fetch (seed 1)")
    ap.add_argument("--get-batch", type=int, default=4096, help="configs[4]: ids per batch (-K)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-read-lengths leg of the default run")
    ap.add_argument("--mixed-reads", type=int, default=262144, help="mixed leg: reads per GPU")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (file to file) object of the default run")
    ap.add_argument("--e2e-reads", type=int, default=1_000_000, help="e2e: reads in the files (4000 samples each; /dev/shm)")
    ap.add_argument("--e2e-cpu-reads", type=int, default=262_144, help="e2e: records of the same files the CPU twins convert per point")
    ap.add_argument("--min-leg-seconds", type=float, default=1.2, help="every GPU leg keeps the device busy for about this long at least")
    ap.add_argument("--min-leg-steps-svb", type=int, default=300, help="configs[1] leg: steps (2.8 ms each on 1 M reads)")
    ap.add_argument("--min-leg-steps-long", type=int, default=40, help="configs[3] leg: steps (28 ms each on 65536 reads)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): make the N ranks here — the same module the driver
    # uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), this file and these
    # arguments again — so that `--gpus 8` can never quietly be eight copies' worth of one rank (`n_gpus: 1`).  Rank 0 prints the line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # S5BENCH_ALIAS_DEVICES=1 (pre-flight of the N > 1 path on a one-GPU box, tests/test_multi_device.py): ranks share device 0
    alias = os.environ.get("S5BENCH_ALIAS_DEVICES", "") not in ("", "0")
    if alias:
        local_rank = 0
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if alias:
            dist.init_process_group("gloo", rank=rank, world_size=world)      # (RCCL refuses two ranks on one device)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        t = torch.ones(1, dtype=torch.int32, device="cpu" if alias else dev)
        dist.all_reduce(t)                     # the one collective of the run that is not a timing barrier: who is here
        ranks_seen = int(t.item())

    import oracle_bind as ob
    from slow5tools_amd import _lib, press, shard

    L = _lib.lib()
    _lib.check(L.s5gpu_init(local_rank), "s5gpu_init")
    for kv in filter(None, os.environ.get("S5BENCH_OPTIONS", "").split(",")):   # tools: library options for A/B runs, e.g. staged_fused=0
        k, v = kv.split("=")
        _lib.check(L.s5gpu_set_option(k.encode(), int(v)), "s5gpu_set_option " + kv)
    K = args.steps
    global TIMING_DEV
    TIMING_DEV = "cpu" if alias else dev

    def finish(line):
        if line is not None:
            print(json.dumps(line))
        if world > 1:
            dist.destroy_process_group()

    if args.long:
        leg = long_leg(args, L, _lib, press, shard, ob, rank, world, dev, K, args.warmup)
        if rank != 0:
            return finish(None)
        line = {"metric": "blow5_encode_raw_signal_throughput", "value": leg["value"], "unit": "GB/s", "n_gpus": world, "steps": leg["steps"],
                "warmup": args.warmup, "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "int16->u8", "data": "synthetic", "config": {"workload": leg["workload"], "record_press": "zlib", "signal_press": "svb-zd",
                                                                      "parallelism": "read-index space sharded over %d GPU(s), no collective" % world}}
        line.update({k: leg[k] for k in ("reads_per_s", "bytes_per_sample", "parity_spot_check", "kernel_ms", "roofline", "scale_factor_vs_10M_reads")})
        line["cpu_baseline"] = None
        return finish(line)

    n = args.samples
    if args.mixed:
        n_reads = args.reads if args.reads != 1_000_000 else 262144
        rng = np.random.default_rng(5)
        ns = np.clip(np.exp(rng.normal(np.log(6000), 0.9, n_reads)), 200, 400000).astype(np.uint64)
        if os.environ.get("S5BENCH_MIXED_SORT"):          # tools: what the order of the reads is worth (longest first / shortest first)
            ns = np.sort(ns)[::-1].copy() if os.environ["S5BENCH_MIXED_SORT"] == "desc" else np.sort(ns)
        # the device entry point only knows the longest read; the host batch calls name this budget from the lengths themselves
        b = press.DeviceBatch(ns, device=dev, lds_payload_cap=args.fused_cap)
        tot = b.sig.numel()
        # one long synthetic trace cut into the reads (the event model is position-keyed)
        _lib.check(L.s5gpu_synth_dev(b.sig.data_ptr(), 1, tot - 64, tot, 0x5105 + rank, 0, b._stream()), "synth")
        _lib.check(L.s5gpu_synth_hdr_dev(b.hdr.data_ptr(), n_reads, rank * n_reads, b._stream()), "hdr")
        raw_bytes = int(2 * ns.sum())
    else:
        n_reads = args.reads
        ns = None
        first = rank * n_reads                      # each rank encodes its own shard of the read index space
        b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), device=dev)
        b.synth(seed=0x5105, first=first)
        raw_bytes = 2 * n * n_reads
    torch.cuda.synchronize()

    if args.decode:     # configs[4] alone: the records of one encode pass, then the decode leg as the whole line
        b.encode_stream()
        torch.cuda.synchronize()
        assert b.stream_ok(), "a read overflowed the LDS budget: --decode needs the single-pass stream"
        leg = decode_leg(args, L, _lib, shard, ob, b, n_reads, n, rank, world, dev, world == 1 and args.cpu_seconds > 0)
        if rank != 0:
            return finish(None)
        line = {"metric": leg["metric"], "value": leg["value"], "unit": "GB/s", "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": leg["dtype"], "data": "synthetic", "config": {"workload": leg["workload"]}}
        line.update({k: v for k, v in leg.items() if k not in ("metric", "value", "unit", "n_gpus", "scaling", "dtype", "workload")})
        return finish(line)

    # Full encode, default: ONE launch per step — k_encode_stream writes every record straight to its place in the
    # contiguous BLOW5 record stream (ordered single pass, decoupled look-back).  It needs every read to fit the LDS
    # budget (true for 4000-sample reads); long reads, --two-pass, --mixed and --svb-only use slots + the compaction pass.
    single_pass = not args.svb_only and not args.two_pass and not args.mixed and b.tot["max_payload"] * 100 // 325 <= 16384

    def run_step():
        if single_pass:
            b.encode_stream()
        else:
            (b.svbzd_encode if args.svb_only else b.encode)()

    def second_half():
        if not single_pass:
            b.compact()

    st = b._stream()
    for _ in range(args.warmup):
        run_step()
        second_half()
    torch.cuda.synchronize()
    if single_pass and not b.stream_ok():   # a read overflowed the LDS budget: the stream is invalid, use the two-pass path
        single_pass = False
        run_step(); second_half()
        torch.cuda.synchronize()

    evs = make_events(L, _lib, 3 * K)

    def k_steps():
        for k in range(K):
            L.s5gpu_event_record(evs[3 * k], st)
            run_step()
            L.s5gpu_event_record(evs[3 * k + 1], st)
            second_half()
            L.s5gpu_event_record(evs[3 * k + 2], st)

    dt = timed(shard, torch, dev, k_steps)
    enc_ms = [elapsed_ms(L, _lib, evs[3 * k], evs[3 * k + 1]) for k in range(K)]
    cmp_ms = [elapsed_ms(L, _lib, evs[3 * k + 1], evs[3 * k + 2]) for k in range(K)]
    for e in evs:
        L.s5gpu_event_destroy(e)

    default_shape = not (args.svb_only or args.mixed or args.two_pass) and n == 4000
    # the same step again for about a second (not what `value` is computed on: that is the K steps above)
    sustained = None
    if default_shape and args.min_leg_seconds > 0:
        reps = max(1, int(args.min_leg_seconds / max(dt / K, 1e-6)))
        reps = min(reps, 2000)
        dts = timed(shard, torch, dev, lambda: [(run_step(), second_half()) for _ in range(reps)])
        sustained = {"steps": reps, "seconds": round(dts, 3), "value": round(raw_bytes * world * reps / dts / 1e9, 3), "unit": "GB/s"}

    main_out_len = b.out_len[:n_reads].cpu().numpy().astype(np.int64)
    main_recs = None
    sig_cpu_t = None
    if rank == 0:
        idx = [0, 1, n_reads // 2, n_reads - 1] if n_reads >= 4 else list(range(n_reads))
        main_recs = (idx, b.stream_records(idx) if single_pass else b.records(idx), b.stream_ok() if single_pass else True)
        if world == 1 and args.cpu_seconds > 0 and not args.svb_only and not args.mixed:   # the CPU baseline is an N = 1 figure
            stride = (n + 7) // 8 * 8
            m = min(n_reads, 262144)
            sig_cpu_t = b.sig[: m * stride].cpu().numpy().reshape(m, stride)[:, :n].copy()

    # ---- the other configs of BASELINE.json as legs of the same line (every rank takes part) ----
    leg1 = leg3 = leg4 = None
    if default_shape and not args.no_legs:
        if single_pass:     # configs[4] decodes the records the headline wrote; it runs before configs[1] reuses the stream buffer
            leg4 = decode_leg(args, L, _lib, shard, ob, b, n_reads, n, rank, world, dev, world == 1 and args.cpu_seconds > 0)
        leg1 = svb_leg(args, L, _lib, shard, ob, b, n_reads, n, rank, world, dev, K, args.warmup)
    del b                                       # free the main batch before the other legs allocate
    torch.cuda.empty_cache()
    if default_shape and not args.no_long:
        leg3 = long_leg(args, L, _lib, press, shard, ob, rank, world, dev, K, args.warmup)
    leg_mixed = None
    if default_shape and not args.no_mixed:
        leg_mixed = mixed_leg(args, L, _lib, press, shard, rank, world, dev)

    if rank != 0:
        return finish(None)

    # ---- N > 1: the HOST-FED curve beside the device-resident one (round 5): one process, one batch call, all N devices ----
    host_fed = None
    if default_shape and world > 1 and not args.no_e2e:
        import bench_e2e
        # (every rank has left its legs — each ends on a barrier — and the other ranks are on their way out: rank 0 goes on alone)
        try:
            host_fed = bench_e2e.host_fed_multi(world, n_reads, n, bool(os.environ.get("S5BENCH_ALIAS_DEVICES")))
        except Exception as e:
            host_fed = {"error": repr(e)}

    # ---- end to end through files + the host-buffer (PCIe-inclusive) batch call: N = 1 figures, like the CPU baseline ----
    e2e_obj = pcie_obj = None
    if default_shape and world == 1 and not args.no_e2e:
        import bench_e2e
        torch.cuda.empty_cache()
        try:
            e2e_obj = bench_e2e.e2e(args, L, _lib, press, torch, dev, ob, want_cpu=args.cpu_seconds > 0)
        except Exception as e:      # (never fatal for the line)
            e2e_obj = {"error": repr(e)}
        try:
            pcie_obj = bench_e2e.pcie_inclusive(L, _lib, press, n_reads, n)
        except Exception as e:
            pcie_obj = {"error": repr(e)}

    # ---- results, rank 0 ----
    z_bytes = int(main_out_len.sum())
    total_reads = n_reads * world
    reads_per_s = total_reads * K / dt
    value = raw_bytes * world * K / dt / 1e9
    # algorithmic HBM bytes of the dominant kernel per launch:
    #   full encode: 2N + H (74-byte head) + Z (record incl. 8-byte prefix), per read; svb only: 2N + S
    alg_bytes = raw_bytes + (0 if args.svb_only else 74 * n_reads) + z_bytes
    kern_s = float(np.mean(enc_ms)) / 1e3
    achieved = alg_bytes / kern_s / 1e9
    if args.svb_only:
        kernel = "k_svbzd_encode"
    elif single_pass:
        kernel = "k_encode_stream"
    elif args.mixed:
        kernel = "k_encode_fused+k_pack+k_deflate_staged"
    elif (n * 13 // 4) * 100 // 325 <= 4 * 16384:
        kernel = "k_encode_fused"
    else:
        kernel = "k_pack+k_deflate_staged"
    traffic, traffic_src = (None, None) if args.mixed else pmc_traffic(kernel, n, n_reads)

    # parity spot check of this very run (outside the timed region)
    import zlib

    idx, recs, parity = main_recs
    if not args.mixed:
        for i, rec in zip(idx, recs):
            sig = ob.synth_read(0x5105, rank * n_reads + i, n)
            if args.svb_only:
                parity &= rec == ob.svbzd_encode(sig)
            else:
                r, keep = ob.make_rec(ob.synth_read_id(rank * n_reads + i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
                parity &= zlib.decompress(rec[8:]) == ob.rec_pack(r, ob.SIG_SVB_ZD)
    else:
        for i, rec in zip(idx, recs):          # mixed lengths: stock zlib must inflate every sampled record to a payload of the right shape
            pay = zlib.decompress(rec[8:])
            parity &= len(pay) > 82 and int.from_bytes(pay[82:86], "little") == int(ns[i])

    cpu = None
    if sig_cpu_t is not None:
        cpu = cpu_encode_baseline(ob, sig_cpu_t, 0, n, args.cpu_seconds, args.cpu_sweep_seconds)

    if args.mixed:
        workload = "read lengths of a real run: %d reads, log-normal lengths (median %d, max %d samples, %.2f G samples), full BLOW5 encode, 8 KiB fused budget" % (
            n_reads, int(np.median(ns)), int(ns.max()), ns.sum() / 1e9)
    else:
        workload = "BASELINE configs[%d]: %s, %d reads x %d int16 samples per GPU" % (
            1 if args.svb_only else 2, "svb-zd only" if args.svb_only else "full BLOW5 encode (svb-zd + DEFLATE, zlib framing)", n_reads, n)
    line = {
        "metric": "blow5_encode_raw_signal_throughput" if not args.svb_only else "svbzd_encode_raw_signal_throughput",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16->u8", "data": "synthetic",
        "config": {"workload": workload, "reads_per_gpu": n_reads, "samples_per_read": n if not args.mixed else "mixed",
                   "record_press": "none" if args.svb_only else "zlib",
                   "signal_press": "svb-zd", "parallelism": "reads sharded over %d GPU(s), no collective" % world},
        "ranks_seen": ranks_seen,
        "reads_per_s": round(reads_per_s, 1),
        "bytes_per_sample": round(z_bytes / (raw_bytes / 2), 4),
        "parity_spot_check": bool(parity),
        "kernel_ms": {"encode": round(float(np.mean(enc_ms)), 3), "compact": round(float(np.mean(cmp_ms)), 3) if not single_pass else 0.0},
        "output": "ordered single-pass record stream (k_encode_stream)" if single_pass else "worst-case slots + compaction pass",
        "roofline": {"bound": "hbm", "kernel": kernel,
                     "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # the nominal roofline above is HBM (the bytes are what the job is); the ceiling that BINDS the kernel is printed next to it
                     "issue": pmc_issue("k_encode_stream", n, n_reads / (float(np.mean(enc_ms)) * 1e-3)) if single_pass else None},
        "sustained": sustained,
        "cpu_baseline": cpu,
        "configs1": leg1,
        "configs3": leg3,
        "configs4": leg4,
        "mixed": leg_mixed,
        "e2e": e2e_obj,
        "pcie_inclusive": pcie_obj,
        "host_fed": host_fed,
    }
    # the other legs' headline numbers in ONE small object inside `roofline` (the driver's record keeps roofline / cpu_baseline / config whole and
    # only the NAMES of further top-level keys): every figure below is measured in this run, its full object is the top-level key of the same name
    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    line["roofline"]["legs"] = {
        "mixed_GB_per_s": pick(leg_mixed, "value"), "mixed_frac": pick(leg_mixed, "roofline", "frac"),
        "configs3_GB_per_s": pick(leg3, "value"), "configs3_frac": pick(leg3, "roofline", "frac"),
        "configs4_k4096_batch_latency_ms": pick(leg4, "batch_latency_ms"), "configs4_k4096_reads_per_s": pick(leg4, "reads_per_s"),
        "configs4_k4096_frac": pick(leg4, "roofline", "frac"),
        "bulk_decode_one_call": {k: pick(leg4, "bulk_decode_one_call", k) for k in ("reads", "ms", "reads_per_s")},
        "bulk_decode_frac": pick(leg4, "bulk_decode_one_call", "roofline", "frac"),
        "stock_zlib_records_reads_per_s": pick(leg4, "bulk_decode_one_call", "stock_zlib_records", "reads_per_s"),
        "pcie_inclusive_GB_per_s": {k: {"malloc": pick(pcie_obj, k, "GB_per_s"), "arena": pick(pcie_obj, k, "arena", "GB_per_s"),
                                        "two_in_flight": pick(pcie_obj, k, "two_in_flight", "GB_per_s")}
                                    for k in ("batch_4096", "batch_10000", "batch_65536", "batch_%d" % n_reads)} if isinstance(pcie_obj, dict) else None,
        "e2e_whole_process_s": {k: pick(e2e_obj, k, "gpu", "whole_process_s") for k in ("slow5_to_blow5", "blow5_to_blow5")} if isinstance(e2e_obj, dict) else None,
        "e2e_get_100k_whole_process_s": pick(e2e_obj, "get_100k", "gpu", "benchmark", "whole_process_s"),
        "host_fed_all_devices_GB_per_s": {k: pick(host_fed, k, "arena", "GB_per_s") for k in ("batch_65536", "batch_%d" % n_reads)} if isinstance(host_fed, dict) else None,
    }
    finish(line)


if __name__ == "__main__":
    main()
