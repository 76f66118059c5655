#!/usr/bin/env python3
"""bench.py — BLOW5 encode throughput (svb-zd + DEFLATE, BASELINE.json config 3) on N MI355X.

A step = one pass of the hot path over one resident batch of synthetic reads, ending in the contiguous BLOW5
record stream the ordered fwrite loop emits:
    k_encode_stream (svb-zd -> pack -> DEFLATE -> zlib frame, one read per workgroup, records placed by a
                     decoupled look-back: one launch)                                   [default]
    or k_encode_fused into worst-case slots + s5gpu_compact (--two-pass, long reads, --svb-only)
Inputs (int16 signals, 74-byte record heads) are already in HBM when the timed region starts.
Reads shard across ranks with no collective (weak scaling: every rank encodes its own batch).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--samples", type=int, default=4000, help="int16 samples per read")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--svb-only", action="store_true", help="config 2: svb-zd stage alone")
    ap.add_argument("--two-pass", action="store_true", help="encode into worst-case slots + compaction pass instead of the ordered single-pass stream")
    ap.add_argument("--decode", action="store_true", help="config 5: random get-style decode (inflate + svb-zd unpack)")
    ap.add_argument("--get-reads", type=int, default=100_000, help="--decode: random read ids to fetch (seed 1)")
    ap.add_argument("--get-batch", type=int, default=4096, help="--decode: ids per batch (-K)")
    args = ap.parse_args()
    if args.decode:
        return bench_decode(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    from slow5tools_amd import _lib, press, shard

    L = _lib.lib()
    _lib.check(L.s5gpu_init(local_rank), "s5gpu_init")

    n_reads, n = args.reads, args.samples
    first = rank * n_reads                      # each rank encodes its own shard of the read index space
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), device=dev)
    b.synth(seed=0x5105, first=first)
    torch.cuda.synchronize()

    # Full encode, default: ONE launch per step — k_encode_stream writes every record straight to its place in the
    # contiguous BLOW5 record stream (ordered single pass, decoupled look-back).  It needs every read to fit the LDS
    # budget (true for 4000-sample reads); long reads, --two-pass and --svb-only use slots + the compaction pass.
    single_pass = not args.svb_only and not args.two_pass and b.tot["max_payload"] * 100 // 325 <= 16384

    def run_step():
        if single_pass:
            b.encode_stream()
        else:
            (b.svbzd_encode if args.svb_only else b.encode)()

    def second_half():
        if not single_pass:
            b.compact()

    st = b._stream()

    barrier = shard.barrier

    for _ in range(args.warmup):
        run_step()
        second_half()
    torch.cuda.synchronize()
    if single_pass and not b.stream_ok():   # a read overflowed the LDS budget: the stream is invalid, use the two-pass path
        single_pass = False
        run_step(); second_half()
        torch.cuda.synchronize()

    K = args.steps
    evs = []
    for _ in range(3 * K):
        e = C.c_void_p()
        _lib.check(L.s5gpu_event_create(C.byref(e)))
        evs.append(e)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        L.s5gpu_event_record(evs[3 * k], st)
        run_step()
        L.s5gpu_event_record(evs[3 * k + 1], st)
        second_half()
        L.s5gpu_event_record(evs[3 * k + 2], st)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, device=dev)

    enc_ms, cmp_ms = [], []
    ms = C.c_float()
    for k in range(K):
        _lib.check(L.s5gpu_event_elapsed_ms(evs[3 * k], evs[3 * k + 1], C.byref(ms)))
        enc_ms.append(ms.value)
        _lib.check(L.s5gpu_event_elapsed_ms(evs[3 * k + 1], evs[3 * k + 2], C.byref(ms)))
        cmp_ms.append(ms.value)
    for e in evs:
        L.s5gpu_event_destroy(e)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- results, rank 0 ----
    out_len = b.out_len[:n_reads].cpu().numpy().astype(np.int64)
    z_bytes = int(out_len.sum())
    raw_bytes = 2 * n * n_reads
    total_reads = n_reads * world
    reads_per_s = total_reads * K / dt
    value = reads_per_s * 2 * n / 1e9
    # algorithmic HBM bytes of the dominant kernel per launch:
    #   full encode: 2N + H (74-byte head) + Z (record incl. 8-byte prefix), per read; svb only: 2N + S
    alg_bytes = raw_bytes + (0 if args.svb_only else 74 * n_reads) + z_bytes
    kern_s = float(np.mean(enc_ms)) / 1e3
    achieved = alg_bytes / kern_s / 1e9
    peak = 8000.0
    # HBM traffic per launch from the committed PMC pass (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc.sh); only valid
    # for the kernel/config it was collected on
    traffic = None
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if not args.svb_only and t.get("samples_per_read") == n and t.get("kernel") == ("k_encode_stream" if single_pass else "k_encode_fused"):
            traffic = int(t["hbm_bytes_per_read"] * n_reads)
    except Exception:
        pass

    # parity spot check of this very run (outside the timed region)
    import zlib

    import oracle_bind as ob

    idx = [0, 1, n_reads // 2, n_reads - 1] if n_reads >= 4 else list(range(n_reads))
    recs = b.stream_records(idx) if single_pass else b.records(idx)
    parity = b.stream_ok() if single_pass else True
    for i, rec in zip(idx, recs):
        sig = ob.synth_read(0x5105, first + i, n)
        if args.svb_only:
            parity &= rec == ob.svbzd_encode(sig)
        else:
            r, keep = ob.make_rec(ob.synth_read_id(first + i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
            parity &= zlib.decompress(rec[8:]) == ob.rec_pack(r, ob.SIG_SVB_ZD)

    # ---- CPU baseline: the oracle's reference-shaped pthread batch encode on this box's host cores ----
    cpu = None
    if args.cpu_seconds > 0 and not args.svb_only:
        cores = os.cpu_count() or 1
        stride = (n + 7) // 8 * 8
        probe_reads = min(n_reads, 256 * cores)
        sig_probe = b.sig[: probe_reads * stride].cpu().numpy().reshape(probe_reads, stride)[:, :n]
        _, secs, _ = ob.encode_batch_mt(sig_probe, first, cores, 4096)
        rate = probe_reads / max(secs, 1e-6)
        m = int(min(n_reads, max(probe_reads, rate * args.cpu_seconds)))
        sig_cpu = b.sig[: m * stride].cpu().numpy().reshape(m, stride)[:, :n]
        tot, secs, _ = ob.encode_batch_mt(sig_cpu, first, cores, 4096)
        cpu = {"value": round(m * 2 * n / secs / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
               "reads_per_s": round(m / secs, 1), "bytes_per_sample": round(tot / (m * n), 4),
               "sample": "first %d reads of the same batch (%d samples each), compute phase of view -t %d -K 4096 "
                         "(svb-zd + zlib-1.2.11 level 6, per-record deflateInit), %.1f s" % (m, n, cores, secs)}

    line = {
        "metric": "blow5_encode_raw_signal_throughput" if not args.svb_only else "svbzd_encode_raw_signal_throughput",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16->u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: %s, %d reads x %d int16 samples per GPU"
                               % (1 if args.svb_only else 2, "svb-zd only" if args.svb_only else "full BLOW5 encode (svb-zd + DEFLATE, zlib framing)", n_reads, n),
                   "reads_per_gpu": n_reads, "samples_per_read": n, "record_press": "none" if args.svb_only else "zlib",
                   "signal_press": "svb-zd", "parallelism": "reads sharded over %d GPU(s), no collective" % world},
        "reads_per_s": round(reads_per_s, 1),
        "bytes_per_sample": round(z_bytes / (n_reads * n), 4),
        "parity_spot_check": bool(parity),
        "kernel_ms": {"encode": round(float(np.mean(enc_ms)), 3), "compact": round(float(np.mean(cmp_ms)), 3) if not single_pass else 0.0},
        "output": "ordered single-pass record stream (k_encode_stream)" if single_pass else "worst-case slots + compaction pass",
        "roofline": {"bound": "hbm", "kernel": "k_svbzd_encode" if args.svb_only else ("k_encode_stream" if single_pass else "k_encode_fused" if b.tot["max_payload"] * 100 // 325 <= 4 * 16384 else "k_pack+k_deflate_staged"),
                     "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes},
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def bench_decode(args):
    """BASELINE config 5: decode for random `get` over an index of --reads records, batches of --get-batch ids.
    Per batch (what src/get.c:321-386 does per -K batch, minus the preads): build the batch's record
    descriptors from the index, upload them, inflate + unpack on the GPU, synchronise.  Every decoded signal
    is compared with the generator.  Prints one JSON line (not the headline metric)."""
    import numpy as np
    import torch

    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    n_reads, n = args.reads, args.samples
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), device=dev)
    b.synth(seed=0x5105, first=0)
    b.encode()
    b.compact()
    torch.cuda.synchronize()
    rec_off = b.rec_off.cpu().numpy().astype(np.int64)          # the "index": offset/size per read (Appendix A.5)
    rng = np.random.default_rng(1)
    ids = rng.integers(0, n_reads, args.get_reads)
    K = args.get_batch
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    sig_cap = (n + 7) // 8 * 8
    payload = torch.empty(K * pay_cap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(K * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(K * 64, dtype=torch.uint8, device=dev)
    desc_dev = torch.empty(K * _lib.REC_DESC.itemsize, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.rec_method, a.sig_method = 1, 1
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc_dev.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
    lat, ok, done = [], True, 0
    t_all = time.perf_counter()

    def run_batch(sel):
        k = len(sel)
        d = np.zeros(k, dtype=_lib.REC_DESC)
        d["in_off"] = rec_off[sel] + 8
        d["in_len"] = rec_off[sel + 1] - rec_off[sel] - 8
        d["pay_off"] = np.arange(k, dtype=np.uint64) * pay_cap
        d["pay_cap"] = pay_cap
        d["sig_off"] = np.arange(k, dtype=np.uint64) * sig_cap
        d["sig_cap"] = sig_cap
        # plain synchronous H2D of the 160 KB descriptor block (torch's pinned + non_blocking path stalls ~90 ms every few
        # batches on this stack — measured, tools note in DESIGN.md — which has nothing to do with the decode)
        desc_dev[: d.nbytes].copy_(torch.from_numpy(d.view(np.uint8)))
        a.n_recs = k
        _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()), "s5gpu_decode_dev")
        torch.cuda.synchronize()

    # pass 1: latency, nothing but the decode between the clock reads (the first two batches are warm-up)
    for lo in range(0, len(ids), K):
        sel = ids[lo:lo + K]
        t0 = time.perf_counter()
        run_batch(sel)
        lat.append(time.perf_counter() - t0)
        if lo >= 2 * K:
            done += len(sel)
    # pass 2: the same batches again, every decoded signal compared with the generator (untimed: the comparison allocates)
    for lo in range(0, len(ids), K):
        sel = ids[lo:lo + K]
        k = len(sel)
        run_batch(sel)
        st = fields[: k * 64].view(torch.int32).view(k, 16)[:, 0]
        got = sig[: k * sig_cap].view(k, sig_cap)[:, :n]
        want = b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[torch.from_numpy(sel).to(dev)][:, :n]
        ok &= bool((st == 0).all().item()) and bool((got == want).all().item())
    # pass 3: the whole index in one call (what `view` / `merge` decode per batch when the batch is large): device time of
    # inflate + unpack by HIP events, every signal compared afterwards
    del payload, sig, fields, desc_dev
    torch.cuda.empty_cache()
    bulk = None
    try:
        d = np.zeros(n_reads, dtype=_lib.REC_DESC)
        d["in_off"] = rec_off[:-1] + 8
        d["in_len"] = rec_off[1:] - rec_off[:-1] - 8
        d["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pay_cap
        d["pay_cap"] = pay_cap
        d["sig_off"] = np.arange(n_reads, dtype=np.uint64) * sig_cap
        d["sig_cap"] = sig_cap
        big_desc = torch.from_numpy(d.view(np.uint8)).to(dev)
        big_pay = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device=dev)
        big_sig = torch.empty(n_reads * sig_cap + 64, dtype=torch.int16, device=dev)
        big_fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
        a.n_recs = n_reads
        a.desc, a.payload, a.sig_out, a.fields = big_desc.data_ptr(), big_pay.data_ptr(), big_sig.data_ptr(), big_fields.data_ptr()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()), "s5gpu_decode_dev")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts[1:])
        st = big_fields.view(torch.int32).view(n_reads, 16)[:, 0]
        same = bool((st == 0).all().item()) and bool(torch.equal(big_sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n],
                                                                  b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]))
        bulk = {"reads": n_reads, "ms": round(ms, 2), "reads_per_s": round(n_reads / ms * 1e3, 1),
                "raw_signal_GB_per_s": round(n_reads * 2 * n / ms / 1e6, 2), "roundtrip_identical": same}
        ok &= same
    except torch.OutOfMemoryError:
        bulk = None
    wall = time.perf_counter() - t_all
    lat_ms = np.array(lat[2:]) * 1e3
    busy = float(np.sum(lat[2:]))
    line = {"metric": "blow5_get_decode_throughput", "value": round(done * 2 * n / busy / 1e9, 3), "unit": "GB/s",
            "n_gpus": 1, "higher_is_better": True, "dtype": "u8->int16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: random get decode (inflate + svb-zd unpack), %d ids (seed 1) over a %d-read index, batches of %d, %d samples/read" % (len(ids), n_reads, K, n)},
            "reads_per_s": round(done / busy, 1), "batch_latency_ms": {"p50": round(float(np.percentile(lat_ms, 50)), 3), "p99": round(float(np.percentile(lat_ms, 99)), 3)},
            "per_read_latency_us_p50": round(float(np.percentile(lat_ms, 50)) * 1e3 / K, 3),
            "bulk_decode_one_call": bulk,
            "roundtrip_identical": bool(ok), "wall_s_including_verification": round(wall, 2)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
