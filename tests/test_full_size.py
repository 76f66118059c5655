"""BASELINE.json's full size (configs[2]: 1 M reads x 4000 samples, full encode) through size-independent properties:
  - the ordered single-pass stream is well formed: u64 size prefixes chain exactly to the end, offsets strictly increase;
  - decode(encode(x)) == x for EVERY read (the GPU inflate is an independent implementation, itself pinned on zlib-written
    fixtures), all 1 M statuses 0 — this includes the Adler-32 of every record, verified on the device;
  - a 1-in-997 sample of records is inflated by STOCK zlib on the host and equals the oracle's payload byte for byte;
  - a checksum of checksums: the sum of all Adler-32 trailers equals the sum recomputed from the decoded payloads' halves;
  - output size stays below zlib level 6's on the sample (size is tracked next to speed).
Mirrors test/test_view_integrity.sh:62-68 (round trip) at the scale the metric is quoted on.  ~15 s on an MI355X."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import oracle_bind as ob

pytestmark = pytest.mark.gpu

N_READS, N = 1_000_000, 4000


def test_one_million_reads_round_trip_and_sampled_zlib_parity():
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    b = press.DeviceBatch(np.full(N_READS, N, dtype=np.uint64), device=dev)
    b.synth(seed=0x5105, first=0)
    b.encode_stream()
    torch.cuda.synchronize()
    assert b.stream_ok()
    off = b.rec_off.cpu().numpy().astype(np.int64)
    assert off[0] == 0 and (np.diff(off) > 8).all()
    total = int(off[N_READS])
    # framing: every u64 prefix equals the distance to the next record
    stream = b.stream_out[:total]
    prefix_idx = torch.from_numpy(off[:-1]).to(dev)
    pre = torch.zeros(N_READS, dtype=torch.int64, device=dev)
    for k in range(8):
        pre |= stream[prefix_idx + k].to(torch.int64) << (8 * k)
    assert torch.equal(pre, torch.from_numpy(np.diff(off) - 8).to(dev))
    # decode everything on the device
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    sig_cap = (N + 7) // 8 * 8
    payload = torch.empty(N_READS * pay_cap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(N_READS * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(N_READS * 64, dtype=torch.uint8, device=dev)
    d = np.zeros(N_READS, dtype=_lib.REC_DESC)
    d["in_off"] = off[:-1] + 8
    d["in_len"] = np.diff(off) - 8
    d["pay_off"] = np.arange(N_READS, dtype=np.uint64) * pay_cap
    d["pay_cap"] = pay_cap
    d["sig_off"] = np.arange(N_READS, dtype=np.uint64) * sig_cap
    d["sig_cap"] = sig_cap
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method = N_READS, 1, 1
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()), "s5gpu_decode_dev")
    torch.cuda.synchronize()
    f32 = fields.view(torch.int32).view(N_READS, 16)
    assert int(f32[:, 0].abs().sum().item()) == 0, "a record failed to decode (status != 0)"
    assert bool((f32[:, 2] == N).all().item())                                     # n_samples
    got = sig[: N_READS * sig_cap].view(N_READS, sig_cap)[:, :N]
    want = b.sig[: N_READS * sig_cap].view(N_READS, sig_cap)[:, :N]
    assert torch.equal(got, want), "decode(encode(x)) != x"
    # checksum of checksums: big-endian Adler-32 trailers vs the halves recomputed from the decoded payloads
    plen = f32[:, 1].to(torch.int64)                                               # payload_len
    trailer_idx = torch.from_numpy(off[1:] - 4).to(dev)
    adler = torch.zeros(N_READS, dtype=torch.int64, device=dev)
    for k in range(4):
        adler = (adler << 8) | stream[trailer_idx + k].to(torch.int64)
    pay2d = payload[: N_READS * pay_cap].view(N_READS, pay_cap)
    col = torch.arange(pay_cap, device=dev).unsqueeze(0)
    valid = col < plen.unsqueeze(1)
    chunk = 50_000
    sum_a = torch.zeros(N_READS, dtype=torch.int64, device=dev)
    sum_b = torch.zeros(N_READS, dtype=torch.int64, device=dev)
    for lo in range(0, N_READS, chunk):
        x = torch.where(valid[lo:lo + chunk], pay2d[lo:lo + chunk].to(torch.int64), torch.zeros((), dtype=torch.int64, device=dev))
        w = (plen[lo:lo + chunk].unsqueeze(1) - col).clamp(min=0)
        sum_a[lo:lo + chunk] = x.sum(1)
        sum_b[lo:lo + chunk] = (x * w).sum(1)
    a_want = (1 + sum_a) % 65521
    b_want = (plen + sum_b) % 65521
    assert torch.equal(adler, (b_want << 16) | a_want), "Adler-32 trailers do not match the decoded payloads"
    # sampled host parity against stock zlib and the oracle's payload
    idx = list(range(0, N_READS, 997))
    recs = b.stream_records(idx)
    gpu_bytes = ref_bytes = 0
    for i, rec in zip(idx, recs):
        s = ob.synth_read(0x5105, i, N)
        r, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0, s)
        pay = ob.rec_pack(r, ob.SIG_SVB_ZD)
        assert struct.unpack_from("<Q", rec, 0)[0] == len(rec) - 8
        assert zlib.decompress(rec[8:]) == pay
        gpu_bytes += len(rec) - 8
        ref_bytes += len(zlib.compress(pay, 6))
    assert gpu_bytes <= ref_bytes, (gpu_bytes, ref_bytes)


def test_one_million_reads_svbzd_stage_bit_exact_sample_and_full_round_trip():
    """BASELINE configs[1] (svb-zd stage alone) at full size: every blob decodes back to its read; a 1-in-997 sample is
    compared bit for bit with the oracle's encoder; blob lengths satisfy 4 + ceil(N/4) + sum(code+1) on the device."""
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    b = press.DeviceBatch(np.full(N_READS, N, dtype=np.uint64), rec_method=press.REC_NONE, sig_method=press.SIG_SVB_ZD, device=dev,
                          with_stream_out=False)
    b.synth(seed=0x5105, first=0)
    b.svbzd_encode()
    torch.cuda.synchronize()
    lens = b.out_len[:N_READS].cpu().numpy().astype(np.int64)
    slot_off = b.desc_np["out_off"].astype(np.int64)
    assert (lens >= 4 + N // 4 + N).all() and (lens <= 4 + N // 4 + 3 * N).all()
    sig_cap = (N + 7) // 8 * 8
    sig = torch.empty(N_READS * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(N_READS * 64, dtype=torch.uint8, device=dev)
    d = np.zeros(N_READS, dtype=_lib.REC_DESC)
    d["in_off"] = slot_off
    d["in_len"] = lens
    d["sig_off"] = np.arange(N_READS, dtype=np.uint64) * sig_cap
    d["sig_cap"] = sig_cap
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method = N_READS, 0, 1
    a.desc, a.in_, a.sig_out, a.fields = desc.data_ptr(), b.slots.data_ptr(), sig.data_ptr(), fields.data_ptr()
    _lib.check(L.s5gpu_svbzd_decode_dev(C.byref(a), b._stream()), "s5gpu_svbzd_decode_dev")
    torch.cuda.synchronize()
    f32 = fields.view(torch.int32).view(N_READS, 16)
    assert int(f32[:, 0].abs().sum().item()) == 0
    got = sig[: N_READS * sig_cap].view(N_READS, sig_cap)[:, :N]
    want = b.sig[: N_READS * sig_cap].view(N_READS, sig_cap)[:, :N]
    assert torch.equal(got, want)
    for i in range(0, N_READS, 997):
        blob = b.slots[int(slot_off[i]): int(slot_off[i]) + int(lens[i])].cpu().numpy().tobytes()
        assert blob == ob.svbzd_encode(ob.synth_read(0x5105, i, N)), i
    # the one-pass form (blobs straight into the stream) against slots + compaction, all 1 M reads: offsets and a hash of the stream
    del sig, fields
    b.stream_out = torch.empty(int(lens.sum()) + 64, dtype=torch.uint8, device=dev)
    b.compact()
    torch.cuda.synchronize()
    want_off = b.rec_off.clone()
    total = int(want_off[N_READS].item())
    assert total == int(lens.sum())
    want = b.stream_out[:total].clone()
    b.stream_out.zero_()
    b.svbzd_encode_stream()
    torch.cuda.synchronize()
    assert b.stream_ok()
    assert torch.equal(b.rec_off, want_off) and torch.equal(b.stream_out[:total], want)


@pytest.mark.gpu
def test_zstd_full_size_round_trip_checksum():
    """BASELINE configs[2] through the zstd record press: 1 M reads encode -> compact -> decode; every signal comes back (checksum
    of checksums against the synthetic source), every record carries a frame whose declared content size is the payload length"""
    import ctypes as C
    import torch
    from slow5tools_amd import _lib, press
    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    n_reads, n_samp = 1_000_000, 4000
    b = press.DeviceBatch([n_samp] * n_reads, rec_method=press.REC_ZSTD, with_stream_out=True)
    b.synth()
    b.encode()
    b.compact()
    torch.cuda.synchronize()
    off = b.rec_off.cpu().numpy()
    lens = (off[1:] - off[:-1]).astype(np.int64)
    assert lens.min() > 17 and lens.max() < 8 + s5_slot(press, n_samp)
    # frame headers: magic + single-segment descriptor + 4-byte content size
    heads = b.stream_out[torch.from_numpy(off[:-1].astype(np.int64)).to(b.dev)[:, None] + torch.arange(8, 17, device=b.dev)[None, :]].cpu().numpy()
    assert (heads[:, :5] == np.frombuffer(b"\x28\xb5\x2f\xfd\xa0", np.uint8)).all()
    declared = heads[:, 5:9].copy().view('<u4')[:, 0]
    dev = b.dev
    desc = np.zeros(n_reads, dtype=_lib.REC_DESC)
    desc["in_off"] = off[:-1] + 8
    desc["in_len"] = (lens - 8).astype(np.uint32)
    pcap = (b.tot["max_payload"] + 31) // 16 * 16
    stride = (n_samp + 7) // 8 * 8
    desc["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pcap
    desc["pay_cap"] = pcap - 16
    desc["sig_off"] = np.arange(n_reads, dtype=np.uint64) * stride
    desc["sig_cap"] = n_samp
    t_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    pay = torch.empty(n_reads * pcap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(n_reads * stride + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_reads * _lib.REC_FIELDS.itemsize, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method = n_reads, press.REC_ZSTD, press.SIG_SVB_ZD
    a.desc, a.in_, a.payload, a.sig_out, a.fields = t_desc.data_ptr(), b.stream_out.data_ptr(), pay.data_ptr(), sig.data_ptr(), fields.data_ptr()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), None), "s5gpu_decode_dev")
    torch.cuda.synchronize()
    f = fields.cpu().numpy().view(_lib.REC_FIELDS)
    assert (f["status"] == 0).all() and (f["n_samples"] == n_samp).all()
    assert (f["payload_len"] == declared).all()                 # every frame header names its payload's length
    got = sig[: n_reads * stride].view(n_reads, stride)[:, :n_samp]
    want = b.sig[: n_reads * stride].view(n_reads, stride)[:, :n_samp]
    assert torch.equal(got, want)


def s5_slot(press, n_samp):
    from slow5tools_amd import _lib
    return _lib.lib().s5gpu_slot_bound(n_samp, 74, 0, press.REC_ZSTD, press.SIG_SVB_ZD)


def test_long_reads_configs3_shape_round_trip_and_sampled_zlib_parity():
    """BASELINE configs[3]'s read shape (100 000 samples per read) at a chunk of 4096 reads = 0.8 GB of raw signal, through the
    HBM-staged kernels (k_pack + k_deflate_staged, 16 KiB DEFLATE blocks) and the compaction:
      - the record stream is well formed (u64 prefixes chain to the end);
      - decode(encode(x)) == x for every read (GPU inflate + unpack, Adler-32 verified on the device);
      - a 1-in-173 sample is inflated by STOCK zlib and equals the oracle's payload byte for byte;
      - the sampled records are not larger than zlib level 6's."""
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    n_reads, n = 4096, 100_000
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), device=dev)
    b.synth(seed=0x5105, first=12345)
    b.encode()
    b.compact()
    torch.cuda.synchronize()
    assert int(b.ovf[0].item()) == 0                      # the whole batch is staged: nothing goes through the overflow list
    off = b.rec_off.cpu().numpy().astype(np.int64)
    lens = np.diff(off)
    assert off[0] == 0 and (lens > 8).all()
    total = int(off[n_reads])
    stream = b.stream_out[:total]
    prefix_idx = torch.from_numpy(off[:-1]).to(dev)
    pre = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    for k in range(8):
        pre |= stream[prefix_idx + k].to(torch.int64) << (8 * k)
    assert torch.equal(pre, torch.from_numpy(lens - 8).to(dev))
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    sig_cap = (n + 7) // 8 * 8
    payload = torch.empty(n_reads * pay_cap + 64, dtype=torch.uint8, device=dev)
    sig = torch.empty(n_reads * sig_cap + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
    d = np.zeros(n_reads, dtype=_lib.REC_DESC)
    d["in_off"] = off[:-1] + 8
    d["in_len"] = lens - 8
    d["pay_off"] = np.arange(n_reads, dtype=np.uint64) * pay_cap
    d["pay_cap"] = pay_cap
    d["sig_off"] = np.arange(n_reads, dtype=np.uint64) * sig_cap
    d["sig_cap"] = sig_cap
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method = n_reads, 1, 1
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()), "s5gpu_decode_dev")
    torch.cuda.synchronize()
    f32 = fields.view(torch.int32).view(n_reads, 16)
    assert int(f32[:, 0].abs().sum().item()) == 0, "a record failed to decode (status != 0)"
    assert bool((f32[:, 2] == n).all().item())
    got = sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]
    want = b.sig[: n_reads * sig_cap].view(n_reads, sig_cap)[:, :n]
    assert torch.equal(got, want), "decode(encode(x)) != x"
    idx = list(range(0, n_reads, 173)) + [n_reads - 1]
    gpu_bytes = ref_bytes = 0
    for i, rec in zip(idx, b.stream_records(idx)):
        s = ob.synth_read(0x5105, 12345 + i, n)
        r, keep = ob.make_rec(ob.synth_read_id(12345 + i), 0, 8192.0, 23.0, 1467.61, 4000.0, s)
        pay = ob.rec_pack(r, ob.SIG_SVB_ZD)
        assert struct.unpack_from("<Q", rec, 0)[0] == len(rec) - 8
        assert zlib.decompress(rec[8:]) == pay
        gpu_bytes += len(rec) - 8
        ref_bytes += len(zlib.compress(pay, 6))
    assert gpu_bytes <= ref_bytes, (gpu_bytes, ref_bytes)
