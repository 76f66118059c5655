"""S5GPU_DEC_NO_PAYLOAD at the sizes bench.py times it on (`configs4`: tickets, launch-order list, reused scratch slots), call after call.

The decode form `get` uses (/root/reference/src/get.c:37-66: every fetched read must equal the source, test/test_get.sh) keeps a record's
uncompressed bytes in a scratch slot that the same workgroup reuses for its next record.  Round 3 saw ONE session in which that bulk form
returned wrong samples with status 0 and never again; these tests are the tripwire that stays in the suite:
  - 1 M records x 4000 samples (BASELINE configs[4]'s index), default scratch, >= 20 back-to-back calls, the signal buffer and the fields
    cleared between calls, EVERY sample of every call compared with the generator's reads;
  - the same through a 3-slot scratch (two workgroups take all the tickets: every record lands in a reused slot) on a smaller batch;
  - 262 144 records with the read lengths of a real run through the launch-order list (longest first);
  - zstd records and ex-zd signals through the same mode.
`tools/np_tripwire.py` runs the same loops on a variant build whose kernel re-checksums the slot before and after the unpack."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np_args(_lib, L, torch, n_rec, in_ptr, off, sig_caps, rec_method, sig_method, pay_cap, scratch_bytes, dev):
    so = np.concatenate([[0], np.cumsum((np.asarray(sig_caps, dtype=np.int64) + 7) // 8 * 8)]).astype(np.int64)
    d = np.zeros(n_rec, dtype=_lib.REC_DESC)
    d["in_off"] = off[:-1] + 8
    d["in_len"] = np.diff(off) - 8
    d["sig_off"] = so[:-1]
    d["sig_cap"] = sig_caps
    desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    sig = torch.empty(int(so[-1]) + 64, dtype=torch.int16, device=dev)
    fields = torch.zeros(n_rec * 64, dtype=torch.uint8, device=dev)
    if scratch_bytes is None:
        L.s5gpu_decode_scratch_bytes.restype = C.c_uint64
        L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
        scratch_bytes = int(L.s5gpu_decode_scratch_bytes(pay_cap))
    scr = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method, a.flags = n_rec, rec_method, sig_method, _lib.DEC_NO_PAYLOAD
    a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), in_ptr, scr.data_ptr(), sig.data_ptr(), fields.data_ptr()
    a.payload_bytes, a.max_pay_cap = scratch_bytes, pay_cap
    return a, so, (desc, sig, fields, scr)


def _soak(L, _lib, torch, a, keep, n_rec, calls, check, stream):
    desc, sig, fields, scr = keep
    for call in range(calls):
        sig.zero_()
        fields.zero_()
        _lib.check(L.s5gpu_decode_dev(C.byref(a), stream), "s5gpu_decode_dev")
        torch.cuda.synchronize()
        st = fields.view(torch.int32).view(n_rec, 16)[:, 0]
        bad = int((st != 0).sum().item())
        assert bad == 0, "call %d: %d records with a status (first %s)" % (call, bad, st[st != 0][:8].tolist())
        check(call, sig, fields)


def _uniform_batch(press, torch, n_reads, n, rec_method, sig_method, dev):
    b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), rec_method=rec_method, sig_method=sig_method, device=dev)
    b.synth(seed=0x5105, first=0)
    if rec_method == press.REC_ZLIB:
        b.encode_stream()
        torch.cuda.synchronize()
        assert b.stream_ok()
    else:
        b.encode()
        b.compact()
        torch.cuda.synchronize()
    return b, b.rec_off.cpu().numpy().astype(np.int64)


def _uniform_check(torch, b, n_reads, n):
    stride = (n + 7) // 8 * 8
    want = b.sig[: n_reads * stride].view(n_reads, stride)[:, :n]

    def check(call, sig, fields):
        got = sig[: n_reads * stride].view(n_reads, stride)[:, :n]
        if not torch.equal(got, want):
            neq = torch.nonzero((got != want).any(dim=1)).flatten()
            raise AssertionError("call %d: %d reads decode to other samples than were encoded (status 0), first %s" % (call, neq.numel(), neq[:8].tolist()))
        ns = fields.view(torch.int32).view(n_reads, 16)[:, 2]
        assert bool((ns == n).all().item())
    return check


@pytest.mark.parametrize("form", ["zlib+svb-zd", "zlib+ex-zd", "zstd+svb-zd"])
def test_one_million_records_no_payload_twenty_calls(form):
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    rec_method = press.REC_ZSTD if form.startswith("zstd") else press.REC_ZLIB
    sig_method = press.SIG_EX_ZD if form.endswith("ex-zd") else press.SIG_SVB_ZD
    n_reads, n = (1_000_000, 4000) if form == "zlib+svb-zd" else (262_144, 4000)
    calls = 20 if form == "zlib+svb-zd" else 8
    b, off = _uniform_batch(press, torch, n_reads, n, rec_method, sig_method, dev)
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    a, so, keep = _np_args(_lib, L, torch, n_reads, b.stream_out.data_ptr(), off, np.full(n_reads, n), rec_method, sig_method, pay_cap, None, dev)
    _soak(L, _lib, torch, a, keep, n_reads, calls, _uniform_check(torch, b, n_reads, n), b._stream())


def test_every_record_through_a_reused_slot_three_slot_scratch():
    """two workgroups draw every ticket: each record is inflated into a slot that held another record a moment ago"""
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    n_reads, n = 8192, 4000
    b, off = _uniform_batch(press, torch, n_reads, n, press.REC_ZLIB, press.SIG_SVB_ZD, dev)
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    a, so, keep = _np_args(_lib, L, torch, n_reads, b.stream_out.data_ptr(), off, np.full(n_reads, n), press.REC_ZLIB, press.SIG_SVB_ZD, pay_cap, 64 + 3 * (pay_cap + 32), dev)
    _soak(L, _lib, torch, a, keep, n_reads, 20, _uniform_check(torch, b, n_reads, n), b._stream())


def test_real_run_read_lengths_through_the_launch_order_list():
    """262 144 records, log-normal lengths (median 6000, longest > 300 k samples): the batch is counting-sorted by compressed length on the
    device and the tickets follow that list; records of every length class share the reused slots"""
    import torch
    from slow5tools_amd import _lib, press

    L = _lib.lib()
    _lib.check(L.s5gpu_init(0), "s5gpu_init")
    dev = "cuda:0"
    n_reads = 262_144
    rng = np.random.default_rng(5)
    ns = np.clip(np.exp(rng.normal(np.log(6000), 0.9, n_reads)), 200, 400000).astype(np.uint64)
    b = press.DeviceBatch(ns, device=dev, lds_payload_cap=8192)
    tot = b.sig.numel()
    _lib.check(L.s5gpu_synth_dev(b.sig.data_ptr(), 1, tot - 64, tot, 0x5105, 0, b._stream()), "synth")      # one long trace cut into the reads
    _lib.check(L.s5gpu_synth_hdr_dev(b.hdr.data_ptr(), n_reads, 0, b._stream()), "hdr")
    b.encode()
    b.compact()
    torch.cuda.synchronize()
    off = b.rec_off.cpu().numpy().astype(np.int64)
    pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
    caps = ns.astype(np.int64)
    a, so, keep = _np_args(_lib, L, torch, n_reads, b.stream_out.data_ptr(), off, caps, press.REC_ZLIB, press.SIG_SVB_ZD, pay_cap, None, dev)
    # the source samples sit at the batch's own (8-sample aligned) offsets, the decoded ones at so[]: the same layout
    assert np.array_equal(so[:-1], b.desc_np["sig_off"].astype(np.int64))
    total = int(so[-1])
    # the padding between reads (up to 7 samples) is not written by the decoder: mask = +1 at every read's start, -1 behind its last sample, summed
    m = torch.zeros(total + 1, dtype=torch.int8, device=dev)
    starts = torch.from_numpy(so[:-1]).to(dev)
    m.index_add_(0, starts, torch.ones(n_reads, dtype=torch.int8, device=dev))
    m.index_add_(0, starts + torch.from_numpy(ns.astype(np.int64)).to(dev), torch.full((n_reads,), -1, dtype=torch.int8, device=dev))
    mask = torch.cumsum(m, 0, dtype=torch.int8)[:total] > 0
    del m
    want = torch.where(mask, b.sig[:total], torch.zeros((), dtype=torch.int16, device=dev))
    nst = torch.from_numpy(ns.astype(np.int32)).to(dev)

    def check(call, sig, fields):
        got = sig[:total]
        if not torch.equal(got, want):
            pos = torch.nonzero(got != want).flatten()[:4].cpu().numpy()
            reads = np.searchsorted(so, pos, side="right") - 1
            raise AssertionError("call %d: samples differ (status 0), first in reads %s (lengths %s)" % (call, reads.tolist(), ns[reads].tolist()))
        assert torch.equal(fields.view(torch.int32).view(n_reads, 16)[:, 2], nst)

    _soak(L, _lib, torch, a, keep, n_reads, 6, check, b._stream())
