"""Host-side mirror of the press path on top of the C ABI.

Two layers, matching include/slow5gpu.h:
  * encode_records / decode_records : host buffers (numpy / bytes) through s5gpu_encode_batch /
    s5gpu_decode_batch — the batch that stands where slow5tools' work_db() stands
    (/root/reference/src/view.c:292, src/merge.c:440, src/get.c:364).
  * DeviceBatch : device-resident buffers (torch tensors used only as HBM allocations + stream
    handles) through s5gpu_encode_dev / s5gpu_compact_dev — what bench.py times.
"""
import ctypes as C
import struct

import numpy as np

from . import _lib
from ._lib import REC_NONE, REC_ZLIB, REC_ZSTD, SIG_EX_ZD, SIG_NONE, SIG_SVB_ZD, S5GpuError, check  # noqa: F401


def pack_hdr(read_id, read_group, digitisation, offset, rng, sampling_rate):
    """The record bytes that precede the u64 length field."""
    rid = read_id if isinstance(read_id, (bytes, bytearray)) else read_id.encode()
    return struct.pack("<H", len(rid)) + bytes(rid) + struct.pack("<Idddd", read_group, digitisation, offset, rng,
                                                                   sampling_rate)


def encode_records(signals, hdrs, auxs=None, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD):
    """slow5_rec_to_mem for a batch: returns list of bytes, each [u64 size][record]."""
    L = _lib.lib()
    n = len(signals)
    if n == 0:
        return []
    sigs = [np.ascontiguousarray(s, dtype=np.int16) for s in signals]
    hb = [bytes(h) for h in hdrs]
    ab = [bytes(a) for a in auxs] if auxs is not None else None
    vp = C.c_void_p
    sig_p = (vp * n)(*[s.ctypes.data if s.size else None for s in sigs])
    ns = (C.c_uint64 * n)(*[s.size for s in sigs])
    hbuf = [C.create_string_buffer(h, len(h)) for h in hb]
    hdr_p = (vp * n)(*[C.addressof(b) for b in hbuf])
    hl = (C.c_uint32 * n)(*[len(h) for h in hb])
    if ab is not None:
        abuf = [C.create_string_buffer(a, max(len(a), 1)) for a in ab]
        aux_p = (vp * n)(*[C.addressof(b) for b in abuf])
        al = (C.c_uint32 * n)(*[len(a) for a in ab])
    else:
        aux_p, al = None, None
    out = (vp * n)()
    out_len = (C.c_size_t * n)()
    check(L.s5gpu_encode_batch(n, sig_p, ns, hdr_p, hl, aux_p, al, rec_method, sig_method, out, out_len),
          "s5gpu_encode_batch")
    libc = C.CDLL(None)
    libc.free.argtypes = [vp]
    res = []
    for i in range(n):
        res.append(C.string_at(out[i], out_len[i]))
        libc.free(out[i])
    return res


def decode_records(records, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD, raise_on_error=True):
    """slow5_rec_depress_parse for a batch of records (bytes without the u64 prefix).
    Returns list of dicts; a corrupt record has only {'status': code}."""
    L = _lib.lib()
    n = len(records)
    if n == 0:
        return []
    vp = C.c_void_p
    rb = [bytes(r) for r in records]
    rbuf = [C.create_string_buffer(r, max(len(r), 1)) for r in rb]
    rec_p = (vp * n)(*[C.addressof(b) for b in rbuf])
    rl = (C.c_size_t * n)(*[len(r) for r in rb])
    pay = (vp * n)()
    sig = (vp * n)()
    fields = np.zeros(n, dtype=_lib.REC_FIELDS)
    rc = L.s5gpu_decode_batch(n, rec_p, rl, rec_method, sig_method, pay, sig, fields.ctypes.data_as(vp))
    if rc != 0 and (raise_on_error or rc != -5):
        _free_all(pay, sig, n)
        check(rc, "s5gpu_decode_batch")
    libc = C.CDLL(None)
    libc.free.argtypes = [vp]
    out = []
    for i in range(n):
        f = fields[i]
        if f["status"] != 0:
            out.append(dict(status=int(f["status"])))
            continue
        payload = C.string_at(pay[i], int(f["payload_len"]))
        s = np.frombuffer(C.string_at(sig[i], 2 * int(f["n_samples"])), dtype=np.int16).copy()
        idl = int(f["read_id_len"])
        out.append(dict(status=0, read_id=payload[2:2 + idl], read_group=int(f["read_group"]),
                        digitisation=float(f["digitisation"]), offset=float(f["offset"]), range=float(f["range"]),
                        sampling_rate=float(f["sampling_rate"]), signal=s,
                        aux=payload[int(f["aux_off"]):int(f["aux_off"]) + int(f["aux_len"])], payload=payload))
        libc.free(pay[i])
        libc.free(sig[i])
    return out


def decode_signals_dev(records, rec_method=REC_ZLIB, max_pay_cap=None, sig_caps=None, scratch_bytes=None, device="cuda:0", sig_method=SIG_SVB_ZD,
                       max_in_len=None):
    """s5gpu_decode_dev with S5GPU_DEC_NO_PAYLOAD: fields + signals only, the uncompressed records stay in reused scratch slots
    (what `get` needs of /root/reference/src/get.c:37-66 when the caller holds the read ids).  records: bytes without the u64 prefix.
    Returns (fields as a numpy REC_FIELDS array, list of int16 arrays — empty where status != 0)."""
    import torch

    L = _lib.lib()
    n = len(records)
    lens = np.array([len(r) for r in records], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum((lens + 15) // 16 * 16)]).astype(np.int64)
    blob = np.zeros(int(offs[-1]) + 64, dtype=np.uint8)
    for r, o in zip(records, offs[:-1]):
        blob[o:o + len(r)] = np.frombuffer(bytes(r), dtype=np.uint8)
    if max_pay_cap is None:
        max_pay_cap = int(8 * lens.max() + 4096) if n else 64
    if sig_caps is None:
        sig_caps = np.full(n, max_pay_cap, dtype=np.int64)
    sig_caps = np.asarray(sig_caps, dtype=np.int64)
    so = np.concatenate([[0], np.cumsum((sig_caps + 15) // 8 * 8)]).astype(np.int64)
    d = np.zeros(n, dtype=_lib.REC_DESC)
    d["in_off"], d["in_len"], d["sig_off"], d["sig_cap"] = offs[:-1], lens, so[:-1], sig_caps
    dev = torch.device(device)
    t_in = torch.from_numpy(blob).to(dev)
    t_desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
    t_sig = torch.zeros(int(so[-1]) + 64, dtype=torch.int16, device=dev)
    t_fields = torch.zeros(max(n, 1) * 64, dtype=torch.uint8, device=dev)
    if scratch_bytes is None:
        L.s5gpu_decode_scratch_bytes.restype = C.c_uint64
        L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
        scratch_bytes = int(L.s5gpu_decode_scratch_bytes(max_pay_cap))
    t_scr = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
    a = _lib.DecodeArgs()
    a.n_recs, a.rec_method, a.sig_method, a.flags = n, rec_method, sig_method, _lib.DEC_NO_PAYLOAD
    a.desc, a.in_, a.payload, a.sig_out, a.fields = t_desc.data_ptr(), t_in.data_ptr(), t_scr.data_ptr(), t_sig.data_ptr(), t_fields.data_ptr()
    a.payload_bytes, a.max_pay_cap = scratch_bytes, max_pay_cap
    a.max_in_len = int(lens.max()) if (max_in_len is None and n) else int(max_in_len or 0)   # (a hint: short records stay in LDS, include/slow5gpu.h)
    check(L.s5gpu_decode_dev(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "s5gpu_decode_dev")
    torch.cuda.synchronize(dev)
    f = t_fields.cpu().numpy().view(_lib.REC_FIELDS)[:n].copy()
    sig = t_sig.cpu().numpy()
    out = [sig[int(so[i]):int(so[i]) + int(f["n_samples"][i])].copy() if f["status"][i] == 0 else np.zeros(0, np.int16) for i in range(n)]
    return f, out


def _free_all(pay, sig, n):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for i in range(n):
        if pay[i]:
            libc.free(pay[i])
        if sig[i]:
            libc.free(sig[i])


def _up(x, a):
    return (x + a - 1) // a * a


def make_read_desc(n_samples, hdr_len, aux_len, rec_method, sig_method):
    """numpy READ_DESC array + (total samples, hdr bytes, aux bytes, slot bytes, max payload)."""
    L = _lib.lib()
    n_samples = np.asarray(n_samples, dtype=np.uint64)
    n = len(n_samples)
    hdr_len = np.broadcast_to(np.asarray(hdr_len, dtype=np.uint64), (n,))
    aux_len = np.broadcast_to(np.asarray(aux_len, dtype=np.uint64), (n,))
    d = np.zeros(n, dtype=_lib.READ_DESC)
    sig_pad = (n_samples + 7) // 8 * 8
    if sig_method == SIG_SVB_ZD:
        sigb = 4 + (n_samples + 3) // 4 + 3 * n_samples
    elif sig_method == SIG_EX_ZD:
        sigb = 24 + 2 * ((n_samples + 3) // 4 + 4 * n_samples) + n_samples
    else:
        sigb = 2 * n_samples
    pay = hdr_len + 8 + sigb + aux_len
    if rec_method == REC_ZLIB:
        slot = pay + 6 * (pay // 16384 + 1) + 14
    elif rec_method == REC_ZSTD:
        slot = pay + 4 * (pay // 16384 + 1) + 24
    else:
        slot = pay + 8
    slot = (slot + 16 + 15) // 16 * 16
    # cross-check the vectorised bound against the library for the first read
    if n:
        assert int(slot[0]) == L.s5gpu_slot_bound(int(n_samples[0]), int(hdr_len[0]), int(aux_len[0]), rec_method, sig_method)
    d["n_samples"] = n_samples
    d["hdr_len"] = hdr_len
    d["aux_len"] = aux_len
    d["slot_cap"] = slot
    d["sig_off"] = np.concatenate([[0], np.cumsum(sig_pad)[:-1]]) if n else []
    d["hdr_off"] = np.concatenate([[0], np.cumsum(hdr_len)[:-1]]) if n else []
    d["aux_off"] = np.concatenate([[0], np.cumsum(aux_len)[:-1]]) if n else []
    d["out_off"] = np.concatenate([[0], np.cumsum(slot)[:-1]]) if n else []
    totals = dict(samples=int(sig_pad.sum()), hdr=int(hdr_len.sum()), aux=int(aux_len.sum()), slots=int(slot.sum()),
                  max_payload=int(pay.max()) if n else 0)
    return d, totals


class DeviceBatch:
    """Device-resident encode batch.  torch is plumbing only: HBM allocations and the stream handle."""

    def __init__(self, n_samples, hdr_len=74, aux_len=0, rec_method=REC_ZLIB, sig_method=SIG_SVB_ZD, device="cuda:0",
                 with_stream_out=True, lds_payload_cap=0, share=None):
        """share: another DeviceBatch whose slots / stream_out (at least as large) this one reuses — chunks of a long job
        that run one after the other need one set of output buffers, not one per chunk"""
        import torch

        self.torch = torch
        self.dev = torch.device(device)
        self.rec_method, self.sig_method = rec_method, sig_method
        self.desc_np, self.tot = make_read_desc(n_samples, hdr_len, aux_len, rec_method, sig_method)
        self.n = len(self.desc_np)
        u8 = torch.uint8
        self.desc = torch.from_numpy(self.desc_np.view(np.uint8).copy()).to(self.dev)
        self.sig = torch.zeros(self.tot["samples"] + 64, dtype=torch.int16, device=self.dev)
        self.hdr = torch.zeros(self.tot["hdr"] + 64, dtype=u8, device=self.dev)
        self.aux = torch.zeros(self.tot["aux"] + 64, dtype=u8, device=self.dev)
        if share is not None:
            assert share.slots.numel() >= self.tot["slots"] + 64
            self.slots = share.slots
        else:
            self.slots = torch.empty(self.tot["slots"] + 64, dtype=u8, device=self.dev)
        self.out_len = torch.zeros(self.n + 1, dtype=torch.int32, device=self.dev)
        self.ovf = torch.zeros(self.n + 4, dtype=torch.int32, device=self.dev)
        self.rec_off = torch.zeros(self.n + 1, dtype=torch.int64, device=self.dev)
        self.tmp = torch.zeros(self.n // 1024 + 8, dtype=torch.int64, device=self.dev)
        if share is not None and share.stream_out is not None:
            self.stream_out = share.stream_out
        else:
            self.stream_out = torch.empty(self.tot["slots"] + 64, dtype=u8, device=self.dev) if with_stream_out else None
        a = _lib.EncodeArgs()
        a.n_reads, a.rec_method, a.sig_method = self.n, rec_method, sig_method
        a.desc, a.sig, a.hdr, a.aux = self.desc.data_ptr(), self.sig.data_ptr(), self.hdr.data_ptr(), self.aux.data_ptr()
        a.slots, a.out_len = self.slots.data_ptr(), self.out_len.data_ptr()
        a.max_payload = self.tot["max_payload"]
        a.lds_payload_cap = lds_payload_cap
        a.ovf = self.ovf.data_ptr()
        self.args = a

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def upload(self, signals, hdrs, auxs=None):
        sig = np.zeros(self.tot["samples"], dtype=np.int16)
        hdr = np.zeros(self.tot["hdr"], dtype=np.uint8)
        aux = np.zeros(self.tot["aux"], dtype=np.uint8)
        for i, d in enumerate(self.desc_np):
            s = np.asarray(signals[i], dtype=np.int16)
            sig[int(d["sig_off"]):int(d["sig_off"]) + s.size] = s
            hdr[int(d["hdr_off"]):int(d["hdr_off"]) + int(d["hdr_len"])] = np.frombuffer(hdrs[i], dtype=np.uint8)
            if auxs is not None and d["aux_len"]:
                aux[int(d["aux_off"]):int(d["aux_off"]) + int(d["aux_len"])] = np.frombuffer(auxs[i], dtype=np.uint8)
        t = self.torch
        self.sig[: sig.size].copy_(t.from_numpy(sig))
        self.hdr[: hdr.size].copy_(t.from_numpy(hdr))
        if aux.size:
            self.aux[: aux.size].copy_(t.from_numpy(aux))

    def synth(self, seed=0x5105, first=0):
        """uniform-length batches only: fills sig + hdr on device (bit-identical to oracle/synth.c)"""
        n = int(self.desc_np["n_samples"][0])
        assert (self.desc_np["n_samples"] == n).all() and (self.desc_np["hdr_len"] == 74).all()
        stride = (n + 7) // 8 * 8
        L = _lib.lib()
        check(L.s5gpu_synth_dev(self.sig.data_ptr(), self.n, n, stride, seed, first, self._stream()), "s5gpu_synth_dev")
        check(L.s5gpu_synth_hdr_dev(self.hdr.data_ptr(), self.n, first, self._stream()), "s5gpu_synth_hdr_dev")

    def encode(self):
        check(_lib.lib().s5gpu_encode_dev(C.byref(self.args), self._stream()), "s5gpu_encode_dev")

    def pack_parked(self):
        """step 1 of the staged (long-read) encode alone: payloads parked in the slots' tails"""
        check(_lib.lib().s5gpu_pack_parked_dev(C.byref(self.args), self._stream()), "s5gpu_pack_parked_dev")

    def deflate_parked(self):
        """step 2: the parked payloads compressed in place (pack_parked + deflate_parked = encode on long reads)"""
        check(_lib.lib().s5gpu_deflate_parked_dev(C.byref(self.args), self._stream()), "s5gpu_deflate_parked_dev")

    def encode_stream(self):
        """ordered single-pass encode straight into stream_out / rec_off (no slots, no compaction pass)"""
        if not hasattr(self, "lb_state"):
            t = self.torch
            self.lb_state = t.zeros(self.n + 1, dtype=t.int64, device=self.dev)
            self.lb_ctl = t.zeros(4, dtype=t.int32, device=self.dev)
        check(_lib.lib().s5gpu_encode_stream_dev(C.byref(self.args), self.stream_out.data_ptr(), self.rec_off.data_ptr(),
                                                 self.lb_state.data_ptr(), self.lb_ctl.data_ptr(), self._stream()),
              "s5gpu_encode_stream_dev")

    def svbzd_encode_stream(self):
        """svb-zd blobs straight into stream_out / rec_off in one pass (svbzd_encode + compact)"""
        if not hasattr(self, "lb_state"):
            t = self.torch
            self.lb_state = t.zeros(self.n + 1, dtype=t.int64, device=self.dev)
            self.lb_ctl = t.zeros(4, dtype=t.int32, device=self.dev)
        check(_lib.lib().s5gpu_svbzd_encode_stream_dev(C.byref(self.args), self.stream_out.data_ptr(), self.rec_off.data_ptr(),
                                                       self.lb_state.data_ptr(), self.lb_ctl.data_ptr(), self._stream()),
              "s5gpu_svbzd_encode_stream_dev")

    def stream_ok(self):
        """after a synchronise: True if the single-pass stream is valid (no LDS overflow, no look-back timeout)"""
        c = self.lb_ctl.cpu().numpy()
        return int(c[0]) == 0 and int(c[2]) == 0

    def svbzd_encode(self):
        check(_lib.lib().s5gpu_svbzd_encode_dev(C.byref(self.args), self._stream()), "s5gpu_svbzd_encode_dev")

    def compact(self):
        check(_lib.lib().s5gpu_compact_dev(self.n, self.desc.data_ptr(), self.slots.data_ptr(), self.out_len.data_ptr(),
                                           self.rec_off.data_ptr(), self.stream_out.data_ptr(), self.tmp.data_ptr(),
                                           self._stream()), "s5gpu_compact_dev")

    def records(self, idx=None):
        """download record slots as list of bytes"""
        self.torch.cuda.synchronize(self.dev)
        lens = self.out_len[: self.n].cpu().numpy()
        idx = range(self.n) if idx is None else idx
        out = []
        for i in idx:
            o = int(self.desc_np["out_off"][i])
            out.append(self.slots[o:o + int(lens[i])].cpu().numpy().tobytes())
        return out

    def stream_records(self, idx):
        """records idx out of the contiguous stream (after compact() or encode_stream())"""
        self.torch.cuda.synchronize(self.dev)
        off = self.rec_off.cpu().numpy()
        return [self.stream_out[int(off[i]):int(off[i + 1])].cpu().numpy().tobytes() for i in idx]

    def stream_bytes(self):
        self.torch.cuda.synchronize(self.dev)
        off = self.rec_off.cpu().numpy()
        return self.stream_out[: int(off[self.n])].cpu().numpy().tobytes(), off
