"""Soak of the S5GPU_DEC_NO_PAYLOAD bulk decode (round-4 review item 1): the call bench.py's configs4 leg times, thousands of times, every sample
of every call compared with the source, signals and fields cleared between calls.

    python tools/np_tripwire.py [calls] [reads] [samples] [scratch: default|three]            (product build)
    S5GPU_LIB=slow5tools_amd/_variants/libs5_trip.so python tools/np_tripwire.py ...          (tools/variant.sh trip -DS5_NP_TRIPWIRE)

With the tripwire variant the decoding wave itself re-checksums its scratch slot (an independent byte-wise Adler-32, through L1 and around
it) right before the unpack and right after, and a single lane re-decodes the svb-zd blob and compares it with the samples the unpack stored:
  status 20  the slot differs from what the inflate verified (both views)          -> the slot changed between inflate and unpack
  status 23  only the view through this CU's L1 differs                            -> stale L1 lines of the slot's previous record
  status 24  only the view around L1 differs
  status 21  the slot changed while the unpack was reading it
  status 22  the slot is intact and the stored samples differ from it              -> the unpack read or computed wrong
One line per run: calls, records, wrong calls, statuses seen; exit code 1 if anything was wrong."""
import collections
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from slow5tools_amd import _lib, press

L = _lib.lib()
_lib.check(L.s5gpu_init(0), "init")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
scratch = sys.argv[4] if len(sys.argv) > 4 else "default"
dev = "cuda:0"
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64), device=dev)
b.synth()
b.encode_stream()
torch.cuda.synchronize()
assert b.stream_ok()
off = b.rec_off.cpu().numpy().astype(np.int64)
pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16)
stride = (n + 7) // 8 * 8
d = np.zeros(n_reads, dtype=_lib.REC_DESC)
d["in_off"] = off[:-1] + 8
d["in_len"] = np.diff(off) - 8
d["sig_off"] = np.arange(n_reads, dtype=np.uint64) * stride
d["sig_cap"] = n
desc = torch.from_numpy(d.view(np.uint8).copy()).to(dev)
sig = torch.empty(n_reads * stride + 64, dtype=torch.int16, device=dev)
fields = torch.zeros(n_reads * 64, dtype=torch.uint8, device=dev)
L.s5gpu_decode_scratch_bytes.restype = C.c_uint64
L.s5gpu_decode_scratch_bytes.argtypes = [C.c_uint32]
sb = 64 + 3 * (pay_cap + 32) if scratch == "three" else int(L.s5gpu_decode_scratch_bytes(pay_cap))
scr = torch.empty(sb, dtype=torch.uint8, device=dev)
a = _lib.DecodeArgs()
a.n_recs, a.rec_method, a.sig_method, a.flags = n_reads, 1, 1, _lib.DEC_NO_PAYLOAD
a.desc, a.in_, a.payload, a.sig_out, a.fields = desc.data_ptr(), b.stream_out.data_ptr(), scr.data_ptr(), sig.data_ptr(), fields.data_ptr()
a.payload_bytes, a.max_pay_cap = sb, pay_cap
want = b.sig[: n_reads * stride].view(n_reads, stride)[:, :n]
st = b._stream()
seen = collections.Counter()
wrong_calls = 0
t0 = time.time()
for call in range(calls):
    sig.zero_()
    fields.zero_()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), st), "decode")
    torch.cuda.synchronize()
    s32 = fields.view(torch.int32).view(n_reads, 16)[:, 0]
    nbad = int((s32 != 0).sum().item())
    same = torch.equal(sig[: n_reads * stride].view(n_reads, stride)[:, :n], want)
    if nbad or not same:
        wrong_calls += 1
        codes = collections.Counter(s32[s32 != 0].cpu().tolist())
        seen.update(codes)
        neq = torch.nonzero((sig[: n_reads * stride].view(n_reads, stride)[:, :n] != want).any(dim=1)).flatten()
        print("  call %d: statuses %s; %d reads with other samples than the source (first %s)" % (call, dict(codes), neq.numel(), neq[:8].tolist()), flush=True)
dt = time.time() - t0
print("np_tripwire lib=%s scratch=%s: %d calls x %d records x %d samples in %.1f s: %d wrong calls, statuses %s" % (
    os.path.basename(os.environ.get("S5GPU_LIB", "libslow5gpu.so")), scratch, calls, n_reads, n, dt, wrong_calls, dict(seen)))
sys.exit(1 if wrong_calls else 0)
