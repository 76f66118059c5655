import sys, time, ctypes as C
"""Decode latency vs batch size (k_inflate + k_unpack), one line per batch."""
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0))
n_reads, n = 200000, 4000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode(); b.compact(); torch.cuda.synchronize()
rec_off = b.rec_off.cpu().numpy().astype(np.int64)
K = 4096
pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16); sig_cap = (n + 7) // 8 * 8
payload = torch.empty(K * pay_cap + 64, dtype=torch.uint8, device="cuda:0"); sig = torch.empty(K * sig_cap + 64, dtype=torch.int16, device="cuda:0")
fields = torch.zeros(K * 64, dtype=torch.uint8, device="cuda:0"); desc_dev = torch.empty(K * _lib.REC_DESC.itemsize, dtype=torch.uint8, device="cuda:0")
a = _lib.DecodeArgs(); a.rec_method, a.sig_method = 1, 1
a.desc, a.in_, a.payload, a.sig_out, a.fields = desc_dev.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
rng = np.random.default_rng(1)
for k in (4096, 4096, 4096, 1696, 1696, 4096, 1000, 1000, 100, 100, 4095, 4095, 4096):
    sel = rng.integers(0, n_reads, k)
    d = np.zeros(k, dtype=_lib.REC_DESC)
    d["in_off"] = rec_off[sel] + 8; d["in_len"] = rec_off[sel + 1] - rec_off[sel] - 8
    d["pay_off"] = np.arange(k, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
    d["sig_off"] = np.arange(k, dtype=np.uint64) * sig_cap; d["sig_cap"] = sig_cap
    desc_dev[: d.nbytes].copy_(torch.from_numpy(d.view(np.uint8)))
    a.n_recs = k
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream())); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("k=%d launch %.3f ms total %.3f ms" % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
