import sys, time, os, ctypes as C
"""Per-batch host/GPU time breakdown of get-style decode batches (used to find the 90 ms stall of torch pinned non_blocking copies)."""
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0))
n_reads, n = 1000000, 4000
b = press.DeviceBatch(np.full(n_reads, n, dtype=np.uint64)); b.synth(); b.encode(); b.compact(); torch.cuda.synchronize()
rec_off = b.rec_off.cpu().numpy().astype(np.int64)
K = 4096
pay_cap = 16 * ((int(b.tot["max_payload"]) + 31) // 16); sig_cap = (n + 7) // 8 * 8
payload = torch.empty(K * pay_cap + 64, dtype=torch.uint8, device="cuda:0"); sig = torch.empty(K * sig_cap + 64, dtype=torch.int16, device="cuda:0")
fields = torch.zeros(K * 64, dtype=torch.uint8, device="cuda:0"); desc_dev = torch.empty(K * _lib.REC_DESC.itemsize, dtype=torch.uint8, device="cuda:0")
desc_pin = torch.empty(K * _lib.REC_DESC.itemsize, dtype=torch.uint8).pin_memory()
a = _lib.DecodeArgs(); a.rec_method, a.sig_method = 1, 1
a.desc, a.in_, a.payload, a.sig_out, a.fields = desc_dev.data_ptr(), b.stream_out.data_ptr(), payload.data_ptr(), sig.data_ptr(), fields.data_ptr()
rng = np.random.default_rng(1)
ids = rng.integers(0, n_reads, 100000)
for lo in range(0, len(ids), K):
    sel = ids[lo:lo + K]; k = len(sel)
    t0 = time.perf_counter()
    d = np.zeros(k, dtype=_lib.REC_DESC)
    d["in_off"] = rec_off[sel] + 8; d["in_len"] = rec_off[sel + 1] - rec_off[sel] - 8
    d["pay_off"] = np.arange(k, dtype=np.uint64) * pay_cap; d["pay_cap"] = pay_cap
    d["sig_off"] = np.arange(k, dtype=np.uint64) * sig_cap; d["sig_cap"] = sig_cap
    t1 = time.perf_counter()
    desc_pin[: d.nbytes].copy_(torch.from_numpy(d.view(np.uint8)))
    desc_dev[: d.nbytes].copy_(desc_pin[: d.nbytes], non_blocking=True)
    t2 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.n_recs = k
    e0.record()
    _lib.check(L.s5gpu_decode_dev(C.byref(a), b._stream()))
    e1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("batch %2d k=%d numpy %.2f copy %.2f launch %.2f sync %.2f total %.2f ms | gpu %.2f ms  maxlen %d" % (lo // K, k, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3, e0.elapsed_time(e1), int(d["in_len"].max())))
