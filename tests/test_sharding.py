"""N>1 path on CPU (gloo, world_size 2): reads shard by index with no data-path collective; the shards
written back in rank order are byte-identical to the single-process result (the ordered fwrite of
/root/reference/src/view.c:296-299).  The per-shard compute is the oracle here (no GPU in this container)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _encode_shard(lo, hi, n):
    import oracle_bind as ob

    out = []
    for i in range(lo, hi):
        sig = ob.synth_read(0x5105, i, n)
        r, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
        out.append(ob.rec_to_mem(r, ob.REC_ZLIB, ob.SIG_SVB_ZD))
    return b"".join(out)


def _worker(rank, world, port, n_total, n, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time

    import torch.distributed as dist

    from slow5tools_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_total, rank, world)
    shard.barrier()
    t0 = time.perf_counter()
    blob = _encode_shard(lo, hi, n)
    shard.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0 + 0.01 * rank)
    open(os.path.join(tmp, "shard%d.bin" % rank), "wb").write(blob)
    open(os.path.join(tmp, "time%d.txt" % rank), "w").write(repr(dt))
    dist.destroy_process_group()


def test_shard_ranges_partition_the_batch():
    sys.path.insert(0, ROOT)
    from slow5tools_amd import shard

    for n in (0, 1, 7, 4096, 1_000_003):
        for w in (1, 2, 3, 8):
            edges = [shard.shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_two_rank_gloo_sharded_encode_matches_single_process(tmp_path):
    n_total, n, world = 37, 1500, 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n_total, n, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    whole = _encode_shard(0, n_total, n)
    parts = b"".join(open(tmp_path / ("shard%d.bin" % r), "rb").read() for r in range(world))
    assert parts == whole
    times = [float(open(tmp_path / ("time%d.txt" % r)).read()) for r in range(world)]
    assert times[0] == times[1] and times[0] >= 0.01       # MAX over ranks reached every rank
