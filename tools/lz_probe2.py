"""Calibrated offline model of a greedy hash-chain matcher (zlib's deflate_fast shape) on the reference's
exp_1_lossless_zlib.blow5 payload: how do window, chain depth and the block size of the device kernel move the size?
python tools/lz_probe2.py"""
import heapq, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from blow5_fixture import Blow5, golden
from lz_probe import lsym, dsym, LEXT, DEXT, huff_cost

def parse(b, window, chain, too_far=4096, nice=258, hbits=15, hash4=False):
    n = len(b); heads = {}; prev = [-1] * n; toks = []
    p = 0
    def ins(q):
        if q + 2 < n:
            key = (b[q], b[q + 1], b[q + 2]) if not hash4 else (b[q], b[q + 1], b[q + 2], b[q + 3] if q + 3 < n else 0)
            prev[q] = heads.get(key, -1); heads[key] = q
    while p < n:
        bl, bd = 0, 0
        if p + 2 < n:
            key = (b[p], b[p + 1], b[p + 2]) if not hash4 else (b[p], b[p + 1], b[p + 2], b[p + 3] if p + 3 < n else 0)
            c = heads.get(key, -1); k = 0
            while c >= 0 and p - c <= window and k < chain:
                l = 3
                while p + l < n and l < 258 and b[c + l] == b[p + l]: l += 1
                if l > bl: bl, bd = l, p - c
                if l >= nice: break
                c = prev[c]; k += 1
        if bl == 3 and bd > too_far: bl = 0
        if bl >= 3:
            toks.append((p, bl, bd))
            for q in range(p, p + bl): ins(q)
            p += bl
        else:
            toks.append((p, 0, 0)); ins(p); p += 1
    return toks

def cost(b, toks, blk):
    """dynamic Huffman per block of `blk` input bytes (token starts decide the block), 100 bytes of header each"""
    bits = 0; cur = None; fl = fd = None; extra = 0
    def flush():
        nonlocal bits
        if fl is not None:
            fl[256] += 1; bits += huff_cost(fl) + huff_cost(fd) + extra + 800
    for p, l, d in toks:
        k = p // blk
        if k != cur:
            flush(); cur = k; fl = [0] * 286; fd = [0] * 30; extra = 0
        if l: ls, ds = lsym(l), dsym(d); fl[257 + ls] += 1; fd[ds] += 1; extra += LEXT[ls] + DEXT[ds]
        else: fl[b[p]] += 1
    flush()
    return bits // 8 + 6

pay = zlib.decompress(Blow5(golden("exp_1_lossless_zlib.blow5")).records[0]); ref = 72640
print("zlib level 1 / 2 / 3 / 6: %s" % [len(zlib.compress(pay, l)) for l in (1, 2, 3, 6)])
for name, kw, blk in [("window 32K chain 4 (zlib -1 shape), blocks 48K", dict(window=32768, chain=4, nice=8), 49152),
                      ("window 32K chain 8 (zlib -2 shape), blocks 48K", dict(window=32768, chain=8, nice=16), 49152),
                      ("window 32K chain 8, blocks 16K", dict(window=32768, chain=8), 16384),
                      ("window 32K chain 16, blocks 16K", dict(window=32768, chain=16), 16384),
                      ("window 32K chain 32, blocks 16K", dict(window=32768, chain=32), 16384),
                      ("window 16K chain 16, blocks 16K", dict(window=16384, chain=16), 16384),
                      ("window 16K chain 32, blocks 16K", dict(window=16384, chain=32), 16384),
                      ("window 16K chain 64, blocks 16K", dict(window=16384, chain=64), 16384),
                      ("window 8K chain 32, blocks 16K", dict(window=8192, chain=32), 16384),
                      ("window 16K chain 1000 (all), blocks 16K", dict(window=16384, chain=1000), 16384),
                      ("window 32K chain 1000 (all), blocks 16K", dict(window=32768, chain=1000), 16384),
                      ("window 32K chain 1000 (all), 4-byte hash, blocks 16K", dict(window=32768, chain=1000, hash4=True), 16384)]:
    t = parse(pay, **kw); c = cost(pay, t, blk)
    print("%-52s %6d bytes  %.4f x reference   (%d tokens, %d matches)" % (name, c, c / ref, len(t), sum(1 for x in t if x[1])))
