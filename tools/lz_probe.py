"""Offline probe for the LZ77 matcher of the DEFLATE kernel (record press zlib over signal press none): what does a simple
matcher buy on the reference's exp_1_lossless_zlib.blow5 payload?  Variants: candidates = {3-byte hash table, one entry, last
writer wins} + fixed short distances; greedy parse; blocks of 16 KiB, history inside the block only; dynamic Huffman per block.
python tools/lz_probe.py"""
import heapq, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blow5_fixture import Blow5, golden

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def lsym(l):
    for i in range(28, -1, -1):
        if l >= LBASE[i]: return i
def dsym(d):
    for i in range(29, -1, -1):
        if d >= DBASE[i]: return i

def huff_cost(f):
    h = [(int(x), i) for i, x in enumerate(f) if x]
    if len(h) <= 1: return int(sum(f))
    heapq.heapify(h); tot = 0
    while len(h) > 1:
        a = heapq.heappop(h); b = heapq.heappop(h); tot += a[0] + b[0]; heapq.heappush(h, (a[0] + b[0], -1))
    return tot

def encode_block(blk, hbits=12, shorts=(1, 2, 4), chain=1, minlen=3, seg=None, lazy=False):
    n = len(blk); b = blk
    table = {}
    best = [(0, 0)] * n
    def mlen(p, c):
        l = 0
        while p + l < n and l < 258 and b[c + l] == b[p + l]: l += 1
        return l
    for p in range(n):
        cands = [p - d for d in shorts if p - d >= 0]
        if p + 2 < n:
            h = (b[p] | b[p + 1] << 8 | b[p + 2] << 16) * 2654435761 >> (32 - hbits) & ((1 << hbits) - 1)
            lst = table.get(h, [])
            # seg: positions of the same segment are not yet in the table when the segment looks up (parallel build)
            for c in lst[:chain]:
                if seg is None or (c // seg) < (p // seg): cands.append(c)
            table[h] = ([p] + lst)[:chain]
        bl, bd = 0, 0
        for c in cands:
            l = mlen(p, c)
            if l > bl or (l == bl and l and p - c < bd): bl, bd = l, p - c
        best[p] = (bl, bd) if bl >= minlen else (0, 0)
    fl = [0] * 286; fd = [0] * 30; extra = 0; p = 0; ntok = 0
    while p < n:
        l, d = best[p]
        if lazy and l and p + 1 < n and best[p + 1][0] > l: l = 0
        if l:
            ls, ds = lsym(l), dsym(d)
            fl[257 + ls] += 1; fd[ds] += 1; extra += LEXT[ls] + DEXT[ds]; p += l
        else:
            fl[b[p]] += 1; p += 1
        ntok += 1
    fl[256] += 1
    return huff_cost(fl) + huff_cost(fd) + extra + 400   # ~ header

pay = zlib.decompress(Blow5(golden("exp_1_lossless_zlib.blow5")).records[0])
ref = len(Blow5(golden("exp_1_lossless_zlib.blow5")).records[0])
print("payload %d bytes, reference record %d bytes, zlib-6 here %d" % (len(pay), ref, len(zlib.compress(pay, 6))))
for name, kw in [("RLE only (dist 1)", dict(shorts=(1,), chain=0)),
                 ("shorts 1,2,4 only", dict(chain=0)),
                 ("hash 4K x1 + shorts", dict()),
                 ("hash 4K x1 + shorts, seg 256", dict(seg=256)),
                 ("hash 32K x1 + shorts", dict(hbits=15)),
                 ("hash 4K x4 + shorts", dict(chain=4)),
                 ("hash 4K x1 + shorts + lazy", dict(lazy=True)),
                 ("hash 4K x1 + shorts, minlen 4", dict(minlen=4)),
                 ("hash 4K x4 + shorts + lazy", dict(chain=4, lazy=True)),
                 ("hash 32K x8 + shorts + lazy", dict(hbits=15, chain=8, lazy=True))]:
    bits = sum(encode_block(pay[o:o + 16384], **kw) for o in range(0, len(pay), 16384))
    print("%-36s %7d bytes  %.3f x reference" % (name, bits // 8 + 6, (bits // 8 + 6) / ref))

print("\ncost-aware selection (literal costs from the block's byte histogram; match cost = 7 + 5 + distance extra bits + length extra bits):")
import math
def encode_block2(blk, hbits=12, shorts=(1, 2, 4), chain=1, window=None, hist=b"", lmatch=7, dmatch=5):
    """hist: bytes in front of the block that may be referenced (a 32 K window across blocks)"""
    b = hist + blk; o = len(hist); n = len(b)
    cnt = np.bincount(np.frombuffer(blk, dtype=np.uint8), minlength=256).astype(float)
    cl = np.where(cnt > 0, -np.log2(np.maximum(cnt, 1) / cnt.sum()), 20.0)
    pre = np.concatenate([[0.0], np.cumsum(cl[np.frombuffer(b, dtype=np.uint8)])])
    table = {}
    best = [(0, 0, 0.0)] * n
    def mlen(p, c):
        l = 0
        while p + l < n and l < 258 and b[c + l] == b[p + l]: l += 1
        return l
    for p in range(n):
        cands = [p - d for d in shorts if p - d >= 0]
        if p + 2 < n:
            h = (b[p] | b[p + 1] << 8 | b[p + 2] << 16) * 2654435761 >> (32 - hbits) & ((1 << hbits) - 1)
            lst = table.get(h, [])
            cands += [c for c in lst[:chain] if p - c <= 32768]
            table[h] = ([p] + lst)[:chain]
        if p < o: continue
        bs, bl, bd = 0.0, 0, 0
        for c in cands:
            l = mlen(p, c)
            if l < 3: continue
            d = p - c
            cost = lmatch + dmatch + DEXT[dsym(d)] + LEXT[lsym(l)]
            sav = pre[p + l] - pre[p] - cost
            if sav > bs: bs, bl, bd = sav, l, d
        best[p] = (bl, bd, bs)
    fl = [0] * 286; fd = [0] * 30; extra = 0; p = o
    while p < n:
        l, d, s = best[p]
        if l:
            ls, ds = lsym(l), dsym(d)
            fl[257 + ls] += 1; fd[ds] += 1; extra += LEXT[ls] + DEXT[ds]; p += l
        else:
            fl[b[p]] += 1; p += 1
    fl[256] += 1
    return huff_cost(fl) + huff_cost(fd) + extra + 400

for name, kw, cross in [("hash 4K x1 + shorts", dict(), False), ("hash 4K x1 + shorts 1,2,3,4,6,8", dict(shorts=(1, 2, 3, 4, 6, 8)), False),
                        ("hash 4K x4 + shorts", dict(chain=4), False), ("hash 32K x4 + shorts", dict(hbits=15, chain=4), False),
                        ("hash 4K x1 + shorts, 32 K window", dict(), True), ("hash 32K x4 + shorts, 32 K window", dict(hbits=15, chain=4), True),
                        ("hash 32K x16 + shorts, 32 K window", dict(hbits=15, chain=16), True)]:
    bits = 0
    for o in range(0, len(pay), 16384):
        bits += encode_block2(pay[o:o + 16384], hist=pay[max(0, o - 32768):o] if cross else b"", **kw)
    print("%-40s %7d bytes  %.3f x reference" % (name, bits // 8 + 6, (bits // 8 + 6) / ref))
