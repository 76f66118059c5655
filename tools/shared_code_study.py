import sys, os, glob, struct, zlib, heapq
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle_bind as ob

def tokens_hist(p):
    """lit/len histogram (286) of the run-length tokeniser: E = equals previous byte; a run of R >= 3 E-positions -> R//258 x 258, rest >= 3 one match else literals"""
    b = np.frombuffer(p, dtype=np.uint8)
    h = np.zeros(286, dtype=np.int64)
    n = len(b)
    e = np.zeros(n, dtype=bool); e[1:] = b[1:] == b[:-1]
    # run boundaries of E
    i = 0
    # vectorised: find E runs
    d = np.diff(np.concatenate(([0], e.view(np.int8), [0])))
    starts = np.where(d == 1)[0]; ends = np.where(d == -1)[0]
    lit = np.ones(n, dtype=bool)
    nm = 0; xb = 0
    for s, t in zip(starts, ends):
        R = t - s
        if R >= 3:
            nf, rem = divmod(R, 258)
            h[285] += nf; nm += nf
            lit[s:t] = False
            if rem >= 3:
                sym, eb = len_sym(rem); h[sym] += 1; nm += 1; xb += eb
            elif rem:
                lit[t - rem:t] = True
    h[:256] += np.bincount(b[lit], minlength=256)
    h[256] += 1
    return h, nm, xb

LB = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LE = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
def len_sym(L):
    for i in range(28, -1, -1):
        if L >= LB[i]: return 257 + i, LE[i]

def huff_lens(f, maxl=15):
    """Huffman code lengths, limited to maxl by the zlib-style heuristic (good enough for an estimate)"""
    f = np.asarray(f, dtype=np.float64)
    idx = [i for i in range(len(f)) if f[i] > 0]
    L = np.zeros(len(f), dtype=np.int64)
    if len(idx) == 1: L[idx[0]] = 1; return L
    heap = [(f[i], i, None, None) for i in idx]
    heapq.heapify(heap); cnt = len(f)
    nodes = {}
    while len(heap) > 1:
        a = heapq.heappop(heap); b = heapq.heappop(heap)
        nodes[cnt] = (a, b); heapq.heappush(heap, (a[0] + b[0], cnt, a, b)); cnt += 1
    def walk(n, d):
        if n[2] is None: L[n[1]] = max(d, 1)
        else: walk(n[2], d + 1); walk(n[3], d + 1)
    sys.setrecursionlimit(10000)
    walk(heap[0], 0)
    if L.max() > maxl:   # crude: package via Kraft repair
        L = np.minimum(L, maxl)
        while sum(2.0 ** -l for l in L[L > 0]) > 1.0 + 1e-12:
            # lengthen the least frequent symbol that is shorter than maxl
            c = [i for i in idx if L[i] < maxl]
            j = min(c, key=lambda i: (f[i], -L[i])); L[j] += 1
    return L

def entropy_bits(h):
    f = h[h > 0].astype(np.float64); N = f.sum()
    return float(N * np.log2(N) - (f * np.log2(f)).sum())

def payload(sig, rid=b"read_0000001"):
    r, keep = ob.make_rec(rid, 0, 8192.0, 23.0, 1467.61, 4000.0, np.ascontiguousarray(sig))
    return ob.rec_pack(r, ob.SIG_SVB_ZD)

def real_records():
    out = []
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/ref/**/*.blow5"), recursive=True)):
        b = open(path, "rb").read()
        if b[:6] != b"BLOW5\x01": continue
        rm, sm = b[9], b[14]
        (hl,) = struct.unpack_from("<I", b, 64)
        off = 68 + hl
        try:
            while b[off:off + 5] != b"5WOLB":
                (sz,) = struct.unpack_from("<Q", b, off)
                body = b[off + 8:off + 8 + sz]
                pl = zlib.decompress(body) if rm == 1 else (ob.zstd_decompress(body) if rm == 2 else body)
                d = ob.rec_parse(pl, sm)
                out.append((os.path.basename(os.path.dirname(path)) + "/" + os.path.basename(path), d["read_id"], d["signal"]))
                off += 8 + sz
        except Exception as e:
            pass
    return out

if __name__ == "__main__":
    H = []; tag = []
    for i in range(300):
        sig = ob.synth_read(0x5105, i * 37, 4000)
        h, nm, xb = tokens_hist(payload(sig, ob.synth_read_id(i * 37))); H.append(h); tag.append("synth")
    seen = set()
    for path, rid, sig in real_records():
        key = (bytes(rid), sig.size)
        if key in seen or sig.size < 500: continue
        seen.add(key)
        h, nm, xb = tokens_hist(payload(sig, bytes(rid)[:40])); H.append(h); tag.append("real:" + path.split("/")[0])
    H = np.array(H); tag = np.array(tag)
    print("records:", len(H), "synth", (tag == "synth").sum(), "real", (tag != "synth").sum())
    np.save("/tmp/H.npy", H); np.save("/tmp/tag.npy", tag)
