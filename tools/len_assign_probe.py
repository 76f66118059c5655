"""Offline (CPU, numpy) probe behind the parallel code-length assignment of the DEFLATE kernel: for the lit/len histograms of
real fixture records and of synthetic reads, compare the bit cost of
  (a) the optimal Huffman code (what the round-based merge builds),
  (b) Shannon lengths ceil(-log2 p) with the Kraft slack handed out greedily by benefit per Kraft unit (one histogram pass),
      optionally repeated,
against each other.  Tokens as the device makes them: runs of >= 4 equal bytes -> literal + (length, distance 1) matches.
python tools/len_assign_probe.py"""
import glob, heapq, math, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind as ob
from blow5_fixture import Blow5, golden

def len_sym(l):
    l = int(l)
    l -= 3
    if l == 255: return 285, 0
    if l < 8: return 257 + l, 0
    nb = l.bit_length() - 3
    return 261 + 4 * nb + ((l >> nb) & 3), nb

def tokenize(buf):
    f = np.zeros(286, dtype=np.int64); extra = 0; nm = 0
    b = np.frombuffer(buf, dtype=np.uint8)
    n = len(b); i = 0
    brk = np.flatnonzero(np.concatenate([[True], b[1:] != b[:-1]]))
    ends = np.concatenate([brk[1:], [n]])
    for s, e in zip(brk, ends):
        f[b[s]] += 1
        body = e - s - 1
        if body >= 3:
            while body >= 3:
                L = min(258, body); sym, eb = len_sym(L); f[sym] += 1; extra += eb; nm += 1; body -= L
            f[b[s]] += body
        else:
            f[b[s]] += body
    f[256] += 1
    return f, extra, nm

def huffman_lengths(f, maxbits=15):
    syms = [i for i in range(len(f)) if f[i]]
    if len(syms) == 1: return {syms[0]: 1}
    h = [(int(f[s]), s, None, None) for s in syms]; heapq.heapify(h); cnt = len(f)
    while len(h) > 1:
        a = heapq.heappop(h); b = heapq.heappop(h); cnt += 1
        heapq.heappush(h, (a[0] + b[0], cnt, a, b))
    out = {}
    def walk(nd, d):
        if nd[2] is None: out[nd[1]] = d
        else: walk(nd[2], d + 1); walk(nd[3], d + 1)
    walk(h[0], 0)
    assert max(out.values()) <= maxbits
    return out

def shannon_fill(f, passes=2, bins=64, maxbits=15):
    """ceil(-log2 p); then per pass: symbols sorted (by histogram bin) by f * 2^l descending get one bit shorter while the Kraft sum stays <= 1"""
    N = f.sum(); idx = np.flatnonzero(f)
    l = np.ceil(-np.log2(f[idx] / N)).astype(int); l = np.clip(l, 1, maxbits)
    for _ in range(passes):
        K = (2.0 ** -l).sum(); slack = 1.0 - K
        if slack <= 0: break
        ratio = f[idx] * (2.0 ** l) / N          # in [1, 2) after ceil
        q = np.clip(((np.log2(ratio)) * bins).astype(int), 0, bins - 1)   # bin by log of the ratio
        order_bins = np.zeros(bins); np.add.at(order_bins, q, 2.0 ** -l)  # Kraft cost of shortening = 2^-l each
        # highest bins first
        cum = np.cumsum(order_bins[::-1])[::-1]
        ok = cum <= slack
        thr = bins
        for bb in range(bins - 1, -1, -1):
            if ok[bb]: thr = bb
            else: break
        sel = (q >= thr) & (l > 1)
        l = np.where(sel, l - 1, l)
    return dict(zip(idx.tolist(), l.tolist()))

def cost(f, lens): return sum(int(f[s]) * l for s, l in lens.items())

def payloads():
    for name in ["exp_1_lossless_zlib_svb_v0.2.0.blow5", "sp1_dna.blow5", "merged_expected_zlib_svb.blow5", "example_multi_rg_v0.2.0.blow5", "gridr10dna_b3.blow5"]:
        try:
            b = Blow5(golden(name))
        except Exception: continue
        if b.rec_method != 1: continue
        for r in b.records: yield name, zlib.decompress(r)
    for i in range(40):
        sig = ob.synth_read(0x5105, i, 4000)
        r, keep = ob.make_rec(ob.synth_read_id(i), 0, 8192.0, 23.0, 1467.61, 4000.0, sig)
        yield "synth", ob.rec_pack(r, ob.SIG_SVB_ZD)

tot = {}
for name, pay in payloads():
    for off in range(0, len(pay), 16384):
        blk = pay[off:off + 16384]
        f, extra, nm = tokenize(blk)
        h = cost(f, huffman_lengths(f)); 
        res = [h]
        for p in (1, 2, 3):
            L = shannon_fill(f, passes=p)
            assert sum(2.0 ** -v for v in L.values()) <= 1 + 1e-12
            res.append(cost(f, L))
        ent = -(f[f > 0] * np.log2(f[f > 0] / f.sum())).sum()
        t = tot.setdefault(name, np.zeros(6)); t += np.array(res + [ent, 8 * len(blk)])
for name, t in tot.items():
    print("%-40s huffman %.4f bits/byte | shannon+fill x1 %+.3f%% x2 %+.3f%% x3 %+.3f%% | entropy %+.3f%%" % (name, t[0] / t[5] * 8, (t[1] / t[0] - 1) * 100, (t[2] / t[0] - 1) * 100, (t[3] / t[0] - 1) * 100, (t[4] / t[0] - 1) * 100))

def greedy_skip(f, bins=None, rounds=1, maxbits=15):
    """ceil lengths, then symbols in descending f*2^l order are shortened whenever their Kraft cost still fits (skipping those that do not)"""
    N = f.sum(); idx = np.flatnonzero(f)
    l = np.clip(np.ceil(-np.log2(f[idx] / N)).astype(int), 1, maxbits)
    for _ in range(rounds):
        slack = 1.0 - (2.0 ** -l).sum()
        ratio = f[idx] * (2.0 ** l)
        if bins: ratio = np.floor(np.log2(ratio / N) * bins)        # quantised order, ties by symbol index
        order = np.lexsort((idx, -ratio))
        for k in order:
            c = 2.0 ** -l[k]
            if c <= slack + 1e-15 and l[k] > 1:
                slack -= c; l[k] -= 1
    return dict(zip(idx.tolist(), l.tolist()))

print()
tot = {}
for name, pay in payloads():
    for off in range(0, len(pay), 16384):
        blk = pay[off:off + 16384]
        f, extra, nm = tokenize(blk)
        h = cost(f, huffman_lengths(f))
        res = [h, cost(f, greedy_skip(f)), cost(f, greedy_skip(f, rounds=2)), cost(f, greedy_skip(f, bins=64)), cost(f, greedy_skip(f, bins=64, rounds=2)), cost(f, greedy_skip(f, bins=16, rounds=2))]
        t = tot.setdefault(name, np.zeros(len(res))); t += np.array(res)
for name, t in tot.items():
    print("%-40s greedy-skip exact %+.3f%% x2 %+.3f%% | 64 bins %+.3f%% x2 %+.3f%% | 16 bins x2 %+.3f%%" % ((name,) + tuple((t[k] / t[0] - 1) * 100 for k in range(1, 6))))

def threshold_passes(f, bins=64, passes=4, maxbits=15, allow_twice=True):
    """the parallel form: per pass, among symbols whose Kraft cost still fits the slack, take whole priority bins from the top while they fit"""
    N = f.sum(); idx = np.flatnonzero(f)
    l = np.clip(np.ceil(-np.log2(f[idx] / N)).astype(int), 1, maxbits)
    U = 1 << maxbits
    done = np.zeros(len(idx), dtype=bool)
    for p in range(passes):
        slack = U - (U >> l).sum()
        if slack <= 0: break
        c = U >> l
        elig = (c <= slack) & (l > 1) & (~done if not allow_twice else True)
        if not elig.any(): break
        pr = f[idx] * (2.0 ** l) / N
        q = np.clip(np.floor(np.log2(pr) * bins).astype(int) + (bins if allow_twice else 0), 0, 2 * bins - 1)   # ratio in [0.5, 2): already-shortened symbols sit one octave lower
        hb = np.zeros(2 * bins, dtype=np.int64); np.add.at(hb, q[elig], c[elig])
        cum = np.cumsum(hb[::-1])[::-1]
        thr = 2 * bins
        for bb in range(2 * bins - 1, -1, -1):
            if cum[bb] <= slack: thr = bb
            else: break
        sel = elig & (q >= thr)
        if not sel.any():
            # nothing fits bin-wise: take the single best eligible symbol (a lane-level arg-max on the device)
            k = np.flatnonzero(elig)[np.argmax(pr[elig])]
            sel = np.zeros(len(idx), dtype=bool); sel[k] = True
        l = np.where(sel, l - 1, l); done |= sel
    assert (U >> l).sum() <= U
    return dict(zip(idx.tolist(), l.tolist()))

print()
for bins, passes in ((64, 2), (64, 4), (64, 6), (64, 8), (32, 6), (16, 6)):
    tot = {}
    for name, pay in payloads():
        for off in range(0, len(pay), 16384):
            f, extra, nm = tokenize(pay[off:off + 16384])
            res = [cost(f, huffman_lengths(f)), cost(f, threshold_passes(f, bins, passes))]
            t = tot.setdefault(name, np.zeros(2)); t += np.array(res)
    print("bins %d passes %d: " % (bins, passes) + "  ".join("%s %+.3f%%" % (n[:12], (t[1] / t[0] - 1) * 100) for n, t in tot.items()))

def device_form(f, nb=128, passes=5, maxbits=15, stats=None):
    """what the kernel does (integer arithmetic, one wave, 5 symbols per lane in symbol order):
    exact ceil lengths; per pass linear priority bins over ratio [0.5, 2), whole bins from the top while they fit, then the next
    non-empty bin partially in symbol order; finisher by length classes until the code is complete"""
    N = int(f.sum()); idx = np.flatnonzero(f); fs = f[idx].astype(np.int64)
    U = 1 << maxbits
    l = np.array([max(1, (N - 1).bit_length() - (int(x)).bit_length() + (1 if (int(x) << ((N - 1).bit_length() - int(x).bit_length())) < N else 0)) if N > 1 else 1 for x in fs])
    # check: smallest l with f << l >= N
    for x, ll in zip(fs, l): assert (int(x) << int(ll)) >= N and (ll == 1 or (int(x) << int(ll - 1)) < N), (x, ll, N)
    R = U - int((U >> l).sum())
    assert R >= 0
    np_ = 0
    for p in range(passes):
        if R == 0: break
        np_ += 1
        c = U >> l
        elig = (c <= R) & (l > 1)
        if not elig.any(): break
        q = np.clip((fs << l) * (nb // 2) // N, 0, nb - 1)          # ratio * nb/2: [nb/4, nb) for ratio in [0.5, 2)
        hb = np.zeros(nb + 1, dtype=np.int64); np.add.at(hb, q[elig], c[elig])
        cum = np.cumsum(hb[::-1])[::-1]                              # cum[b] = cost of bins >= b
        thr = nb
        while thr > 0 and cum[thr - 1] <= R: thr -= 1
        sel = elig & (q >= thr)
        R -= int(cum[thr])
        # the next non-empty bin below thr, partially, in symbol order
        pb = thr - 1
        while pb >= 0 and hb[pb] == 0: pb -= 1
        if pb >= 0:
            for k in np.flatnonzero(elig & (q == pb)):
                if c[k] <= R: sel[k] = True; R -= int(c[k])
                else: break                                            # prefix only (a wave prefix sum on the device)
        l = np.where(sel, l - 1, l)
    nfin = 0
    while R > 0:                                                       # finisher: by length class, shortest codes (largest units) first
        nfin += 1
        for L in range(2, maxbits + 1):
            unit = U >> L
            cand = np.flatnonzero(l == L)
            t = min(len(cand), R // unit)
            if t:
                l[cand[:t]] -= 1; R -= t * unit
    assert (U >> l).sum() == U
    if stats is not None: stats.append((np_, nfin))
    return dict(zip(idx.tolist(), l.tolist()))

print()
for nb, passes in ((128, 3), (128, 4), (128, 5), (128, 6), (64, 5), (256, 5)):
    tot = {}; st = []
    for name, pay in payloads():
        for off in range(0, len(pay), 16384):
            f, extra, nm = tokenize(pay[off:off + 16384])
            res = [cost(f, huffman_lengths(f)), cost(f, device_form(f, nb, passes, stats=st))]
            t = tot.setdefault(name, np.zeros(2)); t += np.array(res)
    print("bins %d passes %d: " % (nb, passes) + "  ".join("%s %+.3f%%" % (n[:12], (t[1] / t[0] - 1) * 100) for n, t in tot.items()) + "  finisher used in %d of %d blocks (max %d rounds)" % (sum(1 for a, b in st if b), len(st), max(b for a, b in st)))
