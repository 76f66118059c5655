"""Hand-made dynamic DEFLATE blocks for the header tests (test infrastructure; every stream is first inflated by STOCK zlib, which is the
referee of what is valid — RFC 1951 3.2.7: HLIT / HDIST / HCLEN, the 3-bit code-length-code lengths in the order 16 17 18 0 8 7 9 …, the
code-length sequence with its repeat symbols 16 / 17 / 18).  The block holds literals only; what varies is the HEADER:

  lit/len code lengths   chosen by the caller (any complete canonical code)
  sequence tokens        "plain": one token per length, "rle": zlib-like runs, "rep0": symbol 16 behind zero lengths as well
  code-length code       from the token frequencies (Huffman, limited to 7 bits), or from caller's weights (shapes with 1-bit codes)
"""
import heapq
import struct
import zlib

ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nb):                     # nb bits of v, least significant first
        self.acc |= (v & ((1 << nb) - 1)) << self.n
        self.n += nb
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nb):             # a Huffman code: first bit of the code first
        for i in range(nb - 1, -1, -1):
            self.put((code >> i) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def huff_lengths(weights, limit):
    """code lengths of a Huffman code over the symbols with weight > 0 (at least two), every length <= limit (asserted: the callers
    choose weights that fit); a complete code"""
    live = [(w, i) for i, w in enumerate(weights) if w > 0]
    assert len(live) >= 2
    heap = [(w, i, (i,)) for w, i in live]
    heapq.heapify(heap)
    depth = [0] * len(weights)
    tick = len(weights)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        for s in a[2] + b[2]:
            depth[s] += 1
        heapq.heappush(heap, (a[0] + b[0], tick, a[2] + b[2]))
        tick += 1
    assert max(depth) <= limit, (max(depth), limit)
    return depth


def canonical(lengths):
    """RFC 1951 3.2.2: code of every symbol (None for length 0)"""
    bl = [0] * 17
    for l in lengths:
        bl[l] += 1
    bl[0] = 0
    nxt, code = [0] * 17, 0
    for b in range(1, 17):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    out = []
    for l in lengths:
        if l:
            out.append(nxt[l])
            nxt[l] += 1
        else:
            out.append(None)
    return out


def tokens_of(seq, mode):
    """the code-length sequence as (symbol, extra value) tokens.  plain: one token per length; rle: zero runs as 18 / 17, other runs as the
    length followed by 16s (what zlib writes); rep0: zero runs ALSO as "0, then 16s" (a repeat of a zero length: legal, zlib never writes it)"""
    out, i, n = [], 0, len(seq)
    while i < n:
        v = seq[i]
        j = i
        while j < n and seq[j] == v:
            j += 1
        run = j - i
        if mode == "plain":
            out += [(v, 0)] * run
        elif v == 0 and mode == "rle":
            while run >= 11:
                r = min(run, 138); out.append((18, r - 11)); run -= r
            if run >= 3:
                out.append((17, run - 3)); run = 0
            out += [(0, 0)] * run
        else:
            out.append((v, 0)); run -= 1
            while run >= 3:
                r = min(run, 6); out.append((16, r - 3)); run -= r
            out += [(v, 0)] * run
        i = j
    return out


def dynamic_block(payload, litlens, mode="rle", cl_weights=None, final=True, dlens=(1, 1)):
    """one dynamic block of literals: payload bytes under the lit/len code `litlens` (>= 257 entries, complete, every used byte and
    symbol 256 with a length), distance code lengths dlens"""
    assert len(litlens) >= 257 and len(litlens) <= 286 and litlens[256]
    seq = list(litlens) + list(dlens)
    toks = tokens_of(seq, mode)
    freq = [0] * 19
    for s, _ in toks:
        freq[s] += 1
    if cl_weights is not None:
        w = [cl_weights.get(s, 0) for s in range(19)]
        assert all(w[s] > 0 for s in range(19) if freq[s])
    else:
        w = list(freq)
    if sum(1 for x in w if x) < 2:
        w[[s for s in range(19) if not w[s]][0]] = 1
    cll = huff_lengths(w, 7)
    clc = canonical(cll)
    hclen = 19
    while hclen > 4 and cll[ORDER[hclen - 1]] == 0:
        hclen -= 1
    b = Bits()
    b.put(1 if final else 0, 1)
    b.put(2, 2)
    b.put(len(litlens) - 257, 5)
    b.put(len(dlens) - 1, 5)
    b.put(hclen - 4, 4)
    for k in range(hclen):
        b.put(cll[ORDER[k]], 3)
    hdr_start = 3 + 14 + 3 * hclen
    nbits = 0
    for s, x in toks:
        b.put_code(clc[s], cll[s]); nbits += cll[s]
        if s == 16:
            b.put(x, 2); nbits += 2
        elif s == 17:
            b.put(x, 3); nbits += 3
        elif s == 18:
            b.put(x, 7); nbits += 7
    lc = canonical(list(litlens))
    for x in payload:
        assert litlens[x], "byte %d has no code" % x
        b.put_code(lc[x], litlens[x])
    b.put_code(lc[256], litlens[256])
    return b, hdr_start, nbits


def zlib_stream(payload, litlens, **kw):
    """-> (zlib stream, bits of the code-length sequence); checked against stock zlib"""
    b, _, nbits = dynamic_block(payload, litlens, **kw)
    s = b"\x78\x9c" + b.done() + struct.pack(">I", zlib.adler32(payload) & 0xFFFFFFFF)
    assert zlib.decompress(s) == payload
    return s, nbits


def raw_stream_with_sequence(toks, cll, hlit, hdist, tail_bits=64):
    """a zlib stream whose dynamic header carries exactly these (symbol, extra) tokens under the code-length code `cll` — for headers that
    are NOT valid (a repeat with nothing in front, a run over the end); nothing follows but zero bits and a dummy trailer"""
    clc = canonical(cll)
    hclen = 19
    while hclen > 4 and cll[ORDER[hclen - 1]] == 0:
        hclen -= 1
    b = Bits()
    b.put(1, 1); b.put(2, 2); b.put(hlit - 257, 5); b.put(hdist - 1, 5); b.put(hclen - 4, 4)
    for k in range(hclen):
        b.put(cll[ORDER[k]], 3)
    for s, x in toks:
        b.put_code(clc[s], cll[s])
        if s == 16:
            b.put(x, 2)
        elif s == 17:
            b.put(x, 3)
        elif s == 18:
            b.put(x, 7)
    b.put(0, tail_bits)
    return b"\x78\x9c" + b.done() + b"\x00\x00\x00\x01"
