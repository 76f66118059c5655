/*
 * zstd_enc.c — CPU ORACLE side of the device zstd ENCODER (csrc/zstd_enc_dev.h).  TEST INFRASTRUCTURE ONLY.
 *
 * slow5lib compresses a record with libzstd's ZSTD_compress (level 1).  Like the DEFLATE side, the device encoder does not
 * reproduce libzstd's bytes (its match finder is a serial hash chain): the contract is a VALID Zstandard frame that libzstd
 * itself decompresses to the identical payload.  The device writes "literals-only" frames: the payload is cut into blocks of
 * at most 16 KiB (the LDS stage of the staged path), each block is one of
 *   - a raw block            (too small, or Huffman would not shrink it),
 *   - an RLE block           (one distinct byte),
 *   - a compressed block     = literals (Huffman-coded in 4 streams, raw, or RLE) + a sequences section that holds the block's
 *                              RUNS: a run of >= ZE_RMIN equal bytes is its first byte as a literal + one match at offset 1 —
 *                              the frame's first repeat offset, i.e. offset code 0 for every sequence (the offset table is one
 *                              RLE byte, no offset bits at all); literal and match lengths in the predefined FSE tables.  That is
 *                              what libzstd's match finder gets out of the key bytes of an svb-zd record.
 * This file states that frame layout once, on the CPU, in the plainest possible way, so that the format details the device
 * has to get right (tree description with FSE-compressed weights, the two interleaved FSE states, normalised-count header,
 * reverse bit order of the streams, jump table, section headers) are pinned against libzstd where iteration is cheap
 * (tests/test_oracle_golden.py::test_zstd_literals_only_frames_are_valid).  The Huffman code LENGTHS are a free choice of
 * the encoder; this file uses a plain two-queue Huffman with a Kraft repair, the device its round-based construction —
 * frames differ in bytes, both decode to the payload.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "s5oracle.h"

#define ZE_BLK 16384
#define ZE_MAXBITS 11

typedef struct { uint8_t *p; uint64_t acc; int n; } bitw;   /* forward LSB-first bit writer */
static void bw_add(bitw *b, uint32_t v, int nb) {
    b->acc |= (uint64_t)v << b->n;
    b->n += nb;
    while (b->n >= 8) { *b->p++ = (uint8_t)b->acc; b->acc >>= 8; b->n -= 8; }
}
static void bw_close(bitw *b) { bw_add(b, 1, 1); if (b->n) { *b->p++ = (uint8_t)b->acc; b->acc = 0; b->n = 0; } }   /* end mark */
static int hb(uint32_t v) { int n = -1; while (v) { v >>= 1; n++; } return n; }

/* code lengths (<= ZE_MAXBITS) for the bytes of a block: heap-free Huffman by repeated minimum search (256 symbols: cheap) */
static void huf_lengths(const uint32_t *freq, uint8_t *len) {
    int parent[512], alive[512], n = 0;
    uint64_t w[512];
    for (int s = 0; s < 256; s++) { len[s] = 0; if (freq[s]) { } }
    int node_of[256];
    for (int s = 0; s < 256; s++) if (freq[s]) { w[n] = freq[s]; alive[n] = 1; parent[n] = -1; node_of[s] = n; n++; } else node_of[s] = -1;
    int leaves = n, total = n;
    for (int step = 0; step + 1 < leaves; step++) {
        int a = -1, b = -1;
        for (int i = 0; i < total; i++) if (alive[i]) {
            if (a < 0 || w[i] < w[a]) { b = a; a = i; }
            else if (b < 0 || w[i] < w[b]) b = i;
        }
        w[total] = w[a] + w[b]; alive[total] = 1; parent[total] = -1;
        alive[a] = alive[b] = 0; parent[a] = parent[b] = total;
        total++;
    }
    uint32_t cnt[64];
    memset(cnt, 0, sizeof cnt);
    for (int s = 0; s < 256; s++) if (node_of[s] >= 0) {
        int d = 0;
        for (int p = node_of[s]; parent[p] >= 0; p = parent[p]) d++;
        if (d > ZE_MAXBITS) d = ZE_MAXBITS;
        len[s] = (uint8_t)d;
        cnt[d]++;
    }
    /* Kraft repair after the clamp: lengthen the cheapest shorter codes until the sum fits */
    uint32_t kraft = 0;
    for (int L = 1; L <= ZE_MAXBITS; L++) kraft += cnt[L] << (ZE_MAXBITS - L);
    while (kraft > (1u << ZE_MAXBITS)) {
        int best = -1;
        for (int s = 0; s < 256; s++) if (len[s] && len[s] < ZE_MAXBITS && (best < 0 || len[s] > len[best] || (len[s] == len[best] && freq[s] < freq[best]))) best = s;
        kraft -= 1u << (ZE_MAXBITS - len[best] - 1);
        len[best]++;
    }
    /* and the other way: a clamp can leave room; shorten the most frequent longest codes while it fits */
    for (;;) {
        int best = -1;
        for (int s = 0; s < 256; s++) if (len[s] > 1 && kraft + (1u << (ZE_MAXBITS - len[s])) <= (1u << ZE_MAXBITS) && (best < 0 || freq[s] > freq[best])) best = s;
        if (best < 0) break;
        kraft += 1u << (ZE_MAXBITS - len[best]);
        len[best]--;
    }
}

/* FSE decode table of a normalised distribution, as the decoder builds it (oracle/zstd_dec.c fse_build) */
typedef struct { uint8_t sym, nb; uint16_t base; } fcell;
static void fse_cells(fcell *t, const int16_t *norm, int maxsym, int log) {
    const int size = 1 << log;
    uint16_t next[16];
    uint8_t cell[64];
    int high = size - 1;
    for (int s = 0; s <= maxsym; s++) { if (norm[s] == -1) { cell[high--] = (uint8_t)s; next[s] = 1; } else next[s] = (uint16_t)norm[s]; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s <= maxsym; s++)
        for (int i = 0; i < norm[s]; i++) { cell[pos] = (uint8_t)s; do { pos = (pos + step) & mask; } while (pos > high); }
    for (int i = 0; i < size; i++) {
        const int s = cell[i];
        const uint32_t ns = next[s]++;
        const int nb = log - hb(ns);
        t[i].sym = (uint8_t)s; t[i].nb = (uint8_t)nb; t[i].base = (uint16_t)((ns << nb) - (uint32_t)size);
    }
}

/* Huffman tree description for weights w[0..n) (the weight of symbol n is implied).  Returns bytes written, 0 = cannot. */
static size_t tree_desc(const uint8_t *w, int n, uint8_t *out) {
    if (n <= 128) {                                   /* direct: 4 bits per weight */
        out[0] = (uint8_t)(127 + n);
        for (int i = 0; i < n; i += 2) out[1 + i / 2] = (uint8_t)((w[i] << 4) | (i + 1 < n ? w[i + 1] : 0));
        return 1 + (size_t)(n + 1) / 2;
    }
    /* FSE-compressed weights, table log 6 */
    const int log = 6, size = 64;
    uint32_t cnt[13];
    int16_t norm[13];
    memset(cnt, 0, sizeof cnt);
    int maxw = 0;
    for (int i = 0; i < n; i++) { cnt[w[i]]++; if (w[i] > maxw) maxw = w[i]; }
    int big = 0, sum = 0;
    for (int s = 0; s <= maxw; s++) {
        norm[s] = 0;
        if (!cnt[s]) continue;
        if (cnt[s] == (uint32_t)n) return 0;          /* a single weight value: FSE cannot end such a stream */
        int v = (int)((uint64_t)cnt[s] * size / (uint32_t)n);
        norm[s] = (int16_t)(v < 1 ? 1 : v);
        sum += norm[s];
        if (cnt[s] > cnt[big] || !cnt[big]) big = s;
    }
    if (sum < size) norm[big] = (int16_t)(norm[big] + size - sum);
    while (sum > size) {                              /* take from the largest entries */
        int m = 0;
        for (int s = 0; s <= maxw; s++) if (norm[s] > norm[m]) m = s;
        norm[m]--; sum--;
    }
    /* normalised counts (the writer side of fse_read_ncount) */
    bitw b = {out + 1, 0, 0};
    bw_add(&b, (uint32_t)(log - 5), 4);
    int remaining = size + 1, threshold = size, nbits = log + 1, prev0 = 0, s = 0;
    while (s <= maxw && remaining > 1) {
        if (prev0) {
            int start = s;
            while (!norm[s]) s++;                     /* a present symbol follows: remaining > 1 */
            while (s >= start + 3) { start += 3; bw_add(&b, 3, 2); }
            bw_add(&b, (uint32_t)(s - start), 2);
        }
        int count = norm[s++];
        const int maxv = (2 * threshold - 1) - remaining;
        remaining -= count;
        count++;
        if (count >= threshold) count += maxv;
        bw_add(&b, (uint32_t)count, nbits - (count < maxv));
        prev0 = count == 1;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    if (b.n) { *b.p++ = (uint8_t)b.acc; b.acc = 0; b.n = 0; }
    /* the two interleaved state chains, walked backwards (decoder: state1 emits w[0], state2 w[1], state1 w[2] ...) */
    fcell t[64];
    fse_cells(t, norm, maxw, log);
    int first[13];
    for (int q = 0; q <= maxw; q++) first[q] = -1;
    for (int i = size - 1; i >= 0; i--) first[t[i].sym] = i;     /* lowest cell of a symbol = its largest bit cost (>= 1 bit) */
    int st[2];
    st[(n - 1) & 1] = first[w[n - 1]];
    st[(n - 2) & 1] = first[w[n - 2]];
    bitw fb = {b.p, 0, 0};
    for (int k = n - 3; k >= 0; k--) {
        const int next = st[k & 1];
        int found = -1;
        for (int i = 0; i < size; i++) if (t[i].sym == w[k] && next >= t[i].base && next < t[i].base + (1 << t[i].nb)) found = i;
        bw_add(&fb, (uint32_t)(next - t[found].base), t[found].nb);
        st[k & 1] = found;
    }
    bw_add(&fb, (uint32_t)st[1], log);
    bw_add(&fb, (uint32_t)st[0], log);
    bw_close(&fb);
    const size_t total = (size_t)(fb.p - (out + 1));
    if (total >= 128) return 0;
    out[0] = (uint8_t)total;
    return 1 + total;
}

/* ---- sequences: runs only ---- */
#define ZE_RMIN 5      /* shortest run that becomes literal + match (0.8735 B/sample on the bench reads; 4: 0.8729, 8: 0.8783, none: 0.8959).
                          5 and not 4: the device keeps its 4-byte sequence records in the bytes the matches free */

static const int16_t LL_NORM[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_NORM[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* FSE encoding table of a normalised distribution (table log 6): next-state table sorted by symbol, and per symbol the two
 * deltas of the usual formulation: bits out = (state + dnb) >> 16, next = table[(state >> bits) + dfs] */
typedef struct { uint16_t next[64]; int32_t dnb[53], dfs[53]; } fse_ct;
static void fse_ctable(fse_ct *ct, const int16_t *norm, int nsym) {
    const int log = 6, size = 64, mask = 63, step = (size >> 1) + (size >> 3) + 3;
    int cumul[54], high = size - 1;
    uint8_t cell[64];
    cumul[0] = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { cumul[s + 1] = cumul[s] + 1; cell[high--] = (uint8_t)s; }
        else cumul[s + 1] = cumul[s] + norm[s];
    }
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) { cell[pos] = (uint8_t)s; do { pos = (pos + step) & mask; } while (pos > high); }
    for (int u = 0; u < size; u++) ct->next[cumul[cell[u]]++] = (uint16_t)(size + u);
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == 0) { ct->dnb[s] = ((log + 1) << 16) - size; ct->dfs[s] = 0; }
        else if (norm[s] == 1 || norm[s] == -1) { ct->dnb[s] = (log << 16) - size; ct->dfs[s] = total - 1; total++; }
        else {
            const int maxbits = log - hb((uint32_t)norm[s] - 1), minplus = norm[s] << maxbits;
            ct->dnb[s] = (maxbits << 16) - minplus; ct->dfs[s] = total - norm[s]; total += norm[s];
        }
    }
}
static uint32_t fse_first(const fse_ct *ct, int sym) {                 /* the state a chain starts in (no bits) */
    const uint32_t nb = (uint32_t)(ct->dnb[sym] + (1 << 15)) >> 16;
    const uint32_t v = (nb << 16) - (uint32_t)ct->dnb[sym];
    return ct->next[(int)(v >> nb) + ct->dfs[sym]];
}
static uint32_t fse_step(const fse_ct *ct, uint32_t state, int sym, bitw *b) {
    const uint32_t nb = (state + (uint32_t)ct->dnb[sym]) >> 16;
    bw_add(b, state & ((1u << nb) - 1), (int)nb);
    return ct->next[(int)(state >> nb) + ct->dfs[sym]];
}
static int ll_code(uint32_t ll) {
    if (ll < 16) return (int)ll;
    if (ll < 24) return 16 + (int)((ll - 16) >> 1);
    if (ll < 32) return 20 + (int)((ll - 24) >> 2);
    if (ll < 48) return 22 + (int)((ll - 32) >> 3);
    if (ll < 64) return 24;
    return hb(ll) + 19;
}
static int ml_code(uint32_t mb) {                                      /* mb = match length - 3 */
    if (mb < 32) return (int)mb;
    if (mb < 40) return 32 + (int)((mb - 32) >> 1);
    if (mb < 48) return 36 + (int)((mb - 40) >> 2);
    if (mb < 64) return 38 + (int)((mb - 48) >> 3);
    if (mb < 96) return 40 + (int)((mb - 64) >> 4);
    if (mb < 128) return 42;
    return hb(mb) + 36;
}

/* literals section of lit[0..n): Huffman-coded (4 streams), RLE, or raw, whichever is smallest.  Returns bytes written. */
static size_t enc_literals(const uint8_t *in, size_t n, uint8_t *body) {
    uint32_t freq[256];
    memset(freq, 0, sizeof freq);
    for (size_t i = 0; i < n; i++) freq[in[i]]++;
    int distinct = 0, maxsym = 0;
    for (int s = 0; s < 256; s++) if (freq[s]) { distinct++; maxsym = s; }
    size_t blen = 0;
    if (n >= 64 && distinct > 1) {
        uint8_t len[256], w[256];
        huf_lengths(freq, len);
        int maxbits = 0;
        for (int s = 0; s < 256; s++) if (len[s] > maxbits) maxbits = len[s];
        for (int s = 0; s < 256; s++) w[s] = len[s] ? (uint8_t)(maxbits + 1 - len[s]) : 0;
        /* codes: longest first, symbol order inside a length */
        uint32_t code[256], cntL[ZE_MAXBITS + 2], base[ZE_MAXBITS + 2];
        memset(cntL, 0, sizeof cntL);
        for (int s = 0; s < 256; s++) cntL[len[s]]++;
        base[maxbits] = 0;
        for (int L = maxbits - 1; L >= 1; L--) base[L] = (base[L + 1] + cntL[L + 1]) >> 1;
        for (int s = 0; s < 256; s++) if (len[s]) code[s] = base[len[s]]++;
        uint8_t desc[160];
        const size_t dl = tree_desc(w, maxsym, desc);
        if (dl) {
            /* section: header | tree | jump table | 4 streams */
            static uint8_t tmp[4][ZE_BLK / 4 * 11 / 8 + 16];
            size_t sl[4];
            const size_t per = (n + 3) / 4;
            for (int k = 0; k < 4 && 3 * per <= n; k++) {
                const size_t from = (size_t)k * per, to = k == 3 ? n : from + per;
                bitw b = {tmp[k], 0, 0};
                for (size_t i = to; i > from; i--) bw_add(&b, code[in[i - 1]], len[in[i - 1]]);
                bw_close(&b);
                sl[k] = (size_t)(b.p - tmp[k]);
            }
            const size_t csize = 3 * per <= n ? dl + 6 + sl[0] + sl[1] + sl[2] + sl[3] : 0;
            const int hl = (n <= 1023 && csize <= 1023) ? 3 : (n <= 16383 && csize <= 16383) ? 4 : 5;
            if (csize && hl + csize + 1 < n) {
                uint64_t h = 2u | ((uint64_t)(hl - 2) << 2) | ((uint64_t)n << 4);          /* compressed literals, 4 streams */
                h |= (uint64_t)csize << (hl == 3 ? 14 : hl == 4 ? 18 : 22);
                for (int i = 0; i < hl; i++) body[blen++] = (uint8_t)(h >> (8 * i));
                memcpy(body + blen, desc, dl); blen += dl;
                for (int k = 0; k < 3; k++) { body[blen++] = (uint8_t)sl[k]; body[blen++] = (uint8_t)(sl[k] >> 8); }
                for (int k = 0; k < 4; k++) { memcpy(body + blen, tmp[k], sl[k]); blen += sl[k]; }
                return blen;
            }
        }
    }
    /* raw (type 0) or RLE (type 1) literals: 1-, 2- or 3-byte header with the regenerated size */
    const uint32_t type = n >= 2 && distinct == 1 ? 1u : 0u;
    if (n < 32) body[blen++] = (uint8_t)(type | (n << 3));
    else if (n < 4096) { const uint32_t h = type | (1u << 2) | ((uint32_t)n << 4); body[blen++] = (uint8_t)h; body[blen++] = (uint8_t)(h >> 8); }
    else { const uint32_t h = type | (3u << 2) | ((uint32_t)n << 4); body[blen++] = (uint8_t)h; body[blen++] = (uint8_t)(h >> 8); body[blen++] = (uint8_t)(h >> 16); }
    if (type) body[blen++] = in[0];
    else { memcpy(body + blen, in, n); blen += n; }
    return blen;
}

/* one block; returns bytes written (header included) */
static size_t enc_block(const uint8_t *in, size_t n, int last, uint8_t *out) {
    static uint8_t body[2 * ZE_BLK + 1024], lit[ZE_BLK];
    static uint16_t sq_lit[ZE_BLK / ZE_RMIN + 1], sq_ml[ZE_BLK / ZE_RMIN + 1];
    size_t blen = 0, nlit = 0, nseq = 0;
    int type = 0;
    /* runs: [s, e) maximal with equal bytes */
    for (size_t s = 0; s < n;) {
        size_t e = s + 1;
        while (e < n && in[e] == in[s]) e++;
        if (e - s >= ZE_RMIN) { lit[nlit++] = in[s]; sq_lit[nseq] = (uint16_t)nlit; sq_ml[nseq] = (uint16_t)(e - s - 1); nseq++; }
        else for (size_t i = s; i < e; i++) lit[nlit++] = in[i];
        s = e;
    }
    /* sequences only where they certainly pay: even with raw literals and every state transition at its 6-bit maximum the
     * block stays below n (extra bits of a literal length L are <= L / 4, of a match length M <= (M - 3) / 8); and only where the
     * device has room for its sequence records behind the compacted literals */
    if (nseq && !(nlit + 4 * nseq + 8 <= n && 3 + nlit + 4 + ((12 * nseq + (nlit >> 2) + ((n - nlit) >> 3) + 20) >> 3) + 1 < n) && !(nseq == 1 && nlit == 1)) nseq = 0;
    if (n >= 64 && nseq == 1 && nlit == 1) type = 1;                        /* one distinct byte: an RLE block */
    else if (n >= 64) {
        blen = enc_literals(nseq ? lit : in, nseq ? nlit : n, body);
        if (!nseq) body[blen++] = 0;                                         /* no sequences */
        else {
            static fse_ct LLT, MLT;
            static int have;
            if (!have) { fse_ctable(&LLT, LL_NORM, 36); fse_ctable(&MLT, ML_NORM, 53); have = 1; }
            if (nseq < 128) body[blen++] = (uint8_t)nseq;
            else { body[blen++] = (uint8_t)(128 + (nseq >> 8)); body[blen++] = (uint8_t)nseq; }
            body[blen++] = 0x10;                                             /* literal lengths predefined | offsets RLE | match lengths predefined */
            body[blen++] = 0;                                                /* the one offset code: 0 = repeat offset 1 (no extra bits) */
            bitw b = {body + blen, 0, 0};
            uint32_t sl = 0, sm = 0;
            for (size_t k = nseq; k-- > 0;) {
                const uint32_t ll = sq_lit[k] - (k ? sq_lit[k - 1] : 0u), mb = sq_ml[k] - 3u;
                const int lc = ll_code(ll), mc = ml_code(mb);
                if (k == nseq - 1) { sm = fse_first(&MLT, mc); sl = fse_first(&LLT, lc); }
                else { sm = fse_step(&MLT, sm, mc, &b); sl = fse_step(&LLT, sl, lc, &b); }   /* (the offset state has no bits) */
                bw_add(&b, ll & ((1u << LL_BITS[lc]) - 1), LL_BITS[lc]);
                bw_add(&b, mb & ((1u << ML_BITS[mc]) - 1), ML_BITS[mc]);
            }
            bw_add(&b, sm & 63u, 6);
            bw_add(&b, sl & 63u, 6);
            bw_close(&b);
            blen = (size_t)(b.p - body);
        }
        if (blen < n) type = 2;
    }
    const uint32_t bsize = type == 2 ? (uint32_t)blen : (uint32_t)n;
    const uint32_t bh = (uint32_t)(last ? 1 : 0) | ((uint32_t)type << 1) | (bsize << 3);
    out[0] = (uint8_t)bh; out[1] = (uint8_t)(bh >> 8); out[2] = (uint8_t)(bh >> 16);
    if (type == 2) { memcpy(out + 3, body, blen); return 3 + blen; }
    if (type == 1) { out[3] = in[0]; return 4; }
    memcpy(out + 3, in, n);
    return 3 + n;
}

/* the encoding tables of the predefined distributions (which: 0 literal lengths, 1 match lengths), for tests of the device's constants */
int s5o_zstd_seq_ctable(int which, uint16_t *next, int32_t *dnb, int32_t *dfs) {
    fse_ct ct;
    const int n = which ? 53 : 36;
    fse_ctable(&ct, which ? ML_NORM : LL_NORM, n);
    memcpy(next, ct.next, sizeof ct.next);
    memcpy(dnb, ct.dnb, (size_t)n * sizeof(int32_t));
    memcpy(dfs, ct.dfs, (size_t)n * sizeof(int32_t));
    return n;
}

size_t s5o_zstd_literals_bound(size_t n) { return n + 3 * (n / ZE_BLK + 1) + 16; }

/* literals-only Zstandard frame of in[0..n); out must hold s5o_zstd_literals_bound(n) bytes */
size_t s5o_zstd_literals_compress(const uint8_t *in, size_t n, uint8_t *out) {
    size_t o = 0;
    out[o++] = 0x28; out[o++] = 0xB5; out[o++] = 0x2F; out[o++] = 0xFD;
    out[o++] = 0xA0;                                  /* single segment, 4-byte content size, no checksum, no dictionary */
    for (int i = 0; i < 4; i++) out[o++] = (uint8_t)((uint64_t)n >> (8 * i));
    size_t done = 0;
    do {
        const size_t bl = n - done < ZE_BLK ? n - done : ZE_BLK;
        o += enc_block(in + done, bl, done + bl == n, out + o);
        done += bl;
    } while (done < n);
    return o;
}
