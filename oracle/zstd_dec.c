/*
 * zstd_dec.c — CPU ORACLE: a plain-C restatement of the Zstandard frame decoder (RFC 8878).  TEST INFRASTRUCTURE ONLY.
 *
 * slow5lib's third record press is zstd (SLOW5_COMPRESS_ZSTD, /root/reference/src/misc.c:259; built only with `zstd=1`,
 * /root/reference/Makefile:101-140), i.e. libzstd's ZSTD_compress / ZSTD_decompress on the record payload (level 1: the
 * fixture test/data/exp/one_fast5/exp_1_lossless_zstd_v0.2.0.blow5 is byte-identical to ZSTD_compress(payload, 1) of
 * libzstd 1.4.8).  This file restates the DECODER of the published format so that the device decoder (csrc/zstd_dev.h) has a
 * line-by-line twin that runs on the CPU; it is pinned against the real libzstd (dlopen'ed by tests/oracle_bind.py) on the
 * reference's zstd fixtures and on thousands of generated frames (tests/test_oracle_golden.py::test_zstd_*).
 *
 * Supported: single / multi block frames, raw / RLE / compressed blocks, all four literal modes (1 and 4 streams, direct and
 * FSE-compressed Huffman weights, treeless), all sequence modes (predefined / RLE / FSE / repeat), repeat offsets, optional
 * content checksum (XXH64: verified when the frame carries one), skippable frames are rejected, dictionaries are rejected.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "s5oracle.h"

typedef struct { uint8_t sym, nb; uint16_t base; } fse_ent;

typedef struct {            /* backward bit reader */
    const uint8_t *p;       /* start of the stream */
    int64_t bits;           /* bits not yet consumed (position of the next bit to read, counted from the start) */
} brd;

static int highbit(uint32_t v) { int n = -1; while (v) { v >>= 1; n++; } return n; }

static int brd_init(brd *b, const uint8_t *p, size_t len) {
    if (len == 0 || p[len - 1] == 0) return -1;
    b->p = p;
    b->bits = (int64_t)(len - 1) * 8 + highbit(p[len - 1]);   /* the bits below the final marker bit */
    return 0;
}
/* read n (<= 32) bits; bits beyond the start of the stream read as zero (and push `bits` negative: the caller checks) */
static uint32_t brd_read(brd *b, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        b->bits--;
        uint32_t bit = 0;
        if (b->bits >= 0) bit = (b->p[b->bits >> 3] >> (b->bits & 7)) & 1;
        v = (v << 1) | bit;
    }
    return v;
}

/* forward bit reader for FSE table descriptions */
typedef struct { const uint8_t *p; size_t len; uint64_t pos; } frd;
static uint32_t frd_peek(const frd *f, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        const uint64_t q = f->pos + (uint64_t)i;
        if ((q >> 3) < f->len) v |= (uint32_t)((f->p[q >> 3] >> (q & 7)) & 1) << i;
    }
    return v;
}

/* FSE_readNCount: returns bytes consumed (0 on error); norm[] gets the counts (-1 = "less than one"), *maxsym, *log */
static size_t fse_read_ncount(const uint8_t *p, size_t len, int16_t *norm, int *maxsym, int *log, int max_log, int max_sym) {
    frd f = {p, len, 0};
    const int al = (int)frd_peek(&f, 4) + 5;
    f.pos += 4;
    if (al > max_log) return 0;
    *log = al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            int n0 = 0;
            for (;;) {
                const int rep = (int)frd_peek(&f, 2);
                f.pos += 2;
                n0 += rep;
                if (rep != 3) break;
            }
            while (n0-- > 0) { if (sym > max_sym) return 0; norm[sym++] = 0; }
            prev0 = 0;
            if (sym > max_sym) break;
        }
        const int maxv = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t bits = frd_peek(&f, nbits);
        if ((int)(bits & (uint32_t)(threshold - 1)) < maxv) {
            count = (int)(bits & (uint32_t)(threshold - 1));
            f.pos += (uint64_t)(nbits - 1);
        } else {
            count = (int)(bits & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= maxv;
            f.pos += (uint64_t)nbits;
        }
        count--;                                   /* -1: probability "less than one" */
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = count == 0;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    if (remaining != 1 || sym > max_sym + 1) return 0;
    *maxsym = sym - 1;
    const size_t used = (size_t)((f.pos + 7) >> 3);
    return used <= len ? used : 0;
}

static int fse_build(fse_ent *t, const int16_t *norm, int maxsym, int log, const uint16_t *base, const uint8_t *extra) {
    const int size = 1 << log;
    uint16_t next[64];
    uint8_t cell[512];
    int high = size - 1;
    if (maxsym >= 64) return -1;
    for (int s = 0; s <= maxsym; s++) {
        if (norm[s] == -1) { cell[high--] = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s <= maxsym; s++)
        for (int i = 0; i < norm[s]; i++) {
            cell[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        const int s = cell[i];
        const uint32_t ns = next[s]++;
        const int nb = log - highbit(ns);
        t[i].sym = (uint8_t)s;
        t[i].nb = (uint8_t)nb;
        t[i].base = (uint16_t)((ns << nb) - (uint32_t)size);
    }
    (void)base; (void)extra;
    return 0;
}

typedef struct { uint8_t sym, nb; } huf_ent;
typedef struct {
    huf_ent huf[1 << 11];
    int huf_log;                      /* 0: no table yet */
    fse_ent ll[512], of[256], ml[512];
    int ll_log, of_log, ml_log;       /* -1: no table yet */
    uint32_t rep[3];
} zctx;

static const int16_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* Huffman table from the tree description at p; returns bytes consumed, 0 on error */
static size_t huf_read_table(zctx *z, const uint8_t *p, size_t len) {
    uint8_t w[256];
    int nsym;
    size_t used;
    if (len < 1) return 0;
    const int hb = p[0];
    if (hb >= 128) {
        nsym = hb - 127;
        used = 1 + (size_t)(nsym + 1) / 2;
        if (used > len) return 0;
        for (int i = 0; i < nsym; i++) w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    } else {
        used = 1 + (size_t)hb;
        if (used > len || hb < 1) return 0;
        int16_t norm[16];
        int maxsym, log;
        const size_t h = fse_read_ncount(p + 1, (size_t)hb, norm, &maxsym, &log, 6, 12);
        if (!h || h >= (size_t)hb) return 0;
        fse_ent t[64];
        if (fse_build(t, norm, maxsym, log, NULL, NULL)) return 0;
        brd b;
        if (brd_init(&b, p + 1 + h, (size_t)hb - h)) return 0;
        uint32_t s1 = brd_read(&b, log), s2 = brd_read(&b, log);
        nsym = 0;
        for (;;) {                                 /* two interleaved states, RFC 8878 4.2.1.2 */
            if (nsym >= 255) return 0;
            w[nsym++] = t[s1].sym;
            if (b.bits < t[s1].nb) { w[nsym++] = t[s2].sym; break; }
            s1 = t[s1].base + brd_read(&b, t[s1].nb);
            if (nsym >= 255) return 0;
            w[nsym++] = t[s2].sym;
            if (b.bits < t[s2].nb) { w[nsym++] = t[s1].sym; break; }
            s2 = t[s2].base + brd_read(&b, t[s2].nb);
        }
        if (nsym > 255) return 0;
    }
    /* the last weight is implied: the sum of 2^(w-1) must reach a power of two */
    uint32_t sum = 0;
    for (int i = 0; i < nsym; i++) { if (w[i] > 11) return 0; if (w[i]) sum += 1u << (w[i] - 1); }
    if (sum == 0) return 0;
    const int maxbits = highbit(sum) + 1;
    if (maxbits > 11) return 0;
    const uint32_t left = (1u << maxbits) - sum;
    if (left & (left - 1)) return 0;               /* must be a power of two */
    w[nsym++] = (uint8_t)(highbit(left) + 1);
    uint32_t rank_start[13], cnt[13];
    memset(cnt, 0, sizeof cnt);
    for (int i = 0; i < nsym; i++) cnt[w[i]]++;
    if (cnt[1] < 2 || (cnt[1] & 1)) return 0;
    uint32_t at = 0;
    for (int r = 1; r <= maxbits; r++) { rank_start[r] = at; at += cnt[r] << (r - 1); }
    for (int i = 0; i < nsym; i++) {
        const int r = w[i];
        if (!r) continue;
        const uint32_t span = 1u << (r - 1);
        for (uint32_t k = 0; k < span; k++) { z->huf[rank_start[r] + k].sym = (uint8_t)i; z->huf[rank_start[r] + k].nb = (uint8_t)(maxbits + 1 - r); }
        rank_start[r] += span;
    }
    z->huf_log = maxbits;
    return used;
}

static int huf_stream(const zctx *z, const uint8_t *p, size_t len, uint8_t *out, size_t n) {
    brd b;
    if (brd_init(&b, p, len)) return -1;
    const int L = z->huf_log;
    for (size_t i = 0; i < n; i++) {
        /* peek L bits (zero padded past the start), consume nb */
        brd t = b;
        const uint32_t idx = brd_read(&t, L);
        const huf_ent e = z->huf[idx];
        b.bits -= e.nb;
        if (b.bits < 0) return -1;
        out[i] = e.sym;
    }
    return b.bits == 0 ? 0 : -1;
}

/* one of the three sequence tables; returns bytes consumed by its description (0 allowed), -1 on error */
static int64_t seq_table(int mode, const uint8_t *p, size_t len, fse_ent *t, int *log, const int16_t *def, int def_n, int def_log,
                         int max_log, int max_sym) {
    int16_t norm[64];
    if (mode == 0) { if (fse_build(t, def, def_n - 1, def_log, NULL, NULL)) return -1; *log = def_log; return 0; }
    if (mode == 1) { if (len < 1 || p[0] > max_sym) return -1; t[0].sym = p[0]; t[0].nb = 0; t[0].base = 0; *log = 0; return 1; }
    if (mode == 2) {
        int maxsym, l;
        const size_t h = fse_read_ncount(p, len, norm, &maxsym, &l, max_log, max_sym);
        if (!h) return -1;
        if (fse_build(t, norm, maxsym, l, NULL, NULL)) return -1;
        *log = l;
        return (int64_t)h;
    }
    return *log < 0 ? -1 : 0;                      /* repeat: the previous table must exist */
}

/* returns the decompressed size, or (size_t)-1 on error.  out must hold cap bytes. */
/* decode cells of a predefined distribution (0 literal lengths, 1 offsets, 2 match lengths; RFC 8878 3.1.1.3.2.2 and appendix A), packed as
 * the device keeps them in csrc/zstd_seq_tables.h: symbol | (bits | baseline << 4) << 8; returns the number of cells (0: no such table) */
int s5o_zstd_seq_dtable(int which, uint32_t *cells) {
    fse_ent t[64];
    const int16_t *def = which == 0 ? LL_DEF : which == 1 ? OF_DEF : which == 2 ? ML_DEF : NULL;
    const int n = which == 0 ? 36 : which == 1 ? 29 : 53, log = which == 1 ? 5 : 6;
    if (!def || fse_build(t, def, n - 1, log, NULL, NULL)) return 0;
    for (int i = 0; i < (1 << log); i++) cells[i] = (uint32_t)t[i].sym | (((uint32_t)t[i].nb | (((uint32_t)t[i].base & 0xFFFu) << 4)) << 8);
    return 1 << log;
}

/* XXH64, seed 0 (the published algorithm: four accumulators over 32-byte stripes, merge, tail, avalanche) */
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t xxh64(const uint8_t *p, size_t n) {
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    uint64_t h;
    size_t i = 0;
    if (n >= 32) {
        uint64_t v[4] = {P1 + P2, P2, 0, 0 - P1};
        for (; i + 32 <= n; i += 32)
            for (int k = 0; k < 4; k++) v[k] = rotl64(v[k] + rd64(p + i + 8 * k) * P2, 31) * P1;
        h = rotl64(v[0], 1) + rotl64(v[1], 7) + rotl64(v[2], 12) + rotl64(v[3], 18);
        for (int k = 0; k < 4; k++) h = (h ^ (rotl64(v[k] * P2, 31) * P1)) * P1 + P4;
    } else h = P5;
    h += n;
    for (; i + 8 <= n; i += 8) h = rotl64(h ^ (rotl64(rd64(p + i) * P2, 31) * P1), 27) * P1 + P4;
    if (i + 4 <= n) { h = rotl64(h ^ ((uint64_t)rd32(p + i) * P1), 23) * P2 + P3; i += 4; }
    for (; i < n; i++) h = rotl64(h ^ ((uint64_t)p[i] * P5), 11) * P1;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

size_t s5o_zstd_restated_decompress(const uint8_t *in, size_t len, uint8_t *out, size_t cap) {
    const size_t ERR = (size_t)-1;
    if (len < 6 || in[0] != 0x28 || in[1] != 0xB5 || in[2] != 0x2F || in[3] != 0xFD) return ERR;
    size_t p = 4;
    const int fhd = in[p++];
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 8) return ERR;
    if (did) return ERR;                            /* no dictionaries */
    if (!single) { if (p >= len) return ERR; p++; } /* window descriptor: the whole output is addressable here */
    const int fcs_bytes = fcs_flag == 0 ? single : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
    if (p + (size_t)fcs_bytes > len) return ERR;
    uint64_t fcs = 0;
    for (int i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)in[p + i] << (8 * i);
    if (fcs_flag == 1) fcs += 256;
    p += (size_t)fcs_bytes;
    zctx *z = (zctx *)malloc(sizeof *z);
    uint8_t *lit = (uint8_t *)malloc(128 * 1024 + 32);
    if (!z || !lit) { free(z); free(lit); return ERR; }
    z->huf_log = 0; z->ll_log = z->of_log = z->ml_log = -1;
    z->rep[0] = 1; z->rep[1] = 4; z->rep[2] = 8;
    size_t o = 0;
    int last = 0, bad = 0;
    while (!last && !bad) {
        if (p + 3 > len) { bad = 1; break; }
        const uint32_t bh = in[p] | (in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
        p += 3;
        last = bh & 1;
        const int type = (bh >> 1) & 3;
        const uint32_t bsize = bh >> 3;
        if (type == 3 || bsize > 128 * 1024) { bad = 1; break; }
        if (type == 0) {
            if (p + bsize > len || o + bsize > cap) { bad = 1; break; }
            memcpy(out + o, in + p, bsize); o += bsize; p += bsize;
            continue;
        }
        if (type == 1) {
            if (p + 1 > len || o + bsize > cap) { bad = 1; break; }
            memset(out + o, in[p], bsize); o += bsize; p += 1;
            continue;
        }
        if (p + bsize > len || bsize < 2) { bad = 1; break; }
        const uint8_t *b = in + p, *bend = b + bsize;
        p += bsize;
        /* ---- literals ---- */
        const int ltype = b[0] & 3, sf = (b[0] >> 2) & 3;
        size_t lsize, csize = 0, hl;
        int streams = 1;
        if (ltype < 2) {
            if (sf == 0 || sf == 2) { lsize = b[0] >> 3; hl = 1; }
            else if (sf == 1) { lsize = (b[0] >> 4) | ((size_t)b[1] << 4); hl = 2; }
            else { if (bsize < 3) { bad = 1; break; } lsize = (b[0] >> 4) | ((size_t)b[1] << 4) | ((size_t)b[2] << 12); hl = 3; }
        } else {
            if (bsize < 5) { bad = 1; break; }
            const uint64_t v = (uint64_t)b[0] | ((uint64_t)b[1] << 8) | ((uint64_t)b[2] << 16) | ((uint64_t)b[3] << 24) | ((uint64_t)b[4] << 32);
            if (sf == 0 || sf == 1) { lsize = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; hl = 3; streams = sf ? 4 : 1; }
            else if (sf == 2) { lsize = (v >> 4) & 0x3FFF; csize = (v >> 18) & 0x3FFF; hl = 4; streams = 4; }
            else { lsize = (v >> 4) & 0x3FFFF; csize = (v >> 22) & 0x3FFFF; hl = 5; streams = 4; }
        }
        if (lsize > 128 * 1024) { bad = 1; break; }
        const uint8_t *q = b + hl;
        if (ltype == 0) { if (q + lsize > bend) { bad = 1; break; } memcpy(lit, q, lsize); q += lsize; }
        else if (ltype == 1) { if (q + 1 > bend) { bad = 1; break; } memset(lit, q[0], lsize); q += 1; }
        else {
            if (q + csize > bend) { bad = 1; break; }
            const uint8_t *c = q, *cend = q + csize;
            q = cend;
            if (ltype == 2) {
                const size_t u = huf_read_table(z, c, (size_t)(cend - c));
                if (!u) { bad = 1; break; }
                c += u;
            } else if (!z->huf_log) { bad = 1; break; }
            if (streams == 1) { if (huf_stream(z, c, (size_t)(cend - c), lit, lsize)) { bad = 1; break; } }
            else {
                if (cend - c < 6) { bad = 1; break; }
                const size_t s1 = c[0] | (c[1] << 8), s2 = c[2] | (c[3] << 8), s3 = c[4] | (c[5] << 8);
                c += 6;
                if (s1 + s2 + s3 > (size_t)(cend - c)) { bad = 1; break; }
                const size_t s4 = (size_t)(cend - c) - s1 - s2 - s3, per = (lsize + 3) / 4;
                if (3 * per > lsize) { bad = 1; break; }
                if (huf_stream(z, c, s1, lit, per) || huf_stream(z, c + s1, s2, lit + per, per) ||
                    huf_stream(z, c + s1 + s2, s3, lit + 2 * per, per) || huf_stream(z, c + s1 + s2 + s3, s4, lit + 3 * per, lsize - 3 * per)) { bad = 1; break; }
            }
        }
        /* ---- sequences ---- */
        if (q >= bend) { bad = 1; break; }
        uint32_t nseq = *q++;
        if (nseq >= 128) {
            if (nseq == 255) { if (q + 2 > bend) { bad = 1; break; } nseq = q[0] + (q[1] << 8) + 0x7F00; q += 2; }
            else { if (q + 1 > bend) { bad = 1; break; } nseq = ((nseq - 128) << 8) + q[0]; q += 1; }
        }
        size_t li = 0;
        if (nseq) {
            if (q >= bend) { bad = 1; break; }
            const int modes = *q++;
            if (modes & 3) { bad = 1; break; }
            int64_t u;
            if ((u = seq_table(modes >> 6, q, (size_t)(bend - q), z->ll, &z->ll_log, LL_DEF, 36, 6, 9, 35)) < 0) { bad = 1; break; }
            q += u;
            if ((u = seq_table((modes >> 4) & 3, q, (size_t)(bend - q), z->of, &z->of_log, OF_DEF, 29, 5, 8, 31)) < 0) { bad = 1; break; }
            q += u;
            if ((u = seq_table((modes >> 2) & 3, q, (size_t)(bend - q), z->ml, &z->ml_log, ML_DEF, 53, 6, 9, 52)) < 0) { bad = 1; break; }
            q += u;
            brd br;
            if (brd_init(&br, q, (size_t)(bend - q))) { bad = 1; break; }
            uint32_t sl = brd_read(&br, z->ll_log), so = brd_read(&br, z->of_log), sm = brd_read(&br, z->ml_log);
            for (uint32_t s = 0; s < nseq && !bad; s++) {
                const int ofc = z->of[so].sym, mlc = z->ml[sm].sym, llc = z->ll[sl].sym;
                if (ofc > 31 || mlc > 52 || llc > 35) { bad = 1; break; }
                const uint32_t ofv = (1u << ofc) + brd_read(&br, ofc);
                const uint32_t mlen = ML_BASE[mlc] + brd_read(&br, ML_BITS[mlc]);
                const uint32_t llen = LL_BASE[llc] + brd_read(&br, LL_BITS[llc]);
                uint32_t offset;
                if (ofv > 3) { offset = ofv - 3; z->rep[2] = z->rep[1]; z->rep[1] = z->rep[0]; z->rep[0] = offset; }
                else {
                    uint32_t idx = ofv - 1 + (llen == 0);
                    if (idx == 0) offset = z->rep[0];
                    else {
                        offset = idx < 3 ? z->rep[idx] : z->rep[0] - 1;
                        if (idx > 1) z->rep[2] = z->rep[1];
                        z->rep[1] = z->rep[0];
                        z->rep[0] = offset;
                    }
                }
                if (s + 1 < nseq) {                   /* state updates: LL, ML, OF */
                    sl = z->ll[sl].base + brd_read(&br, z->ll[sl].nb);
                    sm = z->ml[sm].base + brd_read(&br, z->ml[sm].nb);
                    so = z->of[so].base + brd_read(&br, z->of[so].nb);
                }
                if (br.bits < 0 || li + llen > lsize || o + llen + mlen > cap || offset == 0 || offset > o + llen) { bad = 1; break; }
                memcpy(out + o, lit + li, llen); o += llen; li += llen;
                for (uint32_t k = 0; k < mlen; k++) out[o + k] = out[o + k - offset];
                o += mlen;
            }
            if (!bad && br.bits != 0) bad = 1;
        }
        if (bad) break;
        if (o + (lsize - li) > cap) { bad = 1; break; }
        memcpy(out + o, lit + li, lsize - li); o += lsize - li;
    }
    if (!bad && checksum) {   /* the low 32 bits of XXH64(content, seed 0), little-endian (RFC 8878 3.1.1) */
        if (p + 4 > len) bad = 1;
        else {
            const uint32_t want = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24);
            if ((uint32_t)xxh64(out, o) != want) bad = 1;
            p += 4;
        }
    }
    if (!bad && p != len) bad = 1;
    if (!bad && fcs_bytes && fcs != o) bad = 1;
    free(z); free(lit);
    return bad ? ERR : o;
}
