// inflate_simt_dev.h — zlib (RFC 1950/1951) decoder, ONE RECORD PER LANE.
//
// DEFLATE decoding is bit-serial inside a stream, and serial code costs a GPU wave ~7 cycles per instruction
// whether 1 or 64 lanes are active.  So the throughput design is SIMT across records: each lane of a wave64
// walks its own stream with its own tables (8-bit lookup + canonical fallback, 1.2 KiB of LDS per lane),
// and one VALU instruction advances 64 records.  Records of a batch have the same structure (header, keys,
// data), so the lanes stay largely convergent.  Same contract and status codes as inflate_dev.h (the
// wave-per-record decoder, kept for records this one hands back: none — any conforming stream is accepted).
#pragma once
#include "dev_common.h"
#include "inflate_dev.h"

namespace s5 {

constexpr int SL_LBITS = 8;
struct LaneTables {                    // per lane, in LDS
    uint16_t llut[1 << SL_LBITS];      // sym | len << 9 (0 = long code); doubles as the code-length array while tables are built
    uint16_t lsym[288];                // canonical order
    uint16_t lcount[16], dcount[16];
    uint16_t dsym[32];
    uint32_t lstate;                   // canonical-walk state after SL_LBITS bits: first << 16 | index (also makes the dword stride odd, 305)
};
static_assert(sizeof(LaneTables) == 512 + 576 + 64 + 64 + 4, "LaneTables layout");

struct LaneBits {
    const uint32_t *p;    // next aligned dword to fetch
    uint32_t ahead;       // dword already fetched (software prefetch: its HBM/L2 latency overlaps ~4 symbols of decoding)
    uint64_t buf;
    int cnt;
    uint64_t taken;       // bits loaded into buf so far
};
__device__ __forceinline__ void lb_init(LaneBits &b, const uint8_t *src) {
    b.buf = 0; b.cnt = 0; b.taken = 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(src);
    const int mis = (int)(a & 3);
    b.p = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    if (mis) {   // first partial dword
        const uint32_t w = *b.p++;
        b.buf = w >> (8 * mis);
        b.cnt = 32 - 8 * mis;
        b.taken = (uint64_t)b.cnt;
    }
    b.ahead = *b.p++;
}
__device__ __forceinline__ void lb_need32(LaneBits &b) {   // afterwards cnt >= 33 (reads may run <= 11 bytes past the stream)
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.ahead << b.cnt;
        b.ahead = *b.p++;
        b.cnt += 32;
        b.taken += 32;
    }
}
__device__ __forceinline__ uint32_t lb_get(LaneBits &b, int n) {
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
__device__ __forceinline__ uint64_t lb_consumed(const LaneBits &b) { return b.taken - (uint64_t)b.cnt; }

// canonical decode (count / syms), needs cnt >= 15
__device__ __forceinline__ int lane_slow(LaneBits &b, const uint16_t *count, const uint16_t *syms) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)lb_get(b, 1);
        const int c = count[len];
        if (code - c < first) return syms[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Build count/syms (+ LUT when lut != nullptr) from lens[0..n) (uint16 per length, any storage).  Returns nonzero if over-subscribed.
__device__ __forceinline__ int lane_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *syms) {
    for (int i = 0; i < 16; i++) count[i] = 0;
    for (int s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return 1; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; s++) { const int l = lens[s]; if (l) syms[offs[l]++] = (uint16_t)s; }
    return 0;
}
// canonical walk resumed after the LUT missed: the first SL_LBITS bits (LSB-first, already consumed) are `bits`
__device__ __forceinline__ int lane_slow_resume(LaneBits &b, const uint16_t *count, const uint16_t *syms, uint32_t bits, uint32_t state) {
    int code = (int)(__brev(bits) >> (32 - SL_LBITS)) << 1;
    int first = (int)(state >> 16), index = (int)(state & 0xFFFF);
    for (int len = SL_LBITS + 1; len <= 15; len++) {
        code |= (int)lb_get(b, 1);
        const int c = count[len];
        if (code - c < first) return syms[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// LUT from the canonical order: walking syms by increasing length enumerates the codes in increasing order
__device__ __forceinline__ void lane_fill_lut(const uint16_t *count, const uint16_t *syms, uint16_t *lut, int lutbits, int lenshift) {
    for (int i = 0; i < (1 << lutbits); i++) lut[i] = 0;
    uint32_t code = 0;
    int idx = 0;
    for (int l = 1; l <= lutbits; l++) {
        for (int k = 0; k < count[l]; k++, idx++, code++) {
            const uint32_t rev = __brev(code) >> (32 - l);
            const uint16_t ent = (uint16_t)(syms[idx] | (l << lenshift));
            for (uint32_t e = rev; e < (1u << lutbits); e += (1u << l)) lut[e] = ent;
        }
        code <<= 1;
    }
}

// Inflate one zlib stream with ONE lane.  `in` needs 16 readable bytes of padding after in + in_len.
__device__ __forceinline__ int zlib_inflate_lane(LaneTables &T, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap,
                                                 uint32_t *out_len) {
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    *out_len = 0;
    if (in_len < 6) return INF_ERR_TRUNC;
    {
        const uint32_t cmf = in[0], flg = in[1];
        if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return INF_ERR_HEADER;
    }
    const uint8_t *src = in + 2;
    const uint64_t total_bits = 8ull * (in_len - 6);
    LaneBits b;
    lb_init(b, src);
    uint32_t o = 0;
    uint32_t adA = 1, adB = 0;   // Adler-32 folded in as bytes are produced, reduced every 2048 bytes
    uint32_t since = 0;
    // Output goes out as aligned dwords: four bytes are collected in `pend` (the payload slot is 16-B aligned),
    // so a lane issues one scattered store per four symbols instead of one per symbol.  `lastb` keeps the most
    // recent byte in a register: distance-1 matches (all matches of our own encoder) never touch memory.
    uint32_t pend = 0, lastb = 0;
    auto put = [&](uint32_t x) {
        pend |= x << (8 * (o & 3));
        if ((o & 3) == 3) {
            if (o < cap) *reinterpret_cast<uint32_t *>(out + (o - 3)) = pend;
            else for (uint32_t q = o - 3; q <= o; q++) if (q < cap) out[q] = (uint8_t)(pend >> (8 * (q & 3)));
            pend = 0;
        }
        o++;
        lastb = x;
        adA += x;
        adB += adA;
        if (++since == 2048) { adA %= 65521u; adB %= 65521u; since = 0; }
    };
    auto flush_pending = [&]() {   // make bytes [o & ~3, o) visible in memory (before a match reads them back)
        for (uint32_t q = o & ~3u; q < o; q++) if (q < cap) out[q] = (uint8_t)(pend >> (8 * (q & 3)));
    };
    int last = 0;
    while (!last) {
        lb_need32(b);
        const uint32_t hdr = lb_get(b, 3);
        if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
        last = hdr & 1;
        const int type = hdr >> 1;
        if (type == 3) return INF_ERR_DATA;
        if (type == 0) {
            lb_get(b, b.cnt & 7);
            lb_need32(b);
            const uint32_t len = lb_get(b, 16), nlen = lb_get(b, 16);
            if ((len ^ 0xFFFFu) != nlen) return INF_ERR_DATA;
            const uint64_t pos = lb_consumed(b) >> 3;
            if (8 * pos > total_bits || pos + len > in_len - 6) return INF_ERR_TRUNC;
            for (uint32_t i = 0; i < len; i++) put(src[pos + i]);
            lb_init(b, src + pos + len);
            b.taken += 8 * (pos + len);
            continue;
        }
        int nl, nd;
        uint8_t *lens = reinterpret_cast<uint8_t *>(T.llut);   // the LUT storage (512 B) holds the <= 316 code lengths until the tables are built
        if (type == 1) {
            for (int s = 0; s < 288; s++) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            nl = 288;
            nd = 30;
            if (lane_build(lens, nl, T.lcount, T.lsym)) return INF_ERR_DATA;
            for (int s = 0; s < 30; s++) lens[s] = 5;
            if (lane_build(lens, nd, T.dcount, T.dsym)) return INF_ERR_DATA;
        } else {
            lb_need32(b);
            const uint32_t hd = lb_get(b, 14);
            nl = (int)(hd & 31) + 257;
            nd = (int)((hd >> 5) & 31) + 1;
            const int ncl = (int)(hd >> 10) + 4;
            if (nl > 286 || nd > 30) return INF_ERR_DATA;
            // code-length code: 19 lengths -> canonical tables in the distance table storage
            uint8_t *cl = reinterpret_cast<uint8_t *>(T.lsym);   // scratch for the 19 lengths (lsym is free until the lit/len tables are built)
            for (int i = 0; i < 19; i++) cl[i] = 0;
            {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (int i = 0; i < ncl; i++) { lb_need32(b); cl[order[i]] = (uint8_t)lb_get(b, 3); }
            }
            if (lane_build(cl, 19, T.dcount, T.dsym)) return INF_ERR_DATA;
            uint8_t prev = 0;
            int idx = 0;
            const int tot = nl + nd;
            while (idx < tot) {
                lb_need32(b);
                const int sym = lane_slow(b, T.dcount, T.dsym);
                if (sym < 0) return INF_ERR_DATA;
                if (sym < 16) { prev = (uint8_t)sym; lens[idx++] = prev; }
                else {
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) { if (idx == 0) return INF_ERR_DATA; v = prev; rep = 3 + (int)lb_get(b, 2); }
                    else if (sym == 17) rep = 3 + (int)lb_get(b, 3);
                    else rep = 11 + (int)lb_get(b, 7);
                    if (idx + rep > tot) return INF_ERR_DATA;
                    while (rep--) lens[idx++] = v;
                    if (sym != 16) prev = 0;
                }
            }
            if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
            if (lens[256] == 0) return INF_ERR_DATA;
            // distance tables first (their lengths sit behind the lit/len ones), then lit/len
            if (lane_build(lens + nl, nd, T.dcount, T.dsym)) return INF_ERR_DATA;
            if (lane_build(lens, nl, T.lcount, T.lsym)) return INF_ERR_DATA;
        }
        lane_fill_lut(T.lcount, T.lsym, T.llut, SL_LBITS, 9);   // overwrites the code lengths: no longer needed
        {
            uint32_t first = 0, index = 0;
            for (int l = 1; l <= SL_LBITS; l++) { index += T.lcount[l]; first += T.lcount[l]; first <<= 1; }
            T.lstate = (first << 16) | index;
        }
        const uint32_t lstate = T.lstate;
        for (;;) {
            lb_need32(b);
            int sym;
            const uint32_t e = T.llut[b.buf & ((1 << SL_LBITS) - 1)];
            if (e >> 9) { sym = e & 511; lb_get(b, e >> 9); }
            else { const uint32_t bits = lb_get(b, SL_LBITS); sym = lane_slow_resume(b, T.lcount, T.lsym, bits, lstate); }
            if (sym < 0) return INF_ERR_DATA;
            if (sym < 256) { put((uint32_t)sym); if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC; continue; }
            if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return INF_ERR_DATA;
            const uint32_t mlen = lbase[sym] + lb_get(b, lext[sym]);
            lb_need32(b);
            const int ds = lane_slow(b, T.dcount, T.dsym);
            if (ds < 0 || ds >= 30) return INF_ERR_DATA;
            const uint32_t mdist = dbase[ds] + lb_get(b, dext[ds]);
            if (mdist > o) return INF_ERR_DATA;
            if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
            if (mdist == 1) {
                const uint32_t x = lastb;
                for (uint32_t k = 0; k < mlen; k++) put(x);
            } else {
                for (uint32_t k = 0; k < mlen; k++) {
                    flush_pending();   // the source may be among the bytes still held in `pend`
                    const uint32_t sp = o - mdist;
                    put(sp < cap ? out[sp] : 0u);
                }
            }
        }
    }
    flush_pending();
    *out_len = o;
    adA %= 65521u; adB %= 65521u;
    const uint8_t *t = in + in_len - 4;
    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (o > cap) return INF_ERR_OVERFLOW;
    if (((adB << 16) | adA) != want) return INF_ERR_ADLER;
    return INF_OK;
}

}  // namespace s5
