// inflate_dev.h — zlib (RFC 1950/1951) stream decoder, one record per wave64.
//
// Replaces the record-depress stage of slow5lib reached through slow5_rec_depress_parse / slow5_get
// (/root/reference/src/view.c:38, src/get.c:45).  Accepts any conforming stream (stored / fixed /
// dynamic blocks, distances up to 32 KiB): the reference's fixtures were written by stock zlib.
// DEFLATE decoding is bit-serial per stream, so parallelism is across records: 64 lanes build the
// lookup tables cooperatively, lane 0 walks the bit stream, all lanes replicate matches.
#pragma once
#include "dev_common.h"

namespace s5 {

// same-wave LDS/HBM hand-off: memory ops of one wave execute in order; this only pins the compiler
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int INF_LBITS = 10;   // primary lit/len lookup bits
constexpr int INF_DBITS = 8;    // primary distance lookup bits

struct InflShared {              // per wave
    uint16_t llut[1 << INF_LBITS];   // sym | len << 9   (len == 0: long code -> canonical walk)
    uint16_t dlut[1 << INF_DBITS];   // sym | len << 5
    uint16_t lsym[288];              // canonical order symbols
    uint16_t dsym[32];
    uint16_t lcount[16], dcount[16];
    uint8_t lens[352];           // [0,19) code-length code | [32, 32+316) dynamic lit/len+dist; fixed: [0,288)+[288,320)
    uint32_t misc[4];
};

enum { INF_OK = 0, INF_ERR_HEADER = 1, INF_ERR_DATA = 2, INF_ERR_TRUNC = 3, INF_ERR_ADLER = 4, INF_ERR_OVERFLOW = 5 };

struct BitIn {
    const uint8_t *p, *end;
    uint64_t buf;
    int cnt;
    int over;   // bytes consumed past the end (error if any bit of them is used)
};
__device__ __forceinline__ void bi_refill(BitIn &b) {
    while (b.cnt <= 56) {
        uint64_t v = 0;
        if (b.p < b.end) v = *b.p; else b.over++;
        b.p++;
        b.buf |= v << b.cnt;
        b.cnt += 8;
    }
}
__device__ __forceinline__ uint32_t bi_get(BitIn &b, int n) {   // n <= 32, after refill
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// Build canonical tables + LUT for one alphabet from lens[0..n).  All 64 lanes of the wave.
// Returns 0 ok, nonzero if over-subscribed (incomplete codes are tolerated like zlib does for
// single-code distance trees; an invalid code simply never matches and reports INF_ERR_DATA).
__device__ __forceinline__ int infl_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *syms, uint16_t *lut,
                                          int lutbits, int lenshift) {
    const int lane = lane_id();
    if (lane < 16) count[lane] = 0;
    for (int i = lane; i < (1 << lutbits); i += 64) lut[i] = 0;
    wave_sync();
    // counts per length and canonical first codes (serial over 15 lengths, cheap)
    uint32_t next[16], offs[16];
    {
        uint32_t cnt[16];
#pragma unroll
        for (int b = 0; b < 16; b++) cnt[b] = 0;
        for (int base = 0; base < n; base += 64) {
            const int s = base + lane;
            const int l = s < n ? lens[s] : 0;
#pragma unroll
            for (int b = 1; b < 16; b++) cnt[b] += __popcll(__ballot(l == b));
        }
        uint32_t c = 0, o = 0;
        int left = 1;
        int bad = 0;
        next[0] = 0; offs[0] = 0;
#pragma unroll
        for (int b = 1; b < 16; b++) {
            c = (c + cnt[b - 1] * (b > 1)) << 1;
            next[b] = c;
            offs[b] = o;
            o += cnt[b];
            left = (left << 1) - (int)cnt[b];
            if (left < 0) bad = 1;
            if (lane == 0) count[b] = (uint16_t)cnt[b];
        }
        if (bad) return 1;
    }
    const uint64_t lt = (1ull << lane) - 1;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const int l = s < n ? lens[s] : 0;
        uint32_t code = 0, idx = 0;
#pragma unroll
        for (int b = 1; b < 16; b++) {
            const uint64_t mask = __ballot(l == b);
            const uint32_t r = __popcll(mask & lt);
            if (l == b) { code = next[b] + r; idx = offs[b] + r; }
            next[b] += __popcll(mask);
            offs[b] += __popcll(mask);
        }
        if (l) {
            syms[idx] = (uint16_t)s;
            if (l <= lutbits) {
                const uint32_t rev = __brev(code) >> (32 - l);
                const uint16_t ent = (uint16_t)(s | (l << lenshift));
                for (uint32_t k = rev; k < (1u << lutbits); k += (1u << l)) lut[k] = ent;
            }
        }
    }
    wave_sync();
    return 0;
}

// canonical bit-by-bit walk for codes longer than the LUT (puff-style); returns symbol or -1
__device__ __forceinline__ int infl_slow(BitIn &b, const uint16_t *count, const uint16_t *syms) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)bi_get(b, 1);
        const int c = count[len];
        if (code - c < first) return syms[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Inflate one zlib stream.  out may be HBM.  cap = bytes available at out.  Returns status; *out_len =
// decoded length (also when INF_ERR_OVERFLOW: the size needed, nothing beyond cap is written).
__device__ __forceinline__ int zlib_inflate_wave(InflShared &T, const uint8_t *in, uint32_t in_len, uint8_t *out,
                                                 uint32_t cap, uint32_t *out_len) {
    const int lane = lane_id();
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
    BitIn b;
    b.p = in + 2;
    b.end = in + in_len - 4;   // the Adler-32 trailer is not part of the deflate data
    b.buf = 0;
    b.cnt = 0;
    b.over = 0;
    uint32_t o = 0;          // bytes produced (uniform: broadcast from lane 0 after each block step)
    int status = INF_OK;
    int last = 0;
    while (!last && status == INF_OK) {
        // ---- block header (lane 0 reads, broadcast) ----
        uint32_t hdr = 0;
        if (lane == 0) { bi_refill(b); hdr = bi_get(b, 3); if (b.over > 8) hdr = 8; }
        hdr = __shfl(hdr, 0);
        if (hdr == 8) { status = INF_ERR_TRUNC; break; }   // ran off the end of the input
        last = hdr & 1;
        const int type = hdr >> 1;
        if (type == 3) { status = INF_ERR_DATA; break; }
        if (type == 0) {
            // stored: align to byte, LEN/NLEN, raw copy by all lanes
            uint32_t len = 0, src_off = 0, bad = 0;
            if (lane == 0) {
                bi_get(b, b.cnt & 7);
                bi_refill(b);
                len = bi_get(b, 16);
                const uint32_t nlen = bi_get(b, 16);
                bad = (len ^ 0xFFFFu) != nlen;
                // un-read the whole bytes still buffered so p points at the data
                b.p -= b.cnt >> 3;
                if (b.over) { const int back = min(b.over, b.cnt >> 3); b.over -= back; }
                b.buf = 0;
                b.cnt = 0;
                src_off = (uint32_t)(b.p - in);
                if (b.p + len > b.end) bad |= 2;
                b.p += len;
            }
            len = __shfl(len, 0);
            src_off = __shfl(src_off, 0);
            bad = __shfl(bad, 0);
            if (bad) { status = (bad & 2) ? INF_ERR_TRUNC : INF_ERR_DATA; break; }
            for (uint32_t i = lane; i < len; i += 64)
                if (o + i < cap) out[o + i] = in[src_off + i];
            o += len;
            continue;
        }
        // ---- code lengths ----
        int nl, nd;
        if (type == 1) {
            for (int s = lane; s < 288; s += 64) T.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            if (lane < 32) T.lens[288 + lane] = 5;
            nl = 288;
            nd = 30;
            wave_sync();
        } else {
            uint32_t hd = 0;
            if (lane == 0) { bi_refill(b); hd = bi_get(b, 14); }
            hd = __shfl(hd, 0);
            nl = (int)(hd & 31) + 257;
            nd = (int)((hd >> 5) & 31) + 1;
            const int ncl = (int)(hd >> 10) + 4;
            if (nl > 286 || nd > 30) { status = INF_ERR_DATA; break; }
            if (lane < 19) T.lens[lane] = 0;
            wave_sync();
            if (lane == 0) {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (int i = 0; i < ncl; i++) { bi_refill(b); T.lens[order[i]] = (uint8_t)bi_get(b, 3); }
            }
            wave_sync();
            // the code-length code reuses the distance tables' storage (built before the real ones)
            if (infl_build(T.lens, 19, T.dcount, T.dsym, T.dlut, 7, 5)) { status = INF_ERR_DATA; break; }
            int bad = 0;
            if (lane == 0) {
                uint8_t tmp_prev = 0;
                int idx = 0;
                const int tot = nl + nd;
                while (idx < tot && !bad) {
                    bi_refill(b);
                    const uint32_t e = T.dlut[b.buf & 127];
                    int sym;
                    if (e >> 5) { sym = e & 31; bi_get(b, e >> 5); } else { sym = -1; bad = 1; break; }
                    if (sym < 16) { tmp_prev = (uint8_t)sym; T.lens[32 + idx++] = tmp_prev; }
                    else {
                        int rep; uint8_t v = 0;
                        if (sym == 16) { if (idx == 0) { bad = 1; break; } v = tmp_prev; rep = 3 + (int)bi_get(b, 2); }
                        else if (sym == 17) rep = 3 + (int)bi_get(b, 3);
                        else rep = 11 + (int)bi_get(b, 7);
                        if (idx + rep > tot) { bad = 1; break; }
                        while (rep--) T.lens[32 + idx++] = v;
                        if (sym != 16) tmp_prev = 0;
                    }
                }
            }
            bad = __shfl(bad, 0);
            if (bad) { status = INF_ERR_DATA; break; }
            wave_sync();
        }
        // tables: dynamic lengths sit at T.lens[32 ..] (lit/len then dist); fixed at [0..288) + [288..)
        const uint8_t *ll = type == 1 ? T.lens : T.lens + 32;
        const uint8_t *dl = type == 1 ? T.lens + 288 : T.lens + 32 + nl;
        if (type == 2 && ll[256] == 0) { status = INF_ERR_DATA; break; }
        if (infl_build(ll, nl, T.lcount, T.lsym, T.llut, INF_LBITS, 9)) { status = INF_ERR_DATA; break; }
        if (infl_build(dl, nd, T.dcount, T.dsym, T.dlut, INF_DBITS, 5)) { status = INF_ERR_DATA; break; }
        // ---- symbols: lane 0 decodes, the wave replicates matches ----
        for (;;) {
            // lane 0 decodes up to the next match or end of block, writing literals itself
            uint32_t mlen = 0, mdist = 0, st = 0;   // st: 0 match, 1 end of block, 2 error, 3 truncated
            if (lane == 0) {
                for (;;) {
                    bi_refill(b);
                    if (b.over > 8) { st = 3; break; }   // zero-fill past the end must not decode forever
                    int sym;
                    const uint32_t e = T.llut[b.buf & ((1 << INF_LBITS) - 1)];
                    if (e >> 9) { sym = e & 511; bi_get(b, e >> 9); } else sym = infl_slow(b, T.lcount, T.lsym);
                    if (sym < 0) { st = 2; break; }
                    if (sym < 256) { if (o < cap) out[o] = (uint8_t)sym; o++; continue; }
                    if (sym == 256) { st = 1; break; }
                    sym -= 257;
                    if (sym >= 29) { st = 2; break; }
                    mlen = lbase[sym] + bi_get(b, lext[sym]);
                    bi_refill(b);
                    int ds;
                    const uint32_t de = T.dlut[b.buf & ((1 << INF_DBITS) - 1)];
                    if (de >> 5) { ds = de & 31; bi_get(b, de >> 5); } else ds = infl_slow(b, T.dcount, T.dsym);
                    if (ds < 0 || ds >= 30) { st = 2; break; }
                    mdist = dbase[ds] + bi_get(b, dext[ds]);
                    if (mdist > o) { st = 2; break; }
                    break;
                }
            }
            st = __shfl(st, 0);
            o = __shfl(o, 0);
            if (st >= 2) { status = st == 2 ? INF_ERR_DATA : INF_ERR_TRUNC; break; }
            if (st == 1) break;
            mlen = __shfl(mlen, 0);
            mdist = __shfl(mdist, 0);
            // out[o+k] = out[o-dist + k mod dist]: every source byte precedes o, so lanes are independent
            for (uint32_t k = lane; k < mlen; k += 64)
                if (o + k < cap) out[o + k] = out[o - mdist + (k % mdist)];
            o += mlen;
        }
    }
    // trailer
    int over = __shfl(b.over, 0);
    int cntbits = __shfl(b.cnt, 0);
    if (status == INF_OK) {
        // bits of bytes past the end must be unused: consumed bytes = (p - in) - cnt/8 <= in_len - 4
        if (over * 8 > cntbits) status = INF_ERR_TRUNC;
    }
    *out_len = o;
    if (status == INF_OK && o > cap) status = INF_ERR_OVERFLOW;
    return status;
}

}  // namespace s5
