// inflate_simt_dev.h — zlib (RFC 1950/1951) decoder, ONE RECORD PER LANE.
//
// DEFLATE decoding is bit-serial inside a stream, and serial code costs a GPU wave ~5 cycles per instruction
// whether 1 or 64 lanes are active.  So the throughput design is SIMT across records: each lane of a wave64
// walks its own stream with its own tables, and one VALU instruction advances 64 records.
//
// What bounds this kernel was measured, not guessed (tools/simt_probe.py, rocprofv3 PMC):
//   - throughput is linear in resident waves per CU (1 wave 4.2 M records/s, 2 waves 8.05 M): every wave sits alone on
//     its SIMD, and the per-lane tables in LDS decide how many waves fit;
//   - 64 DIFFERENT records per wave execute 2.58x the instructions of 64 identical ones (130 vs 50 VALU per symbol): the
//     lanes disagree on which path a symbol takes, above all on "lookup table hit" vs "walk the canonical code", and the
//     wave runs the union.  Input addresses being scattered costs nothing (copies of one record at 4096 addresses run as
//     fast as one address), nor does HBM latency (L2-hot input: no change).
// Hence this form: NO lookup table.  A symbol's length comes from comparing the next 15 bits (MSB-first) against the 15
// left-justified canonical limits, which live in registers — the same ~30 VALU instructions for every lane and every code
// length, no divergence — and the tables shrink from 1284 to 420 bytes of LDS per lane: symbols as bytes plus a 288-bit plane
// for the ninth bit; what only the table construction needs (code lengths as nibbles, counters) sits in private memory.
// Six waves per CU instead of two.
// Same contract and status codes as inflate_dev.h (the wave-per-record decoder).
#pragma once
#include "dev_common.h"
#include "inflate_dev.h"

namespace s5 {

struct LaneTables {                    // per lane, in LDS
    uint8_t lsym[288];                 // lit/len symbols in canonical order, low 8 bits
    uint32_t lhi[9];                   // ... and their bit 8 (length codes and end-of-block), one bit per entry
    int16_t ladj[16];                  // canonical index - first code, per length
};
static_assert(sizeof(LaneTables) == 356 && (sizeof(LaneTables) / 4) % 2 == 1, "LaneTables layout: odd dword stride spreads the lanes over the banks");

// What only the table construction needs lives in PRIVATE memory (scratch: per-lane, swizzled so that the 64 lanes' copies of one
// element are contiguous), not in LDS: the LDS footprint per lane is what decides how many waves a CU holds.
struct LaneBuild {
    uint8_t lens4[160];                // code lengths of the block header, 4 bits each (<= 316 of them)
    uint16_t tmp[16];                  // counts, then next free index per length
    uint16_t dcount[16];               // distance code (and, before it, the code-length code): counts per length.  Used once per
    uint8_t dsym[32];                  // match (and per header entry), not per symbol: private memory is good enough, and the 64 bytes
                                       // they took per lane in LDS were the difference between six and seven waves per CU
};

typedef short lane_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short lane_u2 __attribute__((ext_vector_type(2)));
// limit of length l = (first code of length l + codes of length l) << (15 - l), left-justified in 15 bits; kept minus one, two per
// register (lengths 2k+1 | 2k+2), so that sixteen 16-bit subtractions in eight packed instructions compare them all
struct LaneLimits { lane_s2 m1[8]; };

typedef const uint32_t __attribute__((address_space(1))) *lane_gptr32;   // input words: global_load, not flat (a flat access also waits on the LDS counter)
struct LaneBits {
    lane_gptr32 p;        // next aligned dword to fetch
    uint32_t ahead;       // dword already fetched (software prefetch: its HBM/L2 latency overlaps ~4 symbols of decoding)
    uint64_t buf;
    int cnt;
    uint64_t taken;       // bits loaded into buf so far
};
__device__ __forceinline__ void lb_init(LaneBits &b, const uint8_t *src) {
    b.buf = 0; b.cnt = 0; b.taken = 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(src);
    const int mis = (int)(a & 3);
    b.p = (lane_gptr32)(a & ~(uintptr_t)3);
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

__device__ __forceinline__ uint32_t nib_get(const uint8_t *a, int i) { return (a[i >> 1] >> ((i & 1) * 4)) & 15u; }
__device__ __forceinline__ void nib_set(uint8_t *a, int i, uint32_t v) {   // the other nibble of the byte is kept
    const uint32_t sh = (uint32_t)(i & 1) * 4;
    a[i >> 1] = (uint8_t)((a[i >> 1] & ~(15u << sh)) | (v << sh));
}

// canonical decode by walking the lengths (count / syms): the code-length code and the distance code (rare symbols).  cnt >= 15.
__device__ __forceinline__ int lane_slow(LaneBits &b, const uint16_t *count, const uint8_t *syms) {
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
// count / syms of a small alphabet (n <= 32) from lengths get(s).  Nonzero if over-subscribed.
template <typename F>
__device__ __forceinline__ int lane_build_small(F get, int n, uint16_t *count, uint8_t *syms) {
    for (int i = 0; i < 16; i++) count[i] = 0;
    for (int s = 0; s < n; s++) count[get(s)]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return 1; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; s++) { const int l = (int)get(s); if (l) syms[offs[l]++] = (uint8_t)s; }
    return 0;
}
// lit/len tables from the nibbles lens4[0..n): symbols in canonical order (T.lsym / T.lhi), T.ladj, and the limits.
__device__ __forceinline__ int lane_build_litlen(LaneTables &T, LaneBuild &B, int n, LaneLimits &lim) {
    for (int i = 0; i < 16; i++) B.tmp[i] = 0;
    for (int i = 0; i < 9; i++) T.lhi[i] = 0;
    for (int s = 0; s < n; s++) B.tmp[nib_get(B.lens4, s)]++;
    B.tmp[0] = 0;
    int left = 1;
    uint32_t first = 0, index = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        const uint32_t c = B.tmp[l];
        left = (left << 1) - (int)c;
        if (left < 0) return 1;
        {
            const short m1 = (short)(int)(((first + c) << (15 - l)) - 1u);
            if ((l - 1) & 1) lim.m1[(l - 1) >> 1].y = m1; else lim.m1[(l - 1) >> 1].x = m1;
        }
        T.ladj[l] = (int16_t)((int)index - (int)first);
        B.tmp[l] = (uint16_t)index;          // next free canonical index of this length
        index += c;
        first = (first + c) << 1;
    }
    for (int s = 0; s < n; s++) {
        const uint32_t l = nib_get(B.lens4, s);
        if (l) {
            const uint32_t at = B.tmp[l]++;
            T.lsym[at] = (uint8_t)s;
            if (s >= 256) T.lhi[at >> 5] |= 1u << (at & 31);
        }
    }
    return 0;
}
// one lit/len symbol: -1 = no such code.  cnt >= 15.
__device__ __forceinline__ int lane_litlen(LaneBits &b, const LaneTables &T, const LaneLimits &lim) {
    const uint32_t v = __brev((uint32_t)b.buf) >> 17;      // the next 15 bits, first bit of the code on top
    const lane_s2 vv = {(short)v, (short)v};
    lane_u2 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) acc += __builtin_bit_cast(lane_u2, (lane_s2)(lim.m1[k] - vv)) >> (unsigned short)15;   // sign bit: v >= limit
    const uint32_t len = 1u + acc.x + acc.y;
    if (len > 15) return -1;
    const uint32_t idx = (uint32_t)((int)T.ladj[len] + (int)(v >> (15 - len)));
    b.buf >>= len;
    b.cnt -= (int)len;
    return (int)((uint32_t)T.lsym[idx] | (((T.lhi[idx >> 5] >> (idx & 31)) & 1u) << 8));
}

// Inflate one zlib stream with ONE lane.  `in` needs 16 readable bytes of padding after in + in_len.
__device__ __forceinline__ int zlib_inflate_lane(LaneTables &T, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap,
                                                 uint32_t *out_len) {
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
    LaneBuild B;
    LaneLimits lim;
    lim.m1[7].y = 0x7FFF;                                          // there is no sixteenth length: never counted
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
        if (type == 1) {
            for (int s = 0; s < 288; s++) nib_set(B.lens4, s, s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u);
            nl = 288;
            nd = 30;
            if (lane_build_small([](int) { return 5u; }, nd, B.dcount, B.dsym)) return INF_ERR_DATA;
        } else {
            lb_need32(b);
            const uint32_t hd = lb_get(b, 14);
            nl = (int)(hd & 31) + 257;
            nd = (int)((hd >> 5) & 31) + 1;
            const int ncl = (int)(hd >> 10) + 4;
            if (nl > 286 || nd > 30) return INF_ERR_DATA;
            // code-length code: its 19 lengths are parked in the (still unused) symbol table, its canonical tables in the
            // distance tables' storage
            uint8_t *cl = T.lsym;
            for (int i = 0; i < 19; i++) cl[i] = 0;
            {
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (int i = 0; i < ncl; i++) { lb_need32(b); cl[order[i]] = (uint8_t)lb_get(b, 3); }
            }
            if (lane_build_small([&](int s) { return (uint32_t)cl[s]; }, 19, B.dcount, B.dsym)) return INF_ERR_DATA;
            uint32_t prev = 0;
            int idx = 0;
            const int tot = nl + nd;
            while (idx < tot) {
                lb_need32(b);
                const int sym = lane_slow(b, B.dcount, B.dsym);
                if (sym < 0) return INF_ERR_DATA;
                if (sym < 16) { prev = (uint32_t)sym; nib_set(B.lens4, idx++, prev); }
                else {
                    int rep;
                    uint32_t v = 0;
                    if (sym == 16) { if (idx == 0) return INF_ERR_DATA; v = prev; rep = 3 + (int)lb_get(b, 2); }
                    else if (sym == 17) rep = 3 + (int)lb_get(b, 3);
                    else rep = 11 + (int)lb_get(b, 7);
                    if (idx + rep > tot) return INF_ERR_DATA;
                    while (rep--) nib_set(B.lens4, idx++, v);
                    if (sym != 16) prev = 0;
                }
            }
            if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
            if (nib_get(B.lens4, 256) == 0) return INF_ERR_DATA;
            // distance tables (their lengths sit behind the lit/len ones)
            if (lane_build_small([&](int s) { return nib_get(B.lens4, nl + s); }, nd, B.dcount, B.dsym)) return INF_ERR_DATA;
        }
        if (lane_build_litlen(T, B, nl, lim)) return INF_ERR_DATA;
        for (;;) {
            lb_need32(b);
            int sym = lane_litlen(b, T, lim);
            if (sym < 0) return INF_ERR_DATA;
            if (sym < 256) { put((uint32_t)sym); if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC; continue; }
            if (lb_consumed(b) > total_bits) return INF_ERR_TRUNC;
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return INF_ERR_DATA;
            // length code: 3..10 one each, then 4 codes per extra-bit count, 258 on its own — arithmetic, no table in memory
            const uint32_t le = sym < 8 || sym == 28 ? 0u : (uint32_t)(sym >> 2) - 1u;
            const uint32_t mlen = (sym == 28 ? 258u : sym < 8 ? 3u + (uint32_t)sym : 3u + ((4u + ((uint32_t)sym & 3u)) << le)) + lb_get(b, (int)le);
            lb_need32(b);
            const int ds = lane_slow(b, B.dcount, B.dsym);
            if (ds < 0 || ds >= 30) return INF_ERR_DATA;
            const uint32_t de = ds < 4 ? 0u : (uint32_t)(ds >> 1) - 1u;
            const uint32_t mdist = (ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << de)) + lb_get(b, (int)de);
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
