// zstd_enc_dev.h — Zstandard frames on the device, encode side (the zstd record press).
//
// slow5lib compresses a record with ZSTD_compress(level 1) (/root/reference/src/misc.c:259 names the method).  libzstd's
// match finder is a serial hash chain over the record; like the DEFLATE side this encoder does not reproduce its bytes but
// writes a VALID frame that libzstd decompresses to the identical payload.  The frame is "literals only": blocks of at most
// 16 KiB (the LDS stage of the staged path), each raw, RLE, or compressed = Huffman literals in 4 streams + an empty
// sequences section.  On nanopore records that is within ~2 % of libzstd level 1 (75 370 B svb-zd payload: 51 047 B here,
// 49 895 B libzstd 1.4.8, 50 766 B zlib): the svb-zd bytes hold little for a match finder to find.
//
// One record per 256-thread workgroup, one block at a time:
//   histogram     4 per-wave sub-histograms in the (still dead) build scratch, summed into S.freq;
//   code lengths  build_lengths<> of the DEFLATE side, capped at 11 bits (deflate_dev.h);
//   wave 0        weights, canonical codes (longest first, symbol order), the Huffman tree description: direct nibbles for
//                 <= 128 weights, else FSE-compressed — normalised counts by lane 0, the two interleaved state chains walked
//                 backwards with one table cell per lane (the cell whose interval holds the next state is a ballot away);
//   streams       wave k packs stream k: a lane owns a contiguous run of bytes, a wave suffix sum of the code lengths gives
//                 its bit offset (the LAST byte of a stream sits at bit 0: zstd reads its streams backwards);
//   output        the same LDS bit buffer / ZOut / flush_words machinery as the DEFLATE blocks.
// The layout is pinned on the CPU by oracle/zstd_enc.c (checked against libzstd there).
#pragma once
#include "deflate_dev.h"

namespace s5 {

constexpr int ZSTD_MAXBITS = 11;

// scratch of the tree description (wave 0); lives in DeflShared fields the zstd path does not otherwise use
struct ZstdDesc {
    uint8_t bytes[192];      // the description itself
    uint32_t cnt[16];        // weights histogram
    int32_t norm[16];
    uint32_t cell[64];       // FSE decode cell: symbol | bits << 8 | base << 16
};
static_assert(sizeof(ZstdDesc) <= sizeof(DeflShared::code) + sizeof(DeflShared::clseq), "ZstdDesc overlays S.code and S.clseq");

__device__ __forceinline__ void zput_bytes(uint32_t *obuf, const ZOut &z, uint32_t bitpos, uint64_t v, int nbytes) {   // one lane
    put_bits(obuf, z, bitpos, (uint32_t)v, nbytes >= 4 ? 32 : 8 * nbytes);
    if (nbytes > 4) put_bits(obuf, z, bitpos + 32, (uint32_t)(v >> 32), 8 * (nbytes - 4));
}

// Huffman tree description into D.bytes (wave 0, all 64 lanes).  Returns its length, 0 = not representable.
__device__ __forceinline__ uint32_t zstd_tree_desc(ZstdDesc &D, const uint8_t *lens, int n, int maxbits) {
    const int lane = lane_id();
    auto weight = [&](int s) -> uint32_t { const uint32_t l = lens[s]; return l ? (uint32_t)(maxbits + 1) - l : 0u; };
    if (n <= 128) {
        if (lane == 0) D.bytes[0] = (uint8_t)(127 + n);
        const int nb = (n + 1) / 2;
        if (lane < nb) {
            const uint32_t hi = weight(2 * lane), lo = 2 * lane + 1 < n ? weight(2 * lane + 1) : 0u;
            D.bytes[1 + lane] = (uint8_t)((hi << 4) | lo);
        }
        wave_sync();
        return 1u + (uint32_t)nb;
    }
    constexpr int LOG = 6, SIZE = 64;
    // histogram of the weights
    uint32_t mycnt = 0;                                            // lane q < 13 ends with the count of weight q
    int maxw = 0;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t w = s < n ? weight(s) : 99u;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            const uint32_t c = (uint32_t)__popcll(__ballot(w == (uint32_t)q));
            if (lane == q) mycnt += c;
            if (c && q > maxw) maxw = q;
        }
    }
    if (__ballot(lane < 13 && mycnt == (uint32_t)n)) return 0;     // one weight value only: an FSE stream of it cannot end
    int v = 0;
    if (lane < 13 && mycnt) { v = (int)(mycnt * SIZE / (uint32_t)n); if (v < 1) v = 1; }
    if (lane < 16) { D.cnt[lane] = lane < 13 ? mycnt : 0; D.norm[lane] = v; }
    wave_sync();
    uint32_t dl = 0;
    if (lane == 0) {
        int sum = 0, big = 0;
        for (int q = 0; q <= maxw; q++) { sum += D.norm[q]; if (D.cnt[q] > D.cnt[big]) big = q; }
        if (sum < SIZE) D.norm[big] += SIZE - sum;
        while (sum > SIZE) {
            int m = 0;
            for (int q = 1; q <= maxw; q++) if (D.norm[q] > D.norm[m]) m = q;
            D.norm[m]--; sum--;
        }
        // normalised counts (writer side of z_ncount / oracle fse_read_ncount)
        uint64_t acc = (uint64_t)(LOG - 5);
        int nacc = 4;
        uint32_t o = 1;
        int remaining = SIZE + 1, threshold = SIZE, nbits = LOG + 1, prev0 = 0, s = 0;
        while (s <= maxw && remaining > 1) {
            if (prev0) {
                int start = s;
                while (!D.norm[s]) s++;
                while (s >= start + 3) { start += 3; acc |= 3ull << nacc; nacc += 2; }
                acc |= (uint64_t)(s - start) << nacc; nacc += 2;
            }
            int count = D.norm[s++];
            const int maxv = (2 * threshold - 1) - remaining;
            remaining -= count;
            count++;
            if (count >= threshold) count += maxv;
            acc |= (uint64_t)count << nacc;
            nacc += nbits - (count < maxv);
            prev0 = count == 1;
            while (remaining < threshold) { nbits--; threshold >>= 1; }
            while (nacc >= 8) { D.bytes[o++] = (uint8_t)acc; acc >>= 8; nacc -= 8; }
        }
        if (nacc) D.bytes[o++] = (uint8_t)acc;
        dl = o;
        // the decoder's table for this distribution (z_fse_build), one cell per entry of D.cell
        uint32_t next[13];
        uint8_t *spread = reinterpret_cast<uint8_t *>(D.cnt);       // 64 bytes: cnt is dead
        for (int q = 0; q < 13; q++) next[q] = q <= maxw ? (uint32_t)D.norm[q] : 0u;
        const int step = (SIZE >> 1) + (SIZE >> 3) + 3, mask = SIZE - 1;
        int pos = 0;
        for (int q = 0; q <= maxw; q++)
            for (int i = 0; i < D.norm[q]; i++) { spread[pos] = (uint8_t)q; pos = (pos + step) & mask; }
        for (int i = 0; i < SIZE; i++) {
            const int q = spread[i];
            uint32_t ns = 0;
#pragma unroll
            for (int t = 0; t < 13; t++) if (t == q) { ns = next[t]; next[t]++; }
            const int nb = LOG - (31 - __clz((int)ns));
            D.cell[i] = (uint32_t)q | ((uint32_t)nb << 8) | (((ns << nb) - (uint32_t)SIZE) << 16);
        }
    }
    wave_sync();
    dl = (uint32_t)__builtin_amdgcn_readfirstlane((int)dl);
    const uint32_t cell = D.cell[lane];
    const uint32_t csym = cell & 255u, cnb = (cell >> 8) & 255u, cbase = cell >> 16;
    auto first_cell = [&](uint32_t w) -> uint32_t { return (uint32_t)__ffsll((long long)__ballot(csym == w)) - 1u; };   // its costliest state
    uint32_t st0, st1;
    {
        const uint32_t a = first_cell(weight(n - 1)), b = first_cell(weight(n - 2));
        if ((n - 1) & 1) { st1 = a; st0 = b; } else { st0 = a; st1 = b; }
    }
    uint64_t acc = 0;
    int nacc = 0;
    uint32_t o = dl;
    for (int k = n - 3; k >= 0; k--) {
        const uint32_t next = (k & 1) ? st1 : st0, wk = weight(k);
        const uint64_t m = __ballot(csym == wk && next >= cbase && next < cbase + (1u << cnb));
        const int found = __ffsll((long long)m) - 1;               // exactly one cell of a symbol covers a state
        const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)cbase, found), fn = (uint32_t)__builtin_amdgcn_readlane((int)cnb, found);
        acc |= (uint64_t)(next - fb) << nacc;
        nacc += (int)fn;
        if (k & 1) st1 = (uint32_t)found; else st0 = (uint32_t)found;
        if (nacc >= 32) {
            if (lane == 0) { D.bytes[o] = (uint8_t)acc; D.bytes[o + 1] = (uint8_t)(acc >> 8); D.bytes[o + 2] = (uint8_t)(acc >> 16); D.bytes[o + 3] = (uint8_t)(acc >> 24); }
            o += 4; acc >>= 32; nacc -= 32;
            if (o > 180) return 0;
        }
    }
    acc |= (uint64_t)st1 << nacc; nacc += LOG;
    acc |= (uint64_t)st0 << nacc; nacc += LOG;
    acc |= 1ull << nacc; nacc += 1;
    while (nacc > 0) { if (lane == 0) D.bytes[o] = (uint8_t)acc; o++; acc >>= 8; nacc -= 8; }
    if (o - 1 >= 128) return 0;
    if (lane == 0) D.bytes[0] = (uint8_t)(o - 1);
    wave_sync();
    return o;
}

// One zstd block of blen <= DEFL_BLK bytes at LDS `stage` into the bit buffer (B overlays obuf, as deflate_block MODE 2).
__device__ __forceinline__ void zstd_block(DeflShared &S, BuildScratch &B, uint32_t *obuf, uint32_t obuf_words, const uint8_t *stage,
                                           uint32_t blen, bool last, ZOut &z) {
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    ZstdDesc &D = *reinterpret_cast<ZstdDesc *>(S.code);
    // ---- histogram: one sub-histogram per wave, in scratch that is dead until build_lengths ----
    uint32_t *sub = wv == 0 ? S.freq : wv == 1 ? B.lf : wv == 2 ? B.nf : B.sort.bm;
    for (int i = lane; i < 256; i += 64) sub[i] = 0;
    if (tid < 16) S.red[tid & 7] = 0;
    wave_sync();
    {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage);
        const uint32_t nw = blen >> 2;
        for (uint32_t i = tid; i < nw; i += NT) {
            const uint32_t x = s32[i];
            atomicAdd(&sub[x & 255u], 1u); atomicAdd(&sub[(x >> 8) & 255u], 1u); atomicAdd(&sub[(x >> 16) & 255u], 1u); atomicAdd(&sub[x >> 24], 1u);
        }
        if ((uint32_t)tid < (blen & 3u)) atomicAdd(&sub[stage[4 * nw + tid]], 1u);
    }
    __syncthreads();
    const uint32_t f = S.freq[tid] + B.lf[tid] + B.nf[tid] + B.sort.bm[tid];
    __syncthreads();
    S.freq[tid] = f;
    if (tid < 64) S.freq[256 + tid] = 0;
    {   // distinct symbols and the largest one
        const uint64_t m = __ballot(f != 0);
        if (lane == 0 && m) { atomicAdd(&S.red[0], (uint32_t)__popcll(m)); atomicMax(&S.red[1], (uint32_t)(wv * 64 + 63 - __clzll((long long)m))); }
    }
    __syncthreads();
    const uint32_t distinct = S.red[0], maxsym = S.red[1];
    int type = 0;                                                   // 0 raw, 1 RLE, 2 compressed
    uint32_t dl = 0, hl = 0, csize = 0, sbytes[4] = {0, 0, 0, 0};
    const uint32_t per = (blen + 3) >> 2;
    uint32_t mybits = 0, mytotal = 0;
    uint32_t cs = 0, c0 = 0, c1 = 0;                                // my run of stream wv: bytes [c0, c1) of the block
    if (blen >= 64 && distinct == 1) type = 1;
    else if (blen >= 64) {
        if (tid == 0) S.dbg = 0;
        build_lengths(S, B, &B.sort, S.freq, 256, ZSTD_MAXBITS, S.lens, S.blcount, S.icount);
        // ---- stream k on wave k: bits of my run, then (wave 0) codes and the tree description ----
        const uint32_t sfrom = (uint32_t)wv * per, sto = wv == 3 ? blen : sfrom + per;
        const uint32_t count = sto - sfrom;
        cs = (count + 63) >> 6;
        c0 = min(sfrom + (uint32_t)lane * cs, sto); c1 = min(c0 + cs, sto);
        for (uint32_t i = c0; i < c1; i++) mybits += S.lens[stage[i]];
        mytotal = wave_sum(mybits);
        if (lane == 0) S.ws[wv] = mytotal;
        if (wv == 0) {
            int maxbits = 0;
#pragma unroll
            for (int L = 1; L <= ZSTD_MAXBITS; L++) if (__builtin_amdgcn_readfirstlane((int)S.blcount[L])) maxbits = L;
            // canonical codes: the longest codes take the smallest values, symbol order inside a length
            uint32_t next[ZSTD_MAXBITS + 2];
            next[ZSTD_MAXBITS + 1] = 0;
#pragma unroll
            for (int L = ZSTD_MAXBITS; L >= 1; L--) {
                const uint32_t above = L < ZSTD_MAXBITS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)S.blcount[L + 1]) : 0u;
                next[L] = L >= maxbits ? 0u : (next[L + 1] + above) >> 1;
            }
            for (int base = 0; base < 256; base += 64) {
                const int s = base + lane;
                const int l = S.lens[s];
                uint32_t mine = 0;
#pragma unroll
                for (int b = 1; b <= ZSTD_MAXBITS; b++) {
                    const uint64_t mask = __ballot(l == b);
                    if (mask == 0) continue;
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (l == b) mine = next[b] + below;
                    next[b] += (uint32_t)__popcll(mask);
                }
                S.freq[s] = l ? mine | ((uint32_t)l << 16) : 0u;     // the histogram is dead: S.freq now holds the codes
            }
            const uint32_t d = zstd_tree_desc(D, S.lens, (int)maxsym, maxbits);
            if (lane == 0) S.red[2] = d;
        }
        __syncthreads();
        dl = S.red[2];
        if (dl && 3 * per <= blen) {
#pragma unroll
            for (int k = 0; k < 4; k++) sbytes[k] = (S.ws[k] >> 3) + 1;   // + the end mark
            csize = dl + 6 + sbytes[0] + sbytes[1] + sbytes[2] + sbytes[3];
            hl = (blen <= 1023 && csize <= 1023) ? 3 : (blen <= 16383 && csize <= 16383) ? 4 : 5;
            if (hl + csize + 1 < blen) type = 2;
        }
    }
    // ---- B is dead: its storage becomes the bit buffer ----
    __syncthreads();
    for (uint32_t i = tid; i < obuf_words; i += NT) obuf[i] = 0;
    __syncthreads();
    if (tid == 0) obuf[0] = z.carry;
    __syncthreads();
    const uint32_t bsize = type == 2 ? hl + csize + 1 : type == 1 ? 1u : blen;
    const uint32_t p0 = z.bitpos;
    if (tid == 0) zput_bytes(obuf, z, p0, (last ? 1u : 0u) | ((uint32_t)type << 1) | ((type == 2 ? bsize : blen) << 3), 3);
    if (type == 0) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage);
        const uint32_t nw = blen >> 2;
        for (uint32_t i = tid; i < nw; i += NT) put_bits(obuf, z, p0 + 24 + 32 * i, s32[i], 32);
        if ((uint32_t)tid < (blen & 3u)) put_bits(obuf, z, p0 + 24 + 32 * nw + 8 * tid, stage[4 * nw + tid], 8);
    } else if (type == 1) {
        if (tid == 0) put_bits(obuf, z, p0 + 24, stage[0], 8);
    } else {
        const uint32_t lit0 = p0 + 24;
        if (tid == 0) {
            const uint64_t h = 2u | ((uint64_t)(hl - 2) << 2) | ((uint64_t)blen << 4) | ((uint64_t)csize << (hl == 3 ? 14 : hl == 4 ? 18 : 22));
            zput_bytes(obuf, z, lit0, h, (int)hl);
            const uint32_t jt = lit0 + 8 * (hl + dl);
            put_bits(obuf, z, jt, sbytes[0] | (sbytes[1] << 16), 32);
            put_bits(obuf, z, jt + 32, sbytes[2], 16);
        }
        for (uint32_t i = tid; i < dl; i += NT) put_bits(obuf, z, lit0 + 8 * (hl + i), D.bytes[i], 8);
        // my stream starts after the tree, the jump table and the streams before it; my run's bits sit above those of the lanes after me
        uint32_t sb = lit0 + 8 * (hl + dl + 6);
#pragma unroll
        for (int k = 0; k < 3; k++) if (k < wv) sb += 8 * sbytes[k];
        const uint32_t incl = wave_incl_add(mybits);
        uint32_t pos = sb + (mytotal - incl);
        if (lane == 0) put_bits(obuf, z, sb + mytotal, 1u, 1);      // end mark
        uint32_t w = (pos >> 5) - z.flushed;
        uint64_t acc = 0;
        uint32_t nacc = pos & 31u;
        for (uint32_t i = c1; i > c0; i--) {
            const uint32_t c = S.freq[stage[i - 1]];
            acc |= (uint64_t)(c & 0xFFFFu) << nacc;
            nacc += c >> 16;
            if (nacc >= 32) { atomicOr(&obuf[w], (uint32_t)acc); acc >>= 32; nacc -= 32; w++; }
        }
        if (nacc) atomicOr(&obuf[w], (uint32_t)acc);
        // the sequences section: no sequences (the byte is already zero)
    }
    z.bitpos = p0 + 8 * (3 + bsize);
    __syncthreads();
}

// One record: payload -> [u64 size][zstd frame] in the record's slot.  STAGED: the payload is parked in HBM (`src`) and passes
// through the LDS `stage` 16 KiB at a time, in place (k_deflate_staged's argument: a block is in LDS before its output is
// written, the slot bound leaves room for the framing); else `src` is the payload in LDS and blocks are taken where they lie.
// Returns the record length (prefix included).
template <bool STAGED>
__device__ __forceinline__ uint32_t zstd_record(DeflShared &S, uint32_t *obuf, uint32_t obuf_words, const uint8_t *src, uint8_t *stage,
                                                uint32_t plen, uint8_t *out) {
    const int tid = threadIdx.x;
    BuildScratch &B = *reinterpret_cast<BuildScratch *>(obuf);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    if (tid == 0) {   // frame header: magic, single segment + 4-byte content size; words 2, 3 and the first byte of word 4
        out32[2] = 0xFD2FB528u;
        out32[3] = 0xA0u | (plen << 8);
    }
    ZOut z;
    z.bitpos = 64 + 72;
    z.flushed = 4;
    z.carry = plen >> 24;
    uint32_t done = 0;
    do {
        const uint32_t blen = min(plen - done, (uint32_t)DEFL_BLK);
        const bool last = done + blen == plen;
        const uint8_t *blk = src + done;
        if (STAGED) {
            __syncthreads();
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src + done);
            uint4 *d4 = reinterpret_cast<uint4 *>(stage);
            for (uint32_t i = tid; i < (blen + 15) / 16; i += NT) d4[i] = s4[i];
            blk = stage;
            __syncthreads();
        }
        zstd_block(S, B, obuf, obuf_words, blk, blen, last, z);
        done += blen;
        if (!last) {
            flush_words(obuf, out32, z, false);
            z.carry = obuf[0];
            __syncthreads();
        }
    } while (done < plen);
    flush_words(obuf, out32, z, true);
    const uint32_t total = z.bitpos >> 3;
    __syncthreads();
    if (tid == 0) *reinterpret_cast<uint64_t *>(out) = (uint64_t)(total - 8);
    return total;
}

}   // namespace s5
