// zstd_enc_dev.h — Zstandard frames on the device, encode side (the zstd record press).
//
// slow5lib compresses a record with ZSTD_compress(level 1) (/root/reference/src/misc.c:259 names the method).  libzstd's
// match finder is a serial hash chain over the record; like the DEFLATE side this encoder does not reproduce its bytes but
// writes a VALID frame that libzstd decompresses to the identical payload.  Blocks of at most 16 KiB (the LDS stage of the staged
// path), each raw, RLE, or compressed = literals (Huffman in 4 streams, RLE or raw) + a sequences section holding the block's RUNS
// (a run of >= 5 equal bytes = one literal + one match at the frame's first repeat offset: see zstd_tokenise below).  On nanopore
// svb-zd records that is what a match finder finds: 0.8763 B/sample on the bench reads against libzstd level 1's 0.8767
// (literals only, round 1: 0.8959 with the record cut at its seams, 0.9467 as one block).
//
// One record per 256-thread workgroup, one block at a time:
//   runs          zstd_tokenise: runs -> sequence records, the literals compacted to the block's first bytes;
//   histogram     4 per-wave sub-histograms in the (still dead) build scratch, summed into S.freq;
//   code lengths  assign_lengths_wave<11> of the DEFLATE side (no tree; deflate_dev.h), the exact build_lengths<> behind it;
//   two waves     canonical codes (longest first, symbol order) on one; on another the Huffman tree description: direct
//                 nibbles for <= 128 weights, else FSE-compressed — normalised counts in uniform code, one decode cell per
//                 lane, the two interleaved state chains walked backwards (the cell whose interval holds the next state
//                 is a ballot away), transition bits packed by a prefix sum.  Which two waves rotates with the workgroup
//                 id, so the resident workgroups of a CU spread this serial work over its four SIMDs (15.0 -> 10.5 ms);
//   streams       wave k packs stream k: a lane owns a contiguous run of bytes, a wave suffix sum of the code lengths gives
//                 its bit offset (the LAST byte of a stream sits at bit 0: zstd reads its streams backwards);
//   sequences     zstd_sequences_wave: one wave, the two FSE state chains on readlane, bits placed by a prefix sum;
//   output        the same LDS bit buffer / ZOut / flush_words machinery as the DEFLATE blocks.
// The layout is pinned on the CPU by oracle/zstd_enc.c (checked against libzstd there).
#pragma once
#include "deflate_dev.h"
#include "zstd_seq_tables.h"

namespace s5 {

constexpr int ZSTD_MAXBITS = 11;

// scratch of the tree description (wave 0); lives in DeflShared fields the zstd path does not otherwise use
struct ZstdDesc {
    uint32_t words[48];      // the description itself (192 bytes), assembled with byte stores and word ORs
    uint8_t state[256];      // state[k]: FSE state that emits weight k
    uint32_t cell[64];       // FSE decode cell: symbol | bits << 8 | base << 16
};
static_assert(sizeof(ZstdDesc) <= sizeof(DeflShared::code) + sizeof(DeflShared::clseq), "ZstdDesc overlays S.code and S.clseq");

__device__ __forceinline__ void zput_bytes(uint32_t *obuf, const ZOut &z, uint32_t bitpos, uint64_t v, int nbytes) {   // one lane
    put_bits(obuf, z, bitpos, (uint32_t)v, nbytes >= 4 ? 32 : 8 * nbytes);
    if (nbytes > 4) put_bits(obuf, z, bitpos + 32, (uint32_t)(v >> 32), 8 * (nbytes - 4));
}

// Huffman tree description into D.words, in three steps so that the one serial part can run on two waves at once.
// Weights w[0..n) are sent (the weight of symbol n, the largest one in use, is implied).  Up to 128 weights go as nibbles.
// More are FSE-compressed (table log 6).  zstd_desc_head (one wave): the 13-bin weight histogram is normalised and written as
// the count header in uniform code; the 64 decode cells are built one per lane (cell i is slot 3i mod 64 of the spread:
// 43 * 3 = 1 mod 64).  zstd_desc_chain: the two interleaved state chains are walked backwards, one chain per wave — one
// ballot per weight finds the cell of symbol w[k] whose interval holds state[k + 2].  zstd_desc_pack (one wave): the
// transition bits are packed by all lanes with a prefix sum of their widths.
// zstd_desc_head returns 0 = not representable, the finished length for the nibble form, (count header bytes | 1 << 31) for FSE.
__device__ __forceinline__ uint32_t zstd_desc_head(ZstdDesc &D, const uint8_t *lens, int n, int maxbits) {
    const int lane = lane_id();
    uint8_t *bytes = reinterpret_cast<uint8_t *>(D.words);
    uint32_t wr[4];                                                // weights of symbols lane, lane + 64, lane + 128, lane + 192
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t l = lens[lane + 64 * j]; wr[j] = l ? (uint32_t)(maxbits + 1) - l : 0u; }
    if (n <= 128) {
        if (lane == 0) bytes[0] = (uint8_t)(127 + n);
        // byte j = w[2j] << 4 | w[2j + 1]: pair up neighbouring lanes
        const uint32_t nx0 = wave_next(wr[0], 0), nx1 = wave_next(wr[1], 0);
        if (!(lane & 1)) {
            if (lane < n) bytes[1 + (lane >> 1)] = (uint8_t)((wr[0] << 4) | (lane + 1 < n ? nx0 : 0u));
            if (lane + 64 < n) bytes[1 + 32 + (lane >> 1)] = (uint8_t)((wr[1] << 4) | (lane + 65 < n ? nx1 : 0u));
        }
        wave_sync();
        return 1u + (uint32_t)(n + 1) / 2;
    }
    constexpr int LOG = 6, SIZE = 64;
    if (lane < 48) D.words[lane] = 0;
    // histogram of the weights
    uint32_t mycnt = 0;                                            // lane q < 13 ends with the count of weight q
    int maxw = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t w = lane + 64 * j < n ? wr[j] : 99u;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            const uint32_t c = (uint32_t)__popcll(__ballot(w == (uint32_t)q));
            if (lane == q) mycnt += c;
            if (c && q > maxw) maxw = q;
        }
    }
    if (__ballot(lane < 13 && mycnt == (uint32_t)n)) return 0;     // one weight value only: an FSE stream of it cannot end
    // normalise to 64 slots, every weight in use at least one; lane q holds norm[q].  Uniform code: no lane-0 section.
    uint32_t v = 0;
    if (lane < 13 && mycnt) { v = mycnt * SIZE / (uint32_t)n; if (v < 1) v = 1; }
    {
        uint32_t sum = wave_sum(v);
        const int kbig = __builtin_amdgcn_readlane(wave_incl_max(lane < 13 ? (int)((mycnt << 4) | (uint32_t)(15 - lane)) : 0), 63);
        if (sum < (uint32_t)SIZE && lane == 15 - (kbig & 15)) v += (uint32_t)SIZE - sum;
        while (sum > (uint32_t)SIZE) {                             // take from the largest entries
            const int km = __builtin_amdgcn_readlane(wave_incl_max(lane < 13 ? (int)((v << 4) | (uint32_t)(15 - lane)) : 0), 63);
            if (lane == 15 - (km & 15)) v--;
            sum--;
        }
    }
    // normalised counts (writer side of z_ncount / oracle fse_read_ncount); bits 0-7 are the header byte, filled in last
    uint32_t dl;
    {
        uint64_t acc = (uint64_t)(LOG - 5) << 8;
        int nacc = 12;
        uint32_t ow = 0;
        int remaining = SIZE + 1, threshold = SIZE, nbits = LOG + 1, prev0 = 0, sy = 0;
        while (sy <= maxw && remaining > 1) {
            if (prev0) {
                int start = sy;
                while (!__builtin_amdgcn_readlane((int)v, sy)) sy++;
                while (sy >= start + 3) { start += 3; acc |= 3ull << nacc; nacc += 2; }
                acc |= (uint64_t)(sy - start) << nacc; nacc += 2;
            }
            int count = __builtin_amdgcn_readlane((int)v, sy);
            sy++;
            const int maxv = (2 * threshold - 1) - remaining;
            remaining -= count;
            count++;
            if (count >= threshold) count += maxv;
            acc |= (uint64_t)count << nacc;
            nacc += nbits - (count < maxv);
            prev0 = count == 1;
            while (remaining < threshold) { nbits--; threshold >>= 1; }
            if (nacc >= 32) { if (lane == 0) D.words[ow] = (uint32_t)acc; ow++; acc >>= 32; nacc -= 32; }
        }
        if (nacc && lane == 0) D.words[ow] = (uint32_t)acc;
        dl = 4 * ow + (uint32_t)((nacc + 7) >> 3);
    }
    wave_sync();
    // my decode cell: slot j of the spread holds the symbol whose cumulative count covers j; its k-th cell (in cell order) is
    // the decoder's "next state" count + k
    uint32_t csym = 0, cnb, cbase;
    {
        const uint32_t slot = (3u * (uint32_t)lane) & 63u;
        uint32_t cum = 0, mynorm = 0;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            const uint32_t nq = (uint32_t)__builtin_amdgcn_readlane((int)v, q);
            if (slot >= cum && slot < cum + nq) { csym = (uint32_t)q; mynorm = nq; }
            cum += nq;
        }
        uint32_t ns = 0;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            const uint64_t m = __ballot(csym == (uint32_t)q);
            if (csym == (uint32_t)q) ns = mynorm + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        }
        cnb = (uint32_t)LOG - (uint32_t)(31 - __clz((int)ns));
        cbase = (ns << cnb) - (uint32_t)SIZE;
        D.cell[lane] = csym | (cnb << 8) | (cbase << 16);
    }
    wave_sync();
    return dl | 0x80000000u;                                        // FSE: the chains and the packing follow
}

// One of the two interleaved state chains, backwards: par 0 walks k = n-1, n-3, ..., par 1 walks k = n-2, n-4, ...
// (any wave; the cells come from D.cell).  state[k] = the cell of weight w[k] whose interval holds state[k + 2].
__device__ __forceinline__ void zstd_desc_chain(ZstdDesc &D, const uint8_t *lens, int n, int maxbits, int par) {
    const int lane = lane_id();
    // lanes 0..31 hold 8 weights each, 4 bits apiece: w[k] is one v_readlane and a bit-field extract away
    uint32_t wp = 0;
    if (lane < 32) {
        const uint32_t *l32 = reinterpret_cast<const uint32_t *>(lens);   // S.lens is 16-byte aligned
        const uint32_t lo = l32[2 * lane], hi = l32[2 * lane + 1];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t l = ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 255u;
            wp |= (l ? (uint32_t)(maxbits + 1) - l : 0u) << (4 * i);
        }
    }
    const uint32_t cell = D.cell[lane];
    const uint32_t csym = cell & 255u, cnb = (cell >> 8) & 255u, cbase = cell >> 16;
    const uint32_t cend = cbase + (1u << cnb);
    auto wget = [&](int k) -> uint32_t { return ((uint32_t)__builtin_amdgcn_readlane((int)wp, k >> 3) >> ((k & 7) * 4)) & 15u; };   // uniform
    int k = n - 1 - par;
    uint32_t st = (uint32_t)__ffsll((long long)__ballot(csym == wget(k))) - 1u;   // the last symbol of a chain: its costliest state (>= 1 bit)
    D.state[k] = (uint8_t)st;                                       // every lane stores the same byte: no exec juggling in the loop
    for (k -= 2; k >= 0; k -= 2) {
        const uint32_t w = wget(k);
        st = (uint32_t)__ffsll((long long)__ballot(csym == w && st >= cbase && st < cend)) - 1u;   // exactly one cell of a symbol covers a state
        D.state[k] = (uint8_t)st;
    }
    wave_sync();
}

// The transition bits, the two first states and the end mark behind the count header of dl bytes.  Returns the length of the
// whole description, 0 if it would not fit the 127 bytes its header byte can say.
__device__ __forceinline__ uint32_t zstd_desc_pack(ZstdDesc &D, int n, uint32_t dl) {
    const int lane = lane_id();
    constexpr int LOG = 6;
    uint8_t *bytes = reinterpret_cast<uint8_t *>(D.words);
    // transition k (k = n-3 first, at the lowest bits): value state[k+2] - base(state[k]), width bits(state[k])
    uint32_t run = 8 * dl;                                          // bit position in D.words
    for (int t0 = 0; t0 < n - 2; t0 += 64) {
        const int t = t0 + lane, k = n - 3 - t;
        uint32_t nb = 0, val = 0;
        if (t < n - 2) {
            const uint32_t cell = D.cell[D.state[k]];
            nb = (cell >> 8) & 255u;
            val = (uint32_t)D.state[k + 2] - (cell >> 16);
        }
        const uint32_t incl = wave_incl_add(nb);
        const uint32_t pos = run + incl - nb;
        if (nb) {
            atomicOr(&D.words[pos >> 5], val << (pos & 31));
            if ((pos & 31) + nb > 32) atomicOr(&D.words[(pos >> 5) + 1], val >> (32 - (pos & 31)));
        }
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (run > 8 * 150) return 0;                                // cannot end below 128 bytes
    }
    if (lane == 0) {                                                // the two final states and the end mark
        const uint32_t tail = (uint32_t)D.state[1] | ((uint32_t)D.state[0] << LOG) | (1u << (2 * LOG));
        atomicOr(&D.words[run >> 5], tail << (run & 31));
        if ((run & 31) + 13 > 32) atomicOr(&D.words[(run >> 5) + 1], tail >> (32 - (run & 31)));
    }
    const uint32_t o = (run + 13 + 7) >> 3;
    if (o - 1 >= 128) return 0;
    if (lane == 0) bytes[0] = (uint8_t)(o - 1);
    wave_sync();
    return o;
}

// ---- runs as sequences ----
// A run of >= ZSTD_RMIN equal bytes is sent as its first byte (a literal) + ONE match at offset 1.  Offset 1 is the frame's first
// repeat offset and stays it (nothing else is ever used), so every sequence carries offset code 0: the offsets table is one RLE
// byte and costs no bits; literal and match lengths go through the predefined FSE tables (zstd_seq_tables.h).  This is what
// libzstd's match finder gets out of the key bytes of an svb-zd record (0.8959 -> 0.8735 B/sample on the bench reads, libzstd
// level 1: 0.8767), and it makes the key | data block split unnecessary.
// zstd_tokenise: lane t owns bytes [t K, t K + K) of the block (K a multiple of 4);
// break mask by byte compares, the run bounds outside the chunk from a block prefix max / suffix min of the break positions;
// a prefix sum of the literal and sequence counts; then the literals are compacted (through the bit buffer, dead at this point)
// to the block's first bytes and the sequence records (literal index | match length << 16, in stream order) go into the freed
// bytes behind them.  Returns false and leaves the block alone if there is no run, no room, or no certain gain.
constexpr uint32_t ZSTD_RMIN = 5;

__device__ __forceinline__ bool zstd_tokenise(DeflShared &S, uint8_t *stage, uint8_t *tmp, uint32_t blen, uint32_t &nlit_tot, uint32_t &nseq_tot) {
    const int tid = threadIdx.x;
    const int K = (((int)blen + NT - 1) / NT + 3) & ~3;                     // <= 64 (uniform)
    const int base = tid * K;
    const int kk = max(0, min(K, (int)blen - base));
    uint64_t brk = 0;
    {
        uint32_t prev = kk > 0 && base > 0 ? (uint32_t)stage[base - 1] : 0x100u;
        if (kk > 0 && base == 0) brk = 1;                                  // the block's first byte starts a run
        for (int q = 0; 4 * q < kk; q++) {
            const uint32_t x = *reinterpret_cast<const uint32_t *>(stage + base + 4 * q);   // (stage is 4-byte aligned; bytes past blen are masked below)
            const uint32_t d = x ^ ((x << 8) | (prev & 0xFFu));
            const uint32_t m = ((d & 0xFFu) ? 1u : 0u) | ((d & 0xFF00u) ? 2u : 0u) | ((d & 0xFF0000u) ? 4u : 0u) | ((d >> 24) ? 8u : 0u);
            brk |= (uint64_t)m << (4 * q);
            prev = x >> 24;
        }
    }
    const uint64_t valid = kk >= 64 ? ~0ull : (1ull << kk) - 1ull;
    brk &= valid;
    const int local_last = brk ? base + 63 - __clzll((long long)brk) : -1;
    const int local_first = brk ? base + __ffsll((long long)brk) - 1 : (int)blen;
    const int lastb = block_excl_max(local_last, -1, S.ws);                 // last break in front of my chunk (position 0 is one)
    const int nextb = block_suffix_excl_min(local_first, (int)blen, S.ws);  // first break behind it, or the block's end
    // runs that touch my chunk: cov = my bytes inside a match; the loop body runs for long runs only
    uint64_t cov = 0;
    uint32_t myseq = 0;
    const uint64_t I = ~brk & valid;
    uint64_t cand = 0;
    if (kk > 0) {
        if (!(brk & 1ull)) {                                               // the run that comes in from the lanes in front
            const int e_loc = brk ? __ffsll((long long)brk) - 1 : kk;
            const int e = brk ? base + e_loc : nextb;
            if (e - lastb >= (int)ZSTD_RMIN) cov |= e_loc >= 64 ? ~0ull : (1ull << e_loc) - 1ull;
        }
        cand = brk & (I >> 1) & (I >> 2) & (I >> 3) & (I >> 4);            // RMIN - 1 in-run bytes follow inside my chunk
        if (brk) cand |= 1ull << (63 - __clzll((long long)brk));           // my last break: its run may go on behind my chunk
    }
    {
        uint64_t t = cand;
        while (t) {
            const int j = __ffsll((long long)t) - 1;
            t &= t - 1;
            const uint64_t hi = j < 63 ? brk & ~((2ull << j) - 1ull) : 0ull;
            const int jn = hi ? __ffsll((long long)hi) - 1 : kk;           // my bytes (j, jn) belong to the run
            const int e = hi ? base + jn : nextb;
            if (e - (base + j) >= (int)ZSTD_RMIN) {
                myseq++;
                const uint64_t upto = jn >= 64 ? ~0ull : (1ull << jn) - 1ull;
                cov |= upto & ~((2ull << j) - 1ull);
            }
        }
    }
    cov &= valid;
    const uint32_t mylit = (uint32_t)kk - (uint32_t)__popcll(cov);
    uint32_t packed_tot;
    const uint32_t packed = block_excl_add(mylit | (myseq << 16), S.ws, packed_tot);
    const uint32_t nlit = packed_tot & 0xFFFFu, nseq = packed_tot >> 16;    // (nlit <= 16384: no carry into the upper half)
    nlit_tot = nlit; nseq_tot = nseq;
    const bool use = nseq != 0 && !(nseq == 1 && nlit == 1) && nlit + 4 * nseq + 8 <= blen &&
                     3 + nlit + 4 + ((12 * nseq + (nlit >> 2) + ((blen - nlit) >> 3) + 20) >> 3) + 1 < blen;
    if (!use) return false;
    // ---- compaction: my literals into the (dead) scratch `tmp`, then the whole literal string back to the block's first bytes ----
    const uint32_t litbase = packed & 0xFFFFu;
    {
        uint32_t d = litbase;
        uint64_t lits = ~cov & valid;
        while (lits) {
            const int j = __ffsll((long long)lits) - 1;
            lits &= lits - 1;
            tmp[d++] = stage[base + j];
        }
    }
    __syncthreads();
    {
        const uint32_t *t32 = reinterpret_cast<const uint32_t *>(tmp);
        uint32_t *s32 = reinterpret_cast<uint32_t *>(stage);
        for (uint32_t i = tid; i < (nlit + 3u) >> 2; i += NT) s32[i] = t32[i];
    }
    // the sequence records behind them (their bytes are all covered or copied: nothing there is read again)
    uint32_t *rec = reinterpret_cast<uint32_t *>(stage + ((nlit + 3u) & ~3u));
    {
        uint32_t k = packed >> 16;
        uint64_t t = cand;
        while (t) {
            const int j = __ffsll((long long)t) - 1;
            t &= t - 1;
            const uint64_t hi = j < 63 ? brk & ~((2ull << j) - 1ull) : 0ull;
            const int e = hi ? base + __ffsll((long long)hi) - 1 : nextb;
            const int R = e - (base + j);
            if (R >= (int)ZSTD_RMIN) {
                const uint32_t litpos = litbase + (uint32_t)__popcll(~cov & valid & ((2ull << j) - 1ull));   // literals up to and including the head
                rec[k++] = litpos | ((uint32_t)(R - 1) << 16);
            }
        }
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ uint32_t zstd_ll_code(uint32_t ll) {
    return ll < 16 ? ll : ll < 24 ? 16 + ((ll - 16) >> 1) : ll < 32 ? 20 + ((ll - 24) >> 2) : ll < 48 ? 22 + ((ll - 32) >> 3) : ll < 64 ? 24u : 50u - (uint32_t)__clz((int)ll);
}
__device__ __forceinline__ uint32_t zstd_ml_code(uint32_t mb) {            // mb = match length - 3
    return mb < 32 ? mb : mb < 40 ? 32 + ((mb - 32) >> 1) : mb < 48 ? 36 + ((mb - 40) >> 2) : mb < 64 ? 38 + ((mb - 48) >> 3)
         : mb < 96 ? 40 + ((mb - 64) >> 4) : mb < 128 ? 42u : 67u - (uint32_t)__clz((int)mb);
}

// The sequences section at bit position pos0 (a byte boundary) of the bit buffer, by ONE wave: count, modes (predefined | RLE |
// predefined), the offset code, then the bitstream — written from the LAST sequence to the first, 64 sequences at a time:
// the lanes work out codes and extra bits, a serial loop walks the two FSE state chains (the 64-entry next-state tables sit one
// entry per lane: a transition is two v_readlane and a few scalar instructions, no memory access), the lanes place their bits
// with a prefix sum.  Returns the section's length in bytes.
__device__ __forceinline__ uint32_t zstd_sequences_wave(uint32_t *obuf, const ZOut &z, uint32_t pos0, const uint32_t *rec, uint32_t nseq) {
    const int lane = lane_id();
    uint32_t pos = pos0;
    if (lane == 0) {
        if (nseq < 128) { put_bits(obuf, z, pos, nseq | (0x10u << 8), 24); }                           // count, modes, offset code 0
        else { put_bits(obuf, z, pos, (128u + (nseq >> 8)) | ((nseq & 255u) << 8) | (0x10u << 16), 32); }
    }
    pos += nseq < 128 ? 24 : 32;
    const uint32_t nextL = ZSEQ_LL_NEXT[lane], nextM = ZSEQ_ML_NEXT[lane];
    uint32_t stL = 0, stM = 0;                                              // uniform
    for (int hi = (int)nseq - 1; hi >= 0; hi -= 64) {
        const int k = hi - lane;
        uint32_t lc = 0, mc = 0, llx = 0, mlx = 0, llb = 0, mlb = 0;
        int dnbL = 0, dfsL = 0, dnbM = 0, dfsM = 0;
        if (k >= 0) {
            const uint32_t r = rec[k], prevlit = k ? rec[k - 1] & 0xFFFFu : 0u;
            const uint32_t ll = (r & 0xFFFFu) - prevlit, mb = (r >> 16) - 3u;
            lc = zstd_ll_code(ll); mc = zstd_ml_code(mb);
            llb = ZSEQ_LL_BITS[lc]; mlb = ZSEQ_ML_BITS[mc];
            llx = ll & ((1u << llb) - 1u); mlx = mb & ((1u << mlb) - 1u);
            dnbL = ZSEQ_LL_DNB[lc]; dfsL = ZSEQ_LL_DFS[lc]; dnbM = ZSEQ_ML_DNB[mc]; dfsM = ZSEQ_ML_DFS[mc];
        }
        const int cnt = min(64, hi + 1);
        uint32_t sb = 0;                                                    // my state bits: value | width << 16 (match-length state first)
        for (int i = 0; i < cnt; i++) {
            const uint32_t aL = (uint32_t)__builtin_amdgcn_readlane(dnbL, i), aM = (uint32_t)__builtin_amdgcn_readlane(dnbM, i);
            const int fL = __builtin_amdgcn_readlane(dfsL, i), fM = __builtin_amdgcn_readlane(dfsM, i);
            uint32_t out = 0;
            if (hi == (int)nseq - 1 && i == 0) {                            // the chains start here: a state, no bits
                const uint32_t nbM = (aM + (1u << 15)) >> 16, nbL = (aL + (1u << 15)) >> 16;
                stM = (uint32_t)__builtin_amdgcn_readlane((int)nextM, (int)((((nbM << 16) - aM) >> nbM) + (uint32_t)fM));
                stL = (uint32_t)__builtin_amdgcn_readlane((int)nextL, (int)((((nbL << 16) - aL) >> nbL) + (uint32_t)fL));
            } else {
                const uint32_t nbM = (stM + aM) >> 16, nbL = (stL + aL) >> 16;
                out = (stM & ((1u << nbM) - 1u)) | ((stL & ((1u << nbL) - 1u)) << nbM) | ((nbM + nbL) << 16);
                stM = (uint32_t)__builtin_amdgcn_readlane((int)nextM, (int)((stM >> nbM) + (uint32_t)fM));
                stL = (uint32_t)__builtin_amdgcn_readlane((int)nextL, (int)((stL >> nbL) + (uint32_t)fL));
            }
            if (lane == i) sb = out;
        }
        const uint32_t nsb = sb >> 16;
        const uint32_t nb = k >= 0 ? nsb + llb + mlb : 0u;                  // <= 12 + 16 + 16
        const uint64_t v = (uint64_t)(sb & 0xFFFFu) | ((uint64_t)llx << nsb) | ((uint64_t)mlx << (nsb + llb));
        const uint32_t incl = wave_incl_add(nb);
        if (nb) {
            const uint32_t p = pos + incl - nb;
            const uint32_t wd = (p >> 5) - z.flushed, sh = p & 31u;
            const uint64_t lo = v << sh;
            atomicOr(&obuf[wd], (uint32_t)lo);
            if (sh + nb > 32) atomicOr(&obuf[wd + 1], (uint32_t)(lo >> 32));
            if (sh + nb > 64) atomicOr(&obuf[wd + 2], (uint32_t)(v >> (64 - sh)));
        }
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) put_bits(obuf, z, pos, (stM & 63u) | ((stL & 63u) << 6) | (1u << 12), 13);          // final states, end mark
    pos += 13;
    return ((pos + 7) >> 3) - (pos0 >> 3);
}

// One zstd block of blen <= DEFL_BLK bytes at LDS `stage` into the bit buffer (B overlays obuf, as deflate_block MODE 2).
// seq_on: runs go out as sequences (zstd_tokenise compacts the block's literals in place: `stage` is consumed).
__device__ __forceinline__ void zstd_block(DeflShared &S, BuildScratch &B, uint32_t *obuf, uint32_t obuf_words, uint8_t *stage,
                                           uint32_t blen, bool last, ZOut &z, uint32_t dbg = 0, bool force_raw = false, bool seq_on = true) {
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    ZstdDesc &D = *reinterpret_cast<ZstdDesc *>(S.code);
    // ---- runs -> sequences; what is left of the block are its nl literals, stage[0 .. nl) ----
    uint32_t nl = blen, nseq = 0;
    bool seq = false;
    if (seq_on && blen >= 64 && !force_raw && (reinterpret_cast<uintptr_t>(stage) & 3) == 0) {
        seq = zstd_tokenise(S, stage, reinterpret_cast<uint8_t *>(obuf), blen, nl, nseq);
        if (!seq) { nl = blen; nseq = 0; }
    }
    if (dbg == 5) { z.bitpos += nl + nseq; return; }   // tools/ cut-offs (S5GPU_DEBUG_STAGE)
    // ---- histogram: one sub-histogram per wave, in scratch that is dead until build_lengths ----
    uint32_t *sub = wv == 0 ? S.freq : wv == 1 ? B.lf : wv == 2 ? B.nf : B.sort.bm;
    for (int i = lane; i < 256; i += 64) sub[i] = 0;
    if (tid < 16) S.red[tid & 7] = 0;
    wave_sync();
    {
        // the block may start anywhere in the payload (the key | data split): bytes up to the first aligned dword, dwords, tail
        const uint32_t head = min(nl, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(stage) & 3)) & 3));
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(stage + head);
        const uint32_t nw = (nl - head) >> 2;
        for (uint32_t i = tid; i < nw; i += NT) {
            const uint32_t x = s32[i];
            atomicAdd(&sub[x & 255u], 1u); atomicAdd(&sub[(x >> 8) & 255u], 1u); atomicAdd(&sub[(x >> 16) & 255u], 1u); atomicAdd(&sub[x >> 24], 1u);
        }
        if ((uint32_t)tid < head) atomicAdd(&sub[stage[tid]], 1u);
        const uint32_t tail0 = head + 4 * nw;
        if ((uint32_t)tid < nl - tail0) atomicAdd(&sub[stage[tail0 + tid]], 1u);
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
    if (dbg == 1) { z.bitpos += distinct; return; }
    int type = 0;                                                   // block: 0 raw, 1 RLE, 2 compressed
    int lit_mode = 0;                                               // literals section of a compressed block: 0 raw, 1 RLE, 2 Huffman
    uint32_t dl = 0, hl = 0, csize = 0, sbytes[4] = {0, 0, 0, 0};
    const uint32_t per = (nl + 3) >> 2;
    const uint32_t rawl = (nl < 32 ? 1u : nl < 4096 ? 2u : 3u) + nl;   // a raw literals section
    uint32_t mybits = 0, mytotal = 0;
    uint32_t cs = 0, c0 = 0, c1 = 0;                                // my run of stream wv: literals [c0, c1)
    if (!seq && blen >= 64 && distinct == 1 && !force_raw) type = 1;
    else if (nl >= 64 && distinct > 1 && !force_raw) {
        // code lengths without a tree (assign_lengths_wave of the DEFLATE side, capped at 11 bits: 2.2 -> 0.5 ms per 262 k blocks); the
        // round-based exact construction only if the clamp over-subscribes the code space (never seen on signal payloads)
        if (tid == 0) S.dbg = 0;
        __syncthreads();
        if (wv == 0) {
            // (a handful of symbols is where the greedy hand-out is worst in relative terms — 3.5 % on a 6-symbol block — and where the
            // exact construction costs next to nothing)
            const bool okl = distinct > 16 && assign_lengths_wave<ZSTD_MAXBITS>(S.freq, 256, S.lens, S.blcount, S.bins);
            if (lane == 0 && !okl) S.dbg = 1;
        }
        __syncthreads();
        if (S.dbg) {
            __syncthreads();
            if (tid == 0) S.dbg = 0;
            build_lengths(S, B, &B.sort, S.freq, 256, ZSTD_MAXBITS, S.lens, S.blcount, S.icount);
        }
        if (dbg == 2) { z.bitpos += S.lens[tid]; return; }
        // ---- stream k on wave k: bits of my run, then (wave 0) codes and the tree description ----
        const uint32_t sfrom = min((uint32_t)wv * per, nl), sto = wv == 3 ? nl : min(sfrom + per, nl);
        const uint32_t count = sto - sfrom;
        cs = (count + 63) >> 6;
        c0 = min(sfrom + (uint32_t)lane * cs, sto); c1 = min(c0 + cs, sto);
        if (dbg != 33) for (uint32_t i = c0; i < c1; i++) mybits += S.lens[stage[i]];
        mytotal = wave_sum(mybits);
        if (lane == 0) S.ws[wv] = mytotal;
        // The serial jobs go to different waves, and to different ones from one workgroup to the next: the waves of the
        // workgroups resident on a CU then spread this work over its four SIMDs instead of piling it on one.
        // role 0: tree description (head, chain of the odd-from-the-end weights, packing); role 1: canonical codes;
        // role 2: the other chain.
        const int role = (wv + (int)blockIdx.x) & 3;
        int maxbits = 0;
#pragma unroll
        for (int L = 1; L <= ZSTD_MAXBITS; L++) if (__builtin_amdgcn_readfirstlane((int)S.blcount[L])) maxbits = L;
        if (role == 1 && dbg != 32) {
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
        } else if (role == 0) {
            const uint32_t h = dbg == 31 ? 1u : zstd_desc_head(D, S.lens, (int)maxsym, maxbits);
            if (lane == 0) S.red[2] = h;
        }
        __syncthreads();
        const uint32_t h = S.red[2];
        if (h >> 31) {                                              // FSE-compressed weights: one state chain on each of two waves
            if (role == 0) zstd_desc_chain(D, S.lens, (int)maxsym, maxbits, 0);
            else if (role == 2) zstd_desc_chain(D, S.lens, (int)maxsym, maxbits, 1);
            __syncthreads();
            if (role == 0) {
                const uint32_t o = zstd_desc_pack(D, (int)maxsym, h & 0x7FFFFFFFu);
                if (lane == 0) S.red[2] = o;
            }
            __syncthreads();
        }
        dl = S.red[2];
        if (dbg == 3 || dbg > 30) { z.bitpos += dl + S.ws[0]; return; }
        if (dl && 3 * per <= nl) {
#pragma unroll
            for (int k = 0; k < 4; k++) sbytes[k] = (S.ws[k] >> 3) + 1;   // + the end mark
            csize = dl + 6 + sbytes[0] + sbytes[1] + sbytes[2] + sbytes[3];
            hl = (nl <= 1023 && csize <= 1023) ? 3 : (nl <= 16383 && csize <= 16383) ? 4 : 5;
            if (seq) { if (hl + csize < rawl) lit_mode = 2; }
            else if (hl + csize + 1 < blen) { type = 2; lit_mode = 2; }
        }
    }
    if (seq) {                                                      // (zstd_tokenise made sure the block shrinks even with raw literals)
        type = 2;
        if (lit_mode != 2) lit_mode = distinct == 1 && nl >= 2 ? 1 : 0;
    }
    // ---- B is dead: its storage becomes the bit buffer ----
    __syncthreads();
    for (uint32_t i = tid; i < obuf_words; i += NT) obuf[i] = 0;
    __syncthreads();
    if (tid == 0) obuf[0] = z.carry;
    __syncthreads();
    const uint32_t p0 = z.bitpos;
    uint32_t bsize = type == 1 ? 1u : blen;
    auto put_raw = [&](uint32_t at, const uint8_t *from, uint32_t n) {   // n bytes at bit position `at` (a byte boundary), by all lanes
        const uint32_t head = min(n, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(from) & 3)) & 3));
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(from + head);
        const uint32_t nw = (n - head) >> 2, tail0 = head + 4 * nw;
        for (uint32_t i = tid; i < nw; i += NT) put_bits(obuf, z, at + 8 * head + 32 * i, s32[i], 32);
        if ((uint32_t)tid < head) put_bits(obuf, z, at + 8 * tid, from[tid], 8);
        if ((uint32_t)tid < n - tail0) put_bits(obuf, z, at + 8 * (tail0 + tid), from[tail0 + tid], 8);
    };
    if (type == 0) put_raw(p0 + 24, stage, blen);
    else if (type == 1) { if (tid == 0) put_bits(obuf, z, p0 + 24, stage[0], 8); }
    else {
        const uint32_t lit0 = p0 + 24;
        uint32_t litbytes;
        if (lit_mode == 2) {
            litbytes = hl + csize;
            if (tid == 0) {
                const uint64_t h = 2u | ((uint64_t)(hl - 2) << 2) | ((uint64_t)nl << 4) | ((uint64_t)csize << (hl == 3 ? 14 : hl == 4 ? 18 : 22));
                zput_bytes(obuf, z, lit0, h, (int)hl);
                const uint32_t jt = lit0 + 8 * (hl + dl);
                put_bits(obuf, z, jt, sbytes[0] | (sbytes[1] << 16), 32);
                put_bits(obuf, z, jt + 32, sbytes[2], 16);
            }
            for (uint32_t i = tid; i < dl; i += NT) put_bits(obuf, z, lit0 + 8 * (hl + i), reinterpret_cast<const uint8_t *>(D.words)[i], 8);
            // my stream starts after the tree, the jump table and the streams before it; my run's bits sit above those of the lanes after me
            uint32_t sb = lit0 + 8 * (hl + dl + 6);
#pragma unroll
            for (int k = 0; k < 3; k++) if (k < wv) sb += 8 * sbytes[k];
            const uint32_t incl = wave_incl_add(mybits);
            uint32_t pos = sb + (mytotal - incl);
            if (lane == 0) put_bits(obuf, z, sb + mytotal, 1u, 1);      // end mark
            if (dbg == 4) { z.bitpos += pos; return; }
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
        } else {
            // raw or RLE literals: 1-, 2- or 3-byte header with the regenerated size
            const uint32_t hb = nl < 32 ? 1u : nl < 4096 ? 2u : 3u;
            if (tid == 0) {
                const uint32_t h = (uint32_t)lit_mode | (hb == 1 ? nl << 3 : ((hb == 2 ? 1u : 3u) << 2) | (nl << 4));
                put_bits(obuf, z, lit0, h, 8 * hb);
                if (lit_mode == 1) put_bits(obuf, z, lit0 + 8 * hb, stage[0], 8);
            }
            if (lit_mode == 0) put_raw(lit0 + 8 * hb, stage, nl);
            litbytes = hb + (lit_mode == 1 ? 1u : nl);
        }
        uint32_t seqbytes = 1;                                        // no sequences: one zero byte (already there)
        if (seq) {
            __syncthreads();                                          // (the literal streams and the sequences section may share a word)
            if (wv == 0) {
                const uint32_t sbts = zstd_sequences_wave(obuf, z, lit0 + 8 * litbytes, reinterpret_cast<const uint32_t *>(stage + ((nl + 3u) & ~3u)), nseq);
                if (lane == 0) S.red[3] = sbts;
            }
            __syncthreads();
            seqbytes = S.red[3];
        }
        bsize = litbytes + seqbytes;
    }
    if (tid == 0) zput_bytes(obuf, z, p0, (last ? 1u : 0u) | ((uint32_t)type << 1) | ((type == 2 ? bsize : blen) << 3), 3);
    z.bitpos = p0 + 8 * (3 + bsize);
    __syncthreads();
}

// One record: payload -> [u64 size][zstd frame] in the record's slot.  STAGED: the payload is parked in HBM (`src`) and passes
// through the LDS `stage` 16 KiB at a time, in place (k_deflate_staged's argument: a block is in LDS before its output is
// written, the slot bound leaves room for the framing); else `src` is the payload in LDS and blocks are taken where they lie.
// Returns the record length (prefix included).
template <bool STAGED>
__device__ __forceinline__ uint32_t zstd_record(DeflShared &S, uint32_t *obuf, uint32_t obuf_words, const uint8_t *src, uint8_t *stage,
                                                uint32_t plen, uint8_t *out, uint32_t dbg = 0, uint32_t split = 0, uint32_t head = 0,
                                                bool seq_on = true) {
    // split != 0 (payload in LDS only): the first block ends there.  An svb-zd payload is `head | key bytes | data bytes`; the key
    // bytes are ~94 % zeros and share nothing with the data bytes, and one Huffman table over both costs 6.6 % of the record
    // (0.9466 -> 0.8838 B/sample on the bench reads: what libzstd's match finder gets out of the key area, 0.878)
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
        // head != 0: the record head in front of the key bytes goes as a raw block (its doubles would push the key block's
        // alphabet past 128 symbols, i.e. into FSE-compressed weights, for nothing)
        const bool raw_head = !STAGED && split && head && done == 0;
        const uint32_t blen = raw_head ? head : !STAGED && split && done < split ? split - done : min(plen - done, (uint32_t)DEFL_BLK);
        const bool last = done + blen == plen;
        uint8_t *blk = const_cast<uint8_t *>(src) + done;   // (not STAGED: the payload in LDS — a block is consumed where it lies)
        if (STAGED) {
            __syncthreads();
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src + done);
            uint4 *d4 = reinterpret_cast<uint4 *>(stage);
            for (uint32_t i = tid; i < (blen + 15) / 16; i += NT) d4[i] = s4[i];
            blk = stage;
            __syncthreads();
        }
        zstd_block(S, B, obuf, obuf_words, blk, blen, last, z, dbg, raw_head, seq_on);
        if (dbg) return z.bitpos >> 3;
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
