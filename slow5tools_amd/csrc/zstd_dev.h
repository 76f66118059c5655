// zstd_dev.h — Zstandard frame decoder, one record per wave64 (the zstd record press).
//
// slow5lib's third record method is libzstd's ZSTD_decompress on the whole record
// (/root/reference/src/misc.c:259 names the method; test/data/exp/one_fast5/exp_1_lossless_zstd*_v0.2.0.blow5 are its fixtures).
// A frame is a chain of blocks; a compressed block is (a) a literals section — raw, RLE or Huffman-coded in 1 or 4
// backward bit streams — and (b) a sequences section: (literal run, match length, offset) triples whose codes come
// out of three interleaved FSE (tANS) state machines.  What the format leaves parallel is little, and this is how
// the wave uses it:
//   - the 4 Huffman streams of a literals section are decoded by 4 lanes, each with its own 64-bit bit container;
//     the decoded literals are parked at the END of the record's own payload slot (the frame's output can never
//     catch up with them: bytes still to come >= literals still unread), so no side buffer exists;
//   - the Huffman decode table (<= 2^11 entries) is filled by all 64 lanes, symbol by symbol;
//   - the FSE state machines are inherently serial: lane 0 walks them, one sequence per step, and hands
//     (literal run, match length, offset) to the wave, which copies with all 64 lanes (matches replicate as
//     out[o+k] = out[o-d + k mod d], sources precede o);
//   - table descriptions (normalised counts, weights) are tiny and decoded by lane 0.
// The CPU twin of this file is oracle/zstd_dec.c (same structure, checked against libzstd itself).
// The optional content checksum (XXH64; slow5lib's one-shot ZSTD_compress never writes it) is verified when a frame carries one (round 5).
// Rejected: dictionaries, skippable frames, reserved bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_common.h"
#include "inflate_dev.h"   // INF_* status codes
#include "zstd_seq_tables.h"   // decode cells of the predefined sequence distributions

namespace s5 {

struct ZstdShared {                  // per wave
    uint16_t huf[2048];              // symbol | bits << 8
    // FSE decode tables, 3 bytes per cell (bits | baseline << 4 in 16 bits, the symbol in 8): the footprint of these tables and
    // of the Huffman table decides how many waves a CU holds, and the decoder is latency bound (13 -> 16 waves per CU)
    uint16_t ll_e[512], ml_e[512], of_e[256];
    uint8_t ll_s[512], ml_s[512], of_s[256];
    uint32_t llsym[36], mlsym[53];   // extra bits | base << 8
    union {
        struct {                     // while tables are built ...
            uint8_t w[256];          // Huffman weights
            uint16_t start[256];     // first table cell of each symbol / FSE spread cells (as bytes)
            int16_t norm[64];
            uint16_t wt_e[64];       // FSE table of the Huffman weights
            uint8_t wt_s[64];
            uint8_t slot[512];       // FSE build: symbol of every spread slot
        };
        struct {                     // ... and while sequences are executed, 64 at a time: what lane 0's chain decoded, and where the literals go
            uint32_t sq_ll[64], sq_ml[64], sq_of[64];   // literal run, match length, offset (repeat offsets resolved; 0: the bit stream was overrun)
            uint32_t sq_cl[65], sq_sh[64];              // literals in front of sequence s (in the batch; [n] = all), match bytes in front of it
        };
    };
    uint32_t x[8];                   // lane 0 -> wave mailbox
};

// the weights scratch of k_zstd_weights (zstd_first_tree_lane below): one record per frame
constexpr uint32_t ZW_NIB = 128;                 // bytes of nibbles in a record of the weights scratch ...
constexpr uint32_t ZW_REC = 144;                 // ... then u32 offset of the description in the frame (0: none), u32 weights sent; 16-byte stride
constexpr int ZW_LANE_DW = 45;                   // LDS dwords per lane (odd: the lanes' tables start in different banks)

// tools/zstd_phases.py (variant build -DS5_ZPROBE): clock ticks a frame's wave spends in each phase, summed over the batch
#ifdef S5_ZPROBE
__device__ unsigned long long g_zprobe[16];
#define ZP_DECL unsigned long long zp_t = __builtin_readcyclecounter(), zp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define ZP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); zp_acc[i] += n_ - zp_t; zp_t = n_; }
#define ZP_FLUSH { if (lane_id() == 0) { for (int i_ = 0; i_ < 10; i_++) atomicAdd(&g_zprobe[i_], zp_acc[i_]); atomicAdd(&g_zprobe[15], 1ull); } }
#else
#define ZP_DECL
#define ZP(i)
#define ZP_FLUSH
#endif
typedef uint64_t __attribute__((aligned(1))) z_u64u;
typedef uint32_t __attribute__((aligned(1))) z_u32u;

struct ZFse { uint16_t *e; uint8_t *s; };                       // one FSE decode table
__device__ __forceinline__ uint32_t zfse_get(const ZFse &t, uint32_t i) {   // symbol | bits << 8 | baseline << 16
    const uint32_t e = t.e[i];
    return (uint32_t)t.s[i] | ((e & 15u) << 8) | ((e >> 4) << 16);
}

__device__ __forceinline__ int z_highbit(uint32_t v) { return 31 - __clz((int)v); }   // v != 0

// backward bit reader: a 64-bit container over the bytes [ptr, ptr + 8) of the stream, `used` bits of it consumed from the top
struct ZBits {
    const uint8_t *base;
    uint32_t ptr, used;
    uint64_t c;
    __device__ __forceinline__ bool init(const uint8_t *p, uint32_t len) {
        base = p; ptr = 0; used = 64; c = 0;
        if (len == 0) return false;
        const uint32_t last = p[len - 1];
        if (last == 0) return false;
        if (len >= 8) { ptr = len - 8; c = *(const z_u64u *)(p + ptr); used = 8 - (uint32_t)z_highbit(last); }
        else {
            for (uint32_t i = 0; i < len; i++) c |= (uint64_t)p[i] << (8 * i);
            used = 64 - 8 * len + 8 - (uint32_t)z_highbit(last);
        }
        return true;
    }
    __device__ __forceinline__ void reload() {
        if (ptr == 0) return;
        uint32_t nb = used >> 3;
        if (nb > ptr) nb = ptr;
        ptr -= nb; used -= 8 * nb;
        c = *(const z_u64u *)(base + ptr);
    }
    __device__ __forceinline__ void need(uint32_t n) { if (used + n > 64) reload(); }
    // n <= 32 bits at the read position; bits before the start of the stream read as zero
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return used < 64 ? (uint32_t)(((c << used) >> 1) >> (63 - n)) : 0u; }
    __device__ __forceinline__ uint32_t get(uint32_t n) { const uint32_t v = peek(n); used += n; return v; }
    __device__ __forceinline__ uint32_t left() const { return 8 * ptr + 64 - used; }     // may wrap when overrun: check overrun() first
    __device__ __forceinline__ bool overrun() const { return used > 64; }
    __device__ __forceinline__ bool done() const { return ptr == 0 && used == 64; }
};

// forward bits of an FSE table description (one lane).  The description is either read where it lies (five byte loads per peek: the
// lane-per-frame pass, whose 64 lanes wait together) or out of LDS words the wave staged (z_stage_desc: lane 0 parsing a sequence table
// waited for memory ~100 times per table — 11 of a libzstd-written frame's 46 ms per 1 M were the three count headers)
__device__ __forceinline__ uint32_t z_fpeek(const uint8_t *p, uint32_t len, uint32_t pos, int n) {
    uint64_t v = 0;
    const uint32_t b = pos >> 3;
    for (uint32_t i = 0; i < 5; i++) if (b + i < len) v |= (uint64_t)p[b + i] << (8 * i);
    return (uint32_t)(v >> (pos & 7)) & ((1u << n) - 1);
}
constexpr uint32_t Z_DESC_STAGE = 256;           // bytes of a description staged (a count header of 53 symbols at table log 9 is < 80)
__device__ __forceinline__ uint32_t z_fpeek_lds(const uint32_t *w, uint32_t len, uint32_t pos, int n) {   // len <= Z_DESC_STAGE; bytes past len are zero
    const uint32_t i = pos >> 5;
    const uint64_t v = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);             // (the stage is followed by as many zero words)
    return (uint32_t)(v >> (pos & 31)) & ((1u << n) - 1);
}
// the first min(len, Z_DESC_STAGE) bytes at p into LDS words, zero behind them (all lanes; a dword per lane where four bytes are there)
__device__ __forceinline__ void z_stage_desc(uint32_t *w, const uint8_t *p, uint32_t len) {
    const uint32_t lane = (uint32_t)lane_id(), n = min(len, Z_DESC_STAGE), at = 4u * lane;
    uint32_t v = 0;
    if (at + 4 <= n) v = *(const z_u32u *)(p + at);
    else for (uint32_t k = 0; at + k < n; k++) v |= (uint32_t)p[at + k] << (8 * k);
    w[lane] = v;
    w[64 + lane] = 0;                            // (a malformed header may be read a few hundred bits past its end: zeros, as in memory form)
    wave_sync();
}

// normalised counts of an FSE table; bytes consumed, 0 on error (oracle/zstd_dec.c fse_read_ncount)
template <bool LDS>
__device__ __noinline__ uint32_t z_ncount_t(const void *src, uint32_t len, int16_t *norm, int *maxsym, int *log, int max_log, int max_sym) {
    const uint8_t *p = static_cast<const uint8_t *>(src);
    const uint32_t *w = static_cast<const uint32_t *>(src);
#define z_fpeek(p_, len_, pos_, n_) (LDS ? z_fpeek_lds(w, len_, pos_, n_) : z_fpeek(p_, len_, pos_, n_))
    uint32_t pos = 4;
    const int al = (int)z_fpeek(p, len, 0, 4) + 5;
    if (al > max_log) return 0;
    *log = al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            int n0 = 0;
            for (;;) {
                const int rep = (int)z_fpeek(p, len, pos, 2);
                pos += 2;
                n0 += rep;
                if (rep != 3 || pos > 8 * len) break;
            }
            if (sym + n0 > max_sym + 1) return 0;
            while (n0-- > 0) norm[sym++] = 0;
            prev0 = 0;
            if (sym > max_sym) break;
        }
        const int maxv = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t bits = z_fpeek(p, len, pos, nbits);
        if ((int)(bits & (uint32_t)(threshold - 1)) < maxv) { count = (int)(bits & (uint32_t)(threshold - 1)); pos += (uint32_t)(nbits - 1); }
        else {
            count = (int)(bits & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= maxv;
            pos += (uint32_t)nbits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = count == 0;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    if (remaining != 1 || sym > max_sym + 1) return 0;
    *maxsym = sym - 1;
    const uint32_t used = (pos + 7) >> 3;
    return used <= len ? used : 0;
#undef z_fpeek
}
__device__ __forceinline__ uint32_t z_ncount(const uint8_t *p, uint32_t len, int16_t *norm, int *maxsym, int *log, int max_log, int max_sym) {
    return z_ncount_t<false>(p, len, norm, maxsym, log, max_log, max_sym);
}

// Decode table of an FSE distribution (oracle/zstd_dec.c fse_build is the serial statement of it).
// Built by the whole wave (norm[0..maxsym] in LDS, maxsym < 64).  Lane s looks after symbol s: the symbols
// with probability "less than one" take the top cells, every other symbol writes itself over its run of spread slots; then
// a lane per spread step t places slot k(t) (= accepted steps before t: a prefix count) at cell (t * step) mod size; and the
// rank of a cell among the cells of its symbol — the decoder's "next state" counter — comes from one ballot per distinct
// symbol of a 64-cell slice, the running counts living in the symbol lanes.  Was ~150 k cycles of lane-0 code per 512-cell table.
__device__ __forceinline__ int z_fse_build_wave(ZstdShared &T, const ZFse &t, int maxsym, int log) {
    const int lane = lane_id();
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint8_t *cell = reinterpret_cast<uint8_t *>(T.start);
    const int mine = lane <= maxsym ? (int)T.norm[lane] : 0;
    const uint64_t lowm = __ballot(mine == -1);
    const int nlow = __popcll(lowm), high = size - 1 - nlow;
    if (mine == -1) cell[size - 1 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(lowm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lowm, 0u))] = (uint8_t)lane;
    const uint32_t cntp = mine > 0 ? (uint32_t)mine : 0u;
    const uint32_t incl = wave_incl_add(cntp);
    if ((int)__builtin_amdgcn_readlane((int)incl, 63) != size - nlow) return -1;
    for (uint32_t k = incl - cntp; k < incl; k++) T.slot[k] = (uint8_t)lane;
    wave_sync();
    uint32_t carry = 0;
    for (int t0 = 0; t0 < size; t0 += 64) {
        const int tt = t0 + lane;
        const int pos = (tt * step) & mask;
        const bool acc = tt < size && pos <= high;
        const uint64_t m = __ballot(acc);
        if (acc) cell[pos] = T.slot[carry + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))];
        carry += (uint32_t)__popcll(m);
    }
    wave_sync();
    uint32_t nextc = mine == -1 ? 1u : cntp;                          // lane s: next state counter of symbol s
    for (int c0 = 0; c0 < size; c0 += 64) {
        const int ci = c0 + lane;
        const uint32_t sym = ci < size ? cell[ci] : 0xFFu;
        uint64_t todo = __ballot(ci < size);
        uint32_t ns = 0;
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const uint32_t sy = (uint32_t)__builtin_amdgcn_readlane((int)sym, first);
            const uint64_t m = __ballot(sym == sy) & todo;
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)nextc, (int)sy);
            if (sym == sy) ns = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if ((uint32_t)lane == sy) nextc += (uint32_t)__popcll(m);
            todo &= ~m;
        }
        if (ci < size) {
            const int nb = log - z_highbit(ns);
            t.s[ci] = (uint8_t)sym;
            t.e[ci] = (uint16_t)((uint32_t)nb | ((((ns << nb) - (uint32_t)size) & 0xFFFu) << 4));
        }
    }
    wave_sync();
    return 0;
}

// extra bits | base << 8 of the literal-length / match-length codes (RFC 8878 3.1.1.3.2.1.1)
__device__ __forceinline__ uint32_t z_ll_sym(uint32_t c) {
    if (c < 16) return c << 8;
    if (c < 20) return 1u | ((16 + 2 * (c - 16)) << 8);
    if (c < 22) return 2u | ((24 + 4 * (c - 20)) << 8);
    if (c < 24) return 3u | ((32 + 8 * (c - 22)) << 8);
    if (c == 24) return 4u | (48u << 8);
    return (c - 19) | ((1u << (c - 19)) << 8);                    // 25: 6 bits, base 64 ... 35: 16 bits, base 65536
}
__device__ __forceinline__ uint32_t z_ml_sym(uint32_t c) {
    if (c < 32) return (c + 3) << 8;
    if (c < 36) return 1u | ((35 + 2 * (c - 32)) << 8);
    if (c < 38) return 2u | ((43 + 4 * (c - 36)) << 8);
    if (c < 40) return 3u | ((51 + 8 * (c - 38)) << 8);
    if (c < 42) return 4u | ((67 + 16 * (c - 40)) << 8);
    if (c == 42) return 5u | (99u << 8);
    return (c - 36) | (((1u << (c - 36)) + 3) << 8);              // 43: 7 bits, base 131 ... 52: 16 bits, base 65539
}

// an FSE-compressed sequence table (all lanes): count header by lane 0 out of staged LDS words, the table by the wave.  A function of its
// own: frames whose tables are predefined never come here, and inlined three times it cost them registers in the loops that matter
#ifdef S5_Z_SEQTAB_INLINE
#define S5_Z_SEQTAB_ATTR __forceinline__
#else
#define S5_Z_SEQTAB_ATTR __noinline__
#endif
__device__ S5_Z_SEQTAB_ATTR int z_seq_table_fse(ZstdShared &T, const uint8_t *p, uint32_t len, uint16_t *te, uint8_t *ts, int *log, int max_log, int max_sym) {
    const int lane = lane_id();
    const ZFse t = {te, ts};
    uint32_t *stage = reinterpret_cast<uint32_t *>(T.slot);        // (the spread slots' storage: the table is built after the header is read)
    z_stage_desc(stage, p, len);
    if (lane == 0) {
        int maxsym = 0, l = 0;
        const uint32_t h = z_ncount_t<true>(stage, min(len, Z_DESC_STAGE), T.norm, &maxsym, &l, max_log, max_sym);
        T.x[4] = h; T.x[5] = (uint32_t)maxsym; T.x[6] = (uint32_t)l;
    }
    wave_sync();
    const uint32_t h = T.x[4];
    if (!h) return -1;
    if (z_fse_build_wave(T, t, (int)T.x[5], (int)T.x[6])) return -1;
    *log = (int)T.x[6];
    return (int)h;
}

// one of the three sequence tables (all lanes; the count header is read by lane 0): bytes of description consumed, -1 on error
__device__ __forceinline__ int z_seq_table(ZstdShared &T, int mode, const uint8_t *p, uint32_t len, const ZFse &t, int *log, const uint32_t *def,
                                           int def_log, int max_log, int max_sym) {
    const int lane = lane_id();
    if (mode == 0) {                                               // predefined: the cells are constants (zstd_seq_tables.h), one per lane
        if (lane < (1 << def_log)) { const uint32_t v = def[lane]; t.s[lane] = (uint8_t)v; t.e[lane] = (uint16_t)(v >> 8); }
        wave_sync();
        *log = def_log;
        return 0;
    }
    if (mode == 1) {
        if (len < 1 || p[0] > max_sym) return -1;
        if (lane == 0) { t.s[0] = p[0]; t.e[0] = 0; }
        wave_sync();
        *log = 0;
        return 1;
    }
    if (mode == 2) return z_seq_table_fse(T, p, len, t.e, t.s, log, max_log, max_sym);
    return *log < 0 ? -1 : 0;
}

// the FSE-compressed form, walked inside the frame's wave (frames of small batches, and every tree of a frame but its first: the weights
// pass does the rest); a function of its own, so that the frames that never come here do not carry its registers
__device__ __noinline__ uint32_t z_huf_weights_fse(ZstdShared &T, const uint8_t *p, uint32_t hb, uint32_t *nsym_out) {
    const int lane = lane_id();
    if (lane == 0) {
        int maxsym = 0, log = 0;
        const uint32_t h = z_ncount(p + 1, hb, T.norm, &maxsym, &log, 6, 12);
        T.x[4] = (h && h < hb) ? h : 0u; T.x[5] = (uint32_t)maxsym; T.x[6] = (uint32_t)log;
    }
    wave_sync();
    const uint32_t h = T.x[4];
    const int log = (int)T.x[6];
    if (!h) return 0;
    const ZFse wt = {T.wt_e, T.wt_s};
    if (z_fse_build_wave(T, wt, (int)T.x[5], log)) return 0;
    {
        // The two interleaved states share one bit cursor: a serial chain of ~255 steps.  Walked with the state in scalar registers:
        // the table (<= 64 cells) sits one cell per lane and the stream (<= 127 bytes) one dword per lane, so a step is a few
        // v_readlane and scalar shifts — no LDS or global round trip on the chain
        const uint8_t *sp = p + 1 + h;
        const uint32_t slen = hb - h;                                   // 1 .. 126
        const uint32_t cell = lane < (1 << log) ? zfse_get(wt, (uint32_t)lane) : 0u;   // symbol | bits << 8 | baseline << 16
        uint32_t sw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t at = 4u * (uint32_t)lane + k; if (at < slen) sw |= (uint32_t)sp[at] << (8 * k); }
        const uint32_t lastb = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp[slen - 1]);
        uint32_t nsym = 0;
        if (lastb) {
            int hi = 8 * (int)(slen - 1) + z_highbit(lastb);            // bits in front of the cursor (uniform)
            if (hi >= 2 * log) {
#define ZW_GET(dst, n) { hi -= (int)(n); const uint32_t wi_ = (uint32_t)hi >> 5; \
                         const uint64_t two_ = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)sw, (int)wi_) | \
                                               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)sw, (int)min(wi_ + 1u, 63u)) << 32); \
                         dst = (uint32_t)(two_ >> ((uint32_t)hi & 31u)) & ((1u << (n)) - 1u); }
                uint32_t s1, s2, t;
                ZW_GET(s1, (uint32_t)log); ZW_GET(s2, (uint32_t)log);
                for (;;) {
                    if (nsym >= 254) { nsym = 0; break; }
                    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)cell, (int)s1), e2 = (uint32_t)__builtin_amdgcn_readlane((int)cell, (int)s2);
                    const uint32_t n1 = (e1 >> 8) & 255u, n2 = (e2 >> 8) & 255u;
                    if (lane == 0) { T.w[nsym] = (uint8_t)e1; T.w[nsym + 1] = (uint8_t)e2; }
                    nsym += 2;
                    if (hi < (int)n1) break;
                    ZW_GET(t, n1); s1 = (e1 >> 16) + t;
                    if (hi < (int)n2) { if (lane == 0) T.w[nsym] = (uint8_t)__builtin_amdgcn_readlane((int)cell, (int)s1); nsym++; break; }
                    ZW_GET(t, n2); s2 = (e2 >> 16) + t;
                }
#undef ZW_GET
                if (nsym > 255) nsym = 0;
            }
        }
        if (lane == 0) T.x[4] = nsym;
    }
    wave_sync();
    if (!T.x[4]) return 0;
    *nsym_out = T.x[4];
    return 1 + hb;
}

// Huffman weights of a tree description: bytes consumed (0 on error); T.w[0..*nsym_out)
// `pre` (or nullptr): what k_zstd_weights left for this frame — if it names the description at frame offset `at`, the weights are there
__device__ __forceinline__ uint32_t z_huf_weights(ZstdShared &T, const uint8_t *p, uint32_t len, uint32_t *nsym_out, const uint8_t *pre, uint32_t at) {   // all lanes
    const int lane = lane_id();
    if (len < 1) return 0;
    const uint32_t hb = p[0];
    if (hb >= 128) {                                              // nibbles, two per byte
        const uint32_t nsym = hb - 127, used = 1 + (nsym + 1) / 2;
        if (used > len) return 0;
        for (uint32_t i = lane; i < nsym; i += 64) T.w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
        wave_sync();
        *nsym_out = nsym;
        return used;
    }
    // FSE-compressed: count header by lane 0, table by the wave, the two interleaved states walked by lane 0
    const uint32_t used = 1 + hb;
    if (used > len || hb < 1) return 0;
    if (pre) {
        const uint32_t *m = reinterpret_cast<const uint32_t *>(pre + ZW_NIB);
        const uint32_t pn = m[1];
        if (m[0] == at && pn) {                                   // four weights per lane, a nibble each
            const uint32_t v = reinterpret_cast<const uint16_t *>(pre)[lane];
#pragma unroll
            for (int k = 0; k < 4; k++) T.w[4 * lane + k] = (uint8_t)((v >> (4 * k)) & 15u);
            wave_sync();
            *nsym_out = pn;
            return used;
        }
    }
    return z_huf_weights_fse(T, p, hb, nsym_out);
}

// ---- the first Huffman tree description of a frame, one frame per LANE (k_zstd_weights, a pass in front of the decoder) ----
// An FSE-compressed tree description is a serial chain of up to 255 dependent table steps (plus the count header and the table behind it):
// walked inside the frame's wave it keeps 63 lanes waiting for ~a quarter of the frame's time (9 + 2 of 43 ms per 1 M frames, round 2's
// cut-offs).  The chain is tiny — <= 127 bytes in, <= 255 nibbles out, a 64-cell table — so 64 frames' chains fit one wave: each lane
// parses its frame's header, first block header and literals header (the same checks as zstd_decode_wave), reads the count header, builds
// the table serially (oracle/zstd_dec.c fse_build) in its own 45 dwords of LDS, walks the two states and leaves the weights, a nibble
// each, with the description's offset in the frame.  Anything irregular leaves "none": the frame's wave then does it all itself and is
// the one that reports the error.  Only the FIRST tree of a frame is served (a record of up to 16 KiB as this library writes it, of up
// to 128 KiB as libzstd does, has one).
__device__ __forceinline__ void zstd_first_tree_lane(uint32_t *lds, const uint8_t *in, uint32_t len, uint8_t *rec) {
    uint16_t *cell = reinterpret_cast<uint16_t *>(lds);                       // 64 cells: symbol | bits << 4 | baseline << 8
    int16_t *norm = reinterpret_cast<int16_t *>(lds + 32);                    // 13 counts
    uint8_t *next = reinterpret_cast<uint8_t *>(lds + 39);                    // 13 next-state counters
    uint32_t *out32 = reinterpret_cast<uint32_t *>(rec);
    uint32_t at = 0, nsym = 0;
    do {
        if (len < 6 || in[0] != 0x28 || in[1] != 0xB5 || in[2] != 0x2F || in[3] != 0xFD) break;
        const uint32_t fhd = in[4];
        if ((fhd & 8) || (fhd & 3)) break;
        const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1;
        uint32_t p = 5 + (single ? 0u : 1u) + (fcs_flag == 0 ? single : fcs_flag == 1 ? 2u : fcs_flag == 2 ? 4u : 8u);
        if (p + 3 > len) break;
        const uint32_t bh = in[p] | (in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
        p += 3;
        const uint32_t bsize = bh >> 3;
        if (((bh >> 1) & 3) != 2 || bsize < 5 || bsize > 128 * 1024 || p + bsize > len) break;
        const uint8_t *b = in + p;
        if ((b[0] & 3) != 2) break;                                              // literals with a tree of their own
        const uint32_t sf = (b[0] >> 2) & 3;
        const uint64_t v = (uint64_t)b[0] | ((uint64_t)b[1] << 8) | ((uint64_t)b[2] << 16) | ((uint64_t)b[3] << 24) | ((uint64_t)b[4] << 32);
        const uint32_t q = sf < 2 ? 3u : sf == 2 ? 4u : 5u;
        const uint32_t csize = sf < 2 ? (uint32_t)(v >> 14) & 0x3FF : sf == 2 ? (uint32_t)(v >> 18) & 0x3FFF : (uint32_t)(v >> 22) & 0x3FFFF;
        if (q + csize > bsize || csize < 1) break;
        const uint8_t *t = b + q;
        const uint32_t hb = t[0];
        if (hb >= 128 || hb < 1 || 1 + hb > csize) break;                       // (nibble descriptions cost the frame's wave nothing)
        int maxsym = 0, log = 0;
        const uint32_t h = z_ncount(t + 1, hb, norm, &maxsym, &log, 6, 12);
        if (!h || h >= hb) break;
        const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
        int high = size - 1;
        for (int sy = 0; sy <= maxsym; sy++) {
            const int n = norm[sy];
            if (n == -1) { cell[high--] = (uint16_t)sy; next[sy] = 1; } else next[sy] = (uint8_t)n;
        }
        int pos = 0;
        for (int sy = 0; sy <= maxsym; sy++)
            for (int i = 0; i < (int)norm[sy]; i++) {
                cell[pos] = (uint16_t)sy;
                do { pos = (pos + step) & mask; } while (pos > high);
            }
        if (pos != 0) break;
        for (int i = 0; i < size; i++) {
            const uint32_t sy = cell[i];
            const uint32_t ns = next[sy]++;
            const uint32_t nb = (uint32_t)(log - z_highbit(ns));
            cell[i] = (uint16_t)(sy | (nb << 4) | ((((ns << nb) - (uint32_t)size) & 63u) << 8));
        }
        ZBits br;
        if (!br.init(t + 1 + h, hb - h)) break;
        if (br.left() < 2u * (uint32_t)log) break;
        br.need(12);
        uint32_t s1 = br.get((uint32_t)log), s2 = br.get((uint32_t)log);
        uint32_t acc = 0, n = 0;
        bool ok = true;
#define ZW_EMIT(sym_) { acc |= ((sym_) & 15u) << (4u * (n & 7u)); if ((n & 7u) == 7u) { out32[n >> 3] = acc; acc = 0; } n++; }
        for (;;) {
            if (n >= 254) { ok = false; break; }
            const uint32_t e1 = cell[s1], e2 = cell[s2];
            const uint32_t n1 = (e1 >> 4) & 15u, n2 = (e2 >> 4) & 15u;
            ZW_EMIT(e1); ZW_EMIT(e2);
            br.need(12);
            if (br.left() < n1) break;
            s1 = (e1 >> 8) + br.get(n1);
            if (br.left() < n2) { const uint32_t e = cell[s1]; ZW_EMIT(e); break; }
            s2 = (e2 >> 8) + br.get(n2);
        }
#undef ZW_EMIT
        if (!ok || n > 255) break;
        if (n & 7u) out32[n >> 3] = acc;
        at = (uint32_t)(t - in);
        nsym = n;
    } while (false);
    out32[ZW_NIB / 4] = nsym ? at : 0u;
    out32[ZW_NIB / 4 + 1] = nsym;
}

// From the weights T.w[0..nsym) to the table layout, all 64 lanes: the implied last weight, the code length limit, and
// T.start[i] = first cell of symbol i (cells ascend by weight, symbol order inside a weight).  Lane l looks after symbols
// l, l + 64, l + 128, l + 192; counts per weight are ballots, ranks inside a weight the set bits below the lane.
// Returns the number of symbols (0: not a valid description); *maxbits_out = longest code.
__device__ __forceinline__ uint32_t z_huf_ranks(ZstdShared &T, uint32_t nsym, int *maxbits_out) {
    const int lane = lane_id();
    uint32_t w[4], part = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = (uint32_t)lane + 64u * j;
        w[j] = i < nsym ? T.w[i] : 0u;
        if (w[j] > 11) { bad = true; w[j] = 0; }
        if (w[j]) part += 1u << (w[j] - 1);
    }
    if (__ballot(bad)) return 0;
    const uint32_t sum = wave_sum(part);
    if (sum == 0) return 0;
    const int maxbits = z_highbit(sum) + 1;
    if (maxbits > 11) return 0;
    const uint32_t rest = (1u << maxbits) - sum;
    if (rest & (rest - 1)) return 0;
    const uint32_t wl = (uint32_t)z_highbit(rest) + 1;             // the weight of the last symbol is implied
#pragma unroll
    for (int j = 0; j < 4; j++) if ((uint32_t)lane + 64u * j == nsym) { w[j] = wl; T.w[nsym] = (uint8_t)wl; }
    nsym++;
    uint32_t at = 0;                                               // first cell of the weight class at hand
#pragma unroll
    for (uint32_t r = 1; r <= 11; r++) {
        uint32_t seen = 0;                                         // symbols of this weight in the slices before
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t m = __ballot(w[j] == r);
            if (w[j] == r) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                T.start[lane + 64 * j] = (uint16_t)(at + ((seen + below) << (r - 1)));
            }
            seen += (uint32_t)__popcll(m);
        }
        if (r == 1 && (seen < 2 || (seen & 1))) return 0;
        at += seen << (r - 1);
    }
    wave_sync();
    *maxbits_out = maxbits;
    return nsym;
}

// one Huffman stream on the calling lane: n symbols to dst; false on a malformed stream
__device__ __forceinline__ bool z_huf_stream(const uint16_t *huf, uint32_t L, const uint8_t *p, uint32_t len, uint8_t *dst, uint32_t n) {
    ZBits b;
    if (!b.init(p, len)) return false;
    uint32_t i = 0;
    const uint32_t sh = 64u - L;
    for (; i + 4 <= n; i += 4) {                                   // 4 symbols <= 44 bits per refill
        b.need(44);
        // the unread bits on top of one 64-bit word: a symbol is a shift for the index and a shift to drop its code
        // (bits before the start of the stream read as zero, as peek() has it)
        uint64_t w = b.used < 64 ? b.c << b.used : 0ull;
        uint32_t v = 0, took = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t e = huf[(uint32_t)(w >> sh)];
            const uint32_t nb = e >> 8;
            w <<= nb;
            took += nb;
            v |= (e & 255u) << (8 * q);
        }
        b.used += took;
        *(z_u32u *)(dst + i) = v;
    }
    for (; i < n; i++) {
        b.need(11);
        const uint32_t e = huf[b.peek(L)];
        b.used += e >> 8;
        dst[i] = (uint8_t)e;
    }
    return !b.overrun() && b.done();
}

// Huffman literals by the WHOLE wave (the serial form above keeps 4 of 64 lanes busy for a thousand dependent steps each): every
// stream gets G = 64 / streams lanes, each lane a G-th of the stream's bits.  A Huffman stream falls into step with the true code
// boundaries after a few codes wherever the decoder is dropped (the same observation as inflate_par_dev.h): every lane decodes
// the symbols that start in its bit range and reports where its last one ends = where the next lane's range REALLY starts; the
// pass is repeated with the corrected starts until none moves (lane k of a group is certainly right after pass k: G passes
// bound it), the symbol counts give the output offsets, one more pass writes the bytes.  zstd reads its streams from the last bit
// down: positions are "bits still in front of the reader" (hi), falling.  All 64 lanes call; p / sl / dst / ns are the lane's
// stream's.  false: malformed (symbol count or end position wrong).
__device__ __forceinline__ void z_bits_at(ZBits &b, const uint8_t *p, int hi) {          // reader with hi bits in front of it
    b.base = p;
    if (hi <= 0) { b.ptr = 0; b.used = 64; b.c = 0; return; }
    const uint32_t top = (uint32_t)(hi - 1) >> 3;                                        // byte that holds the next bit
    b.ptr = top >= 7 ? top - 7 : 0;
    b.c = *(const z_u64u *)(p + b.ptr);                                                  // (may reach 7 bytes past the stream: readable, skipped by `used`)
    b.used = 8 * b.ptr + 64 - (uint32_t)hi;
}
// the same reader with the NEXT eight bytes already on their way: a refill shifts them in and asks for the eight below — the
// global load is off the lane's critical path (what is left on it is the table lookup per symbol)
struct ZPre {
    const uint8_t *base;
    uint32_t ptr, used;
    uint64_t c, below;
    __device__ __forceinline__ void at(const uint8_t *p, int hi) {
        base = p;
        if (hi <= 0) { ptr = 0; used = 64; c = 0; below = 0; return; }
        const uint32_t top = (uint32_t)(hi - 1) >> 3;
        ptr = top >= 7 ? top - 7 : 0;
        c = *(const z_u64u *)(p + ptr);
        below = *(const z_u64u *)(p + ptr - 8);                      // (bytes in front of a stream are the block's own: readable, never used past ptr = 0)
        used = 8 * ptr + 64 - (uint32_t)hi;
    }
    __device__ __forceinline__ void refill() {                        // afterwards used <= 7 (or the stream's first byte is in the container)
        const uint32_t nb = min(used >> 3, ptr);
        if (nb) {
            c = nb == 8 ? below : (c << (8 * nb)) | (below >> (64 - 8 * nb));
            ptr -= nb; used -= 8 * nb;
            below = *(const z_u64u *)(base + ptr - 8);
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return used < 64 ? (uint32_t)(((c << used) >> 1) >> (63 - n)) : 0u; }
};
// one lane's range of its stream, from the reader position `st` (bits in front of it) down to seg_lo: symbols that START above seg_lo are
// the lane's; cross = where the last one ends.  The unread bits ride on top of one 64-bit word: a symbol is one 32-bit shift for the table
// index and one 64-bit shift to drop its code; four symbols (<= 44 bits) per refill, and every lane refills at the same place — a lane
// that refilled when it ran dry would make the wave wait for a global load in nearly every step.  WRITE: the symbols go to q, four at a
// time as one (unaligned) dword while the lane has four.
#ifndef S5_ZH_TAIL
#define S5_ZH_TAIL 128           // bits of a range the first pass decodes to find where the next range really starts
#endif
template <bool WRITE>
__device__ __forceinline__ void z_huf_range(const uint16_t *huf, uint32_t L, const uint8_t *p, int st, int seg_lo, int &cross, uint32_t &cnt, uint8_t *q) {
    ZPre b;
    b.at(p, st);
    int hi = st;
    const uint32_t sh = 32u - L;
    cnt = 0;
    while (__ballot(hi > seg_lo)) {
        b.refill();
        uint64_t w = b.used < 64 ? b.c << b.used : 0ull;               // (bits before the start of the stream read as zero)
        uint32_t took = 0, v = 0, nv = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const uint32_t e = huf[(uint32_t)(w >> 32) >> sh];
            const uint32_t nb = max(e >> 8, 1u);                         // (a complete table has no empty cell; never stall on one)
            if (hi > seg_lo) {
                took += nb;
                hi -= (int)nb;
                if (WRITE) v |= (e & 255u) << (8 * q4);
                nv++;
            }
            w <<= nb;
        }
        b.used += took;
        cnt += nv;
        if (WRITE) {
            if (nv == 4) *(z_u32u *)q = v;
            else for (uint32_t k = 0; k < nv; k++) q[k] = (uint8_t)(v >> (8 * k));
            q += nv;
        }
    }
    cross = hi;
}
__device__ __forceinline__ bool z_huf_streams_par(const uint16_t *huf, uint32_t L, const uint8_t *p, uint32_t sl, uint8_t *dst, uint32_t ns, int G) {
    const int lane = lane_id(), j = lane & (G - 1), g0 = lane & ~(G - 1);
    const uint32_t last = sl ? p[sl - 1] : 0u;
    if (__ballot(last == 0u)) return false;
    const int Bs = 8 * (int)(sl - 1) + z_highbit(last);                                  // bits below the end mark
    const int B = max((Bs + G - 1) / G, 1);
    const int seg_hi = Bs - j * B, seg_lo = max(seg_hi - B, 0);                          // my symbols start at hi in (seg_lo, seg_hi]
    int st, cross = seg_hi;
    uint32_t cnt = 0;
    // first only the tail of every range, from wherever: after a few codes the decoder is in step, and where it leaves the range is
    // (nearly always) where the next lane really starts — a full pass from guessed starts would be decoded for that alone
    z_huf_range<false>(huf, L, p, min(seg_hi, seg_lo + S5_ZH_TAIL), seg_lo, cross, cnt, nullptr);
    st = __shfl(cross, max(lane - 1, 0));
    if (j == 0) st = Bs;
    for (int pass = 0; pass <= G; pass++) {
        z_huf_range<false>(huf, L, p, st, seg_lo, cross, cnt, nullptr);
        int nst = __shfl(cross, max(lane - 1, 0));
        if (j == 0) nst = Bs;
        const bool moved = nst != st;
        st = nst;
#ifdef S5_ZPROBE
        if (lane == 0) atomicAdd(&g_zprobe[10], 1ull);               // full passes of the literal streams
#endif
        if (!__ballot(moved)) break;
    }
    // the group's symbols must be exactly ns and the last one must end at the stream's first bit
    const uint32_t incl = wave_incl_add(cnt);
    const uint32_t prevg = (uint32_t)__shfl((int)incl, max(g0 - 1, 0));                  // (every lane takes part: a shuffle reads nothing from a lane that sits it out)
    const uint32_t before = g0 ? prevg : 0u;
    const uint32_t total = (uint32_t)__shfl((int)incl, g0 + G - 1) - before;
    const int fin = __shfl(cross, g0 + G - 1);
    if (__ballot(total != ns || fin != 0)) return false;
    uint32_t cnt2;
    int cross2;
    z_huf_range<true>(huf, L, p, st, seg_lo, cross2, cnt2, dst + (incl - cnt - before));
    return true;
}

// n bytes from s to d by the whole wave, d <= s when the two ranges overlap (literals parked behind the output are moved down in
// place): 16 bytes per lane and step — all loads of a step are issued before its stores, and later steps read further up
__device__ __forceinline__ void z_wave_move_down(uint8_t *d, const uint8_t *s, uint32_t n) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t body = n & ~15u;
    for (uint32_t k = 16u * lane; k < body; k += 1024u) {
        const u4 v = *reinterpret_cast<const u4 *>(s + k);
        *reinterpret_cast<u4 *>(d + k) = v;
    }
    wave_sync();
    uint8_t b = 0;
    if (lane < n - body) b = s[body + lane];
    wave_sync();
    if (lane < n - body) d[body + lane] = b;
}

// XXH64 (seed 0) of n bytes: the content checksum of a frame that carries one is its low 32 bits (RFC 8878 3.1.1).  slow5lib's one-shot
// ZSTD_compress never sets the flag, so this is off the hot path: the four accumulators of the 32-byte stripes on lanes 0-3, the merge and
// the tail on lane 0.  Uniform result.
__device__ __forceinline__ uint64_t z_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t z_xxh64_wave(const uint8_t *p, uint64_t n) {
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    typedef uint64_t u64u __attribute__((aligned(1)));
    typedef uint32_t u32u __attribute__((aligned(1)));
    const int lane = lane_id();
    auto round = [&](uint64_t acc, uint64_t in) { return z_rotl64(acc + in * P2, 31) * P1; };
    uint64_t h;
    uint64_t i = 0;
    if (n >= 32) {
        uint64_t v = lane == 0 ? P1 + P2 : lane == 1 ? P2 : lane == 2 ? 0ull : 0ull - P1;
        for (; i + 32 <= n; i += 32)
            if (lane < 4) v = round(v, *reinterpret_cast<const u64u *>(p + i + 8 * lane));
        const uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        h = z_rotl64(v1, 1) + z_rotl64(v2, 7) + z_rotl64(v3, 12) + z_rotl64(v4, 18);
        h = (h ^ round(0, v1)) * P1 + P4;
        h = (h ^ round(0, v2)) * P1 + P4;
        h = (h ^ round(0, v3)) * P1 + P4;
        h = (h ^ round(0, v4)) * P1 + P4;
    } else h = P5;
    h += n;
    if (lane == 0) {
        for (; i + 8 <= n; i += 8) h = z_rotl64(h ^ round(0, *reinterpret_cast<const u64u *>(p + i)), 27) * P1 + P4;
        if (i + 4 <= n) { h = z_rotl64(h ^ ((uint64_t)*reinterpret_cast<const u32u *>(p + i) * P1), 23) * P2 + P3; i += 4; }
        for (; i < n; i++) h = z_rotl64(h ^ ((uint64_t)p[i] * P5), 11) * P1;
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    }
    return __shfl(h, 0);
}

// One frame -> out (olen bytes).  INF_OK, or INF_ERR_HEADER / DATA / TRUNC / OVERFLOW (olen = bytes needed when the frame says);
// INF_ERR_ADLER when the frame carries a content checksum and the decoded bytes do not hash to it.
// `pre`: this frame's record of the weights scratch (k_zstd_weights ran over the batch), or nullptr
__device__ __forceinline__ int zstd_decode_wave(ZstdShared &T, const uint8_t *in, uint32_t len, uint8_t *out, uint32_t cap, uint32_t *olen_out, const uint8_t *pre = nullptr) {
    const int lane = lane_id();
    *olen_out = 0;
    ZP_DECL
    if (len < 6) return INF_ERR_TRUNC;
    if (in[0] != 0x28 || in[1] != 0xB5 || in[2] != 0x2F || in[3] != 0xFD) return INF_ERR_HEADER;
    uint32_t p = 5;
    const uint32_t fhd = in[4];
    const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1;
    if ((fhd & 8) || (fhd & 3)) return INF_ERR_HEADER;            // reserved bit / dictionary
    if (!single) p++;
    const uint32_t fcs_bytes = fcs_flag == 0 ? single : fcs_flag == 1 ? 2u : fcs_flag == 2 ? 4u : 8u;
    if (p + fcs_bytes > len) return INF_ERR_TRUNC;
    uint64_t fcs = 0;
    for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)in[p + i] << (8 * i);
    if (fcs_flag == 1) fcs += 256;
    p += fcs_bytes;
    if (fcs_bytes && fcs > cap) { *olen_out = fcs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)fcs; return INF_ERR_OVERFLOW; }
    const uint32_t E = fcs_bytes ? (uint32_t)fcs : cap;           // end of the output region: Huffman literals park below it
    if (lane < 36) T.llsym[lane] = z_ll_sym((uint32_t)lane);
    if (lane < 53) T.mlsym[lane] = z_ml_sym((uint32_t)lane);
    const ZFse tll = {T.ll_e, T.ll_s}, tml = {T.ml_e, T.ml_s}, tof = {T.of_e, T.of_s};
    int huf_log = 0, ll_log = -1, of_log = -1, ml_log = -1;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;                        // lane 0's copy is the live one
    uint32_t o = 0;
    int status = INF_OK;
    bool last = false;
    while (!last) {
        if (p + 3 > len) { status = INF_ERR_TRUNC; break; }
        const uint32_t bh = in[p] | (in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
        p += 3;
        last = bh & 1;
        const uint32_t type = (bh >> 1) & 3, bsize = bh >> 3;
        if (type == 3 || bsize > 128 * 1024) { status = INF_ERR_DATA; break; }
        if (type < 2) {                                           // raw / RLE block
            const uint32_t nin = type == 0 ? bsize : 1u;
            if (p + nin > len) { status = INF_ERR_TRUNC; break; }
            if (o + bsize > E) { status = fcs_bytes ? INF_ERR_DATA : INF_ERR_OVERFLOW; break; }
            for (uint32_t k = lane; k < bsize; k += 64) out[o + k] = in[p + (type == 0 ? k : 0u)];
            o += bsize; p += nin;
            wave_sync();
            continue;
        }
        if (p + bsize > len) { status = INF_ERR_TRUNC; break; }
        if (bsize < 2) { status = INF_ERR_DATA; break; }
        const uint8_t *b = in + p;
        const uint32_t bend = bsize;                              // offsets below are relative to b
        p += bsize;
        // ---- literals section ----
        const uint32_t ltype = b[0] & 3, sf = (b[0] >> 2) & 3;
        uint32_t lsize, csize = 0, q, streams = 1;
        if (ltype < 2) {
            if (sf == 0 || sf == 2) { lsize = b[0] >> 3; q = 1; }
            else if (sf == 1) { lsize = (b[0] >> 4) | ((uint32_t)b[1] << 4); q = 2; }
            else { if (bsize < 3) { status = INF_ERR_DATA; break; } lsize = (b[0] >> 4) | ((uint32_t)b[1] << 4) | ((uint32_t)b[2] << 12); q = 3; }
        } else {
            if (bsize < 5) { status = INF_ERR_DATA; break; }
            const uint64_t v = (uint64_t)b[0] | ((uint64_t)b[1] << 8) | ((uint64_t)b[2] << 16) | ((uint64_t)b[3] << 24) | ((uint64_t)b[4] << 32);
            if (sf < 2) { lsize = (uint32_t)(v >> 4) & 0x3FF; csize = (uint32_t)(v >> 14) & 0x3FF; q = 3; streams = sf ? 4 : 1; }
            else if (sf == 2) { lsize = (uint32_t)(v >> 4) & 0x3FFF; csize = (uint32_t)(v >> 18) & 0x3FFF; q = 4; streams = 4; }
            else { lsize = (uint32_t)(v >> 4) & 0x3FFFF; csize = (uint32_t)(v >> 22) & 0x3FFFF; q = 5; streams = 4; }
        }
        if (lsize > 128 * 1024) { status = INF_ERR_DATA; break; }
        ZP(0)
#if defined(S5_ZCUT) && S5_ZCUT == 1   // variant builds (tools/zstd_cuts.sh): where a frame's time goes, by leaving early
        break;
#endif
        const uint8_t *lit = nullptr;                             // where literal li lives (raw: in the input; Huffman: parked in `out`)
        uint32_t lit_fill = 0;
        bool lit_parked = false;
        if (ltype == 0) { if (q + lsize > bend) { status = INF_ERR_DATA; break; } lit = b + q; q += lsize; }
        else if (ltype == 1) { if (q + 1 > bend) { status = INF_ERR_DATA; break; } lit_fill = b[q]; q += 1; }
        else {
            if (q + csize > bend) { status = INF_ERR_DATA; break; }
            if (lsize > E || o > E - lsize) { status = fcs_bytes ? INF_ERR_DATA : INF_ERR_OVERFLOW; break; }
            uint32_t c = q;
            const uint32_t cend = q + csize;
            q = cend;
            if (ltype == 2) {
                uint32_t nw = 0;
                const uint32_t u = z_huf_weights(T, b + c, cend - c, &nw, pre, (uint32_t)(b + c - in));
                if (!u) { status = INF_ERR_DATA; break; }
                const uint32_t nsym = z_huf_ranks(T, nw, &huf_log);
                if (!nsym) { huf_log = 0; status = INF_ERR_DATA; break; }
                c += u;
                // the table: a lane fills the cells of its own symbols when they are few (weights up to 5: at most 16 cells);
                // the handful of frequent symbols with hundreds of cells each are filled by the whole wave, one after the other
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t i = (uint32_t)lane + 64u * j;
                    const uint32_t wgt = i < nsym ? T.w[i] : 0u;
                    const uint32_t at = i < nsym ? T.start[i] : 0u;
                    const uint32_t e = i | ((uint32_t)(huf_log + 1 - (int)wgt) << 8);
                    if (wgt && wgt <= 5) { const uint32_t span = 1u << (wgt - 1); for (uint32_t k = 0; k < span; k++) T.huf[at + k] = (uint16_t)e; }
                    uint64_t big = __ballot(wgt >= 6);
                    while (big) {
                        const int l = __ffsll((long long)big) - 1;
                        const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)wgt, l), ba = (uint32_t)__builtin_amdgcn_readlane((int)at, l);
                        const uint32_t be = (uint32_t)__builtin_amdgcn_readlane((int)e, l), span = 1u << (bw - 1);
                        for (uint32_t k = lane; k < span; k += 64) T.huf[ba + k] = (uint16_t)be;
                        big &= big - 1;
                    }
                }
                wave_sync();
            } else if (!huf_log) { status = INF_ERR_DATA; break; }
            ZP(1)
#if defined(S5_ZCUT) && S5_ZCUT == 2
            break;
#endif
            uint8_t *park = out + (E - lsize);
            bool ok = true;
            if (streams == 1) {
                if (lane == 0) ok = z_huf_stream(T.huf, (uint32_t)huf_log, b + c, cend - c, park, lsize);
            } else {
                if (cend - c < 6) { status = INF_ERR_DATA; break; }
                const uint32_t s1 = b[c] | (b[c + 1] << 8), s2 = b[c + 2] | (b[c + 3] << 8), s3 = b[c + 4] | (b[c + 5] << 8);
                c += 6;
                const uint32_t per = (lsize + 3) / 4;
                if (s1 + s2 + s3 > cend - c || 3 * per > lsize) { status = INF_ERR_DATA; break; }
                if (lsize >= 512) {                                  // 16 lanes per stream (z_huf_streams_par)
                    const int k = lane >> 4;
                    const uint32_t from = k == 0 ? 0 : k == 1 ? s1 : k == 2 ? s1 + s2 : s1 + s2 + s3;
                    const uint32_t sl = k == 0 ? s1 : k == 1 ? s2 : k == 2 ? s3 : cend - c - s1 - s2 - s3;
                    ok = z_huf_streams_par(T.huf, (uint32_t)huf_log, b + c + from, sl, park + (uint32_t)k * per, k == 3 ? lsize - 3 * per : per, 16);
                } else if (lane < 4) {
                    const uint32_t from = lane == 0 ? 0 : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
                    const uint32_t sl = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : cend - c - s1 - s2 - s3;
                    ok = z_huf_stream(T.huf, (uint32_t)huf_log, b + c + from, sl, park + (uint32_t)lane * per, lane == 3 ? lsize - 3 * per : per);
                }
            }
            if (__ballot(!ok)) { status = INF_ERR_DATA; break; }
            wave_sync();
            lit = park;
            lit_parked = true;
        }
        ZP(2)
#if defined(S5_ZCUT) && S5_ZCUT == 3
        break;
#endif
        // ---- sequences section ----
        if (q >= bend) { status = INF_ERR_DATA; break; }
        uint32_t nseq = b[q++];
        if (nseq >= 128) {
            if (nseq == 255) { if (q + 2 > bend) { status = INF_ERR_DATA; break; } nseq = b[q] + (b[q + 1] << 8) + 0x7F00; q += 2; }
            else { if (q + 1 > bend) { status = INF_ERR_DATA; break; } nseq = ((nseq - 128) << 8) + b[q]; q += 1; }
        }
        uint32_t li = 0;
        if (nseq) {
            if (q >= bend) { status = INF_ERR_DATA; break; }
            const uint32_t modes = b[q++];
            if (modes & 3) { status = INF_ERR_DATA; break; }
            {
                int u, bad = 0;
                uint32_t qq = q;
                if ((u = z_seq_table(T, (int)(modes >> 6), b + qq, bend - qq, tll, &ll_log, ZSEQ_LL_DEC, 6, 9, 35)) < 0) bad = 1; else qq += (uint32_t)u;
                if (!bad) { if ((u = z_seq_table(T, (int)((modes >> 4) & 3), b + qq, bend - qq, tof, &of_log, ZSEQ_OF_DEC, 5, 8, 31)) < 0) bad = 1; else qq += (uint32_t)u; }
                if (!bad) { if ((u = z_seq_table(T, (int)((modes >> 2) & 3), b + qq, bend - qq, tml, &ml_log, ZSEQ_ML_DEC, 6, 9, 52)) < 0) bad = 1; else qq += (uint32_t)u; }
                if (lane == 0) { T.x[0] = (uint32_t)bad; T.x[1] = qq; }
            }
            wave_sync();
            if (T.x[0]) { status = INF_ERR_DATA; break; }
            q = T.x[1];
            ZP(3)
#if defined(S5_ZCUT) && S5_ZCUT == 4
            break;
#endif
            ZBits br;
            br.base = b; br.ptr = 0; br.used = 64; br.c = 0;
            uint32_t sl = 0, so = 0, sm = 0;
            bool ok = true;
            if (lane == 0) {
                ok = q < bend && br.init(b + q, bend - q);
                if (ok) {
                    sl = br.get((uint32_t)ll_log);
                    br.need(32);
                    so = br.get((uint32_t)of_log);
                    sm = br.get((uint32_t)ml_log);
                }
            }
            if (__shfl((int)ok, 0) == 0) { status = INF_ERR_DATA; break; }
            // Up to 64 sequences at a time.  Executed one by one, a sequence costs the wave two dependent trips to memory (its literals, then
            // its match): ~50 sequences a frame were 16 of a frame's 29 ms.  So lane 0 walks the chain for the whole batch first (into LDS);
            // then ALL the batch's literals move — one stream of bytes cut at the sequence boundaries, 16 bytes per lane and step, ascending
            // (parked literals only ever move DOWN, and by less the later they come: every step loads before it stores and later steps read
            // further up, as in z_wave_move_down); then every lane copies ITS match as soon as the matches that reach into its source are
            // done (the rule of inflate_par_dev.h's waiting matches: destinations ascend, two binary searches over the lanes give the
            // range), long ones by the whole wave.
            for (uint32_t s0 = 0; s0 < nseq && status == INF_OK; s0 += 64) {
                const uint32_t nb = min(nseq - s0, 64u);
                ZP(4)
                if (lane == 0) {
                    for (uint32_t s = 0; s < nb; s++) {
                        uint32_t llen, mlen, offset;
                        const uint32_t eo = zfse_get(tof, so), em = zfse_get(tml, sm), el = zfse_get(tll, sl);
                        const uint32_t ofc = eo & 255, mls = T.mlsym[em & 255], lls = T.llsym[el & 255];
                        br.need(ofc);
                        const uint32_t ofv = (1u << ofc) + br.get(ofc);
                        br.need(32);
                        mlen = (mls >> 8) + br.get(mls & 255);
                        llen = (lls >> 8) + br.get(lls & 255);
                        if (ofv > 3) { offset = ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                        else {
                            const uint32_t idx = ofv - 1 + (llen == 0);
                            if (idx == 0) offset = rep0;
                            else {
                                offset = idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
                                if (idx > 1) rep2 = rep1;
                                rep1 = rep0;
                                rep0 = offset;
                            }
                        }
                        if (s0 + s + 1 < nseq) {
                            br.need(27);
                            sl = (el >> 16) + br.get((el >> 8) & 255);
                            sm = (em >> 16) + br.get((em >> 8) & 255);
                            so = (eo >> 16) + br.get((eo >> 8) & 255);
                        }
                        if (br.overrun()) offset = 0;                  // reported as corrupt below
                        T.sq_ll[s] = llen; T.sq_ml[s] = mlen; T.sq_of[s] = offset;
                    }
                }
                wave_sync();
                ZP(5)
                const bool have = (uint32_t)lane < nb;
                const uint32_t ll = have ? T.sq_ll[lane] : 0u, ml = have ? T.sq_ml[lane] : 0u, of = have ? T.sq_of[lane] : 1u;
                const uint32_t cl_in = wave_incl_add(ll), cm_in = wave_incl_add(ml);
                const uint32_t cl = cl_in - ll, cm = cm_in - ml;
                const uint32_t LB = (uint32_t)__builtin_amdgcn_readlane((int)cl_in, 63), MB = (uint32_t)__builtin_amdgcn_readlane((int)cm_in, 63);
                const uint64_t dmat = (uint64_t)o + cl_in + cm;                // where my match goes (my literals end there)
                {   // the checks of the one-by-one form, for the first sequence that fails one
                    int bad = 0;
                    if (have) {
                        const uint64_t lim = lit_parked ? (uint64_t)E - lsize + li + cl_in : (uint64_t)E;   // not into parked literals still unread, nor past the slot
                        if ((uint64_t)li + cl_in > lsize || of == 0 || (uint64_t)of > dmat) bad = INF_ERR_DATA;
                        else if (dmat + ml > lim) bad = fcs_bytes ? INF_ERR_DATA : INF_ERR_OVERFLOW;
                    }
                    const uint64_t bm = __ballot(bad != 0);
                    if (bm) { status = __builtin_amdgcn_readlane(bad, __ffsll((long long)bm) - 1); break; }
                }
                T.sq_cl[lane] = cl; T.sq_sh[lane] = cm;
                if (lane == 0) T.sq_cl[64] = LB;
                wave_sync();
#if defined(S5_ZCUT) && S5_ZCUT == 5   // the chains and the bookkeeping without the copies
                o += LB + MB; li += LB;
                continue;
#endif
                ZP(6)
                // ---- the batch's literals ----
                for (uint32_t k0 = 16u * (uint32_t)lane; k0 < LB; k0 += 1024u) {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
                    const uint32_t n = min(16u, LB - k0);
                    uint64_t vlo = (uint64_t)lit_fill * 0x0101010101010101ull, vhi = vlo;
                    u4 v = {(uint32_t)vlo, (uint32_t)vlo, (uint32_t)vlo, (uint32_t)vlo};
                    if (lit) {
                        if (n == 16) { v = *reinterpret_cast<const u4 *>(lit + li + k0); vlo = (uint64_t)v.x | ((uint64_t)v.y << 32); vhi = (uint64_t)v.z | ((uint64_t)v.w << 32); }
                        else {
                            vlo = 0; vhi = 0;
#pragma unroll
                            for (uint32_t j = 0; j < 15; j++) if (j < n) { const uint64_t x = lit[li + k0 + j]; if (j < 8) vlo |= x << (8 * j); else vhi |= x << (8 * (j - 8)); }
                        }
                    }
                    // every lane of the wave has LOADED its 16 bytes before any lane stores (parked literals move down inside `out`: one lane's
                    // destination can be another lane's source); the barrier is a convergence point, so the two load shapes above cannot be
                    // merged with the two store shapes below into load-store / load-store
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    int lo = 0, hi = (int)nb - 1;                               // the last sequence whose literals start at or before k0
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (lo < hi) { if (T.sq_cl[mid] <= k0) lo = mid; else hi = mid - 1; }
                    }
                    uint32_t sq = (uint32_t)lo;
                    uint8_t *dstl = out + (uint64_t)o + k0;
                    if (n == 16 && k0 + 16 <= T.sq_cl[sq + 1 < nb ? sq + 1 : 64]) {
                        *reinterpret_cast<u4 *>(dstl + T.sq_sh[sq]) = v;
                    } else {                                                    // a boundary inside (or the stream's last bytes): byte by byte
                        uint32_t end = sq + 1 < nb ? T.sq_cl[sq + 1] : LB, sh = T.sq_sh[sq];
                        for (uint32_t j = 0; j < n; j++) {
                            while (k0 + j >= end && sq + 1 < nb) { sq++; end = sq + 1 < nb ? T.sq_cl[sq + 1] : LB; sh = T.sq_sh[sq]; }
                            dstl[j + sh] = (uint8_t)((j < 8 ? vlo : vhi) >> (8 * (j & 7)));
                        }
                    }
                }
                wave_sync();
                ZP(7)
                // ---- the batch's matches ----
                {
                    // positions relative to the batch's first byte (< 2^25: 64 sequences of at most 128 KiB + 128 KiB); a source may start in front of it
                    const int mop = (int)(cl_in + cm), mlen = (int)ml;
                    const int64_t ss64 = (int64_t)mop - (int64_t)of;
                    const int ss = (int)(ss64 < -1 ? -1 : ss64), se = min((int)(ss64 + mlen < 0 ? 0 : ss64 + mlen), mop);   // source bytes [ss, se) exist before this match starts writing
                    int lo = 0, hi = lane;
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const int mid = (lo + hi) >> 1;
                        const int v = __builtin_amdgcn_ds_bpermute(mid << 2, mop);
                        if (lo < hi) { if (v < se) lo = mid + 1; else hi = mid; }
                    }
                    const int cand = lo - 1;                                   // the last match in front of me that starts before se
                    int lo2 = 0, hi2 = lane;                                   // the first one that ends behind ss
                    const int e_op = mop + mlen;
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const int mid = (lo2 + hi2) >> 1;
                        const int v = __builtin_amdgcn_ds_bpermute(mid << 2, e_op);
                        if (lo2 < hi2) { if (v <= ss) lo2 = mid + 1; else hi2 = mid; }
                    }
                    const uint64_t need = cand >= lo2 ? (2ull << cand) - (1ull << lo2) : 0ull;
                    const bool big = mlen > 128;                               // copied by the whole wave
                    uint8_t *const ob = out + (uint64_t)o;
                    uint64_t donem = ~__ballot(have);
                    while (~donem) {
                        const bool ready = !((donem >> lane) & 1ull) && !(need & ~donem);
                        if (ready && !big) {
                            uint8_t *q = ob + mop;
                            if (of == 1) {                                     // a run: dwords wherever they start, the last one overlapping
                                typedef uint32_t u1 __attribute__((aligned(1)));
                                const uint32_t x4 = (uint32_t)q[-1] * 0x01010101u;
                                for (int k = 0; k + 4 < mlen; k += 4) *reinterpret_cast<u1 *>(q + k) = x4;
                                *reinterpret_cast<u1 *>(q + mlen - 4) = x4;    // (mlen >= 3: at worst this rewrites q[-1], which holds x)
                            } else {
                                // the bytes in front of q are periodic with period `of`: every step copies as much as is known, then twice as much is
                                int k = 0;
                                int64_t d = (int64_t)of;
                                while (k < mlen) {
                                    const int n = (int)min((int64_t)(mlen - k), d);
                                    copy_ends(q + k, q + k - d, n);
                                    k += n;
                                    d += d;
                                }
                            }
                        }
                        uint64_t bigm = __ballot(ready && big);
                        while (bigm) {
                            const int l = __ffsll((long long)bigm) - 1;
                            bigm &= bigm - 1;
                            const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane(mop, l), bl = (uint32_t)__builtin_amdgcn_readlane(mlen, l);
                            const uint32_t bf = (uint32_t)__builtin_amdgcn_readlane((int)of, l);
                            uint8_t *bq = ob + bo;
                            const uint8_t *src = bq - bf;
                            if (bf >= bl) { for (uint32_t k = lane; k < bl; k += 64) bq[k] = src[k]; }
                            else { for (uint32_t k = lane; k < bl; k += 64) bq[k] = src[k % bf]; }
                        }
                        // a match copied in this step may be the source of one in the next: the same release / acquire pair as the
                        // waiting matches of inflate_par_dev.h (the bytes travel through global memory between lanes of the wave)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        donem |= __ballot(ready);
                    }
                }
                o += LB + MB;
                li += LB;
                wave_sync();
                ZP(8)
            }
            if (status != INF_OK) break;
            if (__shfl((int)(br.done() && !br.overrun()), 0) == 0) { status = INF_ERR_DATA; break; }
        }
        const uint32_t restl = lsize - li;
        if ((uint64_t)o + restl > E) { status = fcs_bytes ? INF_ERR_DATA : INF_ERR_OVERFLOW; break; }
        if (lit) z_wave_move_down(out + o, lit + li, restl);
        else { for (uint32_t k = lane; k < restl; k += 64) out[o + k] = (uint8_t)lit_fill; }
        o += restl;
        wave_sync();
    }
    ZP(9)
    ZP_FLUSH
    if (status != INF_OK) return status;
    uint32_t want_sum = 0;
    if (checksum) { if (p + 4 > len) return INF_ERR_TRUNC; want_sum = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24); p += 4; }
    if (p != len) return INF_ERR_DATA;
    if (fcs_bytes && fcs != o) return INF_ERR_DATA;
    if (checksum) {                                               // libzstd rejects a frame whose content does not hash to its checksum: so do we
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if ((uint32_t)z_xxh64_wave(out, o) != want_sum) return INF_ERR_ADLER;
    }
    *olen_out = o;
    return INF_OK;
}

}   // namespace s5
