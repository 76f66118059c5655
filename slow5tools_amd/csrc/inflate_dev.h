// inflate_dev.h — zlib (RFC 1950/1951) stream decoder, one record per wave64.
//
// Replaces the record-depress stage of slow5lib reached through slow5_rec_depress_parse / slow5_get
// (/root/reference/src/view.c:38, src/get.c:45).  Accepts any conforming stream (stored / fixed /
// dynamic blocks, distances up to 32 KiB): the reference's fixtures were written by stock zlib.
//
// DEFLATE decoding is bit-serial per stream, so the parallelism is across records (thousands of waves in
// flight) and inside a wave only where the format allows it:
//   - the compressed stream is staged through a 2 KiB LDS window (all lanes refill it with aligned
//     dword loads + byte shift); lane 0 walks the bit stream out of LDS, never out of HBM
//   - 64 lanes build the canonical tables and the 10-bit / 8-bit lookup tables with ballots
//   - literals go to a 4 KiB LDS output ring, flushed to HBM by all lanes (coalesced) every 2 KiB;
//     the Adler-32 is accumulated on the flushed bytes
//   - matches are replicated by all lanes: out[o+k] = out[o-d + k mod d] (sources precede o)
#pragma once
#include "dev_common.h"

namespace s5 {

// tools/inflate_phases.py (variant build -DS5_IPROBE): clock ticks a record's wave spends in each phase of the parallel inflate, summed over the batch
#ifdef S5_IPROBE
__device__ unsigned long long g_iprobe[20];
struct IProbe { unsigned long long t; };
#define IPP_DECL IProbe ipp_; ipp_.t = __builtin_readcyclecounter(); IProbe *ipp = &ipp_;
#define IPP(i) { if (ipp) { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane_id() == 0) atomicAdd(&g_iprobe[i], n_ - ipp->t); ipp->t = __builtin_readcyclecounter(); } }
#define IPP_FLUSH { if (ipp && lane_id() == 0) atomicAdd(&g_iprobe[19], 1ull); }
#define IPP_ARG , IProbe *ipp = nullptr
#define IPP_PASS , ipp
#else
#define IPP_DECL
#define IPP(i)
#define IPP_FLUSH
#define IPP_ARG
#define IPP_PASS
#endif
// tools/par_probe.py (variant build -DS5_PAR_PROBE): cut-offs INSIDE the block header's parser, for instruction counts per step
#ifdef S5_PAR_PROBE
#define IPC(n) { if (cut == (n)) return INF_OK; }
#else
#define IPC(n)
#endif


constexpr int INF_LBITS = 10;      // primary lit/len lookup bits
constexpr int INF_DBITS = 8;       // primary distance lookup bits
constexpr int INF_IW = 2048;       // input window, bytes
constexpr int INF_OW = 4096;       // output ring, bytes
constexpr int INF_FLUSH = 2048;    // flush the ring when this many bytes are pending

struct InflShared {                  // per wave
    uint32_t win[INF_IW / 4];        // compressed bytes [wbase, wbase + INF_IW) of the deflate data
    uint8_t ring[INF_OW];            // output bytes, position o lives at ring[o % INF_OW]
    uint16_t llut[1 << INF_LBITS];   // sym | len << 9   (len == 0: long code -> canonical walk)
    uint16_t dlut[1 << INF_DBITS];   // sym | len << 5
    uint16_t lsym[288];              // canonical order symbols
    uint16_t dsym[32];
    uint16_t lcount[16], dcount[16];
    uint8_t lens[352];               // [0,19) code-length code | [32, 32+316) dynamic lit/len+dist; fixed: [0,288)+[288,320)
    __device__ __forceinline__ uint16_t *cl_lut() { return dlut; }   // the code-length code's 7-bit table: built before the distance table, in its storage
};

enum { INF_OK = 0, INF_ERR_HEADER = 1, INF_ERR_DATA = 2, INF_ERR_TRUNC = 3, INF_ERR_ADLER = 4, INF_ERR_OVERFLOW = 5 };

// Bit reader over the LDS window (lane 0).  Refills in aligned dwords.
struct BitIn {
    uint64_t buf;
    int cnt;           // valid bits in buf
    uint32_t wpos;     // next dword of the window
    uint32_t wbase;    // deflate-data byte offset of window dword 0
};
__device__ __forceinline__ void bi_need32(BitIn &b, const uint32_t *win) {   // afterwards cnt >= 33
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)win[b.wpos++] << b.cnt;
        b.cnt += 32;
    }
}
// wave-uniform variant: every lane runs it on the same state; the LDS word is forced scalar so the
// compiler keeps the bit reader in SGPRs and branches on SCC instead of juggling the exec mask
__device__ __forceinline__ void bi_need32_u(BitIn &b, const uint32_t *win) {
    if (b.cnt <= 32) {
        const uint32_t w = __builtin_amdgcn_readfirstlane(win[b.wpos]);
        b.wpos++;
        b.buf |= (uint64_t)w << b.cnt;
        b.cnt += 32;
    }
}
__device__ __forceinline__ uint32_t bi_get(BitIn &b, int n) {   // n <= 32 <= cnt
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
__device__ __forceinline__ uint64_t bi_consumed_bits(const BitIn &b) { return 8ull * b.wbase + 32ull * b.wpos - (uint64_t)b.cnt; }

// All 64 lanes: load the window so that window byte 0 = deflate byte `from` (any alignment).  Bytes at or
// past `total` read as zero.  src = first deflate byte (stream + 2).
__device__ __forceinline__ void infl_load_window(uint32_t *win, const uint8_t *src, uint32_t from, uint32_t total) {
    const int lane = lane_id();
    const uintptr_t addr = reinterpret_cast<uintptr_t>(src) + from;
    const uint32_t *g = reinterpret_cast<const uint32_t *>(addr & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(addr & 3) * 8;
    const uint32_t avail = from < total ? total - from : 0;   // bytes that exist from `from` on
#pragma unroll 2
    for (int q = 0; q < INF_IW / 4 / 64; q++) {
        const uint32_t i = q * 64 + lane;
        uint32_t w = 0;
        if (4 * i < avail) {
            const uint32_t lo = g[i];
            w = lo;
            if (sh) {
                // the high dword may lie past the record; the C ABI asks for 8 readable bytes after `in`
                const uint32_t hi = g[i + 1];
                w = (lo >> sh) | (hi << (32 - sh));
            }
            const uint32_t left = avail - 4 * i;
            if (left < 4) w &= (1u << (8 * left)) - 1;
        }
        win[i] = w;
    }
    wave_sync();
}

// Slide the window forward.  The bits still buffered in b stay inside the new window (it starts at the dword that holds the
// first unread bit), so a bit address relative to the window — 32 * wpos - cnt — is valid at all times.
__device__ __forceinline__ void infl_reload(uint32_t *win, const uint8_t *src, uint32_t total, BitIn &b) {
    const uint32_t back = ((uint32_t)b.cnt + 31u) >> 5;            // window dwords the buffered bits came from (<= 2)
    const uint32_t from = b.wbase + 4 * (b.wpos - back);
    infl_load_window(win, src, from, total);
    b.wbase = from;
    b.wpos = back;
}

// Build canonical tables + LUT for one alphabet from lens[0..n).  All 64 lanes of the wave.
// Returns nonzero if over-subscribed (incomplete codes are tolerated like zlib does for single-code
// distance trees; a code that does not exist simply never matches and reports INF_ERR_DATA).
__device__ __forceinline__ int infl_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *syms, uint16_t *lut,
                                          int lutbits, int lenshift) {
    const int lane = lane_id();
    if (lane < 16) count[lane] = 0;
    for (int i = lane; i < (1 << lutbits); i += 64) lut[i] = 0;
    wave_sync();
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
        if (l) syms[idx] = (uint16_t)s;
        // lookup-table entries: a code of l bits owns 2^(lutbits - l) of them.  A lane that fills its own symbol's entries keeps
        // the other 63 waiting for as long as the shortest code takes (the two 1-bit distance codes of a run-length stream: 128
        // stores each): codes with 16 entries or more are filled by the whole wave, one symbol at a time, the rest per lane
        const bool inl = l != 0 && l <= lutbits;
        const uint32_t rev = inl ? __brev(code) >> (32 - l) : 0u;
        const uint32_t ent = (uint32_t)s | ((uint32_t)l << lenshift);
        const bool wide = inl && l + 4 <= lutbits;
        uint64_t wm = __ballot(wide);
        while (wm) {
            const int from = __ffsll((long long)wm) - 1;
            wm &= wm - 1;
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)rev, from);
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ent, from);
            const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane(l, from);
            for (uint32_t k = r0 + ((uint32_t)lane << l0); k < (1u << lutbits); k += 64u << l0) lut[k] = (uint16_t)e0;
        }
        if (inl && !wide)
            for (uint32_t k = rev; k < (1u << lutbits); k += (1u << l)) lut[k] = (uint16_t)ent;
    }
    wave_sync();
    return 0;
}

// canonical bit-by-bit walk for codes longer than the LUT (needs cnt >= 15); returns symbol or -1
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

// Code lengths of a fixed (type 1) or dynamic (type 2) block and the decode tables of both alphabets, by the whole wave with the
// wave-uniform bit reader (the dynamic header is read out of the LDS window T.win, INF_IW bytes).  T: any struct with the table
// fields of InflShared.  Returns INF_OK or an error status; nl / nd = number of lit/len and distance codes.
// Canonical tables WITHOUT a lookup table (the parallel decoder): count[l] and the symbols in canonical order.  Counts by LDS
// atomics; a symbol's place = first place of its length + the symbols of that length in front of it, and the ballots that count
// those run over the lengths that OCCUR in a slice of 64 symbols (six or seven), not over all fifteen.
__device__ __forceinline__ int infl_build_syms(const uint8_t *lens, int n, uint16_t *count, uint16_t *syms, uint32_t *scratch16) {
    const int lane = lane_id();
    if (lane < 16) scratch16[lane] = 0;
    wave_sync();
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        if (l) atomicAdd(&scratch16[l], 1u);
    }
    wave_sync();
    // first place of every length, over-subscription check (lane l holds length l)
    const uint32_t c = lane >= 1 && lane < 16 ? scratch16[lane] : 0u;
    const uint32_t incl = wave_incl_add(c);
    uint32_t next = incl - c;                                  // my length's first place
    {
        // Kraft: sum c_l * 2^(15 - l) <= 2^15
        const uint32_t k = wave_sum(lane >= 1 && lane < 16 ? c << (15 - lane) : 0u);
        if (k > 32768u) return 1;
    }
    if (lane < 16) count[lane] = (uint16_t)c;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        uint64_t rem = __ballot(l != 0u);
        uint32_t place = 0;
        while (rem) {
            const uint32_t L0 = (uint32_t)__builtin_amdgcn_readlane((int)l, __ffsll((long long)rem) - 1);
            const uint64_t m = __ballot(l == L0);
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)L0);
            if (l == L0) place = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if ((uint32_t)lane == L0) next += (uint32_t)__popcll(m);
            rem &= ~m;
        }
        if (l) syms[place] = (uint16_t)s;
    }
    wave_sync();
    return 0;
}

// The code-length sequence of a dynamic header (hlit + hdist lengths, run-length coded in the code-length code) decoded by the
// whole wave instead of token by token on the uniform reader: lane L looks up the token that WOULD start L bits ahead of the
// reader (code-length codes are at most 7 bits: the 7-bit table always resolves them), three rounds of pointer doubling across
// the lanes find which of the 64 offsets the token chain really visits (8 tokens per scalar hop), a prefix sum of the tokens'
// run lengths places them, and a "repeat previous" token finds the length it repeats through a ballot.  ~9 rounds per header
// instead of ~150 dependent steps.  lens32 = T.lens + 32 must be zero (zero runs are not written).  Returns 0, or 1 bad data.
template <class TT>
__device__ __forceinline__ int infl_cl_sequence_wave(TT &T, BitIn &b, int tot, int idx = 0, uint32_t prev = 0xFFu) {   // prev: last length written (0xFF: none yet)
    const int lane = lane_id();
    while (idx < tot) {
        bi_need32_u(b, T.win);
        const uint32_t abit = 32u * b.wpos - (uint32_t)b.cnt + (uint32_t)lane;      // window bit address of my offset
        const uint32_t w0 = T.win[abit >> 5], w1 = T.win[(abit >> 5) + 1];
        const uint32_t bits = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (abit & 31));
        const uint32_t e = T.cl_lut()[bits & 127u];
        const uint32_t sym = e & 31u, clen = sym == 31u ? 0u : (e >> 5) & 7u;     // (the table of infl_build_cl7: the token's total bit count stands above; 31: no code)
        const uint32_t ext = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
        const uint32_t x = (bits >> clen) & ((1u << ext) - 1u);
        const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + x : 3u + x;
        const bool ok = clen != 0u;
        // E: offset reached after up to 8 tokens from mine (an invalid code absorbs), R: offsets visited on the way
        uint32_t E = ok ? (uint32_t)lane + clen + ext : (uint32_t)lane;
        uint32_t Rlo = ok && lane < 32 ? 1u << lane : 0u, Rhi = ok && lane >= 32 ? 1u << (lane - 32) : 0u;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int src = (int)(E & 63u) << 2;
            const uint32_t e2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)E);
            const uint32_t r2lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)Rlo);
            const uint32_t r2hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)Rhi);
            if (E < 64u) { Rlo |= r2lo; Rhi |= r2hi; E = e2; }
        }
        const uint64_t okm = __ballot(ok);
        uint32_t off = 0;
        uint64_t visited = 0;
        while ((okm >> off) & 1) {
            visited |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)Rlo, (int)off) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)Rhi, (int)off) << 32);
            off = (uint32_t)__builtin_amdgcn_readlane((int)E, (int)off);
            if (off >= 64) break;
        }
        if (!visited) return 1;                                   // the token at the reader is not a code of the code-length code
        const bool mine = (visited >> lane) & 1;
        const uint32_t incl = wave_incl_add(mine ? rep : 0u);
        // the sequence ends exactly at `tot` lengths: the first visited token that reaches it is the last one
        const uint64_t reach = __ballot(mine && (uint32_t)idx + incl >= (uint32_t)tot);
        int last_lane = 64;
        if (reach) {
            last_lane = __ffsll((long long)reach) - 1;
            if ((uint32_t)idx + (uint32_t)__builtin_amdgcn_readlane((int)incl, last_lane) > (uint32_t)tot) return 1;   // a run over the end
            visited &= last_lane == 63 ? ~0ull : (2ull << last_lane) - 1;
            off = (uint32_t)__builtin_amdgcn_readlane((int)(lane + clen + ext), last_lane);
        }
        const bool use = (visited >> lane) & 1;
        // the length a "repeat previous" token repeats: the nearest earlier token of this round that is not one itself, else the
        // length carried in from the round before
        const uint64_t defm = __ballot(use && sym != 16u);
        const uint64_t below = defm & ((1ull << lane) - 1);
        const int from = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t dv = sym < 16u ? sym : 0u;                  // 17 / 18: zero runs
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)dv);
        const uint32_t val = sym != 16u ? dv : (below ? got : prev);
        if (__ballot(use && sym == 16u && val == 0xFFu)) return 1;   // "repeat previous" with nothing in front of it
        if (use && sym <= 16u && val != 0u) {
            const uint32_t at = 32u + (uint32_t)idx + incl - rep;
            for (uint32_t k = 0; k < rep; k++) T.lens[at + k] = (uint8_t)val;
        }
        // carry: the value of the round's last token
        const int lastv = 63 - __clzll((long long)visited);
        prev = (uint32_t)__builtin_amdgcn_readlane((int)val, lastv);
        idx += (int)(uint32_t)__builtin_amdgcn_readlane((int)incl, lastv);
        while (off) {                                             // advance the uniform reader
            bi_need32_u(b, T.win);
            const uint32_t t = off < 32u ? off : 32u;
            bi_get(b, (int)t);
            off -= t;
        }
    }
    wave_sync();
    return 0;
}

// ---- round 6, third session: the dynamic header with a quarter of the instructions ----
// PMC cut-offs of the parallel inflate (tools/par_probe_pmc.sh, 262 144 own records): of 13.3 k vector instructions per record the block header
// took 4.1 k, the code-length sequence alone 1.8 k — infl_cl_sequence_wave looks at 64 bit offsets per round, a fifth of which start a token,
// and pays ~165 instructions per round for 11 rounds.  Below: the generic table builder replaced by a 19-symbol one that resolves every
// table entry once (infl_build_cl7), and the sequence taken 1024 bits per round (infl_cl_sequence_wave2).

// 7-bit lookup table of the code-length code (19 symbols, lengths 0..7): entry = symbol | code length << 5 | (code length + extra bits) << 8;
// seven bits that start no code: "symbol 31", a token of 14 bits — the chain of infl_cl_sequence_wave2 walks over it like over any other
// token (behind the sequence lies the block's data, garbage to this code), and only a lane that meets it INSIDE the sequence reports it.  Built by DECODING, like the lit/len table of inflate_par_dev.h: entry i is the code that starts the seven bits i, found by
// comparing them (first bit on top) with the left-justified end of every length's codes — uniform values.  sorted: >= 19 entries of scratch,
// adj: >= 8.  Returns 1 if the lengths are over-subscribed (an incomplete set is tolerated: its unused entries are CL7_NOCODE).
constexpr uint32_t CL7_NOCODE = 31u | (7u << 5) | (14u << 8);
__device__ __forceinline__ int infl_build_cl7(const uint8_t *lens19, uint16_t *lut, uint16_t *sorted, uint16_t *adj) {
    const int lane = lane_id();
    const uint32_t l = lane < 19 ? (uint32_t)lens19[lane] : 0u;
    uint32_t first = 0, offs = 0, place = 0, adjv = 0;
    uint32_t lim[8];
    int left = 1;
    bool over = false;
#pragma unroll
    for (int bb = 1; bb <= 7; bb++) {
        const uint64_t m = __ballot(l == (uint32_t)bb);
        const uint32_t c = (uint32_t)__popcll(m);
        if (l == (uint32_t)bb) place = offs + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == bb) adjv = offs - first;                      // index of a length's first symbol in `sorted` - its first code
        lim[bb] = (first + c) << (7 - bb);
        offs += c;
        first = (first + c) << 1;
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
    }
    if (over) return 1;
    if (l) sorted[place] = (uint16_t)lane;
    if (lane >= 1 && lane < 8) adj[lane] = (uint16_t)adjv;
    wave_sync();
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const uint32_t i = (uint32_t)(it * 64 + lane);
        const uint32_t v = __brev(i) >> 25;
        uint32_t len = 1;
#pragma unroll
        for (int bb = 1; bb <= 7; bb++) len += v >= lim[bb] ? 1u : 0u;      // (the ends never decrease: a length without codes ends where the one below it does)
        uint32_t ent = CL7_NOCODE;
        if (len <= 7u) {
            const uint32_t idx = ((uint32_t)(int)(short)adj[len] + (v >> (7u - len))) & 31u;
            const uint32_t sym = sorted[idx];
            const uint32_t ext = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
            ent = sym | (len << 5) | ((len + ext) << 8);
        }
        lut[i] = (uint16_t)ent;
    }
    wave_sync();
    return 0;
}

// Tables of an alphabet of at most 32 symbols and lengths up to 15 — the distance code — in the manner of infl_build_cl7 (one lane per symbol,
// every lookup entry decoded once) instead of the generic builder's two loops over fifteen lengths and its symbol-by-symbol table fill:
// count[1..15] and the symbols in canonical order (what the canonical walk behind a lookup miss reads), lut[1 << LB] = symbol | length << 5 for
// codes of up to LB bits, 0 for everything longer or unused.  adj: 16 entries of scratch.  Returns 1 if over-subscribed.
template <int LB>
__device__ __forceinline__ int infl_build_small(const uint8_t *lens, int n, uint16_t *count, uint16_t *syms, uint16_t *lut, uint16_t *adj) {
    const int lane = lane_id();
    const uint32_t l = lane < n && lane < 32 ? (uint32_t)lens[lane] : 0u;
    uint32_t first = 0, offs = 0, place = 0, adjv = 0, cntv = 0;
    uint32_t lim[LB + 1];
    int left = 1;
    bool over = false;
#pragma unroll
    for (int bb = 1; bb <= 15; bb++) {
        const uint64_t m = __ballot(l == (uint32_t)bb);
        const uint32_t c = (uint32_t)__popcll(m);
        if (m) {                                                   // (uniform: a run-length stream has two lengths' worth of work here, not fifteen)
            if (l == (uint32_t)bb) place = offs + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == bb) cntv = c;
        }
        if (lane == bb) adjv = offs - first;
        if (bb <= LB) lim[bb] = (first + c) << (LB - bb);
        offs += c;
        first = (first + c) << 1;
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
    }
    if (over) return 1;
    if (l) syms[place] = (uint16_t)lane;
    if (lane < 16) { count[lane] = (uint16_t)cntv; adj[lane] = (uint16_t)adjv; }
    wave_sync();
#pragma unroll
    for (int it = 0; it < (1 << LB) / 64; it++) {
        const uint32_t i = (uint32_t)(it * 64 + lane);
        const uint32_t v = __brev(i) >> (32 - LB);
        uint32_t len = 1;
#pragma unroll
        for (int bb = 1; bb <= LB; bb++) len += v >= lim[bb] ? 1u : 0u;
        uint32_t ent = 0;
        if (len <= (uint32_t)LB) {
            const uint32_t idx = ((uint32_t)(int)(short)adj[len] + (v >> ((uint32_t)LB - len))) & 31u;
            ent = (uint32_t)syms[idx] | (len << 5);
        }
        lut[i] = (uint16_t)ent;
    }
    wave_sync();
    return 0;
}

// wave-uniform: put the reader at window bit `abit`
__device__ __forceinline__ void bi_seek_u(BitIn &b, const uint32_t *win, uint32_t abit) {
    const uint32_t w = abit >> 5, sh = abit & 31u;
    const uint32_t word = __builtin_amdgcn_readfirstlane(win[w]);
    b.buf = (uint64_t)(word >> sh);
    b.cnt = (int)(32u - sh);
    b.wpos = w + 1u;
}

// The code-length sequence, 1024 bits per round: lane L owns the sixteen bit offsets [16 L, 16 L + 16) behind the reader.
//   1. every lane looks up the token that WOULD start at each of its offsets and keeps its total bit count (code + extra bits, <= 14) as a
//      nibble: 64 bits per lane;
//   2. where the real token chain ENTERS each lane is a serial walk over the lanes (a token never skips a lane: 15 + 14 < 32).  Every lane first
//      folds its sixteen offsets into a FUNCTION entry offset -> exit offset (sixteen nibbles, from the last offset down); the walk is then one
//      readlane and one scalar shift per lane.  Seven bits that start no code count as a 14-bit token (CL7_NOCODE): the chain never breaks, and
//      whether such a place was an error is decided by where the sequence ended;
//   3. every lane walks its own tokens from its entry — at most eight static slots (nine or more tokens in sixteen bits: the 64-offset
//      parser above takes the header instead; only before anything was written) — and sums their run lengths; a prefix sum places them,
//      the first token whose cumulative count reaches `tot` is the last one (beyond it: error), "repeat previous" takes the value of the
//      nearest token in front that is not one itself (inside the lane in order, across lanes by a ballot, across rounds in `prev`);
//   4. lengths are written (zero runs are not: T.lens + 32 was cleared), the reader is put behind the last token.
// Our own records' sequences are ~800 bits: one round instead of eleven.  Returns 0, or 1 bad data.
template <class TT>
__device__ __forceinline__ int infl_cl_sequence_wave2(TT &T, BitIn &b, int tot) {
    const int lane = lane_id();
    const uint16_t *lut = T.cl_lut();
    uint32_t A = 32u * b.wpos - (uint32_t)b.cnt;                    // window bit address of the round's first bit (uniform)
    int idx = 0;
    uint32_t prev = 0xFFu;
    for (int round = 0; round < 4; round++) {                       // (316 lengths of at most 14 bits: three rounds)
        const uint32_t base = A + 16u * (uint32_t)lane;
        const uint32_t w = base >> 5, sh = base & 31u;
        const uint32_t d0 = T.win[w], d1 = T.win[w + 1], d2 = T.win[w + 2];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
        uint32_t tlo = 0, thi = 0;
#pragma unroll
        for (int o = 0; o < 16; o++) {
            const uint32_t bits = o ? __builtin_amdgcn_alignbit(hi, lo, (uint32_t)o) : lo;
            const uint32_t t = (uint32_t)lut[bits & 127u] >> 8;
            if (o < 8) tlo |= t << (4 * o);
            else thi |= t << (4 * (o - 8));
        }
        // where the chain leaves my sixteen offsets, for every offset it could enter at: x[o] = o + t - 16 if the token at o reaches out of
        // them, else x[o + t] — from the last offset down, nibbles again (exits are 0 .. 13)
        uint32_t xlo = 0, xhi = 0;
#pragma unroll
        for (int o = 15; o >= 0; o--) {
            const uint32_t t = o < 8 ? __builtin_amdgcn_ubfe(tlo, 4 * o, 4) : __builtin_amdgcn_ubfe(thi, 4 * (o - 8), 4);
            const uint32_t tgt = (uint32_t)o + t;                                         // > o
            uint32_t x = __builtin_amdgcn_ubfe(xhi, (tgt << 2) & 31u, 4);                 // tgt in 8 .. 15
            if (o < 7) { const uint32_t xl = __builtin_amdgcn_ubfe(xlo, (tgt << 2) & 31u, 4); x = tgt < 8u ? xl : x; }
            x = tgt >= 16u ? tgt - 16u : x;
            if (o < 8) xlo |= x << (4 * o);
            else xhi |= x << (4 * (o - 8));
        }
        // (uniform, scalar unit) e: entry into lane L — one hop per lane: a readlane and a shift; the entries collect as nibbles, sixteen lanes to
        // a 64-bit scalar (this compiler has no writelane builtin), and every lane picks its own.  (First form: the chain followed token by token
        // on the scalar unit from the nibbles of step 1 — 4.6 k scalar instructions per record, serial: the kernel went from 6.9 to 9.4 ms per
        // 262 k records.  The CU has ONE scalar unit: it issues as many instructions per cycle as the four vector units together.)
        uint32_t e = 0;
        uint64_t e64[4];
#pragma unroll
        for (int blk = 0; blk < 4; blk++) {
            uint64_t acc = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int L = blk * 16 + j;
                acc |= (uint64_t)e << (4 * j);
                // (both halves fetched, one 64-bit shift: choosing the half first compiled into two branches per hop — 13 scalar instructions)
                const uint64_t xx = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)xlo, L) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)xhi, L) << 32);
                e = (uint32_t)(xx >> (4u * e)) & 15u;
            }
            e64[blk] = acc;
        }
        uint32_t ent;                                                 // my entry offset
        {
            const int q = lane >> 4;
            const uint64_t mine = q == 0 ? e64[0] : q == 1 ? e64[1] : q == 2 ? e64[2] : e64[3];
            ent = (uint32_t)(mine >> (4 * (lane & 15))) & 15u;
        }
        // my tokens: run length | symbol << 8 | offset behind the token << 16
        uint32_t tk[8];
        uint32_t o = ent, sum = 0, lastdef = 0;
        bool bad = false, has_def = false;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, o);
            const uint32_t en = lut[bits & 127u];
            const uint32_t sym = en & 31u, clen = (en >> 5) & 7u, t = en >> 8;
            const uint32_t x = (bits >> clen) & ((1u << (t - clen)) - 1u);
            const uint32_t rp = sym < 16u ? 1u : (sym == 18u ? 11u : 3u) + x;
            const bool act = o < 16u;
            const bool valid = act && sym != 31u;
            if (act && sym == 31u) bad = true;
            o = valid ? o + t : 16u;
            tk[k] = valid ? rp | (sym << 8) | (o << 16) : 0u;
            if (valid) sum += rp;
            if (valid && sym != 16u) { lastdef = sym < 16u ? sym : 0u; has_def = true; }
            if (k >= 3 && k < 7 && !__ballot(o < 16u)) {              // (uniform: most rounds need four or five slots)
#pragma unroll
                for (int q = k + 1; q < 8; q++) tk[q] = 0u;
                break;
            }
        }
        if (__ballot(o < 16u)) {
            // nine or more tokens in some lane's sixteen bits (code lengths of one and two bits): the 64-offset parser from this round's start
            bi_seek_u(b, T.win, A);
            return infl_cl_sequence_wave(T, b, tot, idx, prev);
        }
        const uint32_t incl = wave_incl_add(sum);
        const uint64_t reach = __ballot((uint32_t)idx + incl >= (uint32_t)tot);
        const int endlane = reach ? __ffsll((long long)reach) - 1 : 64;
        const uint64_t badm = __ballot(bad);
        if (badm && __ffsll((long long)badm) - 1 < endlane) return 1;             // no code at a place the sequence has to pass
        // the value a "repeat previous" at the head of my tokens repeats
        const uint64_t defm = __ballot(has_def);
        const uint64_t below = defm & ((1ull << lane) - 1);
        const int from = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)lastdef);
        uint32_t cur = below ? got : prev;
        uint32_t cum = (uint32_t)idx + incl - sum;                      // lengths in front of my next token
        uint32_t endpos = 0;
        bool err = false, isend = false;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k >= 4 && !__ballot(tk[k] != 0u)) break;              // (uniform; slots fill from the front)
            const uint32_t rp = tk[k] & 255u, sym = (tk[k] >> 8) & 31u;
            const bool used = tk[k] != 0u && cum < (uint32_t)tot;
            const uint32_t v = sym < 16u ? sym : sym == 16u ? cur : 0u;
            if (used) {
                if ((sym == 16u && cur == 0xFFu) || cum + rp > (uint32_t)tot) err = true;      // nothing to repeat; a run over the end
                else if (sym <= 16u && v != 0u)
                    for (uint32_t r = 0; r < rp; r++) T.lens[32u + cum + r] = (uint8_t)v;
                cur = v;
                cum += rp;
                if (cum == (uint32_t)tot) { isend = true; endpos = tk[k] >> 16; }
            }
        }
        if (__ballot(err)) return 1;
        const uint64_t endm = __ballot(isend);
        if (endm) {
            const int el = __ffsll((long long)endm) - 1;
            bi_seek_u(b, T.win, (uint32_t)__builtin_amdgcn_readlane((int)base, el) + (uint32_t)__builtin_amdgcn_readlane((int)endpos, el));
            wave_sync();
            return 0;
        }
        if (badm) return 1;                                             // (cannot happen: a bad token in front of the end was caught above)
        idx += (int)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (defm) prev = (uint32_t)__builtin_amdgcn_readlane((int)lastdef, 63 - __clzll((long long)defm));
        A += 1024u + e;
    }
    return 1;
}

// LITLUT = 0: no lit/len lookup table (the parallel decoder resolves those codes by comparison); PARCL: the code-length sequence by
// the whole wave (infl_cl_sequence_wave)
template <class TT, int LITLUT = INF_LBITS, bool PARCL = false, int DBITS = INF_DBITS>
__device__ __forceinline__ int infl_block_tables(TT &T, const uint8_t *src, uint32_t total, uint64_t total_bits, BitIn &b, int type, int &nl, int &nd IPP_ARG, uint32_t cut = 0) {
    const int lane = lane_id();
    // ---- code lengths ----
    if (type == 1) {
        for (int s = lane; s < 288; s += 64) T.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        if (lane < 32) T.lens[288 + lane] = 5;
        nl = 288;
        nd = 30;
        wave_sync();
    } else {
        // the whole dynamic header is at most 14 + 57 + 316 * 14 bits = 562 bytes: make sure it is in the window
        if (b.wpos > INF_IW / 4 - 160) {
            infl_reload(T.win, src, total, b);
        }
        bi_need32_u(b, T.win);
        const uint32_t hd = bi_get(b, 14);
        nl = (int)(hd & 31) + 257;
        nd = (int)((hd >> 5) & 31) + 1;
        const int ncl = (int)(hd >> 10) + 4;
        if (nl > 286 || nd > 30) return INF_ERR_DATA;
        if (lane < 19) T.lens[lane] = 0;
        wave_sync();
        {   // the 3-bit lengths of the code-length code, one per lane, straight out of the window; then the reader steps over them
            const uint32_t abit = 32u * b.wpos - (uint32_t)b.cnt + 3u * (uint32_t)lane;
            if (lane < ncl) {
                const uint32_t w0 = T.win[abit >> 5], w1 = T.win[(abit >> 5) + 1];
                // order[]: 16 17 18 0 | 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (even lanes count up from 8, odd ones down from 7)
                const int at = lane < 3 ? 16 + lane : lane == 3 ? 0 : (lane & 1) ? 7 - ((lane - 5) >> 1) : 8 + ((lane - 4) >> 1);
                T.lens[at] = (uint8_t)(__builtin_amdgcn_alignbit(w1, w0, abit & 31u) & 7u);
            }
            uint32_t off = 3u * (uint32_t)ncl;
            while (off) {
                bi_need32_u(b, T.win);
                const uint32_t t = off < 32u ? off : 32u;
                bi_get(b, (int)t);
                off -= t;
            }
        }
        wave_sync();
        IPP(1)
        IPC(1)
        // the code-length code reuses the distance tables' storage (built before the real ones)
        if (PARCL) { if (infl_build_cl7(T.lens, T.cl_lut(), T.dsym, T.dcount)) return INF_ERR_DATA; }
        else if (infl_build(T.lens, 19, T.dcount, T.dsym, T.cl_lut(), 7, 5)) return INF_ERR_DATA;
        IPP(2)
        IPC(2)
        int bad = 0;
        if (PARCL) {
            for (int i = lane; i < 320; i += 64) T.lens[32 + i] = 0;
            wave_sync();
#ifndef S5_CLSEQ_V1
            bad = infl_cl_sequence_wave2(T, b, nl + nd);
#else
            bad = infl_cl_sequence_wave(T, b, nl + nd);
#endif
            if (!bad && bi_consumed_bits(b) > total_bits) bad = 2;
        } else {
            uint8_t prev = 0;
            int idx = 0;
            const int tot = nl + nd;
            while (idx < tot) {
                bi_need32_u(b, T.win);
                const uint32_t e = __builtin_amdgcn_readfirstlane((uint32_t)T.cl_lut()[(uint32_t)b.buf & 127]);
                if (!(e >> 5)) { bad = 1; break; }
                const int sym = e & 31;
                bi_get(b, e >> 5);
                if (sym < 16) { prev = (uint8_t)sym; if (lane == 0) T.lens[32 + idx] = prev; idx++; }
                else {
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) { if (idx == 0) { bad = 1; break; } v = prev; rep = 3 + (int)bi_get(b, 2); }
                    else if (sym == 17) rep = 3 + (int)bi_get(b, 3);
                    else rep = 11 + (int)bi_get(b, 7);
                    if (idx + rep > tot) { bad = 1; break; }
                    if (lane < rep) T.lens[32 + idx + lane] = v;   // rep <= 138: two rounds at most
                    if (lane + 64 < rep) T.lens[32 + idx + lane + 64] = v;
                    if (lane + 128 < rep) T.lens[32 + idx + lane + 128] = v;
                    idx += rep;
                    if (sym != 16) prev = 0;
                }
            }
            if (!bad && bi_consumed_bits(b) > total_bits) bad = 2;
        }
        if (bad) return bad == 2 ? INF_ERR_TRUNC : INF_ERR_DATA;
        wave_sync();
        IPP(3)
        IPC(3)
    }
    // tables: dynamic lengths sit at T.lens[32 ..] (lit/len then dist); fixed at [0..288) + [288..)
    const uint8_t *ll = type == 1 ? T.lens : T.lens + 32;
    const uint8_t *dl = type == 1 ? T.lens + 288 : T.lens + 32 + nl;
    if (type == 2 && ll[256] == 0) return INF_ERR_DATA;
    if constexpr (LITLUT == 0) { if (infl_build_syms(ll, nl, T.lcount, T.lsym, reinterpret_cast<uint32_t *>(T.llut))) return INF_ERR_DATA; }   // (T.llut: >= 64 bytes of scratch)
    else { if (infl_build(ll, nl, T.lcount, T.lsym, T.llut, LITLUT, 9)) return INF_ERR_DATA; }
    IPP(4)
    IPC(4)
    if constexpr (PARCL) { if (infl_build_small<DBITS>(dl, nd, T.dcount, T.dsym, T.dlut, T.ladj)) return INF_ERR_DATA; }     // (T.ladj: scratch until the lit/len limits are set up)
    else { if (infl_build(dl, nd, T.dcount, T.dsym, T.dlut, DBITS, 5)) return INF_ERR_DATA; }
    IPP(5)
    IPC(5)
    return INF_OK;
}

// All lanes: write ring bytes [from, to) to HBM (nothing at or beyond cap) and fold them into the Adler-32.
__device__ __forceinline__ void infl_flush(const uint8_t *ring, uint8_t *out, uint32_t from, uint32_t to, uint32_t cap,
                                           uint32_t &adA, uint32_t &adB) {
    const int lane = lane_id();
    const uint32_t n = to - from;
    uint32_t sa = 0;
    uint64_t sb = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t x = ring[(from + i) & (INF_OW - 1)];
        if (from + i < cap) out[from + i] = (uint8_t)x;
        sa += x;
        sb += (uint64_t)(n - i) * x;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sa += __shfl_xor(sa, d);
        sb += __shfl_xor(sb, d);
    }
    adB = (uint32_t)(((uint64_t)adB + (uint64_t)n * adA + sb) % 65521u);
    adA = (adA + sa) % 65521u;
}

// Inflate one zlib stream.  out: HBM, cap bytes.  Returns status; *out_len = decoded length (also when
// INF_ERR_OVERFLOW: the size needed; nothing at or beyond cap is written).  Adler-32 verified.
// `in` must have 8 readable bytes after in + in_len.
// HEAD: only the first `cap` bytes are wanted (a record's head: read_id_len | read_id | ..., for the index builder): decoding stops as soon
// as they are there — no Adler-32 (the stream is not read to its end), *out_len = bytes written (<= cap); `in` may be just the front
// of the stream: running out of input before cap bytes are there is INF_ERR_TRUNC.
template <bool HEAD = false>
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
    const uint8_t *src = in + 2;
    const uint32_t total = in_len - 6;            // deflate data bytes (the Adler-32 trailer is not part of them)
    const uint64_t total_bits = 8ull * total;
    BitIn b;
    b.buf = 0; b.cnt = 0; b.wpos = 0; b.wbase = 0;
    infl_load_window(T.win, src, 0, total);
    uint32_t o = 0;              // bytes produced (uniform)
    uint32_t flushed = 0;        // bytes already written to HBM (uniform)
    uint32_t ring_lo = 0;        // lowest position whose byte is certainly still in the ring
    uint32_t adA = 1, adB = 0;
    int status = INF_OK;
    int last = 0;
    // The bit reader state `b` is WAVE-UNIFORM: every lane executes the same reads on the same LDS words, so
    // no broadcast is ever needed and the compiler can keep the reader on the scalar unit.
    while (!last && status == INF_OK) {
        // ---- block header ----
        if (b.wpos > INF_IW / 4 - 3) {   // window nearly used up: reload first
            infl_reload(T.win, src, total, b);
            continue;
        }
        bi_need32_u(b, T.win);
        const uint32_t hdr = bi_get(b, 3);
        if (bi_consumed_bits(b) > total_bits) { status = INF_ERR_TRUNC; break; }
        last = hdr & 1;
        const int type = hdr >> 1;
        if (type == 3) { status = INF_ERR_DATA; break; }
        if (type == 0) {
            // stored: align to a byte, LEN/NLEN, then raw bytes straight HBM -> HBM
            bi_get(b, b.cnt & 7);
            bi_need32_u(b, T.win);
            const uint32_t len = bi_get(b, 16);
            const uint32_t nlen = bi_get(b, 16);
            const uint32_t pos = (uint32_t)(bi_consumed_bits(b) >> 3);   // byte offset of the raw data
            if ((len ^ 0xFFFFu) != nlen) { status = INF_ERR_DATA; break; }
            if (8ull * pos > total_bits || (uint64_t)pos + len > total) { status = INF_ERR_TRUNC; break; }
            wave_sync();
            infl_flush(T.ring, out, flushed, o, cap, adA, adB);
            {
                uint32_t sa = 0;
                uint64_t sb = 0;
                for (uint32_t i = lane; i < len; i += 64) {
                    const uint32_t x = src[pos + i];
                    if (o + i < cap) out[o + i] = (uint8_t)x;
                    sa += x;
                    sb += (uint64_t)(len - i) * x;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
                adB = (uint32_t)(((uint64_t)adB + (uint64_t)len * adA + sb) % 65521u);
                adA = (adA + sa) % 65521u;
            }
            o += len;
            flushed = o;
            ring_lo = o;   // the ring holds none of these bytes: later matches read them back from HBM
            if (HEAD && o >= cap) { last = 1; continue; }
            infl_load_window(T.win, src, pos + len, total);
            b.wbase = pos + len; b.wpos = 0; b.buf = 0; b.cnt = 0;
            continue;
        }
        // ---- code lengths and tables ----
        int nl, nd;
        { const int rc = infl_block_tables(T, src, total, total_bits, b, type, nl, nd); if (rc != INF_OK) { status = rc; break; } }
        // Codes longer than the LUT: every length tested at once.  Lane L (1..15) holds limit = first[L] + count[L] and
        // base = offs[L] - first[L] of the canonical lit/len code; the code of length L is the top L bits of the bit-reversed
        // peek, the true length is the smallest L with code < limit (one ballot), the symbol is lsym[code + base].
        uint32_t ll_lim = 0;
        int ll_base = 0;
        {
            uint32_t first = 0, offs = 0;
            for (int l = 1; l <= 15; l++) {
                const uint32_t c = T.lcount[l];
                if (lane == l) { ll_lim = first + c; ll_base = (int)offs - (int)first; }
                offs += c;
                first = (first + c) << 1;
            }
        }
        auto long_code = [&](BitIn &bb) -> int {   // needs >= 15 buffered bits; returns the symbol or -1, consumes the code
            const uint32_t rev = __brev((uint32_t)bb.buf & 0x7FFFu);
            const uint32_t c = (lane >= 1 && lane <= 15) ? rev >> (32 - lane) : 0xFFFFFFFFu;
            const uint64_t hit = __ballot(lane >= 1 && lane <= 15 && c < ll_lim);
            if (!hit) return -1;
            const int L = __ffsll((long long)hit) - 1;
            const int idx = __builtin_amdgcn_readlane((int)c + ll_base, L);
            bi_get(bb, L);
            return (int)__builtin_amdgcn_readfirstlane((uint32_t)T.lsym[idx]);
        };
        // ---- symbols: lane 0 decodes out of LDS; the wave refills / flushes / replicates matches ----
        for (;;) {
            // st: 0 match, 1 end of block, 2 data error, 3 truncated, 4 input window low, 5 ring needs a flush
            uint32_t mlen = 0, mdist = 0, st = 0;
            {   // wave-uniform: all lanes walk the same bits (scalar unit); only lane 0 stores the literal
                for (;;) {
                    if (HEAD && o >= cap) { st = 6; break; }
                    if (b.wpos > INF_IW / 4 - 3) { st = 4; break; }
                    if (o - flushed >= INF_FLUSH) { st = 5; break; }
                    bi_need32_u(b, T.win);
                    // A run of literals in one round: lane L looks up the code that WOULD start L bits ahead of the reader (its
                    // bits come straight from the LDS window, so all 64 offsets are live; one LUT read for the whole wave); the
                    // scalar unit then hops from code to code through those results (v_readlane with a scalar index: a few
                    // cycles per symbol instead of an LDS round trip each), and the lanes at the visited offsets store their
                    // literals side by side.  Stops at the first symbol that is not a LUT-resolved literal (match, end of
                    // block, long code); that symbol takes the one-at-a-time route below.
                    {
                        const uint32_t abit = 32u * b.wpos - (uint32_t)b.cnt + (uint32_t)lane;   // window bit address of my offset
                        const uint32_t w0 = T.win[abit >> 5], w1 = T.win[(abit >> 5) + 1];
                        const uint32_t ev = T.llut[(uint32_t)((((uint64_t)w1 << 32) | w0) >> (abit & 31)) & ((1 << INF_LBITS) - 1)];
                        // per lane: is the code at my offset a LUT-resolved literal, and where would the next code start
                        const bool is_lit = (ev >> 9) != 0 && (ev & 511u) < 256u;
                        const uint64_t lit_at = __ballot(is_lit);
                        // Where does a walk that starts at my offset stand after up to 4 literal codes, and which offsets did it
                        // visit?  Two rounds of pointer doubling across the lanes (ds_bpermute): E = offset reached (a code that
                        // is not a literal absorbs: E = its own offset; >= 64 = beyond this round), R = literal offsets visited.
                        // The scalar unit then hops four codes at a time — it is the one unit the 16 waves of a CU share, and the
                        // code-by-code hop (a v_readlane with a scalar index per code) was what bounded this kernel.
                        uint32_t E = is_lit ? (uint32_t)lane + (ev >> 9) : (uint32_t)lane;
                        uint32_t Rlo = is_lit && lane < 32 ? 1u << lane : 0u, Rhi = is_lit && lane >= 32 ? 1u << (lane - 32) : 0u;
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int src = (int)(E & 63u) << 2;
                            const uint32_t e2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)E);
                            const uint32_t r2lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)Rlo);
                            const uint32_t r2hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)Rhi);
                            if (E < 64u) { Rlo |= r2lo; Rhi |= r2hi; E = e2; }
                        }
                        uint32_t off = 0;
                        uint64_t visited = 0;
                        while ((lit_at >> off) & 1) {
                            visited |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)Rlo, (int)off) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)Rhi, (int)off) << 32);
                            off = (uint32_t)__builtin_amdgcn_readlane((int)E, (int)off);
                            if (off >= 64) break;
                        }
                        if (visited) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(visited >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)visited, 0u));
                            if ((visited >> lane) & 1) T.ring[(o + rank) & (INF_OW - 1)] = (uint8_t)ev;
                            o += (uint32_t)__popcll(visited);
                            while (off) {                       // advance the scalar reader by `off` (<= 78) bits
                                bi_need32_u(b, T.win);
                                const uint32_t t = off < 32u ? off : 32u;
                                bi_get(b, (int)t);
                                off -= t;
                            }
                            continue;
                        }
                    }
                    int sym;
                    const uint32_t e = __builtin_amdgcn_readfirstlane((uint32_t)T.llut[(uint32_t)b.buf & ((1 << INF_LBITS) - 1)]);
                    if (e >> 9) { sym = e & 511; bi_get(b, e >> 9); }
                    else sym = long_code(b);
                    if (sym < 0) { st = 2; break; }
                    if (sym < 256) { if (lane == 0) T.ring[o & (INF_OW - 1)] = (uint8_t)sym; o++; continue; }
                    if (sym == 256) { st = 1; break; }
                    sym -= 257;
                    if (sym >= 29) { st = 2; break; }
                    mlen = lbase[sym] + bi_get(b, lext[sym]);
                    bi_need32_u(b, T.win);
                    int ds;
                    const uint32_t de = __builtin_amdgcn_readfirstlane((uint32_t)T.dlut[(uint32_t)b.buf & ((1 << INF_DBITS) - 1)]);
                    if (de >> 5) { ds = de & 31; bi_get(b, de >> 5); }
                    else ds = __builtin_amdgcn_readfirstlane(infl_slow(b, T.dcount, T.dsym));
                    if (ds < 0 || ds >= 30) { st = 2; break; }
                    mdist = dbase[ds] + bi_get(b, dext[ds]);
                    if (mdist > o) { st = 2; break; }
                    break;
                }
                if (st != 2 && bi_consumed_bits(b) > total_bits) st = 3;   // decoded out of the zero padding past the end
            }
            if (st == 2 || st == 3) { status = st == 2 ? INF_ERR_DATA : INF_ERR_TRUNC; break; }
            if (st == 1) break;
            if (st == 6) { last = 1; break; }              // HEAD: enough bytes
            if (st == 4) {
                infl_reload(T.win, src, total, b);
                continue;
            }
            if (st == 5) {
                wave_sync();
                infl_flush(T.ring, out, flushed, o, cap, adA, adB);
                flushed = o;
                continue;
            }
            wave_sync();
            // sources older than what the ring still holds (or bytes of a stored block) come back from HBM
            const uint32_t in_ring = max(ring_lo, o + mlen > (uint32_t)INF_OW ? o + mlen - INF_OW : 0u);
            for (uint32_t k = lane; k < mlen; k += 64) {
                const uint32_t sp = o - mdist + (k % mdist);
                const uint8_t x = sp >= in_ring ? T.ring[sp & (INF_OW - 1)] : (sp < cap ? out[sp] : (uint8_t)0);
                T.ring[(o + k) & (INF_OW - 1)] = x;
            }
            wave_sync();
            o += mlen;
        }
    }
    wave_sync();
    if (HEAD) {
        if (status == INF_OK) infl_flush(T.ring, out, flushed, o, cap, adA, adB);
        *out_len = o < cap ? o : cap;
        return status;
    }
    if (status == INF_OK) {
        infl_flush(T.ring, out, flushed, o, cap, adA, adB);
        const uint8_t *t = in + in_len - 4;
        const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        if (o <= cap && ((adB << 16) | adA) != want) status = INF_ERR_ADLER;
    }
    *out_len = o;
    if (status == INF_OK && o > cap) status = INF_ERR_OVERFLOW;
    return status;
}

}  // namespace s5
