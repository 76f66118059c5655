// inflate_par_dev.h — zlib (RFC 1950/1951) decoder, one record per wave64, PARALLEL INSIDE THE RECORD.
//
// The two older decoders leave a record's symbol stream serial: inflate_dev.h walks it on the scalar unit (one record per
// wave: ~26 dependent instructions per symbol, 0.83 ms for a `get` batch of 4096 records however few of them there are), and
// inflate_simt_dev.h gives every lane a record of its own (64 private table sets: six waves per CU, latency bound at 20 M
// records/s).  A Huffman stream does not have to be decoded from its first bit, though: a decoder dropped at an arbitrary bit
// falls into step with the true code boundaries after a few codes (self-synchronisation), and that is all the parallelism
// this decoder needs:
//   window   up to 4 KiB of the compressed stream in LDS, cut into 64 segments of equal bit length (>= 384 bits: a lane needs a
//            few dozen codes to fall into step), one per lane; ONE set of tables for the whole wave.  Lit/len codes through a
//            two-level lookup table (round 6: 9 root bits, 852 entries, built by resolving every prefix ONCE with the 15 packed
//            compares against the canonical limits that rounds 2-5 ran for every token) — every lane the same two reads;
//   sync     every lane decodes the tokens that start in its segment — literal, end of block, or length + distance with
//            their extra bits — and reports where its last token ends; that is where the next lane's segment REALLY
//            starts.  The first pass walks only the tail of every segment (a first guess of every start), the second the whole
//            segment from the corrected start; repeated until no start moves (2.1 passes on average; pass k is certain to have
//            lanes 0..k-1 right, so nothing can hang).  The walk goes on behind an end-of-block code (stopping would cut the chain);
//   count    the same pass counts the bytes each lane's tokens produce and the matches that will have to WAIT (source in another
//            lane's output, or behind a match that waits itself): prefix sums give every lane its output offset and its place in
//            the round's one list of waiting matches, in stream order;
//   output   one more pass writes the bytes straight to HBM, a lane's literals four at a time as one dword store (round 6).  A run of
//            a byte the lane knows goes on a list and the wave fills all runs at once afterwards; a match whose whole source the lane has written itself is copied on the spot; a waiting
//            match leaves its position (16 bits) on the list and parks its length and distance in the first three bytes it will
//            produce;
//   waiting  64 at a time, each copied by ITS lane as soon as nothing it reads is still to come: destinations are ascending and
//            disjoint, so only the last earlier entry that starts in front of my source's end can reach into it (a binary search
//            over the lanes in front), and I wait until everything up to it is done; copies double the known period every step.
//            Stock zlib leaves ~110 such matches in a 4000-sample svb-zd record, ~190 per window of real signal, several hundred
//            in the key bytes of a long read or in raw-signal records: if a round's do not fit the list (768), only the lanes in
//            front go out and the next round starts behind them;
//   Adler    a last coalesced pass over the finished record.
// A record larger than one window takes several rounds; block headers (stored / fixed / dynamic) are parsed by the wave (the
// code-length sequence too: infl_cl_sequence_wave).  What this decoder declines — a payload slot that is too small, a single
// segment that expands to 64 KiB or holds more waiting matches than the list — it reports as INF_NEED_FALLBACK and the
// wave-per-record decoder redoes that record (k_inflate_fallback).  Same contract and status codes as inflate_dev.h otherwise.
#pragma once
#include "inflate_dev.h"

namespace s5 {

#ifndef S5_IP_SPAN
#define S5_IP_SPAN 4096
#endif
constexpr int IP_SPAN = S5_IP_SPAN;    // compressed bytes per round (a multiple of 16)
constexpr int IP_WIN_DW = IP_SPAN / 4 + 4;   // window dwords: the round's bytes + what a 32-bit look behind the last bit reaches, a multiple of four
#ifndef S5_IP_WAIT
#define S5_IP_WAIT 768
#endif
#ifndef S5_IP_FILL
#define S5_IP_FILL 64
#endif
constexpr int IP_WAIT = S5_IP_WAIT;    // waiting matches per round (the whole wave's, in stream order)
constexpr int IP_FILL = S5_IP_FILL;    // runs (distance-1 matches) per round that the whole wave fills afterwards
#ifndef S5_IP_TAIL
#define S5_IP_TAIL 224
#endif
constexpr uint32_t IP_TAIL = S5_IP_TAIL;   // bits of its segment a lane walks in the first pass
constexpr uint32_t IP_MINSEG = 384;    // shortest segment, bits
#ifndef S5_IP_LENSTEP
#define S5_IP_LENSTEP 4
#endif
constexpr uint32_t IP_LENSTEP = S5_IP_LENSTEP;   // a power of two: the length / distance part of the token loop runs every IP_LENSTEP-th step
constexpr int INF_NEED_FALLBACK = 8;
#ifndef S5_IP_DBITS
#define S5_IP_DBITS 8
#endif
constexpr int IP_DBITS = S5_IP_DBITS;  // primary distance lookup bits (>= 7: the code-length code's 7-bit table is built in the same storage)
static_assert(IP_DBITS >= 5 && IP_DBITS <= INF_DBITS, "");
#ifndef S5_IP_DBITS_SVB
#define S5_IP_DBITS_SVB 7
#endif
constexpr int IP_DBITS_SVB = S5_IP_DBITS_SVB;     // ... in the instantiation for svb-zd records: their distance codes are a handful of short ones
static_assert(IP_DBITS_SVB >= 5 && IP_DBITS_SVB <= INF_DBITS, "");   // (the code-length code's 7-bit table stands in the lit/len table's storage: cl_lut)

#ifndef S5_IP_WAIT_SVB
#define S5_IP_WAIT_SVB 256
#endif
constexpr int IP_WAIT_SVB = S5_IP_WAIT_SVB;       // ... in the instantiation for svb-zd records (below)
// Lit/len lookup table (round 6): 9 root bits + second-level tables for longer codes, 16-bit entries
//   literal   0x0000 | byte << 4 | len          length   0x4000 | (symbol - 257) << 4 | len
//   stop      0x8000 | len (end of block)       invalid  0x8010 | 15
//   sublink   0xC000 | first entry << 4 | bits of the second-level index
// 512 + 340 entries hold any code of <= 286 symbols and <= 15 bits (zlib's ENOUGH_LENS for a 9-bit root: a second-level table is
// sized by the longest code under its 9-bit prefix).
constexpr int IP_LROOT = 9;
#ifndef S5_IP_LSUB
#define S5_IP_LSUB 340
#endif
constexpr int IP_LSUB = S5_IP_LSUB;    // (340 holds every code; at least 288: the canonical-order symbols stand there while the table is built.  Round 6 also
                                       // measured the struct trimmed to 6640 bytes — 4032-byte window, 6-bit distance table, 192 waiting matches, 288 entries here —
                                       // for a 24th wave per CU: 24.80 against 24.70 ms per 1 M of our own records, 32.4 against 31.9 M stock-zlib records/s.
                                       // Not worth the narrower limits: profiles/r06_inflate_variants.txt)
// PAYB > 0 (round 6, the no-payload decode of short records): the window's storage also takes the UNCOMPRESSED record — PAYB bytes, written
// by the output pass, which then reads the compressed bits out of global memory (L2) instead: the record never leaves the CU
// (zlib_inflate_par<.., LDSOUT>).
template <int WAIT, int DB, int PAYB = 0>
struct InflParSharedT {                 // per wave: 6.9 KiB — the kernel's speed follows the number of resident waves (measured: + 4 KiB of
                                       // LDS per wave = + 33 % time), so nothing here is larger than it has to be
    alignas(16) uint32_t win[PAYB / 4 > IP_WIN_DW ? PAYB / 4 : IP_WIN_DW];     // window (16-byte aligned: filled 16 bytes per lane); the header parser uses its first INF_IW bytes
    union {
        uint16_t wq[WAIT];             // waiting match: position in the round's output (< 64 Ki); its length and distance wait in the
                                       // first three of the bytes it will produce — a match is at least three bytes long
        struct {
            uint16_t llut[32];         // (64 bytes of scratch for the header parser's symbol sort; no lit/len lookup table there)
            uint16_t ladj[16];         // while the table is built: index of a length's first symbol in lsym - its first code
        };
    };
    uint16_t dlut[1 << DB];
    union {
        uint16_t ltab[(1 << IP_LROOT) + IP_LSUB];
        struct {
            uint16_t ltab_root_[1 << IP_LROOT];
            uint16_t lsym[IP_LSUB];    // while the table is built: the symbols in canonical order (288 at most), where the second level will stand
        };
    };
    uint16_t dsym[32];
    uint16_t lcount[16], dcount[16];
    union {
        uint8_t lens[352];             // code lengths: dead once the tables stand ...
        struct {
            uint32_t fill_a[IP_FILL];  // ... run: position in the round's output | length << 20
            uint8_t fill_x[IP_FILL];   //     its byte
            uint32_t nfill;
        };
    };
#ifdef S5_IP_PAD
    uint8_t pad[S5_IP_PAD];            // tools only: how the kernel's time follows the number of resident waves
#endif
    static constexpr int N_WAIT = WAIT;
    static constexpr int DBITS = DB;
    static constexpr uint32_t LDS_PAY = PAYB;
    __device__ __forceinline__ uint16_t *cl_lut() { return ltab; }   // the code-length code's 7-bit table (dead before the lit/len table is built)
};
static_assert(IP_LSUB >= 288, "the canonical-order symbols are sorted where the second-level tables will stand");
// Two sizes of the waiting list (round 3).  The kernel's speed follows the number of resident waves, and at 80 VGPRs the register file
// holds 24 per CU: 7.4 KiB of LDS per wave allow 21, 6.4 KiB all 24 (measured on 1 M own records: 30.9 -> 28.6 ms).  What the list has to
// hold depends on what was compressed: stock zlib leaves ~110 waiting matches in a 4000-sample svb-zd record (a list of 256 takes a record
// per round; long reads take a few more rounds in their key bytes), but several hundred per window in a RAW-SIGNAL record (four tokens
// out of five are far matches: 768 entries, 15.8 ms per 8192 fixture records against 25 with 512).  The caller of s5gpu_decode_dev names
// the signal press, so svb-zd records get the small list; inflate-only calls and raw-signal records keep the large one.
using InflParShared = InflParSharedT<IP_WAIT, IP_DBITS>;
using InflParSharedSvb = InflParSharedT<IP_WAIT_SVB, IP_DBITS_SVB>;
#ifndef S5_IP_LDS_PAY
#define S5_IP_LDS_PAY 5360         // (8192 bytes of LDS per wave with the tables and a waiting list of 192: 20 waves per CU, five per SIMD — a byte more and
                                   // the compiler plans for four, 16 per CU, which costs 14 %: profiles/r06_inflate_variants.txt)
#endif
using InflParSharedLp = InflParSharedT<192, IP_DBITS_SVB, S5_IP_LDS_PAY>;
static_assert(sizeof(InflParSharedLp) <= 8192, "");
static_assert(S5_IP_LDS_PAY % 16 == 0, "");
static_assert(INF_IW <= IP_SPAN, "the header parser's window is the head of the round window");

// 32 bits of the window starting at bit p (two aligned dwords + one alignbit)
__device__ __forceinline__ uint32_t ip_peek(const uint32_t *win, uint32_t p) {
    const uint32_t w = p >> 5;
    return __builtin_amdgcn_alignbit(win[w + 1], win[w], p & 31u);
}
// window byte 0 = deflate byte `from` (a multiple of 4 keeps the copy aligned when src is); bytes at or past `total` read as zero
__device__ __forceinline__ void ip_load_window(uint32_t *win, const uint8_t *src, uint32_t from, uint32_t total) {
    const int lane = lane_id();
    const uint32_t avail = from < total ? total - from : 0;
#ifndef S5_IP_WIN4   // (round 4; -DS5_IP_WIN4: the dword form — 28.41 ms per 1 M own records against 28.05, K = 4096 batches 0.2376 ms against 0.2350)
    // sixteen bytes per lane and step: an UNALIGNED 16-byte global load (the record lies wherever the file put it) into an aligned 16-byte LDS store, for
    // every vector that lies inside the record; the dwords around the record's end (cut, masked, zero) the old way
    typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u4a __attribute__((ext_vector_type(4)));
    constexpr uint32_t NV = IP_WIN_DW / 4;
    static_assert(IP_WIN_DW % 4 == 0, "");
    const uint32_t nfull = min(avail >> 4, NV);
    for (uint32_t j = lane; j < nfull; j += 64) reinterpret_cast<u4a *>(win)[j] = *reinterpret_cast<const u4u *>(src + from + 16u * j);
    const uint32_t i0 = 4u * nfull;
#else
    const uint32_t i0 = 0;
#endif
    const uintptr_t addr = reinterpret_cast<uintptr_t>(src) + from;
    const uint32_t *g = reinterpret_cast<const uint32_t *>(addr & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(addr & 3) * 8;
    for (uint32_t i = i0 + lane; i < (uint32_t)IP_WIN_DW; i += 64) {
        uint32_t w = 0;
        if (4 * i < avail) {
            const uint32_t lo = g[i];
            w = lo;
            if (sh) w = (lo >> sh) | (g[i + 1] << (32 - sh));   // 8 readable bytes follow the record (C ABI)
            const uint32_t left = avail - 4 * i;
            if (left < 4) w &= (1u << (8 * left)) - 1;
        }
        win[i] = w;
    }
    wave_sync();
}

struct IpSeg {            // what a lane learns about its segment
    uint32_t cross;       // window bit where its last token ends (= the next segment's real start)
    uint32_t nout;        // bytes its tokens produce (up to the end-of-block code, if the segment holds one)
    uint32_t eob;         // 1: the segment holds the end-of-block code ...
    uint32_t eobpos;      // ... and this is the bit behind it
    uint32_t bad;         // 1: an invalid code (meaningless unless the segment was decoded from a real boundary)
    uint32_t nwait;       // matches that have to wait (source in another lane's output, or behind a match that waits itself); up to the end-of-block code
};
// Rounds 2-5 resolved EVERY token's lit/len code without a lookup table (with 64 lanes at 64 places of the stream some lane meets a code
// longer than a one-level table in almost every step): the next 15 bits, first bit on top, compared against the 15 left-justified canonical
// limits — wave-uniform values in scalar registers — the code's length being the number of limits not above it, plus one; ~30 VALU
// instructions, no loop, no divergence.  Round 6 keeps that chain for the CONSTRUCTION of a two-level table (ip_build_ltab: every 9-bit
// prefix and every second-level entry is resolved once) and the token walk reads the table: two dependent reads, the same two for every lane.
typedef short ip_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short ip_u2 __attribute__((ext_vector_type(2)));
// limit of length l = (first code of length l + codes of length l) << (15 - l), left-justified in 15 bits; kept minus one, two
// per word (lengths 2k+1 | 2k+2): eight packed 16-bit subtractions compare all fifteen, and no compare ever goes through VCC
// (on gfx950 a VALU write of VCC needs wait states before a VALU read of it: the cmp / addc form paid a nop per limit)
struct IpLimits { ip_s2 m1[8]; uint32_t base; int np; uint32_t end; };   // pairs from the shortest length in use on (lengths below it always count, the longest one's
                                                           // limit is the end of the code space and never does): np pairs matter; end: the code space in
                                                           // use, of 32768 (less: an incomplete code — what lies behind it is no code)
// (carrying the index adjustment of the code's length along in the same compare chain — a conditional move per limit instead of
// the T.ladj read — was measured: 9 % slower; the wave is short of issue slots, not of LDS latency)

// Decode the tokens that start in [st, end) of the window.  WRITE: also produce the bytes, at dst + obase (dst = the record's
// output at the round's first byte, o_abs0 bytes into the record; positions are relative to the round), and stop at the
// end-of-block code.  The bytes go straight to HBM: a lane writes its own run of positions, and L2 collects the lines.
// Without WRITE (the synchronisation passes) the walk goes on behind an end-of-block code: whatever follows is another
// block's header, garbage to this decoder, but walking on keeps the lane's end position self-synchronised — a lane that
// stopped there would cut the chain, and every lane behind it would have to be revived one pass at a time.
// The loop is wave-uniform (it runs while any lane has tokens left) and the length / distance part is entered only in steps in
// which some lane stands at a length code: a lane that is done, or at a literal, rides along predicated instead of parking
// behind nested exec masks.
// length of the code that starts the 15 bits v (first bit on top): the number of limits not above it, plus one; 16: behind the last code
// of an incomplete code.  Only the table construction resolves codes this way now (round 6).
__device__ __forceinline__ uint32_t ip_code_len(const IpLimits &L, uint32_t v) {
    const ip_s2 vv = {(short)v, (short)v};
    ip_u2 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) acc += __builtin_bit_cast(ip_u2, (ip_s2)(L.m1[k] - vv)) >> (unsigned short)15;   // sign bit: v >= limit
    if (L.np > 4) {   // (uniform: a lit/len code whose lengths span more than nine values; svb-zd records usually stay below)
#pragma unroll
        for (int k = 4; k < 6; k++) acc += __builtin_bit_cast(ip_u2, (ip_s2)(L.m1[k] - vv)) >> (unsigned short)15;
        if (L.np > 6) {
#pragma unroll
            for (int k = 6; k < 8; k++) acc += __builtin_bit_cast(ip_u2, (ip_s2)(L.m1[k] - vv)) >> (unsigned short)15;
        }
    }
    // (only the limits of the lengths in use are compared, and the longest one's never: what lies behind the last code of an incomplete
    // code is found by its own compare)
    return v >= L.end ? 16u : L.base + acc.x + acc.y;
}
// table entry of the code that starts the 15 bits v (its length: len <= 15, or 16: no such code)
template <class SH>
__device__ __forceinline__ uint32_t ip_entry_of(const SH &T, uint32_t v, uint32_t len) {
    if (len > 15u) return 0x801Fu;
    uint32_t idx = (uint32_t)((int)(short)T.ladj[len] + (int)(v >> (15u - len)));
    idx = min(idx, 287u);
    const uint32_t sym = (uint32_t)T.lsym[idx];
    return sym < 256u ? (sym << 4) | len : sym == 256u ? 0x8000u | len : 0x4000u | ((sym - 257u) << 4) | len;
}
// Lookup table of the lit/len code (layout: InflParSharedT).  Built by DECODING: entry i of the root is the code that starts the nine
// bits i — resolved once per entry with the packed compare chain against the canonical limits (ip_code_len: the decoder of rounds 2-5
// ran that chain for every token of every pass) — and a 9-bit prefix under which codes longer than nine bits stand links to a
// second-level table sized by the longest code under it.  Canonical codes grow with their length, so those prefixes form one ZONE per
// length 10..15, in that order: zs[k] = first prefix whose longest code has 10 + k bits, zb[k] = table index of that prefix's second
// level; both follow from the counts alone.  The symbols in canonical order stand where the second level goes: its entries are computed
// first (six per lane at most) and stored behind a barrier.  Returns 1 if the second level would not fit (never, for <= 286 symbols).
template <class SH>
__device__ __forceinline__ int ip_build_ltab(SH &T, const IpLimits &L) {
    const int lane = lane_id();
    uint32_t zs[7], zb[7];
    {
        uint32_t first = 0;
#pragma unroll
        for (int l = 1; l <= 15; l++) {
            const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)T.lcount[l]);
            if (l >= 10) zs[l - 10] = (first << (15 - l)) >> 6;
            if (l == 15) zs[6] = (first + c + 63u) >> 6;               // behind the last prefix that holds a code
            first = (first + c) << 1;
        }
        zb[0] = 1u << IP_LROOT;
#pragma unroll
        for (int k = 0; k < 6; k++) zb[k + 1] = zb[k] + ((zs[k + 1] - zs[k]) << (k + 1));
        if (zb[6] > (uint32_t)((1 << IP_LROOT) + IP_LSUB)) return 1;
    }
    // root: entry i is nine stream bits = nine code bits pfx, first bit on top.  The trips run over pfx IN CODE ORDER (the entry's index is
    // pfx bit-reversed: a scattered 16-bit store per trip): canonical codes grow with their length, so the prefixes under which long codes
    // stand are the LAST ones, and the forty instructions that place a link to the second level run in the one or two trips that hold such
    // prefixes instead of in all eight (stream order mixed them into every trip)
#pragma unroll 2
    for (int it = 0; it < (1 << IP_LROOT) / 64; it++) {
        const uint32_t pfx = (uint32_t)(it * 64 + lane), i = __brev(pfx) >> (32 - IP_LROOT);
        const uint32_t v = pfx << (15 - IP_LROOT);
        const uint32_t len = ip_code_len(L, v);
        uint32_t e = ip_entry_of(T, v, len);
        if ((uint32_t)(it * 64 + 63) >= zs[0] && len > (uint32_t)IP_LROOT && len <= 15u) {      // (zs[0]: the first prefix that holds a code of ten bits or more)
            uint32_t k = 0;
#pragma unroll
            for (int q = 1; q < 6; q++) k += pfx >= zs[q] ? 1u : 0u;
            uint32_t zsk = zs[0], zbk = zb[0];
#pragma unroll
            for (int q = 1; q < 6; q++) if (k == (uint32_t)q) { zsk = zs[q]; zbk = zb[q]; }
            e = 0xC000u | ((zbk + ((pfx - zsk) << (k + 1u))) << 4) | (k + 1u);
        }
        T.ltab[i] = (uint16_t)e;
    }
    // second level: entry j of zone k stands for prefix zs[k] + (j >> (k + 1)) followed by the k + 1 stream bits j & mask
    const uint32_t nsub = zb[6] - (1u << IP_LROOT);
    uint32_t sub[(IP_LSUB + 63) / 64];
#pragma unroll
    for (int it = 0; it < (IP_LSUB + 63) / 64; it++) {
        sub[it] = 0u;
        if (it * 64 >= (int)nsub) continue;                       // (uniform: our own records need 150-odd entries, three trips of six)
        const uint32_t j = (uint32_t)(it * 64 + lane) + (1u << IP_LROOT);
        uint32_t k = 0;
#pragma unroll
        for (int q = 1; q < 6; q++) k += j >= zb[q] ? 1u : 0u;
        uint32_t zsk = zs[0], zbk = zb[0];
#pragma unroll
        for (int q = 1; q < 6; q++) if (k == (uint32_t)q) { zsk = zs[q]; zbk = zb[q]; }
        const uint32_t rel = j - zbk, sb = k + 1u;
        const uint32_t pfx = zsk + (rel >> sb), tail = __brev(rel & ((1u << sb) - 1u)) >> (32u - sb);     // the tail's first stream bit on top
        const uint32_t v = (pfx << (15 - IP_LROOT)) | (tail << (15u - IP_LROOT - sb));
        sub[it] = ip_entry_of(T, v, ip_code_len(L, v));
    }
    wave_sync();
#pragma unroll
    for (int it = 0; it < (IP_LSUB + 63) / 64; it++) {
        const uint32_t j = (uint32_t)(it * 64 + lane);
        if (j < nsub) T.ltab[(1 << IP_LROOT) + j] = (uint16_t)sub[it];
    }
    wave_sync();
    return 0;
}

// Decode the tokens that start in [st, end) of the window.  WRITE: also produce the bytes, at dst + obase (dst = the record's
// output at the round's first byte, o_abs0 bytes into the record; positions are relative to the round), and stop at the
// end-of-block code.  The bytes go straight to HBM: a lane writes its own run of positions, and L2 collects the lines.
// Without WRITE (the synchronisation passes) the walk goes on behind an end-of-block code: whatever follows is another
// block's header, garbage to this decoder, but walking on keeps the lane's end position self-synchronised — a lane that
// stopped there would cut the chain, and every lane behind it would have to be revived one pass at a time.
// The loop is wave-uniform (it runs while any lane has tokens left) and the length / distance part is entered only in steps in
// which some lane stands at a length code: a lane that is done, or at a literal, rides along predicated instead of parking
// behind nested exec masks.
// GB (with WRITE, the output pass of zlib_inflate_par<.., LDSOUT>): the window's storage is being overwritten with the output, so the bits
// come out of GLOBAL memory — gsrc = the window's byte 0 in the record, dwords up to index gmaxw may be read — through a 64-bit buffer per
// lane: `cnt` valid bits, refilled 32 at a time from a dword that was requested one refill earlier.
// XONLY (without WRITE; round 6, third session): only `cross` is wanted — the first pass of a window, whose walk starts late in the segment and whose
// counts a later pass replaces in any case: no byte counting, no match classification, a match costs its bit lengths only.
template <bool WRITE, class SH, bool GB = false, bool XONLY = false>
__device__ __forceinline__ IpSeg ip_decode_segment(SH &T, uint32_t lenmask, uint32_t st, uint32_t end, uint32_t obase, uint32_t o_abs0,
                                                   uint8_t *dst, uint32_t wbase = 0, uint32_t wmax = 0, const uint8_t *gsrc = nullptr, uint32_t gmaxw = 0) {
    static_assert(!GB || WRITE, "");
    static_assert(!XONLY || !WRITE, "");
    IpSeg r;
    r.cross = st; r.nout = 0; r.eob = 0; r.eobpos = 0; r.bad = 0; r.nwait = 0;
    uint32_t p = st, o = obase;
    uint32_t lim = end;            // the lane walks while p < lim (a stop clears it)
    uint32_t wait_end = obase;     // output position behind this lane's last waiting match / pending run: nothing in front of it is in memory yet
    uint32_t lastb = 0x100u;       // WRITE: the byte this lane produced last (0x100: not known — nothing yet, or a waiting match)
    uint32_t unk_at = obase;       // a synchronisation pass only has to know WHETHER that byte is known (which matches will have to wait is counted there, so
                                   // that the output pass can put every lane's waiting matches at their place in ONE list in stream order): it is not exactly
                                   // when nothing has been produced since the start or since the last waiting match, i.e. while o == unk_at — nothing to keep
                                   // up to date per literal
    // Steps come in GROUPS of lenmask + 1 (round 6): inside a group a lane takes literals — two table reads, the same two for every lane
    // and every code (rounds 2-5: fifteen packed compares + two reads) — and a lane that meets anything else holds its position; no vote
    // is taken inside a group (a vote on a condition that is not a plain compare costs two vector instructions on this compiler).  The
    // length / distance part behind the group costs several literal steps and runs with the few lanes that stand at such a code.
    const uint32_t group = lenmask + 1u;
    // (every trip moves every live lane by a bit at least; the trip count is bounded all the same — a scalar counter — so that no table
    // content whatsoever can hang the wave: what is left over then counts as bad data)
    // WRITE: literals leave the lane FOUR AT A TIME, as one (unaligned) dword store — a store instruction of 64 lanes at 64 places costs the
    // memory pipeline the same whatever its width, and the output pass is bound by exactly that (profiles/r06_inflate_phases.txt).  acc holds
    // the acc8 / 8 literals in front of o that are not in memory yet; they go out as bytes before anything else is written or read back.
    uint32_t acc = 0, acc8 = 0;
    typedef uint32_t u1 __attribute__((aligned(1)));
    uint64_t gbuf = 0;
    uint32_t gcnt = 0, gnxt = 0, gw = 0;
    auto gld = [&](uint32_t i) { return *reinterpret_cast<const u1 *>(gsrc + 4u * min(i, gmaxw)); };
    auto refill = [&]() { if (gcnt < 32u) { gbuf |= (uint64_t)gnxt << gcnt; gcnt += 32u; gnxt = gld(gw); gw += 1u; } };
    if (GB) {
        const uint32_t w = st >> 5;
        const uint32_t d0 = gld(w), d1 = gld(w + 1u);
        gnxt = gld(w + 2u);
        gw = w + 3u;
        gbuf = (((uint64_t)d1 << 32) | d0) >> (st & 31u);
        gcnt = 64u - (st & 31u);
    }
    auto flush_acc = [&]() {
        if (acc8 >= 8u) dst[o - (acc8 >> 3)] = (uint8_t)acc;
        if (acc8 >= 16u) dst[o - (acc8 >> 3) + 1u] = (uint8_t)(acc >> 8);
        if (acc8 >= 24u) dst[o - (acc8 >> 3) + 2u] = (uint8_t)(acc >> 16);
        acc = 0; acc8 = 0;
    };
    for (uint32_t trips = 0; __ballot(p < lim); trips++) {
        if (trips > 8u * IP_SPAN) { if (p < lim) r.bad = 1; break; }
        uint32_t bits = 0, e = 0;
        bool held = false;
        for (uint32_t g = 0; g < group; g++) {
            if (GB) { refill(); bits = (uint32_t)gbuf; }
            else bits = ip_peek(T.win, p);                                  // (a lane that is done stands at most a token behind its limit: inside the window)
            e = T.ltab[bits & ((1u << IP_LROOT) - 1u)];
            if (e >= 0xC000u) e = T.ltab[((e >> 4) & 1023u) + __builtin_amdgcn_ubfe(bits, IP_LROOT, e & 15u)];
            // (the second read without the branch around it — index clamped, result selected — measured: no-payload decode 23.74 against 23.19 ms
            // per 1 M records, full decode 22.85 against 22.97.)
            // (letting a code longer than the root's nine bits HOLD like a length code, resolved behind the group, was measured: 9 % of our own
            // records' tokens are such codes, and what the held lanes lose in steps is what the second read costs — 13316 -> 13576 vector
            // instructions per record, profiles/r06_inflate_variants.txt)
            const bool act = p < lim;
            const bool lit = act && e < 0x4000u;
            held = act && !(e < 0x4000u);
            if (lit) {
                if (WRITE) {
#ifndef S5_IP_BYTE_STORES
                    if (GB) dst[o] = (uint8_t)(e >> 4);                      // (into LDS: a byte store is as good as any there)
                    else {
                        acc |= ((e >> 4) & 255u) << acc8;
                        acc8 += 8u;
                        if (acc8 == 32u) { *reinterpret_cast<u1 *>(dst + o - 3u) = acc; acc = 0; acc8 = 0; }
                    }
#else
                    dst[o] = (uint8_t)(e >> 4);
#endif
                }
                if (WRITE) lastb = e >> 4;
                p += e & 15u;
                if (!XONLY) o += 1u;
                if (GB) { gbuf >>= e & 15u; gcnt -= e & 15u; }
            }
        }
        if (__ballot(held)) {
            const uint32_t len = e & 15u;
            uint32_t adv = len, nby = 0;
            uint32_t gdone = 0;            // GB: bits of this token that have left the buffer already
            bool stop = false;
            if (held) {
                if (WRITE) flush_acc();
                // an invalid code (behind the last code of an incomplete code) is a stop like a bad length symbol: sym 0x3FF
                const uint32_t sym = e < 0x8000u ? 257u + ((e >> 4) & 31u) : (e & 0x10u) ? 0x3FFu : 256u;
                if (sym == 256u) {
                    if (!XONLY && !r.eob) { r.eob = 1; r.eobpos = p + len; r.nout = o - obase; }
                    if (WRITE) stop = true;
                } else if (sym - 257u >= 29u) {
                    r.bad = 1; stop = true; adv = 0;
                } else {
                    const uint32_t ls = sym - 257u;
                    uint32_t b2 = bits >> len;
                    // length code: 3..10 one each, then 4 codes per extra-bit count, 258 on its own — arithmetic, no table in memory
                    const uint32_t le = ls < 8u || ls == 28u ? 0u : (ls >> 2) - 1u;
                    const uint32_t mlen = (ls == 28u ? 258u : ls < 8u ? 3u + ls : 3u + ((4u + (ls & 3u)) << le)) + (b2 & ((1u << le) - 1u));
                    adv += le;
                    if (GB) { gbuf >>= adv; gcnt -= adv; gdone = adv; refill(); b2 = (uint32_t)gbuf; }   // (>= 32 valid bits again: a distance code with its extra bits takes 28)
                    else b2 = ip_peek(T.win, p + adv);
                    const uint32_t de = T.dlut[b2 & ((1u << SH::DBITS) - 1)];
                    uint32_t ds, dlen = de >> 5;
                    if (dlen) ds = de & 31u;
                    else {   // a distance code longer than the lookup table: canonical walk (rare)
                        uint32_t code = 0, first = 0, index = 0;
                        ds = 0xFFu;
                        for (dlen = 1; dlen <= 15; dlen++) {
                            code |= (b2 >> (dlen - 1)) & 1u;
                            const uint32_t c = T.dcount[dlen];
                            if (code < first + c) { ds = T.dsym[index + (code - first)]; break; }
                            index += c;
                            first = (first + c) << 1;
                            code <<= 1;
                        }
                    }
                    if (ds >= 30u) { r.bad = 1; stop = true; adv = 0; }
                    else {
                        adv += dlen;
                        b2 >>= dlen;
                        const uint32_t dx = ds < 4u ? 0u : (ds >> 1) - 1u;
                        const uint32_t mdist = (ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx)) + (b2 & ((1u << dx) - 1u));
                        adv += dx;
                        nby = mlen;
                        if (XONLY) { }
                        else if (WRITE && mdist > o_abs0 + o) { r.bad = 1; stop = true; }               // reaches in front of the record
                        else if (mdist == 1 && (WRITE ? lastb < 0x100u : o != unk_at)) {
                            // a run of the byte this lane produced last.  A lane that filled it itself would keep the other 63 waiting
                            // (the key bytes of an svb-zd record are runs of zeros, and they all sit in the first two segments): the run
                            // goes on a list and the wave fills all of them at once after the pass
                            if (WRITE) {
                                const uint32_t slot = atomicAdd(&T.nfill, 1u);
                                if (slot < (uint32_t)IP_FILL) { T.fill_a[slot] = o | (mlen << 20); T.fill_x[slot] = (uint8_t)lastb; }
                                else for (uint32_t k = 0; k < mlen; k++) dst[o + k] = (uint8_t)lastb;
                            }
                            wait_end = o + mlen;                                                      // (also when the list was full: both kinds of pass must agree)
                        } else if (o >= mdist && o - mdist >= wait_end) {                             // the whole source is bytes this lane has written: copy now
                            if (WRITE) {
                                for (uint32_t k = 0; k < mlen; k++) dst[o + k] = dst[o + k - mdist];
                                lastb = dst[o + mlen - 1];
                            }
                        } else {                                                                      // another lane's bytes, or bytes that wait themselves
                            if (WRITE) {
                                if (r.nwait < wmax) {
                                    T.wq[wbase + r.nwait] = (uint16_t)o;
                                    dst[o] = (uint8_t)(mlen - 3u); dst[o + 1] = (uint8_t)(mdist - 1u); dst[o + 2] = (uint8_t)((mdist - 1u) >> 8);
                                }
                                r.nwait++;
                            } else if (!r.eob) r.nwait++;
                            wait_end = o + mlen;
                            lastb = 0x100u;
                            unk_at = o + mlen;
                        }
                    }
                }
                p += adv;
                o += nby;
                if (GB && adv > gdone) { gbuf >>= adv - gdone; gcnt -= adv - gdone; }
                if (stop) lim = 0u;
            }
        }
    }
    if (WRITE) flush_acc();
    r.cross = p;
    if (!r.eob) r.nout = o - obase;
    return r;
}

// Inflate one zlib stream with one wave.  Returns a status of inflate_dev.h or INF_NEED_FALLBACK (nothing usable was written).
// LDSOUT (round 6; SH::LDS_PAY > 0): the output goes into the WINDOW'S OWN STORAGE (T.win, SH::LDS_PAY bytes) instead of `out`, for a caller
// that consumes the record on the spot (the no-payload decode: parse + svb-zd unpack by the same wave) — the uncompressed record never
// touches HBM.  Only a record whose one and only block ends inside the first window and whose output fits qualifies (every 4000-sample
// read does); anything else — a second window, a second block, a stored block, a longer record — is declined (INF_NEED_FALLBACK) before a
// byte is written, and the wave-per-record decoder takes it into its HBM slot.
template <class SH, bool LDSOUT = false>
__device__ __forceinline__ int zlib_inflate_par(SH &T, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap,
                                                uint32_t *out_len, uint32_t *dbg = nullptr, uint32_t tail = IP_TAIL) {
    static_assert(!LDSOUT || SH::LDS_PAY >= 1024u, "");
    if (LDSOUT) {
        out = reinterpret_cast<uint8_t *>(T.win);
        cap = min(cap, SH::LDS_PAY - 16u);
    }
    const int lane = lane_id();
    *out_len = 0;
    IPP_DECL
    if (in_len < 6) return INF_ERR_TRUNC;
    // the two header bytes and the Adler-32 trailer are fetched here and looked at behind the first window load / at the very end: their
    // round trips to memory run under the window's instead of in front of it and behind everything (bulk decode 26.9 -> 26.5 ms per 1 M
    // records on the sources of that moment)
    const uint32_t cmf = in[0], flg = in[1];
    const uint32_t want = ((uint32_t)in[in_len - 4] << 24) | ((uint32_t)in[in_len - 3] << 16) | ((uint32_t)in[in_len - 2] << 8) | in[in_len - 1];
    bool hdr_checked = false;
    const uint8_t *src = in + 2;
    const uint32_t total = in_len - 6;
    const uint64_t total_bits = 8ull * total;
    uint64_t pos = 0;            // bit position in the deflate data (uniform)
    uint32_t o = 0;              // bytes produced and flushed (uniform)
    uint32_t adA = 1, adB = 0;
    int last = 0;
    while (!last) {
        // ---- block header, with the wave-uniform reader of inflate_dev.h on the head of the window ----
        BitIn b;
        b.buf = 0; b.cnt = 0; b.wpos = 0;
        b.wbase = (uint32_t)(pos >> 3) & ~3u;
        ip_load_window(T.win, src, b.wbase, total);                           // the whole round window: the tokens behind the header are in it
        if (!hdr_checked) {
            hdr_checked = true;
            if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return INF_ERR_HEADER;
        }
        IPP(0)
        const uint32_t hdr_wb = b.wbase;
        bool win_fresh = true;
        bi_need32_u(b, T.win);
        bi_get(b, (int)(pos - 8ull * b.wbase));
        bi_need32_u(b, T.win);
        const uint32_t hdr = bi_get(b, 3);
        if (bi_consumed_bits(b) > total_bits) return INF_ERR_TRUNC;
        last = hdr & 1;
        const int type = hdr >> 1;
        if (type == 3) return INF_ERR_DATA;
        if (type == 0) {
            if (LDSOUT) return INF_NEED_FALLBACK;
            bi_get(b, b.cnt & 7);
            bi_need32_u(b, T.win);
            const uint32_t len = bi_get(b, 16), nlen = bi_get(b, 16);
            const uint32_t at = (uint32_t)(bi_consumed_bits(b) >> 3);
            if ((len ^ 0xFFFFu) != nlen) return INF_ERR_DATA;
            if ((uint64_t)at + len > total) return INF_ERR_TRUNC;
            if (o + len > cap) return INF_NEED_FALLBACK;
            for (uint32_t i = lane; i < len; i += 64) out[o + i] = src[at + i];
            o += len;
            pos = 8ull * ((uint64_t)at + len);
            continue;
        }
        int nl, nd;
        { const int rc = infl_block_tables<SH, 0, true, SH::DBITS>(T, src, total, total_bits, b, type, nl, nd IPP_PASS, dbg && dbg[3] >= 11 ? dbg[3] - 10 : 0u); if (rc != INF_OK) return rc; }
        if (dbg && dbg[3] >= 11) return INF_OK;
        pos = bi_consumed_bits(b);
        if (b.wbase != hdr_wb) win_fresh = false;                             // (the header parser slid its window: never, for a window that starts at the header)
        if (dbg && dbg[3] == 1) return INF_OK;       // tools/par_probe.py cut-off: block header and tables only
        uint32_t lenmask;
        {   // canonical limits and index adjustments of the lit/len code (uniform).  The limits go through LDS (the code lengths'
            // bytes are dead by now) so that the pairs can start at the shortest length in use
            IpLimits L;
            uint32_t first = 0, offs = 0;
            int minlen = 16, maxlen = 0;
            // how often the token loop enters its length / distance part: the code lengths tell how common such tokens are (the
            // share of the code space the length symbols own).  Few (svb-zd payloads: 2-5 %): every fourth step; a quarter or more
            // (raw-signal records: four tokens out of five are matches): every step
            {
                const uint8_t *ll = type == 1 ? T.lens : T.lens + 32;
                const uint32_t l = lane < 29 && 257 + lane < nl ? (uint32_t)ll[257 + lane] : 0u;
                const uint32_t mass = wave_sum(l ? 1u << (15u - l) : 0u);     // of 32768
                lenmask = mass >= 8192u ? 0u : mass >= 3072u ? 1u : IP_LENSTEP - 1u;
            }
            wave_sync();
            int16_t *lim = reinterpret_cast<int16_t *>(T.lens);               // lim[l], l = 1 .. 32
#pragma unroll
            for (int l = 1; l <= 15; l++) {
                const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)T.lcount[l]);
                if (c) { if (minlen == 16) minlen = l; maxlen = l; }
                if (lane == l) { T.ladj[l] = (uint16_t)(short)((int)offs - (int)first); lim[l] = (int16_t)((int)((first + c) << (15 - l)) - 1); }
                offs += c;
                first = (first + c) << 1;
            }
            if (lane >= 16 && lane < 33) lim[lane] = 0x7FFF;                  // there is no sixteenth length: never counted
            wave_sync();
#pragma unroll
            for (int k = 0; k < 8; k++) {
                L.m1[k].x = (short)__builtin_amdgcn_readfirstlane((int)lim[minlen + 2 * k]);
                L.m1[k].y = (short)__builtin_amdgcn_readfirstlane((int)lim[minlen + 2 * k + 1]);
            }
            L.base = (uint32_t)minlen;
            L.end = first >> 1;                                               // (first: twice the left-justified end of the 15-bit codes)
            L.np = (maxlen - minlen + 1) >> 1;                                // limits of minlen .. maxlen - 1
            wave_sync();
            if (ip_build_ltab(T, L)) { if (dbg) dbg[2] = 5; return INF_NEED_FALLBACK; }
        }
        IPP(6)
        if (dbg && dbg[3] == 5) return INF_OK;       // cut-off: + canonical limits and the lit/len lookup table
        // ---- the block's tokens, a window at a time ----
        for (;;) {
            if (pos >= total_bits) return INF_ERR_TRUNC;                      // no end-of-block code before the data ran out
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");            // bytes flushed by earlier rounds may be read back
            // window byte 0: the header's window again if the tokens start in its first quarter (the usual case: one load per record)
            uint32_t wb = (uint32_t)(pos >> 3) & ~3u;
            if (win_fresh && pos - 8ull * hdr_wb < 8ull * (IP_SPAN / 4)) wb = hdr_wb;
            else ip_load_window(T.win, src, wb, total);
            win_fresh = false;
            const uint32_t rel0 = (uint32_t)(pos - 8ull * wb);               // 0..31, or up to a quarter of the window when it is the header's
            const uint64_t rem = total_bits - pos;
            const uint32_t room = (uint32_t)(8 * IP_SPAN - 64) - (rel0 & ~31u);
            const uint32_t span = rem < (uint64_t)room ? (uint32_t)rem : room;
            uint32_t B = (span + 63u) / 64u;
            if (B < IP_MINSEG) B = IP_MINSEG;                                 // a lane needs a few dozen codes to fall into step: short streams get
                                                                              // fewer, longer segments (with 64-bit segments a record of 64 samples
                                                                              // took 56 passes, one lane corrected per pass); and a token, at most
                                                                              // 48 bits, never skips a segment
            const uint32_t wend = rel0 + span;
            const int nseg = (int)((span + B - 1u) / B);                      // lanes [0, nseg) have a segment (>= 1: span > 0)
            const uint32_t seg_lo = rel0 + (uint32_t)lane * B;
            const uint32_t seg_end = min(seg_lo + B, wend);
            // first pass: a lane only has to fall into step by the END of its segment, so it starts late in it (a lane that is not in
            // step by then is caught by the next pass, like any other wrong start).  Lane 0 too, although it knows its real start: the
            // pass takes as long as its longest walk, and all this pass is for is a first guess of every other lane's start — lane 0
            // walks its whole segment in the second pass, with everybody else
            // `tail` (uniform): IP_TAIL in a bulk launch.  In a `get` batch — one record per resident wave — the kernel's time is the SLOWEST record's,
            // and one record in ten needs a third pass because some lane was not in step after 224 bits (two passes of 4096 records: 0.062 ms of a
            // 0.154 ms kernel for a mean of 2.1 passes): there the first pass walks the WHOLE segment from its nominal start, after which nearly every
            // lane's end is right and the second pass is the last
            uint32_t st = min(seg_lo, wend);
            if (nseg > 1 && seg_end - st > tail) st = seg_end - tail;
            IpSeg sg;
            if (dbg) dbg[1]++;
            for (int pass = 0; pass < 67; pass++) {
                if (dbg) dbg[0]++;
                // (pass 0 of a window with more than one segment only guesses the other lanes' starts: it walks for its end position alone, and a
                // counting pass follows whatever it finds)
                const bool xo = pass == 0 && nseg > 1;
                if (xo) sg = ip_decode_segment<false, SH, false, true>(T, lenmask, st, st < seg_end ? seg_end : st, 0u, 0u, nullptr);
                else sg = ip_decode_segment<false>(T, lenmask, st, st < seg_end ? seg_end : st, 0u, 0u, nullptr);
                uint32_t ns = wave_prev(sg.cross, 0u);
                if (lane == 0) ns = rel0;
                else if (lane >= nseg) ns = st;                              // (a lane without a segment has nothing to correct: left alone, or
                const bool moved = ns != st;                                 // a change at the last segment's end would ripple on one lane per pass)
                st = ns;
                if (dbg && dbg[3] == 6) return INF_OK;                       // cut-off: + window load and the first (tail) pass
                if (!xo && !__ballot(moved)) break;
            }
            IPP(7)
            if (dbg && dbg[3] == 2) return INF_OK;   // cut-off: + window load and synchronisation passes
            // which lanes hold real tokens of this block, and where their bytes go
            const uint64_t eobs = __ballot(sg.eob != 0u && st < seg_end);
            const int eob_lane = eobs ? __ffsll((long long)eobs) - 1 : 64;
            int m = eob_lane < nseg ? eob_lane + 1 : nseg;                     // lanes [0, m) go out
            // every lane's waiting matches (counted by the last synchronisation pass) get their place in ONE list, in stream order.  If
            // they do not all fit, only the lanes in front go out in this round and the next round starts behind them (stock zlib
            // leaves ~190 matches in 4 KiB of an svb-zd stream of real signal, and several hundred in the key bytes of a long read)
            const uint32_t w_all = st < seg_end && lane < m ? sg.nwait : 0u;
            const uint32_t wincl = wave_incl_add(w_all);
            const uint32_t n_all = st < seg_end && lane < m ? sg.nout : 0u;
            const uint32_t nincl = wave_incl_add(n_all);
            {
                // (the sums never decrease: the lanes that fit are a prefix; list positions have 16 bits)
                const uint64_t over = __ballot(wincl > (uint32_t)SH::N_WAIT || nincl > 0xFFFFu);
                const int mfit = over ? __ffsll((long long)over) - 1 : 64;
                if (mfit < m) m = mfit;
                if (m == 0) { if (dbg) dbg[2] = 3; return INF_NEED_FALLBACK; }   // one segment over the list's capacity or with >= 64 KiB of output
            }
            const uint32_t w_act = lane < m ? w_all : 0u;
            const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)wincl, m - 1);
            const uint32_t n_act = lane < m ? n_all : 0u;
            const uint32_t incl = lane < m ? nincl : 0u;
            const uint32_t round_out = (uint32_t)__builtin_amdgcn_readlane((int)nincl, m - 1);
            if (o + round_out > cap) { if (dbg) dbg[2] = 2; return INF_NEED_FALLBACK; }        // payload slot too small: the old decoder reports the size needed
            if (LDSOUT && !(last && m - 1 == eob_lane && o == 0u)) return INF_NEED_FALLBACK;     // (the output is about to overwrite the window: this must be all there is)
            const uint32_t obase = incl - n_act;
            uint8_t *dst = out + o;
            // ---- output pass ----
            if (lane == 0) T.nfill = 0;
            wave_sync();
            IpSeg wr;
            wr.nwait = 0; wr.bad = 0; wr.eob = 0; wr.eobpos = 0; wr.cross = st; wr.nout = 0;
            IPP(8)
            if (LDSOUT) {
                wave_sync();                                                     // (every lane has read what it needs of the window: the tables stand elsewhere)
                const uint32_t gavail = total > wb ? total - wb : 0u;            // bytes of the record from the window's byte 0 on; 8 readable bytes follow the record (C ABI)
                if (lane < m && st < seg_end)
                    wr = ip_decode_segment<true, SH, true>(T, lenmask, st, seg_end, obase, o, dst, wincl - w_act, w_act, src + wb, (gavail + 4u) >> 2);
            } else if (lane < m && st < seg_end) wr = ip_decode_segment<true>(T, lenmask, st, seg_end, obase, o, dst, wincl - w_act, w_act);
            IPP(9)
            if (dbg && dbg[3] == 7) return INF_OK;   // cut-off: + output pass (without runs and waiting matches)
            if (__ballot(wr.bad != 0u)) return INF_ERR_DATA;
            if (__ballot(wr.nwait != w_act)) { if (dbg) dbg[2] = 4; return INF_NEED_FALLBACK; }   // (the two kinds of pass disagree: never seen)
            wave_sync();
            {   // ---- runs: one per lane, 64 at a time (they depend on nothing) ----
                const uint32_t nf = min(__builtin_amdgcn_readfirstlane(T.nfill), (uint32_t)IP_FILL);
                for (uint32_t f0 = 0; f0 < nf; f0 += 64) {
                    if (f0 + (uint32_t)lane < nf) {
                        const uint32_t a = T.fill_a[f0 + lane], op = a & 0xFFFFFu, mlen = a >> 20;
                        const uint32_t x = T.fill_x[f0 + lane], x4 = x * 0x01010101u;
                        uint8_t *q = dst + op;
                        typedef uint32_t u1 __attribute__((aligned(1)));
                        if (mlen >= 4) {                                     // dwords wherever they start; the last one overlaps
                            for (uint32_t k = 0; k + 4 < mlen; k += 4) *reinterpret_cast<u1 *>(q + k) = x4;
                            *reinterpret_cast<u1 *>(q + mlen - 4) = x4;
                        } else if (mlen) { q[0] = (uint8_t)x; q[mlen >> 1] = (uint8_t)x; q[mlen - 1] = (uint8_t)x; }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            wave_sync();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            IPP(10)
            // ---- waiting matches: 64 at a time, each copied by ITS lane as soon as nothing it reads is still to come.  The list is
            // in stream order, so the destinations — starts and ends — are ascending and disjoint: the entries that reach into my
            // source are a contiguous range of the lanes in front of me, from the first whose end lies behind my source's start to
            // the last that starts in front of my source's end (two binary searches over the lanes); I go as soon as exactly those
            // are done.  Stock zlib leaves ~110 such matches in an svb-zd record of 4000 samples (half of them in the key bytes):
            // ~7 dependency steps per 64 entries (waiting for everything in front of the last one, as rounds 2-3 did, took 13).
            for (uint32_t c0 = 0; c0 < wtot; c0 += 64) {
                if (dbg && dbg[3] == 4) break;                   // probe cut-off: what the waiting matches cost (the output is wrong without them)
                const uint32_t j = c0 + (uint32_t)lane;
                const bool have = j < wtot;
                const int op = have ? (int)T.wq[j] : 0;
                uint32_t dist = 1u;
                int mlen = 0;
                if (have) { const uint8_t *h = dst + op; mlen = 3 + (int)h[0]; dist = 1u + (uint32_t)h[1] + ((uint32_t)h[2] << 8); }
                const int ss = op - (int)dist, se = min(ss + mlen, op);          // source bytes [ss, se) exist before this match starts writing
                int lo = 0, hi = lane;
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi) >> 1;
                    const int v = __builtin_amdgcn_ds_bpermute(mid << 2, op);
                    if (lo < hi) { if (v < se) lo = mid + 1; else hi = mid; }
                }
                const int cand = lo - 1;                                          // the last entry in front of me that starts before se
                int lo2 = 0, hi2 = lane;                                          // the first one that ends behind ss
                const int e_op = op + mlen;
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo2 + hi2) >> 1;
                    const int v = __builtin_amdgcn_ds_bpermute(mid << 2, e_op);
                    if (lo2 < hi2) { if (v <= ss) lo2 = mid + 1; else hi2 = mid; }
                }
                const uint64_t need = cand >= lo2 ? (2ull << cand) - (1ull << lo2) : 0ull;
                uint64_t donem = ~__ballot(have);
                while (~donem) {
                    const bool ready = !((donem >> lane) & 1ull) && !(need & ~donem);
                    if (ready) {
                        // the bytes in front of q are periodic with period dist; every step copies as much as is known without
                        // reading what it writes, and then twice as much is known (a run at distance 1 takes 9 steps, not 258)
                        uint8_t *q = dst + op;
                        int k = 0, d = (int)dist;
                        while (k < mlen) {
                            const int nb = min(mlen - k, d);
                            copy_ends(q + k, q + k - d, nb);
                            k += nb;
                            d += d;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    wave_sync();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    donem |= __ballot(ready);
                    if (dbg) dbg[2] += 0x10000u;                 // (probe builds: dependency steps of the waiting matches in the high half, ...
                }
                if (dbg) dbg[2] += 1u;                           // ... batches of 64 in the low half)
            }
            IPP(11)
            o += round_out;
            // where the round ended: behind the end-of-block code, or at the last emitted lane's last token
            const uint32_t endbit = (uint32_t)__builtin_amdgcn_readlane((int)wr.cross, m - 1);
            pos = 8ull * wb + endbit;
            if (m - 1 == eob_lane) break;                                    // the block is done
            wave_sync();
        }
        if (pos > total_bits) return INF_ERR_TRUNC;
    }
    *out_len = o;
    if (dbg && dbg[3] == 3) return INF_OK;           // cut-off: everything but the Adler-32 pass
    {   // Adler-32 over the finished record (coalesced dwords of bytes that are still in L2; the payload slot is 16-byte aligned),
        // in pieces short enough for 64-bit sums: A += sum x, B += n A_old + sum (n - i) x_i
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t from = 0; from < o; from += 1u << 20) {
            const uint32_t n = min(o - from, 1u << 20);
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(out + from);
            uint32_t sa = 0;
            uint64_t sb = 0;
#ifndef S5_IP_ADLER4   // four dwords per lane and step (the slot is 16-byte aligned, `from` a multiple of 2^20); round 4: with the 16-byte window load
                       // 28.41 -> 27.74 ms per 1 M own records (35.2 -> 36.1 M reads/s), K = 4096 batches 0.2376 -> 0.2322 ms (-DS5_IP_ADLER4: the dword form)
            const uint4 *w128 = reinterpret_cast<const uint4 *>(w32);
            const uint32_t nvec = n / 16;
            for (uint32_t i = lane; i < nvec; i += 64) {
                const uint4 v = w128[i];
                const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t s4 = __builtin_amdgcn_udot4(ww[k], 0x01010101u, 0u, false);
                    sa += s4;
                    sb += (uint64_t)((n - 16 * i - 4 * k) * s4 - __builtin_amdgcn_udot4(ww[k], 0x03020100u, 0u, false));
                }
            }
            for (uint32_t i = 4 * nvec + lane; i < n / 4; i += 64) {
#else
            for (uint32_t i = lane; i < n / 4; i += 64) {
#endif
                const uint32_t w = w32[i];
                const uint32_t s4 = __builtin_amdgcn_udot4(w, 0x01010101u, 0u, false);
                sa += s4;
                sb += (uint64_t)((n - 4 * i) * s4 - __builtin_amdgcn_udot4(w, 0x03020100u, 0u, false));
            }
            if ((uint32_t)lane < (n & 3u)) { const uint32_t i = (n & ~3u) + lane; const uint32_t x = out[from + i]; sa += x; sb += (uint64_t)(n - i) * x; }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
            adB = (uint32_t)(((uint64_t)adB + (uint64_t)n * adA + sb) % 65521u);
            adA = (adA + sa) % 65521u;
        }
    }
    IPP(12)
    IPP_FLUSH
    if (((adB << 16) | adA) != want) return INF_ERR_ADLER;
    return INF_OK;
}

}  // namespace s5
