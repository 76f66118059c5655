// lz_dev.h — DEFLATE block encoder with a real LZ77 matcher (distance codes of any size), for payloads whose redundancy is
// NOT run-length: records written with signal press "none" (raw int16 samples: slow5tools view -s none -c zlib,
// /root/reference/test/test_view.sh:89-91, every v0.1.0 file) and byte ranges handed to the solo zlib press.  svb-zd / ex-zd
// payloads keep the run-length tokeniser of deflate_dev.h (an LZ77 window buys < 1 % there, SURVEY.md §0 finding 6).
//
// What the data asks for (tools/lz_probe2.py on the reference's exp_1_lossless_zlib.blow5, whose zlib-6 stream covers 82 % of
// its bytes with matches of length 3-5 at distances spread evenly up to 32 K): a raw signal is noise around slowly moving
// levels, so almost every 4-byte string (two samples) has occurred somewhere in the last 16 K samples.  Finding SOME
// earlier occurrence of the next 4 bytes matters, the search depth hardly does, and 3-byte matches at long distances cost
// more than the literals they replace.  Hence: minimum match 4; a 4-way table of most recent positions keyed by a 13-bit hash
// of the 4 bytes (64 KiB of LDS), plus the four nearest distances 1..4; greedy parse (no lazy evaluation).  Measured on that
// record: 1.02 x zlib level 6's size (zlib level 1: 1.036, level 2: 1.029), against 1.127 for run-length + Huffman alone.
//
// One workgroup per record, 16 KiB blocks, window = previous block + current block, both in LDS:
//   match   rounds of 256 positions (one per lane): look up the table (positions of EARLIER rounds), verify by comparing
//           bytes in the window — table entries are 16-bit positions and may be stale, the comparison is what makes a match —
//           then enter the round's own positions, wave k into way k (same-address stores of one instruction resolve in a fixed
//           lane order and different waves never share a way: the output is deterministic)
//   parse   greedy and serial by nature; here: a lane parses its own 64 positions from an entry offset, hands the overhang of
//           its last match to the next lane, and the workgroup iterates until no entry offset changes (parses started at
//           different offsets fall into step after a few tokens, so two or three rounds settle everything; an all-equal
//           payload, one match chain from end to end, takes as many rounds as there are lanes under a match)
//   code    histograms of lit/len and distance symbols; both alphabets get their lengths from assign_lengths_wave on two
//           waves at once; the header and the token bits as in deflate_dev.h, with real distance codes
#pragma once
#include "deflate_dev.h"

namespace s5 {

constexpr int LZ_BLK = DEFL_BLK;
constexpr int LZ_MINLEN = 4, LZ_WAYS = 4;          // (four ways = four waves: wave k fills way k)

// Two shapes of the same encoder.  LzLong: records of any length — 16 KiB blocks, the previous block kept as history, a 13-bit table
// (64 KiB): ~150 KiB of LDS, one workgroup per CU.  LzShort (round 3): records whose payload fits ONE block of 8 KiB (a 4000-sample
// raw-signal record is 8086 bytes) need no history, a quarter of the positions and a table an eighth the size, and the table is dead
// when the bit buffer comes to life, so the two share storage: 38 KiB, four workgroups per CU — the matcher is a chain of dependent
// LDS round trips, so what it gains is resident waves (tools/lz_time.py).
// TN = threads of the workgroup.  LzLong holds a CU by itself (150 KiB of LDS) and every phase of it is a chain of dependent LDS round trips —
// table lookup -> candidates -> compare; a lane's serial greedy parse; its token loops — so the only thing that hides the latency is more
// waves ON that one workgroup: round 4 runs it with S5_LZ_TN threads (16 waves at 1024) instead of 256; a lane then owns BLK / TN positions.
#ifndef S5_LZ_TN
#define S5_LZ_TN 1024
#endif
// tools/lz_phases.py (variant build -DS5_LZPROBE): clock ticks thread 0 of a workgroup spends in each phase of deflate_block_lz, summed over the batch
#ifdef S5_LZPROBE
__device__ unsigned long long g_lzprobe[16];
#define LZP_DECL unsigned long long lzp_t = __builtin_readcyclecounter();
#define LZP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_lzprobe[i], n_ - lzp_t); lzp_t = n_; }
#else
#define LZP_DECL
#define LZP(i)
#endif
template <int BLK_, int HBITS_, bool HIST_, int TN_ = NT>
struct LzCfg { static constexpr int BLK = BLK_, HBITS = HBITS_, TN = TN_; static constexpr bool HIST = HIST_; static constexpr int WOFF = HIST_ ? BLK_ : 0; };
using LzLong = LzCfg<LZ_BLK, 13, true, S5_LZ_TN>;
using LzShort = LzCfg<8192, 10, false>;


template <class C>
struct LzSharedT {
    alignas(16) uint8_t win[C::WOFF + C::BLK + 16];        // [previous block |] current block, slack for dword reads at the end
    union {
        alignas(16) uint16_t table[(1 << C::HBITS) * LZ_WAYS];  // position & 0xFFFF of the most recent occurrences of a hash
        alignas(16) uint32_t obuf_alias[C::HIST ? 4 : (C::BLK + 64) / 4];   // LzShort: the bit buffer lives where the table was
    };
    alignas(16) uint16_t D[C::BLK + 8];                     // per position: match distance (0 none); after the parse D[p + 1] = length of the match starting at p
    uint16_t entry[C::TN + 2];
    uint32_t blcount_d[16];
    uint32_t bins_d[64];
    // Rounds of more than 256 positions: the positions of the CURRENT round are not in `table` yet (it is filled behind the round), and with 1024
    // of them a payload of period 255 would see nothing for its first four periods.  cur[] holds, per 10-bit hash, the EARLIEST position of the
    // round with that hash (an LDS atomic min, so it does not depend on the waves' timing): one more candidate for every later position of the round.
    uint32_t cur[C::TN > 256 ? 1024 : 1];   // (0xFFFF - round) << 16 | position in the block; cleared per block
};
using LzShared = LzSharedT<LzLong>;

// four bytes at any LDS byte address (two aligned dwords + one alignbit)
__device__ __forceinline__ uint32_t lds_load32u(const uint8_t *p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbit(q[1], q[0], (uint32_t)(a & 3) * 8u);
}
// length of the match between window positions a (current) and b < a, at most maxl; the first LZ_MINLEN bytes are known equal
__device__ __forceinline__ uint32_t lz_match_len(const uint8_t *win, uint32_t a, uint32_t b, uint32_t maxl, uint32_t from = LZ_MINLEN) {
    uint32_t l = from;
    while (l + 4 <= maxl) {
        const uint32_t x = lds_load32u(win + a + l) ^ lds_load32u(win + b + l);
        if (x) return l + ((uint32_t)__ffs((int)x) - 1u) / 8u;
        l += 4;
    }
    while (l < maxl && win[a + l] == win[b + l]) l++;
    return l;
}
__device__ __forceinline__ void lz_len_sym(uint32_t len, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    const uint32_t l = len - 3;
    eb = 0; ev = 0;
    if (len == 258) sym = 285;
    else if (l < 8) sym = 257 + l;
    else {
        const uint32_t nb = 29u - (uint32_t)__clz((int)l);
        sym = 261 + 4 * nb + ((l >> nb) & 3u);
        eb = nb;
        ev = l & ((1u << nb) - 1u);
    }
}
__device__ __forceinline__ void lz_dist_sym(uint32_t dist, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    const uint32_t d = dist - 1;
    if (d < 4) { sym = d; eb = 0; ev = 0; return; }
    const uint32_t nb = 30u - (uint32_t)__clz((int)d);
    sym = 2 * nb + 2 + ((d >> nb) & 1u);
    eb = nb;
    ev = d & ((1u << nb) - 1u);
}

// length of the match the matcher found at block position pos (d = X.D[pos], not 0)
template <class C>
__device__ __forceinline__ uint32_t lz_len_of(const LzSharedT<C> &X, uint32_t d, int pos, int len) {
    if (!C::HIST) {
        const uint32_t lc = d >> 13;
        if (lc < 7u) return lc + 4u;
        d &= 0x1FFFu;
    }
    return lz_match_len(X.win, (uint32_t)C::WOFF + (uint32_t)pos, (uint32_t)C::WOFF + (uint32_t)pos - d, min(258u, (uint32_t)(len - pos)));
}

// Encode the `len` <= LZ_BLK bytes at X.win + LZ_BLK as one DEFLATE block into the bit buffer.  `hist` = bytes of history in
// front of them in the window (0 for a record's first block, LZ_BLK afterwards), abs0 = position of the block in the record.
// Same contract as deflate_block MODE 2: the bit buffer is cleared here and the stream's partial word travels in z.carry.
template <class C>
__device__ __forceinline__ void deflate_block_lz(DeflShared &S, LzSharedT<C> &X, uint32_t *obuf, uint32_t obuf_words, int len, uint32_t hist,
                                                 uint32_t abs0, bool final, ZOut &z, uint32_t &adA, uint32_t &adB) {
    const int tid = threadIdx.x;
    constexpr uint32_t WOFF = (uint32_t)C::WOFF;
    const uint8_t *cur = X.win + WOFF;
    constexpr int TN = C::TN;
    static_assert(TN % 256 == 0 && TN / 64 <= 16, "S.ws holds 16 wave sums");
    constexpr int K = C::BLK / TN;   // positions per lane (64 / 32 at 256 threads, 16 at 1024)
    const int base = tid * K;
    if (len == 0) {   // empty stream: a fixed block holding only end-of-block
        for (uint32_t i = tid; i < obuf_words; i += TN) obuf[i] = 0;
        __syncthreads();
        if (tid == 0) { obuf[0] = z.carry; put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (1u << 1), 10); }
        z.bitpos += 10;
        __syncthreads();
        return;
    }
    LZP_DECL
    for (int i = tid; i < 320; i += TN) S.freq[i] = 0;
    if (tid < 8) S.red[tid] = 0;
    if (tid < 20) S.clfreq[tid] = 0;
    // ---- match: rounds of TN positions ----
    const int nround = (len + TN - 1) / TN;
    // (wide rounds) a round's positions enter cur[] BEFORE the round: round 0's here, round r + 1's under the last table writes of round r, so
    // that the barrier that ends those writes also completes cur[] — five barriers per round of 1024 positions instead of six
    uint32_t w_n = 0, h_n = 0;                               // my position of the coming round: its four bytes and their hash
    auto enter_cur = [&](int r) {
        const int i = r * TN + tid;
        if (i + LZ_MINLEN <= len) {
            w_n = lds_load32u(X.win + WOFF + (uint32_t)i);
            h_n = (w_n * 2654435761u) >> (32 - C::HBITS);
            atomicMin(&X.cur[h_n & 1023u], ((0xFFFFu - (uint32_t)r) << 16) | (uint32_t)i);
        }
    };
    if constexpr (TN > 256) {
        for (int i = tid; i < 1024; i += TN) X.cur[i] = 0xFFFFFFFFu;
        __syncthreads();
        enter_cur(0);
    }
    for (int r = 0; r < nround; r++) {
        const int i = r * TN + tid;
        const bool act = i + LZ_MINLEN <= len;
        uint32_t w = 0, h = 0, best_l = 0, best_d = 0;
        const uint32_t p16 = (abs0 + (uint32_t)i) & 0xFFFFu;
        uint32_t dcur = 0;                                   // distance to the round's earliest position with my hash (0: none in front of me)
        if constexpr (TN > 256) {
            const uint32_t rkey = (0xFFFFu - (uint32_t)r) << 16;
            __syncthreads();                                 // cur[] holds this round; the table holds every round before it
            if (act) {
                const uint32_t ce = X.cur[h_n & 1023u];
                if ((ce & 0xFFFF0000u) == rkey && (ce & 0xFFFFu) < (uint32_t)i) dcur = (uint32_t)i - (ce & 0xFFFFu);
            }
        }
        if (act) {
            const uint32_t widx = WOFF + (uint32_t)i;
            const uint32_t maxl = min(258u, (uint32_t)(len - i)), avail = (uint32_t)i + hist;
            // my four bytes, the four in front of them and the four behind them: twelve consecutive bytes = four ALIGNED dwords (neighbouring lanes read
            // the same ones: no bank conflicts) and three byte-aligns in registers, instead of three unaligned loads of two dwords each.
            // prev4: the candidates at distances 1..4 come out of (prev4 : w) in registers (round 3: a third of the matcher's LDS reads);
            // w2: a verified candidate's length is settled by ONE more dword compare in nearly every case (raw signals match over 4 - 5
            // bytes); only a candidate that agrees over all eight bytes walks on
            uint32_t prev4, w2;
            if (WOFF > 0 || widx >= 4u) {
                const uint32_t a = widx - 4u;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(X.win + (a & ~3u));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], sh = (a & 3u) * 8u;
                prev4 = __builtin_amdgcn_alignbit(q1, q0, sh);
                w = __builtin_amdgcn_alignbit(q2, q1, sh);
                w2 = __builtin_amdgcn_alignbit(q3, q2, sh);
            } else {
                prev4 = 0u;
                w = lds_load32u(X.win + widx);
                w2 = lds_load32u(X.win + widx + 4);
            }
            h = (w * 2654435761u) >> (32 - C::HBITS);
            const uint2 e = *reinterpret_cast<const uint2 *>(&X.table[h * LZ_WAYS]);
            constexpr int NC = TN > 256 ? 9 : 8;             // table ways 0..3, distances 1..4, (wide rounds) the round's earliest position
            // (loading the first dwords of all far candidates side by side, before any is looked at, was measured: 38.1 -> 32.8 GB/s on the long shape,
            // 64.6 -> 61.4 on the short one — the matcher is bound by LDS throughput now, not by the candidates' round trips)
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const uint32_t ent = k == 0 ? e.x & 0xFFFFu : k == 1 ? e.x >> 16 : k == 2 ? e.y & 0xFFFFu : e.y >> 16;
                const uint32_t d = k < 4 ? (p16 - ent) & 0xFFFFu : k < 8 ? (uint32_t)(k - 3) : dcur;
                bool hit;
                if (k < 4 || k == 8) hit = d >= (k == 8 ? 5u : 1u) && d <= avail && d <= 32768u && lds_load32u(X.win + widx - d) == w;
                else hit = avail >= 4u && __builtin_amdgcn_alignbyte(w, prev4, 4u - d) == w;
                if (hit) {
                    const uint32_t x = (k < 4 || k == 8 ? lds_load32u(X.win + widx - d + 4) : __builtin_amdgcn_alignbyte(w2, w, 4u - d)) ^ w2;
                    uint32_t l = x ? 4u + ((uint32_t)__ffs((int)x) - 1u) / 8u : 8u;
                    if (l >= maxl) l = maxl;                                   // (the bytes behind the block's end do not count)
                    else if (!x) l = lz_match_len(X.win, widx, widx - d, maxl, 8u);
                    if (l > best_l || (l == best_l && d < best_d)) { best_l = l; best_d = d; }
                }
            }
        }
        // LzShort: a distance needs 13 bits, the three above them carry the match length (4 .. 10; 7 = 11 or more: measure again), so the parse and
        // the passes behind it need not walk the window again for nearly every match (raw signals: matches of 4 - 5 bytes)
        if (i < len) X.D[i] = best_l >= (uint32_t)LZ_MINLEN ? (uint16_t)(best_d | (C::HIST ? 0u : (min(best_l, 11u) - 4u) << 13)) : (uint16_t)0;
        __syncthreads();
        // wave k fills way k & 3: see the header note on determinism.  With more than four waves the waves that share a way write in turn
        // (lower positions first), a barrier between them, so that the survivor of a bucket does not depend on the waves' timing
#pragma unroll
        for (int g = 0; g < TN / 256; g++) {
            if (act && (wave_id() >> 2) == g) X.table[h * LZ_WAYS + (wave_id() & 3)] = (uint16_t)p16;
            if (TN == 256 || g + 1 < TN / 256) __syncthreads();
        }
        if constexpr (TN > 256) { if (r + 1 < nround) enter_cur(r + 1); }   // (the barrier at the top of the next round ends the last group's writes too)
    }
    LZP(0)
    // ---- parse: lane-local greedy parses, entry offsets handed on until they settle ----
    uint64_t tok = 0, mat = 0;
    {
        uint32_t my_entry = 0;
        const int end = min(base + K, len);
        for (int guard = 0; guard <= TN + 1; guard++) {
            tok = 0; mat = 0;
            int pos = base + (int)my_entry;
            while (pos < end) {
                const uint32_t d = X.D[pos];
                const uint64_t bit = 1ull << (pos - base);
                tok |= bit;
                if (d) {
                    mat |= bit;
                    pos += (int)lz_len_of<C>(X, d, pos, len);
                } else pos++;
            }
            X.entry[tid + 1] = (uint16_t)(pos > base + K ? pos - (base + K) : 0);
            __syncthreads();
            const uint32_t ne = tid ? X.entry[tid] : 0u;
            const int changed = ne != my_entry;
            my_entry = ne;
            LZP(1)
#ifdef S5_LZPROBE
            if (threadIdx.x == 0) atomicAdd(&g_lzprobe[9], 1ull);
#endif
            if (!__syncthreads_or(changed)) break;
        }
        // the settled parse: park every match's length behind its distance (position p + 1 lies inside the match)
        uint64_t m = mat;
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int pos = base + j;
            const uint32_t d = X.D[pos];
            X.D[pos + 1] = (uint16_t)lz_len_of<C>(X, d, pos, len);
            if (!C::HIST) X.D[pos] = (uint16_t)(d & 0x1FFFu);              // the plain distance for the passes behind
        }
    }
    __syncthreads();
    LZP(2)
    // ---- histograms, Adler-32 partial sums ----
    uint32_t nextra = 0, a_sum = 0, b_sum = 0;
    {
        uint64_t t = tok;
        while (t) {
            const int j = __ffsll((long long)t) - 1;
            t &= t - 1;
            const int pos = base + j;
            if ((mat >> j) & 1ull) {
                uint32_t ls, le, lv, ds, de, dv;
                lz_len_sym(X.D[pos + 1], ls, le, lv);
                lz_dist_sym(X.D[pos], ds, de, dv);
                atomicAdd(&S.freq[ls], 1u);
                atomicAdd(&S.freq[DOFF + ds], 1u);
                nextra += le + de;
            } else atomicAdd(&S.freq[cur[pos]], 1u);
        }
        const int kk = max(0, min(K, len - base));
        int j = 0;
        for (; j + 4 <= kk; j += 4) {
            const uint32_t wd = *reinterpret_cast<const uint32_t *>(cur + base + j);
            b_sum = __builtin_amdgcn_udot4(wd, 0x01020304u, b_sum + 4u * a_sum, false);
            a_sum = __builtin_amdgcn_udot4(wd, 0x01010101u, a_sum, false);
        }
        for (; j < kk; j++) { a_sum += cur[base + j]; b_sum += a_sum; }
        b_sum += a_sum * (uint32_t)max(0, len - (base + kk));
    }
    nextra = wave_sum(nextra);
    a_sum = wave_sum(a_sum);
    b_sum = wave_sum(b_sum % 65521u);
    if (lane_id() == 0) {
        atomicAdd(&S.red[1], nextra);
        atomicAdd(&S.red[2], a_sum);
        atomicAdd(&S.red[3], b_sum);
    }
    if (tid == 0) atomicAdd(&S.freq[256], 1u);
    __syncthreads();
    LZP(3)
    // ---- code lengths and codes: lit/len on wave 0, distances on wave 1; waves 2-3 clear the bit buffer ----
    if (wave_id() == 0) {
        const bool ok = assign_lengths_wave(S.freq, NLIT, S.lens, S.blcount, S.bins);
        if (lane_id() == 0) S.dbg = ok ? 0u : 1u;
        assign_codes_wave(S.blcount, S.lens, NLIT, S.code);
    } else if (wave_id() == 1) {
        const bool ok = assign_lengths_wave(S.freq + DOFF, 30, S.lens + DOFF, X.blcount_d, X.bins_d);
        if (lane_id() == 0 && !ok) S.red[7] = 1;
        assign_codes_wave(X.blcount_d, S.lens + DOFF, 30, S.code + DOFF);
    } else {
        for (uint32_t i = tid - 128; i < obuf_words; i += TN - 128) obuf[i] = 0;
    }
    __syncthreads();
    LZP(4)
    if (tid == 0) obuf[0] = z.carry;
    // ---- header (wave 0), block costs (waves 1-3) ----
    if (wave_id() == 0) {
        int hd = lane_id() < 30 && S.lens[DOFF + lane_id()] ? lane_id() + 1 : 1;
        const int hdist = __builtin_amdgcn_readlane(wave_incl_max(hd), 63);
        if (lane_id() == 0) S.icount[0] = (uint32_t)hdist;
        cl_header_wave(S, hdist);
    } else {
        uint32_t dynb = 0, fixb = 0;
        for (int s = tid - 64; s < 320; s += TN - 64) {
            const uint32_t f = S.freq[s];
            if (s < NLIT) { dynb += f * S.lens[s]; fixb += f * fixed_len(s); }
            else if (s >= DOFF && s < DOFF + 30) { dynb += f * S.lens[s]; fixb += f * 5u; }
        }
        dynb = wave_sum(dynb);
        fixb = wave_sum(fixb);
        if (lane_id() == 0) { atomicAdd(&S.red[4], dynb); atomicAdd(&S.red[5], fixb); }
    }
    __syncthreads();
    LZP(5)
    const uint32_t extra = S.red[1], hdist = S.icount[0];
    const uint32_t hdr_dyn = 17 + 3 * S.hclen + S.red[6];
    const uint32_t dyn_total = hdr_dyn + S.red[4] + extra;
    const uint32_t fix_total = 3 + S.red[5] + extra;
    const uint32_t sto_total = 3 + ((0u - (z.bitpos + 3)) & 7) + 32 + 8u * (uint32_t)len;
    {
        const uint32_t nb = (uint32_t)(((uint64_t)adB + (uint64_t)len * adA + S.red[3]) % 65521u);
        adA = (adA + S.red[2]) % 65521u;
        adB = nb;
    }
    if (sto_total <= dyn_total && sto_total <= fix_total) {
        const uint32_t bytepos = (z.bitpos + 3 + 7) >> 3;
        uint8_t *ob8 = reinterpret_cast<uint8_t *>(obuf) + (bytepos - z.flushed * 4);
        if (tid == 0) {
            put_bits(obuf, z, z.bitpos, final ? 1u : 0u, 3);
            put_bits(obuf, z, bytepos * 8, (uint32_t)len | ((~(uint32_t)len) << 16), 32);
        }
        __syncthreads();
        for (int i = tid; i < len; i += TN) ob8[4 + i] = cur[i];
        z.bitpos = (bytepos + 4 + (uint32_t)len) * 8;
        __syncthreads();
        return;
    }
    const bool use_fixed = fix_total < dyn_total || S.dbg != 0 || S.red[7] != 0;
    uint32_t pos0;
    uint32_t clv[2] = {0, 0}, clnb[2] = {0, 0};
    if (use_fixed) {
        __syncthreads();   // every lane has read the dynamic code's costs
        for (int s = tid; s < 288; s += TN) S.code[s] = fixed_code(s);
        if (tid < 30) S.code[DOFF + tid] = (__brev((uint32_t)tid) >> 27) | (5u << 16);
        if (tid == 0) put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (1u << 1), 3);
        pos0 = z.bitpos + 3;
        __syncthreads();
    } else {
        const uint32_t hclen = S.hclen;
        if (tid == 0) put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (2u << 1) | ((S.hlit - 257) << 3) | ((hdist - 1) << 8) | ((hclen - 4) << 13), 17);
        if (tid < (int)hclen) {
            const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            put_bits(obuf, z, z.bitpos + 17 + 3 * tid, S.cllens[order[tid]], 3);
        }
        const int ncl = S.ncl;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int e = 2 * tid + q;
            if (e < ncl) {
                const uint32_t ent = S.clseq[e];
                const uint32_t sym = ent & 31, cc = S.clcode[sym], cl = cc >> 16;
                const uint32_t eb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
                clv[q] = (cc & 0xFFFF) | ((ent >> 5) << cl);
                clnb[q] = cl + eb;
            }
        }
        pos0 = z.bitpos + hdr_dyn;
    }
    // ---- bit totals, one packed scan (13 bits of header entries, 19 of token bits), pack ----
    uint32_t mybits = 0;
    {
        uint64_t t = tok;
        while (t) {
            const int j = __ffsll((long long)t) - 1;
            t &= t - 1;
            const int pos = base + j;
            if ((mat >> j) & 1ull) {
                uint32_t ls, le, lv, ds, de, dv;
                lz_len_sym(X.D[pos + 1], ls, le, lv);
                lz_dist_sym(X.D[pos], ds, de, dv);
                mybits += (S.code[ls] >> 16) + le + (S.code[DOFF + ds] >> 16) + de;
            } else mybits += S.code[cur[pos]] >> 16;
        }
    }
    LZP(6)
    uint32_t packed_total;
    const uint32_t packed = block_excl_add_w<TN / 64>((clnb[0] + clnb[1]) | (mybits << 13), S.ws, packed_total);
    const uint32_t total_bits = packed_total >> 13;
    if (!use_fixed) {
        const uint32_t p = z.bitpos + 17 + 3 * S.hclen + (packed & 0x1FFF);
        if (clnb[0]) put_bits(obuf, z, p, clv[0], clnb[0]);
        if (clnb[1]) put_bits(obuf, z, p + clnb[0], clv[1], clnb[1]);
    }
    {
        uint32_t pos_b = pos0 + (packed >> 13);
        auto or_bits = [&](uint64_t v, uint32_t nb) {          // nb <= 48
            const uint32_t wd = (pos_b >> 5) - z.flushed, sh = pos_b & 31;
            const uint64_t lo = v << sh;
            atomicOr(&obuf[wd], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&obuf[wd + 1], (uint32_t)(lo >> 32));
            if (sh + nb > 64) atomicOr(&obuf[wd + 2], (uint32_t)(v >> (64 - sh)));
            pos_b += nb;
        };
        uint64_t t = tok;
        while (t) {
            const int j = __ffsll((long long)t) - 1;
            t &= t - 1;
            const int pos = base + j;
            if ((mat >> j) & 1ull) {
                uint32_t ls, le, lv, ds, de, dv;
                lz_len_sym(X.D[pos + 1], ls, le, lv);
                lz_dist_sym(X.D[pos], ds, de, dv);
                const uint32_t lc = S.code[ls], dc = S.code[DOFF + ds];
                uint32_t nb = lc >> 16;
                uint64_t v = (uint64_t)(lc & 0xFFFFu) | ((uint64_t)lv << nb);
                nb += le;
                v |= (uint64_t)(dc & 0xFFFFu) << nb;
                nb += dc >> 16;
                v |= (uint64_t)dv << nb;
                nb += de;
                or_bits(v, nb);
            } else {
                const uint32_t cc = S.code[cur[pos]];
                or_bits((uint64_t)(cc & 0xFFFFu), cc >> 16);
            }
        }
    }
    const uint32_t eob = S.code[256];
    if (tid == 0) put_bits(obuf, z, pos0 + total_bits, eob & 0xFFFF, eob >> 16);
    z.bitpos = pos0 + total_bits + (eob >> 16);
    __syncthreads();
    LZP(7)
#ifdef S5_LZPROBE
    if (tid == 0) atomicAdd(&g_lzprobe[15], 1ull);
#endif
}

}  // namespace s5
