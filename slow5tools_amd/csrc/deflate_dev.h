// deflate_dev.h — per-workgroup DEFLATE (RFC 1951) block encoder for gfx950, zlib (RFC 1950) framing.
//
// Replaces the record-press stage of slow5lib that slow5tools reaches through slow5_rec_to_mem
// (/root/reference/src/view.c:49; zlib via -lz, /root/reference/Makefile:10).  Output is NOT
// byte-identical to zlib's (no encoder is required to be); the contract is that stock zlib
// inflates it to the identical payload (tests/test_gpu_parity.py).
//
// Design (SURVEY.md §0 finding 6): on svb-zd payloads LZ77 buys < 1 %, run-length + dynamic Huffman
// does the work.  Per block (<= 16 KiB of payload held in LDS), one workgroup of 256 lanes:
//   A  tokenise: lane t owns K = ceil(len/256) contiguous bytes and a break mask; the neighbouring breaks come from one
//      ballot per wave; in-run positions are classified with mask arithmetic, only long runs reach per-segment code;
//      tokens = literal | (length 3..258, distance 1); LDS-atomic histogram of the 286 lit/len symbols; Adler-32 partials
//   B  code construction: counting sort of the symbol frequencies (all lanes), round-based Huffman merge in one wave
//      (DPP / permlane butterflies), leaf depths by parallel parent walks, length limiting, canonical codes via wave
//      ballots; the 19-symbol code-length code is picked from two static prefix codes by cost
//   C  cost compare dynamic / fixed / stored; per-lane bit totals -> workgroup prefix scan of bit offsets -> each lane ORs
//      its tokens into the LDS bit buffer, four byte positions (<= 60 bits) at a time
// then coalesced word copy LDS -> HBM.  No MFMA: there is no contraction anywhere in this path.  DESIGN.md §4.1 has the
// measurements behind each of these choices.
#pragma once
#include "dev_common.h"

namespace s5 {

constexpr int DEFL_BLK = NT * 64;    // max payload bytes per DEFLATE block (64 per lane: one break mask)
constexpr int NLIT = 286;            // literal/length symbols in use
constexpr int DOFF = 288;            // distance codes live at [DOFF, DOFF+32) in the shared tables

// Scratch of one Huffman construction.  The 286-symbol instance (3.4 KiB) is only live between the
// histogram and the code lengths, so the fused kernel overlays it on the (not yet used) bit buffer.
template <int CAP>
struct BuildScratchT {
    alignas(16) uint32_t lf[CAP];   // leaf weights, ascending
    alignas(16) uint32_t nf[CAP];   // sort keys, then internal-node weights (creation order = ascending)
    uint16_t npar[CAP];             // internal node -> parent
    uint16_t rsym[CAP];             // rank -> symbol
};
// Scratch of the counting sort of the 286 lit/len frequencies (lives right behind the build scratch).
struct SortScratch {
    uint32_t bm[64 * 9];     // per frequency bucket: 288-bit membership bitmap of its symbols
    uint32_t cnt[64];        // symbols per bucket (bucket = min(freq, 63))
    uint32_t ovl[288];       // keys of the symbols in the overflow bucket (freq >= 63), in symbol order
};
struct BuildScratch : BuildScratchT<288> { SortScratch sort; };

struct DeflShared {
    alignas(16) uint32_t freq[320];   // histogram: [0,288) lit/len, [288,320) dist
    alignas(16) uint32_t clfreq[20];
    uint32_t code[320];      // bit-reversed code | nbits << 16
    uint16_t clseq[320];     // code-length sequence: sym | extra_value << 5
    uint8_t lens[320];       // code lengths, same indexing as freq
    uint32_t clcode[20];
    uint8_t cllens[20];
    uint32_t blcount[16];
    uint32_t icount[16];     // internal nodes per depth
    uint32_t ws[16];         // cross-wave scan scratch
    uint32_t bins[64];       // assign_lengths_wave: Kraft cost per priority bin
    uint32_t red[8];         // 0 matches, 1 extra bits, 2 adler A part, 3 adler B part, 4 dyn bits, 5 fixed bits, 6 cl bits
    uint32_t ncl, hlit, hclen, dbg;
    uint32_t wtot[32];       // deflate_block2: per wave [0, 8) dynamic body bits, [8, 16) fixed, [16, 24) extra bits, [24, 32) matches
    uint32_t lalloc, lpad[3];   // deflate_block2: tokens handed out of the token list (which lives in freq[])
    uint32_t wad[16];        // deflate_block2: Adler-32 partial sums per wave ([0, 8) sum of bytes, [8, 16) weighted sum mod 65521)
};

// Ordered single-pass output: as soon as a single-block record's final size is known (right after the bit-offset
// scan, before the tokens are packed) it is published for the look-back of the following reads.
struct EarlySize {
    unsigned long long *state;   // nullptr: nothing to publish
    uint32_t r;
};
__device__ __forceinline__ void publish_size(const EarlySize &es, uint32_t end_bitpos) {
    if (es.state && threadIdx.x == 0) {
        const unsigned long long total = ((end_bitpos + 7) >> 3) + 4;   // + Adler-32
        __hip_atomic_store(&es.state[es.r], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct ZOut {                // replicated uniformly in every lane's registers
    uint32_t bitpos;         // absolute bit position in the record slot (slot byte 0 = bit 0)
    uint32_t flushed;        // 32-bit words already copied to HBM; obuf[0] holds word `flushed`
    uint32_t carry = 0;      // MODE 2 only: the partial word obuf[0] must hold when a block starts writing (see deflate_block)
};

__device__ __forceinline__ void put_bits(uint32_t *obuf, const ZOut &z, uint32_t pos, uint32_t v, uint32_t nb) {
    uint32_t w = (pos >> 5) - z.flushed, sh = pos & 31;
    atomicOr(&obuf[w], v << sh);
    if (sh + nb > 32) atomicOr(&obuf[w + 1], v >> (32 - sh));
}

// ---- length-limited Huffman code lengths for freq[0..n), 256 < n <= 288 (the lit/len alphabet), all NT lanes ----
template <int CAP>
__device__ __forceinline__ void build_lengths(DeflShared &S, BuildScratchT<CAP> &B, SortScratch *Q, const uint32_t *freq, int n,
                                              int maxbits, uint8_t *lens, uint32_t *blcount, uint32_t *icount) {
    static_assert(CAP > NT, "the counting sort below is laid out for the 286-symbol alphabet");
    const int tid = (int)threadIdx.x;
    constexpr int NTH = NT;
    auto sync = [&]() { __syncthreads(); };
    for (int s = tid; s < n; s += NTH) lens[s] = 0;
    if (tid < 16) { blcount[tid] = 0; icount[tid] = 0; }
    // Sort the symbols by (frequency, symbol).
    const uint32_t f0 = tid < n ? freq[tid] : 0;
    const uint32_t f1 = tid + NT < n ? freq[tid + NT] : 0;
    const uint32_t key0 = f0 ? (f0 << 9) | (uint32_t)tid : 0xFFFFFFFFu;
    const uint32_t key1 = f1 ? (f1 << 9) | (uint32_t)(tid + NT) : 0xFFFFFFFFu;
    int r0 = 0, r1 = 0, m = 0;
    {
        // Counting sort, deterministic.  Frequencies are small: bucket = min(freq, 63).  Each bucket keeps a
        // 288-bit bitmap of its symbols, so a symbol's rank inside its bucket is a popcount of the bits below
        // it (symbol order), and the bucket's first rank is a 64-entry prefix sum.  Only the overflow bucket
        // (freq >= 63: a handful of symbols) needs real comparisons, among its own members.
        for (int i = tid; i < 64 * 9 + 64; i += NT) Q->bm[i] = 0;   // bm and cnt are contiguous
        sync();
        const uint32_t b0 = min(f0, 63u), b1 = min(f1, 63u);
        if (f0) { atomicAdd(&Q->cnt[b0], 1u); atomicOr(&Q->bm[b0 * 9 + (tid >> 5)], 1u << (tid & 31)); }
        if (f1) { atomicAdd(&Q->cnt[b1], 1u); atomicOr(&Q->bm[b1 * 9 + 8], 1u << (tid & 31)); }   // symbols 256.. sit in word 8
        sync();
        if (wave_id() == 0) {   // first rank of every bucket (B.nf is free until the merge)
            const uint32_t c = Q->cnt[lane_id()];
            const uint32_t incl = wave_incl_add(c);
            B.nf[lane_id()] = incl - c;
            if (lane_id() == 63) B.nf[64] = incl;
        }
        uint32_t w0 = 0, w1 = 0;
        if (f0) {
            const int w = tid >> 5;
            w0 = __popc(Q->bm[b0 * 9 + w] & ((1u << (tid & 31)) - 1));
            for (int q = 0; q < w; q++) w0 += __popc(Q->bm[b0 * 9 + q]);
            if (b0 == 63) Q->ovl[w0] = key0;
        }
        if (f1) {
            w1 = __popc(Q->bm[b1 * 9 + 8] & ((1u << (tid & 31)) - 1));
            for (int q = 0; q < 8; q++) w1 += __popc(Q->bm[b1 * 9 + q]);
            if (b1 == 63) Q->ovl[w1] = key1;
        }
        sync();
        const uint32_t novf = Q->cnt[63];
        if (f0 && b0 == 63) { w0 = 0; for (uint32_t q = 0; q < novf; q++) w0 += Q->ovl[q] < key0; }
        if (f1 && b1 == 63) { w1 = 0; for (uint32_t q = 0; q < novf; q++) w1 += Q->ovl[q] < key1; }
        r0 = (int)(B.nf[b0] + w0);
        r1 = (int)(B.nf[b1] + w1);
        m = (int)B.nf[64];
    }
    sync();
    if (f0) { B.lf[r0] = f0; B.rsym[r0] = (uint16_t)tid; }
    if (f1) { B.lf[r1] = f1; B.rsym[r1] = (uint16_t)(tid + NT); }
    sync();
    if (S.dbg == 31) return;   // tools/stage_time.py cut-off: after the sort
    if (m <= 1) {   // degenerate: keep the code complete with two 1-bit codes
        if (tid == 0) {
            int sym = m ? B.rsym[0] : 0;
            lens[sym] = 1;
            lens[sym == 0 ? 1 : 0] = 1;
            blcount[1] = 2;
        }
        sync();
        return;
    }
    if (wave_id() == 0) {
      // direction of every butterfly step as a per-lane constant: the upper lane of a pair keeps the max (med3 with ~0), the lower the min
      const uint32_t dir32 = (lane_id() & 32) ? ~0u : 0u, dir16 = (lane_id() & 16) ? ~0u : 0u, dir8 = (lane_id() & 8) ? ~0u : 0u;
      const uint32_t dir4 = (lane_id() & 4) ? ~0u : 0u, dir2 = (lane_id() & 2) ? ~0u : 0u, dir1 = (lane_id() & 1) ? ~0u : 0u;
      // Huffman tree by ROUNDS instead of one merge per step (the serial two-queue loop costs ~400 cycles per merge on a
      // GPU).  Leaves ascending in B.lf, internal nodes are produced ascending into B.nf.  Per round, one wave64 takes the
      // next leaves and the next nodes, bitonic-merges them ([leaves asc | nodes desc] is a bitonic sequence), and pairs up
      // EVERY item not larger than T = X0 + X1 at once — no node created in this round can be smaller than T, so the pairs
      // are exactly the ones the serial algorithm would form.  The smallest remaining weight at least doubles per round:
      // ~12 rounds for a 210-symbol alphabet (m-1 rounds only for Fibonacci-like weights).  While more than 32 leaves or 32
      // nodes are pending the window is 64 + 64 keys in two registers; after that (about half of the rounds, and
      // always when fewer than 33 symbols are in use) 32 + 32 keys in one register.
      const int lane = lane_id();
      const uint32_t INF = 0xFFFFFFFFu;
      int i = 0, j = 0, k = 0;
      while (k < m - 1) {
        if (m - i <= 32 && k - j <= 32) {
            // leaves ascending in lanes 0..31, nodes descending in lanes 32..63; key = weight << 8 | is_node << 6 | window index
            const int qn = 63 - lane;
            const uint32_t ov = lane < 32 ? (i + lane < m ? B.lf[i + lane] : INF) : (j + qn < k ? B.nf[j + qn] : INF);
            uint32_t x = ov == INF ? INF : (ov << 8) | (lane < 32 ? (uint32_t)lane : 64u | (uint32_t)qn);
            x = umed3(x, wave_xor<32>(x), dir32); x = umed3(x, wave_xor<16>(x), dir16); x = umed3(x, wave_xor<8>(x), dir8);
            x = umed3(x, wave_xor<4>(x), dir4); x = umed3(x, wave_xor<2>(x), dir2); x = umed3(x, wave_xor<1>(x), dir1);
            const uint32_t x0 = __builtin_amdgcn_readlane(x, 0), x1 = __builtin_amdgcn_readlane(x, 1);
            const uint32_t T = (x0 >> 8) + (x1 >> 8);          // everything pending is inside the window
            const uint32_t vx = x >> 8;
            int c = __popcll(__ballot(x != INF && vx <= T));
            c &= ~1;
            c = max(c, 2);
            c = min(c, 2 * (m - 1 - k));
            const uint32_t sx = vx + wave_xor<1>(vx);
            if (!(lane & 1) && lane < c) B.nf[k + (lane >> 1)] = sx;
            const bool inx = lane < c;
            if (inx && (x & 64u)) B.npar[j + (x & 63u)] = (uint16_t)(k + (lane >> 1));
            const int nn = __popcll(__ballot(inx && (x & 64u)));
            i += c - nn;
            j += nn;
            k += c >> 1;
        } else {
            // key = weight << 8 | is_node << 6 | window index; weights < 2^24
            const uint32_t lv = i + lane < m ? B.lf[i + lane] : INF;
            const int qn = 63 - lane;   // node window is loaded descending
            const uint32_t nv = j + qn < k ? B.nf[j + qn] : INF;
            uint32_t a = lv == INF ? INF : (lv << 8) | (uint32_t)lane;
            uint32_t b = nv == INF ? INF : (nv << 8) | 64u | (uint32_t)qn;
            {   // bitonic merge of 128 keys: stride 64 across the two registers, then 32..1 inside each
                const uint32_t lo = min(a, b), hi = max(a, b);
                a = lo; b = hi;
                a = umed3(a, wave_xor<32>(a), dir32); b = umed3(b, wave_xor<32>(b), dir32);
                a = umed3(a, wave_xor<16>(a), dir16); b = umed3(b, wave_xor<16>(b), dir16);
                a = umed3(a, wave_xor<8>(a), dir8); b = umed3(b, wave_xor<8>(b), dir8);
                a = umed3(a, wave_xor<4>(a), dir4); b = umed3(b, wave_xor<4>(b), dir4);
                a = umed3(a, wave_xor<2>(a), dir2); b = umed3(b, wave_xor<2>(b), dir2);
                a = umed3(a, wave_xor<1>(a), dir1); b = umed3(b, wave_xor<1>(b), dir1);
            }
            // a[lane] = X[lane], b[lane] = X[64 + lane] in ascending order
            const uint32_t x0 = __builtin_amdgcn_readlane(a, 0), x1 = __builtin_amdgcn_readlane(a, 1);
            uint32_t T = (x0 >> 8) + (x1 >> 8);
            if (i + 64 < m) T = min(T, __builtin_amdgcn_readlane(lv, 63));      // leaves beyond the window
            if (j + 64 < k) T = min(T, __builtin_amdgcn_readlane(nv, 0));       // nodes beyond the window
            const uint32_t va = a >> 8, vb = b >> 8;
            int c = __popcll(__ballot(a != INF && va <= T)) + __popcll(__ballot(b != INF && vb <= T));
            c &= ~1;
            c = max(c, 2);
            c = min(c, 2 * (m - 1 - k));
            // pairs (X[2p], X[2p+1]) -> node k + p
            const uint32_t sa = va + wave_xor<1>(va), sb = vb + wave_xor<1>(vb);
            if (!(lane & 1) && lane < c) B.nf[k + (lane >> 1)] = sa;
            if (!(lane & 1) && 64 + lane < c) B.nf[k + 32 + (lane >> 1)] = sb;
            const bool ina = lane < c, inb = 64 + lane < c;
            if (ina && (a & 64u)) B.npar[j + (a & 63u)] = (uint16_t)(k + (lane >> 1));
            if (inb && (b & 64u)) B.npar[j + (b & 63u)] = (uint16_t)(k + 32 + (lane >> 1));
            const int nn = __popcll(__ballot(ina && (a & 64u))) + __popcll(__ballot(inb && (b & 64u)));
            i += c - nn;
            j += nn;
            k += c >> 1;
        }
      }
    }
    sync();
    if (S.dbg == 32) return;   // cut-off: after the merge
    // Depth of every internal node by a parallel parent walk (root = node m-2, depth 0).  Leaves at
    // depth L = 2 * I[L-1] - I[L]; sorted order makes depth monotone in rank, so counts are enough.
    for (int q = tid; q < m - 1; q += NTH) {
        int d = 0, p = q;
        while (p != m - 2) { p = B.npar[p]; d++; }
        atomicAdd(&icount[min(d, maxbits)], 1u);
    }
    sync();
    if (wave_id() == 0) {
        // leaves per depth from the internal-node counts, one depth per lane; the clamp bucket takes the rest
        const int L = lane_id();
        uint32_t c = 0;
        if (L >= 1 && L < maxbits) c = 2 * icount[L - 1] - icount[L];
        const uint32_t used = wave_sum(c);
        if (L >= 1 && L < maxbits) blcount[L] = c;
        if (L == 0) blcount[maxbits] = (uint32_t)m - used;   // every leaf at depth >= maxbits, clamped
        wave_sync();
        if (L == 0 && icount[maxbits]) {   // some leaf was deeper than maxbits: repair the Kraft sum on the counts
            uint32_t total = 0;
            for (int b = maxbits; b >= 1; b--) total += blcount[b] << (maxbits - b);
            while (total != (1u << maxbits)) {
                blcount[maxbits]--;
                for (int b = maxbits - 1; b > 0; b--)
                    if (blcount[b]) { blcount[b]--; blcount[b + 1] += 2; break; }
                total--;
            }
        }
    }
    sync();
    for (int r = tid; r < m; r += NTH) {   // rarest symbols get the longest codes
        uint32_t cum = 0;
        int L = maxbits;
        for (; L > 1; L--) { cum += blcount[L]; if ((uint32_t)r < cum) break; }
        lens[B.rsym[r]] = (uint8_t)L;
    }
    sync();
}


// ---- code lengths without a tree: Shannon lengths + greedy hand-out of the Kraft slack, ONE wave, no sort ----
// Replaces sort + Huffman merge + depth walk for the DEFLATE lit/len alphabet (1.8 k of the 8.9 k VALU instructions per read,
// 4.4 of 16 ms).  l_s = ceil(log2(N / f_s)) (exact integer arithmetic) never violates the Kraft inequality and leaves a slack
// R = 2^15 - sum 2^(15 - l_s) < 2^14.  Shortening symbol s by one bit costs 2^(15 - l_s) of slack and saves f_s bits; the
// benefit per unit of slack, f_s * 2^l_s, lies in [N, 2N), so the order is fine-grained but the stakes are even: handing the
// slack out greedily in that order — whole priority bins from the top while they fit, then the next bin partially in symbol
// order, and again with whatever still fits until nothing is left — lands within 0.03 % of the optimal (Huffman) cost on
// every fixture record and 0.003 % on the bench reads (tools/len_assign_probe.py; a symbol may be shortened more than once).
// The code comes out COMPLETE (slack exactly 0), which zlib's inflate demands of a lit/len code: the longest code's unit
// always divides the slack, so a pass always finds a taker.  Lane t owns symbols 5t .. 5t+4 in registers; a pass is one LDS
// histogram over 64 bins (ratio 0.5 .. 2 in steps of 1/32) and two wave scans.  n <= 320, N <= 2^15 (a block holds <= 16 KiB).
// Returns false (uniform) if the pass limit was hit — never seen; the caller then sends the block with the fixed code.
// MAXL: longest code allowed (15 DEFLATE, 11 zstd literals).  A Shannon length above MAXL is clamped; if the clamped lengths
// over-subscribe the code space the function returns false as well (the caller then uses the exact construction).
// NH: the frequencies are the sum of NH histograms that lie 288 words apart (deflate2_dev.h keeps one per wave); EOB1: symbol 256 (end of
// block) is not in the histograms and counts once.
// ABSORB (round 5; 0 = off): after that many passes whatever slack is left goes to UNUSED symbols — one per set bit of the slack, the lowest
// unused symbols first (holes among the literals in use, so the code-length header's zero runs are not cut) — instead of more passes over the
// used ones.  A code for a symbol that never occurs costs only its header entry (~8 bits); the body loses the bits the slack would still have
// bought: measured on every 16 KiB block of the fixture records and on the bench reads (tools/len_assign_probe.py's model), 2 passes + absorb
// = +0.02 .. +0.10 % bits (+0.06 % on the bench reads), 3 passes +0.01 .. +0.05 %.  The passes are ~100 dependent instructions each on ONE
// wave while the workgroup's other waves wait: the loop runs 4-6 of them.  Falls back to the loop when too few symbols are unused.
template <int MAXL = 15, int NH = 1, bool EOB1 = false, int ABSORB = 0>
__device__ __forceinline__ bool assign_lengths_wave(const uint32_t *freq, int n, uint8_t *lens, uint32_t *blcount, uint32_t *bins) {
    const int lane = lane_id();
    constexpr uint32_t HUGE_C = 0x80000000u;   // "cannot be shortened": no symbol / already 1 bit
    uint32_t f[5], c[5], b[5];
    int l[5];
    uint32_t fsum = 0, used = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int s = 5 * lane + q;
        f[q] = s < n ? freq[s] : 0u;
#pragma unroll
        for (int h = 1; h < NH; h++) f[q] += s < n ? freq[288 * h + s] : 0u;
        if (EOB1 && s == 256) f[q] += 1u;
        fsum += f[q];
        used += f[q] != 0u;
    }
    if (lane < 16) blcount[lane] = 0;
    const uint32_t N = wave_sum(fsum), m = wave_sum(used);
    if (m <= 1) {   // degenerate: keep the code complete with two 1-bit codes
        const int other = __builtin_amdgcn_readfirstlane((int)f[0]) ? 1 : 0;   // (lane 0 holds symbol 0) the second 1-bit code goes to a symbol that is not in use
#pragma unroll
        for (int q = 0; q < 5; q++) { const int s = 5 * lane + q; if (s < n) lens[s] = (f[q] || s == other || (m == 0 && s == 1)) ? 1 : 0; }
        if (lane == 0) blcount[1] = 2;
        wave_sync();
        return true;
    }
    const int a = 32 - __clz((int)(N - 1));   // N - 1 has a bits: 2^(a-1) < N <= 2^a
    const float inv = 32.0f / (float)N;
    uint32_t csum = 0;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        if (f[q]) {
            const int l0 = a - (32 - __clz((int)f[q]));           // f << l0 has a bits
            const int ls = l0 + (((f[q] << l0) < N) ? 1 : 0);      // smallest l with f << l >= N; >= 1 because f < N
            const int lq = min(ls, MAXL);
            l[q] = lq;
            csum += 1u << (MAXL - lq);
            c[q] = lq > 1 ? 1u << (MAXL - lq) : HUGE_C;
            b[q] = min(63u, (uint32_t)((float)(f[q] << ls) * inv) >> (ls - lq));   // floor(32 * ratio), ratio in [1, 2); a clamped symbol ranks lower
        } else { l[q] = 0; c[q] = HUGE_C; b[q] = 0; }
    }
    const uint32_t used_c = wave_sum(csum);
    if (used_c > (1u << MAXL)) return false;                         // (only possible with MAXL < 15: too many clamped symbols)
    uint32_t R = (1u << MAXL) - used_c;
    int guard = 0;
    while (R) {   // uniform
        if (ABSORB > 0 && guard == ABSORB) {
            uint32_t nun = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) nun += (5 * lane + q < n && f[q] == 0u) ? 1u : 0u;
            const uint32_t incl = wave_incl_add(nun);
            if ((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) >= (uint32_t)__popc(R)) {   // (uniform)
                uint32_t r = R;                                    // my first unused symbol takes the (incl - nun)-th set bit of R from the bottom
                const uint32_t skip = incl - nun;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)MAXL; i++) if (i < skip) r &= r - 1u;
#pragma unroll
                for (int q = 0; q < 5; q++)
                    if (5 * lane + q < n && f[q] == 0u && r) {
                        l[q] = MAXL - (__ffs((int)r) - 1);         // slack unit 2^j = a code of MAXL - j bits
                        r &= r - 1u;
                    }
                R = 0;
                break;
            }
        }
        if (++guard > 40) return false;
        bins[lane] = 0;
        wave_sync();
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (c[q] <= R) atomicAdd(&bins[b[q]], c[q]);
        wave_sync();
        const uint32_t hr = bins[63 - lane];                          // lane j looks at bin 63 - j: the scan runs from the top bin down
        const uint32_t incl = wave_incl_add(hr);
        const uint64_t fit = __ballot(incl <= R);                      // a prefix of the lanes: incl never decreases
        const int nfit = ~fit ? __ffsll((long long)~fit) - 1 : 64;
        const uint32_t whole = nfit ? (uint32_t)__builtin_amdgcn_readlane((int)incl, nfit - 1) : 0u;
        const uint64_t below = nfit < 64 ? (__ballot(hr != 0u) >> nfit) << nfit : 0ull;
        const uint32_t thr = 64u - (uint32_t)nfit;                     // bins >= thr go whole ...
        const uint32_t pb = below ? 63u - (uint32_t)(__ffsll((long long)below) - 1) : 0xFFFFu;   // ... the next bin in use partially
        const uint32_t Rold = R;
        R -= whole;
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) mine += (c[q] <= Rold && b[q] == pb) ? c[q] : 0u;
        uint32_t run = wave_incl_add(mine) - mine, got = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            bool sel = c[q] <= Rold && b[q] >= thr && b[q] != pb;
            if (c[q] <= Rold && b[q] == pb) { run += c[q]; if (run <= R) { sel = true; got += c[q]; } }
            if (sel) {
                l[q] -= 1;
                c[q] = l[q] > 1 ? c[q] << 1 : HUGE_C;
                b[q] >>= 1;                                            // the ratio halves
            }
        }
        R -= wave_sum(got);
    }
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int s = 5 * lane + q;
        if (s < n) {
            lens[s] = (uint8_t)l[q];
            if (l[q]) atomicAdd(&blcount[l[q]], 1u);
        }
    }
    wave_sync();
    return true;
}

// ---- canonical codes from lengths (wave 0; S.blcount must match lens) ----
__device__ __forceinline__ void assign_codes_wave(const uint32_t *blcount, const uint8_t *lens, int n, uint32_t *code_out) {
    {
        uint32_t next[16];
        uint32_t c = 0;
        next[0] = 0;
#pragma unroll
        for (int b = 1; b < 16; b++) {   // first code of every length, wave-uniform: keep the table in SGPRs
            c = (c + (b > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)blcount[b - 1]) : 0u)) << 1;
            next[b] = c;
        }
        for (int base = 0; base < n; base += 64) {
            const int s = base + lane_id();
            const int l = s < n ? lens[s] : 0;
            uint32_t mine = 0;
#pragma unroll
            for (int b = 1; b < 16; b++) {
                const uint64_t mask = __ballot(l == b);
                if (mask == 0) continue;                                  // uniform: no symbol of this length in the slice
                // rank among the lanes with the same length = set bits of the ballot below my lane (v_mbcnt)
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (l == b) mine = next[b] + below;
                next[b] += __popcll(mask);
            }
            if (s < n) code_out[s] = l ? ((__brev(mine) >> (32 - l)) | ((uint32_t)l << 16)) : 0u;
        }
    }
    wave_sync();
}

__device__ __forceinline__ int fixed_len(int s) { return s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8; }
__device__ __forceinline__ uint32_t fixed_code(int s) {
    uint32_t c, l;
    if (s < 144) { c = 0x30 + s; l = 8; }
    else if (s < 256) { c = 0x190 + (s - 144); l = 9; }
    else if (s < 280) { c = s - 256; l = 7; }
    else { c = 0xC0 + (s - 280); l = 8; }
    return (__brev(c) >> (32 - l)) | (l << 16);
}

// ---- the code-length header of a dynamic block, one wave: HLIT, run-length coding of the hlit + hdist code lengths
// (S.lens[0 .. hlit) then S.lens[DOFF .. DOFF + hdist)), choice of the code-length code, header cost in S.red[6] ----
__device__ __forceinline__ void cl_header_wave(DeflShared &S, int hdist) {
    const int lane = lane_id();
    if (lane < 20) S.clfreq[lane] = 0;   // (this wave is the counters' only user: cleared here, ordered by the wave-scope syncs below)
    int hl = lane < 29 && S.lens[257 + lane] ? 258 + lane : 257;
    const int hlit = __builtin_amdgcn_readlane(wave_incl_max(hl), 63), n = hlit + hdist;
    if (lane == 0) S.hlit = (uint32_t)hlit;
    wave_sync();
    {
        // Run-length code the hlit + hdist code lengths (RFC 1951 3.2.7, symbols 16/17/18): lane t owns
        // positions 5t..5t+4; run bounds from a wave prefix-max and suffix-min; every position decides
        // alone whether it emits an entry; a wave prefix sum compacts the entries.
        constexpr int PP = 5;
        const int p0 = PP * lane;
        int v[PP];
        int prevv = -1;
        if (p0 - 1 >= 0 && p0 - 1 < n) prevv = p0 - 1 < hlit ? S.lens[p0 - 1] : S.lens[DOFF + p0 - 1 - hlit];
        uint32_t bk = 0;
#pragma unroll
        for (int q = 0; q < PP; q++) {
            const int p = p0 + q;
            v[q] = p < n ? (p < hlit ? S.lens[p] : S.lens[DOFF + p - hlit]) : -2 - q;   // past the end: never equal
            if (p < n && v[q] != (q == 0 ? prevv : v[q - 1])) bk |= 1u << q;
        }
        const int local_last = bk ? p0 + 31 - __clz(bk) : -1;
        const int local_first = bk ? p0 + __ffs(bk) - 1 : n;
        int lastb = wave_incl_max(local_last);
        lastb = __shfl_up(lastb, 1);
        if (lane == 0) lastb = -1;
        int nextb = wave_suffix_incl_min(local_first);
        nextb = __shfl_down(nextb, 1);
        if (lane == 63) nextb = n;
        uint32_t ent[PP], nent = 0;
#pragma unroll
        for (int q = 0; q < PP; q++) {
            const int p = p0 + q;
            ent[q] = 0xFFFFFFFFu;
            if (p >= n) continue;
            const uint32_t lo = bk & ((2u << q) - 1), hi = bk >> (q + 1);
            const int s = lo ? p0 + 31 - __clz(lo) : lastb;
            const int e = hi ? p + __ffs(hi) : nextb;
            const int R = e - s, rel = p - s;
            if (v[q] == 0) {
                const int c = rel / 138, off = rel - c * 138, Lc = min(138, R - c * 138);
                if (Lc >= 11) { if (off == 0) ent[q] = 18u | ((uint32_t)(Lc - 11) << 5); }
                else if (Lc >= 3) { if (off == 0) ent[q] = 17u | ((uint32_t)(Lc - 3) << 5); }
                else ent[q] = 0;
            } else if (rel == 0) {
                ent[q] = (uint32_t)v[q];
            } else {
                const int mm = rel - 1, c = mm / 6, off = mm - c * 6, Lc = min(6, R - 1 - c * 6);
                if (Lc >= 3) { if (off == 0) ent[q] = 16u | ((uint32_t)(Lc - 3) << 5); }
                else ent[q] = (uint32_t)v[q];
            }
            nent += ent[q] != 0xFFFFFFFFu;
        }
        const uint32_t incl = wave_incl_add(nent);
        uint32_t at = incl - nent;
#pragma unroll
        for (int q = 0; q < PP; q++)
            if (ent[q] != 0xFFFFFFFFu) {
                S.clseq[at++] = (uint16_t)ent[q];
                atomicAdd(&S.clfreq[ent[q] & 31], 1u);
            }
        if (lane == 63) S.ncl = incl;
    }
    wave_sync();
    {
        // The 19-symbol code-length code is not built per read: it is picked from two static prefix codes by cost.
        // Its lengths travel in the block header (HCLEN x 3 bits), so any complete code is valid DEFLATE.  Code A is the
        // 7-bit-limited optimum (package-merge) for the aggregate code-length statistics of svb-zd signal payloads
        // (tools/clfreq_dump.py: 0.08 % larger records than a per-read optimum); code B covers every symbol for payloads
        // that use code lengths 14 / 15.  Saves the serial Huffman construction on the critical wave.
        static constexpr uint32_t CLA[19] = {0x30001, 0x7002f, 0x7006f, 0x7001f, 0x7005f, 0x50003, 0x50013, 0x5000b, 0x5001b, 0x50007,
                                             0x30005, 0x20000, 0x20002, 0x7003f, 0x00000, 0x00000, 0x50017, 0x6000f, 0x7007f};
        static constexpr uint32_t CLB[19] = {0x30002, 0x60017, 0x7002f, 0x7006f, 0x60037, 0x5000b, 0x40005, 0x5001b, 0x50007, 0x4000d,
                                             0x30006, 0x30001, 0x20000, 0x7001f, 0x7005f, 0x7003f, 0x40003, 0x6000f, 0x7007f};
        const uint32_t ca = lane < 19 ? CLA[lane] : 0u, cb = lane < 19 ? CLB[lane] : 0u;
        const uint32_t f = lane < 19 ? S.clfreq[lane] : 0u;
        const uint32_t eb = lane == 16 ? 2u : lane == 17 ? 3u : lane == 18 ? 7u : 0u;
        const uint32_t costA = wave_sum(f * ((ca >> 16) + eb) + ((f && !(ca >> 16)) ? (1u << 24) : 0u));   // A lacks symbols 14, 15
        const uint32_t costB = wave_sum(f * ((cb >> 16) + eb));
        const bool useA = costA <= costB;
        if (lane < 19) { const uint32_t c = useA ? ca : cb; S.clcode[lane] = c; S.cllens[lane] = (uint8_t)(c >> 16); }
        if (lane == 0) { S.red[6] = useA ? costA : costB; S.hclen = useA ? 18u : 19u; }   // A: trailing zero length of symbol 15 is not sent
    }
}

// Copy completed words obuf -> HBM slot and slide the partial word to obuf[0].
// final_all: copy everything including the last partial word, no slide.
template <int TN = NT>   // threads of the calling workgroup
__device__ __forceinline__ void flush_words(uint32_t *obuf, uint32_t *out32, ZOut &z, bool final_all, uint32_t n_words = 0xFFFFFFFFu) {
    const int tid = threadIdx.x;
    const uint32_t full = final_all ? (z.bitpos + 31) >> 5 : z.bitpos >> 5;
    const uint32_t n = n_words != 0xFFFFFFFFu ? n_words : full - z.flushed;   // (n_words: copy exactly that many words, nothing else)
    {   // four words per lane: an aligned 16-byte LDS read, a 16-byte store wherever the stream stands (z.flushed is any word count)
        typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t u4a __attribute__((ext_vector_type(4)));
        const u4a *s16 = reinterpret_cast<const u4a *>(obuf);
        const uint32_t nv = n >> 2;
        for (uint32_t j = tid; j < nv; j += TN) {
            const uint32_t w = z.flushed + 4u * j;
            const u4a v = s16[j];
            if (w >= 2) *reinterpret_cast<u4u *>(out32 + w) = v;
            else { out32[w + 2] = v.z; out32[w + 3] = v.w; }   // (w = 0) words 0,1 = u64 size prefix, written last by lane 0
        }
        for (uint32_t i = 4u * nv + tid; i < n; i += TN) {
            const uint32_t w = z.flushed + i;
            if (w >= 2) out32[w] = obuf[i];
        }
    }
    if (final_all) return;
    __syncthreads();
    const uint32_t partial = obuf[n];
    __syncthreads();
    for (uint32_t i = tid; i <= n; i += TN) obuf[i] = i == 0 ? partial : 0u;
    __syncthreads();
    z.flushed = full;
}

}  // namespace s5
#include "deflate2_dev.h"   // the slab form of the block encoder (deflate_block2; round 5 — the position-per-lane encoder of rounds 1-4 is in the history, docs/history.md)
namespace s5 {

// Fused path: zlib-frame a payload of at most DEFL_BLK bytes that sits in LDS (`pay`) as ONE DEFLATE block.
// Leaves the whole record in the LDS bit buffer, bytes [8, total): 78 9c | block | adler32 BE (bytes [0, 8) are
// reserved for the u64 size prefix) and returns total.  obuf_words >= max(plen + 64, sizeof(BuildScratch)) / 4.
template <typename M>
__device__ __forceinline__ uint32_t zlib_frame_fused(DeflShared &S, uint32_t *obuf, uint32_t obuf_words, const uint8_t *pay,
                                                     uint32_t plen, ZOut &z, uint32_t dbg = 0, EarlySize es = EarlySize{nullptr, 0}, uint32_t gen_hint = 0,
                                                     bool prepared = false) {
    const int tid = threadIdx.x;
    z.bitpos = 80;   // 64 bits of size prefix + 16 bits of zlib header, both written later
    z.flushed = 0;
    uint32_t adA = 1, adB = 0;
    deflate_block2<1, NT>(S, obuf, obuf_words, pay, (int)plen, true, z, adA, adB, dbg, es, gen_hint, prepared);
    if (dbg) return 16;
    z.bitpos = (z.bitpos + 7) & ~7u;
    if (tid == 0 && plen) {   // deflate_block2<1> leaves the block's Adler sums per wave: A = 1 + sum x_i, B = len + sum (len - i) x_i
        uint32_t a = 0, b = 0;
        for (int w = 0; w < NW; w++) { a += S.wad[w]; b += S.wad[8 + w]; }
        adB = (uint32_t)(((uint64_t)plen + b) % 65521u);
        adA = (1u + a) % 65521u;
    }
    if (tid == 0) put_bits(obuf, z, z.bitpos, __builtin_bswap32((adB << 16) | adA), 32);
    z.bitpos += 32;
    __syncthreads();
    return z.bitpos >> 3;
}

// ... then into a 16-B aligned HBM slot: [u64 size][record]
template <typename M>
__device__ __forceinline__ uint32_t zlib_compress_fused(DeflShared &S, uint32_t *obuf, uint32_t obuf_words,
                                                        const uint8_t *pay, uint32_t plen, uint8_t *out, uint32_t dbg = 0, uint32_t gen_hint = 0, bool prepared = false) {
    ZOut z;
    const uint32_t total = zlib_frame_fused<M>(S, obuf, obuf_words, pay, plen, z, dbg, EarlySize{nullptr, 0}, gen_hint, prepared);
    if (dbg) {
        if (threadIdx.x == 0) *reinterpret_cast<uint32_t *>(out) = z.bitpos;
        if (dbg == 41 && threadIdx.x < 20) reinterpret_cast<uint32_t *>(out)[4 + threadIdx.x] = threadIdx.x < 19 ? S.clfreq[threadIdx.x] : S.red[6];   // tools/clfreq_dump.py
        return 16;
    }
    flush_words(obuf, reinterpret_cast<uint32_t *>(out), z, true);
    if (threadIdx.x == 0) *reinterpret_cast<uint64_t *>(out) = (uint64_t)(total - 8);
    return total;
}

// ---- ordered single-pass output (decoupled look-back over the record sizes) ----
// state word per read: flag << 62 | value; flag 1 = this read's size, 2 = inclusive prefix up to and including it.
// One 8-byte agent-scope store carries flag and value together, so no fence is needed (G16 R2).
constexpr uint64_t LB_MASK = (1ull << 62) - 1;
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// wave 0 of the workgroup that owns read r (reads are taken in ticket = start order, so every predecessor is
// already resident and will publish).  Returns the exclusive prefix (byte offset of this record in the stream).
// E: state words per lane and step — the window of one look-back step is 64 * E predecessors.  Inclusive prefixes travel one window per
// step (a store seen by a load on another XCD: about a microsecond), so a kernel that retires R reads per microsecond needs a window wider
// than R.  (Round 3 measured E = 1, 2, 4 on both stream kernels: no gain on either — k_encode_stream 13.63 / 13.75 / 13.94 ms — so E stays 1;
// what held the svb-zd stream kernel back was its million tickets on one counter and sizes published only at the end, see there.)
template <int E = 1>
__device__ __forceinline__ uint64_t lookback_offset(unsigned long long *st, uint32_t r, uint64_t mysize, uint32_t *err, bool published) {
    const int lane = lane_id();
    if (!published && lane == 0) __hip_atomic_store(&st[r], (1ull << 62) | mysize, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t acc = 0;                     // per lane; summed over the wave once at the end
    long long base = (long long)r - 1;
    uint32_t spins = 0;
    for (;;) {
        uint64_t v[E];                    // E rows of 64 consecutive words (a row = 512 contiguous bytes), nearest row first
#pragma unroll
        for (int e = 0; e < E; e++) {
            const long long idx = base - 64 * e - lane;
            v[e] = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
        }
        int outcome = 0;                  // 1: reached an inclusive prefix, 2: an unpublished predecessor (wait for it)
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (outcome) continue;
            const uint32_t f = (uint32_t)(v[e] >> 62);
            const uint64_t m0 = __ballot(f == 0), m2 = __ballot(f == 2);
            const int first0 = m0 ? __ffsll((long long)m0) - 1 : 64;
            const int first2 = m2 ? __ffsll((long long)m2) - 1 : 64;
            if (first2 < first0) {        // a prefix is reachable through published sizes: done
                acc += lane <= first2 ? (v[e] & LB_MASK) : 0ull;
                outcome = 1;
            } else {                      // take the published sizes in front of the first unpublished predecessor
                acc += lane < first0 ? (v[e] & LB_MASK) : 0ull;
                if (first0 < 64) { base -= 64 * e + first0; outcome = 2; }
            }
        }
        if (outcome == 1) break;
        if (outcome == 0) { base -= 64 * E; continue; }
        if (++spins > (1u << 22)) { if (lane == 0) *err = 1; break; }   // never hang the GPU: report instead
        __builtin_amdgcn_s_sleep(2);
    }
    const uint64_t excl = wave_sum64(acc);
    if (lane == 0) __hip_atomic_store(&st[r], (2ull << 62) | (excl + mysize), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}
// copy `total` bytes of the LDS record image (word-aligned at obuf) to a byte-aligned HBM destination
// `total` bytes from LDS (obuf: 16-byte aligned) to dst (any alignment: BLOW5 framing has no padding).  Round 4: a lane moves 16 bytes — one aligned
// ds_read_b128, one UNALIGNED 16-byte global store (the memory system takes it: a wave still writes one contiguous kilobyte) — instead of a dword put
// together from two LDS dwords by a funnel shift; the last total % 16 bytes go one per lane
__device__ __forceinline__ void copy_record_out(const uint32_t *obuf, uint32_t total, uint8_t *dst) {
    const int tid = threadIdx.x;
    typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u4a __attribute__((ext_vector_type(4)));
    const u4a *s16 = reinterpret_cast<const u4a *>(obuf);
    const uint32_t nv = total >> 4;
    for (uint32_t i = tid; i < nv; i += NT) *reinterpret_cast<u4u *>(dst + 16u * i) = s16[i];
    const uint32_t tail0 = 16u * nv;
    const uint8_t *ob8 = reinterpret_cast<const uint8_t *>(obuf);
    if ((uint32_t)tid < total - tail0) dst[tail0 + tid] = ob8[tail0 + tid];
}

}  // namespace s5
