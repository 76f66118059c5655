// deflate2_dev.h — the DEFLATE block encoder of round 5: slabs of 256 positions per wave step instead of a contiguous chunk per lane.
//
// Same contract as deflate_block (deflate_dev.h): one block of <= 16 KiB of payload that sits in LDS -> run-length tokens
// (literal | length 3..258 at distance 1) -> dynamic / fixed / stored DEFLATE block in the LDS bit buffer; stock zlib inflates the
// stream to the payload (/root/reference/src/view.c:49, zlib behind slow5_rec_to_mem).  What changed is who owns which byte:
//
//   round 1-4  lane t owns K = ceil(len / 256) CONTIGUOUS bytes.  The bit offset of a lane's first token needs the bit total of every
//              lane in front of it: one pass over all bytes only to count bits (851 of 7331 VALU instructions per 4000-sample read),
//              and the lanes of the svb key area (long zero runs, a handful of tokens) walk their tokens one by one while the lanes
//              of the data area (every byte a literal) run groups of four — the first wave executes both loops with most lanes off
//              (profiles/r04_encode_stages.txt: the emit stage 1534 VALU at 32 active lanes of 64).
//   round 5    a wave walks its region in SLABS of 256 positions, lane t taking four consecutive positions of each slab: every LDS
//              access of the two passes is one aligned dword per lane, the bit offset inside a slab is one wave prefix scan on the
//              DPP path, and the offset of a wave's region comes from PER-WAVE HISTOGRAMS (sum of frequency x code length) —
//              no bit-count pass.  A slab is classified with byte-parallel mask arithmetic on the dword: E = "equal to the
//              previous byte" per position; a position is a literal unless it lies in a sequence of >= 3 E-positions (a run of
//              >= 4 equal bytes), and such a sequence is sent by its LAST position as length-258 matches + one shorter match (or
//              one / two literals), so every decision looks at most two positions ahead — inside the lane's own dword, because the
//              lane's four positions are taken two to the left of it ("centred frame": positions 256 k + 4 t - 2 .. + 1 from the
//              dwords at 256 k + 4 t - 4 and 256 k + 4 t).  Slabs without such a sequence (nearly all of the data area) run
//              without masks and without branches: 4 histogram adds in the first pass; 4 code loads, one 64-bit OR value, one scan
//              in the second.  Only a sequence that started in front of the slab needs its start: a wave prefix-max over the
//              last break of every lane, and behind that a backward search (rare).
//
// The token sequence is the one deflate_block produces (same literals, same matches), so sizes and the "<= zlib level 6" property of
// tests/test_full_size.py carry over; code lengths, code-length header and block choice are the same functions.
#pragma once
// (included by deflate_dev.h in front of its zlib framing functions: everything above that point is in scope)

namespace s5 {

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// 0x80 in every byte of x that is zero (exact: no carries between bytes)
__device__ __forceinline__ uint32_t zbytes(uint32_t x) {
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;   // bit 7: the low seven bits are not all zero
    return ~(t | x | 0x7F7F7F7Fu);
}
// spread mask (bit 8q + 7 for slot q) of the slots of a four-position frame starting at a0 that lie in [first, len)
__device__ __forceinline__ uint32_t frame_mask(int a0, int first, int len) {
    const int lo = min(4, max(0, first - a0)), hi = min(4, max(0, len - a0));
    const uint32_t mh = hi >= 4 ? 0xFFFFFFFFu : (1u << (8 * hi)) - 1u;
    const uint32_t ml = lo >= 4 ? 0xFFFFFFFFu : (1u << (8 * lo)) - 1u;
    return mh & ~ml & 0x80808080u;
}
__device__ __forceinline__ int top_slot(uint32_t m) { return (31 - __clz((int)m)) >> 3; }   // m != 0, bits at 8q + 7

// length symbol of a match of L bytes (3..258): symbol, extra bits, extra value
__device__ __forceinline__ void length_symbol(int L, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    const int l = L - 3;
    eb = 0; ev = 0;
    if (L == 258) sym = 285;
    else if (l < 8) sym = 257 + l;
    else {
        const int nb = 29 - __clz(l);
        sym = 261 + 4 * nb + ((l >> nb) & 3);
        eb = nb;
        ev = l & ((1 << nb) - 1);
    }
}

// last position p < pos with a break (p == 0 or buf[p] != buf[p - 1]); pos >= 1.  All 64 lanes, uniform result.
__device__ __forceinline__ int find_break_before(const uint8_t *__restrict__ buf, int pos) {
    const int lane = lane_id();
    for (int hi = pos - 1;; hi -= 64) {
        const int p = hi - lane;
        const bool brk = p >= 0 && (p == 0 || buf[p] != buf[p - 1]);
        const uint64_t m = __ballot(brk);
        if (m) return hi - (__ffsll((long long)m) - 1);
    }
}

// What a lane knows of its four positions of a slab (centred frame: positions p0 .. p0 + 3, p0 = 256 k + 4 lane - 2).
struct SlabCls {
    uint32_t bytes;     // the four bytes
    uint32_t C;         // E (equal to the previous byte, valid positions only) per slot
    uint32_t Cp1;       // E of the position after each slot
    uint32_t member;    // slots inside a sequence of >= 3 E-positions
    uint32_t V;         // valid slots (0 <= p < len)
};
// carryE: E of the aligned frame of lane 63 of the previous slab (uniform); updated.
template <bool EDGE>
__device__ __forceinline__ SlabCls classify_slab(const uint32_t *__restrict__ buf32, int k, int len, uint32_t &carryE) {
    const int lane = lane_id();
    const int di = 64 * k + lane;
    uint32_t w, wp;
    if (EDGE) {
        const int ndw = (len + 3) >> 2;
        w = di < ndw ? buf32[di] : 0u;
        wp = di - 1 < ndw ? buf32[di - 1] : 0u;      // (di = 0: the word in front of the buffer — LDS of this workgroup, masked out below)
    } else {
        w = buf32[di];
        wp = buf32[di - 1];
    }
    uint32_t E = zbytes(w ^ alignbit(w, wp, 24));    // aligned frame: positions 4 di .. 4 di + 3 (bits 8q + 7 only)
    if (EDGE) E &= frame_mask(4 * di, 1, len);
    const uint32_t Ep = dpp_u32<DPP_WAVE_SHR1>(carryE, E);
    carryE = (uint32_t)__builtin_amdgcn_readlane((int)E, 63);
    SlabCls c;
    c.bytes = alignbit(w, wp, 16);
    c.C = alignbit(E, Ep, 16);
    const uint32_t Cm1 = alignbit(E, Ep, 8);
    c.Cp1 = alignbit(E, Ep, 24);
    // E at p - 2 is Ep, at p + 2 is E.  A slot is a member if it is E and one of the three windows of three around it is all E.
    c.member = c.C & ((c.Cp1 & (E | Cm1)) | (Ep & Cm1));
    c.V = EDGE ? frame_mask(256 * k + 4 * lane - 2, 0, len) : 0x80808080u;
    return c;
}

// The tail a lane sends for the sequence of >= 3 E-positions that ENDS at one of its slots: body = positions of the sequence.
struct SlabTail {
    uint32_t ql;        // slot of the sequence's last position (4: none)
    uint32_t nfull;     // matches of 258
    uint32_t rem;       // then: >= 3 one match of rem, 1 / 2 literals of the run byte, 0 nothing
    uint32_t runbyte;
};
struct WaveCarry {      // per wave, uniform
    int lastb;          // last break position in front of the current slab, when known
    bool ok;
};
// Uniform call (all lanes) for a slab that holds members.  k: slab, c: its classification.
__device__ __forceinline__ SlabTail slab_tail(const uint8_t *__restrict__ buf, int k, const SlabCls &c, WaveCarry &wc) {
    const int lane = lane_id();
    const int p0 = 256 * k + 4 * lane - 2;
    const uint32_t lastmem = c.member & ~c.Cp1;                   // at most one slot per lane: two sequences are >= 4 positions apart
    const uint32_t Bm = c.V & ~c.C;                                // breaks
    const int lb = Bm ? p0 + top_slot(Bm) : -1;
    const int Lincl = wave_incl_max(lb);
    const int Lprev = (int)wave_prev((uint32_t)Lincl, 0xFFFFFFFFu);
    const int slab_last = __builtin_amdgcn_readlane(Lincl, 63);
    SlabTail t;
    t.ql = 4; t.nfull = 0; t.rem = 0; t.runbyte = 0;
    int s = 0;
    const bool has = lastmem != 0u;
    if (has) {
        t.ql = (uint32_t)top_slot(lastmem);
        const uint32_t below = Bm & ((1u << (8 * t.ql)) - 1u);
        s = below ? p0 + top_slot(below) : Lprev;
    }
    if (__ballot(has && s < 0)) {                                  // (uniform) a sequence that started in front of this slab
        if (!wc.ok) { wc.lastb = find_break_before(buf, 256 * k - 2); wc.ok = true; }
        if (has && s < 0) s = wc.lastb;
    }
    if (slab_last >= 0) { wc.lastb = slab_last; wc.ok = true; }
    if (has) {
        const int body = p0 + (int)t.ql - s;                       // >= 3
        t.nfull = (uint32_t)((body * 16257) >> 22);                // body / 258 for body < 70000
        t.rem = (uint32_t)body - 258u * t.nfull;
        t.runbyte = (c.bytes >> (8 * t.ql)) & 255u;
    }
    return t;
}

// extra bits of a length symbol
__device__ __forceinline__ uint32_t length_extra_bits(uint32_t sym) { return sym >= 265u && sym < 285u ? (sym - 261u) >> 2 : 0u; }
// E of the aligned dword in front of slab k (lane 0's left neighbour there): uniform addresses, uniform result
__device__ __forceinline__ uint32_t carry_e_before(const uint32_t *__restrict__ buf32, int k, int len) {
    const int di = 64 * k - 1;                              // positions 256 k - 4 .. 256 k - 1
    const uint32_t w = buf32[di], wp = buf32[di - 1];
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(zbytes(w ^ alignbit(w, wp, 24)) & frame_mask(4 * di, 1, len)));
}

// Token list (round 5, second step).  A "general" slab — one that holds a sequence of >= 3 E-positions, or touches an end of the block —
// costs three to four times a plain one in either pass (masks, the tail analysis with its wave prefix-max, per-slot predicates), and
// on an svb-zd payload the key area is made of such slabs: a fifth of the bytes, half of the two passes' instructions.  So pass 1
// does the analysis ONCE and leaves the slab's tokens — few: a key slab of 256 positions holds ~40 — as 16-bit words
// (symbol | extra value << 9) in a list that lives in S.freq (not otherwise used here); the histogram is counted from the list, and
// pass 2 sends a listed slab 64 tokens per step: one load, one code lookup, one scan, one OR per lane.  DEFL2_LIST_CAP tokens for
// the whole workgroup, handed out by one LDS counter; a slab that finds the list full is redone from the bytes in pass 2.
constexpr uint32_t DEFL2_LIST_CAP = 640;
// Staged (multi-block) form: a 16 KiB block of the key area of a long read holds ~2700 tokens in 64 general slabs — four times what S.freq
// takes, and the slabs that found the list full were redone at ~220 instructions each in pass 2.  Such a block's OUTPUT is small (that is what
// makes its slabs general), so the list's overflow lives in the bit buffer itself, growing DOWN from the histograms (word wf_at) while the
// block's bits grow up from word 0: K bytes of keys make ~K / 6 tokens (K / 3 bytes of list) and shrink the output by ~0.5 K bytes, so the two
// do not meet.  Pass 2 checks that they did not; otherwise the list's words are cleared and those slabs are redone from the bytes.
constexpr uint32_t DEFL2_LISTB_MIN = 64;    // the list never reaches below this word
constexpr uint32_t SEG_FAST = 0xFFFFFFFFu, SEG_REDO = 0xFFFFFFFEu, SEG_MASK = 0xFFFFFFFDu;

// What has to be zero before pass 1 of a block: the per-wave histograms and the token list's counter.  (Everything else the block
// leaves in S is written before it is read.)  No barrier in here: the caller orders it.
template <int TN = NT>
__device__ __forceinline__ void deflate2_prepare(DeflShared &S, uint32_t *obuf, uint32_t obuf_words) {
    constexpr int NWV = TN / 64;
    typedef uint32_t u4a __attribute__((ext_vector_type(4)));
    u4a *w16 = reinterpret_cast<u4a *>(obuf + (obuf_words - (uint32_t)(NWV * 288)));   // (obuf is 16-byte aligned, obuf_words and 288 are multiples of four)
    for (int i = threadIdx.x; i < NWV * 72; i += TN) w16[i] = u4a{0u, 0u, 0u, 0u};
    if (threadIdx.x == 0) { S.lalloc = 0; S.red[7] = 0; }
}

template <int MODE, int TN = NT>
__device__ __forceinline__ void deflate_block2(DeflShared &S, uint32_t *obuf, uint32_t obuf_words, const uint8_t *__restrict__ buf, int len,
                                               bool final, ZOut &z, uint32_t &adA, uint32_t &adB, uint32_t dbg = 0,
                                               EarlySize es = EarlySize{nullptr, 0}, uint32_t gen_hint = 0, bool prepared = false) {
    static_assert(MODE == 1 || MODE == 2, "fused single block (1) or staged multi-block (2): the bit buffer is this function's to clear");
    constexpr bool FUSED = MODE == 1;
    constexpr int NWV = TN / 64;
    static_assert(TN % 64 == 0 && NWV >= 4 && NWV <= 8, "S.wtot holds four words per wave");
    const int tid = threadIdx.x, lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane(wave_id());   // (rotating the roles with the workgroup number was measured: by its low bits no change,
                                                                // by a hash of it 3 % slower — the hardware already spreads the waves of successive workgroups)
    if (len == 0) {   // empty stream: a fixed block holding only end-of-block
        for (uint32_t i = tid; i < obuf_words; i += TN) obuf[i] = 0;
        __syncthreads();
        if (tid == 0) { if (FUSED) put_bits(obuf, z, 64, 0x9c78u, 16); else obuf[0] = z.carry; }
        if (tid == 0) put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (1u << 1), 10);
        z.bitpos += 10;
        publish_size(es, z.bitpos);
        __syncthreads();
        return;
    }
    // one histogram of the 286 lit/len symbols per wave, in the tail of the bit buffer (dead until the tokens are packed)
    const uint32_t wf_at = obuf_words - (uint32_t)(NWV * 288);
    uint32_t *wfa = obuf + wf_at;
    uint32_t *wf = wfa + wv * 288;
    if (!prepared) {      // (uniform; the fused kernels clear these under their signal loads: deflate2_prepare)
        deflate2_prepare<TN>(S, obuf, obuf_words);
        __syncthreads();
    }

    const uint32_t *buf32 = reinterpret_cast<const uint32_t *>(buf);
    constexpr bool POOLB = MODE == 2;                       // the token list: S.freq, then (staged) the bit buffer below the histograms
    const bool poolb_ok = POOLB && wf_at >= DEFL2_LISTB_MIN + 64u;
    // token index t: t < DEFL2_LIST_CAP lies in S.freq, the ones above in the bit buffer (staged only); a slab's tokens never straddle the two
    uint16_t *tla = reinterpret_cast<uint16_t *>(S.freq);
    uint16_t *tlb_top = reinterpret_cast<uint16_t *>(obuf + wf_at);    // a slab's tokens [st, st + total) of the second pool end (st - CAP) tokens below this
    const uint32_t list_cap = DEFL2_LIST_CAP + (poolb_ok ? (wf_at - DEFL2_LISTB_MIN) * 2u : 0u);
    const int nsl = (len + 2 + 255) >> 8;                   // slabs: centred frames cover positions [-2, 256 nsl - 2)
    // Which slabs a wave takes: contiguous regions of about equal COST.  gen_hint = bytes at the front of the block that the caller expects
    // to be run-heavy (an svb-zd payload's head and key area): their slabs are general ones, about four times the work of a plain slab, and in
    // equal-sized regions the first wave would do most of pass 1 alone.  Only a hint: any split is correct.
#ifndef S5_DEFL2_GEN_W
#define S5_DEFL2_GEN_W 4
#endif
#ifndef S5_DEFL2_GEN_W_FUSED
#define S5_DEFL2_GEN_W_FUSED 5
#endif
    // weight of a general slab (a plain one: 1).  Measured with pass 1's plain slabs taken in pairs: fused 4 / 5 / 6 = 11.68 / 11.62 / 11.63 ms per
    // 1 M reads; staged (mixed | long reads) 4 / 5 / 6 = 536 | 743, 526 | 726, 528 | 729 GB/s
    constexpr int GW = FUSED ? S5_DEFL2_GEN_W_FUSED : S5_DEFL2_GEN_W;
    const int gs = min(nsl, gen_hint ? (int)((gen_hint + 2u + 255u) >> 8) : 0);
    const int cost_all = (GW - 1) * gs + nsl;
    auto bound = [&](int w) -> int {
        if (w >= NWV) return nsl;
        const int T = (cost_all * w + NWV / 2) / NWV;         // cost in front of wave w's first slab
        return T <= GW * gs ? (T + GW / 2) / GW : gs + (T - GW * gs);
    };
    const int k0 = bound(wv), k1 = bound(wv + 1);
    uint32_t segs = SEG_FAST;                               // lane j: what pass 2 does with slab k0 + j (start | count << 16 of its tokens)

    // ---- pass 1: histogram, Adler-32 partial sums, match / extra-bit counts; token lists of the general slabs ----
    {
        uint32_t a_acc = 0, b_acc = 0, d_acc = 0, nmatch = 0, nextra = 0;
        uint32_t carryE = k0 > 0 && k0 < k1 ? carry_e_before(buf32, k0, len) : 0u;
        WaveCarry wc{-1, k0 == 0};
        for (int k = k0; k < k1; k++) {
#ifndef S5_DEFL2_NO_PAIRS
            // TWO plain slabs in one step (as pass 2 does), eight consecutive positions per lane: three loads instead of four, one DPP instead of
            // two, one membership test, one trip — 58 instructions for what two single steps do in 84.  Tried behind the hinted run-heavy front only
            // (there every attempt would fail); a pair that holds a sequence after all is done slab by slab below.
            if (k >= gs && k > 0 && k + 1 < k1 && 256 * k + 512 <= len) {
                const int d0 = 64 * k + 2 * lane;
                const uint32_t wp = buf32[d0 - 1], w0 = buf32[d0], w1 = buf32[d0 + 1];
                const uint32_t E0 = zbytes(w0 ^ alignbit(w0, wp, 24)), E1 = zbytes(w1 ^ alignbit(w1, w0, 24));
                const uint32_t EpA = dpp_u32<DPP_WAVE_SHR1>(carryE, E1);
                auto member_of = [](uint32_t E, uint32_t Ep) {           // (classify_slab's rule)
                    const uint32_t C = alignbit(E, Ep, 16), Cm1 = alignbit(E, Ep, 8), Cp1 = alignbit(E, Ep, 24);
                    return C & ((Cp1 & (E | Cm1)) | (Ep & Cm1));
                };
                if (__ballot((member_of(E0, EpA) | member_of(E1, E0)) != 0u) == 0ull) {
                    carryE = (uint32_t)__builtin_amdgcn_readlane((int)E1, 63);
                    const uint32_t bA = alignbit(w0, wp, 16), bB = alignbit(w1, w0, 16);
                    atomicAdd(&wf[bA & 255u], 1u);
                    atomicAdd(&wf[(bA >> 8) & 255u], 1u);
                    atomicAdd(&wf[(bA >> 16) & 255u], 1u);
                    atomicAdd(&wf[bA >> 24], 1u);
                    atomicAdd(&wf[bB & 255u], 1u);
                    atomicAdd(&wf[(bB >> 8) & 255u], 1u);
                    atomicAdd(&wf[(bB >> 16) & 255u], 1u);
                    atomicAdd(&wf[bB >> 24], 1u);
                    const uint32_t wA = (uint32_t)(len - (256 * k + 8 * lane - 2));     // Adler: weight of the lane's first position
                    const uint32_t sA = __builtin_amdgcn_udot4(bA, 0x01010101u, 0u, false), sB = __builtin_amdgcn_udot4(bB, 0x01010101u, 0u, false);
                    a_acc += sA + sB;
                    b_acc = __umul24(wA, sA) + __umul24(wA - 4u, sB) + b_acc;
                    d_acc = __builtin_amdgcn_udot4(bA, 0x03020100u, d_acc, false);
                    d_acc = __builtin_amdgcn_udot4(bB, 0x03020100u, d_acc, false);
                    wc.ok = false;
                    k++;
                    continue;
                }
            }
#endif
            const bool full = k > 0 && 256 * k + 256 <= len;     // every slot of every lane is a valid position >= 1
            const uint32_t wgt = (uint32_t)(len - (256 * k + 4 * lane - 2));   // Adler: weight of slot 0
            SlabCls c;
            if (full) c = classify_slab<false>(buf32, k, len, carryE);
            else c = classify_slab<true>(buf32, k, len, carryE);
            const bool anymem = __ballot(c.member != 0u) != 0ull;
            if (full && !anymem) {
                atomicAdd(&wf[c.bytes & 255u], 1u);
                atomicAdd(&wf[(c.bytes >> 8) & 255u], 1u);
                atomicAdd(&wf[(c.bytes >> 16) & 255u], 1u);
                atomicAdd(&wf[c.bytes >> 24], 1u);
                const uint32_t s4 = __builtin_amdgcn_udot4(c.bytes, 0x01010101u, 0u, false);
                a_acc += s4;
                b_acc = __umul24(wgt, s4) + b_acc;
                d_acc = __builtin_amdgcn_udot4(c.bytes, 0x03020100u, d_acc, false);
                wc.ok = false;
                continue;
            }
            {   // Adler: bytes of invalid slots count as zero
                uint32_t bytes = c.bytes;
                if (!full) { const uint32_t vb = c.V >> 7; bytes &= (vb << 8) - vb; }
                const uint32_t s4 = __builtin_amdgcn_udot4(bytes, 0x01010101u, 0u, false);
                a_acc += s4;
                b_acc = __umul24(wgt, s4) + b_acc;
                d_acc = __builtin_amdgcn_udot4(bytes, 0x03020100u, d_acc, false);
            }
            const uint32_t lit = c.V & ~c.member;
            if (!anymem) {
                // an end slab without a run: plain literals under a validity mask in both passes, no list
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (lit & (0x80u << (8 * q))) atomicAdd(&wf[(c.bytes >> (8 * q)) & 255u], 1u);
                if (lane == k - k0) segs = SEG_MASK;
                wc.ok = false;
                continue;
            }
            const SlabTail t = slab_tail(buf, k, c, wc);
            uint32_t tsym = 0, teb = 0, tev = 0;           // the tail's match of rem bytes
            if (t.ql < 4 && t.rem >= 3) length_symbol((int)t.rem, tsym, teb, tev);
            const uint32_t ntail = t.ql < 4 ? t.nfull + (t.rem >= 3 ? 1u : t.rem) : 0u;
            // literal slots below each slot (token index inside the lane, before the tail is spliced in)
            const uint32_t i1 = (lit >> 7) & 1u, i2 = i1 + ((lit >> 15) & 1u), i3 = i2 + ((lit >> 23) & 1u);
            const uint32_t cnt = i3 + (lit >> 31) + ntail;
            const uint32_t incl = wave_incl_add(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t st = 0;
            if (lane == 0) {
                st = atomicAdd(&S.lalloc, total);
                if (st < DEFL2_LIST_CAP && st + total > DEFL2_LIST_CAP) {
                    // would straddle the two pools: the slab takes `total` entries of the SECOND pool — the counter is first raised to the pool's
                    // start (atomicMax: whoever else got in between has raised it already), then `total` are taken there.  (Round 5 took
                    // total + CAP - st a second time: the second pool lost up to CAP entries per straddle, listb_words grew with them.)
                    atomicMax(&S.lalloc, (uint32_t)DEFL2_LIST_CAP);
                    st = atomicAdd(&S.lalloc, total);
                }
            }
            st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
            const bool listed = st + total <= list_cap && st + total < 65536u;
            uint16_t *tl = st < DEFL2_LIST_CAP ? tla : tlb_top - (st - DEFL2_LIST_CAP) - total - st;     // (indexed by st + ...)
            // tokens in position order: a literal slot in front of the tail's slot (only slot 0 under a tail at slot 3 can be one), the tail,
            // the literal slots behind it.  The histogram is counted on the way.
            const uint32_t base = st + incl - cnt;
            const uint32_t idx[4] = {0u, i1, i2, i3};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (lit & (0x80u << (8 * q))) {
                    const uint32_t b = (c.bytes >> (8 * q)) & 255u;
                    atomicAdd(&wf[b], 1u);
                    if (listed) tl[base + idx[q] + ((uint32_t)q > t.ql ? ntail : 0u)] = (uint16_t)b;
                }
            if (t.ql < 4) {
                uint32_t at = base + (t.ql == 3 ? (lit >> 7) & 1u : 0u);   // (slots ql - 1, ql - 2 are members: only slot 0 under ql = 3 can be a literal)
                if (t.nfull) {
                    atomicAdd(&wf[285], t.nfull);
                    nmatch += t.nfull;
                    if (listed) for (uint32_t i = 0; i < t.nfull; i++) tl[at++] = (uint16_t)285;
                }
                if (t.rem >= 3) {
                    atomicAdd(&wf[tsym], 1u);
                    nmatch += 1;
                    nextra += teb;
                    if (listed) tl[at] = (uint16_t)(tsym | (tev << 9));
                } else if (t.rem) {
                    atomicAdd(&wf[t.runbyte], t.rem);
                    if (listed) for (uint32_t i = 0; i < t.rem; i++) tl[at++] = (uint16_t)t.runbyte;
                }
            }
            if (lane == k - k0) segs = listed ? st | (total << 16) : SEG_REDO;
        }
        // Adler: sum of (len - p) x_p = sum over slabs of wgt * (x0 + x1 + x2 + x3) - (0 x0 + 1 x1 + 2 x2 + 3 x3)
        const uint32_t a_w = wave_sum(a_acc);
        const uint32_t b_w = wave_sum((b_acc - d_acc) % 65521u);
        const uint32_t mx = wave_sum(nmatch | (nextra << 16));     // (a block of <= 16 KiB: < 5462 matches, < 27310 extra bits — one sum for both)
        if (lane == 0) {
            S.wad[wv] = a_w;
            S.wad[8 + wv] = b_w;
            S.wtot[16 + wv] = mx >> 16;
            S.wtot[24 + wv] = mx & 0xFFFFu;
        }
    }
    __syncthreads();
    if (dbg == 2) { z.bitpos += wfa[tid] + S.wad[tid & 15]; return; }   // tools/stage_time.py cut-off

    // (uniform) words of the bit buffer, below the histograms, that hold tokens
    const uint32_t la = S.lalloc;
    const uint32_t listb_words = poolb_ok && la > DEFL2_LIST_CAP ? (((la < list_cap ? la : list_cap) - DEFL2_LIST_CAP + 7u) >> 3) << 2 : 0u;
    const bool poolb_used = listb_words != 0u;
    // ---- code lengths (wave 0, no tree: assign_lengths_wave over the sum of the histograms); the other waves clear the bit buffer ----
    if (wv == 0) {
#ifndef S5_DEFL2_ABSORB
#define S5_DEFL2_ABSORB 2
#endif
        const bool ok = assign_lengths_wave<15, NWV, true, S5_DEFL2_ABSORB>(wfa, NLIT, S.lens, S.blcount, S.bins);
        if (lane == 0) S.dbg = ok ? 0u : 1u;
    } else {
        typedef uint32_t u4a __attribute__((ext_vector_type(4)));
        u4a *o16 = reinterpret_cast<u4a *>(obuf);
        const uint32_t zend = (wf_at - listb_words) / 4;   // (staged: the words above hold the block's token list, if it needed them)
        for (uint32_t i = (uint32_t)(wv * 64 + lane) - 64u; i < zend; i += TN - 64) o16[i] = u4a{0u, 0u, 0u, 0u};   // (waves 1 ..)
    }
    __syncthreads();
    if (dbg == 3) { z.bitpos += S.lens[tid]; return; }
    if (tid == 0) {
        if (FUSED) put_bits(obuf, z, 64, 0x9c78u, 16);   // CMF/FLG 78 9c (deflate, 32K window, default level)
        else obuf[0] = z.carry;                          // the stream's pending partial word
    }
    // ---- wave 0: code-length header; wave 1: canonical codes; waves 2..: body bits of every wave's region under the dynamic / fixed code ----
    if (wv == 0) {
        if (lane == 0) {   // two 1-bit distance codes (complete code; only code 0 = distance 1 is ever sent)
            S.lens[DOFF] = 1; S.lens[DOFF + 1] = 1;
            S.code[DOFF] = 0u | (1u << 16); S.code[DOFF + 1] = 1u | (1u << 16);
        }
        cl_header_wave(S, 2);
    } else if (wv == 1) {
        assign_codes_wave(S.blcount, S.lens, NLIT, S.code);
    } else {
        // fixed code: 8 bits for every symbol, 9 for 144..255, 7 for 256..279 -> 8 * count + count(144..255) - count(256..279)
        for (int h = wv - 2; h < NWV; h += NWV - 2) {
            const uint32_t *fh = wfa + h * 288;
            const uint32_t f0 = fh[lane], f1 = fh[64 + lane], f2 = fh[128 + lane], f3 = fh[192 + lane], f4 = lane < NLIT - 256 ? fh[256 + lane] : 0u;
            uint32_t dynb = f0 * S.lens[lane] + f1 * S.lens[64 + lane] + f2 * S.lens[128 + lane] + f3 * S.lens[192 + lane] + f4 * S.lens[256 + lane];
            uint32_t fixb = 8u * (f0 + f1 + f2 + f3 + f4) + (lane >= 16 ? f2 : 0u) + f3 - (lane < 24 ? f4 : 0u);
            dynb = wave_sum(dynb);
            fixb = wave_sum(fixb);
            if (lane == 0) { S.wtot[h] = dynb; S.wtot[8 + h] = fixb; }
            // this wave was the histogram's last reader: its words become bit buffer right here (words 286, 287 were never counted into), so the
            // barrier below is the only one between the codes and the tokens
            uint32_t *fz = wfa + h * 288;
            fz[lane] = 0u; fz[64 + lane] = 0u; fz[128 + lane] = 0u; fz[192 + lane] = 0u;
            if (lane < NLIT - 256) fz[256 + lane] = 0u;
        }
    }
    __syncthreads();
    if (dbg == 4 || dbg == 41) { z.bitpos += S.wtot[tid & 31] + S.red[6] + S.code[tid]; return; }
    // per-wave totals: lane h of every wave holds wave h's numbers
    uint32_t my_dyn = 0, my_fix = 0, my_x = 0, my_m = 0;
    if (lane < NWV) { my_dyn = S.wtot[lane]; my_fix = S.wtot[8 + lane]; my_x = S.wtot[16 + lane]; my_m = S.wtot[24 + lane]; }
    // (the numbers live in lanes 0 .. NWV - 1 <= 7 of one row: an inclusive scan is two or three row shifts, not the wave's six steps)
    auto row8_incl = [](uint32_t v) {
        v += dpp_u32<DPP_ROW_SHR1>(0, v);
        v += dpp_u32<DPP_ROW_SHR2>(0, v);
        if (NWV > 4) v += dpp_u32<DPP_ROW_SHR4>(0, v);
        return v;
    };
    const uint32_t dyn_body_all = (uint32_t)__builtin_amdgcn_readlane((int)row8_incl(my_dyn + my_x + my_m), NWV - 1);          // tokens incl. extra and distance bits
    const uint32_t fix_body_all = (uint32_t)__builtin_amdgcn_readlane((int)row8_incl(my_fix + my_x + 5u * my_m), NWV - 1);
    const uint32_t eob_dyn = S.lens[256];
    const uint32_t hdr_dyn = 17 + 3 * S.hclen + S.red[6];
    const uint32_t dyn_total = hdr_dyn + dyn_body_all + eob_dyn;
    const uint32_t fix_total = 3 + fix_body_all + 7;
    const uint32_t sto_total = 3 + ((0u - (z.bitpos + 3)) & 7) + 32 + 8u * (uint32_t)len;
    if (!FUSED) {   // Adler-32 running update (RFC 1950): B' = B + len * A + sum (len - i) x_i.  (Fused: one block — the frame's writer takes
                    // the per-wave sums (S.wad) itself, on one lane, instead of a 64-bit modulo on every lane of the workgroup.)
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) { a += S.wad[w]; b += S.wad[8 + w]; }
        const uint32_t nb = (uint32_t)(((uint64_t)adB + (uint64_t)len * adA + b) % 65521u);
        adA = (adA + a) % 65521u;
        adB = nb;
    }
    if (sto_total <= dyn_total && sto_total <= fix_total) {
        // ---- stored block ----
        const uint32_t bytepos = (z.bitpos + 3 + 7) >> 3;
        publish_size(es, (bytepos + 4 + (uint32_t)len) * 8);
        uint8_t *ob8 = reinterpret_cast<uint8_t *>(obuf) + (bytepos - z.flushed * 4);
        if (poolb_used) {
            // (staged, uniform) the token list of this block's general slabs stands in the bit buffer below the histograms, and waves 1.. cleared
            // only the words below it: a stored block's bytes are plain stores that may END inside those words, and the stream's next bits — the
            // Adler-32 trailer, the next block's carry — are ORed into whatever the rest of that word and the next one hold.  Clear them first.
            typedef uint32_t u4a __attribute__((ext_vector_type(4)));
            u4a *l16 = reinterpret_cast<u4a *>(obuf + (wf_at - listb_words));
            for (uint32_t i = tid; i < listb_words / 4; i += TN) l16[i] = u4a{0u, 0u, 0u, 0u};
        }
        __syncthreads();                                   // the histogram words are zero
        if (tid == 0) {
            put_bits(obuf, z, z.bitpos, final ? 1u : 0u, 3);
            put_bits(obuf, z, bytepos * 8, (uint32_t)len | ((~(uint32_t)len) << 16), 32);
        }
        __syncthreads();
        for (int i = tid; i < len; i += TN) ob8[4 + i] = buf[i];
        z.bitpos = (bytepos + 4 + (uint32_t)len) * 8;
        __syncthreads();
        return;
    }
    const bool use_fixed = fix_total < dyn_total || S.dbg != 0;   // S.dbg: the length assignment gave up (never seen)
    const uint32_t dist_bits = use_fixed ? 5u : 1u;
    const uint32_t pos0 = z.bitpos + (use_fixed ? 3u : hdr_dyn);
    // bit offset of every wave's region: an exclusive scan over lanes 0 .. NWV - 1
    const uint32_t my_t = (use_fixed ? my_fix : my_dyn) + my_x + my_m * dist_bits;
    const uint32_t t_incl = row8_incl(my_t);
    const uint32_t total_bits = (uint32_t)__builtin_amdgcn_readlane((int)t_incl, NWV - 1);
    uint32_t wbase = pos0 + (wv ? (uint32_t)__builtin_amdgcn_readlane((int)t_incl, wv - 1) : 0u);
    const uint32_t eob = use_fixed ? fixed_code(256) : S.code[256];
    publish_size(es, pos0 + total_bits + (eob >> 16));
    if (use_fixed) {
        __syncthreads();                                   // everyone has read the dynamic code's numbers
        for (int s = tid; s < 288; s += TN) { S.code[s] = fixed_code(s); S.lens[s] = (uint8_t)fixed_len(s); }
    }
    // staged: the token list in the bit buffer is good for pass 2 only if the block's bits stay below it
    const bool list_ok = !poolb_used || ((pos0 + total_bits + 15u + 31u) >> 5) - z.flushed + 4u <= wf_at - listb_words;
    if (!list_ok) {
        typedef uint32_t u4a __attribute__((ext_vector_type(4)));
        u4a *l16 = reinterpret_cast<u4a *>(obuf + (wf_at - listb_words));
        for (uint32_t i = tid; i < listb_words / 4; i += TN) l16[i] = u4a{0u, 0u, 0u, 0u};
    }
    if (use_fixed || !list_ok) __syncthreads();            // (uniform, rare) the fixed codes / the cleared list words are in place
    if (wv == 0) {
        // block header on one wave, beside the other waves' tokens (disjoint bits, atomic ORs)
        if (use_fixed) {
            if (lane == 0) put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (1u << 1), 3);
        } else {
            // BFINAL, BTYPE = 10, HLIT, HDIST (2 codes), HCLEN in one 17-bit field; the HCLEN 3-bit code-length-code lengths one per lane;
            // the code-length sequence five entries per lane
            const uint32_t hclen = S.hclen;
            if (lane == 0) put_bits(obuf, z, z.bitpos, (final ? 1u : 0u) | (2u << 1) | ((S.hlit - 257) << 3) | (1u << 8) | ((hclen - 4) << 13), 17);
            if (lane < (int)hclen) {
                const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                put_bits(obuf, z, z.bitpos + 17 + 3 * lane, S.cllens[order[lane]], 3);
            }
            const int ncl = S.ncl;
            uint32_t cv[5], cn[5], sum = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int e = 5 * lane + q;
                cv[q] = 0; cn[q] = 0;
                if (e < ncl) {
                    const uint32_t ent = S.clseq[e];
                    const uint32_t sym = ent & 31, cc = S.clcode[sym], cl = cc >> 16;
                    const uint32_t eb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
                    cv[q] = (cc & 0xFFFF) | ((ent >> 5) << cl);
                    cn[q] = cl + eb;
                }
                sum += cn[q];
            }
            uint32_t p = z.bitpos + 17 + 3 * hclen + wave_incl_add(sum) - sum;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                if (cn[q]) put_bits(obuf, z, p, cv[q], cn[q]);
                p += cn[q];
            }
        }
    }
    if (dbg == 5) { z.bitpos += wbase; return; }

    // ---- pass 2: the tokens, slab by slab: one scan gives every lane its bit offset ----
    {
        auto or_bits = [&](uint32_t pos, uint64_t v, uint32_t nb) {          // nb <= 60
            const uint32_t w = (pos >> 5) - z.flushed, sh = pos & 31;
            const uint64_t lo = v << sh;
            atomicOr(&obuf[w], (uint32_t)lo);
            atomicOr(&obuf[w + 1], (uint32_t)(lo >> 32));
            if (sh + nb > 64) atomicOr(&obuf[w + 2], (uint32_t)(v >> (64 - sh)));   // rare: four long codes
        };
        // the codes of four literal bytes as one value of <= 60 bits
        auto lit4 = [&](uint32_t bytes, uint64_t &v, uint32_t &nb) {
            const uint32_t c0 = S.code[bytes & 255u], c1 = S.code[(bytes >> 8) & 255u], c2 = S.code[(bytes >> 16) & 255u], c3 = S.code[bytes >> 24];
            const uint32_t n0 = c0 >> 16, n1 = c1 >> 16, n2 = c2 >> 16, n3 = c3 >> 16;
            const uint32_t lo = (c0 & 0xFFFFu) | ((c1 & 0xFFFFu) << n0), nlo = n0 + n1;
            const uint32_t hi = (c2 & 0xFFFFu) | ((c3 & 0xFFFFu) << n2), nhi = n2 + n3;
            v = (uint64_t)lo | ((uint64_t)hi << nlo);
            nb = nlo + nhi;
        };
        for (int k = k0; k < k1; k++) {
            const uint32_t seg = (uint32_t)__builtin_amdgcn_readlane((int)segs, k - k0);
            if (seg == SEG_FAST && k + 1 < k1 && (uint32_t)__builtin_amdgcn_readlane((int)segs, k + 1 - k0) == SEG_FAST) {
                // TWO plain slabs in one step, eight consecutive positions per lane.  With four positions a lane's codes span ~33 bits, so
                // neighbouring lanes often OR into the same word in the same instruction — same-address LDS atomics serialise, and pass 2
                // was bound by the LDS pipe (91 % busy, two thirds of it conflict cycles: profiles/r05_encode_stages.txt).  With eight a
                // lane spans >= 32 bits: the word indices of one instruction rise strictly from lane to lane.  Half the scans, too.
                const int d0 = 64 * k + 2 * lane;
                const uint32_t wp = buf32[d0 - 1], w0 = buf32[d0], w1 = buf32[d0 + 1];
                uint64_t vA, vB;
                uint32_t nA, nB;
                lit4(alignbit(w0, wp, 16), vA, nA);
                lit4(alignbit(w1, w0, 16), vB, nB);
                const uint32_t nb = nA + nB;
                const uint32_t incl = wave_incl_add(nb);
                const uint32_t pos = wbase + incl - nb;
                or_bits(pos, vA, nA);
                or_bits(pos + nA, vB, nB);
                wbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                k++;
                continue;
            }
            if (seg == SEG_FAST) {
                const int di = 64 * k + lane;
                uint64_t v;
                uint32_t nb;
                lit4(alignbit(buf32[di], buf32[di - 1], 16), v, nb);
                const uint32_t incl = wave_incl_add(nb);
                or_bits(wbase + incl - nb, v, nb);
                wbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                continue;
            }
            if (seg == SEG_MASK) {
                // an end slab without a run: the literals of the valid positions
                const int di = 64 * k + lane, ndw = (len + 3) >> 2;
                uint32_t w = 0, wp = 0;
                if (di < ndw) w = buf32[di];
                if (di - 1 < ndw) wp = buf32[di - 1];
                const uint32_t bytes = alignbit(w, wp, 16);
                const uint32_t V = frame_mask(256 * k + 4 * lane - 2, 0, len);
                uint32_t cc[4] = {S.code[bytes & 255u], S.code[(bytes >> 8) & 255u], S.code[(bytes >> 16) & 255u], S.code[bytes >> 24]};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (!(V & (0x80u << (8 * q)))) cc[q] = 0;
                const uint32_t n0 = cc[0] >> 16, n1 = cc[1] >> 16, n2 = cc[2] >> 16, n3 = cc[3] >> 16;
                const uint32_t lo = (cc[0] & 0xFFFFu) | ((cc[1] & 0xFFFFu) << n0), nlo = n0 + n1;
                const uint32_t hi = (cc[2] & 0xFFFFu) | ((cc[3] & 0xFFFFu) << n2), nhi = n2 + n3;
                const uint32_t nb = nlo + nhi;
                const uint32_t incl = wave_incl_add(nb);
                or_bits(wbase + incl - nb, (uint64_t)lo | ((uint64_t)hi << nlo), nb);
                wbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                continue;
            }
            if (seg != SEG_REDO && (list_ok || (seg & 0xFFFFu) < DEFL2_LIST_CAP)) {
                // listed: 64 tokens per step
                const uint32_t st = seg & 0xFFFFu, total = seg >> 16;
                const uint16_t *tl = st < DEFL2_LIST_CAP ? tla : tlb_top - (st - DEFL2_LIST_CAP) - total - st;
                for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                    const uint32_t idx = i0 + (uint32_t)lane;
                    uint32_t v = 0, nb = 0;
                    if (idx < total) {
                        const uint32_t tok = tl[st + idx];
                        const uint32_t sym = tok & 0x1FFu, cc = S.code[sym];
                        nb = cc >> 16;
                        v = (cc & 0xFFFFu) | ((tok >> 9) << nb);
                        nb += length_extra_bits(sym) + (sym > 256u ? dist_bits : 0u);   // (the distance code is all zeros)
                    }
                    const uint32_t incl = wave_incl_add(nb);
                    const uint32_t pos = wbase + incl - nb;
                    const uint32_t w = (pos >> 5) - z.flushed, sh = pos & 31;            // nb <= 25: two words
                    const uint64_t lo = (uint64_t)v << sh;
                    atomicOr(&obuf[w], (uint32_t)lo);
                    atomicOr(&obuf[w + 1], (uint32_t)(lo >> 32));
                    wbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                continue;
            }
            // the list was full in pass 1: the slab's analysis again (rare)
            const bool full = k > 0 && 256 * k + 256 <= len;
            uint32_t carryE = k > 0 ? carry_e_before(buf32, k, len) : 0u;
            WaveCarry wc{-1, k == 0};
            SlabCls c;
            if (full) c = classify_slab<false>(buf32, k, len, carryE);
            else c = classify_slab<true>(buf32, k, len, carryE);
            const bool anymem = __ballot(c.member != 0u) != 0ull;
            const uint32_t c0 = S.code[c.bytes & 255u], c1 = S.code[(c.bytes >> 8) & 255u], c2 = S.code[(c.bytes >> 16) & 255u], c3 = S.code[c.bytes >> 24];
            const uint32_t lit = c.V & ~c.member;
            SlabTail t;
            t.ql = 4; t.nfull = 0; t.rem = 0; t.runbyte = 0;
            if (anymem) t = slab_tail(buf, k, c, wc);
            // literals in front of the tail's slot (all of them without a tail), the tail, the literals behind it
            const uint32_t mA = t.ql >= 4 ? 0xFFFFFFFFu : (1u << (8 * t.ql)) - 1u;
            const uint32_t litA = lit & mA, litB = lit & ~mA;
            const uint32_t cc[4] = {c0, c1, c2, c3};
            uint64_t vA = 0, vB = 0;
            uint32_t nA = 0, nB = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t n = cc[q] >> 16, v = cc[q] & 0xFFFFu;
                if (litA & (0x80u << (8 * q))) { vA |= (uint64_t)v << nA; nA += n; }
                if (litB & (0x80u << (8 * q))) { vB |= (uint64_t)v << nB; nB += n; }
            }
            uint32_t nT = 0, tv = 0, tn = 0;      // tv / tn: the tail's last token (a match of rem bytes, or nothing)
            uint32_t c285 = 0, crun = 0;
            if (t.ql < 4) {
                c285 = S.code[285];
                crun = S.code[t.runbyte];
                nT = t.nfull * ((c285 >> 16) + dist_bits);
                if (t.rem >= 3) {
                    uint32_t sym, eb, ev;
                    length_symbol((int)t.rem, sym, eb, ev);
                    const uint32_t cs = S.code[sym];
                    tn = cs >> 16;
                    tv = (cs & 0xFFFFu) | (ev << tn);
                    tn += eb + dist_bits;                         // (the distance code is all zeros)
                    nT += tn;
                } else nT += t.rem * (crun >> 16);
            }
            const uint32_t nb = nA + nT + nB;
            const uint32_t incl = wave_incl_add(nb);
            uint32_t pos = wbase + incl - nb;
            wbase += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            or_bits(pos, vA, nA);
            pos += nA;
            if (t.ql < 4) {
                const uint32_t n285 = (c285 >> 16) + dist_bits;
                for (uint32_t i = 0; i < t.nfull; i++) { or_bits(pos, (uint64_t)(c285 & 0xFFFFu), n285); pos += n285; }
                if (t.rem >= 3) { or_bits(pos, (uint64_t)tv, tn); pos += tn; }
                else for (uint32_t i = 0; i < t.rem; i++) { or_bits(pos, (uint64_t)(crun & 0xFFFFu), crun >> 16); pos += crun >> 16; }
                or_bits(pos, vB, nB);
            }
        }
#ifdef S5_DEFL2_CHECK   // tests: every wave must end exactly where the next one starts
        {
            const uint32_t want = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)t_incl, wv);
            if (lane == 0 && wbase != want) atomicOr(&S.red[7], 1u);
        }
#endif
        if (dbg == 6) { z.bitpos += wbase; return; }
    }
    if (tid == 0) put_bits(obuf, z, pos0 + total_bits, eob & 0xFFFF, eob >> 16);
    z.bitpos = pos0 + total_bits + (eob >> 16);
    __syncthreads();
}

}  // namespace s5
