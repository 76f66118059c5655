// exzd_dev.h — ex-zd signal codec (slow5lib's SLOW5_COMPRESS_EX_ZD, /root/reference/src/misc.c:261; the default signal
// press of `slow5tools degrade`, src/degrade.c:302) for one read per workgroup.  SURVEY.md §8f row 4.
//
// Layout (pinned on the reference's fixtures, see oracle/exzd.c): u8 version 0 | u64 N | u8 q | u16 z0 | u32 nex |
// [u32 len | svb32(first position, then gap - 1) | u32 len | svb32(z - 256)] if nex | u8 z of every non-exception.
// q = trailing zero bits common to all samples, y = x >> q, z = zigzag(y[i] - y[i-1]), exception = z > 255 (i >= 1).
//
// Encode: three coalesced passes over the signal (OR-reduce for q; count exceptions and section sizes; write), because the
// one-byte array starts behind two sections whose sizes depend on the whole read.  Positions are compacted with workgroup
// prefix sums; the 2-bit StreamVByte keys of the (rare) exceptions go in with 32-bit atomic ORs.
// Decode: exceptions are decoded in aligned chunks of 4096 into LDS (absolute positions by a prefix sum of gap + 1); position
// tiles are bounded so that every exception of a tile is in the chunk at hand; inside a tile a 4096-bit flag map gives every
// position its rank among exceptions / non-exceptions, i.e. where its z comes from.
#pragma once
#include "dev_common.h"

namespace s5 {

constexpr int EXZD_TILE = NT * 16;            // positions (samples after the first) per tile, 16 per lane
constexpr uint32_t EXZD_ERR = 0xFFFFFFFFu;

struct ExzdScratch {                          // LDS of the decoder (also used by the encoder for its carries)
    uint32_t epos[EXZD_TILE];                 // absolute positions of the exception chunk at hand
    uint32_t eval[EXZD_TILE];                 // their z - 256
    uint32_t flag[EXZD_TILE / 32];            // tile flag map: bit p - p0 set <=> position p is an exception
    uint32_t red[8];
};

__device__ __forceinline__ uint32_t svb32_len(uint32_t v) { return 1u + (v > 0xFFu) + (v > 0xFFFFu) + (v > 0xFFFFFFu); }
__device__ __forceinline__ uint32_t zigzag32(int d) { return ((uint32_t)d << 1) ^ (uint32_t)(d >> 31); }

// OR a 2-bit key code into key byte (e >> 2) of a byte-aligned HBM key area that was zeroed before
__device__ __forceinline__ void svb_key_or(uint8_t *keys, uint32_t e, uint32_t code) {
    uint8_t *kb = keys + (e >> 2);
    const uintptr_t a = reinterpret_cast<uintptr_t>(kb);
    uint32_t *w = reinterpret_cast<uint32_t *>(a & ~(uintptr_t)3);
    atomicOr(w, (code << (2 * (e & 3))) << (8 * (uint32_t)(a & 3)));
}
__device__ __forceinline__ void put_le(uint8_t *p, uint32_t v, uint32_t nbytes) {
    for (uint32_t b = 0; b < nbytes; b++) p[b] = (uint8_t)(v >> (8 * b));
}

// ---- encode: signal -> blob at dst (HBM, any alignment; dst - 3 .. dst + bound must be writable for the key ORs' aligned
// words — the payload's 8-byte length field sits right in front).  Returns the blob length (uniform).
// room: bytes dst can take; if the blob needs more, EXZD_ERR comes back (uniform) with at most the 10 header bytes written.
__device__ __forceinline__ uint32_t exzd_encode_wg(const int16_t *__restrict__ x, uint32_t n, uint8_t *dst, uint32_t *ws, uint32_t *red,
                                                   uint32_t room = 0xFFFFFFF0u) {
    const int tid = threadIdx.x;
    // pass 1: q
    uint32_t acc = 0;
    for (uint32_t i = tid; i < n; i += NT) acc |= (uint16_t)x[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc |= __shfl_xor(acc, d);
    if (tid < 8) red[tid] = 0;
    __syncthreads();
    if (lane_id() == 0) atomicOr(&red[0], acc);
    __syncthreads();
    acc = red[0];
    const uint32_t q = acc ? (uint32_t)__ffs((int)acc) - 1 : 0u;
    if (room < 16) return EXZD_ERR;
    if (tid == 0) {
        dst[0] = 0;
        for (int b = 0; b < 8; b++) dst[1 + b] = b < 4 ? (uint8_t)(n >> (8 * b)) : 0;
        dst[9] = (uint8_t)q;
    }
    if (n == 0) return 10;
    const uint32_t np = n - 1;
    // per-tile worker: z of my 16 positions, exception mask; shared by the two passes below
    auto load_tile = [&](uint32_t t0, uint32_t (&z)[16], uint32_t &exm, int &valid) {
        const uint32_t p0 = t0 + 16u * tid;
        valid = p0 >= np ? 0 : (int)min(16u, np - p0);
        exm = 0;
        int prev = valid > 0 ? ((int)x[p0] >> q) : 0;             // sample p0 is position p0's predecessor
#pragma unroll
        for (int k = 0; k < 16; k++) {
            z[k] = 0;
            if (k < valid) {
                const int y = (int)x[p0 + 1 + k] >> q;
                z[k] = zigzag32(y - prev);
                prev = y;
                if (z[k] > 255u) exm |= 1u << k;
            }
        }
    };
    // pass 2a: number of exceptions, data bytes of the two StreamVByte sections
    uint32_t nex = 0, pos_bytes = 0, val_bytes = 0;
    int last_carry = -1;                                           // position of the last exception of the tiles before
    for (uint32_t t0 = 0; t0 < np; t0 += EXZD_TILE) {
        uint32_t z[16], exm;
        int valid;
        load_tile(t0, z, exm, valid);
        const int p0 = (int)(t0 + 16u * tid);
        const int local_last = exm ? p0 + 31 - __clz((int)exm) : -1;
        int lastp = max(block_excl_max(local_last, -1, ws), last_carry);
        if (exm) {                                                 // static indices only: z[] must stay in registers
#pragma unroll
            for (int k = 0; k < 16; k++)
                if ((exm >> k) & 1u) {
                    pos_bytes += svb32_len((uint32_t)(p0 + k - lastp - 1));
                    val_bytes += svb32_len(z[k] - 256u);
                    lastp = p0 + k;
                    nex++;
                }
        }
        // the tile's last exception position, for the next tile
        __syncthreads();
        if (tid == 0) red[1] = 0;
        __syncthreads();
        if (exm) atomicMax(&red[1], (uint32_t)(local_last + 1));
        __syncthreads();
        if (red[1]) last_carry = (int)red[1] - 1;
        __syncthreads();
    }
    nex = wave_sum(nex); pos_bytes = wave_sum(pos_bytes); val_bytes = wave_sum(val_bytes);
    if (lane_id() == 0) { atomicAdd(&red[2], nex); atomicAdd(&red[3], pos_bytes); atomicAdd(&red[4], val_bytes); }
    __syncthreads();
    nex = red[2]; pos_bytes = red[3]; val_bytes = red[4];
    const uint32_t nk = (nex + 3) >> 2;
    const uint32_t pos_sec = nk + pos_bytes, val_sec = nk + val_bytes;
    const uint32_t o_pos = 20, o_val = o_pos + pos_sec + 4, o_rest = nex ? o_val + val_sec : 16;
    if ((uint64_t)o_rest + (np - nex) + 4 > room) return EXZD_ERR;       // + 4: the key ORs touch whole aligned words
    if (tid == 0) {
        const int y0 = (int)x[0] >> q;
        put_le(dst + 10, zigzag32(y0) & 0xFFFFu, 2);
        put_le(dst + 12, nex, 4);
        if (nex) { put_le(dst + 16, pos_sec, 4); put_le(dst + o_val - 4, val_sec, 4); }
    }
    uint8_t *pkeys = dst + o_pos, *pdata = pkeys + nk, *vkeys = dst + o_val, *vdata = vkeys + nk, *rest = dst + o_rest;
    if (nex) {
        for (uint32_t i = tid; i < nk; i += NT) { pkeys[i] = 0; vkeys[i] = 0; }
        // Workgroup scope is enough: the ORs below come from this workgroup and meet the zeroes in the same XCD's L2.  (An
        // agent-scope fence writes the whole L2 back on this multi-XCD part: measured 12x slower for the kernel.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __syncthreads();
    // pass 2b: write
    uint32_t e_run = 0, pd_run = 0, vd_run = 0, rb_run = 0;
    last_carry = -1;
    for (uint32_t t0 = 0; t0 < np; t0 += EXZD_TILE) {
        uint32_t z[16], exm;
        int valid;
        load_tile(t0, z, exm, valid);
        const int p0 = (int)(t0 + 16u * tid);
        const int local_last = exm ? p0 + 31 - __clz((int)exm) : -1;
        int lastp = max(block_excl_max(local_last, -1, ws), last_carry);
        // my section sizes first (same walk as pass 2a), then one scan per quantity
        uint32_t my_ex = (uint32_t)__popc(exm), my_pb = 0, my_vb = 0;
        if (exm) {
            int lp = lastp;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if ((exm >> k) & 1u) {
                    my_pb += svb32_len((uint32_t)(p0 + k - lp - 1));
                    my_vb += svb32_len(z[k] - 256u);
                    lp = p0 + k;
                }
        }
        uint32_t tot_ex, tot_pb, tot_vb;
        uint32_t e = e_run + block_excl_add(my_ex, ws, tot_ex);
        uint32_t pd = pd_run + block_excl_add(my_pb, ws, tot_pb);
        uint32_t vd = vd_run + block_excl_add(my_vb, ws, tot_vb);
        uint32_t rb = rb_run + (uint32_t)(16 * tid) - (e - e_run);   // positions before me in the tile minus exceptions before me
        // kept as a rolled loop on purpose: fully unrolled it pushes the fused kernels (64-VGPR budget, 8 workgroups per CU)
        // into 300-1000 bytes of scratch per lane — measured 2.8x slower for the whole encode
#pragma nounroll
        for (int k = 0; k < 16; k++) {
            if (k >= valid) break;
            if ((exm >> k) & 1u) {
                const uint32_t gap = (uint32_t)(p0 + k - lastp - 1), v = z[k] - 256u;
                const uint32_t lg = svb32_len(gap), lv = svb32_len(v);
                svb_key_or(pkeys, e, lg - 1);
                svb_key_or(vkeys, e, lv - 1);
                put_le(pdata + pd, gap, lg);
                put_le(vdata + vd, v, lv);
                pd += lg; vd += lv; e++;
                lastp = p0 + k;
            } else {
                rest[rb++] = (uint8_t)z[k];
            }
        }
        const uint32_t tile_pos = min((uint32_t)EXZD_TILE, np - t0);
        e_run += tot_ex; pd_run += tot_pb; vd_run += tot_vb; rb_run += tile_pos - tot_ex;
        __syncthreads();
        if (tid == 0) red[1] = 0;
        __syncthreads();
        if (exm) atomicMax(&red[1], (uint32_t)(local_last + 1));
        __syncthreads();
        if (red[1]) last_carry = (int)red[1] - 1;
        __syncthreads();
    }
    return o_rest + (np - nex);
}

// ---- decode ----
// one aligned chunk of a StreamVByte-32 list: entries [c0, c0 + cnt), c0 a multiple of 4096, cnt <= 4096; keys = key area
// of the list; data = its first unread data byte.  Values go to LDS vals[0 .. cnt).  Returns the data bytes consumed.
__device__ __forceinline__ uint32_t svb32_decode_chunk(const uint8_t *keys, uint32_t c0, uint32_t cnt, const uint8_t *data,
                                                       const uint8_t *data_end, uint32_t *vals, uint32_t *ws, int &err) {
    const int tid = threadIdx.x;
    const uint32_t i0 = 16u * tid;
    const int valid = i0 >= cnt ? 0 : (int)min(16u, cnt - i0);
    const int nkb = (valid + 3) >> 2;
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k < nkb) key |= (uint32_t)keys[(c0 >> 2) + 4 * tid + k] << (8 * k);
    uint32_t nbytes = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < valid) nbytes += ((key >> (2 * k)) & 3) + 1;
    uint32_t total;
    const uint32_t off = block_excl_add(nbytes, ws, total);
    const uint8_t *dp = data + off;
    const bool ok = !(valid > 0 && dp + nbytes > data_end);
    if (!ok) err = 1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < valid) {
            uint32_t v = 0;
            if (ok) {
                const uint32_t code = (key >> (2 * k)) & 3;
                for (uint32_t b = 0; b <= code; b++) v |= (uint32_t)dp[b] << (8 * b);
                dp += code + 1;
            }
            vals[i0 + k] = v;
        }
    }
    return total;
}

// blob (HBM) -> int16 samples.  Returns 0 ok, 6 output too small (*n_out = samples needed), 7 malformed.
__device__ __forceinline__ int exzd_decode_wg(const uint8_t *blob, uint64_t L, int16_t *__restrict__ out, uint32_t cap, uint32_t *n_out,
                                              ExzdScratch &X, uint32_t *ws) {
    const int tid = threadIdx.x;
    *n_out = 0;
    if (L < 10 || blob[0] != 0) return 7;
    uint64_t n64 = 0;
    for (int b = 0; b < 8; b++) n64 |= (uint64_t)blob[1 + b] << (8 * b);
    const uint32_t q = blob[9];
    if (n64 == 0) return L == 10 ? 0 : 7;
    if (n64 > 0xFFFFFFF0ull || q > 15 || L < 16) return 7;
    const uint32_t n = (uint32_t)n64, np = n - 1;
    *n_out = n;
    if (n > cap) return 6;
    const uint32_t z0 = (uint32_t)blob[10] | ((uint32_t)blob[11] << 8);
    uint32_t nex = 0;
    for (int b = 0; b < 4; b++) nex |= (uint32_t)blob[12 + b] << (8 * b);
    if (nex > np) return 7;
    const uint32_t nk = (nex + 3) >> 2;
    const uint8_t *pkeys = nullptr, *pdata = nullptr, *pend = nullptr, *vkeys = nullptr, *vdata = nullptr, *vend = nullptr;
    uint64_t at = 16;
    if (nex) {
        uint32_t sl = 0;
        if (at + 4 > L) return 7;
        for (int b = 0; b < 4; b++) sl |= (uint32_t)blob[at + b] << (8 * b);
        at += 4;
        if (sl < nk || at + sl > L) return 7;
        pkeys = blob + at; pdata = pkeys + nk; pend = pkeys + sl;
        at += sl;
        if (at + 4 > L) return 7;
        sl = 0;
        for (int b = 0; b < 4; b++) sl |= (uint32_t)blob[at + b] << (8 * b);
        at += 4;
        if (sl < nk || at + sl > L) return 7;
        vkeys = blob + at; vdata = vkeys + nk; vend = vkeys + sl;
        at += sl;
    }
    if (L - at != (uint64_t)np - nex) return 7;
    const uint8_t *rest = blob + at;
    int y0 = (int)(z0 >> 1) ^ -(int)(z0 & 1);
    if (tid == 0) out[0] = (int16_t)((uint32_t)y0 << q);
    int ycarry = y0, err = 0;
    uint32_t p0 = 0, e0 = 0, rb = 0;            // next position, next exception entry, next byte of `rest`
    uint32_t cb = 0, cc = 0;                    // exception chunk at hand: entries [cb, cb + cc)
    uint32_t pd = 0, vd = 0, acarry = 0;        // data bytes consumed of the two lists; last exception position + 1
    while (p0 < np) {
        if (e0 == cb + cc && e0 < nex) {        // all entries of the chunk used: decode the next one
            cb = e0;
            cc = min((uint32_t)EXZD_TILE, nex - cb);
            __syncthreads();
            pd += svb32_decode_chunk(pkeys, cb, cc, pdata + pd, pend, X.epos, ws, err);
            vd += svb32_decode_chunk(vkeys, cb, cc, vdata + vd, vend, X.eval, ws, err);
            __syncthreads();
            // gaps -> absolute positions: abs_e = acarry + sum_{j <= e} (gap_j + 1) - 1
            uint32_t g[16], s = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) { g[k] = 16u * tid + k < cc ? X.epos[16 * tid + k] + 1u : 0u; s += g[k]; }
            uint32_t tot;
            uint32_t run = acarry + block_excl_add(s, ws, tot);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                run += g[k];
                if (16u * tid + k < cc) X.epos[16 * tid + k] = run - 1u;
            }
            acarry += tot;
            __syncthreads();
            if (X.epos[cc - 1] >= np) { err = 1; break; }            // positions must stay inside the signal (uniform read: uniform break)
        }
        // tile [p0, pe): at most EXZD_TILE positions, and no further than the last exception of the chunk when more follow
        uint32_t pe = min(p0 + (uint32_t)EXZD_TILE, np);
        if (cb + cc < nex) pe = min(pe, X.epos[cc - 1] + 1u);
        if (pe <= p0) { err = 1; break; }                             // cannot happen for a well-formed list
        for (int i = tid; i < EXZD_TILE / 32; i += NT) X.flag[i] = 0;
        __syncthreads();
        uint32_t mine = 0;                                           // chunk entries inside the tile
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t ci = 16u * tid + k;
            if (ci < cc && cb + ci >= e0) {
                const uint32_t p = X.epos[ci];
                if (p >= p0 && p < pe) { atomicOr(&X.flag[(p - p0) >> 5], 1u << ((p - p0) & 31)); mine++; }
            }
        }
        uint32_t in_tile;
        block_excl_add(mine, ws, in_tile);                           // includes the barrier that publishes the flags
        const uint32_t lp = 16u * tid;                               // my first position, tile-relative
        const int valid = p0 + lp >= pe ? 0 : (int)min(16u, pe - p0 - lp);
        uint32_t fl = (X.flag[lp >> 5] >> (lp & 31)) & 0xFFFFu;
        if (valid < 16) fl &= (1u << valid) - 1u;
        uint32_t dummy;
        const uint32_t exb = block_excl_add((uint32_t)__popc(fl), ws, dummy);   // exceptions before my first position
        if (dummy != in_tile) err = 1;                               // two exceptions on one position
        int d[16], sum = 0;
        uint32_t ei = e0 + exb - cb, ri = rb + lp - exb;
        const uint32_t rest_len = np - nex;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            d[k] = 0;
            if (k < valid) {
                uint32_t z;
                if ((fl >> k) & 1u) z = X.eval[ei++] + 256u;
                else { z = ri < rest_len ? rest[ri] : 0u; if (ri >= rest_len) err = 1; ri++; }
                d[k] = (int)(z >> 1) ^ -(int)(z & 1);
            }
            sum += d[k];
        }
        uint32_t tsum;
        int y = ycarry + (int)block_excl_add((uint32_t)sum, ws, tsum);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < valid) { y += d[k]; out[1 + p0 + lp + k] = (int16_t)((uint32_t)y << q); }
        ycarry += (int)tsum;
        rb += (pe - p0) - in_tile;
        e0 += in_tile;
        p0 = pe;
        __syncthreads();
    }
    // errors are per lane: make them uniform
    __syncthreads();
    if (tid == 0) X.red[0] = 0;
    __syncthreads();
    if (err) atomicOr(&X.red[0], 1u);
    __syncthreads();
    if (X.red[0] || e0 != nex || rb != np - nex) return 7;
    return 0;
}

// ---- decode by ONE wave (round 3): the wave that inflated a record decodes its ex-zd blob too (k_inflate_par<2>), as it does for
// svb-zd — the blob it reads is the payload it just wrote.  Same scheme as exzd_decode_wg on a quarter of the scale: exception chunks
// of 512 entries (8 per lane), position tiles of 1024 (16 per lane), a 1024-bit flag map; the scratch is the inflate's own LDS, dead
// by then.  Output: sample i = 1 + position.  A lane whose 16 positions hold no exception (most lanes: 1.5 % of the deltas of a nanopore
// signal are exceptions) takes its 16 bytes with one load and stores index [p0 + lp, p0 + lp + 16) as two 16-byte words — its
// carry-in (the sample in front of its first) and its first 15 samples; a lane's 16th sample is the next lane's carry-in, the very
// last sample is stored behind the loop.
constexpr uint32_t EXZD_WCHUNK = 512, EXZD_WTILE = 1024;
struct ExzdWaveScratch {
    uint32_t epos[EXZD_WCHUNK];
    uint32_t eval[EXZD_WCHUNK];
    uint32_t flag[EXZD_WTILE / 32];
};
__device__ __forceinline__ uint32_t svb32_decode_chunk_wave(const uint8_t *keys, uint32_t c0, uint32_t cnt, const uint8_t *data,
                                                            const uint8_t *data_end, uint32_t *vals, int &err) {
    const int lane = lane_id();
    const uint32_t i0 = 8u * (uint32_t)lane;
    const int valid = i0 >= cnt ? 0 : (int)min(8u, cnt - i0);
    const int nkb = (valid + 3) >> 2;
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (k < nkb) key |= (uint32_t)keys[(c0 >> 2) + 2 * lane + k] << (8 * k);
    uint32_t nbytes = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < valid) nbytes += ((key >> (2 * k)) & 3) + 1;
    const uint32_t incl = wave_incl_add(nbytes);
    const uint8_t *dp = data + (incl - nbytes);
    const bool ok = !(valid > 0 && dp + nbytes > data_end);
    if (!ok) err = 1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < valid) {
            uint32_t v = 0;
            if (ok) {
                const uint32_t code = (key >> (2 * k)) & 3;
                for (uint32_t b = 0; b <= code; b++) v |= (uint32_t)dp[b] << (8 * b);
                dp += code + 1;
            }
            vals[i0 + k] = v;
        }
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}
// blob (the wave's own payload bytes) -> int16 samples at out (16-byte aligned).  Returns 0 ok, 6 output too small (*n_out = samples
// needed), 7 malformed; uniform.  Only lane 0's *n_out store matters to the caller (every lane passes the same pointer).
__device__ __forceinline__ int exzd_decode_wave(const uint8_t *blob, uint64_t L, int16_t *__restrict__ out, uint32_t cap, uint32_t &n_out,
                                                ExzdWaveScratch &X) {
    const int lane = lane_id();
    n_out = 0;
    if (L < 10 || blob[0] != 0) return 7;
    uint64_t n64 = 0;
    for (int b = 0; b < 8; b++) n64 |= (uint64_t)blob[1 + b] << (8 * b);
    const uint32_t q = blob[9];
    if (n64 == 0) return L == 10 ? 0 : 7;
    if (n64 > 0xFFFFFFF0ull || q > 15 || L < 16) return 7;
    const uint32_t n = (uint32_t)n64, np = n - 1;
    n_out = n;
    if (n > cap) return 6;
    const uint32_t z0 = (uint32_t)blob[10] | ((uint32_t)blob[11] << 8);
    uint32_t nex = 0;
    for (int b = 0; b < 4; b++) nex |= (uint32_t)blob[12 + b] << (8 * b);
    if (nex > np) return 7;
    const uint32_t nk = (nex + 3) >> 2;
    const uint8_t *pkeys = nullptr, *pdata = nullptr, *pend = nullptr, *vkeys = nullptr, *vdata = nullptr, *vend = nullptr;
    uint64_t at = 16;
    if (nex) {
        uint32_t sl = 0;
        if (at + 4 > L) return 7;
        for (int b = 0; b < 4; b++) sl |= (uint32_t)blob[at + b] << (8 * b);
        at += 4;
        if (sl < nk || at + sl > L) return 7;
        pkeys = blob + at; pdata = pkeys + nk; pend = pkeys + sl;
        at += sl;
        if (at + 4 > L) return 7;
        sl = 0;
        for (int b = 0; b < 4; b++) sl |= (uint32_t)blob[at + b] << (8 * b);
        at += 4;
        if (sl < nk || at + sl > L) return 7;
        vkeys = blob + at; vdata = vkeys + nk; vend = vkeys + sl;
        at += sl;
    }
    if (L - at != (uint64_t)np - nex) return 7;
    const uint8_t *rest = blob + at;
    const uint32_t rest_len = np - nex;
    const int y0 = (int)(z0 >> 1) ^ -(int)(z0 & 1);
    int ycarry = y0, err = 0;
    uint32_t p0 = 0, e0 = 0, rb = 0;            // next position, next exception entry, next byte of `rest`
    uint32_t cb = 0, cc = 0;                    // exception chunk at hand: entries [cb, cb + cc)
    uint32_t pd = 0, vd = 0, acarry = 0;        // data bytes consumed of the two lists; last exception position + 1
    bool broke = false;
    while (p0 < np) {
        if (e0 == cb + cc && e0 < nex) {        // all entries of the chunk used: decode the next one
            cb = e0;
            cc = min(EXZD_WCHUNK, nex - cb);
            wave_sync();
            pd += svb32_decode_chunk_wave(pkeys, cb, cc, pdata + pd, pend, X.epos, err);
            vd += svb32_decode_chunk_wave(vkeys, cb, cc, vdata + vd, vend, X.eval, err);
            wave_sync();
            uint32_t g[8], sum = 0;             // gaps -> absolute positions: abs_e = acarry + sum_{j <= e} (gap_j + 1) - 1
#pragma unroll
            for (int k = 0; k < 8; k++) { g[k] = 8u * lane + k < cc ? X.epos[8 * lane + k] + 1u : 0u; sum += g[k]; }
            const uint32_t incl = wave_incl_add(sum);
            uint32_t run = acarry + incl - sum;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                run += g[k];
                if (8u * lane + k < cc) X.epos[8 * lane + k] = run - 1u;
            }
            acarry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            wave_sync();
            if (X.epos[cc - 1] >= np) { broke = true; break; }        // positions must stay inside the signal (uniform read)
        }
        // tile [p0, pe): at most EXZD_WTILE positions, and no further than the last exception of the chunk when more follow
        uint32_t pe = min(p0 + EXZD_WTILE, np);
        if (cb + cc < nex) pe = min(pe, X.epos[cc - 1] + 1u);
        if (pe <= p0) { broke = true; break; }                        // cannot happen for a well-formed list
        if (lane < (int)(EXZD_WTILE / 32)) X.flag[lane] = 0;
        wave_sync();
        uint32_t mine = 0;                                            // chunk entries inside the tile
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t ci = 8u * lane + k;
            if (ci < cc && cb + ci >= e0) {
                const uint32_t p = X.epos[ci];
                if (p >= p0 && p < pe) { atomicOr(&X.flag[(p - p0) >> 5], 1u << ((p - p0) & 31)); mine++; }
            }
        }
        const uint32_t in_tile = wave_sum(mine);
        wave_sync();
        const uint32_t lp = 16u * (uint32_t)lane;                     // my first position, tile-relative
        const int valid = p0 + lp >= pe ? 0 : (int)min(16u, pe - p0 - lp);
        uint32_t fl = (X.flag[lp >> 5] >> (lp & 31)) & 0xFFFFu;
        if (valid < 16) fl &= (1u << valid) - 1u;
        const uint32_t pc = (uint32_t)__popc(fl);
        const uint32_t pincl = wave_incl_add(pc);
        const uint32_t exb = pincl - pc;                              // exceptions before my first position
        if ((uint32_t)__builtin_amdgcn_readlane((int)pincl, 63) != in_tile) err = 1;   // two exceptions on one position
        int d[16], sum = 0;
        const uint32_t ri0 = rb + lp - exb;
        if (valid == 16 && fl == 0 && ri0 + 16 <= rest_len) {         // sixteen plain bytes
            typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(1)));
            const v4u w = *reinterpret_cast<const v4u *>(rest + ri0);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t z = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                d[k] = (int)(z >> 1) ^ -(int)(z & 1);
                sum += d[k];
            }
        } else {
            uint32_t ei = e0 + exb - cb, ri = ri0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                d[k] = 0;
                if (k < valid) {
                    uint32_t z;
                    if ((fl >> k) & 1u) z = X.eval[ei++] + 256u;
                    else { z = ri < rest_len ? rest[ri] : 0u; if (ri >= rest_len) err = 1; ri++; }
                    d[k] = (int)(z >> 1) ^ -(int)(z & 1);
                }
                sum += d[k];
            }
        }
        const uint32_t sincl = wave_incl_add((uint32_t)sum);
        int y = ycarry + (int)(sincl - (uint32_t)sum);                // the sample in front of my first
        int16_t *o = out + p0 + lp;
        if (valid == 16 && ((p0 + lp) & 7u) == 0) {
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t lo = ((uint32_t)y << q) & 0xFFFFu;
                y += d[2 * k];
                w[k] = lo | (((uint32_t)y << q) << 16);
                y += d[2 * k + 1];
            }
            uint4 *o4 = reinterpret_cast<uint4 *>(o);
            o4[0] = make_uint4(w[0], w[1], w[2], w[3]);
            o4[1] = make_uint4(w[4], w[5], w[6], w[7]);
        } else if (valid > 0) {
            o[0] = (int16_t)((uint32_t)y << q);
#pragma unroll
            for (int k = 0; k < 15; k++)
                if (k + 1 < valid) { y += d[k]; o[1 + k] = (int16_t)((uint32_t)y << q); }
        }
        ycarry += (int)(uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
        rb += (pe - p0) - in_tile;
        e0 += in_tile;
        p0 = pe;
        wave_sync();
    }
    if (lane == 0 && !broke) out[np] = (int16_t)((uint32_t)ycarry << q);   // the last sample (sample 0 when there is no other)
    if (broke || __ballot(err != 0) || e0 != nex || rb != np - nex) return 7;
    return 0;
}

}  // namespace s5
