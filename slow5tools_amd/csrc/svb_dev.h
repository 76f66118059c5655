// svb_dev.h — svb-zd (StreamVByte-32 + zigzag delta) encode / decode tiles for one workgroup.
//
// Replaces slow5lib's signal press (thirdparty/streamvbyte, absent submodule) as reached from
// slow5_rec_to_mem / slow5_rec_depress_parse (/root/reference/src/view.c:38,49).  Bit layout:
// SURVEY.md Appendix A.3 — u32 N | ceil(N/4) key bytes (2 b/value, LSB first) | 1..4 LE bytes/value
// of z = zigzag32(x[i] - x[i-1]), x[-1] = 0.
//
// One read per workgroup, tiles of 4096 samples: lane t owns 16 consecutive samples = two 16-byte
// coalesced loads = exactly 4 key bytes; data offsets by wave prefix scan + 4-entry cross-wave scan.
#pragma once
#include "dev_common.h"

namespace s5 {

constexpr int SVB_TILE = NT * 16;   // samples per tile

// Encode samples [t0, min(t0+SVB_TILE, n)) of one read.  sig: read base (16-B aligned).
// keys: destination of this tile's key bytes (= key area + t0/4); data: destination of this tile's
// first data byte.  Destinations may be LDS or HBM.  Returns the tile's data byte count (uniform).
// room: data bytes the destination can still take; if the tile needs more, nothing is written and the
// (uniform) return value exceeds room — the caller routes the read to the HBM-staged path.
// In two halves, so that a caller can learn the sizes of several tiles before it writes any of them (k_svbzd_stream): what a lane
// holds of a tile between the two is its 16 zig-zag deltas, its 4 key bytes and its byte count.
struct SvbTileLane {
    uint32_t z[16];
    uint32_t key, nbytes;
    int valid;
};
#ifndef S5_SVB_PACKED
#define S5_SVB_PACKED 1
#endif
__device__ __forceinline__ void svb_tile_classify(const int16_t *__restrict__ sig, uint32_t n, uint32_t t0, SvbTileLane &T) {
    const int tid = threadIdx.x;
    const uint32_t i0 = t0 + 16u * tid;
    const int valid = i0 >= n ? 0 : (int)min(16u, n - i0);
#if S5_SVB_PACKED
    // Round 5: two samples per instruction.  A nanopore signal lives in a few thousand ADC levels; when every sample a wave holds of this tile
    // (and the one in front of each lane) lies in [-16384, 16384), every delta fits 16 bits and so does its zig-zag value: the packed 16-bit
    // forms of subtract / shift / min do two samples each, the key bits come out of two accumulators, the byte count is a population count (a
    // value takes one byte or two).  Lanes without samples ride along on zeros.  Anything else — a sample outside that range, a lane with 1 .. 15
    // samples — sends the whole wave down the general path below: same results, bit for bit (tests/test_gpu_parity.py, tests/test_full_size.py).
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        typedef short ss2 __attribute__((ext_vector_type(2)));
        auto U = [](uint32_t v) { return __builtin_bit_cast(us2, v); };
        auto W = [](us2 v) { return __builtin_bit_cast(uint32_t, v); };
        uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, pw = 0u;
        bool ok = valid == 0;
        if (valid == 16) {
            const uint4 a = *reinterpret_cast<const uint4 *>(sig + i0);
            const uint4 b = *reinterpret_cast<const uint4 *>(sig + i0 + 8);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
            if (i0 > 0) pw = *reinterpret_cast<const uint32_t *>(sig + i0 - 2);   // (i0 is a multiple of 16: aligned; its high half is the sample in front)
            uint32_t rng = W(U(pw) + us2{0x4000, 0x4000});
#pragma unroll
            for (int j = 0; j < 8; j++) rng |= W(U(w[j]) + us2{0x4000, 0x4000});
            ok = (rng & 0x80008000u) == 0u;
        }
        if (__ballot(!ok) == 0ull) {                       // (uniform)
            uint32_t A = 0, B = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t xp = __builtin_amdgcn_alignbit(w[j], j ? w[j - 1] : pw, 16);       // the samples in front of the pair's two
                const ss2 d = __builtin_bit_cast(ss2, w[j]) - __builtin_bit_cast(ss2, xp);
                const us2 z = __builtin_bit_cast(us2, d) << us2{1, 1} ^ __builtin_bit_cast(us2, d >> ss2{15, 15});
                const us2 c = __builtin_elementwise_min(z >> us2{8, 8}, us2{1, 1});               // 0: one byte, 1: two
                T.z[2 * j] = (uint32_t)z.x;
                T.z[2 * j + 1] = (uint32_t)z.y;
                if (j < 4) A |= W(c) << (4 * j); else B |= W(c) << (4 * (j - 4));              // bit 4 j: sample 2 j, bit 16 + 4 j: sample 2 j + 1
            }
            const uint32_t key = ((A | (A >> 14)) & 0xFFFFu) | ((B | (B >> 14)) << 16);
            T.key = key;
            T.nbytes = valid == 16 ? 16u + (uint32_t)__popc(key) : 0u;
            T.valid = valid;
            return;
        }
    }
#endif
    int x[16];
    int prev = 0;
    if (valid == 16) {
        const int4 a = *reinterpret_cast<const int4 *>(sig + i0);
        const int4 b = *reinterpret_cast<const int4 *>(sig + i0 + 8);
        const int w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            x[2 * q] = (int)(short)(w[q] & 0xFFFF);
            x[2 * q + 1] = w[q] >> 16;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = q < valid ? (int)sig[i0 + q] : 0;
    }
    if (valid > 0 && i0 > 0) prev = sig[i0 - 1];
    uint32_t key = 0, nbytes = 0;
    // A lane is almost always full (16 samples) or empty; the full case runs without per-sample predicates.
    const bool full = valid == 16;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int d = x[q] - prev;
        prev = x[q];
        T.z[q] = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
    }
    if (full) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t code = (T.z[q] > 0xFFu ? 1u : 0u) + (T.z[q] >> 16);   // int16 input: z <= 131070, so z >> 16 is 0 or 1
            key |= code << (2 * q);
            nbytes += code;
        }
        nbytes += 16;
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t code = (T.z[q] > 0xFFu) + (T.z[q] > 0xFFFFu);
            if (q < valid) {
                key |= code << (2 * q);
                nbytes += code + 1;
            }
        }
    }
    T.key = key;
    T.nbytes = nbytes;
    T.valid = valid;
}
// dp: the lane's first data byte (data + its exclusive byte offset in the tile)
__device__ __forceinline__ void svb_tile_write(const SvbTileLane &T, uint8_t *keys, uint8_t *dp) {
    const int tid = threadIdx.x;
    if (T.valid == 16) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            *dp++ = (uint8_t)T.z[q];
            if (T.z[q] > 0xFFu) {                        // 1.5 % of the samples of a nanopore signal
                *dp++ = (uint8_t)(T.z[q] >> 8);
                if (T.z[q] > 0xFFFFu) *dp++ = (uint8_t)(T.z[q] >> 16);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (q < T.valid) {
                *dp++ = (uint8_t)T.z[q];
                if (T.z[q] > 0xFFu) *dp++ = (uint8_t)(T.z[q] >> 8);
                if (T.z[q] > 0xFFFFu) *dp++ = (uint8_t)(T.z[q] >> 16);
            }
        }
    }
    const int nk = (T.valid + 3) >> 2;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < nk) keys[4 * tid + q] = (uint8_t)(T.key >> (8 * q));
}
// ws: 16 words of scratch.  ws_fresh: nobody has used ws since the workgroup's last barrier (the fused kernels: one read per workgroup) —
// otherwise a read's first tile waits for the readers of the previous read's last scan.
__device__ __forceinline__ uint32_t svb_encode_tile(const int16_t *__restrict__ sig, uint32_t n, uint32_t t0,
                                                    uint8_t *keys, uint8_t *data, uint32_t *ws, uint32_t room, bool ws_fresh = false) {
    SvbTileLane T;
    svb_tile_classify(sig, n, t0, T);
    if (t0 == 0 && !ws_fresh) __syncthreads();
    uint32_t total;
    const uint32_t off = block_excl_add_alt(T.nbytes, ws, total, t0 / (uint32_t)SVB_TILE);   // (ws: 16 words; one barrier per tile)
    if (total > room) return total;
    svb_tile_write(T, keys, data + off);
    return total;
}

// Decode values [t0, min(t0+SVB_TILE, n)) of one svb-zd blob.  keys: key area base + t0/4;
// data: first data byte of this tile; data_end: end of blob.  `carry` = x[t0-1] (0 for the first tile).
// Writes int16 samples to out[t0..].  Returns the tile's data byte count; updates carry (uniform).
// err is set (not cleared) if a value would read past data_end.
// stage: LDS, SVB_STAGE bytes, 16-byte aligned.  The data bytes of the tile are brought in with 16-byte loads and the lanes
// pick their bytes out of LDS: byte-granular loads straight from HBM cost ~20 vector-memory instructions per lane and tile.
constexpr uint32_t SVB_STAGE = 4u * SVB_TILE + 16u;
__device__ __forceinline__ uint32_t svb_decode_tile(const uint8_t *keys, const uint8_t *data, const uint8_t *data_end,
                                                    uint32_t n, uint32_t t0, int16_t *__restrict__ out, int &carry,
                                                    int &err, uint32_t *ws, uint8_t *stage) {
    const int tid = threadIdx.x;
    const uint32_t i0 = t0 + 16u * tid;
    const int valid = i0 >= n ? 0 : (int)min(16u, n - i0);
    const int nk = (valid + 3) >> 2;
    uint32_t key = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < nk) key |= (uint32_t)keys[4 * tid + q] << (8 * q);
    uint32_t nbytes = 0;
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (q < valid) nbytes += ((key >> (2 * q)) & 3) + 1;
    uint32_t total;
    const uint32_t off = block_excl_add(nbytes, ws, total);
    {   // data[0 .. min(total, bytes left)) -> stage (total <= 4 * SVB_TILE)
        const uint32_t have = (uint32_t)min((uint64_t)total, (uint64_t)(data_end - data));
        typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(1)));
        typedef uint32_t v4a __attribute__((ext_vector_type(4)));
        for (uint32_t k = 16u * tid; k < have; k += 16u * NT) {
            if (k + 16 <= have) *reinterpret_cast<v4a *>(stage + k) = *reinterpret_cast<const v4u *>(data + k);
            else for (uint32_t j = k; j < have; j++) stage[j] = data[j];
        }
        __syncthreads();
    }
    const uint8_t *dp = stage + off;
    const bool ok = !(valid > 0 && data + off + nbytes > data_end);
    if (!ok) { err = 1; }
    int d[16];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        d[q] = 0;
        if (q < valid && ok) {
            const uint32_t code = (key >> (2 * q)) & 3;
            uint32_t zz = dp[0];
            if (code > 0) zz |= (uint32_t)dp[1] << 8;
            if (code > 1) zz |= (uint32_t)dp[2] << 16;
            if (code > 2) zz |= (uint32_t)dp[3] << 24;
            dp += code + 1;
            d[q] = (int)(zz >> 1) ^ -(int)(zz & 1);
        }
        sum += d[q];
    }
    // prefix sum of deltas across the workgroup (wrapping int32 arithmetic, as the CPU decoder)
    uint32_t tsum;
    const uint32_t before = block_excl_add((uint32_t)sum, ws, tsum);
    int acc = carry + (int)before;
    if (valid == 16) {
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc += d[2 * q];
            const uint32_t lo = (uint32_t)acc & 0xFFFFu;
            acc += d[2 * q + 1];
            w[q] = lo | ((uint32_t)acc << 16);
        }
        uint4 *o = reinterpret_cast<uint4 *>(out + i0);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (q < valid) { acc += d[q]; out[i0 + q] = (int16_t)acc; }
    }
    carry += (int)tsum;
    return total;
}

// The same tile by ONE wave (the fused inflate + unpack of k_inflate_par): 64 lanes x 16 values, wave scans instead of block scans,
// stage: SVB_WSTAGE bytes of LDS (the inflate window, dead by then), 16-byte aligned.
constexpr uint32_t SVB_WTILE = 64u * 16u;
constexpr uint32_t SVB_WSTAGE = 4u * SVB_WTILE + 16u;
// STAGED = false (round 6): the blob already lies in LDS (the no-payload decode that inflates into the window's storage) — the lanes pick
// their bytes where they are.
template <bool STAGED = true>
__device__ __forceinline__ uint32_t svb_decode_tile_wave(const uint8_t *keys, const uint8_t *data, const uint8_t *data_end, uint32_t n, uint32_t t0,
                                                         int16_t *__restrict__ out, int &carry, int &err, uint8_t *stage) {
    const int lane = lane_id();
    const uint32_t i0 = t0 + 16u * (uint32_t)lane;
    const int valid = i0 >= n ? 0 : (int)min(16u, n - i0);
    const int nk = (valid + 3) >> 2;
    uint32_t key = 0;
    if (valid == 16) {   // a full lane's four key bytes as one (unaligned) dword
        typedef uint32_t u1 __attribute__((aligned(1)));
        key = *reinterpret_cast<const u1 *>(keys + 4 * lane);
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < nk) key |= (uint32_t)keys[4 * lane + q] << (8 * q);
    }
    // bytes of my values: one each + the sum of the 2-bit codes (bits of values past `valid` are masked out)
    uint32_t nbytes;
    {
        const uint32_t km = valid >= 16 ? key : key & ((1u << (2 * valid)) - 1u);
        uint32_t c = (km & 0x33333333u) + ((km >> 2) & 0x33333333u);
        c = (c + (c >> 4)) & 0x0F0F0F0Fu;
        nbytes = (uint32_t)valid + ((c * 0x01010101u) >> 24);
    }
    const uint32_t incl = wave_incl_add(nbytes);
    const uint32_t off = incl - nbytes;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (STAGED) {   // data[0 .. min(total, bytes left)) -> stage
        const uint32_t have = (uint32_t)min((uint64_t)total, (uint64_t)(data_end - data));
        typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(1)));
        typedef uint32_t v4a __attribute__((ext_vector_type(4)));
        for (uint32_t k = 16u * (uint32_t)lane; k < have; k += 16u * 64u) {
            if (k + 16 <= have) *reinterpret_cast<v4a *>(stage + k) = *reinterpret_cast<const v4u *>(data + k);
            else for (uint32_t j = k; j < have; j++) stage[j] = data[j];
        }
        wave_sync();
    }
    const uint8_t *dp = (STAGED ? stage : data) + off;
    const bool ok = !(valid > 0 && data + off + nbytes > data_end);
    if (!ok) err = 1;
    int dlt[16];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        dlt[q] = 0;
        if (q < valid && ok) {
            const uint32_t code = (key >> (2 * q)) & 3;
            uint32_t zz = dp[0];
            if (code > 0) zz |= (uint32_t)dp[1] << 8;
            if (code > 1) zz |= (uint32_t)dp[2] << 16;
            if (code > 2) zz |= (uint32_t)dp[3] << 24;
            dp += code + 1;
            dlt[q] = (int)(zz >> 1) ^ -(int)(zz & 1);
        }
        sum += dlt[q];
    }
    const uint32_t incl2 = wave_incl_add((uint32_t)sum);
    int acc = carry + (int)(incl2 - (uint32_t)sum);
    if (valid == 16) {
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc += dlt[2 * q];
            const uint32_t lo = (uint32_t)acc & 0xFFFFu;
            acc += dlt[2 * q + 1];
            w[q] = lo | ((uint32_t)acc << 16);
        }
        uint4 *o = reinterpret_cast<uint4 *>(out + i0);
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++)
            if (q < valid) { acc += dlt[q]; out[i0 + q] = (int16_t)acc; }
    }
    carry += (int)(uint32_t)__builtin_amdgcn_readlane((int)incl2, 63);
    wave_sync();                                   // the next tile refills the stage
    return total;
}

}  // namespace s5
