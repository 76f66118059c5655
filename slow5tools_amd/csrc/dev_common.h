// dev_common.h — wave64 / workgroup primitives shared by the gfx950 kernels.
// One read (record) per workgroup; NT = 256 threads = 4 wave64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s5 {

constexpr int NT = 256;          // threads per workgroup
constexpr int NW = NT / 64;      // wave64 per workgroup

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// same-wave LDS/HBM hand-off: memory ops of one wave execute in order; this only pins the compiler
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// n bytes from s to d, the two ranges disjoint, any alignment (one lane; global memory takes unaligned dwords).  The waiting matches it
// serves are mostly 3-8 bytes: both ends of the piece are loaded before anything is stored and the two halves overlap in the middle —
// one round trip to memory whatever the length, at most one of four short paths per lane (8.92 -> 8.62 ms per 262 144 stock-zlib records
// against the 16 / 8 / 4 / byte-loop ladder, profiles/r04_wait_matches.txt)
__device__ __forceinline__ void copy_ends(uint8_t *d, const uint8_t *s, int n) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
    typedef uint32_t u1 __attribute__((aligned(1)));
    if (n >= 16) {
        const u4 e = *reinterpret_cast<const u4 *>(s + n - 16);
        for (int k = 0; k + 16 < n; k += 16) *reinterpret_cast<u4 *>(d + k) = *reinterpret_cast<const u4 *>(s + k);
        *reinterpret_cast<u4 *>(d + n - 16) = e;
    } else if (n >= 8) {
        const u2 a = *reinterpret_cast<const u2 *>(s), e = *reinterpret_cast<const u2 *>(s + n - 8);
        *reinterpret_cast<u2 *>(d) = a; *reinterpret_cast<u2 *>(d + n - 8) = e;
    } else if (n >= 4) {
        const u1 a = *reinterpret_cast<const u1 *>(s), e = *reinterpret_cast<const u1 *>(s + n - 4);
        *reinterpret_cast<u1 *>(d) = a; *reinterpret_cast<u1 *>(d + n - 4) = e;
    } else if (n > 0) {
        const uint8_t a = s[0], b = s[n >> 1], e = s[n - 1];
        d[0] = a; d[n >> 1] = b; d[n - 1] = e;
    }
}

// ---- wave-level scans (64 lanes) ----
// Prefix scans and reductions run on the DPP data path (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 and
// row_bcast:31 carry the row totals up): six VALU instructions, no LDS crossbar traffic and no address arithmetic, against
// six ds_bpermute round trips for the shuffle form.  All 64 lanes must be active at the call.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    v += dpp_u32<DPP_ROW_SHR1>(0, v);
    v += dpp_u32<DPP_ROW_SHR2>(0, v);
    v += dpp_u32<DPP_ROW_SHR4>(0, v);
    v += dpp_u32<DPP_ROW_SHR8>(0, v);
    v += dpp_u32<DPP_ROW_BCAST15, 0xA>(0, v);
    v += dpp_u32<DPP_ROW_BCAST31, 0xC>(0, v);
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v) {
    constexpr uint32_t ID = 0x80000000u;   // INT_MIN
    v = max(v, (int)dpp_u32<DPP_ROW_SHR1>(ID, (uint32_t)v));
    v = max(v, (int)dpp_u32<DPP_ROW_SHR2>(ID, (uint32_t)v));
    v = max(v, (int)dpp_u32<DPP_ROW_SHR4>(ID, (uint32_t)v));
    v = max(v, (int)dpp_u32<DPP_ROW_SHR8>(ID, (uint32_t)v));
    v = max(v, (int)dpp_u32<DPP_ROW_BCAST15, 0xA>(ID, (uint32_t)v));
    v = max(v, (int)dpp_u32<DPP_ROW_BCAST31, 0xC>(ID, (uint32_t)v));
    return v;
}
// value of the previous / next lane (lane 0 / lane 63 get `fill`)
__device__ __forceinline__ uint32_t wave_prev(uint32_t v, uint32_t fill) { return dpp_u32<DPP_WAVE_SHR1>(fill, v); }
__device__ __forceinline__ uint32_t wave_next(uint32_t v, uint32_t fill) { return dpp_u32<DPP_WAVE_SHL1>(fill, v); }
// value of lane (lane ^ D): quad permutes and row rotations on the DPP path, the gfx950 row / half-wave swaps above that —
// no LDS crossbar round trip (ds_bpermute) for the butterfly steps of the bitonic merges
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_bank_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANK_MASK, false);
}
template <int D>
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
    if constexpr (D == 1) return dpp_u32<0xB1>(v, v);                 // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return dpp_u32<0x4E>(v, v);            // quad_perm [2,3,0,1]
    else if constexpr (D == 4) return dpp_bank_u32<0x114, 0xA>(dpp_bank_u32<0x104, 0x5>(v, v), v);   // row_shl:4 into banks 0,2; row_shr:4 into 1,3
    else if constexpr (D == 8) return dpp_u32<0x128>(v, v);           // row_ror:8
    else if constexpr (D == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane_id() & 16) ? r[0] : r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane_id() & 32) ? r[0] : r[1]; }
}
// median of three unsigned values (v_med3_u32).  med3(x, p, 0) = min(x, p), med3(x, p, ~0) = max(x, p): one instruction for a
// compare-exchange whose direction is a per-lane constant
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { return min(max(a, b), max(min(a, b), c)); }
__device__ __forceinline__ int wave_suffix_incl_min(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_down(v, d);
        if (lane_id() + d < 64) v = min(v, t);
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {   // uniform result (an SGPR)
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_add(v), 63);
}

// ---- workgroup scans; ws = LDS scratch of >= NW words. Two barriers each. ----
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t *ws, uint32_t &total) {
    uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t x = ws[w];
        if (w < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}
// The same for a loop that scans once per round: round i uses the half ws[8 * (i & 1) ..] of a 16-word scratch, so ONE barrier per round
// is enough — a wave can only overwrite the half of round i in round i + 2, behind round i + 1's barrier, which no wave passes before all
// have read round i's words.  Contract: ws holds 16 words; NW <= 8; and NO OTHER scan may touch ws between two rounds or right behind the last
// one without a __syncthreads() of its own in front of it (a slower wave may still be reading the half a different scan would overwrite).
__device__ __forceinline__ uint32_t block_excl_add_alt(uint32_t v, uint32_t *ws, uint32_t &total, uint32_t round) {
    static_assert(NW <= 8, "block_excl_add_alt: a half of the 16-word scratch holds one word per wave");
    uint32_t *w8 = ws + 8u * (round & 1u);
    const uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) w8[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t x = w8[w];
        if (w < wave_id()) base += x;
        tot += x;
    }
    total = tot;
    return base + incl - v;
}
// workgroup exclusive prefix sum for a workgroup of NWV waves (block_excl_add above is the NW = 4 form)
template <int NWV>
__device__ __forceinline__ uint32_t block_excl_add_w(uint32_t v, uint32_t *ws, uint32_t &total) {
    const uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) ws[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        const uint32_t x = ws[w];
        if (w < wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}
// exclusive prefix max (identity `ident`)
__device__ __forceinline__ int block_excl_max(int v, int ident, uint32_t *ws) {
    int incl = wave_incl_max(v);
    if (lane_id() == 63) ws[wave_id()] = (uint32_t)incl;
    __syncthreads();
    int base = ident;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        int x = (int)ws[w];
        if (w < wave_id()) base = max(base, x);
    }
    __syncthreads();
    const int prev = (int)wave_prev((uint32_t)incl, (uint32_t)ident);
    return max(base, prev);
}
// exclusive suffix min: min over threads t' > t (identity `ident`)
__device__ __forceinline__ int block_suffix_excl_min(int v, int ident, uint32_t *ws) {
    int incl = wave_suffix_incl_min(v);
    if (lane_id() == 0) ws[wave_id()] = (uint32_t)incl;
    __syncthreads();
    int base = ident;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        int x = (int)ws[w];
        if (w > wave_id()) base = min(base, x);
    }
    __syncthreads();
    int nxt = __shfl_down(incl, 1);
    if (lane_id() == 63) nxt = ident;
    return min(base, nxt);
}

// integer-only synthetic read generator (bit-identical to oracle/synth.c)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace s5
