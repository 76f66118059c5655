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

// ---- wave-level scans (64 lanes) ----
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane_id() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d);
        if (lane_id() >= d) v = max(v, t);
    }
    return v;
}
__device__ __forceinline__ int wave_suffix_incl_min(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_down(v, d);
        if (lane_id() + d < 64) v = min(v, t);
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
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
    int prev = __shfl_up(incl, 1);
    if (lane_id() == 0) prev = ident;
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
