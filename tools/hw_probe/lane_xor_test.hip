#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dppb(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false); }
__device__ __forceinline__ uint32_t xor1(uint32_t v) { return dppb<0xB1>(v, v); }
__device__ __forceinline__ uint32_t xor2(uint32_t v) { return dppb<0x4E>(v, v); }
__device__ __forceinline__ uint32_t xor4(uint32_t v) { uint32_t p = dppb<0x104, 0xF, 0x5>(v, v); return dppb<0x114, 0xF, 0xA>(p, v); }
__device__ __forceinline__ uint32_t xor8(uint32_t v) { return dppb<0x128>(v, v); }
__device__ __forceinline__ uint32_t xor16(uint32_t v) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? r[0] : r[1];
}
__device__ __forceinline__ uint32_t xor32(uint32_t v) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__global__ void k(uint32_t *out) {
    uint32_t v = threadIdx.x * 7 + 3;
    out[0 * 64 + threadIdx.x] = xor1(v);
    out[1 * 64 + threadIdx.x] = xor2(v);
    out[2 * 64 + threadIdx.x] = xor4(v);
    out[3 * 64 + threadIdx.x] = xor8(v);
    out[4 * 64 + threadIdx.x] = xor16(v);
    out[5 * 64 + threadIdx.x] = xor32(v);
}
int main() {
    uint32_t *d; (void)hipMalloc(&d, 6 * 64 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[6 * 64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int ds[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    for (int t = 0; t < 6; t++)
        for (int i = 0; i < 64; i++) {
            uint32_t want = (i ^ ds[t]) * 7 + 3;
            if (h[t * 64 + i] != want) { if (bad < 12) printf("xor%d lane %d got %u want %u\n", ds[t], i, h[t * 64 + i], want); bad++; }
        }
    printf("xor test bad=%d\n", bad);
    return bad != 0;
}
