#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ uint32_t wave_incl_add_dpp(uint32_t v) {
    v += dpp_u32<0x111>(0, v);
    v += dpp_u32<0x112>(0, v);
    v += dpp_u32<0x114>(0, v);
    v += dpp_u32<0x118>(0, v);
    v += dpp_u32<0x142, 0xA>(0, v);
    v += dpp_u32<0x143, 0xC>(0, v);
    return v;
}
__device__ __forceinline__ int wave_incl_max_dpp(int v, int ident) {
    v = max(v, (int)dpp_u32<0x111>((uint32_t)ident, (uint32_t)v));
    v = max(v, (int)dpp_u32<0x112>((uint32_t)ident, (uint32_t)v));
    v = max(v, (int)dpp_u32<0x114>((uint32_t)ident, (uint32_t)v));
    v = max(v, (int)dpp_u32<0x118>((uint32_t)ident, (uint32_t)v));
    v = max(v, (int)dpp_u32<0x142, 0xA>((uint32_t)ident, (uint32_t)v));
    v = max(v, (int)dpp_u32<0x143, 0xC>((uint32_t)ident, (uint32_t)v));
    return v;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill) { return dpp_u32<0x138>(fill, v); }
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t fill) { return dpp_u32<0x130>(fill, v); }
__global__ void k(uint32_t *out) {
    uint32_t v = threadIdx.x * 3 + 1;
    out[threadIdx.x] = wave_incl_add_dpp(v);
    out[64 + threadIdx.x] = (uint32_t)wave_incl_max_dpp((int)((threadIdx.x * 37) % 50) - 5, -100);
    out[128 + threadIdx.x] = wave_shr1(v, 777);
    out[192 + threadIdx.x] = wave_shl1(v, 888);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0; uint32_t acc = 0; int mx = -100;
    for (int i = 0; i < 64; i++) {
        acc += i * 3 + 1; mx = mx > (int)((i * 37) % 50) - 5 ? mx : (int)((i * 37) % 50) - 5;
        if (h[i] != acc) { bad++; if (bad < 5) printf("add lane %d got %u want %u\n", i, h[i], acc); }
        if ((int)h[64 + i] != mx) { bad++; if (bad < 5) printf("max lane %d got %d want %d\n", i, (int)h[64 + i], mx); }
        uint32_t ws = i ? (i - 1) * 3 + 1 : 777;
        if (h[128 + i] != ws) { bad++; if (bad < 9) printf("shr lane %d got %u want %u\n", i, h[128 + i], ws); }
        uint32_t wl = i < 63 ? (i + 1) * 3 + 1 : 888;
        if (h[192 + i] != wl) { bad++; if (bad < 9) printf("shl lane %d got %u want %u\n", i, h[192 + i], wl); }
    }
    printf("dpp test bad=%d\n", bad);
    return bad != 0;
}
