// Which is it: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = active lanes, or active lanes x 4 (a wave64 VALU instruction takes four cycles
// on a SIMD16)?  Two kernels of the same 4096 dependent v_add per wave: every lane active / lanes 0..15 only (the rest masked off by a
// branch).  tools/pmc_lanes.sh prints the ratio for both: the full wave must read 64 (or 256), the quarter wave 16 (or 64).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LANES>
__global__ __launch_bounds__(64) void k_valu(uint32_t *out, uint32_t seed) {
    uint32_t x = seed + threadIdx.x;
    if ((int)threadIdx.x < LANES) {
#pragma unroll 1
        for (int i = 0; i < 512; i++) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(x) : "v"(seed));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 4096 * 64 * 4);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_valu<64>, dim3(4096), dim3(64), 0, 0, d, 7u);
        hipLaunchKernelGGL(k_valu<16>, dim3(4096), dim3(64), 0, 0, d, 7u);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
