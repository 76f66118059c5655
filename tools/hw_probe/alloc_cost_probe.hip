// what the first batch call of a process pays for its workspaces: hipMalloc / hipHostMalloc / hipStreamCreate by size, first and second time
// hipcc --offload-arch=gfx950 -O2 tools/hw_probe/alloc_cost_probe.hip -o /tmp/alloc_cost_probe && /tmp/alloc_cost_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main() {
    double t0 = now();
    hipFree(0);
    printf("runtime + context up: %.1f ms\n", now() - t0);
    t0 = now(); hipStream_t st; hipStreamCreate(&st); printf("hipStreamCreate: %.2f ms\n", now() - t0);
    t0 = now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, (char *)nullptr, (size_t)0); hipStreamSynchronize(st); printf("first launch (code object load): %.2f ms\n", now() - t0);
    const size_t sizes[] = {1u << 20, 16u << 20, 64u << 20, 256u << 20, 1024u << 20};
    for (int rep = 0; rep < 2; rep++)
        for (size_t s : sizes) {
            void *d = nullptr, *h = nullptr;
            t0 = now(); hipMalloc(&d, s); double a = now() - t0;
            t0 = now(); hipLaunchKernelGGL(k_touch, dim3((unsigned)((s / 4096 + 255) / 256)), dim3(256), 0, st, (char *)d, s); hipStreamSynchronize(st); double b = now() - t0;
            t0 = now(); hipHostMalloc(&h, s, hipHostMallocPortable); double c = now() - t0;
            t0 = now(); hipFree(d); double e = now() - t0;
            t0 = now(); hipHostFree(h); double f = now() - t0;
            printf("rep %d  %5zu MB: hipMalloc %7.2f ms  first touch %6.2f ms  hipHostMalloc %7.2f ms  hipFree %6.2f  hipHostFree %6.2f\n", rep, s >> 20, a, b, c, e, f);
        }
    return 0;
}
