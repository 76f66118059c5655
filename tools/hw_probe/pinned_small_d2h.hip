// pinned_small_d2h.hip — stand-alone reproducer attempt for the round-5 fault (DESIGN.md "A fault found on the way"): D2H copies into a SMALL
// pinned buffer that one host thread allocated while another thread was inside its own first kernels "completed without the bytes arriving"
// (1 % of two-worker s5view processes; cured in the library by never pinning less than 2 MiB at a time: csrc/host_api.hip s5_pinned_alloc).
// Nothing of libslow5gpu is linked here: if THIS shows the fault, it is the platform's (runtime / driver), not the library's.
//
// One process = one trial of the library's start-up shape: thread A creates its stream, allocates device memory and a big pinned buffer and
// runs its first kernels; thread B starts a little later (a sweep of delays, so that some trial hits A's first launches), allocates a
// SMALL pinned buffer (4.9 KB by default), runs a kernel on its own stream that fills a device buffer with a pattern, copies it back with
// hipMemcpyAsync into the small buffer, synchronises the stream and compares.  Exit code 3 + a line on stderr if the bytes did not arrive.
//   hipcc --offload-arch=gfx950 -O2 -o pinned_small_d2h tools/hw_probe/pinned_small_d2h.hip -lpthread
//   ./pinned_small_d2h [small_bytes] [delay_us] [rounds] [pieces]      (tools/hw_probe/pinned_small_d2h.sh runs a few hundred processes;
//                                                                       pieces = 1: two small copies of 520 + 16 bytes, the library's case)
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_fill(uint32_t *p, uint32_t n, uint32_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = seed ^ (i * 2654435761u);
}
__global__ void k_busy(uint32_t *p, uint32_t n, uint32_t rounds) {   // thread A's "first batch": scratch-free integer work over a big buffer
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = p[i];
    for (uint32_t r = 0; r < rounds; r++) v = v * 1664525u + 1013904223u;
    p[i] = v;
}

// Ballast (shape 4): the library's thread A does one thing the plain probe did not — its FIRST launch loads a 2 MB code object of 60 kernels
// onto the device and sets a function attribute on 22 of them, under thread B's allocation.  64 instantiations of an unrolled kernel make a
// code object of that size here; thread A launches one of them first and sets the attribute on all.
template <int K>
__global__ void k_ballast(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = p[i];
#pragma unroll
    for (int r = 0; r < 700; r++) v = (v ^ (uint32_t)(r * 2654435761u + K)) * (uint32_t)(2 * r + 2 * K + 1) + (v >> ((r + K) & 15));
    p[i] = v;
}
template <int K>
struct Ballast {
    static void touch(uint32_t *d, uint32_t n, hipStream_t st, bool launch) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ballast<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (launch) k_ballast<K><<<n / 256, 256, 0, st>>>(d, n);
        Ballast<K - 1>::touch(d, n, st, false);
    }
};
template <>
struct Ballast<0> { static void touch(uint32_t *, uint32_t, hipStream_t, bool) {} };

static size_t g_small = 4900;
static unsigned g_delay_us = 0;
static int g_rounds = 8;
static int g_pieces = 0;
static int g_ballast = 0;
static volatile int g_a_started = 0;

static void *thread_a(void *) {
    CHECK(hipSetDevice(0));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    g_a_started = 1;
    uint32_t *d = nullptr;
    void *h = nullptr;
    const uint32_t n = 16u << 20;
    CHECK(hipMalloc(&d, 4ull * n));
    CHECK(hipHostMalloc(&h, 32u << 20, hipHostMallocPortable));
    CHECK(hipMemsetAsync(d, 1, 4ull * n, st));
    if (g_ballast) Ballast<64>::touch(d, n, st, true);       // first launch out of a big code object + 64 attribute calls
    for (int r = 0; r < g_rounds; r++) {
        k_busy<<<n / 256, 256, 0, st>>>(d, n, 64);
        CHECK(hipMemcpyAsync(h, d, 32u << 20, hipMemcpyDeviceToHost, st));
        // ... and the first batch's workspace churn: device and pinned allocations under the runtime's locks while the kernels run
        void *d2 = nullptr, *h2 = nullptr;
        CHECK(hipMalloc(&d2, (size_t)(3 + r) << 20));
        CHECK(hipHostMalloc(&h2, (size_t)(1 + r) << 20, hipHostMallocPortable));
        CHECK(hipMemsetAsync(d2, 0, 1 << 20, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipHostFree(h2));
        CHECK(hipFree(d2));
    }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipHostFree(h));
    CHECK(hipFree(d));
    return nullptr;
}

static int g_bad = 0;
static void *thread_b(void *) {
    CHECK(hipSetDevice(0));
    while (!g_a_started) usleep(10);
    if (g_delay_us) usleep(g_delay_us);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int r = 0; r < g_rounds; r++) {
        // the library's grow-only workspaces: a buffer that is too small is freed and a larger one allocated (Buf::reserve) — round r asks
        // for a little more than round r - 1 did, so the small pinned buffer is freed and re-made every round, next to A's churn
        const size_t small = g_small + (size_t)r * 512;
        const uint32_t n = (uint32_t)(small / 4);
        uint32_t *d = nullptr, *h = nullptr;
        CHECK(hipMalloc(&d, 4ull * n + 64));
        CHECK(hipHostMalloc((void **)&h, small, hipHostMallocPortable));         // the SMALL pinned buffer, allocated while A is busy
        memset(h, 0, small);
        CHECK(hipMemsetAsync(d, 0, 4ull * n + 64, st));
        const uint32_t seed = 0x5105u + (uint32_t)r * 977u;
        k_fill<<<(n + 255) / 256, 256, 0, st>>>(d, n, seed);
        if (g_pieces) {
            // the library's case exactly: the buffer is pinned for ~4.9 KB, what lands in it are TWO small copies back to back — 8 (n + 1) bytes
            // of record offsets and 16 bytes of control words for a batch of 64 records (host_api.hip encode_stream_resident)
            const size_t a = 8 * 65, b = 16;
            CHECK(hipMemcpyAsync(h, d, a, hipMemcpyDeviceToHost, st));
            CHECK(hipMemcpyAsync((uint8_t *)h + a, (uint8_t *)d + a, b, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamSynchronize(st));
            uint32_t wrong = 0;
            for (uint32_t i = 0; i < (a + b) / 4; i++) if (h[i] != (seed ^ (i * 2654435761u))) wrong++;
            if (wrong) {
                fprintf(stderr, "pinned_small_d2h: round %d: %u of %zu words of the two small copies did not arrive in a %zu-byte pinned buffer\n", r, wrong, (a + b) / 4, small);
                g_bad = 1;
            }
            CHECK(hipHostFree(h));
            CHECK(hipFree(d));
            continue;
        }
        CHECK(hipMemcpyAsync(h, d, 4ull * n, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        uint32_t wrong = 0, zeros = 0;
        for (uint32_t i = 0; i < n; i++) { const uint32_t w = seed ^ (i * 2654435761u); if (h[i] != w) { wrong++; if (h[i] == 0) zeros++; } }
        if (wrong) {
            // the library's observation: a synchronous copy into the same buffer does arrive
            CHECK(hipMemcpy(h, d, 4ull * n, hipMemcpyDeviceToHost));
            uint32_t wrong2 = 0;
            for (uint32_t i = 0; i < n; i++) if (h[i] != (seed ^ (i * 2654435761u))) wrong2++;
            fprintf(stderr, "pinned_small_d2h: round %d: %u of %u words did not arrive (%u still zero) in a %zu-byte pinned buffer at %p; after a synchronous hipMemcpy: %u wrong\n",
                    r, wrong, n, zeros, small, (void *)h, wrong2);
            g_bad = 1;
        }
        CHECK(hipHostFree(h));
        CHECK(hipFree(d));
    }
    return nullptr;
}

int main(int argc, char **argv) {
    if (argc > 1) g_small = (size_t)atol(argv[1]);
    if (argc > 2) g_delay_us = (unsigned)atol(argv[2]);
    if (argc > 3) g_rounds = atoi(argv[3]);
    if (argc > 4) g_pieces = atoi(argv[4]);
    if (argc > 5) g_ballast = atoi(argv[5]);
    if (g_small < 64) g_small = 64;
    pthread_t a, b;
    pthread_create(&a, nullptr, thread_a, nullptr);
    pthread_create(&b, nullptr, thread_b, nullptr);
    pthread_join(a, nullptr);
    pthread_join(b, nullptr);
    return g_bad ? 3 : 0;
}
