// Probe for the tree description of csrc/zstd_enc_dev.h (zstd_desc_head / _chain / _pack): runs the tree description for random sets of code lengths on one wave and
// checks the FSE state chain on the host (state[k] is a cell of weight w[k] whose interval holds state[k+2]).
// Lesson it was written for: LDS is NOT zero in a busy kernel — a variant that packed table bytes of rows it had not
// written passed here (fresh LDS) and failed in the encoder until the bytes were masked.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I slow5tools_amd/csrc tools/hw_probe/zstd_desc_probe.hip -o gpurun_tmp/zprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "zstd_enc_dev.h"
using namespace s5;

__global__ void k_probe(const uint8_t *lens_in, int n, int maxbits, uint8_t *out, uint32_t *dl_out) {
    __shared__ DeflShared S;
    ZstdDesc &D = *reinterpret_cast<ZstdDesc *>(S.code);
    for (int i = threadIdx.x; i < 320; i += 64) S.lens[i] = i < 256 ? lens_in[i] : 0;
    __syncthreads();
    uint32_t d = zstd_desc_head(D, S.lens, n, maxbits);
    if (d >> 31) {
        zstd_desc_chain(D, S.lens, n, maxbits, 0);
        zstd_desc_chain(D, S.lens, n, maxbits, 1);
        d = zstd_desc_pack(D, n, d & 0x7FFFFFFFu);
    }
    __syncthreads();
    if (threadIdx.x == 0) *dl_out = d;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(&D);
    for (uint32_t i = threadIdx.x; i < sizeof(ZstdDesc); i += 64) out[i] = p[i];
}

int main(int argc, char **argv) {
    srand(argc > 1 ? atoi(argv[1]) : 1);
    int bad = 0;
    for (int trial = 0; trial < 200; trial++) {
        // random complete-ish code: lengths for 256 symbols via repeated splitting
        uint8_t lens[256];
        int present = (trial & 1) ? 12 + rand() % 60 : 140 + rand() % 116;
        for (int i = 0; i < 256; i++) lens[i] = 0;
        // assign lengths by a random Kraft-complete construction: start with one code of length 0, split random leaves
        int L[256], m = 1; L[0] = 0;
        while (m < present) { int j = rand() % m; if (L[j] >= 11) { int ok = 0; for (int t = 0; t < m; t++) if (L[t] < 11) { j = t; ok = 1; break; } if (!ok) break; } L[j]++; L[m++] = L[j]; }
        int idx[256]; for (int i = 0; i < 256; i++) idx[i] = i;
        for (int i = 255; i > 0; i--) { int j = rand() % (i + 1); int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
        int maxsym = 0, maxbits = 0;
        for (int i = 0; i < m; i++) { lens[idx[i]] = (uint8_t)L[i]; if (i == 0 && idx[i] < 200) { lens[idx[i]] = 0; lens[200 + rand() % 56] = (uint8_t)L[i]; } if (idx[i] > maxsym) maxsym = idx[i]; if (L[i] > maxbits) maxbits = L[i]; }
        uint8_t *d_l, *d_o; uint32_t *d_d;
        hipMalloc(&d_l, 256); hipMalloc(&d_o, sizeof(ZstdDesc)); hipMalloc(&d_d, 4);
        hipMemcpy(d_l, lens, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_l, maxsym, maxbits, d_o, d_d);
        ZstdDesc h; uint32_t dl;
        hipMemcpy(&h, d_o, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&dl, d_d, 4, hipMemcpyDeviceToHost);
        hipFree(d_l); hipFree(d_o); hipFree(d_d);
        if (maxsym <= 128 || dl == 0) continue;
        // check the chain on the host: state[k] must be a cell of weight w[k] whose interval holds state[k+2]
        int n = maxsym, errs = 0;
        for (int k = 0; k < n; k++) {
            const uint32_t c = h.cell[h.state[k] & 63];
            const uint32_t w = lens[k] ? (uint32_t)(maxbits + 1 - lens[k]) : 0u;
            if (h.state[k] > 63 || (c & 255) != w) { if (errs < 3) printf("trial %d k %d: state %d cell sym %u want w %u\n", trial, k, h.state[k], c & 255, w); errs++; continue; }
            if (k + 2 < n) {
                const uint32_t nb = (c >> 8) & 255, base = c >> 16, nx = h.state[k + 2];
                if (nx < base || nx >= base + (1u << nb)) { if (errs < 3) printf("trial %d k %d: next %u not in [%u,+%u)\n", trial, k, nx, base, 1u << nb); errs++; }
            }
        }
        if (errs) bad++;
    }
    printf("bad trials: %d\n", bad);
    return 0;
}
