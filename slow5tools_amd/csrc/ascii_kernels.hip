// ascii_kernels.hip — the raw_signal column of SLOW5 ASCII on the device.
//
// A SLOW5 record line is `read_id \t read_group \t ... \t len_raw_signal \t s0,s1,s2,... \t aux...`
// (/root/reference/test/data/exp/one_fast5/exp_1_lossless.slow5); the comma-separated samples are ~95 % of
// its bytes.  slow5lib parses / prints them one strtol / sprintf at a time inside slow5_rec_depress_parse /
// slow5_rec_to_mem (call sites /root/reference/src/view.c:38,49).  Here one read is one 256-thread workgroup:
//   parse : a lane owns 16 characters; the number that *starts* in its span is its to finish (8 bytes of
//           look-ahead), the sample index is the count of commas before it (workgroup prefix sum);
//   format: a lane owns 8 samples; their printed lengths are prefix-summed into byte offsets.
// Both are HBM-bound byte work: ~4.2 text bytes + 2 signal bytes per sample.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slow5gpu.h"
#include "dev_common.h"

using namespace s5;

extern "C" void s5gpu_set_error(const char *fmt, ...);

#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            s5gpu_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return S5GPU_ERR_HIP;                                                                    \
        }                                                                                            \
    } while (0)

constexpr int PARSE_SPAN = 16;               // characters per lane per tile
constexpr int PARSE_AHEAD = 7;               // "-32768," started at the last owned character ends 7 characters later
constexpr int FMT_SPAN = 8;                  // samples per lane per tile

__global__ __launch_bounds__(NT) void k_ascii_parse(const s5gpu_txt_desc_t *desc, const uint8_t *text, int16_t *sig, int32_t *status) {
    __shared__ uint32_t ws[NW];
    __shared__ uint32_t s_err;
    const s5gpu_txt_desc_t d = desc[blockIdx.x];
    const uint8_t *t = text + d.txt_off;
    int16_t *out = sig + d.sig_off;
    const uint32_t len = d.txt_len, n = d.n_samples;
    if (threadIdx.x == 0) s_err = 0;
    __syncthreads();
    uint32_t err = 0, carry = 0;             // carry = commas before the current tile
    for (uint32_t base = 0; base < len; base += NT * PARSE_SPAN) {
        const uint32_t p0 = base + threadIdx.x * PARSE_SPAN;
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
        uint32_t prev = ',';
        if (p0 < len) {
            // the text may start at any byte (the chunk calls point into a file chunk as it was read): unaligned vector loads
            typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(1)));
            typedef uint32_t u2u __attribute__((ext_vector_type(2), aligned(1)));
            const u4u a = *reinterpret_cast<const u4u *>(t + p0);
            const u2u b = *reinterpret_cast<const u2u *>(t + p0 + 16);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y;
            if (p0) prev = t[p0 - 1];
        }
        uint32_t commas = 0;
#pragma unroll
        for (int j = 0; j < PARSE_SPAN; j++) {
            const uint32_t ch = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            commas += (p0 + j < len && ch == ',') ? 1u : 0u;
        }
        uint32_t total;
        uint32_t idx = carry + block_excl_add(commas, ws, total);
        carry += total;
        if (p0 < len) {
            bool active = prev == ',';       // a number starts at my first character
            bool open = true;                // still scanning
            uint32_t val = 0, nd = 0, neg = 0;
#pragma unroll
            for (int j = 0; j < PARSE_SPAN + PARSE_AHEAD; j++) {
                const uint32_t pos = p0 + j;
                if (j >= PARSE_SPAN && !active) open = false;      // the tail only finishes a number I started
                if (open && pos <= len) {
                    const uint32_t ch = pos < len ? (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu : (uint32_t)',';   // virtual terminator
                    if (ch == ',') {
                        if (active) {
                            if (nd == 0) err = err ? err : 3;
                            else if (val > 32767u + neg) err = err ? err : 2;
                            else if (idx >= n) err = err ? err : 4;
                            else out[idx] = (int16_t)(neg ? -(int)val : (int)val);
                        }
                        idx++;
                        active = true; val = 0; nd = 0; neg = 0;
                        if (j >= PARSE_SPAN - 1 || pos == len) open = false;   // what follows belongs to the next lane
                    } else if (active) {
                        const uint32_t dgt = ch - '0';
                        if (dgt <= 9) { val = val * 10 + dgt; if (++nd > 5) err = err ? err : 2; }
                        else if (ch == '-' && nd == 0 && !neg) neg = 1;
                        else err = err ? err : 1;
                    } else {
                        const uint32_t dgt = ch - '0';
                        if (dgt > 9 && ch != '-') err = err ? err : 1;   // validated by the owner too, but cheap
                    }
                }
            }
            if (open && active && p0 + PARSE_SPAN + PARSE_AHEAD <= len) err = err ? err : 2;   // number longer than the look-ahead
        }
    }
    const uint32_t count = len ? carry + 1 : 0;
    if (count != n) err = err ? err : 4;
    if (err) atomicMax(&s_err, err);
    __syncthreads();
    if (threadIdx.x == 0) status[blockIdx.x] = (int32_t)s_err;
}

// digits of |v| <= 32768, most significant first, packed little-endian (first character in the low byte), then a comma
__device__ __forceinline__ uint64_t print_sample(int v, bool comma, uint32_t &nchar) {
    const uint32_t neg = v < 0;
    uint32_t a = neg ? (uint32_t)(-v) : (uint32_t)v;
    const uint32_t d4 = a / 10000u; a -= d4 * 10000u;
    const uint32_t d3 = a / 1000u; a -= d3 * 1000u;
    const uint32_t d2 = a / 100u; a -= d2 * 100u;
    const uint32_t d1 = a / 10u;
    const uint32_t d0 = a - d1 * 10u;
    uint64_t wv = 0;
    uint32_t k = 0;
    if (neg) { wv |= (uint64_t)'-' << (8 * k); k++; }
    bool lead = false;
    if (d4) { wv |= (uint64_t)('0' + d4) << (8 * k); k++; lead = true; }
    if (lead || d3) { wv |= (uint64_t)('0' + d3) << (8 * k); k++; lead = true; }
    if (lead || d2) { wv |= (uint64_t)('0' + d2) << (8 * k); k++; lead = true; }
    if (lead || d1) { wv |= (uint64_t)('0' + d1) << (8 * k); k++; }
    wv |= (uint64_t)('0' + d0) << (8 * k); k++;
    if (comma) { wv |= (uint64_t)',' << (8 * k); k++; }
    nchar = k;
    return wv;
}

__global__ __launch_bounds__(NT) void k_ascii_format(const s5gpu_txt_desc_t *desc, const int16_t *sig, uint8_t *text, uint32_t *txt_len,
                                                     int32_t *status) {
    __shared__ uint32_t ws[NW];
    const s5gpu_txt_desc_t d = desc[blockIdx.x];
    const int16_t *in = sig + d.sig_off;
    uint8_t *out = text + d.txt_off;
    const uint32_t n = d.n_samples, cap = d.txt_len;
    uint32_t running = 0;
    bool fail = false;
    for (uint32_t base = 0; base < n; base += NT * FMT_SPAN) {
        const uint32_t i0 = base + threadIdx.x * FMT_SPAN;
        uint64_t wv[FMT_SPAN];
        uint32_t nc[FMT_SPAN], mine = 0;
#pragma unroll
        for (int j = 0; j < FMT_SPAN; j++) {
            nc[j] = 0; wv[j] = 0;
            if (i0 + j < n) wv[j] = print_sample(in[i0 + j], i0 + j + 1 < n, nc[j]);
            mine += nc[j];
        }
        uint32_t total;
        uint32_t pos = running + block_excl_add(mine, ws, total);
        if (running + total > cap) { fail = true; break; }     // uniform: every lane sees the same totals
#pragma unroll
        for (int j = 0; j < FMT_SPAN; j++) {
#pragma unroll
            for (int k = 0; k < 7; k++)
                if ((uint32_t)k < nc[j]) out[pos + k] = (uint8_t)(wv[j] >> (8 * k));
            pos += nc[j];
        }
        running += total;
    }
    if (threadIdx.x == 0) { txt_len[blockIdx.x] = fail ? 0 : running; status[blockIdx.x] = fail ? 5 : 0; }
}

__global__ __launch_bounds__(NT) void k_gather(const uint64_t *src_off, const uint32_t *len, const uint64_t *dst_off, const uint8_t *src,
                                               uint8_t *dst) {
    const uint8_t *s = src + src_off[blockIdx.x];
    uint8_t *o = dst + dst_off[blockIdx.x];
    const uint32_t l = len[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < l; i += NT) o[i] = s[i];
}

__global__ void k_ascii_noop(uint32_t *p) { if (p) p[0] = 0; }
int s5ascii_warm(hipStream_t st) {             // s5gpu_warmup (kernels.hip): this file's code object
    hipLaunchKernelGGL(k_ascii_noop, dim3(1), dim3(64), 0, st, (uint32_t *)nullptr);
    return hipGetLastError() == hipSuccess ? S5GPU_OK : S5GPU_ERR_HIP;
}

extern "C" int s5gpu_ascii_parse_dev(uint32_t n, const s5gpu_txt_desc_t *desc, const uint8_t *text, int16_t *sig, int32_t *status,
                                     void *stream_) {
    if (n == 0) return S5GPU_OK;
    if (!desc || !text || !sig || !status) { s5gpu_set_error("s5gpu_ascii_parse_dev: NULL argument"); return S5GPU_ERR_ARG; }
    hipLaunchKernelGGL(k_ascii_parse, dim3(n), dim3(NT), 0, (hipStream_t)stream_, desc, text, sig, status);
    HIP_TRY(hipGetLastError());
    return S5GPU_OK;
}
extern "C" int s5gpu_ascii_format_dev(uint32_t n, const s5gpu_txt_desc_t *desc, const int16_t *sig, uint8_t *text, uint32_t *txt_len,
                                      int32_t *status, void *stream_) {
    if (n == 0) return S5GPU_OK;
    if (!desc || !text || !sig || !txt_len || !status) { s5gpu_set_error("s5gpu_ascii_format_dev: NULL argument"); return S5GPU_ERR_ARG; }
    hipLaunchKernelGGL(k_ascii_format, dim3(n), dim3(NT), 0, (hipStream_t)stream_, desc, sig, text, txt_len, status);
    HIP_TRY(hipGetLastError());
    return S5GPU_OK;
}
extern "C" int s5gpu_gather_dev(uint32_t n, const uint64_t *src_off, const uint32_t *len, const uint64_t *dst_off, const uint8_t *src,
                                uint8_t *dst, void *stream_) {
    if (n == 0) return S5GPU_OK;
    if (!src_off || !len || !dst_off || !src || !dst) { s5gpu_set_error("s5gpu_gather_dev: NULL argument"); return S5GPU_ERR_ARG; }
    hipLaunchKernelGGL(k_gather, dim3(n), dim3(NT), 0, (hipStream_t)stream_, src_off, len, dst_off, src, dst);
    HIP_TRY(hipGetLastError());
    return S5GPU_OK;
}
