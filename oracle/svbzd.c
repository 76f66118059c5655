/*
 * svbzd.c — ORACLE (test infrastructure): scalar restatement of slow5lib's "svb-zd" signal codec.
 *
 * Restates: slow5lib thirdparty/streamvbyte (streamvbyte_slow5 namespace, named at
 * /root/reference/CMakeLists.txt:73, Makefile:140) as driven by slow5lib's signal press
 * (call sites src/view.c:49, src/merge.c:62).  Wire layout verified on the reference's golden
 * fixtures:
 *     u32 N | ceil(N/4) key bytes (2 bits per value, LSB first) | 1..4 LE data bytes per value
 * applied to z = zigzag32(x[i] - x[i-1]), x[-1] = 0, samples widened int16 -> int32 first.
 */
#include "s5oracle.h"
#include <string.h>

size_t s5o_svbzd_bound(uint64_t n) { return 4 + (size_t)((n + 3) / 4) + 4 * (size_t)n; }

size_t s5o_svbzd_encode(const int16_t *x, uint64_t n, uint8_t *out) {
    uint32_t n32 = (uint32_t)n;
    memcpy(out, &n32, 4);
    uint8_t *keys = out + 4;
    uint8_t *data = keys + (n + 3) / 4;
    int32_t prev = 0;
    uint8_t key = 0;
    for (uint64_t i = 0; i < n; i++) {
        int32_t d = (int32_t)x[i] - prev;
        prev = x[i];
        uint32_t z = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
        unsigned code = (z > 0xFFu) + (z > 0xFFFFu) + (z > 0xFFFFFFu);
        for (unsigned b = 0; b <= code; b++) *data++ = (uint8_t)(z >> (8 * b));
        key |= (uint8_t)(code << (2 * (i & 3)));
        if ((i & 3) == 3) { *keys++ = key; key = 0; }
    }
    if (n & 3) *keys++ = key;
    return (size_t)(data - out);
}

int s5o_svbzd_decode(const uint8_t *in, size_t in_len, int16_t *out, uint64_t *n_out) {
    if (in_len < 4) return -1;
    uint32_t n;
    memcpy(&n, in, 4);
    if (n_out) *n_out = n;
    size_t nkeys = ((size_t)n + 3) / 4;
    if (in_len < 4 + nkeys) return -1;
    if (!out) return 0;
    const uint8_t *keys = in + 4;
    const uint8_t *data = keys + nkeys;
    const uint8_t *end = in + in_len;
    int32_t prev = 0;
    for (uint32_t i = 0; i < n; i++) {
        unsigned code = (keys[i >> 2] >> (2 * (i & 3))) & 3;
        if ((size_t)(end - data) < code + 1) return -2;
        uint32_t z = 0;
        for (unsigned b = 0; b <= code; b++) z |= (uint32_t)data[b] << (8 * b);
        data += code + 1;
        int32_t d = (int32_t)(z >> 1) ^ -(int32_t)(z & 1);
        prev += d;
        out[i] = (int16_t)prev;
    }
    return data == end ? 0 : -3;
}
