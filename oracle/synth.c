/*
 * synth.c — ORACLE-side statement of the synthetic read generator.
 *
 * Not a reference algorithm (slow5tools has no generator); this is the measurement workload that
 * BASELINE.json's configs 2-5 ask for ("synthetic reads of stated sample length").  It is written
 * integer-only and counter-based so the device generator in slow5tools_amd/csrc/synth.hip produces
 * bit-identical int16 samples (checked in tests/test_gpu_parity.py).
 *
 * Model (nanopore-like): a read is a sequence of events; sample i starts a new event with
 * probability 1/10 (geometric dwell, mean 10; forced at multiples of 128 to bound the device
 * back-scan); event level ~ clip(520 + 60 g, 200, 1100), sample = level + 10 g', where g, g' are
 * Irwin-Hall(4) approximations to N(0,1) built from one 64-bit hash.
 */
#include "s5oracle.h"
#include <stdio.h>

static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline int64_t ih4(uint64_t h) { /* sum of 4 u16, centred: sd = 65536/sqrt(3) */
    return (int64_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
}
static inline uint64_t read_key(uint64_t seed, uint64_t r) { return mix64(seed + r * 0xD1342543DE82EF95ull); }
static inline int is_boundary(uint64_t key, uint64_t i) {
    if ((i & 127) == 0) return 1;
    uint64_t h = mix64(key ^ (i * 4 + 1));
    return (((h >> 32) * 10ull) >> 32) == 0;
}
static inline int32_t event_level(uint64_t key, uint64_t i) {
    int64_t g = ih4(mix64(key ^ (i * 4 + 2)));
    int64_t L = 520 + ((g * 1663) >> 20);
    return (int32_t)(L < 200 ? 200 : L > 1100 ? 1100 : L);
}
static inline int32_t noise(uint64_t key, uint64_t i) {
    int64_t g = ih4(mix64(key ^ (i * 4 + 0)));
    return (int32_t)((g * 277) >> 20);
}

void s5o_synth_read(uint64_t seed, uint64_t read_idx, uint64_t n, int16_t *out) {
    uint64_t key = read_key(seed, read_idx);
    int32_t level = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (is_boundary(key, i)) level = event_level(key, i);
        int32_t v = level + noise(key, i);
        out[i] = (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
}

void s5o_synth_read_id(uint64_t r, char out[37]) {
    snprintf(out, 37, "%08x-0000-4000-8000-%012llx", (unsigned)(r & 0xFFFFFFFFu),
             (unsigned long long)(r & 0xFFFFFFFFFFFFull));
}
