/*
 * exzd.c — CPU ORACLE for the ex-zd signal codec.  TEST INFRASTRUCTURE ONLY — see s5oracle.h.
 *
 * ex-zd is slow5lib's second signal press (SLOW5_COMPRESS_EX_ZD, /root/reference/src/misc.c:261; the default of
 * `slow5tools degrade`, /root/reference/src/degrade.c:302).  slow5lib is an absent submodule, so the byte layout was
 * read off the reference's fixtures and is PINNED on them: tests/test_oracle_golden.py::test_exzd_* decodes every record
 * of the committed ex-zd files and re-encodes it to the identical blob; tools/validate_reference_exzd.py does the same
 * for all 45 ex-zd records under /root/reference/test/data (10 files: one_fast5/exp_1_*_zlib_ex_zd.blow5 and
 * exp/degrade/*_b{2,3,4}.blow5) — 45 of 45 bit-exact, two of them with exceptions, 43 without.
 *
 * Layout (little-endian):
 *   u8  version = 0
 *   u64 N                         samples
 *   u8  q                         trailing zero bits common to all samples (what `degrade` leaves); y[i] = x[i] >> q is coded
 *   --- nothing more when N == 0 ---
 *   u16 z0                        zigzag16 of y[0]
 *   u32 nex                       number of exceptions among z[1..N): z = zigzag(y[i] - y[i-1]) > 255
 *   if nex: u32 len | StreamVByte-32 (2-bit keys, ceil(nex/4) key bytes, then data) of the exception positions in z[1..):
 *                     first position, then gap - 1
 *           u32 len | StreamVByte-32 of (z - 256) of the exceptions
 *   u8  z[i] for every non-exception i in 1..N-1, in order
 * Not covered by a fixture, chosen here: an all-zero (or empty) signal gets q = 0.
 */
#include <stdlib.h>
#include <string.h>

#include "s5oracle.h"

static size_t svb32_bound(uint64_t n) { return (size_t)((n + 3) / 4 + 4 * n); }
static size_t svb32_encode(const uint32_t *v, uint64_t n, uint8_t *out) {
    const size_t nk = (size_t)((n + 3) / 4);
    memset(out, 0, nk);
    uint8_t *d = out + nk;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t x = v[i];
        const int code = x < (1u << 8) ? 0 : x < (1u << 16) ? 1 : x < (1u << 24) ? 2 : 3;
        out[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
        for (int b = 0; b <= code; b++) *d++ = (uint8_t)(x >> (8 * b));
    }
    return (size_t)(d - out);
}
/* returns bytes consumed, 0 if the section is truncated */
static size_t svb32_decode(const uint8_t *in, size_t len, uint64_t n, uint32_t *v) {
    const size_t nk = (size_t)((n + 3) / 4);
    if (nk > len) return 0;
    size_t p = nk;
    for (uint64_t i = 0; i < n; i++) {
        const int code = (in[i >> 2] >> (2 * (i & 3))) & 3;
        if (p + (size_t)code + 1 > len) return 0;
        uint32_t x = 0;
        for (int b = 0; b <= code; b++) x |= (uint32_t)in[p++] << (8 * b);
        v[i] = x;
    }
    return p;
}

size_t s5o_exzd_bound(uint64_t n) { return 16 + 8 + 2 * svb32_bound(n) + (size_t)n; }

size_t s5o_exzd_encode(const int16_t *x, uint64_t n, uint8_t *out) {
    size_t o = 0;
    out[o++] = 0;
    memcpy(out + o, &n, 8); o += 8;
    unsigned acc = 0;
    for (uint64_t i = 0; i < n; i++) acc |= (uint16_t)x[i];
    uint8_t q = 0;
    while (acc && !((acc >> q) & 1)) q++;
    out[o++] = q;
    if (n == 0) return o;
    uint32_t *pos = (uint32_t *)malloc(sizeof(uint32_t) * n), *val = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint8_t *small = (uint8_t *)malloc(n);
    if (!pos || !val || !small) { free(pos); free(val); free(small); return 0; }
    int32_t prev = x[0] >> q;                                       /* arithmetic shift: the dropped bits are zero */
    const uint16_t z0 = (uint16_t)(((uint32_t)prev << 1) ^ (uint32_t)(prev >> 31));
    memcpy(out + o, &z0, 2); o += 2;
    uint32_t nex = 0;
    uint64_t ns = 0, last = (uint64_t)-1;
    for (uint64_t i = 1; i < n; i++) {
        const int32_t y = x[i] >> q, d = y - prev;
        prev = y;
        const uint32_t z = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
        if (z > 255) {
            const uint64_t p = i - 1;                               /* position inside z[1..) */
            pos[nex] = (uint32_t)(last == (uint64_t)-1 ? p : p - last - 1);
            val[nex] = z - 256;
            last = p;
            nex++;
        } else small[ns++] = (uint8_t)z;
    }
    memcpy(out + o, &nex, 4); o += 4;
    if (nex) {
        uint32_t len = (uint32_t)svb32_encode(pos, nex, out + o + 4);
        memcpy(out + o, &len, 4); o += 4 + len;
        len = (uint32_t)svb32_encode(val, nex, out + o + 4);
        memcpy(out + o, &len, 4); o += 4 + len;
    }
    memcpy(out + o, small, ns); o += ns;
    free(pos); free(val); free(small);
    return o;
}

/* returns 0 on success; *n_out = sample count (call with out = NULL to learn it) */
int s5o_exzd_decode(const uint8_t *in, size_t len, int16_t *out, uint64_t *n_out) {
    if (len < 10 || in[0] != 0) return -1;
    uint64_t n;
    memcpy(&n, in + 1, 8);
    const unsigned q = in[9];
    *n_out = n;
    if (n == 0) return len == 10 ? 0 : -1;
    if (q > 15 || len < 16 || n > 0xFFFFFFF0ull) return -1;
    if (!out) return 0;
    uint16_t z0;
    uint32_t nex;
    memcpy(&z0, in + 10, 2);
    memcpy(&nex, in + 12, 4);
    size_t p = 16;
    if (nex > n - 1) return -1;
    uint32_t *pos = NULL, *val = NULL;
    if (nex) {
        pos = (uint32_t *)malloc(sizeof(uint32_t) * nex);
        val = (uint32_t *)malloc(sizeof(uint32_t) * nex);
        if (!pos || !val) { free(pos); free(val); return -1; }
        for (int s = 0; s < 2; s++) {
            uint32_t sl;
            if (p + 4 > len) { free(pos); free(val); return -1; }
            memcpy(&sl, in + p, 4); p += 4;
            if (sl > len - p || svb32_decode(in + p, sl, nex, s ? val : pos) != sl) { free(pos); free(val); return -1; }
            p += sl;
        }
    }
    if (len - p != n - 1 - nex) { free(pos); free(val); return -1; }
    int32_t y = (int32_t)(z0 >> 1) ^ -(int32_t)(z0 & 1);
    out[0] = (int16_t)((uint32_t)y << q);
    uint64_t e = 0, next_ex = nex ? pos[0] : (uint64_t)-1;
    int rc = 0;
    for (uint64_t i = 1; i < n; i++) {
        uint32_t z;
        if (i - 1 == next_ex) {
            z = val[e] + 256;
            e++;
            next_ex = e < nex ? next_ex + 1 + pos[e] : (uint64_t)-1;
        } else {
            if (p >= len) { rc = -1; break; }
            z = in[p++];
        }
        y += (int32_t)(z >> 1) ^ -(int32_t)(z & 1);
        out[i] = (int16_t)((uint32_t)y << q);
    }
    if (e != nex || p != len) rc = -1;
    free(pos); free(val);
    return rc;
}
