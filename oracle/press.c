/*
 * press.c — ORACLE (test infrastructure): the zlib record codec of slow5lib's press layer.
 *
 * Restates: slow5lib src/slow5_press.c zlib path (absent from /root/reference; call sites
 * src/view.c:43-54).  The reference links system libz (Makefile:10, configure.ac:61-71); every
 * zlib record in the reference's fixtures is byte-identical to deflate(level 6 = Z_DEFAULT_COMPRESSION,
 * windowBits 15, memLevel 8, Z_DEFAULT_STRATEGY, one Z_FINISH call) of this container's zlib 1.2.11
 * (SURVEY.md §0 finding 2) — that is what this file calls.  One deflateInit2/deflateEnd per record,
 * matching the reference's slow5_press_init/free per record (src/view.c:43,54).
 */
#include "s5oracle.h"
#include <string.h>
#include <zlib.h>

size_t s5o_zlib_bound(size_t n) { return compressBound((uLong)n); }

int s5o_zlib_compress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len) {
    z_stream s;
    memset(&s, 0, sizeof s);
    if (deflateInit2(&s, Z_DEFAULT_COMPRESSION, Z_DEFLATED, MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
    s.next_in = (Bytef *)in;
    s.avail_in = (uInt)n;
    s.next_out = out;
    s.avail_out = (uInt)*out_len;
    int ret = deflate(&s, Z_FINISH);
    size_t produced = s.total_out;
    deflateEnd(&s);
    if (ret != Z_STREAM_END) return -2;
    *out_len = produced;
    return 0;
}

/* NOT the reference's shape: one deflate state per THREAD, reset per record (deflateReset) instead of allocated and cleared per record.
 * Only the cpu_baseline's "pooled_zstream" point uses it (bench.py), to show how much of the CPU figure is the reference's own
 * per-record slow5_press_init (src/view.c:43-54: a 256 KiB allocation + memset per record).  Same bytes out. */
static __thread z_stream t_pool;
static __thread int t_pool_ok;
int s5o_pool_zstream = 0;      /* switch for s5o_rec_to_mem (set by s5o_encode_batch_mt's mode argument) */
int s5o_zlib_compress_pooled(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len) {
    if (!t_pool_ok) {
        memset(&t_pool, 0, sizeof t_pool);
        if (deflateInit2(&t_pool, Z_DEFAULT_COMPRESSION, Z_DEFLATED, MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
        t_pool_ok = 1;
    } else if (deflateReset(&t_pool) != Z_OK) return -1;
    t_pool.next_in = (Bytef *)in;
    t_pool.avail_in = (uInt)n;
    t_pool.next_out = out;
    t_pool.avail_out = (uInt)*out_len;
    if (deflate(&t_pool, Z_FINISH) != Z_STREAM_END) return -2;
    *out_len = t_pool.total_out;
    return 0;
}
void s5o_zlib_pool_release(void) { if (t_pool_ok) { deflateEnd(&t_pool); t_pool_ok = 0; } }

int s5o_zlib_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len) {
    z_stream s;
    memset(&s, 0, sizeof s);
    if (inflateInit2(&s, MAX_WBITS) != Z_OK) return -1;
    s.next_in = (Bytef *)in;
    s.avail_in = (uInt)n;
    s.next_out = out;
    s.avail_out = (uInt)*out_len;
    int ret = inflate(&s, Z_FINISH);
    size_t produced = s.total_out;
    size_t consumed = s.total_in;
    inflateEnd(&s);
    if (ret != Z_STREAM_END) return -2;
    if (consumed != n) return -3;
    *out_len = produced;
    return 0;
}

/* RFC 1950 Adler-32, restated directly (checks the GPU's own Adler without going through libz) */
uint32_t s5o_adler32(const uint8_t *p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}
