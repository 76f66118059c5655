/*
 * rec.c — ORACLE (test infrastructure): BLOW5 record pack / parse.
 *
 * Restates slow5lib's slow5_rec_to_mem / slow5_rec_depress_parse for the binary format (absent
 * submodule; call sites src/view.c:38,49; src/merge.c:46,62; src/get.c:59).  The uncompressed layout
 * follows the reference's own literal statement in test/misc/make_blow5.c:76-97 and SURVEY.md
 * Appendix A.2/A.3 (verified on the golden fixtures):
 *   record  = u64 rec_size | rec_size bytes (zlib stream of payload, or payload)
 *   payload = u16 id_len | id | u32 read_group | f64 digitisation | f64 offset | f64 range |
 *             f64 sampling_rate | u64 L | signal bytes | aux bytes
 *   signal none  : L = sample count, N x int16
 *   signal svb-zd: L = byte length of the svb-zd blob that follows
 */
#include "s5oracle.h"
#include <string.h>

static size_t prim_len(const s5o_rec_t *r) { return 2 + (size_t)r->read_id_len + 4 + 4 * 8 + 8; }

size_t s5o_payload_bound(const s5o_rec_t *r, int sig_method) {
    size_t sig = sig_method == S5O_SIG_SVB_ZD ? s5o_svbzd_bound(r->len_raw_signal)
               : sig_method == S5O_SIG_EX_ZD ? s5o_exzd_bound(r->len_raw_signal) : 2 * (size_t)r->len_raw_signal;
    return prim_len(r) + sig + r->aux_len;
}

size_t s5o_rec_pack(const s5o_rec_t *r, int sig_method, uint8_t *out) {
    uint8_t *p = out;
    memcpy(p, &r->read_id_len, 2); p += 2;
    memcpy(p, r->read_id, r->read_id_len); p += r->read_id_len;
    memcpy(p, &r->read_group, 4); p += 4;
    memcpy(p, &r->digitisation, 8); p += 8;
    memcpy(p, &r->offset, 8); p += 8;
    memcpy(p, &r->range, 8); p += 8;
    memcpy(p, &r->sampling_rate, 8); p += 8;
    uint64_t L;
    if (sig_method == S5O_SIG_SVB_ZD) {
        L = s5o_svbzd_encode(r->raw_signal, r->len_raw_signal, p + 8);
    } else if (sig_method == S5O_SIG_EX_ZD) {
        L = s5o_exzd_encode(r->raw_signal, r->len_raw_signal, p + 8);
    } else {
        L = r->len_raw_signal;
        memcpy(p + 8, r->raw_signal, 2 * (size_t)L);
    }
    memcpy(p, &L, 8); p += 8;
    p += sig_method == S5O_SIG_NONE ? 2 * (size_t)L : (size_t)L;
    if (r->aux_len) { memcpy(p, r->aux, r->aux_len); p += r->aux_len; }
    return (size_t)(p - out);
}

size_t s5o_rec_to_mem_bound(const s5o_rec_t *r, int sig_method) {
    return 8 + s5o_zlib_bound(s5o_payload_bound(r, sig_method));
}

size_t s5o_rec_to_mem(const s5o_rec_t *r, int rec_method, int sig_method, uint8_t *scratch, uint8_t *out) {
    uint64_t sz;
    if (rec_method == S5O_REC_ZLIB) {
        size_t plen = s5o_rec_pack(r, sig_method, scratch);
        size_t zl = s5o_zlib_bound(plen);
        if ((s5o_pool_zstream ? s5o_zlib_compress_pooled(scratch, plen, out + 8, &zl) : s5o_zlib_compress(scratch, plen, out + 8, &zl)) != 0) return 0;
        sz = zl;
    } else {
        sz = s5o_rec_pack(r, sig_method, out + 8);
    }
    memcpy(out, &sz, 8);
    return 8 + (size_t)sz;
}

int s5o_rec_parse(const uint8_t *payload, size_t len, int sig_method, s5o_rec_t *r, int16_t *sig_out) {
    const uint8_t *p = payload, *end = payload + len;
    if (len < 2) return -1;
    memcpy(&r->read_id_len, p, 2); p += 2;
    if ((size_t)(end - p) < (size_t)r->read_id_len + 4 + 32 + 8) return -1;
    r->read_id = (const char *)p; p += r->read_id_len;
    memcpy(&r->read_group, p, 4); p += 4;
    memcpy(&r->digitisation, p, 8); p += 8;
    memcpy(&r->offset, p, 8); p += 8;
    memcpy(&r->range, p, 8); p += 8;
    memcpy(&r->sampling_rate, p, 8); p += 8;
    uint64_t L;
    memcpy(&L, p, 8); p += 8;
    if (sig_method == S5O_SIG_SVB_ZD) {
        if ((uint64_t)(end - p) < L) return -2;
        uint64_t n;
        int rc = s5o_svbzd_decode(p, (size_t)L, sig_out, &n);
        if (rc != 0) return rc - 10;
        r->len_raw_signal = n;
        p += L;
    } else if (sig_method == S5O_SIG_EX_ZD) {
        if ((uint64_t)(end - p) < L) return -2;
        uint64_t n;
        int rc = s5o_exzd_decode(p, (size_t)L, sig_out, &n);
        if (rc != 0) return rc - 20;
        r->len_raw_signal = n;
        p += L;
    } else {
        if ((uint64_t)(end - p) < 2 * L) return -2;
        r->len_raw_signal = L;
        if (sig_out) memcpy(sig_out, p, 2 * (size_t)L);
        p += 2 * L;
    }
    r->raw_signal = sig_out;
    r->aux = p;
    r->aux_len = (size_t)(end - p);
    return 0;
}
