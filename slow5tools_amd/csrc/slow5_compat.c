/*
 * slow5_compat.c — slow5lib's record-press entry points (include/slow5_compat.h) in plain C on top of
 * the C ABI of the MI355X kernels (include/slow5gpu.h).  Host code stays C, as in the reference; nothing
 * here computes a codec on the CPU — every call ends in s5gpu_*_batch and fails if there is no GPU.
 *
 * Mirrors the call contracts visible in slow5tools:
 *   src/view.c:35-57   depress_parse -> press_init -> rec_to_mem -> press_free -> rec_free
 *   src/merge.c:43-70  the same with read->read_group rewritten and aux dropped when lossy
 *   src/get.c:37-66    slow5_get (depress+parse) then rec_to_mem
 */
#include "../../include/slow5_compat.h"

#include <stdlib.h>
#include <string.h>

#include "../../include/slow5gpu.h"
#include "../../include/slow5gpu_hooks.h"

__thread int slow5_errno = 0;

/* ---- process-wide switches (src/main.c:246-247, src/get.c:194) ---- */
#include <stdarg.h>
#include <stdio.h>
static enum slow5_log_level_opt g_log_level = SLOW5_LOG_INFO;
static enum slow5_exit_condition_opt g_exit_cond = SLOW5_EXIT_OFF;
int slow5_compat_skip_rid = 0;   /* read by blow5_file.c */
void slow5_set_log_level(enum slow5_log_level_opt l) { g_log_level = l; }
void slow5_set_exit_condition(enum slow5_exit_condition_opt e) { g_exit_cond = e; }
void slow5_set_skip_rid(void) { slow5_compat_skip_rid = 1; }
/* error report of this layer: printed at log level >= ERR, fatal under SLOW5_EXIT_ON_ERR (the policy slow5tools selects) */
void slow5_compat_error(const char *fmt, ...) {
    if (g_log_level >= SLOW5_LOG_ERR) {
        va_list ap;
        va_start(ap, fmt);
        fputs("[slow5gpu::ERROR] ", stderr);
        vfprintf(stderr, fmt, ap);
        fputc('\n', stderr);
        va_end(ap);
    }
    if (g_exit_cond >= SLOW5_EXIT_ON_ERR) exit(EXIT_FAILURE);
}

void slow5_compat_warn(const char *fmt, ...) {
    if (g_log_level >= SLOW5_LOG_WARN) {
        va_list ap;
        va_start(ap, fmt);
        fputs("[slow5gpu::WARNING] ", stderr);
        vfprintf(stderr, fmt, ap);
        fputc('\n', stderr);
        va_end(ap);
    }
    if (g_exit_cond >= SLOW5_EXIT_ON_WARN) exit(EXIT_FAILURE);
}

static int rec_code(enum slow5_press_method m) {
    return m == SLOW5_COMPRESS_NONE ? S5GPU_REC_NONE : m == SLOW5_COMPRESS_ZLIB ? S5GPU_REC_ZLIB : m == SLOW5_COMPRESS_ZSTD ? S5GPU_REC_ZSTD : -1;
}
static int sig_code(enum slow5_press_method m) {
    return m == SLOW5_COMPRESS_NONE ? S5GPU_SIG_NONE : m == SLOW5_COMPRESS_SVB_ZD ? S5GPU_SIG_SVB_ZD : m == SLOW5_COMPRESS_EX_ZD ? S5GPU_SIG_EX_ZD : -1;
}

struct slow5_press *slow5_press_init(slow5_press_method_t method) {
    if (rec_code(method.record_method) < 0 || sig_code(method.signal_method) < 0) {
        slow5_errno = SLOW5_ERR_PRESS;
        return NULL;
    }
    struct slow5_press *p = (struct slow5_press *)calloc(1, sizeof *p);
    struct __slow5_press *r = (struct __slow5_press *)calloc(1, sizeof *r);
    struct __slow5_press *s = (struct __slow5_press *)calloc(1, sizeof *s);
    if (!p || !r || !s) { free(p); free(r); free(s); slow5_errno = SLOW5_ERR_MEM; return NULL; }
    r->method = method.record_method;   /* codec state lives on the device side: nothing to allocate per record */
    s->method = method.signal_method;
    p->record_press = r;
    p->signal_press = s;
    return p;
}

void slow5_press_free(struct slow5_press *comp) {
    if (!comp) return;
    free(comp->record_press);
    free(comp->signal_press);
    free(comp);
}

void *slow5_ptr_compress_solo(enum slow5_press_method method, const void *ptr, size_t count, size_t *n) {
    void *out = NULL;
    size_t len = 0;
    if (method == SLOW5_COMPRESS_NONE) {
        out = malloc(count ? count : 1);
        if (!out) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
        memcpy(out, ptr, count);
        len = count;
    } else {
        const int stage = method == SLOW5_COMPRESS_ZLIB ? 0 : method == SLOW5_COMPRESS_SVB_ZD ? 2 : method == SLOW5_COMPRESS_ZSTD ? 5 : method == SLOW5_COMPRESS_EX_ZD ? 6 : -1;
        const void *in[1] = {ptr};
        if (stage < 0 || s5gpu_solo_batch(stage, 1, in, &count, &out, &len, NULL) != S5GPU_OK) { slow5_errno = SLOW5_ERR_PRESS; return NULL; }
    }
    if (n) *n = len;
    return out;
}

void *slow5_ptr_depress_solo(enum slow5_press_method method, const void *ptr, size_t count, size_t *n) {
    void *out = NULL;
    size_t len = 0;
    if (method == SLOW5_COMPRESS_NONE) {
        out = malloc(count ? count : 1);
        if (!out) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
        memcpy(out, ptr, count);
        len = count;
    } else {
        const int stage = method == SLOW5_COMPRESS_ZLIB ? 1 : method == SLOW5_COMPRESS_SVB_ZD ? 3 : method == SLOW5_COMPRESS_ZSTD ? 4 : method == SLOW5_COMPRESS_EX_ZD ? 7 : -1;
        const void *in[1] = {ptr};
        if (stage < 0 || s5gpu_solo_batch(stage, 1, in, &count, &out, &len, NULL) != S5GPU_OK) { free(out); slow5_errno = SLOW5_ERR_PRESS; return NULL; }
    }
    if (n) *n = len;
    return out;
}

void *slow5_ptr_compress(struct __slow5_press *comp, const void *ptr, size_t count, size_t *n) {
    if (!comp) { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    return slow5_ptr_compress_solo(comp->method, ptr, count, n);
}
void *slow5_ptr_depress(struct __slow5_press *comp, const void *ptr, size_t count, size_t *n) {
    if (!comp) { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    return slow5_ptr_depress_solo(comp->method, ptr, count, n);
}

struct slow5_rec *slow5_rec_init(void) { return (struct slow5_rec *)calloc(1, sizeof(struct slow5_rec)); }

void slow5_rec_free(struct slow5_rec *read) {
    if (!read) return;
    free(read->read_id);
    free(read->raw_signal);
    free(read->aux_blob);
    free(read);
}

/* record head = u16 id_len | id | u32 read_group | 4 x f64   */
static uint8_t *pack_head(const struct slow5_rec *r, uint32_t *len) {
    const uint32_t n = 2u + r->read_id_len + 4u + 32u;
    uint8_t *h = (uint8_t *)malloc(n), *p = h;
    if (!h) return NULL;
    memcpy(p, &r->read_id_len, 2); p += 2;
    memcpy(p, r->read_id, r->read_id_len); p += r->read_id_len;
    memcpy(p, &r->read_group, 4); p += 4;
    memcpy(p, &r->digitisation, 8); p += 8;
    memcpy(p, &r->offset, 8); p += 8;
    memcpy(p, &r->range, 8); p += 8;
    memcpy(p, &r->sampling_rate, 8);
    *len = n;
    return h;
}

static int rec_to_mem_batch_any(int64_t n, struct slow5_rec **reads, int drop_aux, slow5_press_method_t to, void **out, size_t *out_len, void **arena);
int slow5_gpu_rec_to_mem_batch(int64_t n, struct slow5_rec **reads, int drop_aux, slow5_press_method_t to, void **out,
                               size_t *out_len) {
    return rec_to_mem_batch_any(n, reads, drop_aux, to, out, out_len, NULL);
}
static int rec_to_mem_batch_any(int64_t n, struct slow5_rec **reads, int drop_aux, slow5_press_method_t to, void **out, size_t *out_len, void **arena) {
    const int rc_m = rec_code(to.record_method), sg_m = sig_code(to.signal_method);
    if (arena) *arena = NULL;
    if (n < 0 || rc_m < 0 || sg_m < 0) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
    if (n == 0) return 0;
    const int16_t **sig = (const int16_t **)malloc(sizeof(void *) * n);
    uint64_t *ns = (uint64_t *)malloc(sizeof(uint64_t) * n);
    const void **hdr = (const void **)calloc(n, sizeof(void *));
    uint32_t *hl = (uint32_t *)malloc(sizeof(uint32_t) * n);
    const void **aux = (const void **)malloc(sizeof(void *) * n);
    uint32_t *al = (uint32_t *)malloc(sizeof(uint32_t) * n);
    int ret = -1;
    if (!sig || !ns || !hdr || !hl || !aux || !al) { slow5_errno = SLOW5_ERR_MEM; goto done; }
    for (int64_t i = 0; i < n; i++) {
        const struct slow5_rec *r = reads[i];
        sig[i] = r->raw_signal;
        ns[i] = r->len_raw_signal;
        hdr[i] = pack_head(r, &hl[i]);
        if (!hdr[i]) { slow5_errno = SLOW5_ERR_MEM; goto done; }
        aux[i] = drop_aux ? NULL : r->aux_blob;
        al[i] = drop_aux ? 0u : (uint32_t)r->aux_len;
    }
    if ((arena ? s5gpu_encode_batch_arena((uint32_t)n, sig, ns, hdr, hl, aux, al, rc_m, sg_m, out, out_len, arena)
               : s5gpu_encode_batch((uint32_t)n, sig, ns, hdr, hl, aux, al, rc_m, sg_m, out, out_len)) != S5GPU_OK) { slow5_errno = SLOW5_ERR_PRESS; goto done; }
    ret = 0;
done:
    if (hdr) for (int64_t i = 0; i < n; i++) free((void *)hdr[i]);
    free(sig); free(ns); free(hdr); free(hl); free(aux); free(al);
    return ret;
}

void *slow5_rec_to_mem(struct slow5_rec *read, struct slow5_aux_meta *aux_meta, enum slow5_fmt format,
                       struct slow5_press *compress, size_t *n) {
    if (!read || (format != SLOW5_FORMAT_BINARY && format != SLOW5_FORMAT_ASCII)) { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    slow5_press_method_t m = {SLOW5_COMPRESS_NONE, SLOW5_COMPRESS_NONE};
    if (compress && format == SLOW5_FORMAT_BINARY) { m.record_method = compress->record_press->method; m.signal_method = compress->signal_press->method; }
    void *out = NULL;
    size_t len = 0;
    if (slow5_gpu_rec_to_mem_batch(1, &read, aux_meta == NULL, m, &out, &len) != 0) return NULL;
    if (format == SLOW5_FORMAT_ASCII) {
        /* the record line: the uncompressed record goes back through the ASCII formatter, whose raw_signal
         * column is printed on the device; the line ends in a newline like slow5lib's */
        const void *rec = (const char *)out + 8;
        size_t rlen = len - 8, llen = 0;
        void *line = NULL;
        const int rc = s5gpu_blow5_to_ascii_batch(1, &rec, &rlen, S5GPU_REC_NONE, S5GPU_SIG_NONE, aux_meta ? aux_meta->num : 0, aux_meta ? aux_meta->types : NULL, NULL, 0,
                                                  &line, &llen, NULL);
        free(out);
        if (rc != S5GPU_OK) { slow5_errno = SLOW5_ERR_RECPARSE; return NULL; }
        out = line;
        len = llen;
    }
    if (n) *n = len;
    return out;
}

int slow5_rec_fwrite(FILE *fp, struct slow5_rec *read, struct slow5_aux_meta *aux_meta, enum slow5_fmt format, struct slow5_press *compress) {
    if (!fp) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    size_t n = 0;
    void *mem = slow5_rec_to_mem(read, aux_meta, format, compress, &n);
    if (!mem) { slow5_compat_error("slow5_rec_fwrite: the record could not be encoded (%s)", s5gpu_last_error()); return -1; }
    const size_t w = fwrite(mem, 1, n, fp);
    free(mem);
    if (w != n) { slow5_errno = SLOW5_ERR_IO; slow5_compat_error("slow5_rec_fwrite: short write"); return -1; }
    return (int)n;
}

static int fill_rec(struct slow5_rec **pr, const s5gpu_rec_fields_t *f, const uint8_t *payload, int16_t *sig) {
    struct slow5_rec *r = *pr;
    if (!r) { r = slow5_rec_init(); if (!r) return -1; *pr = r; }
    else { free(r->read_id); free(r->raw_signal); free(r->aux_blob); r->read_id = NULL; r->raw_signal = NULL; r->aux_blob = NULL; }
    r->read_id_len = (uint16_t)f->read_id_len;
    r->read_id = (char *)malloc((size_t)f->read_id_len + 1);
    if (!r->read_id) return -1;
    memcpy(r->read_id, payload + 2, f->read_id_len);
    r->read_id[f->read_id_len] = '\0';
    r->read_group = f->read_group;
    r->digitisation = f->digitisation;
    r->offset = f->offset;
    r->range = f->range;
    r->sampling_rate = f->sampling_rate;
    r->len_raw_signal = f->n_samples;
    r->aux_len = f->aux_len;
    if (f->aux_len) {
        r->aux_blob = (uint8_t *)malloc(f->aux_len);
        if (!r->aux_blob) { r->len_raw_signal = 0; r->aux_len = 0; return -1; }   /* sig stays the caller's: nothing to free twice */
        memcpy(r->aux_blob, payload + f->aux_off, f->aux_len);
    }
    r->raw_signal = sig;   /* ownership moves to the record only once every allocation has succeeded */
    return 0;
}

int slow5_gpu_depress_parse_batch(int64_t n, char **mem, size_t *bytes, slow5_press_method_t from, struct slow5_rec **reads) {
    const int rc_m = rec_code(from.record_method), sg_m = sig_code(from.signal_method);
    if (n < 0 || rc_m < 0 || sg_m < 0) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
    if (n == 0) return 0;
    void **pay = (void **)calloc(n, sizeof(void *));
    int16_t **sig = (int16_t **)calloc(n, sizeof(void *));
    s5gpu_rec_fields_t *f = (s5gpu_rec_fields_t *)calloc(n, sizeof *f);
    int ret = -1;
    if (!pay || !sig || !f) { slow5_errno = SLOW5_ERR_MEM; goto done; }
    if (s5gpu_decode_batch((uint32_t)n, (const void *const *)mem, bytes, rc_m, sg_m, pay, sig, f) != S5GPU_OK) { slow5_errno = SLOW5_ERR_RECPARSE; goto done; }
    for (int64_t i = 0; i < n; i++) {
        if (fill_rec(&reads[i], &f[i], (const uint8_t *)pay[i], sig[i]) != 0) { slow5_errno = SLOW5_ERR_MEM; goto done; }
        sig[i] = NULL;
        /* like slow5lib: *mem now holds the uncompressed record, the caller still frees it (src/view.c:41) */
        free(mem[i]);
        mem[i] = (char *)pay[i];
        bytes[i] = f[i].payload_len;
        pay[i] = NULL;
    }
    ret = 0;
done:
    if (pay) for (int64_t i = 0; i < n; i++) free(pay[i]);
    if (sig) for (int64_t i = 0; i < n; i++) free(sig[i]);
    free(pay); free(sig); free(f);
    return ret;
}

int slow5_rec_depress_parse(char **mem, size_t *bytes, const char *read_id, struct slow5_rec **read, struct slow5_file *s5p) {
    (void)read_id;
    if (!mem || !*mem || !bytes || !read || !s5p || s5p->format != SLOW5_FORMAT_BINARY) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    slow5_press_method_t m = {SLOW5_COMPRESS_NONE, SLOW5_COMPRESS_NONE};
    if (s5p->compress) { m.record_method = s5p->compress->record_press->method; m.signal_method = s5p->compress->signal_press->method; }
    return slow5_gpu_depress_parse_batch(1, mem, bytes, m, read);
}

int slow5_decode(char **mem, size_t *bytes, struct slow5_rec **read, struct slow5_file *s5p) {
    const int rc = slow5_rec_depress_parse(mem, bytes, NULL, read, s5p);
    if (rc != 0) { slow5_compat_error("slow5_decode: the record could not be decoded (%s)", s5gpu_last_error()); return slow5_errno ? slow5_errno : SLOW5_ERR_RECPARSE; }
    return 0;
}

int slow5_gpu_convert_batch(int64_t n, char **mem, size_t *bytes, enum slow5_fmt from_fmt, slow5_press_method_t from,
                            const struct slow5_aux_meta *aux_meta, enum slow5_fmt to_fmt, slow5_press_method_t to,
                            const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len) {
    if (from_fmt == SLOW5_FORMAT_BINARY && to_fmt == SLOW5_FORMAT_BINARY)
        return slow5_gpu_recompress_batch(n, mem, bytes, from, to, new_read_group, drop_aux, out, out_len);
    const uint32_t n_aux = aux_meta ? aux_meta->num : 0;
    const uint8_t *types = aux_meta ? aux_meta->types : NULL;
    if (n < 0 || (from_fmt != SLOW5_FORMAT_ASCII && from_fmt != SLOW5_FORMAT_BINARY) || (to_fmt != SLOW5_FORMAT_ASCII && to_fmt != SLOW5_FORMAT_BINARY)) {
        slow5_errno = SLOW5_ERR_ARG;
        return -1;
    }
    if (n == 0) return 0;
    int rc;
    if (from_fmt == SLOW5_FORMAT_ASCII && to_fmt == SLOW5_FORMAT_BINARY) {
        const int tr = rec_code(to.record_method), ts = sig_code(to.signal_method);
        if (tr < 0 || ts < 0) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
        rc = s5gpu_ascii_to_blow5_batch((uint32_t)n, (const char *const *)mem, bytes, n_aux, types, tr, ts, new_read_group, drop_aux, out, out_len, NULL);
    } else if (from_fmt == SLOW5_FORMAT_BINARY) {
        const int fr = rec_code(from.record_method), fs = sig_code(from.signal_method);
        if (fr < 0 || fs < 0) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
        rc = s5gpu_blow5_to_ascii_batch((uint32_t)n, (const void *const *)mem, bytes, fr, fs, n_aux, types, new_read_group, drop_aux, out, out_len, NULL);
    } else {
        /* ASCII -> ASCII: the reference parses and prints again; through BLOW5 (none, none) gives the same canonical text */
        void **mid = (void **)calloc((size_t)n, sizeof *mid);
        size_t *mid_len = (size_t *)calloc((size_t)n, sizeof *mid_len);
        if (!mid || !mid_len) { free(mid); free(mid_len); slow5_errno = SLOW5_ERR_MEM; return -1; }
        rc = s5gpu_ascii_to_blow5_batch((uint32_t)n, (const char *const *)mem, bytes, n_aux, types, S5GPU_REC_NONE, S5GPU_SIG_NONE, new_read_group,
                                        drop_aux, mid, mid_len, NULL);
        if (rc == S5GPU_OK) {
            for (int64_t i = 0; i < n; i++) { memmove(mid[i], (char *)mid[i] + 8, mid_len[i] - 8); mid_len[i] -= 8; }   /* drop the u64 prefix */
            rc = s5gpu_blow5_to_ascii_batch((uint32_t)n, (const void *const *)mid, mid_len, S5GPU_REC_NONE, S5GPU_SIG_NONE, drop_aux ? 0 : n_aux, types, NULL,
                                            0, out, out_len, NULL);
        }
        for (int64_t i = 0; i < n; i++) free(mid[i]);
        free(mid); free(mid_len);
    }
    if (rc != S5GPU_OK) { slow5_errno = SLOW5_ERR_RECPARSE; return -1; }
    for (int64_t i = 0; i < n; i++) { free(mem[i]); mem[i] = NULL; }
    return 0;
}

int slow5_gpu_recompress_batch(int64_t n, char **mem, size_t *bytes, slow5_press_method_t from, slow5_press_method_t to,
                               const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len) {
    const int fr = rec_code(from.record_method), fs = sig_code(from.signal_method);
    const int tr = rec_code(to.record_method), ts = sig_code(to.signal_method);
    if (n < 0 || fr < 0 || fs < 0 || tr < 0 || ts < 0) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
    if (n == 0) return 0;
    /* one device-resident pass: only compressed bytes cross PCIe (decoded signals stay in HBM) */
    if (s5gpu_recompress_batch((uint32_t)n, (const void *const *)mem, bytes, fr, fs, tr, ts, new_read_group, drop_aux, out, out_len, NULL) != S5GPU_OK) {
        slow5_errno = SLOW5_ERR_RECPARSE;
        return -1;
    }
    for (int64_t i = 0; i < n; i++) { free(mem[i]); mem[i] = NULL; }   /* the reference's worker frees the input record (src/view.c:41) */
    return 0;
}


/* ---- slow5gpu_hooks.h: the hooks without any slow5lib type in their signatures (a patched slow5tools includes that header
 * next to <slow5/slow5.h>).  Methods and formats arrive as ints holding slow5lib's enum values. ---- */
static int hook_method_ok(int rec, int sig) {
    return rec_code((enum slow5_press_method)rec) >= 0 && sig_code((enum slow5_press_method)sig) >= 0;
}
int slow5_gpu_hook_init(uint64_t dev_mask) { return s5gpu_init_mask(dev_mask) == S5GPU_OK ? 0 : -1; }
void slow5_gpu_hook_shutdown(void) { s5gpu_shutdown(); }
const char *slow5_gpu_hook_error(void) { return s5gpu_last_error(); }

int slow5_gpu_hook_recompress(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method, int to_record_method,
                              int to_signal_method, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len) {
    slow5_press_method_t from = {(enum slow5_press_method)from_record_method, (enum slow5_press_method)from_signal_method};
    slow5_press_method_t to = {(enum slow5_press_method)to_record_method, (enum slow5_press_method)to_signal_method};
    return slow5_gpu_recompress_batch(n, mem, bytes, from, to, new_read_group, drop_aux, out, out_len);
}

int slow5_gpu_hook_recompress_arena(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method, int to_record_method,
                                    int to_signal_method, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len, void **batch) {
    if (!batch) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    *batch = NULL;
    if (n < 0 || n > 0xFFFFFFFFll || !hook_method_ok(from_record_method, from_signal_method) || !hook_method_ok(to_record_method, to_signal_method)) {
        slow5_errno = SLOW5_ERR_PRESS;
        return -1;
    }
    if (n == 0) return 0;
    if (s5gpu_recompress_batch_arena((uint32_t)n, (const void *const *)mem, bytes, rec_code((enum slow5_press_method)from_record_method),
                                     sig_code((enum slow5_press_method)from_signal_method), rec_code((enum slow5_press_method)to_record_method),
                                     sig_code((enum slow5_press_method)to_signal_method), new_read_group, drop_aux, out, out_len, NULL, batch) != S5GPU_OK) {
        slow5_errno = SLOW5_ERR_RECPARSE;
        return -1;
    }
    for (int64_t i = 0; i < n; i++) { free(mem[i]); mem[i] = NULL; }   /* the reference's worker frees the input record (src/view.c:41) */
    return 0;
}
void slow5_gpu_hook_release(void *batch) { s5gpu_arena_release(batch); }

/* the same batch in flight (slow5gpu_hooks.h): the ticket remembers the input records the worker frees once the batch is through */
struct hook_ticket { void *t; int64_t n; char **mem; };
void *slow5_gpu_hook_recompress_submit(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method, int to_record_method,
                                       int to_signal_method, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len) {
    if (n < 0 || n > 0xFFFFFFFFll || !hook_method_ok(from_record_method, from_signal_method) || !hook_method_ok(to_record_method, to_signal_method)) {
        slow5_errno = SLOW5_ERR_PRESS;
        return NULL;
    }
    struct hook_ticket *h = (struct hook_ticket *)calloc(1, sizeof *h);
    if (!h) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
    h->n = n; h->mem = mem;
    if (n) {
        h->t = s5gpu_recompress_batch_submit((uint32_t)n, (const void *const *)mem, bytes, rec_code((enum slow5_press_method)from_record_method),
                                             sig_code((enum slow5_press_method)from_signal_method), rec_code((enum slow5_press_method)to_record_method),
                                             sig_code((enum slow5_press_method)to_signal_method), new_read_group, drop_aux, out, out_len, NULL, 1);
        if (!h->t) { free(h); slow5_errno = SLOW5_ERR_MEM; return NULL; }
    }
    return h;
}
int slow5_gpu_hook_recompress_wait(void *ticket, void **batch) {
    if (batch) *batch = NULL;
    struct hook_ticket *h = (struct hook_ticket *)ticket;
    if (!h) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    int ret = 0;
    if (h->t && s5gpu_batch_wait(h->t, batch) != S5GPU_OK) { slow5_errno = SLOW5_ERR_RECPARSE; ret = -1; }
    if (ret == 0) for (int64_t i = 0; i < h->n; i++) { free(h->mem[i]); h->mem[i] = NULL; }   /* as the synchronous worker does (src/view.c:41) */
    free(h);
    return ret;
}

void *slow5_gpu_hook_alloc(size_t bytes) { return s5gpu_host_alloc(bytes); }
void slow5_gpu_hook_free(void *p) { s5gpu_host_free(p); }
int slow5_gpu_hook_recompress_chunk(int64_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len,
                                    int from_record_method, int from_signal_method, int to_record_method, int to_signal_method,
                                    const uint32_t *new_read_group, int drop_aux, void *out_buf, size_t out_cap, uint64_t *out_off) {
    if (n < 0 || n > 0xFFFFFFFFll || !hook_method_ok(from_record_method, from_signal_method) || !hook_method_ok(to_record_method, to_signal_method)) {
        slow5_errno = SLOW5_ERR_PRESS;
        return -1;
    }
    const int rc = s5gpu_recompress_stream((uint32_t)n, chunk, chunk_bytes, rec_pos, rec_len, rec_code((enum slow5_press_method)from_record_method),
                                           sig_code((enum slow5_press_method)from_signal_method), rec_code((enum slow5_press_method)to_record_method),
                                           sig_code((enum slow5_press_method)to_signal_method), new_read_group, drop_aux, out_buf, out_cap, out_off, NULL);
    if (rc != S5GPU_OK) { slow5_errno = rc == S5GPU_ERR_NOMEM ? SLOW5_ERR_MEM : SLOW5_ERR_RECPARSE; return -1; }
    return 0;
}

int slow5_gpu_hook_convert(int64_t n, char **mem, size_t *bytes, int from_fmt, int from_record_method, int from_signal_method,
                           const char *aux_types_line, int to_fmt, int to_record_method, int to_signal_method,
                           const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len) {
    slow5_press_method_t from = {(enum slow5_press_method)from_record_method, (enum slow5_press_method)from_signal_method};
    slow5_press_method_t to = {(enum slow5_press_method)to_record_method, (enum slow5_press_method)to_signal_method};
    uint8_t types[1024];
    struct slow5_aux_meta am = {0, types};
    if (aux_types_line) {
        const int k = s5gpu_aux_types_parse(aux_types_line, strlen(aux_types_line), types, sizeof types);
        if (k < 0) { slow5_errno = SLOW5_ERR_TYPE; return -1; }
        am.num = (uint32_t)k;
    }
    return slow5_gpu_convert_batch(n, mem, bytes, (enum slow5_fmt)from_fmt, from, am.num ? &am : NULL, (enum slow5_fmt)to_fmt, to, new_read_group,
                                   drop_aux, out, out_len);
}

int slow5_gpu_hook_depress_parse(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method, slow5_gpu_read_t *reads) {
    if (n < 0 || !hook_method_ok(from_record_method, from_signal_method)) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
    if (n == 0) return 0;
    void **pay = (void **)calloc((size_t)n, sizeof(void *));
    int16_t **sig = (int16_t **)calloc((size_t)n, sizeof(void *));
    s5gpu_rec_fields_t *f = (s5gpu_rec_fields_t *)calloc((size_t)n, sizeof *f);
    int ret = -1;
    if (!pay || !sig || !f) { slow5_errno = SLOW5_ERR_MEM; goto done; }
    if (s5gpu_decode_batch((uint32_t)n, (const void *const *)mem, bytes, rec_code((enum slow5_press_method)from_record_method),
                           sig_code((enum slow5_press_method)from_signal_method), pay, sig, f) != S5GPU_OK) { slow5_errno = SLOW5_ERR_RECPARSE; goto done; }
    for (int64_t i = 0; i < n; i++) {
        slow5_gpu_read_t *r = &reads[i];
        free(mem[i]);
        mem[i] = (char *)pay[i];           /* the uncompressed record; read_id / aux below point into it */
        bytes[i] = f[i].payload_len;
        pay[i] = NULL;
        r->read_id = mem[i] + 2;
        r->read_id_len = (uint16_t)f[i].read_id_len;
        r->read_group = f[i].read_group;
        r->digitisation = f[i].digitisation; r->offset = f[i].offset; r->range = f[i].range; r->sampling_rate = f[i].sampling_rate;
        r->len_raw_signal = f[i].n_samples;
        r->raw_signal = sig[i];
        sig[i] = NULL;
        r->aux = (const uint8_t *)mem[i] + f[i].aux_off;
        r->aux_len = f[i].aux_len;
    }
    ret = 0;
done:
    if (pay) for (int64_t i = 0; i < n; i++) free(pay[i]);
    if (sig) for (int64_t i = 0; i < n; i++) free(sig[i]);
    free(pay); free(sig); free(f);
    return ret;
}

static int hook_rec_to_mem_any(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method, void **out,
                               size_t *out_len, void **arena);
int slow5_gpu_hook_rec_to_mem(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method, void **out,
                              size_t *out_len) {
    return hook_rec_to_mem_any(n, reads, drop_aux, to_record_method, to_signal_method, out, out_len, NULL);
}
int slow5_gpu_hook_rec_to_mem_arena(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method, void **out,
                                    size_t *out_len, void **batch) {
    if (!batch) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    return hook_rec_to_mem_any(n, reads, drop_aux, to_record_method, to_signal_method, out, out_len, batch);
}
static int hook_rec_to_mem_any(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method, void **out,
                               size_t *out_len, void **arena) {
    if (arena) *arena = NULL;
    if (n < 0 || !hook_method_ok(to_record_method, to_signal_method)) { slow5_errno = SLOW5_ERR_PRESS; return -1; }
    if (n == 0) return 0;
    struct slow5_rec *recs = (struct slow5_rec *)calloc((size_t)n, sizeof *recs);
    struct slow5_rec **ptrs = (struct slow5_rec **)calloc((size_t)n, sizeof *ptrs);
    int ret = -1;
    if (!recs || !ptrs) { slow5_errno = SLOW5_ERR_MEM; goto done; }
    for (int64_t i = 0; i < n; i++) {     /* views, nothing is copied: the batch call only reads them */
        const slow5_gpu_read_t *r = &reads[i];
        recs[i].read_id_len = r->read_id_len; recs[i].read_id = (char *)r->read_id; recs[i].read_group = r->read_group;
        recs[i].digitisation = r->digitisation; recs[i].offset = r->offset; recs[i].range = r->range; recs[i].sampling_rate = r->sampling_rate;
        recs[i].len_raw_signal = r->len_raw_signal; recs[i].raw_signal = r->raw_signal;
        recs[i].aux_blob = (uint8_t *)r->aux; recs[i].aux_len = r->aux_len;
        ptrs[i] = &recs[i];
    }
    {
        slow5_press_method_t to = {(enum slow5_press_method)to_record_method, (enum slow5_press_method)to_signal_method};
        ret = rec_to_mem_batch_any(n, ptrs, drop_aux, to, out, out_len, arena);
    }
done:
    free(recs); free(ptrs);
    return ret;
}
