/*
 * batch.c — ORACLE-side CPU baseline with the reference's threading shape.  TEST/BENCH INFRASTRUCTURE.
 *
 * Restates the compute phase of `slow5tools view -t T -K B` for SLOW5->BLOW5 (zlib + svb-zd):
 *   - batch loop of B records (src/view.c:254-300; default B = 4096, src/cmd.h:8)
 *   - work_db: static block partition over T pthreads + single-item work stealing
 *     (src/thread.c:19-37 steal_work, :40-67 pthread_single, :69-111 pthread_db), threads created and
 *     joined per batch (src/thread.c:100-110), serial when T == 1 (src/thread.c:116-121)
 *   - per record (src/view.c:35-57): codec state allocated and freed per record, one malloc'd
 *     output buffer per record, freed after the ordered "write" (src/view.c:296-299)
 * What is timed is the work_db analogue only (time_depress_parse, src/view.c:293,319): the serial
 * read/write phases are excluded, which favours the CPU.  The decode half of the callback
 * (slow5_rec_depress_parse of an ASCII/BLOW5 input) is also excluded: input is int16 in memory.
 */
#define _GNU_SOURCE
#include "s5oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    const int16_t *sig;
    uint64_t n_samples, first_idx, base;
    int rec_method, sig_method;
    int64_t n_batch;
    uint8_t **out;
    size_t *out_len;
    /* decode batches (s5o_decode_batch_mt): records of a BLOW5 stream picked through an id list */
    const uint8_t *stream;
    const uint64_t *rec_off;
    const uint32_t *ids;
    /* SLOW5 text batches (s5o_convert_ascii_batch_mt): record lines of a .slow5 file */
    const char *text;
    const uint64_t *line_off;      /* n + 1 offsets: line i = text[line_off[i], line_off[i + 1]) incl. its newline */
    int failed;
    void (*one)(void *db, int32_t i);
} batch_t;

typedef struct worker {
    batch_t *db;
    volatile int32_t starti;
    int32_t endi;
    struct worker *all;
    int32_t n_threads;
} worker_t;

static void encode_one_(batch_t *db, int32_t i);
static void encode_one(batch_t *db, int32_t i) {
    if (db->one) db->one(db, i); else encode_one_(db, i);
}
static void encode_one_(batch_t *db, int32_t i) {
    s5o_rec_t r;
    char id[37];
    uint64_t ridx = db->first_idx + db->base + (uint64_t)i;
    s5o_synth_read_id(ridx, id);
    r.read_id_len = 36;
    r.read_id = id;
    r.read_group = 0;
    r.digitisation = 8192.0;
    r.offset = 23.0;
    r.range = 1467.61;
    r.sampling_rate = 4000.0;
    r.len_raw_signal = db->n_samples;
    r.raw_signal = db->sig + (db->base + (uint64_t)i) * db->n_samples;
    r.aux = NULL;
    r.aux_len = 0;
    uint8_t *scratch = (uint8_t *)malloc(s5o_payload_bound(&r, db->sig_method));
    uint8_t *out = (uint8_t *)malloc(s5o_rec_to_mem_bound(&r, db->sig_method));
    db->out_len[i] = s5o_rec_to_mem(&r, db->rec_method, db->sig_method, scratch, out);
    db->out[i] = out;
    free(scratch);
}

static int32_t steal(worker_t *all, int32_t n) {
    int32_t c = -1;
    for (int32_t t = 0; t < n; t++)
        if (all[t].endi - all[t].starti > 1) { c = t; break; }
    if (c < 0) return -1;
    int32_t k = __sync_fetch_and_add(&all[c].starti, 1);
    return k >= all[c].endi ? -1 : k;
}

static void *worker_main(void *arg) {
    worker_t *w = (worker_t *)arg;
    int32_t i;
    for (;;) {
        i = __sync_fetch_and_add(&w->starti, 1);
        if (i >= w->endi) break;
        encode_one(w->db, i);
    }
    while ((i = steal(w->all, w->n_threads)) >= 0) encode_one(w->db, i);
    s5o_zlib_pool_release();          /* (threads are created per batch: a pooled deflate state lives as long as its thread) */
    return NULL;
}

static void work_batch(batch_t *db, int n_threads) {
    if (n_threads <= 1) {
        for (int32_t i = 0; i < db->n_batch; i++) encode_one(db, i);
        return;
    }
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    worker_t *ws = (worker_t *)malloc(sizeof(worker_t) * (size_t)n_threads);
    int32_t step = (int32_t)((db->n_batch + n_threads - 1) / n_threads), i = 0;
    for (int t = 0; t < n_threads; t++) {
        ws[t].db = db;
        ws[t].starti = i;
        i += step;
        ws[t].endi = i > db->n_batch ? (int32_t)db->n_batch : i;
        if (ws[t].starti > ws[t].endi) ws[t].starti = ws[t].endi;
        ws[t].all = ws;
        ws[t].n_threads = n_threads;
    }
    for (int t = 0; t < n_threads; t++) pthread_create(&tids[t], NULL, worker_main, &ws[t]);
    for (int t = 0; t < n_threads; t++) pthread_join(tids[t], NULL);
    free(ws);
    free(tids);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

uint64_t s5o_encode_batch_mt(const int16_t *sig, uint64_t n_reads, uint64_t n_samples, uint64_t first_read_idx,
                             int rec_method, int sig_method, int n_threads, int batch_size, double *secs,
                             uint64_t *checksum) {
    batch_t db;
    memset(&db, 0, sizeof db);
    db.sig = sig;
    db.n_samples = n_samples;
    db.first_idx = first_read_idx;
    db.rec_method = rec_method;
    db.sig_method = sig_method;
    db.out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)batch_size);
    db.out_len = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    uint64_t total = 0, ck = 0;
    double t = 0;
    for (uint64_t base = 0; base < n_reads; base += (uint64_t)batch_size) {
        db.base = base;
        db.n_batch = (int64_t)(n_reads - base < (uint64_t)batch_size ? n_reads - base : (uint64_t)batch_size);
        double t0 = now_s();
        work_batch(&db, n_threads);
        t += now_s() - t0;
        for (int64_t i = 0; i < db.n_batch; i++) { /* the ordered fwrite analogue (untimed) */
            total += db.out_len[i];
            ck = ck * 1000003ull + s5o_adler32(db.out[i], db.out_len[i]);
            free(db.out[i]);
        }
    }
    free(db.out);
    free(db.out_len);
    if (secs) *secs = t;
    if (checksum) *checksum = ck;
    return total;
}

/* ---- decode twin: the compute phase of `slow5tools get --benchmark -t T -K B` (src/get.c:52: slow5_get only, no
 * re-encode): per id  inflate (inflateInit/inflate/inflateEnd per record) -> parse -> svb-zd decode into a malloc'd
 * int16 buffer that is freed after the batch, under the same work_db shape.  The preads of slow5_get are excluded
 * (the records are in memory), which favours the CPU.  stream: BLOW5 records [u64 size][zlib stream] back to back,
 * rec_off[i] = byte offset of record i; ids: which records to fetch, in order.  Returns the number of samples decoded. */
static void decode_one(void *dbv, int32_t i) {
    batch_t *db = (batch_t *)dbv;
    const uint32_t id = db->ids[db->base + (uint64_t)i];
    const uint8_t *rec = db->stream + db->rec_off[id];
    uint64_t zlen;
    memcpy(&zlen, rec, 8);
    size_t cap = (size_t)zlen * 4 + 4096, plen = 0;
    uint8_t *pay = NULL;
    for (int attempt = 0; attempt < 4; attempt++) {
        pay = (uint8_t *)malloc(cap);
        plen = cap;
        if (db->rec_method == S5O_REC_ZLIB) {
            if (s5o_zlib_decompress(rec + 8, (size_t)zlen, pay, &plen) == 0) break;
        } else {
            if (zlen <= cap) { memcpy(pay, rec + 8, (size_t)zlen); plen = (size_t)zlen; break; }
        }
        free(pay);
        pay = NULL;
        cap *= 4;
    }
    db->out[i] = NULL;
    db->out_len[i] = 0;
    if (!pay) { db->failed = 1; return; }
    s5o_rec_t r;
    if (s5o_rec_parse(pay, plen, db->sig_method, &r, NULL) != 0) { free(pay); db->failed = 1; return; }
    int16_t *sig = (int16_t *)malloc(r.len_raw_signal ? 2 * (size_t)r.len_raw_signal : 2);
    if (s5o_rec_parse(pay, plen, db->sig_method, &r, sig) != 0) { free(pay); free(sig); db->failed = 1; return; }
    free(pay);
    db->out[i] = (uint8_t *)sig;
    db->out_len[i] = (size_t)r.len_raw_signal;
}

uint64_t s5o_decode_batch_mt(const uint8_t *stream, const uint64_t *rec_off, const uint32_t *ids, uint64_t n_ids,
                             int rec_method, int sig_method, int n_threads, int batch_size, double *secs,
                             uint64_t *checksum) {
    batch_t db;
    memset(&db, 0, sizeof db);
    db.stream = stream;
    db.rec_off = rec_off;
    db.ids = ids;
    db.rec_method = rec_method;
    db.sig_method = sig_method;
    db.one = decode_one;
    db.out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)batch_size);
    db.out_len = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    uint64_t total = 0, ck = 0;
    double t = 0;
    for (uint64_t base = 0; base < n_ids; base += (uint64_t)batch_size) {
        db.base = base;
        db.n_batch = (int64_t)(n_ids - base < (uint64_t)batch_size ? n_ids - base : (uint64_t)batch_size);
        double t0 = now_s();
        work_batch(&db, n_threads);
        t += now_s() - t0;
        for (int64_t i = 0; i < db.n_batch; i++) {
            total += db.out_len[i];
            if (db.out[i] && db.out_len[i]) {   /* cheap fingerprint: first, middle and last sample */
                const int16_t *sg = (const int16_t *)db.out[i];
                ck = ck * 1000003ull + (uint16_t)sg[0] + ((uint64_t)(uint16_t)sg[db.out_len[i] / 2] << 16) + ((uint64_t)(uint16_t)sg[db.out_len[i] - 1] << 32);
            }
            free(db.out[i]);
        }
    }
    free(db.out);
    free(db.out_len);
    if (secs) *secs = t;
    if (checksum) *checksum = ck;
    return db.failed ? 0 : total;
}

/* ---- SLOW5 text in: the whole worker of `slow5tools view in.slow5 -o out.blow5` (src/view.c:35-57, BASELINE configs[0]): per record
 * slow5_rec_depress_parse of the ASCII line (strtol per sample: ascii.c) -> slow5_press_init -> slow5_rec_to_mem (svb-zd + zlib) ->
 * free, under the same work_db shape.  text / line_off: record lines back to back.  Returns total output bytes. */
static void convert_ascii_one(void *dbv, int32_t i) {
    batch_t *db = (batch_t *)dbv;
    const uint64_t k = db->base + (uint64_t)i;
    const char *line = db->text + db->line_off[k];
    const size_t len = (size_t)(db->line_off[k + 1] - db->line_off[k]);
    db->out[i] = NULL;
    db->out_len[i] = 0;
    uint8_t *pay = (uint8_t *)malloc(len + 128);                  /* 2 B/sample binary never exceeds the text that printed it */
    const size_t plen = pay ? s5o_ascii_line_to_payload(line, len, NULL, 0, pay) : 0;
    s5o_rec_t r;
    if (!plen || s5o_rec_parse(pay, plen, S5O_SIG_NONE, &r, NULL) != 0) { free(pay); db->failed = 1; return; }
    int16_t *sig = (int16_t *)malloc(r.len_raw_signal ? 2 * (size_t)r.len_raw_signal : 2);
    if (!sig || s5o_rec_parse(pay, plen, S5O_SIG_NONE, &r, sig) != 0) { free(pay); free(sig); db->failed = 1; return; }
    uint8_t *scratch = (uint8_t *)malloc(s5o_payload_bound(&r, db->sig_method));
    uint8_t *out = (uint8_t *)malloc(s5o_rec_to_mem_bound(&r, db->sig_method));
    db->out_len[i] = s5o_rec_to_mem(&r, db->rec_method, db->sig_method, scratch, out);
    db->out[i] = out;
    free(scratch);
    free(sig);
    free(pay);
}

uint64_t s5o_convert_ascii_batch_mt(const char *text, const uint64_t *line_off, uint64_t n_lines, int rec_method, int sig_method,
                                    int n_threads, int batch_size, double *secs, uint64_t *checksum) {
    batch_t db;
    memset(&db, 0, sizeof db);
    db.text = text;
    db.line_off = line_off;
    db.rec_method = rec_method;
    db.sig_method = sig_method;
    db.one = convert_ascii_one;
    db.out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)batch_size);
    db.out_len = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    uint64_t total = 0, ck = 0;
    double t = 0;
    for (uint64_t base = 0; base < n_lines; base += (uint64_t)batch_size) {
        db.base = base;
        db.n_batch = (int64_t)(n_lines - base < (uint64_t)batch_size ? n_lines - base : (uint64_t)batch_size);
        double t0 = now_s();
        work_batch(&db, n_threads);
        t += now_s() - t0;
        for (int64_t i = 0; i < db.n_batch; i++) {
            total += db.out_len[i];
            if (db.out[i]) ck = ck * 1000003ull + s5o_adler32(db.out[i], db.out_len[i]);
            free(db.out[i]);
        }
    }
    free(db.out);
    free(db.out_len);
    if (secs) *secs = t;
    if (checksum) *checksum = ck;
    return db.failed ? 0 : total;
}
