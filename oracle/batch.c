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

/* =====================================================================================================================
 * END-TO-END twins: the WHOLE loop of `slow5tools view` and `slow5tools get --benchmark` on files, for bench.py's `e2e` object.
 * Same shape as the reference: a serial read phase of K records (src/view.c:265-278: slow5_get_next_mem — one getline / one
 * fread + one malloc per record), the work_db compute phase above (src/view.c:292), a serial ordered write phase (src/view.c:296-299:
 * one fwrite + one free per record); the three phases one after the other per batch.  phases[0..2] = seconds spent in read /
 * compute / write, phases[3] = first read to last write.  TEST / BENCH infrastructure like the rest of this file.
 * ===================================================================================================================== */
#include <stdio.h>
#include <unistd.h>
#include <fcntl.h>

typedef struct {
    batch_t db;               /* db.out / db.out_len: the batch's outputs */
    char **mem;               /* the batch's input records (lines, or BLOW5 record bytes) */
    size_t *bytes;
    int from_blow5;
    uint8_t aux_types[256];   /* text input: the aux columns' types (the header's "#char*..." line) */
    int n_aux;
} view_db_t;

static void view_one(void *dbv, int32_t i) {
    view_db_t *v = (view_db_t *)dbv;            /* (batch_t is the first member) */
    batch_t *db = &v->db;
    db->out[i] = NULL;
    db->out_len[i] = 0;
    s5o_rec_t r;
    uint8_t *pay = NULL;
    size_t plen = 0;
    int from_sig = S5O_SIG_NONE;
    if (!v->from_blow5) {                       /* slow5_rec_depress_parse of an ASCII line */
        pay = (uint8_t *)malloc(v->bytes[i] + 128);
        plen = pay ? s5o_ascii_line_to_payload(v->mem[i], v->bytes[i], v->n_aux ? v->aux_types : NULL, (unsigned)v->n_aux, pay) : 0;
    } else {                                    /* ... of a zlib + svb-zd record */
        size_t cap = v->bytes[i] * 4 + 4096;
        for (int attempt = 0; attempt < 4 && !plen; attempt++) {
            pay = (uint8_t *)malloc(cap);
            size_t l = cap;
            if (pay && s5o_zlib_decompress((const uint8_t *)v->mem[i], v->bytes[i], pay, &l) == 0) { plen = l; break; }
            free(pay); pay = NULL; cap *= 4;
        }
        from_sig = S5O_SIG_SVB_ZD;
    }
    if (!plen || s5o_rec_parse(pay, plen, from_sig, &r, NULL) != 0) { free(pay); db->failed = 1; return; }
    int16_t *sig = (int16_t *)malloc(r.len_raw_signal ? 2 * (size_t)r.len_raw_signal : 2);
    if (!sig || s5o_rec_parse(pay, plen, from_sig, &r, sig) != 0) { free(pay); free(sig); db->failed = 1; return; }
    uint8_t *scratch = (uint8_t *)malloc(s5o_payload_bound(&r, db->sig_method));
    uint8_t *out = (uint8_t *)malloc(s5o_rec_to_mem_bound(&r, db->sig_method));
    db->out_len[i] = s5o_rec_to_mem(&r, db->rec_method, db->sig_method, scratch, out);
    db->out[i] = out;
    free(scratch); free(sig); free(pay);
}

/* in: a .slow5 (text) file or a BLOW5 file with zlib + svb-zd records (sniffed); out: BLOW5, zlib + svb-zd.  max_reads = 0: the whole file.
 * Returns the number of records written, 0 on failure. */
uint64_t s5o_view_file(const char *in_path, const char *out_path, int n_threads, int batch_size, uint64_t max_reads, double phases[4]) {
    FILE *in = fopen(in_path, "rb"), *out = fopen(out_path, "wb");
    if (!in || !out) { if (in) fclose(in); if (out) fclose(out); return 0; }
    static char ibuf[1 << 20], obuf[1 << 20];
    setvbuf(in, ibuf, _IOFBF, sizeof ibuf);
    setvbuf(out, obuf, _IOFBF, sizeof obuf);
    view_db_t V;
    memset(&V, 0, sizeof V);
    uint8_t head[64];
    char *hdr_text = NULL;
    size_t hdr_len = 0;
    int c0 = fgetc(in);
    ungetc(c0, in);
    V.from_blow5 = c0 == 'B';
    if (V.from_blow5) {
        uint32_t hl;
        if (fread(head, 1, 64, in) != 64 || fread(&hl, 4, 1, in) != 1) goto fail;
        hdr_text = (char *)malloc(hl ? hl : 1);
        if (fread(hdr_text, 1, hl, in) != hl) goto fail;
        hdr_len = hl;
    } else {
        /* header lines (#..., @...) up to the first record line; #slow5_version / #num_read_groups go into the binary head */
        memset(head, 0, 64);
        memcpy(head, "BLOW5\1", 6);
        head[6] = 0; head[7] = 2; head[8] = 0;
        uint32_t nrg = 1;
        size_t cap = 0;
        char *line = NULL;
        for (;;) {
            int c = fgetc(in);
            if (c == EOF) break;
            ungetc(c, in);
            if (c != '#' && c != '@') break;
            ssize_t l = getline(&line, &cap, in);
            if (l <= 0) break;
            if (strncmp(line, "#slow5_version\t", 15) == 0) { int a = 0, b = 0, c2 = 0; sscanf(line + 15, "%d.%d.%d", &a, &b, &c2); head[6] = (uint8_t)a; head[7] = (uint8_t)b; head[8] = (uint8_t)c2; continue; }
            if (strncmp(line, "#num_read_groups\t", 17) == 0) { nrg = (uint32_t)strtoul(line + 17, NULL, 10); continue; }
            if (strncmp(line, "#char*\t", 7) == 0) { const int na = s5o_aux_types(line, (size_t)l, V.aux_types, sizeof V.aux_types); V.n_aux = na > 0 ? na : 0; }
            hdr_text = (char *)realloc(hdr_text, hdr_len + (size_t)l);
            memcpy(hdr_text + hdr_len, line, (size_t)l);
            hdr_len += (size_t)l;
        }
        free(line);
        memcpy(head + 10, &nrg, 4);
    }
    head[9] = 1;  /* record press zlib */
    head[14] = 1; /* signal press svb-zd */
    {
        uint32_t hl = (uint32_t)hdr_len;
        fwrite(head, 1, 64, out); fwrite(&hl, 4, 1, out); fwrite(hdr_text, 1, hdr_len, out);
    }
    V.db.rec_method = S5O_REC_ZLIB;
    V.db.sig_method = S5O_SIG_SVB_ZD;
    V.db.one = view_one;
    V.db.out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)batch_size);
    V.db.out_len = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    V.mem = (char **)malloc(sizeof(char *) * (size_t)batch_size);
    V.bytes = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    uint64_t total = 0;
    double tr = 0, tc = 0, tw = 0;
    const double t_first = now_s();
    int eof = 0;
    while (!eof && (!max_reads || total < max_reads)) {
        double t0 = now_s();
        int64_t n = 0;
        while (n < batch_size && (!max_reads || total + (uint64_t)n < max_reads)) {     /* read phase */
            if (V.from_blow5) {
                uint64_t sz;
                if (fread(&sz, 8, 1, in) != 1 || memcmp(&sz, "5WOLB", 5) == 0) { eof = 1; break; }
                V.mem[n] = (char *)malloc((size_t)sz ? (size_t)sz : 1);
                if (fread(V.mem[n], 1, (size_t)sz, in) != (size_t)sz) { free(V.mem[n]); eof = 1; break; }
                V.bytes[n] = (size_t)sz;
            } else {
                char *line = NULL;
                size_t cap = 0;
                ssize_t l = getline(&line, &cap, in);
                if (l <= 0) { free(line); eof = 1; break; }
                if (l && line[l - 1] == '\n') l--;
                V.mem[n] = line;
                V.bytes[n] = (size_t)l;
            }
            n++;
        }
        if (n == 0) break;
        double t1 = now_s();
        V.db.n_batch = n;
        work_batch(&V.db, n_threads);                                                   /* compute phase */
        double t2 = now_s();
        for (int64_t i = 0; i < n; i++) {                                               /* ordered write phase */
            if (V.db.out[i]) fwrite(V.db.out[i], 1, V.db.out_len[i], out);
            free(V.db.out[i]);
            free(V.mem[i]);
        }
        double t3 = now_s();
        tr += t1 - t0; tc += t2 - t1; tw += t3 - t2;
        total += (uint64_t)n;
    }
    fwrite("5WOLB", 1, 5, out);
    fflush(out);
    const double t_last = now_s();
    if (phases) { phases[0] = tr; phases[1] = tc; phases[2] = tw; phases[3] = t_last - t_first; }
    free(V.db.out); free(V.db.out_len); free(V.mem); free(V.bytes); free(hdr_text);
    fclose(in); fclose(out);
    return V.db.failed ? 0 : total;
fail:
    free(hdr_text); fclose(in); fclose(out);
    return 0;
}

/* `get --benchmark -t T -K B` on a file (src/get.c:52, 321-386): per id the worker itself preads the record (slow5_get does), inflates,
 * parses and decodes it; nothing is written.  pos / len: file extents of the records to fetch ([u64 size][bytes]: pos points at the size
 * prefix), in fetch order.  Returns samples decoded (0 on failure); *secs = first batch to last. */
typedef struct { batch_t db; int fd; const uint64_t *pos; const uint32_t *len; } get_db_t;
static void get_one(void *dbv, int32_t i) {
    get_db_t *g = (get_db_t *)dbv;
    batch_t *db = &g->db;
    const uint64_t k = db->base + (uint64_t)i;
    db->out[i] = NULL;
    db->out_len[i] = 0;
    const size_t zl = g->len[k] - 8;
    uint8_t *rec = (uint8_t *)malloc(g->len[k]);
    if (!rec || pread(g->fd, rec, g->len[k], (off_t)g->pos[k]) != (ssize_t)g->len[k]) { free(rec); db->failed = 1; return; }
    size_t cap = zl * 4 + 4096, plen = 0;
    uint8_t *pay = NULL;
    for (int attempt = 0; attempt < 4 && !plen; attempt++) {
        pay = (uint8_t *)malloc(cap);
        size_t l = cap;
        if (pay && s5o_zlib_decompress(rec + 8, zl, pay, &l) == 0) { plen = l; break; }
        free(pay); pay = NULL; cap *= 4;
    }
    free(rec);
    s5o_rec_t r;
    if (!plen || s5o_rec_parse(pay, plen, db->sig_method, &r, NULL) != 0) { free(pay); db->failed = 1; return; }
    int16_t *sig = (int16_t *)malloc(r.len_raw_signal ? 2 * (size_t)r.len_raw_signal : 2);
    if (!sig || s5o_rec_parse(pay, plen, db->sig_method, &r, sig) != 0) { free(pay); free(sig); db->failed = 1; return; }
    free(pay);
    db->out[i] = (uint8_t *)sig;
    db->out_len[i] = (size_t)r.len_raw_signal;
}
uint64_t s5o_get_file(const char *path, const uint64_t *pos, const uint32_t *len, uint64_t n_ids, int n_threads, int batch_size, double *secs) {
    get_db_t G;
    memset(&G, 0, sizeof G);
    G.fd = open(path, O_RDONLY);
    if (G.fd < 0) return 0;
    G.pos = pos; G.len = len;
    G.db.rec_method = S5O_REC_ZLIB;
    G.db.sig_method = S5O_SIG_SVB_ZD;
    G.db.one = get_one;
    G.db.out = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)batch_size);
    G.db.out_len = (size_t *)malloc(sizeof(size_t) * (size_t)batch_size);
    uint64_t total = 0;
    const double t0 = now_s();
    for (uint64_t base = 0; base < n_ids; base += (uint64_t)batch_size) {
        G.db.base = base;
        G.db.n_batch = (int64_t)(n_ids - base < (uint64_t)batch_size ? n_ids - base : (uint64_t)batch_size);
        work_batch(&G.db, n_threads);
        for (int64_t i = 0; i < G.db.n_batch; i++) { total += G.db.out_len[i]; free(G.db.out[i]); }
    }
    if (secs) *secs = now_s() - t0;
    free(G.db.out); free(G.db.out_len);
    close(G.fd);
    return G.db.failed ? 0 : total;
}
