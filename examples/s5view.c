/*
 * s5view.c — the batch loop of `slow5tools view` (SLOW5 / BLOW5 -> SLOW5 / BLOW5) on the GPU press path.
 *
 * Not a CLI re-implementation: this is slow5_convert_parallel (/root/reference/src/view.c:241-323) with the
 * work_db() call at src/view.c:292 replaced by ONE slow5_gpu_convert_batch() per batch, written against
 * include/slow5_compat.h only.  It doubles as the end-to-end harness of tests/test_container.py.
 *
 *   s5view in.[b|s]low5 out.[b|s]low5 [record: none|zlib|zstd] [signal: none|svb-zd|ex-zd] [batch K]   (defaults zlib svb-zd 4096,
 *        src/misc.c:54-58, src/cmd.h:8; the input format is sniffed, the output format follows the extension as in
 *        src/view.c:170-190; press methods are ignored for a .slow5 output).  A 6th argument sets the number of GPU worker
 *        threads of the read || GPU || write pipeline (default 1); 0 runs the reference's serial read / compute / write phases.
 *        BLOW5 -> BLOW5 and SLOW5 -> BLOW5 take the chunked pipeline (no malloc / memcpy per record; S5VIEW_CHUNK_MB, S5VIEW_READERS
 *        tune it, S5VIEW_PER_RECORD=1 forces the per-record pipeline the tests compare it with).
 *   s5view --index in.blow5          writes in.blow5.idx (slow5tools index)
 *   s5view --get in.blow5 read_id    prints len_raw_signal and the first samples of one read (slow5tools get)
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "slow5_compat.h"
#include "slow5gpu.h"

/* S5VIEW_TIMING=1: where the wall time of the whole process goes (stderr, seconds since main() was entered) */
static double g_t_main;
static int g_timing;
static double now_s(void);
static void stamp(const char *what) { if (g_timing) fprintf(stderr, "s5view[t] %8.3f  %s\n", now_s() - g_t_main, what); }

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int die(const char *what) {
    fprintf(stderr, "s5view: %s (slow5_errno %d; %s)\n", what, slow5_errno, s5gpu_last_error());
    return EXIT_FAILURE;
}


/* ---- the three-stage pipeline (read || GPU || write) ---- */
#define NSLOT 4
enum { ST_EMPTY = 0, ST_FILLED, ST_BUSY, ST_DONE };
typedef struct {
    int state;
    int64_t seq, n;
    char **mem;
    size_t *bytes;
    void **bufs;
    size_t *lens;
} slot_t;
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    slot_t slot[NSLOT];
    slow5_file_t *in;
    int64_t K, next_work, total_batches;      /* total_batches < 0 until the reader has seen the end */
    slow5_press_method_t from, to;
    enum slow5_fmt fmt_out;
    int failed;
    char why[256];
} pipe_t;

static void pipe_fail(pipe_t *P, const char *what) {
    pthread_mutex_lock(&P->mu);
    if (!P->failed) { P->failed = 1; snprintf(P->why, sizeof P->why, "%s (slow5_errno %d; %s)", what, slow5_errno, s5gpu_last_error()); }
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}

static void *reader_main(void *arg) {                               /* read phase, src/view.c:265-278 */
    pipe_t *P = (pipe_t *)arg;
    for (int64_t s = 0;; s++) {
        slot_t *b = &P->slot[s % NSLOT];
        pthread_mutex_lock(&P->mu);
        while (!P->failed && b->state != ST_EMPTY) pthread_cond_wait(&P->cv, &P->mu);
        const int stop = P->failed;
        pthread_mutex_unlock(&P->mu);
        if (stop) return NULL;
        int64_t n = 0;
        int eof = 0;
        while (n < P->K) {
            b->mem[n] = (char *)slow5_get_next_mem(&b->bytes[n], P->in);
            if (!b->mem[n]) {
                if (slow5_errno != SLOW5_ERR_EOF) { pipe_fail(P, "bad record framing"); return NULL; }
                eof = 1;
                break;
            }
            n++;
        }
        pthread_mutex_lock(&P->mu);
        if (n) { b->n = n; b->seq = s; b->state = ST_FILLED; }
        if (eof || n == 0) P->total_batches = s + (n ? 1 : 0);
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        if (eof || n == 0) return NULL;
    }
}

static void *worker_main(void *arg) {                               /* compute phase: the work_db() of src/view.c:292 */
    pipe_t *P = (pipe_t *)arg;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        slot_t *b;
        int64_t s;
        for (;;) {
            s = P->next_work;
            b = &P->slot[s % NSLOT];
            if (P->failed || (P->total_batches >= 0 && s >= P->total_batches)) { pthread_mutex_unlock(&P->mu); return NULL; }
            if (b->state == ST_FILLED && b->seq == s) break;
            pthread_cond_wait(&P->cv, &P->mu);
        }
        P->next_work = s + 1;
        b->state = ST_BUSY;
        pthread_mutex_unlock(&P->mu);
        if (slow5_gpu_convert_batch(b->n, b->mem, b->bytes, P->in->format, P->from, P->in->header->aux_meta, P->fmt_out, P->to, NULL, 0, b->bufs,
                                    b->lens) != 0) {
            pipe_fail(P, "GPU press path failed");
            return NULL;
        }
        pthread_mutex_lock(&P->mu);
        b->state = ST_DONE;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

/* ---- BLOW5 -> BLOW5 without a malloc or a memcpy per record ----
 * The reference's loop spends its read phase in slow5_get_next_mem — one fread and one malloc per record (src/view.c:265-278) —
 * and its write phase in one fwrite and one free per record (src/view.c:296-299); with the compute on the GPU those two phases
 * are all that is left.  Here the reader takes the file in chunks of tens of MB straight into pinned memory (several pread
 * threads), frames the records in place (their size prefixes chain through the chunk; a record cut by the chunk's end is
 * carried to the next chunk), the GPU worker hands the whole chunk over in one call and gets the re-encoded records back as
 * one contiguous stream, and the writer issues one write() per chunk.  Same three stages, same order of records. */
#define FSLOT_MAX 16
typedef struct {
    int state;
    int64_t seq;
    uint8_t *in, *out;
    size_t in_have, out_cap, out_total;
    uint32_t n, cap;
    uint64_t *rec_pos, *out_off;
    uint32_t *rec_len;
    uint32_t *nl, nl_cap;       /* SLOW5 text: newline positions found by the pread threads (16 equal shares) */
    int ready;                  /* its buffers are allocated (slots behind the first are pinned while the first chunk is already on its way) */
} fslot_t;
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    fslot_t slot[FSLOT_MAX];
    int nslot;                   /* chunk slots in flight (S5VIEW_SLOTS, default 6) */
    double t_read, t_frame, t_rwait, t_gpu, t_write, t_wwait;   /* S5VIEW_TIMING: where the stages spend their time */
    int fd_in, fd_out;
    uint64_t pos, end;           /* record area of the input file: [pos, end) */
    size_t chunk;
    int64_t next_work, total_batches;
    slow5_press_method_t from, to;
    int failed, readers, oversize;
    int ascii;                   /* the input is a .slow5 file: the chunk is framed into lines, not [u64 size][bytes] records */
    int ascii_out;               /* the output is a .slow5 file (BLOW5 input): every chunk comes back as one block of text lines */
    uint32_t n_aux;
    const uint8_t *aux_type;
    char why[256];
    uint64_t records;
} fpipe_t;
static int rec_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_ZLIB ? S5GPU_REC_ZLIB : m == SLOW5_COMPRESS_ZSTD ? S5GPU_REC_ZSTD : S5GPU_REC_NONE; }
static int sig_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_SVB_ZD ? S5GPU_SIG_SVB_ZD : m == SLOW5_COMPRESS_EX_ZD ? S5GPU_SIG_EX_ZD : S5GPU_SIG_NONE; }
static void fpipe_fail(fpipe_t *P, const char *what) {
    pthread_mutex_lock(&P->mu);
    if (!P->failed) { P->failed = 1; snprintf(P->why, sizeof P->why, "%s (%s)", what, s5gpu_last_error()); }
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}
/* one part of a chunk: read it, and for SLOW5 text note where its newlines are while the bytes are still in this core's cache
 * (the reader thread would otherwise scan the whole chunk again on its own: a third of its time) */
typedef struct { int fd; uint8_t *dst; size_t len; uint64_t off; int ok; const uint8_t *base; uint32_t *nl, nl_cap, nl_n; int nl_over; } pread_job_t;
static void *pread_main(void *arg) {
    pread_job_t *j = (pread_job_t *)arg;
    size_t got = 0;
    while (got < j->len) {
        ssize_t r = pread(j->fd, j->dst + got, j->len - got, (off_t)(j->off + got));
        if (r <= 0) { j->ok = 0; return NULL; }
        got += (size_t)r;
    }
    j->nl_n = 0; j->nl_over = 0;
    if (j->nl) {
        const uint8_t *q = j->dst, *e = j->dst + j->len;
        while (q < e && (q = (const uint8_t *)memchr(q, '\n', (size_t)(e - q)))) {
            if (j->nl_n == j->nl_cap) { j->nl_over = 1; break; }
            j->nl[j->nl_n++] = (uint32_t)(q - j->base);
            q++;
        }
    }
    j->ok = 1;
    return NULL;
}
/* the first newline at or behind p: in the carried bytes (scanned here), then in the parts' lists, in file order */
static const uint8_t *next_newline(const fslot_t *b, const pread_job_t *job, int used, size_t carry, size_t p, int *jt, uint32_t *jk) {
    if (p < carry) {
        const uint8_t *q = (const uint8_t *)memchr(b->in + p, '\n', carry - p);
        if (q) return q;
        p = carry;
    }
    while (*jt < used) {
        const pread_job_t *j = &job[*jt];
        const size_t lo = (size_t)(j->dst - b->in), hi = lo + j->len;
        if (p < hi) {
            if (j->nl_over || !j->nl) {                              /* (lines of a few bytes: the list overflowed) */
                const size_t from = p > lo ? p : lo;
                const uint8_t *q = (const uint8_t *)memchr(b->in + from, '\n', hi - from);
                if (q) return q;
            } else {
                while (*jk < j->nl_n && j->nl[*jk] < p) (*jk)++;
                if (*jk < j->nl_n) return b->in + j->nl[*jk];
            }
        }
        (*jt)++;
        *jk = 0;
    }
    return NULL;
}
static void *freader_main(void *arg) {
    fpipe_t *P = (fpipe_t *)arg;
    size_t carry = 0;
    const uint8_t *carry_from = NULL;
    for (int64_t s = 0;; s++) {
        fslot_t *b = &P->slot[s % P->nslot];
        const double tw0 = now_s();
        pthread_mutex_lock(&P->mu);
        while (!P->failed && (b->state != ST_EMPTY || !b->ready)) pthread_cond_wait(&P->cv, &P->mu);
        const int stop = P->failed;
        pthread_mutex_unlock(&P->mu);
        if (stop) return NULL;
        const double tr0 = now_s();
        P->t_rwait += tr0 - tw0;
        if (carry) memmove(b->in, carry_from, carry);            /* the record the previous chunk's end cut in two */
        size_t want = P->chunk - carry;
        if (want > P->end - P->pos) want = (size_t)(P->end - P->pos);
        pread_job_t job[16];
        int used = 0;
        {   /* the chunk in several pread threads: one thread copies out of the page cache at ~5 GB/s */
            pthread_t th[16];
            const int T = P->readers < 1 ? 1 : P->readers > 16 ? 16 : P->readers;
            const size_t part = (want / T + 4095) & ~(size_t)4095;
            const uint32_t nl_part = b->nl ? b->nl_cap / 16 : 0;
            for (int t = 0; t < T; t++) {
                const size_t lo = (size_t)t * part;
                if (lo >= want) break;
                job[t].fd = P->fd_in; job[t].dst = b->in + carry + lo; job[t].len = want - lo < part ? want - lo : part; job[t].off = P->pos + lo; job[t].ok = 0;
                job[t].base = b->in; job[t].nl = b->nl ? b->nl + (size_t)t * nl_part : NULL; job[t].nl_cap = nl_part;
                used++;
            }
            for (int t = 1; t < used; t++) pthread_create(&th[t], NULL, pread_main, &job[t]);
            if (used) pread_main(&job[0]);
            for (int t = 1; t < used; t++) pthread_join(th[t], NULL);
            for (int t = 0; t < used; t++) if (!job[t].ok) { fpipe_fail(P, "read failed"); return NULL; }
        }
        P->pos += want;
        const double tr1 = now_s();
        P->t_read += tr1 - tr0;
        if (s == 0) stamp("first chunk read");
        const size_t have = carry + want;
        size_t p = 0;
        uint32_t n = 0;
        const int file_done_ = P->pos >= P->end;
        /* SLOW5 text: the next newline comes from the parts' lists (in file order); the carried bytes in front of them, and any part
         * whose list overflowed (lines of a few bytes), are scanned here */
        int jt = 0;
        uint32_t jk = 0;
        while (P->ascii && p < have && n < b->cap) {                 /* one record per line; a line the chunk's end cut is carried */
            const uint8_t *nl = next_newline(b, job, used, carry, p, &jt, &jk);
            if (!nl && !file_done_) break;
            const size_t e = nl ? (size_t)(nl - b->in) : have;         /* (a last line without its newline) */
            size_t l = e - p;
            if (l && b->in[p + l - 1] == '\r') l--;
            if (l) { b->rec_pos[n] = p; b->rec_len[n] = (uint32_t)l; n++; }
            p = nl ? e + 1 : e;
        }
        if (P->ascii && n == 0 && have >= P->chunk) { P->oversize = 1; fpipe_fail(P, "a line larger than the chunk size"); return NULL; }
        while (!P->ascii && p + 8 <= have && n < b->cap) {
            uint64_t sz;
            memcpy(&sz, b->in + p, 8);
            if (sz > P->chunk - 8) { P->oversize = 1; fpipe_fail(P, "a record larger than the chunk size"); return NULL; }   /* main() redoes the file record by record */
            if (p + 8 + sz > have) break;
            b->rec_pos[n] = p + 8;
            b->rec_len[n] = (uint32_t)sz;
            n++;
            p += 8 + (size_t)sz;
        }
        carry = have - p;
        carry_from = b->in + p;
        /* the file is read to its end: what is left over is either whole records the slot had no descriptors for (tiny records:
         * more than `cap` of them in one chunk) — they are framed in the next round, which reads nothing — or a cut record */
        const int file_done = P->pos >= P->end;
        if (file_done && carry && n == 0) { fpipe_fail(P, "bad record framing"); return NULL; }
        const int last = file_done && carry == 0;
        b->in_have = p;
        P->t_frame += now_s() - tr1;
        pthread_mutex_lock(&P->mu);
        if (n) { b->n = n; b->seq = s; b->state = ST_FILLED; }
        if (last || n == 0) P->total_batches = s + (n ? 1 : 0);
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        if (last || n == 0) return NULL;
        if (carry) {   /* the next slot's head takes the cut record: it must still be readable when that slot is claimed */
            /* the bytes stay valid: this slot is not reused before the next one has been filled (at least two slots) */
        }
    }
}
static void *fworker_main(void *arg) {
    fpipe_t *P = (fpipe_t *)arg;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        fslot_t *b;
        int64_t s;
        for (;;) {
            s = P->next_work;
            b = &P->slot[s % P->nslot];
            if (P->failed || (P->total_batches >= 0 && s >= P->total_batches)) { pthread_mutex_unlock(&P->mu); return NULL; }
            if (b->state == ST_FILLED && b->seq == s) break;
            pthread_cond_wait(&P->cv, &P->mu);
        }
        P->next_work = s + 1;
        b->state = ST_BUSY;
        pthread_mutex_unlock(&P->mu);
        const double tg0 = now_s();
        for (int attempt = 0;; attempt++) {
            const int rc = P->ascii_out
                ? s5gpu_blow5_to_ascii_stream(b->n, b->in, b->in_have, b->rec_pos, b->rec_len, rec_code_of(P->from.record_method), sig_code_of(P->from.signal_method),
                                              P->n_aux, P->aux_type, NULL, 0, b->out, b->out_cap, b->out_off, NULL)
                : P->ascii
                ? s5gpu_ascii_to_blow5_stream(b->n, b->in, b->in_have, b->rec_pos, b->rec_len, P->n_aux, P->aux_type, rec_code_of(P->to.record_method),
                                              sig_code_of(P->to.signal_method), NULL, 0, b->out, b->out_cap, b->out_off, NULL)
                : s5gpu_recompress_stream(b->n, b->in, b->in_have, b->rec_pos, b->rec_len, rec_code_of(P->from.record_method), sig_code_of(P->from.signal_method),
                                          rec_code_of(P->to.record_method), sig_code_of(P->to.signal_method), NULL, 0, b->out, b->out_cap, b->out_off, NULL);
            if (rc == S5GPU_OK) break;
            if (rc == S5GPU_ERR_NOMEM && attempt == 0 && b->out_off[0] > b->out_cap) {   /* the output outgrew its buffer: bring a bigger one */
                const size_t need = (size_t)b->out_off[0] + (size_t)b->out_off[0] / 8;
                s5gpu_host_free(b->out);
                b->out = (uint8_t *)s5gpu_host_alloc(need);
                b->out_cap = b->out ? need : 0;
                if (b->out) continue;
            }
            fpipe_fail(P, "GPU press path failed");
            return NULL;
        }
        b->out_total = (size_t)b->out_off[b->n];
        if (s == 0) stamp("first chunk through the GPU call");
        pthread_mutex_lock(&P->mu);
        P->t_gpu += now_s() - tg0;
        b->state = ST_DONE;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}
/* ---- the ordered write phase, one chunk ----
 * One write() per chunk (S5VIEW_WRITERS = 1, the default).  Measured on the round-4 MI355X box into /dev/shm (tools/hw_probe/shm_write_probe.c):
 * one thread's write() 6.5 GB/s — a new file's pages are allocated and zeroed under the inode lock —, 4 pwrite threads on disjoint parts of
 * the chunk 7.1, 8 threads 3.5, ftruncate + mmap + memcpy 5.6 / 3.9: the page cache does not take a second writer, so 3.5 GB of records are
 * 0.54 s of any conversion.  S5VIEW_WRITERS > 1 / S5VIEW_WRITE_MODE = mmap keep the parallel forms for filesystems that do. */
typedef struct { int fd; const uint8_t *src; uint8_t *dst; size_t len; off_t off; int ok; } wjob_t;
static void *wjob_main(void *arg) {
    wjob_t *j = (wjob_t *)arg;
    j->ok = 1;
    if (j->dst) { memcpy(j->dst, j->src, j->len); return NULL; }
    size_t done = 0;
    while (done < j->len) {
        ssize_t w = pwrite(j->fd, j->src + done, j->len - done, j->off + (off_t)done);
        if (w <= 0) { j->ok = 0; return NULL; }
        done += (size_t)w;
    }
    return NULL;
}
static int write_chunk(int fd, const uint8_t *src, size_t len, off_t off, int writers, int use_mmap) {
    if (len == 0) return 0;
    if (off < 0) {                                                 /* not seekable (a pipe): the plain ordered write() */
        size_t done = 0;
        while (done < len) { ssize_t w = write(fd, src + done, len - done); if (w <= 0) return -1; done += (size_t)w; }
        return 0;
    }
    int T = writers < 1 ? 1 : writers > 16 ? 16 : writers;
    if (len < ((size_t)1 << 20)) T = 1;
    uint8_t *map = NULL;
    const off_t base = off & ~(off_t)4095;
    if (use_mmap) {
        if (ftruncate(fd, off + (off_t)len) != 0) return -1;
        map = (uint8_t *)mmap(NULL, (size_t)(off - base) + len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, base);
        if (map == MAP_FAILED) map = NULL;                       /* (a pipe, a filesystem without shared mappings: pwrite below) */
    }
    wjob_t job[16];
    pthread_t th[16];
    const size_t part = ((len / (size_t)T) + 4095) & ~(size_t)4095;
    int used = 0;
    for (int t = 0; t < T; t++) {
        const size_t lo = (size_t)t * part;
        if (lo >= len) break;
        job[t].fd = fd; job[t].src = src + lo; job[t].dst = map ? map + (off - base) + lo : NULL; job[t].len = len - lo < part ? len - lo : part; job[t].off = off + (off_t)lo; job[t].ok = 0;
        used++;
    }
    for (int t = 1; t < used; t++) pthread_create(&th[t], NULL, wjob_main, &job[t]);
    wjob_main(&job[0]);
    for (int t = 1; t < used; t++) pthread_join(th[t], NULL);
    if (map) munmap(map, (size_t)(off - base) + len);
    for (int t = 0; t < used; t++) if (!job[t].ok) return -1;
    return 0;
}

/* buffers of one chunk slot.  Output room: what the conversion usually needs (a chunk that outgrows it is redone with the room it asked for) */
static int fslot_alloc(fpipe_t *P, int i) {
    fslot_t *b = &P->slot[i];
    const char *e;
    b->cap = (uint32_t)(P->chunk / 48) + 1;                      /* descriptors per chunk; a chunk of smaller records than that is framed in two rounds */
    e = getenv("S5VIEW_SLOT_RECS");                             /* tests: fewer descriptors than a chunk holds records */
    if (e && atoi(e) > 0) b->cap = (uint32_t)atoi(e);
    b->in = (uint8_t *)s5gpu_host_alloc(P->chunk + 64);
    /* text in: ~4.5 text bytes per sample against ~0.9 record bytes; text out: ~4.5 against 0.9 the other way; BLOW5 to BLOW5: about the
     * same size unless a compressed file is written out uncompressed */
    const int expands = P->from.record_method != SLOW5_COMPRESS_NONE && P->to.record_method == SLOW5_COMPRESS_NONE;
    b->out_cap = P->ascii ? P->chunk / 2 : P->ascii_out ? P->chunk * 6 : expands ? P->chunk * 3 : P->chunk + P->chunk / 4;
    if (b->out_cap < (1u << 16)) b->out_cap = 1u << 16;
    b->out = (uint8_t *)s5gpu_host_alloc(b->out_cap);
    b->rec_pos = (uint64_t *)malloc(sizeof(uint64_t) * b->cap);
    b->rec_len = (uint32_t *)malloc(sizeof(uint32_t) * b->cap);
    b->out_off = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)b->cap + 1));
    if (P->ascii) { b->nl_cap = 16 * (uint32_t)(P->chunk / 16 / 64 + 64); b->nl = (uint32_t *)malloc(sizeof(uint32_t) * b->nl_cap); }
    if (!b->in || !b->out || !b->rec_pos || !b->rec_len || !b->out_off || (P->ascii && !b->nl)) return -1;
    pthread_mutex_lock(&P->mu);
    b->ready = 1;
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
    return 0;
}
static void *fslot_alloc_rest(void *arg) {
    fpipe_t *P = (fpipe_t *)arg;
    for (int i = 1; i < P->nslot; i++) {
        /* a file that fits the slots already there needs no more of them */
        pthread_mutex_lock(&P->mu);
        const int done = P->failed || P->total_batches >= 0;
        pthread_mutex_unlock(&P->mu);
        if (done) break;
        if (fslot_alloc(P, i) != 0) { fpipe_fail(P, "cannot allocate the chunk buffers"); break; }
    }
    stamp("all chunk slots pinned");
    return NULL;
}
/* returns 0 and the record count, -1, or -2: a record does not fit a chunk (an uncompressed ultra-long read) — the caller redoes the
 * file with the per-record pipeline */
static int fast_view(slow5_file_t *in, FILE *out, slow5_press_method_t from, slow5_press_method_t to, int ascii_out, int workers, uint64_t *total) {
    fpipe_t P;
    memset(&P, 0, sizeof P);
    pthread_mutex_init(&P.mu, NULL);
    pthread_cond_init(&P.cv, NULL);
    struct stat st;
    P.fd_in = fileno(in->fp);
    P.ascii = in->format == SLOW5_FORMAT_ASCII;
    P.ascii_out = ascii_out;
    if ((P.ascii || P.ascii_out) && in->header->aux_meta) { P.n_aux = in->header->aux_meta->num; P.aux_type = in->header->aux_meta->types; }
    if (fstat(P.fd_in, &st) != 0 || (uint64_t)st.st_size < in->meta.start_rec_offset + (P.ascii ? 0 : 5)) return -1;
    if (!P.ascii) {   /* the end marker must close the file (src/quickcheck.c:93-97) */
        char tail[5];
        if (pread(P.fd_in, tail, 5, st.st_size - 5) != 5 || memcmp(tail, "5WOLB", 5) != 0) { fprintf(stderr, "s5view: no BLOW5 end marker\n"); return -1; }
    }
    fflush(out);
    P.fd_out = fileno(out);
    P.pos = in->meta.start_rec_offset;
    P.end = (uint64_t)st.st_size - (P.ascii ? 0 : 5);
    const char *e = getenv("S5VIEW_CHUNK_MB");
    P.chunk = (size_t)(e ? atoi(e) : 32) << 20;
    e = getenv("S5VIEW_CHUNK_KB");                                   /* tests: chunks smaller than a record batch */
    if (e && atoi(e) > 0) P.chunk = (size_t)atoi(e) << 10;
    e = getenv("S5VIEW_READERS");
    P.readers = e ? atoi(e) : (in->format == SLOW5_FORMAT_ASCII ? 8 : 4);   /* (text is twice the bytes per sample: more copy threads) */
    P.from = from; P.to = to; P.total_batches = -1;
    e = getenv("S5VIEW_SLOTS");
    P.nslot = e ? atoi(e) : 6;
    if (P.nslot < 2) P.nslot = 2;
    if (P.nslot > FSLOT_MAX) P.nslot = FSLOT_MAX;
    const double t_alloc = now_s();
    /* slot 0 is pinned here; the others by a helper thread while the first chunk is read and sent (a pinned buffer of tens of MB costs
     * 5-15 ms, and a short job is over before four slots' worth of that would have been spent up front) */
    if (fslot_alloc(&P, 0) != 0) { fprintf(stderr, "s5view: cannot allocate the chunk buffers (%s)\n", s5gpu_last_error()); return -1; }
    stamp("first chunk slot pinned");
    const double t0 = now_s();
    pthread_t rd, wk[8], al;
    const int W = workers > 8 ? 8 : workers;
    const int al_started = pthread_create(&al, NULL, fslot_alloc_rest, &P) == 0;
    if (!al_started) fslot_alloc_rest(&P);                              /* no helper thread: pin the other slots here */
    if (pthread_create(&rd, NULL, freader_main, &P) != 0) { fprintf(stderr, "s5view: cannot start the reader thread\n"); return -1; }
    for (int i = 0; i < W; i++) pthread_create(&wk[i], NULL, fworker_main, &P);
    uint64_t out_bytes = 0;
    off_t out_pos = lseek(P.fd_out, 0, SEEK_CUR);                    /* (the header is flushed: the records start here) */
    e = getenv("S5VIEW_WRITERS");
    const int writers = e ? atoi(e) : 1;     /* (round 4, MI355X box, /dev/shm: one write() 6.5 GB/s, 4 pwrite threads 7.1, 8 threads 3.5: tools/hw_probe/shm_write_probe.c) */
    e = getenv("S5VIEW_WRITE_MODE");
    const int write_mmap = e && strcmp(e, "mmap") == 0;
    for (int64_t s = 0;; s++) {                                      /* ordered write phase: the chunks in order, each by several threads */
        fslot_t *b = &P.slot[s % P.nslot];
        const double tq0 = now_s();
        pthread_mutex_lock(&P.mu);
        while (!P.failed && !(b->state == ST_DONE && b->seq == s) && !(P.total_batches >= 0 && s >= P.total_batches)) pthread_cond_wait(&P.cv, &P.mu);
        const int stop = P.failed || (P.total_batches >= 0 && s >= P.total_batches);
        pthread_mutex_unlock(&P.mu);
        if (stop) break;
        const double tq1 = now_s();
        P.t_wwait += tq1 - tq0;
        if (write_chunk(P.fd_out, b->out, b->out_total, out_pos, writers, write_mmap) != 0) fpipe_fail(&P, "write failed");
        P.t_write += now_s() - tq1;
        if (s == 0) stamp("first chunk written");
        if (out_pos >= 0) out_pos += (off_t)b->out_total;
        out_bytes += b->out_total;
        *total += b->n;
        pthread_mutex_lock(&P.mu);
        b->state = ST_EMPTY;
        pthread_cond_broadcast(&P.cv);
        pthread_mutex_unlock(&P.mu);
    }
    pthread_join(rd, NULL);
    for (int i = 0; i < W; i++) pthread_join(wk[i], NULL);
    if (al_started) pthread_join(al, NULL);
    if (out_pos >= 0 && lseek(P.fd_out, out_pos, SEEK_SET) < 0) fpipe_fail(&P, "seek failed");   /* the end marker follows the last record */
    const double t1 = now_s();
    stamp("last write");
    /* (the process leaves through _exit right after this: un-pinning the chunk buffers — 0.12 ms per MB — is skipped unless S5_FULL_EXIT=1) */
    { const char *fe = getenv("S5_FULL_EXIT");
      if (fe && atoi(fe))
          for (int i = 0; i < P.nslot; i++) { fslot_t *b = &P.slot[i]; s5gpu_host_free(b->in); s5gpu_host_free(b->out); free(b->rec_pos); free(b->rec_len); free(b->out_off); free(b->nl); } }
    if (g_timing) fprintf(stderr, "s5view[t] stages (seconds, summed): reader pread %.3f + framing %.3f + waiting for a free slot %.3f | GPU calls %.3f over %d worker(s) | writer write %.3f + waiting for a chunk %.3f | %d slots\n",
                          P.t_read, P.t_frame, P.t_rwait, P.t_gpu, W, P.t_write, P.t_wwait, P.nslot);
    if (P.failed && P.oversize) return -2;
    if (P.failed) { fprintf(stderr, "s5view: %s\n", P.why); return -1; }
    fprintf(stderr, "s5view: chunked pipeline%s%s: %.3f s for %llu records (%.1f MB in, %.1f MB out; buffers %.3f s), %d pread threads, %d GPU worker(s), chunks of %zu MB\n",
            P.ascii ? " (SLOW5 text in)" : "", P.ascii_out ? " (SLOW5 text out)" : "", t1 - t0, (unsigned long long)*total, (double)(P.end - in->meta.start_rec_offset) / 1e6, (double)out_bytes / 1e6, t0 - t_alloc, P.readers, W, P.chunk >> 20);
    return 0;
}

/* The work is done and every file is closed: leave without the HIP runtime's static destructors (code objects, memory pools: ~0.1 s of a
 * one-second job).  S5_FULL_EXIT=1 takes the ordinary way out (leak checkers). */
static int leave(void) {
    if (fflush(stdout) != 0) { fprintf(stderr, "%s: writing the standard output failed\n", "s5view"); fflush(stderr); _exit(EXIT_FAILURE); }
    fflush(stderr);
    const char *e = getenv("S5_FULL_EXIT");
    if (e && atoi(e)) { s5gpu_shutdown(); return EXIT_SUCCESS; }
    _exit(EXIT_SUCCESS);
}
/* fopen(path, "wb") over an existing multi-gigabyte file gives its pages back before it returns — 0.07 s per 3.5 GB on tmpfs, on the critical
 * path of a conversion that takes one second.  A big old output is moved aside instead, the new one is created beside it, and the old
 * one is unlinked by a thread of its own while the pipeline runs. */
static void *unlink_main(void *arg) {
    char *p = (char *)arg;
    unlink(p);
    free(p);
    return NULL;
}
static FILE *open_output(const char *path) {
    struct stat st;
    if (stat(path, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > (off_t)(64 << 20)) {
        const size_t n = strlen(path) + 32;
        char *old = (char *)malloc(n);
        if (old) {
            snprintf(old, n, "%s.s5old.%ld", path, (long)getpid());
            pthread_t th;
            if (rename(path, old) == 0) {
                if (pthread_create(&th, NULL, unlink_main, old) == 0) pthread_detach(th);
                else { unlink(old); free(old); }
            } else free(old);
        }
    }
    return fopen(path, "wb");
}
static void *early_init_main(void *arg) {
    (void)arg;
    if (s5gpu_warmup() == S5GPU_OK) s5gpu_host_free(s5gpu_host_alloc(4096));   /* (the first pinned allocation brings up its own machinery) */
    stamp("early device initialisation done");
    return NULL;
}
int main(int argc, char **argv) {
    g_t_main = now_s();
    g_timing = getenv("S5VIEW_TIMING") && atoi(getenv("S5VIEW_TIMING"));
    if (argc >= 3 && strcmp(argv[1], "--index") == 0) {
        slow5_file_t *s = slow5_open(argv[2], "r");
        if (!s) return die("cannot open input");
        int rc = slow5_idx_create(s);
        slow5_close(s);
        return rc == 0 ? EXIT_SUCCESS : die("index failed");
    }
    if (argc >= 4 && strcmp(argv[1], "--get") == 0) {
        slow5_file_t *s = slow5_open(argv[2], "r");
        if (!s) return die("cannot open input");
        if (slow5_idx_load(s) != 0) return die("cannot load index");
        slow5_rec_t *rec = NULL;
        if (slow5_get(argv[3], &rec, s) != 0) return die("read not found / corrupt");
        printf("%s\t%u\t%llu", rec->read_id, rec->read_group, (unsigned long long)rec->len_raw_signal);
        for (uint64_t i = 0; i < rec->len_raw_signal && i < 8; i++) printf("%c%d", i ? ',' : '\t', rec->raw_signal[i]);
        printf("\n");
        slow5_rec_free(rec);
        slow5_close(s);
        return EXIT_SUCCESS;
    }
    if (argc < 3) {
        fprintf(stderr, "usage: s5view in.blow5 out.blow5 [none|zlib|zstd] [none|svb-zd|ex-zd] [K]\n");
        return EXIT_FAILURE;
    }
    slow5_press_method_t to = {SLOW5_COMPRESS_ZLIB, SLOW5_COMPRESS_SVB_ZD};
    if (argc > 3) to.record_method = strcmp(argv[3], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(argv[3], "zstd") == 0 ? SLOW5_COMPRESS_ZSTD : SLOW5_COMPRESS_ZLIB;
    if (argc > 4) to.signal_method = strcmp(argv[4], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(argv[4], "ex-zd") == 0 ? SLOW5_COMPRESS_EX_ZD : SLOW5_COMPRESS_SVB_ZD;
    const int64_t K = argc > 5 ? atoll(argv[5]) : 4096;

    {   /* several GPUs: S5VIEW_DEV_MASK (bit d = HIP device d) — every batch call then splits its records over them */
        const char *dm = getenv("S5VIEW_DEV_MASK");
        if (dm && strtoull(dm, NULL, 0) && s5gpu_init_mask(strtoull(dm, NULL, 0)) != S5GPU_OK) return die("cannot initialise the devices of S5VIEW_DEV_MASK");
    }
    /* the HIP runtime and the device context come up (~0.15 s) while the input's header is read and the output is created */
    pthread_t init_th;
    const int early_init = !(getenv("S5VIEW_DEV_MASK") && strtoull(getenv("S5VIEW_DEV_MASK"), NULL, 0));
    int early_init_started = early_init && pthread_create(&init_th, NULL, early_init_main, NULL) == 0;   /* (no thread: the first GPU call initialises) */
    slow5_file_t *in = slow5_open(argv[1], "r");
    if (!in) return die("cannot open input");
    stamp("input opened, header read");
    FILE *out = open_output(argv[2]);
    if (!out) return die("cannot open output");
    const size_t ol = strlen(argv[2]);
    const enum slow5_fmt fmt_out = ol > 6 && strcmp(argv[2] + ol - 6, ".slow5") == 0 ? SLOW5_FORMAT_ASCII : SLOW5_FORMAT_BINARY;
    if (slow5_hdr_fwrite(out, in->header, fmt_out, to) < 0) return die("header write failed");
    slow5_press_method_t from = {in->compress->record_press->method, in->compress->signal_press->method};

    const int workers = argc > 6 ? atoi(argv[6]) : 1;
    uint64_t total = 0;
    if (early_init_started) pthread_join(init_th, NULL);
    stamp("device ready");
    const char *nofast = getenv("S5VIEW_PER_RECORD");
    /* chunked pipeline: BLOW5 -> BLOW5, SLOW5 -> BLOW5 (the conversion BASELINE configs[0] names) and BLOW5 -> SLOW5; text to text takes the per-record one */
    int fast = workers > 0 && (fmt_out == SLOW5_FORMAT_BINARY || in->format == SLOW5_FORMAT_BINARY) && !(nofast && atoi(nofast));
    if (fast) {
        const int rc = fast_view(in, out, from, to, fmt_out == SLOW5_FORMAT_ASCII, workers, &total);
        if (rc == -2) {   /* start the output over, record by record (the chunked reader used pread: in->fp still stands at the first record) */
            fprintf(stderr, "s5view: a record larger than a chunk (S5VIEW_CHUNK_MB): per-record pipeline\n");
            if (in->format == SLOW5_FORMAT_ASCII && fseeko(in->fp, (off_t)in->meta.start_rec_offset, SEEK_SET) != 0) return die("cannot rewind the input");
            total = 0;
            if (fflush(out) != 0 || ftruncate(fileno(out), 0) != 0 || fseek(out, 0, SEEK_SET) != 0 || slow5_hdr_fwrite(out, in->header, fmt_out, to) < 0) return die("cannot restart the output");
            fast = 0;
        } else if (rc != 0) return die("chunked pipeline failed");
    }
    if (fast) {
    } else if (workers > 0) {
        /* read || GPU || write.  The reference runs the three phases one after the other per batch
         * (src/view.c:265-278, 292, 296-299) and its authors note the overlap as the missing 2x (README.md:197). */
        pipe_t P;
        memset(&P, 0, sizeof P);
        pthread_mutex_init(&P.mu, NULL);
        pthread_cond_init(&P.cv, NULL);
        P.in = in; P.K = K; P.from = from; P.to = to; P.fmt_out = fmt_out; P.total_batches = -1;
        for (int i = 0; i < NSLOT; i++) {
            P.slot[i].mem = (char **)calloc(K, sizeof(char *));
            P.slot[i].bytes = (size_t *)calloc(K, sizeof(size_t));
            P.slot[i].bufs = (void **)calloc(K, sizeof(void *));
            P.slot[i].lens = (size_t *)calloc(K, sizeof(size_t));
        }
        pthread_t rd, wk[8];
        const int W = workers > 8 ? 8 : workers;
        pthread_create(&rd, NULL, reader_main, &P);
        for (int i = 0; i < W; i++) pthread_create(&wk[i], NULL, worker_main, &P);
        for (int64_t s = 0;; s++) {                                  /* ordered write phase, src/view.c:296-299 */
            slot_t *b = &P.slot[s % NSLOT];
            pthread_mutex_lock(&P.mu);
            while (!P.failed && !(b->state == ST_DONE && b->seq == s) && !(P.total_batches >= 0 && s >= P.total_batches)) pthread_cond_wait(&P.cv, &P.mu);
            const int stop = P.failed || (P.total_batches >= 0 && s >= P.total_batches);
            pthread_mutex_unlock(&P.mu);
            if (stop) break;
            for (int64_t i = 0; i < b->n; i++) {
                if (fwrite(b->bufs[i], 1, b->lens[i], out) != b->lens[i]) return die("write failed");
                free(b->bufs[i]);
            }
            total += (uint64_t)b->n;
            pthread_mutex_lock(&P.mu);
            b->state = ST_EMPTY;
            pthread_cond_broadcast(&P.cv);
            pthread_mutex_unlock(&P.mu);
        }
        pthread_join(rd, NULL);
        for (int i = 0; i < W; i++) pthread_join(wk[i], NULL);
        if (P.failed) { fprintf(stderr, "s5view: %s\n", P.why); return EXIT_FAILURE; }
    } else {
        char **mem = (char **)calloc(K, sizeof(char *));
        size_t *bytes = (size_t *)calloc(K, sizeof(size_t));
        void **bufs = (void **)calloc(K, sizeof(void *));
        size_t *lens = (size_t *)calloc(K, sizeof(size_t));
        int eof = 0;
        while (!eof) {
            int64_t n = 0;                                              /* read phase, src/view.c:265-278 */
            while (n < K) {
                mem[n] = (char *)slow5_get_next_mem(&bytes[n], in);
                if (!mem[n]) { if (slow5_errno != SLOW5_ERR_EOF) return die("bad record framing"); eof = 1; break; }
                n++;
            }
            if (n == 0) break;
            /* compute phase: the work_db() of src/view.c:292, one call for the whole batch */
            if (slow5_gpu_convert_batch(n, mem, bytes, in->format, from, in->header->aux_meta, fmt_out, to, NULL, 0, bufs, lens) != 0)
                return die("GPU press path failed");
            for (int64_t i = 0; i < n; i++) {                           /* ordered write phase, src/view.c:296-299 */
                if (fwrite(bufs[i], 1, lens[i], out) != lens[i]) return die("write failed");
                free(bufs[i]);
            }
            total += (uint64_t)n;
        }
        free(mem); free(bytes); free(bufs); free(lens);
    }
    if (fmt_out == SLOW5_FORMAT_BINARY && slow5_eof_fwrite(out) < 0) return die("eof write failed");   /* src/view.c:311-313 */
    if (fclose(out) != 0) return die("closing the output failed (its last bytes may not be on disk)");
    slow5_close(in);
    fprintf(stderr, "s5view: %llu records\n", (unsigned long long)total);
    stamp("output closed");
    return leave();
}
