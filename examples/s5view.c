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
 *        threads of the read || GPU || write pipeline (default 1: the reader is the bottleneck); 0 runs the reference's serial read / compute / write phases.
 *   s5view --index in.blow5          writes in.blow5.idx (slow5tools index)
 *   s5view --get in.blow5 read_id    prints len_raw_signal and the first samples of one read (slow5tools get)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slow5_compat.h"
#include "slow5gpu.h"

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

int main(int argc, char **argv) {
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

    slow5_file_t *in = slow5_open(argv[1], "r");
    if (!in) return die("cannot open input");
    FILE *out = fopen(argv[2], "wb");
    if (!out) return die("cannot open output");
    const size_t ol = strlen(argv[2]);
    const enum slow5_fmt fmt_out = ol > 6 && strcmp(argv[2] + ol - 6, ".slow5") == 0 ? SLOW5_FORMAT_ASCII : SLOW5_FORMAT_BINARY;
    if (slow5_hdr_fwrite(out, in->header, fmt_out, to) < 0) return die("header write failed");
    slow5_press_method_t from = {in->compress->record_press->method, in->compress->signal_press->method};

    const int workers = argc > 6 ? atoi(argv[6]) : 1;
    uint64_t total = 0;
    if (workers > 0) {
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
    fclose(out);
    slow5_close(in);
    fprintf(stderr, "s5view: %llu records\n", (unsigned long long)total);
    s5gpu_shutdown();
    return EXIT_SUCCESS;
}
