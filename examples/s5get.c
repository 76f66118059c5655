/*
 * s5get.c — the batch loop of `slow5tools get` (read ids -> records of an indexed BLOW5 file) on the GPU press path.
 *
 * Not a CLI re-implementation: this is the loop of /root/reference/src/get.c:321-386 — load the index (get.c:286), per batch of K ids
 * look each one up, pread its record, run the worker (work_per_single_read_get, get.c:37-66: slow5_get = decode, then slow5_rec_to_mem)
 * for the whole batch in ONE call, write in order — as a pipeline: several reader threads pread batches into pinned memory, the GPU
 * worker takes each batch as one framed chunk, the writer writes one block per batch.  Against include/slow5_compat.h + slow5gpu.h.
 *
 *   s5get in.blow5 ids.txt out.blow5 [record: none|zlib|zstd] [signal: none|svb-zd|ex-zd] [K] [readers]
 *        the records of the ids (one per line), re-encoded with the given methods (defaults zlib svb-zd, K 4096 = src/cmd.h:8, 8 readers);
 *        an id that is not in the index is an error, as in get.c:47 (skipped with a warning when S5GET_SKIP=1: get --skip)
 *   s5get --benchmark in.blow5 ids.txt [K] [readers]
 *        get --benchmark (get.c:52): fetch + decode only, nothing written; prints reads/s and the batch latencies of the decode call
 *   s5get --random in.blow5 N seed out_ids.txt
 *        N ids drawn uniformly (with replacement) from the index: the id list of BASELINE configs[4] (100 k ids, seed 1)
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "slow5_compat.h"
#include "slow5gpu.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int g_timing;                 /* S5VIEW_TIMING=1: a timeline of the process on stderr (as s5view's) */
static double g_t_main;
static void stamp(const char *what) { if (g_timing) fprintf(stderr, "s5get[t] %8.3f  %s\n", now_s() - g_t_main, what); }
static int die(const char *what) {
    fprintf(stderr, "s5get: %s (slow5_errno %d; %s)\n", what, slow5_errno, s5gpu_last_error());
    return EXIT_FAILURE;
}
static int rec_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_ZLIB ? S5GPU_REC_ZLIB : m == SLOW5_COMPRESS_ZSTD ? S5GPU_REC_ZSTD : S5GPU_REC_NONE; }
static int sig_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_SVB_ZD ? S5GPU_SIG_SVB_ZD : m == SLOW5_COMPRESS_EX_ZD ? S5GPU_SIG_EX_ZD : S5GPU_SIG_NONE; }

#define GSLOT 8
enum { ST_EMPTY = 0, ST_FILLED, ST_BUSY, ST_DONE };
typedef struct {
    int state;
    int64_t seq;
    uint32_t n;
    uint8_t *in;                 /* pinned: the batch's records, each behind its u64 size prefix, 16-byte aligned */
    size_t in_cap, in_have;
    uint64_t *rec_pos, *off;     /* off: out_off (records) or sig_off (benchmark), n + 1 */
    uint32_t *rec_len;
    uint8_t *out;                /* pinned: re-encoded records / decoded signals */
    size_t out_cap, out_total;
    s5gpu_rec_fields_t *fields;
    double t_gpu;
} gslot_t;
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    gslot_t slot[GSLOT];
    int nslot;                       /* slots in use (<= GSLOT): a short job pins fewer buffers — pinning costs 0.23 ms per MB */
    slow5_file_t *in;
    int fd;
    char **ids;
    uint64_t n_ids;
    int64_t K, n_batches, next_fill, next_work;
    slow5_press_method_t from, to;
    int benchmark, skip, failed;
    char why[320];
    uint64_t missing;
} gpipe_t;

static void gfail(gpipe_t *P, const char *what, const char *arg) {
    pthread_mutex_lock(&P->mu);
    if (!P->failed) { P->failed = 1; snprintf(P->why, sizeof P->why, "%s%s%s (%s)", what, arg ? " " : "", arg ? arg : "", s5gpu_last_error()); }
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
}

/* A slot's buffers are pinned by the reader thread that fills it first: the readers run side by side, so the slots come up in parallel
 * with the first preads, and a list of a few batches never pays for eight slots (640 MB of pinned memory up front was a third of a
 * 100 k-id job). */
static int gslot_alloc(gpipe_t *P, gslot_t *b) {
    /* all or nothing, and b->in — what the readers test — is set LAST: a slot is either absent or whole, whoever looks at it and when */
    const size_t in_cap = (size_t)P->K * 4096 + 65536;
    const size_t out_cap = (size_t)P->K * (P->benchmark ? 10240 : 6144) + 65536;     /* decoded samples / re-encoded records; a batch that outgrows it is redone */
    uint8_t *in = (uint8_t *)s5gpu_host_alloc(in_cap);
    uint8_t *out = (uint8_t *)s5gpu_host_alloc(out_cap);
    uint64_t *rec_pos = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)P->K);
    uint32_t *rec_len = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)P->K);
    uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)P->K + 1));
    s5gpu_rec_fields_t *fields = (s5gpu_rec_fields_t *)malloc(sizeof(s5gpu_rec_fields_t) * (size_t)P->K);
    if (!(in && out && rec_pos && rec_len && off && fields)) {
        s5gpu_host_free(in); s5gpu_host_free(out); free(rec_pos); free(rec_len); free(off); free(fields);
        return -1;
    }
    b->in_cap = in_cap; b->out_cap = out_cap;
    b->out = out; b->rec_pos = rec_pos; b->rec_len = rec_len; b->off = off; b->fields = fields;
    __sync_synchronize();
    b->in = in;
    return 0;
}

/* read phase (get.c:335-361): one reader thread fills one whole batch; several batches are being filled at once */
static void *greader_main(void *arg) {
    gpipe_t *P = (gpipe_t *)arg;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        const int64_t s = P->next_fill;
        if (P->failed || s >= P->n_batches) { pthread_mutex_unlock(&P->mu); return NULL; }
        P->next_fill = s + 1;
        gslot_t *b = &P->slot[s % P->nslot];
        while (!P->failed && !(b->state == ST_EMPTY && b->seq == s - P->nslot)) pthread_cond_wait(&P->cv, &P->mu);   /* the slot's previous batch has been written */
        const int stop = P->failed;
        pthread_mutex_unlock(&P->mu);
        if (stop) return NULL;
        if (!b->in && gslot_alloc(P, b) != 0) { gfail(P, "cannot allocate the batch buffers", NULL); return NULL; }
        const uint64_t i0 = (uint64_t)s * (uint64_t)P->K, i1 = i0 + (uint64_t)P->K < P->n_ids ? i0 + (uint64_t)P->K : P->n_ids;
        size_t at = 0;
        uint32_t n = 0;
        for (uint64_t i = i0; i < i1; i++) {
            struct slow5_rec_idx e;
            if (slow5_idx_get(P->in->index, P->ids[i], &e) != 0) {
                if (P->skip) { __sync_fetch_and_add(&P->missing, 1); continue; }
                gfail(P, "read id not in the index:", P->ids[i]);
                return NULL;
            }
            if (at + e.size + 32 > b->in_cap) {                         /* (only the reader of this slot touches its buffer) */
                size_t nc = b->in_cap * 2 > at + e.size + 32 ? b->in_cap * 2 : at + e.size + 32 + (b->in_cap >> 1);
                uint8_t *nb = (uint8_t *)s5gpu_host_alloc(nc);
                if (!nb) { gfail(P, "cannot grow the batch buffer", NULL); return NULL; }
                memcpy(nb, b->in, at);
                s5gpu_host_free(b->in);
                b->in = nb; b->in_cap = nc;
            }
            size_t got = 0;
            while (got < e.size) {
                ssize_t r = pread(P->fd, b->in + at + got, e.size - got, (off_t)(e.offset + got));
                if (r <= 0) { gfail(P, "pread failed for", P->ids[i]); return NULL; }
                got += (size_t)r;
            }
            uint64_t sz;
            memcpy(&sz, b->in + at, 8);
            if (sz + 8 != e.size) { gfail(P, "the index does not match the file at", P->ids[i]); return NULL; }
            b->rec_pos[n] = at + 8;
            b->rec_len[n] = (uint32_t)sz;
            n++;
            at = (at + e.size + 15) & ~(size_t)15;
        }
        b->in_have = at;
        pthread_mutex_lock(&P->mu);
        b->n = n; b->seq = s; b->state = ST_FILLED;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

/* compute phase: the work_db() of get.c:364, one call per batch */
static void *gworker_main(void *arg) {
    gpipe_t *P = (gpipe_t *)arg;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        gslot_t *b;
        int64_t s;
        for (;;) {
            s = P->next_work;
            b = &P->slot[s % P->nslot];
            if (P->failed || s >= P->n_batches) { pthread_mutex_unlock(&P->mu); return NULL; }
            if (b->state == ST_FILLED && b->seq == s) break;
            pthread_cond_wait(&P->cv, &P->mu);
        }
        P->next_work = s + 1;
        b->state = ST_BUSY;
        pthread_mutex_unlock(&P->mu);
        const double t0 = now_s();
        for (int attempt = 0; b->n; attempt++) {
            int rc;
            if (P->benchmark)
                rc = s5gpu_decode_stream(b->n, b->in, b->in_have, b->rec_pos, b->rec_len, rec_code_of(P->from.record_method), sig_code_of(P->from.signal_method),
                                         (int16_t *)b->out, b->out_cap / 2, b->off, b->fields);
            else
                rc = s5gpu_recompress_stream(b->n, b->in, b->in_have, b->rec_pos, b->rec_len, rec_code_of(P->from.record_method), sig_code_of(P->from.signal_method),
                                             rec_code_of(P->to.record_method), sig_code_of(P->to.signal_method), NULL, 0, b->out, b->out_cap, b->off, NULL);
            if (rc == S5GPU_OK) break;
            const size_t need_b = P->benchmark ? (size_t)b->off[0] * 2 : (size_t)b->off[0];
            if (rc == S5GPU_ERR_NOMEM && attempt == 0 && need_b > b->out_cap) {     /* the output outgrew its buffer: bring a bigger one */
                const size_t need = need_b + need_b / 8 + 64;
                s5gpu_host_free(b->out);
                b->out = (uint8_t *)s5gpu_host_alloc(need);
                b->out_cap = b->out ? need : 0;
                if (b->out) continue;
            }
            gfail(P, "GPU press path failed", NULL);
            return NULL;
        }
        b->out_total = b->n ? (P->benchmark ? (size_t)b->off[b->n] * 2 : (size_t)b->off[b->n]) : 0;
        b->t_gpu = now_s() - t0;
        pthread_mutex_lock(&P->mu);
        b->state = ST_DONE;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
}

static int cmp_d(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

static char **read_ids(const char *path, uint64_t *n_out) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    uint64_t n = 0, cap = 1024;
    char **ids = (char **)malloc(sizeof(char *) * cap), *line = NULL;
    size_t lc = 0;
    ssize_t got;
    while (ids && (got = getline(&line, &lc, f)) > 0) {
        while (got && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (!got) continue;
        if (n == cap) { cap *= 2; char **ni = (char **)realloc(ids, sizeof(char *) * cap); if (!ni) { free(ids); ids = NULL; break; } ids = ni; }
        ids[n++] = strdup(line);
    }
    free(line);
    fclose(f);
    *n_out = n;
    return ids;
}

static int gslot_alloc(gpipe_t *P, gslot_t *b);
static void *early_init_main(void *arg) {
    gpipe_t *P = (gpipe_t *)arg;
    if (s5gpu_warmup() != S5GPU_OK) return NULL;
    /* Round 5: the batch slots are pinned HERE, under the index load of the main thread.  Pinned by the reader threads at their first
     * batch they cost 0.23 ms per MB in the middle of the job, and they hold the runtime's lock while the first GPU call allocates its
     * workspaces (that call: 130 ms of a 190 ms job).  A slot that could not be pinned here is pinned by its first reader as before. */
    for (int i = 0; i < P->nslot; i++)
        if (!P->slot[i].in && gslot_alloc(P, &P->slot[i]) != 0) break;
    return NULL;
}
/* The work is done and every file is closed: leave without the HIP runtime's static destructors (code objects, memory pools: ~0.1 s of a
 * one-second job).  S5_FULL_EXIT=1 takes the ordinary way out (leak checkers). */
static int leave(void) {
    if (fflush(stdout) != 0) { fprintf(stderr, "%s: writing the standard output failed\n", "s5get"); fflush(stderr); _exit(EXIT_FAILURE); }
    fflush(stderr);
    const char *e = getenv("S5_FULL_EXIT");
    if (e && atoi(e)) { s5gpu_shutdown(); return EXIT_SUCCESS; }
    _exit(EXIT_SUCCESS);
}
int main(int argc, char **argv) {
    if (argc >= 6 && strcmp(argv[1], "--random") == 0) {
        slow5_file_t *s = slow5_open(argv[2], "r");
        if (!s) return die("cannot open input");
        if (slow5_idx_load(s) != 0) return die("cannot load index");
        uint64_t n = 0;
        char **rids = slow5_get_rids(s, &n);
        if (!rids || !n) return die("empty index");
        FILE *o = fopen(argv[5], "w");
        if (!o) return die("cannot open the id list for writing");
        uint64_t x = strtoull(argv[4], NULL, 0) * 0x9E3779B97F4A7C15ull + 0xD1342543DE82EF95ull;
        for (uint64_t k = 0, N = strtoull(argv[3], NULL, 0); k < N; k++) {
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;                     /* xorshift64* */
            fprintf(o, "%s\n", rids[(x * 0x2545F4914F6CDD1Dull >> 11) % n]);
        }
        fclose(o);
        slow5_close(s);
        return EXIT_SUCCESS;
    }
    g_t_main = now_s();
    { const char *e = getenv("S5VIEW_TIMING"); g_timing = e && atoi(e); }
    const int benchmark = argc >= 2 && strcmp(argv[1], "--benchmark") == 0;
    char **av = argv + (benchmark ? 1 : 0);
    const int ac = argc - (benchmark ? 1 : 0);
    if (ac < (benchmark ? 3 : 4)) {
        fprintf(stderr, "usage: s5get in.blow5 ids.txt out.blow5 [none|zlib|zstd] [none|svb-zd|ex-zd] [K] [readers]\n"
                        "       s5get --benchmark in.blow5 ids.txt [K] [readers]\n       s5get --random in.blow5 N seed out_ids.txt\n");
        return EXIT_FAILURE;
    }
    {   /* several GPUs: S5VIEW_DEV_MASK (bit d = HIP device d) — every batch call then splits its records over them */
        const char *dm = getenv("S5VIEW_DEV_MASK");
        if (dm && strtoull(dm, NULL, 0) && s5gpu_init_mask(strtoull(dm, NULL, 0)) != S5GPU_OK) return die("cannot initialise the devices of S5VIEW_DEV_MASK");
    }
    gpipe_t P;
    memset(&P, 0, sizeof P);
    pthread_mutex_init(&P.mu, NULL);
    pthread_cond_init(&P.cv, NULL);
    P.benchmark = benchmark;
    /* the id list first (milliseconds): the number of batches, and with it the number of slots, is known before anything is pinned */
    P.ids = read_ids(av[2], &P.n_ids);
    if (!P.ids) return die("cannot read the id list");
    stamp("id list read");
    {
        const int argk0 = benchmark ? 3 : 6;
        P.K = ac > argk0 ? atoll(av[argk0]) : 4096;
        if (P.K < 1) P.K = 1;
        P.n_batches = (int64_t)((P.n_ids + (uint64_t)P.K - 1) / (uint64_t)P.K);
        /* Round 5: eight slots of K x (4 KiB in + 16 KiB out) are 670 MB of pinned memory — 150 ms of page pinning for a job whose 24 batches
         * take 35 ms of GPU time.  A job gets one slot per six batches (2 .. 8), and no more reader threads than slots. */
        P.nslot = P.n_batches / 6 < 2 ? 2 : P.n_batches / 6 > GSLOT ? GSLOT : (int)(P.n_batches / 6);
    }
    /* the HIP runtime and the device context come up (~0.15 s), and the batch slots are pinned, while the index is loaded (~0.13 s per million reads) */
    pthread_t init_th;
    const int early_init = !(getenv("S5VIEW_DEV_MASK") && strtoull(getenv("S5VIEW_DEV_MASK"), NULL, 0));     /* (a device mask has initialised the library already) */
    const int early_init_started = early_init && pthread_create(&init_th, NULL, early_init_main, &P) == 0;   /* (no thread: the first GPU call initialises) */
    P.in = slow5_open(av[1], "r");
    if (!P.in || P.in->format != SLOW5_FORMAT_BINARY) return die("cannot open input (an indexed BLOW5 file)");
    const double t_idx0 = now_s();
    if (slow5_idx_load(P.in) != 0) return die("cannot load index");
    const double t_idx = now_s() - t_idx0;
    stamp("index loaded");
    if (early_init_started) pthread_join(init_th, NULL);
    stamp("library warm, slots pinned (early-init thread joined)");
    P.from.record_method = P.in->compress->record_press->method; P.from.signal_method = P.in->compress->signal_press->method;
    P.to.record_method = SLOW5_COMPRESS_ZLIB; P.to.signal_method = SLOW5_COMPRESS_SVB_ZD;
    int argk = benchmark ? 3 : 6;
    FILE *out = NULL;
    if (!benchmark) {
        if (ac > 4) P.to.record_method = strcmp(av[4], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(av[4], "zstd") == 0 ? SLOW5_COMPRESS_ZSTD : SLOW5_COMPRESS_ZLIB;
        if (ac > 5) P.to.signal_method = strcmp(av[5], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(av[5], "ex-zd") == 0 ? SLOW5_COMPRESS_EX_ZD : SLOW5_COMPRESS_SVB_ZD;
        out = fopen(av[3], "wb");
        if (!out) return die("cannot open output");
        if (slow5_hdr_fwrite(out, P.in->header, SLOW5_FORMAT_BINARY, P.to) < 0) return die("header write failed");
        fflush(out);
    }
    const int readers = ac > argk + 1 ? atoi(av[argk + 1]) : 8;
    { const char *e = getenv("S5GET_SKIP"); P.skip = e && atoi(e); }
    P.fd = fileno(P.in->fp);
    for (int i = 0; i < P.nslot; i++) P.slot[i].seq = (int64_t)i - P.nslot;     /* (a slot's buffers are pinned by the reader that first fills it: gslot_alloc) */
    double *lat = (double *)malloc(sizeof(double) * (size_t)(P.n_batches ? P.n_batches : 1));
    const double t0 = now_s();
    pthread_t rd[32], wk[4];
    const int R0 = readers < 1 ? 1 : readers > 32 ? 32 : readers;
    const int R = R0 > P.nslot ? P.nslot : R0;
    const int W = 2;
    for (int i = 0; i < R; i++) pthread_create(&rd[i], NULL, greader_main, &P);
    for (int i = 0; i < W; i++) pthread_create(&wk[i], NULL, gworker_main, &P);
    uint64_t total = 0, samples = 0, out_bytes = 0, checksum = 0;
    for (int64_t s = 0; s < P.n_batches; s++) {                          /* ordered write phase (get.c:373-384): one write per batch */
        gslot_t *b = &P.slot[s % P.nslot];
        pthread_mutex_lock(&P.mu);
        while (!P.failed && !(b->state == ST_DONE && b->seq == s)) pthread_cond_wait(&P.cv, &P.mu);
        const int stop = P.failed;
        pthread_mutex_unlock(&P.mu);
        if (stop) break;
        if (benchmark) {
            const int16_t *sg = (const int16_t *)b->out;
            for (uint32_t i = 0; i < b->n; i++) {                         /* use what was decoded: first, middle and last sample of every read */
                const uint64_t a = b->off[i], e = b->off[i + 1];
                if (e > a) checksum = checksum * 1000003ull + (uint16_t)sg[a] + ((uint64_t)(uint16_t)sg[a + (e - a) / 2] << 16) + ((uint64_t)(uint16_t)sg[e - 1] << 32);
                samples += e - a;
            }
        } else {
            size_t done = 0;
            while (done < b->out_total) {
                ssize_t w = write(fileno(out), b->out + done, b->out_total - done);
                if (w <= 0) { gfail(&P, "write failed", NULL); break; }
                done += (size_t)w;
            }
            out_bytes += b->out_total;
        }
        lat[s] = b->t_gpu;
        total += b->n;
        pthread_mutex_lock(&P.mu);
        b->state = ST_EMPTY;
        pthread_cond_broadcast(&P.cv);
        pthread_mutex_unlock(&P.mu);
    }
    for (int i = 0; i < R; i++) pthread_join(rd[i], NULL);
    for (int i = 0; i < W; i++) pthread_join(wk[i], NULL);
    const double dt = now_s() - t0;
    stamp("last batch done");
    if (P.failed) { fprintf(stderr, "s5get: %s\n", P.why); return EXIT_FAILURE; }
    if (out) {
        if (fseek(out, 0, SEEK_END) != 0 || slow5_eof_fwrite(out) < 0) return die("eof write failed");
        if (fclose(out) != 0) return die("closing the output failed (its last bytes may not be on disk)");
    }
    int64_t nl = P.n_batches;
    if (nl > 1 && (uint64_t)P.K * (uint64_t)nl != P.n_ids) nl--;       /* the last batch is a short one */
    qsort(lat, (size_t)nl, sizeof(double), cmp_d);
    fprintf(stderr, "s5get: %llu reads of %llu ids in %.3f s = %.0f reads/s (%s; K %lld, %d pread threads, %d GPU workers; index load %.3f s; %llu ids not found)\n",
            (unsigned long long)total, (unsigned long long)P.n_ids, dt, (double)total / dt, benchmark ? "fetch + decode" : "fetch + decode + re-encode + write",
            (long long)P.K, R, W, t_idx, (unsigned long long)P.missing);
    if (nl > 0) fprintf(stderr, "s5get: GPU call per batch: p50 %.3f ms, p99 %.3f ms over %lld batches\n", 1e3 * lat[nl / 2], 1e3 * lat[(nl * 99) / 100], (long long)nl);
    if (benchmark) printf("%llu\t%llu\t%016llx\n", (unsigned long long)total, (unsigned long long)samples, (unsigned long long)checksum);
    /* The process is about to _exit (leave()): un-pinning ~200 MB of slot buffers costs 90-130 ms and freeing a million index entries 50 — a
     * quarter of a 100 k-id job — for memory the kernel takes back anyway.  S5_FULL_EXIT=1 (leak checkers) releases everything. */
    { const char *fe = getenv("S5_FULL_EXIT");
      if (fe && atoi(fe)) {
          for (int i = 0; i < GSLOT; i++) { gslot_t *b = &P.slot[i]; s5gpu_host_free(b->in); s5gpu_host_free(b->out); free(b->rec_pos); free(b->rec_len); free(b->off); free(b->fields); }
          stamp("slot buffers released");
          slow5_close(P.in);
          stamp("input closed (index freed)");
      } }
    return leave();
}
