/*
 * s5view.c — the batch loop of `slow5tools view` (SLOW5 / BLOW5 -> SLOW5 / BLOW5) on the GPU press path.
 *
 * Not a CLI re-implementation: this is slow5_convert_parallel (/root/reference/src/view.c:241-323) with the
 * work_db() call at src/view.c:292 replaced by ONE slow5_gpu_convert_batch() per batch, written against
 * include/slow5_compat.h only.  It doubles as the end-to-end harness of tests/test_container.py.
 *
 *   s5view in.[b|s]low5 out.[b|s]low5 [record: none|zlib] [signal: none|svb-zd] [batch K]   (defaults zlib svb-zd 4096,
 *        src/misc.c:54-58, src/cmd.h:8; the input format is sniffed, the output format follows the extension as in
 *        src/view.c:170-190; press methods are ignored for a .slow5 output)
 *   s5view --index in.blow5          writes in.blow5.idx (slow5tools index)
 *   s5view --get in.blow5 read_id    prints len_raw_signal and the first samples of one read (slow5tools get)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slow5_compat.h"
#include "slow5gpu.h"

static int die(const char *what) {
    fprintf(stderr, "s5view: %s (slow5_errno %d; %s)\n", what, slow5_errno, s5gpu_last_error());
    return EXIT_FAILURE;
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
        fprintf(stderr, "usage: s5view in.blow5 out.blow5 [none|zlib] [none|svb-zd] [K]\n");
        return EXIT_FAILURE;
    }
    slow5_press_method_t to = {SLOW5_COMPRESS_ZLIB, SLOW5_COMPRESS_SVB_ZD};
    if (argc > 3) to.record_method = strcmp(argv[3], "none") == 0 ? SLOW5_COMPRESS_NONE : SLOW5_COMPRESS_ZLIB;
    if (argc > 4) to.signal_method = strcmp(argv[4], "none") == 0 ? SLOW5_COMPRESS_NONE : SLOW5_COMPRESS_SVB_ZD;
    const int64_t K = argc > 5 ? atoll(argv[5]) : 4096;

    slow5_file_t *in = slow5_open(argv[1], "r");
    if (!in) return die("cannot open input");
    FILE *out = fopen(argv[2], "wb");
    if (!out) return die("cannot open output");
    const size_t ol = strlen(argv[2]);
    const enum slow5_fmt fmt_out = ol > 6 && strcmp(argv[2] + ol - 6, ".slow5") == 0 ? SLOW5_FORMAT_ASCII : SLOW5_FORMAT_BINARY;
    if (slow5_hdr_fwrite(out, in->header, fmt_out, to) < 0) return die("header write failed");
    slow5_press_method_t from = {in->compress->record_press->method, in->compress->signal_press->method};

    char **mem = (char **)calloc(K, sizeof(char *));
    size_t *bytes = (size_t *)calloc(K, sizeof(size_t));
    void **bufs = (void **)calloc(K, sizeof(void *));
    size_t *lens = (size_t *)calloc(K, sizeof(size_t));
    uint64_t total = 0;
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
    if (fmt_out == SLOW5_FORMAT_BINARY && slow5_eof_fwrite(out) < 0) return die("eof write failed");   /* src/view.c:311-313 */
    fclose(out);
    slow5_close(in);
    free(mem); free(bytes); free(bufs); free(lens);
    fprintf(stderr, "s5view: %llu records\n", (unsigned long long)total);
    s5gpu_shutdown();
    return EXIT_SUCCESS;
}
