/*
 * s5merge.c — the batch loop of `slow5tools merge` (several SLOW5 / BLOW5 files -> one BLOW5 file) on the GPU press path.
 *
 * Not a CLI re-implementation: this is the record loop of /root/reference/src/merge.c:383-467 with its worker parallel_reads_model
 * (merge.c:43-70: decode, read_group = list[file][read_group], slow5_rec_to_mem against the OUTPUT header's aux fields) replaced by one
 * batch call, plus as much of the header merge of merge.c:217-350 as the worker depends on:
 *   - read groups: a read group whose run_id the output already has maps onto it (its attributes must agree: merge.c:318-323),
 *     any other is appended (merge.c:325-333);
 *   - header attributes: the union over the inputs, sorted, "." where a read group has none (slow5_hdr_add_rg_data);
 *   - aux fields: enum fields first, in the order they are met (merge.c:249-292), then the others sorted by name (the std::map of
 *     merge.c:218,293,345); a record of a file without one of them gets the missing value ("." in text; 0xFF / type maximum / NaN / empty
 *     array in BLOW5 — the reference's own merged_expected_zlib_svb.blow5 shows 0xFF for a missing enum).
 * Records of a file whose aux fields are the output's, in the same order, go straight through the batch hooks (BLOW5: decode and
 * re-encode device-resident, s5gpu_recompress_batch; SLOW5 text: s5gpu_ascii_to_blow5_batch) with the read_group rewrite on the device;
 * others take the detour over text lines, where the aux columns are re-ordered on the host.
 *
 *   s5merge out.blow5 [-c none|zlib|zstd] [-s none|svb-zd|ex-zd] [-l (lossy: drop aux fields)] [-K batch] in1 in2 ...
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "slow5_compat.h"
#include "slow5gpu.h"

static int die(const char *what, const char *arg) {
    fprintf(stderr, "s5merge: %s%s%s (slow5_errno %d; %s)\n", what, arg ? " " : "", arg ? arg : "", slow5_errno, s5gpu_last_error());
    return EXIT_FAILURE;
}
static int rec_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_ZLIB ? S5GPU_REC_ZLIB : m == SLOW5_COMPRESS_ZSTD ? S5GPU_REC_ZSTD : S5GPU_REC_NONE; }
static int sig_code_of(enum slow5_press_method m) { return m == SLOW5_COMPRESS_SVB_ZD ? S5GPU_SIG_SVB_ZD : m == SLOW5_COMPRESS_EX_ZD ? S5GPU_SIG_EX_ZD : S5GPU_SIG_NONE; }

/* ---- a header as the merge needs it ---- */
typedef struct { char *name; char **val; } attr_t;                     /* val[g] = value of read group g (NULL: none) */
typedef struct { char *name, *type; } auxcol_t;
typedef struct {
    attr_t *attr; size_t n_attr, n_rg;
    auxcol_t *aux; size_t n_aux;
    struct slow5_version version;
} hdr_t;

static char *dupn(const char *p, size_t n) { char *s = (char *)malloc(n + 1); if (s) { memcpy(s, p, n); s[n] = 0; } return s; }

/* split [p, p + n) at tabs: returns the number of fields, f[k] / fl[k] their starts and lengths (at most cap) */
static size_t split_tabs(const char *p, size_t n, const char **f, size_t *fl, size_t cap) {
    size_t k = 0, b = 0;
    while (b <= n && k < cap) {
        const char *t = (const char *)memchr(p + b, '\t', n - b);
        f[k] = p + b;
        fl[k] = t ? (size_t)(t - p) - b : n - b;
        b += fl[k] + 1;
        k++;
        if (!t) break;
    }
    return k;
}

static int hdr_parse(const slow5_file_t *s, hdr_t *h) {
    memset(h, 0, sizeof *h);
    h->version = s->header->version;
    h->n_rg = s->header->num_read_groups;
    const char *d = s->header->data;
    size_t len = s->header->data_len, b = 0;
    const char *types = NULL, *names = NULL;
    size_t tl = 0, nl = 0;
    while (b < len) {
        const char *e = (const char *)memchr(d + b, '\n', len - b);
        const size_t l = e ? (size_t)(e - d) - b : len - b;
        const char *line = d + b;
        if (l && line[0] == '@') {
            const char *f[4096];
            size_t fl[4096];
            const size_t k = split_tabs(line + 1, l - 1, f, fl, 4096);
            if (k < 1) return -1;
            attr_t *na = (attr_t *)realloc(h->attr, sizeof(attr_t) * (h->n_attr + 1));
            if (!na) return -1;
            h->attr = na;
            attr_t *a = &h->attr[h->n_attr++];
            a->name = dupn(f[0], fl[0]);
            a->val = (char **)calloc(h->n_rg ? h->n_rg : 1, sizeof(char *));
            for (size_t g = 0; g < h->n_rg; g++)
                if (g + 1 < k && !(fl[g + 1] == 1 && f[g + 1][0] == '.')) a->val[g] = dupn(f[g + 1], fl[g + 1]);
        } else if (l > 6 && memcmp(line, "#char*", 6) == 0) { types = line; tl = l; }
        else if (l > 8 && memcmp(line, "#read_id", 8) == 0) { names = line; nl = l; }
        b += l + 1;
    }
    if (!types || !names) return -1;
    const char *tf[1100], *nf[1100];
    size_t tfl[1100], nfl[1100];
    const size_t kt = split_tabs(types, tl, tf, tfl, 1100), kn = split_tabs(names, nl, nf, nfl, 1100);
    if (kt != kn || kt < 8) return -1;
    h->n_aux = kt - 8;
    h->aux = (auxcol_t *)calloc(h->n_aux ? h->n_aux : 1, sizeof(auxcol_t));
    for (size_t c = 0; c < h->n_aux; c++) { h->aux[c].name = dupn(nf[8 + c], nfl[8 + c]); h->aux[c].type = dupn(tf[8 + c], tfl[8 + c]); }
    return 0;
}
static const char *hdr_get(const hdr_t *h, const char *name, size_t g) {
    for (size_t i = 0; i < h->n_attr; i++) if (strcmp(h->attr[i].name, name) == 0) return h->attr[i].val[g];
    return NULL;
}
static int is_enum(const char *type) { return strncmp(type, "enum{", 5) == 0; }

/* the output header: attributes sorted by name, one value per output read group */
static attr_t *out_attr(hdr_t *o, const char *name) {
    size_t lo = 0;
    while (lo < o->n_attr && strcmp(o->attr[lo].name, name) < 0) lo++;
    if (lo < o->n_attr && strcmp(o->attr[lo].name, name) == 0) return &o->attr[lo];
    attr_t *na = (attr_t *)realloc(o->attr, sizeof(attr_t) * (o->n_attr + 1));
    if (!na) return NULL;
    o->attr = na;
    memmove(&o->attr[lo + 1], &o->attr[lo], sizeof(attr_t) * (o->n_attr - lo));
    o->n_attr++;
    o->attr[lo].name = strdup(name);
    o->attr[lo].val = (char **)calloc(4096, sizeof(char *));             /* (output read groups: at most 4096 here) */
    return &o->attr[lo];
}
static long out_aux_find(const hdr_t *o, const char *name) {
    for (size_t c = 0; c < o->n_aux; c++) if (strcmp(o->aux[c].name, name) == 0) return (long)c;
    return -1;
}

/* "a\tb\tc" aux tail of a line re-ordered for the output columns: map[c] = input column of output column c, or -1 (".") */
static char *remap_line(const char *line, size_t len, const long *map, size_t n_out, size_t n_in, size_t *out_len) {
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
    const char *f[1100];
    size_t fl[1100];
    const size_t k = split_tabs(line, len, f, fl, 1100);
    if (k != 8 + n_in) return NULL;
    const size_t head = (size_t)(f[7] + fl[7] - line);
    char *o = (char *)malloc(len + 2 * n_out + 2);
    if (!o) return NULL;
    memcpy(o, line, head);
    size_t at = head;
    for (size_t c = 0; c < n_out; c++) {
        o[at++] = '\t';
        if (map[c] < 0) o[at++] = '.';
        else { memcpy(o + at, f[8 + map[c]], fl[8 + map[c]]); at += fl[8 + map[c]]; }
    }
    *out_len = at;
    return o;
}

/* The work is done and every file is closed: leave without the HIP runtime's static destructors (code objects, memory pools: ~0.1 s of a
 * one-second job).  S5_FULL_EXIT=1 takes the ordinary way out (leak checkers). */
static int leave(void) {
    if (fflush(stdout) != 0) { fprintf(stderr, "%s: writing the standard output failed\n", "s5merge"); fflush(stderr); _exit(EXIT_FAILURE); }
    fflush(stderr);
    const char *e = getenv("S5_FULL_EXIT");
    if (e && atoi(e)) { s5gpu_shutdown(); return EXIT_SUCCESS; }
    _exit(EXIT_SUCCESS);
}
int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: s5merge out.blow5 [-c none|zlib|zstd] [-s none|svb-zd|ex-zd] [-l] [-K batch] in1 in2 ...\n"); return EXIT_FAILURE; }
    slow5_press_method_t to = {SLOW5_COMPRESS_ZLIB, SLOW5_COMPRESS_SVB_ZD};
    int lossy = 0;
    int64_t K = 4096;
    const char *files[4096];
    size_t nf = 0;
    for (int i = 2; i < argc; i++) {
        if (strcmp(argv[i], "-c") == 0 && i + 1 < argc) { i++; to.record_method = strcmp(argv[i], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(argv[i], "zstd") == 0 ? SLOW5_COMPRESS_ZSTD : SLOW5_COMPRESS_ZLIB; }
        else if (strcmp(argv[i], "-s") == 0 && i + 1 < argc) { i++; to.signal_method = strcmp(argv[i], "none") == 0 ? SLOW5_COMPRESS_NONE : strcmp(argv[i], "ex-zd") == 0 ? SLOW5_COMPRESS_EX_ZD : SLOW5_COMPRESS_SVB_ZD; }
        else if (strcmp(argv[i], "-l") == 0) lossy = 1;
        else if (strcmp(argv[i], "-K") == 0 && i + 1 < argc) K = atoll(argv[++i]);
        else if (nf < 4096) files[nf++] = argv[i];
    }
    if (nf == 0 || K < 1) return die("no input files", NULL);

    /* ---- pass 1: the output header and, per file, its read-group list and aux column map (merge.c:232-340) ---- */
    hdr_t out;
    memset(&out, 0, sizeof out);
    hdr_t *in = (hdr_t *)calloc(nf, sizeof(hdr_t));
    size_t **rgmap = (size_t **)calloc(nf, sizeof(size_t *));
    auxcol_t *plain = NULL;                                               /* the non-enum aux fields, kept sorted by name */
    size_t n_plain = 0;
    for (size_t i = 0; i < nf; i++) {
        slow5_file_t *s = slow5_open(files[i], "r");
        if (!s) return die("cannot open", files[i]);
        if (hdr_parse(s, &in[i]) != 0) return die("cannot read the header of", files[i]);
        slow5_close(s);
        const hdr_t *h = &in[i];
        if (h->version.major > out.version.major || (h->version.major == out.version.major && (h->version.minor > out.version.minor ||
            (h->version.minor == out.version.minor && h->version.patch > out.version.patch)))) out.version = h->version;
        if (!lossy && h->n_aux == 0) return die("no auxiliary fields (use -l) in", files[i]);          /* merge.c:241-245 */
        for (size_t c = 0; !lossy && c < h->n_aux; c++) {
            if (is_enum(h->aux[c].type)) {
                const long at = out_aux_find(&out, h->aux[c].name);
                if (at < 0) {
                    auxcol_t *na = (auxcol_t *)realloc(out.aux, sizeof(auxcol_t) * (out.n_aux + 1));
                    if (!na) return die("out of memory", NULL);
                    out.aux = na;
                    out.aux[out.n_aux].name = strdup(h->aux[c].name); out.aux[out.n_aux].type = strdup(h->aux[c].type);
                    out.n_aux++;
                } else if (strcmp(out.aux[at].type, h->aux[c].type) != 0) return die("different enum labels in different files for", h->aux[c].name);   /* merge.c:281-289 */
            } else {
                size_t lo = 0;
                while (lo < n_plain && strcmp(plain[lo].name, h->aux[c].name) < 0) lo++;
                if (lo < n_plain && strcmp(plain[lo].name, h->aux[c].name) == 0) {
                    /* STRICTER than the reference on purpose: merge.c:292 keeps the first file's type (std::map::insert ignores the second) and then
                     * copies the other file's aux bytes under it — a header that no longer describes its records.  Refused here. */
                    if (strcmp(plain[lo].type, h->aux[c].type) != 0) return die("different types in different files for", h->aux[c].name);
                } else {
                    auxcol_t *np = (auxcol_t *)realloc(plain, sizeof(auxcol_t) * (n_plain + 1));
                    if (!np) return die("out of memory", NULL);
                    plain = np;
                    memmove(&plain[lo + 1], &plain[lo], sizeof(auxcol_t) * (n_plain - lo));
                    plain[lo].name = strdup(h->aux[c].name); plain[lo].type = strdup(h->aux[c].type);
                    n_plain++;
                }
            }
        }
        rgmap[i] = (size_t *)calloc(h->n_rg ? h->n_rg : 1, sizeof(size_t));
        for (size_t j = 0; j < h->n_rg; j++) {
            const char *run_id = hdr_get(h, "run_id", j);
            if (!run_id) return die("no run_id in", files[i]);
            size_t k = 0;
            for (; k < out.n_rg; k++) {
                attr_t *ra = out_attr(&out, "run_id");
                if (ra->val[k] && strcmp(ra->val[k], run_id) == 0) break;
            }
            if (k < out.n_rg) {                                           /* same run_id: the same read group; its attributes must agree (merge.c:318-323) */
                for (size_t a = 0; a < h->n_attr; a++) {
                    const attr_t *oa = out_attr(&out, h->attr[a].name);
                    const char *x = oa->val[k], *y = h->attr[a].val[j];
                    if ((x == NULL) != (y == NULL) || (x && strcmp(x, y) != 0)) return die("attributes differ for the same run_id; attribute", h->attr[a].name);
                }
            } else {
                if (out.n_rg >= 4096) return die("more than 4096 read groups", NULL);
                for (size_t a = 0; a < h->n_attr; a++) {
                    attr_t *oa = out_attr(&out, h->attr[a].name);
                    if (!oa) return die("out of memory", NULL);
                    if (h->attr[a].val[j]) oa->val[out.n_rg] = strdup(h->attr[a].val[j]);
                }
                out.n_rg++;
            }
            rgmap[i][j] = k;
        }
    }
    for (size_t c = 0; c < n_plain; c++) {                                /* merge.c:344-350 */
        auxcol_t *na = (auxcol_t *)realloc(out.aux, sizeof(auxcol_t) * (out.n_aux + 1));
        if (!na) return die("out of memory", NULL);
        out.aux = na;
        out.aux[out.n_aux++] = plain[c];
    }
    /* header text */
    size_t cap = 1 << 16, at = 0;
    char *txt = (char *)malloc(cap);
#define PUT(p, n) do { const size_t n_ = (n); if (at + n_ + 2 > cap) { cap = (cap + n_) * 2; txt = (char *)realloc(txt, cap); if (!txt) return die("out of memory", NULL); } memcpy(txt + at, (p), n_); at += n_; } while (0)
    for (size_t a = 0; a < out.n_attr; a++) {
        PUT("@", 1); PUT(out.attr[a].name, strlen(out.attr[a].name));
        for (size_t g = 0; g < out.n_rg; g++) { PUT("\t", 1); if (out.attr[a].val[g]) PUT(out.attr[a].val[g], strlen(out.attr[a].val[g])); else PUT(".", 1); }
        PUT("\n", 1);
    }
    const size_t types_at = at;
    static const char TYPES8[] = "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*";
    static const char NAMES8[] = "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal";
    PUT(TYPES8, sizeof TYPES8 - 1);
    for (size_t c = 0; c < out.n_aux; c++) { PUT("\t", 1); PUT(out.aux[c].type, strlen(out.aux[c].type)); }
    const size_t types_len = at - types_at;
    PUT("\n", 1);
    PUT(NAMES8, sizeof NAMES8 - 1);
    for (size_t c = 0; c < out.n_aux; c++) { PUT("\t", 1); PUT(out.aux[c].name, strlen(out.aux[c].name)); }
    PUT("\n", 1);
    uint8_t out_types[1024];
    const int n_out_aux = s5gpu_aux_types_parse(txt + types_at, types_len, out_types, 1024);
    if (n_out_aux < 0 || (size_t)n_out_aux != out.n_aux) return die("cannot parse the merged aux types", NULL);
    struct slow5_hdr oh;
    memset(&oh, 0, sizeof oh);
    oh.version = out.version;
    oh.num_read_groups = (uint32_t)out.n_rg;
    oh.data = txt;
    oh.data_len = (uint32_t)at;
    FILE *fo = fopen(argv[1], "wb");
    if (!fo) return die("cannot open output", argv[1]);
    if (slow5_hdr_fwrite(fo, &oh, SLOW5_FORMAT_BINARY, to) < 0) return die("header write failed", NULL);

    /* ---- pass 2: the records, file after file, batches of K (merge.c:383-467) ---- */
    char **mem = (char **)calloc((size_t)K, sizeof(char *));
    size_t *bytes = (size_t *)calloc((size_t)K, sizeof(size_t));
    void **bufs = (void **)calloc((size_t)K, sizeof(void *)), **tmp = (void **)calloc((size_t)K, sizeof(void *));
    size_t *lens = (size_t *)calloc((size_t)K, sizeof(size_t)), *tlens = (size_t *)calloc((size_t)K, sizeof(size_t));
    uint32_t *nrg = (uint32_t *)calloc((size_t)K, sizeof(uint32_t));
    uint64_t total = 0, detour = 0;
    for (size_t i = 0; i < nf; i++) {
        slow5_file_t *s = slow5_open(files[i], "r");
        if (!s) return die("cannot open", files[i]);
        const hdr_t *h = &in[i];
        long map[1100];
        int identity = lossy || h->n_aux == out.n_aux;
        for (size_t c = 0; !lossy && c < out.n_aux; c++) {
            map[c] = -1;
            for (size_t q = 0; q < h->n_aux; q++) if (strcmp(h->aux[q].name, out.aux[c].name) == 0) map[c] = (long)q;
            if (map[c] != (long)c) identity = 0;
        }
        uint8_t in_types[1024];
        const uint32_t n_in_aux = s->header->aux_meta ? s->header->aux_meta->num : 0;
        if (n_in_aux) memcpy(in_types, s->header->aux_meta->types, n_in_aux);
        const slow5_press_method_t from = {s->compress->record_press->method, s->compress->signal_press->method};
        int eof = 0;
        while (!eof) {
            int64_t n = 0;
            while (n < K) {                                               /* read phase, merge.c:398-424 */
                mem[n] = (char *)slow5_get_next_mem(&bytes[n], s);
                if (!mem[n]) { if (slow5_errno != SLOW5_ERR_EOF) return die("bad record framing in", files[i]); eof = 1; break; }
                n++;
            }
            if (n == 0) break;
            /* compute phase: the work_db() of merge.c:440.  The new read group is list[file][old one] (merge.c:51); the device rewrite
             * takes a value per record, so the old read_group of a multi-group file has to be known here: one-group files (the usual case)
             * need no look; others take the detour below, where the column is in the text */
            int direct = identity && h->n_rg == 1;
            for (int64_t r = 0; r < n; r++) nrg[r] = (uint32_t)rgmap[i][0];
            int rc = 0;
            if (direct && s->format == SLOW5_FORMAT_BINARY)
                rc = s5gpu_recompress_batch((uint32_t)n, (const void *const *)mem, bytes, rec_code_of(from.record_method), sig_code_of(from.signal_method),
                                            rec_code_of(to.record_method), sig_code_of(to.signal_method), nrg, lossy, bufs, lens, NULL);
            else if (direct)
                rc = s5gpu_ascii_to_blow5_batch((uint32_t)n, (const char *const *)mem, bytes, n_in_aux, in_types, rec_code_of(to.record_method),
                                                sig_code_of(to.signal_method), nrg, lossy, bufs, lens, NULL);
            else {
                detour += (uint64_t)n;
                const char **lines = (const char **)mem;
                size_t *ll = bytes;
                if (s->format == SLOW5_FORMAT_BINARY) {                    /* records -> text lines first */
                    rc = s5gpu_blow5_to_ascii_batch((uint32_t)n, (const void *const *)mem, bytes, rec_code_of(from.record_method), sig_code_of(from.signal_method),
                                                    n_in_aux, in_types, NULL, 0, tmp, tlens, NULL);
                    if (rc) return die("GPU press path failed (records to text) for", files[i]);
                    lines = (const char **)tmp; ll = tlens;
                }
                for (int64_t r = 0; r < n; r++) {                          /* read_group column: list[file][old]; aux columns: the output's order */
                    const char *f[3];
                    size_t fl[3];
                    size_t len = ll[r];
                    while (len && (lines[r][len - 1] == '\n' || lines[r][len - 1] == '\r')) len--;
                    if (split_tabs(lines[r], len, f, fl, 3) < 3) return die("malformed record line in", files[i]);
                    const unsigned long old = strtoul(f[1], NULL, 10);
                    if (old >= h->n_rg) return die("read group out of range in", files[i]);
                    nrg[r] = (uint32_t)rgmap[i][old];
                    size_t nl2 = 0;
                    char *nl_ = lossy ? dupn(lines[r], len) : remap_line(lines[r], len, map, out.n_aux, h->n_aux, &nl2);
                    if (!nl_) return die("malformed record line (aux columns) in", files[i]);
                    if (lossy) nl2 = len;
                    if (s->format == SLOW5_FORMAT_BINARY) free(tmp[r]);
                    tmp[r] = nl_; tlens[r] = nl2;
                }
                rc = s5gpu_ascii_to_blow5_batch((uint32_t)n, (const char *const *)tmp, tlens, lossy ? h->n_aux : (uint32_t)out.n_aux, lossy ? in_types : out_types,
                                                rec_code_of(to.record_method), sig_code_of(to.signal_method), nrg, lossy, bufs, lens, NULL);
                for (int64_t r = 0; r < n; r++) { free(tmp[r]); tmp[r] = NULL; }
            }
            if (rc) return die("GPU press path failed for", files[i]);
            for (int64_t r = 0; r < n; r++) {                              /* ordered write phase, merge.c:444-447 */
                if (fwrite(bufs[r], 1, lens[r], fo) != lens[r]) return die("write failed", NULL);
                free(bufs[r]); free(mem[r]);
            }
            total += (uint64_t)n;
        }
        slow5_close(s);
    }
    if (slow5_eof_fwrite(fo) < 0) return die("eof write failed", NULL);
    if (fclose(fo) != 0) return die("closing the output failed (its last bytes may not be on disk)", NULL);
    fprintf(stderr, "s5merge: %llu records of %zu files into %zu read groups, %zu aux fields (%llu records through the text detour)\n",
            (unsigned long long)total, nf, out.n_rg, lossy ? (size_t)0 : out.n_aux, (unsigned long long)detour);
    return leave();
}
