/*
 * blow5_file.c — BLOW5 container framing around the press path, host C.
 *
 * slow5lib's slow5_open / slow5_get_next_mem / slow5_hdr_fwrite / slow5_eof_fwrite / slow5_idx_* as called
 * from /root/reference/src/view.c:192,246,265-278,313, src/get.c:286,45 and src/index.c.  slow5lib is an
 * absent submodule, so the layouts follow SURVEY.md Appendix A (verified on the golden files) and the
 * reference's own literal statement in test/misc/make_blow5.c:11-101.  Header attributes are kept as the opaque
 * text blob; SLOW5 ASCII files (same text after two '#' version lines, one record per line) are framed here too.  No codec work happens here: read ids for the index come from the GPU batch decode.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "../../include/slow5_compat.h"
#include "../../include/slow5gpu.h"

void slow5_compat_error(const char *fmt, ...);   /* slow5_compat.c: printed per log level, fatal under SLOW5_EXIT_ON_ERR */
void slow5_compat_warn(const char *fmt, ...);    /* ... a warning: printed from SLOW5_LOG_WARN on, never fatal on its own */
extern int slow5_compat_skip_rid;                /* slow5_set_skip_rid() */

static const char BLOW5_MAGIC[6] = {'B', 'L', 'O', 'W', '5', '\1'};
static const char BLOW5_EOF[5] = {'5', 'W', 'O', 'L', 'B'};
static const char IDX_MAGIC[9] = {'S', 'L', 'O', 'W', '5', 'I', 'D', 'X', '\1'};
static const char IDX_EOF[8] = {'X', 'D', 'I', '5', 'W', 'O', 'L', 'S'};

struct idx_ent { char *id; uint16_t id_len; uint64_t offset, size; };
struct slow5_idx {
    struct idx_ent *ents;   /* file order */
    uint64_t n, cap;
    uint32_t *table;        /* open addressing: index into ents + 1, 0 = empty */
    uint64_t tsize;
    struct slow5_version version;
    char **rids;            /* slow5_get_rids: the ids in file order (pointers into ents), made on first use */
    char *blob;             /* an index READ from its file: the file's bytes, the ids are NUL-terminated in place (not one malloc each) */
};

static enum slow5_press_method rec_from_code(uint8_t c) { return c == 0 ? SLOW5_COMPRESS_NONE : c == 1 ? SLOW5_COMPRESS_ZLIB : c == 2 ? SLOW5_COMPRESS_ZSTD : (enum slow5_press_method)-1; }
static enum slow5_press_method sig_from_code(uint8_t c) { return c == 0 ? SLOW5_COMPRESS_NONE : c == 1 ? SLOW5_COMPRESS_SVB_ZD : c == 2 ? SLOW5_COMPRESS_EX_ZD : (enum slow5_press_method)-1; }
static int rec_to_code(enum slow5_press_method m) { return m == SLOW5_COMPRESS_NONE ? 0 : m == SLOW5_COMPRESS_ZLIB ? 1 : m == SLOW5_COMPRESS_ZSTD ? 2 : -1; }
static int sig_to_code(enum slow5_press_method m) { return m == SLOW5_COMPRESS_NONE ? 0 : m == SLOW5_COMPRESS_SVB_ZD ? 1 : m == SLOW5_COMPRESS_EX_ZD ? 2 : -1; }

/* aux column types from the header text's types line (the line before "#read_id...") */
static int aux_meta_build(struct slow5_hdr *hd) {
    hd->aux_meta = NULL;
    if (hd->data_len < 2) return 0;
    /* last line = names, the one before = types */
    const char *d = hd->data;
    size_t end = hd->data_len;
    if (d[end - 1] == '\n') end--;
    size_t names = end;
    while (names > 0 && d[names - 1] != '\n') names--;
    if (names == 0) return 0;
    size_t tend = names - 1, types = tend;
    while (types > 0 && d[types - 1] != '\n') types--;
    uint8_t codes[1024];
    int n = s5gpu_aux_types_parse(d + types, tend - types, codes, 1024);
    if (n < 0) return -1;
    if (n == 0) return 0;
    struct slow5_aux_meta *am = (struct slow5_aux_meta *)calloc(1, sizeof *am);
    if (!am) return -1;
    am->types = (uint8_t *)malloc((size_t)n);
    if (!am->types) { free(am); return -1; }
    memcpy(am->types, codes, (size_t)n);
    am->num = (uint32_t)n;
    hd->aux_meta = am;
    return 0;
}

static slow5_file_t *open_ascii(FILE *fp, const char *pathname) {
    /* "#slow5_version\tM.m.p\n#num_read_groups\tN\n" then the header text up to and including the "#read_id" line */
    char *line = NULL;
    size_t cap = 0;
    ssize_t got;
    unsigned maj, min, pat, nrg;
    slow5_file_t *s = (slow5_file_t *)calloc(1, sizeof *s);
    struct slow5_hdr *hd = (struct slow5_hdr *)calloc(1, sizeof *hd);
    char *text = NULL;
    size_t tlen = 0, tcap = 0;
    int ok = 0;
    if (!s || !hd) goto done;
    if ((got = getline(&line, &cap, fp)) <= 0 || sscanf(line, "#slow5_version\t%u.%u.%u", &maj, &min, &pat) != 3) { slow5_errno = SLOW5_ERR_MAGIC; goto done; }
    if ((got = getline(&line, &cap, fp)) <= 0 || sscanf(line, "#num_read_groups\t%u", &nrg) != 1) { slow5_errno = SLOW5_ERR_TRUNC; goto done; }
    slow5_errno = SLOW5_ERR_TRUNC;
    for (;;) {
        if ((got = getline(&line, &cap, fp)) <= 0) goto done;
        if (line[0] != '#' && line[0] != '@') goto done;
        if (tlen + (size_t)got + 1 > tcap) {
            tcap = (tlen + (size_t)got + 1) * 2;
            char *nt = (char *)realloc(text, tcap);
            if (!nt) { slow5_errno = SLOW5_ERR_MEM; goto done; }
            text = nt;
        }
        memcpy(text + tlen, line, (size_t)got);
        tlen += (size_t)got;
        if (strncmp(line, "#read_id", 8) == 0) break;
    }
    if (tlen > 0xFFFFFFFFull) goto done;
    hd->version.major = (uint8_t)maj; hd->version.minor = (uint8_t)min; hd->version.patch = (uint8_t)pat;
    hd->num_read_groups = nrg;
    hd->data = text;
    hd->data_len = (uint32_t)tlen;
    text = NULL;
    if (aux_meta_build(hd) != 0) { slow5_errno = SLOW5_ERR_OTH; goto done; }
    s->fp = fp;
    s->format = SLOW5_FORMAT_ASCII;
    s->header = hd;
    s->meta.pathname = strdup(pathname);
    s->meta.start_rec_offset = (uint64_t)ftello(fp);
    {
        slow5_press_method_t m = {SLOW5_COMPRESS_NONE, SLOW5_COMPRESS_NONE};
        s->compress = slow5_press_init(m);
    }
    ok = s->compress != NULL;
done:
    free(line);
    free(text);
    if (!ok) {
        if (hd) { free(hd->data); if (hd->aux_meta) { free(hd->aux_meta->types); free(hd->aux_meta); } free(hd); }
        free(s);
        fclose(fp);
        return NULL;
    }
    slow5_errno = SLOW5_ERR_OK;
    return s;
}

slow5_file_t *slow5_open(const char *pathname, const char *mode) {
    if (!pathname || !mode || mode[0] != 'r') { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    FILE *fp = fopen(pathname, "rb");
    if (!fp) { slow5_errno = SLOW5_ERR_IO; return NULL; }
    uint8_t h[64];
    uint32_t hl = 0;
    size_t got = fread(h, 1, 64, fp);
    if (got >= 14 && memcmp(h, "#slow5_version", 14) == 0) {
        rewind(fp);
        return open_ascii(fp, pathname);
    }
    if (got != 64 || fread(&hl, 4, 1, fp) != 1) { fclose(fp); slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    if (memcmp(h, BLOW5_MAGIC, 6) != 0) { fclose(fp); slow5_errno = SLOW5_ERR_MAGIC; return NULL; }
    slow5_file_t *s = (slow5_file_t *)calloc(1, sizeof *s);
    struct slow5_hdr *hd = (struct slow5_hdr *)calloc(1, sizeof *hd);
    char *text = (char *)malloc(hl ? hl : 1);
    if (!s || !hd || !text) { free(s); free(hd); free(text); fclose(fp); slow5_errno = SLOW5_ERR_MEM; return NULL; }
    if (hl && fread(text, 1, hl, fp) != hl) { free(s); free(hd); free(text); fclose(fp); slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    hd->version.major = h[6]; hd->version.minor = h[7]; hd->version.patch = h[8];
    memcpy(&hd->num_read_groups, h + 10, 4);
    hd->data = text;
    hd->data_len = hl;
    slow5_press_method_t m = {rec_from_code(h[9]), sig_from_code(h[14])};
    s->fp = fp;
    s->format = SLOW5_FORMAT_BINARY;
    s->header = hd;
    s->meta.pathname = strdup(pathname);
    s->meta.start_rec_offset = 68ull + hl;
    s->compress = slow5_press_init(m);
    if (!s->compress) { slow5_close(s); slow5_errno = SLOW5_ERR_PRESS; return NULL; }
    if (aux_meta_build(hd) != 0) { slow5_close(s); slow5_errno = SLOW5_ERR_OTH; return NULL; }
    return s;
}

slow5_file_t *slow5_open_with(const char *pathname, const char *mode, enum slow5_fmt format) {
    if (format != SLOW5_FORMAT_UNKNOWN && format != SLOW5_FORMAT_ASCII && format != SLOW5_FORMAT_BINARY) { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    slow5_file_t *s = slow5_open(pathname, mode);
    if (s && format != SLOW5_FORMAT_UNKNOWN && s->format != format) {   /* the file is not what the caller said it is */
        slow5_close(s);
        slow5_errno = SLOW5_ERR_MAGIC;
        slow5_compat_error("'%s' is not a %s file", pathname, format == SLOW5_FORMAT_ASCII ? "SLOW5 ASCII" : "BLOW5");
        return NULL;
    }
    return s;
}

int slow5_close(slow5_file_t *s) {
    if (!s) return 0;
    slow5_idx_unload(s);
    if (s->fp) fclose(s->fp);
    if (s->header) {
        if (s->header->aux_meta) { free(s->header->aux_meta->types); free(s->header->aux_meta); }
        free(s->header->data);
        free(s->header);
    }
    slow5_press_free(s->compress);
    free((void *)s->meta.pathname);
    free(s);
    return 0;
}

void *slow5_get_next_mem(size_t *n, const slow5_file_t *s) {
    if (!s || !s->fp) { slow5_errno = SLOW5_ERR_ARG; return NULL; }
    if (s->format == SLOW5_FORMAT_ASCII) {
        char *line = NULL;
        size_t cap = 0;
        ssize_t got = getline(&line, &cap, s->fp);
        if (got <= 0) { free(line); slow5_errno = feof(s->fp) ? SLOW5_ERR_EOF : SLOW5_ERR_IO; return NULL; }
        if (line[got - 1] == '\n') line[--got] = 0;   /* the line without its newline */
        if (n) *n = (size_t)got;
        slow5_errno = SLOW5_ERR_OK;
        return line;
    }
    uint8_t pre[8];
    size_t got = fread(pre, 1, 8, s->fp);
    if (got >= 5 && memcmp(pre, BLOW5_EOF, 5) == 0) {
        /* the marker must be the last thing in the file (src/quickcheck.c:93-97) */
        int trailing = got > 5 || fgetc(s->fp) != EOF;
        slow5_errno = trailing ? SLOW5_ERR_TRUNC : SLOW5_ERR_EOF;
        return NULL;
    }
    if (got != 8) { slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    uint64_t sz;
    memcpy(&sz, pre, 8);
    void *mem = malloc(sz ? sz : 1);
    if (!mem) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
    if (fread(mem, 1, sz, s->fp) != sz) { free(mem); slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    if (n) *n = sz;
    slow5_errno = SLOW5_ERR_OK;
    return mem;
}

int slow5_get_next_bytes(char **mem, size_t *bytes, slow5_file_t *s) {
    if (!mem || !bytes) { slow5_errno = SLOW5_ERR_ARG; return SLOW5_ERR_ARG; }
    size_t n = 0;
    char *m = (char *)slow5_get_next_mem(&n, s);
    if (!m) {
        if (slow5_errno != SLOW5_ERR_EOF) slow5_compat_error("slow5_get_next_bytes: the next record could not be read (slow5_errno %d)", slow5_errno);
        return slow5_errno ? slow5_errno : SLOW5_ERR_UNK;
    }
    *mem = m;
    *bytes = n;
    return 0;
}

int slow5_hdr_fwrite(FILE *fp, struct slow5_hdr *header, enum slow5_fmt format, slow5_press_method_t comp) {
    if (fp && header && format == SLOW5_FORMAT_ASCII) {
        int w = fprintf(fp, "#slow5_version\t%u.%u.%u\n#num_read_groups\t%u\n", header->version.major, header->version.minor,
                        header->version.patch, header->num_read_groups);
        if (w < 0 || (header->data_len && fwrite(header->data, 1, header->data_len, fp) != header->data_len)) { slow5_errno = SLOW5_ERR_IO; return -1; }
        return w + (int)header->data_len;
    }
    if (!fp || !header || format != SLOW5_FORMAT_BINARY || rec_to_code(comp.record_method) < 0 || sig_to_code(comp.signal_method) < 0) {
        slow5_errno = SLOW5_ERR_ARG;
        return -1;
    }
    uint8_t h[64];
    memset(h, 0, sizeof h);
    memcpy(h, BLOW5_MAGIC, 6);
    struct slow5_version v = header->version;
    if (comp.signal_method != SLOW5_COMPRESS_NONE && v.major == 0 && v.minor < 2) { v.minor = 2; v.patch = 0; }
    h[6] = v.major; h[7] = v.minor; h[8] = v.patch;
    h[9] = (uint8_t)rec_to_code(comp.record_method);
    memcpy(h + 10, &header->num_read_groups, 4);
    h[14] = (uint8_t)sig_to_code(comp.signal_method);
    if (fwrite(h, 1, 64, fp) != 64 || fwrite(&header->data_len, 4, 1, fp) != 1 ||
        (header->data_len && fwrite(header->data, 1, header->data_len, fp) != header->data_len)) {
        slow5_errno = SLOW5_ERR_IO;
        return -1;
    }
    return (int)(68 + header->data_len);
}

long slow5_eof_fwrite(FILE *fp) {
    if (!fp || fwrite(BLOW5_EOF, 1, 5, fp) != 5) { slow5_errno = SLOW5_ERR_IO; return -1; }
    return 5;
}

/* ---- index ---- */
static uint64_t hash_id(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; }
    return h;
}
static int idx_push(struct slow5_idx *ix, const char *id, uint16_t id_len, uint64_t off, uint64_t size) {
    if (ix->n == ix->cap) {
        uint64_t nc = ix->cap ? ix->cap * 2 : 1024;
        struct idx_ent *ne = (struct idx_ent *)realloc(ix->ents, nc * sizeof *ne);
        if (!ne) return -1;
        ix->ents = ne;
        ix->cap = nc;
    }
    struct idx_ent *e = &ix->ents[ix->n++];
    e->id = (char *)malloc((size_t)id_len + 1);
    if (!e->id) return -1;
    memcpy(e->id, id, id_len);
    e->id[id_len] = '\0';
    e->id_len = id_len;
    e->offset = off;
    e->size = size;
    return 0;
}
/* The lookup table of a big index is built by a few threads: a million ids are a million hash computations and a million cache misses
 * into an 8 MB table — 0.1 s on one core, which was a third of `get`'s whole process (round 5 review).  Open addressing, insertion by
 * compare-and-swap on the 32-bit slot: any interleaving of the threads leaves every entry findable by idx_find's linear probe. */
struct idx_tab_job { struct slow5_idx *ix; uint64_t lo, hi; };
static void *idx_tab_worker(void *arg) {
    struct idx_tab_job *j = (struct idx_tab_job *)arg;
    struct slow5_idx *ix = j->ix;
    const uint64_t t = ix->tsize;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        uint64_t p = hash_id(ix->ents[i].id, ix->ents[i].id_len) & (t - 1);
        for (;;) {
            if (ix->table[p] == 0 && __sync_bool_compare_and_swap(&ix->table[p], 0u, (uint32_t)(i + 1))) break;
            p = (p + 1) & (t - 1);
        }
    }
    return NULL;
}
static int idx_build_table(struct slow5_idx *ix) {
    uint64_t t = 16;
    while (t < 2 * ix->n + 1) t <<= 1;
    ix->table = (uint32_t *)calloc(t, sizeof(uint32_t));
    if (!ix->table) return -1;
    ix->tsize = t;
    enum { MAXT = 8 };
    long hw = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = ix->n < 65536 ? 1 : hw >= MAXT ? MAXT : hw > 1 ? (int)hw : 1;
    pthread_t th[MAXT];
    int live[MAXT];
    struct idx_tab_job job[MAXT];
    for (int k = 0; k < nt; k++) {
        job[k].ix = ix; job[k].lo = ix->n * (uint64_t)k / (uint64_t)nt; job[k].hi = ix->n * (uint64_t)(k + 1) / (uint64_t)nt;
        live[k] = k + 1 < nt && pthread_create(&th[k], NULL, idx_tab_worker, &job[k]) == 0;
        if (!live[k]) idx_tab_worker(&job[k]);           /* the last share, or one that got no thread: on the caller's */
    }
    for (int k = 0; k < nt; k++) if (live[k]) pthread_join(th[k], NULL);
    return 0;
}
static const struct idx_ent *idx_find(const struct slow5_idx *ix, const char *id) {
    const size_t n = strlen(id);
    uint64_t p = hash_id(id, n) & (ix->tsize - 1);
    while (ix->table[p]) {
        const struct idx_ent *e = &ix->ents[ix->table[p] - 1];
        if (e->id_len == n && memcmp(e->id, id, n) == 0) return e;
        p = (p + 1) & (ix->tsize - 1);
    }
    return NULL;
}
static void idx_free(struct slow5_idx *ix) {
    if (!ix) return;
    if (!ix->blob) for (uint64_t i = 0; i < ix->n; i++) free(ix->ents[i].id);
    free(ix->blob);
    free(ix->ents);
    free(ix->table);
    free(ix->rids);
    free(ix);
}
void slow5_idx_unload(slow5_file_t *s) {
    if (s && s->index) { idx_free(s->index); s->index = NULL; }
}

static char *idx_path(const slow5_file_t *s) {
    char *p = (char *)malloc(strlen(s->meta.pathname) + 5);
    if (p) { strcpy(p, s->meta.pathname); strcat(p, ".idx"); }
    return p;
}

/* Scan the records.  The file is read in chunks of tens of MB, the records are framed in place (their size prefixes chain through the
 * chunk; a record the chunk's end cuts is carried into the next one) and the read ids of a chunk come from ONE s5gpu_record_ids_stream
 * call, which inflates only the head of each record (k_inflate_head) — round 2 decoded every whole record, signal included, with three
 * mallocs each, to read a 36-byte id.  Records that call declines (zstd frames, ids longer than 256 bytes) take the general decode. */
static int idx_ids_general(slow5_file_t *s, struct slow5_idx *ix, const uint8_t *chunk, const uint64_t *pos, const uint32_t *len, const uint64_t *foff,
                           const uint32_t *which, uint32_t m) {
    slow5_press_method_t from = {s->compress->record_press->method, s->compress->signal_press->method};
    char **mem = (char **)calloc(m, sizeof(char *));
    size_t *bytes = (size_t *)calloc(m, sizeof(size_t));
    struct slow5_rec **reads = (struct slow5_rec **)calloc(m, sizeof(void *));
    int err = (!mem || !bytes || !reads) ? SLOW5_ERR_MEM : 0;
    for (uint32_t k = 0; k < m && !err; k++) {
        bytes[k] = len[which[k]];
        mem[k] = (char *)malloc(bytes[k] ? bytes[k] : 1);
        if (!mem[k]) err = SLOW5_ERR_MEM; else memcpy(mem[k], chunk + pos[which[k]], bytes[k]);
    }
    if (!err && slow5_gpu_depress_parse_batch(m, mem, bytes, from, reads) != 0) err = slow5_errno ? slow5_errno : SLOW5_ERR_RECPARSE;
    for (uint32_t k = 0; k < m && !err; k++) {
        struct idx_ent *e = &ix->ents[foff[which[k]]];          /* foff: index of the entry reserved for this record */
        e->id = (char *)malloc((size_t)reads[k]->read_id_len + 1);
        if (!e->id) { err = SLOW5_ERR_MEM; break; }
        memcpy(e->id, reads[k]->read_id, reads[k]->read_id_len);
        e->id[reads[k]->read_id_len] = '\0';
        e->id_len = reads[k]->read_id_len;
    }
    for (uint32_t k = 0; k < m; k++) { if (mem) free(mem[k]); if (reads) slow5_rec_free(reads[k]); }
    free(mem); free(bytes); free(reads);
    return err;
}

static struct slow5_idx *idx_scan(slow5_file_t *s) {
    enum { ID_PITCH = 256 };
    struct slow5_idx *ix = (struct slow5_idx *)calloc(1, sizeof *ix);
    if (!ix) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
    ix->version = s->header->version;
    struct stat fst;
    const int fd = fileno(s->fp);
    if (fstat(fd, &fst) != 0 || (uint64_t)fst.st_size < s->meta.start_rec_offset + 5) { idx_free(ix); slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    {   /* the end marker must close the file (src/quickcheck.c:93-97) */
        char tail[5];
        if (pread(fd, tail, 5, fst.st_size - 5) != 5 || memcmp(tail, BLOW5_EOF, 5) != 0) { idx_free(ix); slow5_errno = SLOW5_ERR_TRUNC; return NULL; }
    }
    const char *ce = getenv("SLOW5_IDX_CHUNK_KB");                       /* tests: chunks smaller than a record */
    size_t chunk = ce && atoi(ce) > 0 ? (size_t)atoi(ce) << 10 : (size_t)64 << 20;
    const int rec_code = rec_to_code(s->compress->record_press->method);
    uint64_t pos = s->meta.start_rec_offset;
    const uint64_t end = (uint64_t)fst.st_size - 5;
    /* the work buffers follow the chunk size (chunk / 48 descriptors, 256 id bytes each): a small file gets small ones */
    if (chunk > end - pos) chunk = (size_t)(end - pos) > 4096 ? (size_t)(end - pos) : 4096;
    uint8_t *buf = (uint8_t *)malloc(chunk + 64);
    uint32_t cap = (uint32_t)(chunk / 48) + 16;
    uint64_t *rpos = (uint64_t *)malloc(sizeof(uint64_t) * cap), *ent = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    uint32_t *rlen = (uint32_t *)malloc(sizeof(uint32_t) * cap), *redo = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    char *ids = (char *)malloc((size_t)cap * ID_PITCH);
    uint16_t *idl = (uint16_t *)malloc(sizeof(uint16_t) * cap);
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * cap);
    int err = (!buf || !rpos || !ent || !rlen || !redo || !ids || !idl || !st) ? SLOW5_ERR_MEM : 0;
    size_t carry = 0;
    while (!err && (pos < end || carry)) {
        size_t want = chunk - carry;
        if (want > end - pos) want = (size_t)(end - pos);
        size_t got = 0;
        while (got < want) {
            ssize_t r = pread(fd, buf + carry + got, want - got, (off_t)(pos + got));
            if (r <= 0) { err = SLOW5_ERR_IO; break; }
            got += (size_t)r;
        }
        if (err) break;
        const uint64_t file_base = pos - carry;                          /* file offset of buf[0] */
        pos += want;
        const size_t have = carry + want;
        size_t p = 0;
        uint32_t n = 0;
        while (p + 8 <= have && n < cap) {
            uint64_t sz;
            memcpy(&sz, buf + p, 8);
            if (sz > 0xFFFFFF00ull || sz > end - (file_base + p + 8)) { err = SLOW5_ERR_TRUNC; break; }   /* the record would leave the file: nothing is grown for it */
            if (sz > chunk - 8) {                                         /* a record larger than the chunk: grow the chunk and start this round over */
                const size_t nc = (size_t)sz + 8 + (chunk >> 1);
                uint8_t *nb = (uint8_t *)realloc(buf, nc + 64);
                if (!nb) { err = SLOW5_ERR_MEM; break; }
                buf = nb; chunk = nc;
                break;
            }
            if (p + 8 + sz > have) break;
            rpos[n] = p + 8; rlen[n] = (uint32_t)sz;
            /* reserve the entry now (file order); its id is filled in below */
            if (idx_push(ix, "", 0, file_base + p, 8 + sz) != 0) { err = SLOW5_ERR_MEM; break; }
            free(ix->ents[ix->n - 1].id); ix->ents[ix->n - 1].id = NULL;
            ent[n] = ix->n - 1;
            n++;
            p += 8 + (size_t)sz;
        }
        if (err) break;
        if (n == 0 && pos >= end && have - p > 0 && have < chunk) { err = SLOW5_ERR_TRUNC; break; }   /* a cut record at the end of the file */
        if (n) {
            uint32_t m = 0;
            if (rec_code == 2 || s5gpu_record_ids_stream(n, buf, have, rpos, rlen, rec_code == 1 ? S5GPU_REC_ZLIB : S5GPU_REC_NONE, ID_PITCH, ids, idl, st) != S5GPU_OK) {
                if (rec_code != 2) { err = SLOW5_ERR_RECPARSE; break; }
                for (uint32_t i = 0; i < n; i++) redo[m++] = i;           /* zstd frames: the general decode */
            } else {
                for (uint32_t i = 0; i < n; i++) {
                    if (st[i] != 0) { redo[m++] = i; continue; }
                    struct idx_ent *e = &ix->ents[ent[i]];
                    e->id = (char *)malloc((size_t)idl[i] + 1);
                    if (!e->id) { err = SLOW5_ERR_MEM; break; }
                    memcpy(e->id, ids + (size_t)i * ID_PITCH, idl[i]);
                    e->id[idl[i]] = '\0';
                    e->id_len = idl[i];
                }
            }
            if (!err && m) err = idx_ids_general(s, ix, buf, rpos, rlen, ent, redo, m);
        }
        carry = have - p;
        if (carry && p) memmove(buf, buf + p, carry);
        if (n == 0 && carry && pos >= end && carry < 8) { err = SLOW5_ERR_TRUNC; break; }
    }
    free(buf); free(rpos); free(ent); free(rlen); free(redo); free(ids); free(idl); free(st);
    if (!err) for (uint64_t i = 0; i < ix->n; i++) if (!ix->ents[i].id) { err = SLOW5_ERR_RECPARSE; break; }
    if (!err && idx_build_table(ix) != 0) err = SLOW5_ERR_MEM;
    if (err) { idx_free(ix); slow5_errno = err; return NULL; }
    return ix;
}

static int idx_write(const struct slow5_idx *ix, const char *path) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    uint8_t h[64];
    memset(h, 0, sizeof h);
    memcpy(h, IDX_MAGIC, 9);
    h[9] = ix->version.major; h[10] = ix->version.minor; h[11] = ix->version.patch;
    int ok = fwrite(h, 1, 64, fp) == 64;
    for (uint64_t i = 0; ok && i < ix->n; i++) {
        const struct idx_ent *e = &ix->ents[i];
        ok = fwrite(&e->id_len, 2, 1, fp) == 1 && fwrite(e->id, 1, e->id_len, fp) == e->id_len &&
             fwrite(&e->offset, 8, 1, fp) == 1 && fwrite(&e->size, 8, 1, fp) == 1;
    }
    ok = ok && fwrite(IDX_EOF, 1, 8, fp) == 8;
    return fclose(fp) == 0 && ok ? 0 : -1;
}

/* The whole file in one read; the entries are walked once, ids stay where they are (NUL-terminated in place: the byte behind an id is the
 * first byte of its offset field, copied out just before) — no fread, malloc and memcpy per entry (round 6: 1 M entries 0.16 -> 0.03 s). */
static struct slow5_idx *idx_read(const char *path) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    struct stat st;
    struct slow5_idx *ix = (struct slow5_idx *)calloc(1, sizeof *ix);
    int ok = ix && fstat(fd, &st) == 0 && st.st_size >= 64 + 8;
    size_t len = ok ? (size_t)st.st_size : 0;
    if (ok) { ix->blob = (char *)malloc(len + 1); ok = ix->blob != NULL; }
    for (size_t got = 0; ok && got < len;) {
        const ssize_t r = read(fd, ix->blob + got, len - got);
        if (r <= 0) ok = 0; else got += (size_t)r;
    }
    close(fd);
    const uint8_t *b = ok ? (const uint8_t *)ix->blob : NULL;
    ok = ok && memcmp(b, IDX_MAGIC, 9) == 0 && memcmp(b + len - 8, IDX_EOF, 8) == 0;
    if (ok) { ix->version.major = b[9]; ix->version.minor = b[10]; ix->version.patch = b[11]; }
    const size_t end = ok ? len - 8 : 0;           /* the end marker is the file's last eight bytes, whatever an entry might look like */
    if (ok) {                                      /* (an entry is at least 18 bytes: a first guess of the count saves the reallocs) */
        ix->cap = (end - 64) / 40 + 16;
        ix->ents = (struct idx_ent *)malloc(ix->cap * sizeof *ix->ents);
        ok = ix->ents != NULL;
    }
    for (size_t p = 64; ok && p < end;) {
        uint16_t l;
        if (end - p < 2) { ok = 0; break; }
        memcpy(&l, b + p, 2);
        if (end - p < (size_t)2 + l + 16) { ok = 0; break; }
        if (ix->n == ix->cap) {
            const uint64_t nc = ix->cap * 2;
            struct idx_ent *ne = (struct idx_ent *)realloc(ix->ents, nc * sizeof *ne);
            if (!ne) { ok = 0; break; }
            ix->ents = ne; ix->cap = nc;
        }
        struct idx_ent *e = &ix->ents[ix->n++];
        e->id = ix->blob + p + 2;
        e->id_len = l;
        memcpy(&e->offset, b + p + 2 + l, 8);
        memcpy(&e->size, b + p + 2 + l + 8, 8);
        ix->blob[p + 2 + l] = '\0';
        p += (size_t)2 + l + 16;
    }
    if (ok && idx_build_table(ix) != 0) ok = 0;
    if (!ok) { idx_free(ix); return NULL; }
    return ix;
}

int slow5_idx_create(slow5_file_t *s) {
    if (!s || !s->fp || s->format != SLOW5_FORMAT_BINARY) { slow5_errno = SLOW5_ERR_ARG; return -1; }   /* BLOW5 only here */
    struct slow5_idx *ix = idx_scan(s);
    if (!ix) return -1;
    char *p = idx_path(s);
    int rc = p ? idx_write(ix, p) : -1;
    free(p);
    idx_free(ix);
    if (rc != 0) slow5_errno = SLOW5_ERR_IO;
    return rc;
}

int slow5_idx_load(slow5_file_t *s) {
    if (!s || !s->fp || s->format != SLOW5_FORMAT_BINARY) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    if (s->index) return 0;
    char *p = idx_path(s);
    if (!p) { slow5_errno = SLOW5_ERR_MEM; return -1; }
    if (access(p, R_OK) != 0 && slow5_idx_create(s) != 0) { free(p); return -1; }
    s->index = idx_read(p);
    if (s->index) {
        /* an index is only as good as the file it was made from: same version, every entry inside the file (size covers the
         * u64 prefix).  An index OLDER than the BLOW5 is only warned about and still used, as slow5lib does: mtimes have one-second
         * granularity and get reordered by cp / rsync, the directory may be read-only, and the index is the user's file */
        struct stat fs, is;
        int stale = 0;
        const struct slow5_version fv = s->header->version, iv = s->index->version;
        if (iv.major != fv.major || iv.minor != fv.minor || iv.patch != fv.patch) stale = 1;
        if (!stale && fstat(fileno(s->fp), &fs) == 0) {
            const uint64_t fsz = (uint64_t)fs.st_size;
            for (uint64_t i = 0; i < s->index->n && !stale; i++) {
                const struct idx_ent *e = &s->index->ents[i];
                if (e->size < 8 || e->offset < s->meta.start_rec_offset || e->offset > fsz || e->size > fsz - e->offset) stale = 1;   /* (no sum: it could wrap) */
            }
            if (!stale && stat(p, &is) == 0 && is.st_mtime < fs.st_mtime)
                slow5_compat_warn("index '%s' is older than '%s'; using it all the same (re-create it with slow5_idx_create if the file was rewritten)", p, s->meta.pathname);
        }
        if (stale) {
            slow5_idx_unload(s);
            slow5_compat_error("index '%s' does not belong to this file (version or record extents differ); remove it", p);
        }
    }
    free(p);
    if (!s->index) { slow5_errno = SLOW5_ERR_NOIDX; return -1; }
    return 0;
}

int slow5_idx_get(struct slow5_idx *index, const char *read_id, struct slow5_rec_idx *read_index) {
    if (!index || !read_id || !read_index) { slow5_errno = SLOW5_ERR_ARG; return -1; }
    const struct idx_ent *e = idx_find(index, read_id);
    if (!e) { slow5_errno = SLOW5_ERR_NOTFOUND; return -1; }
    read_index->offset = e->offset;
    read_index->size = e->size;
    return 0;
}

char **slow5_get_rids(const slow5_file_t *s, uint64_t *n) {
    if (!s || !s->index || !n) { slow5_errno = SLOW5_ERR_NOIDX; return NULL; }
    struct slow5_idx *ix = s->index;
    if (!ix->rids) {
        ix->rids = (char **)malloc(sizeof(char *) * (size_t)(ix->n ? ix->n : 1));
        if (!ix->rids) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
        for (uint64_t i = 0; i < ix->n; i++) ix->rids[i] = ix->ents[i].id;
    }
    *n = ix->n;
    return ix->rids;
}

void *slow5_get_mem(const char *read_id, size_t *n, const slow5_file_t *s) {
    if (!s || !s->index || !read_id) { slow5_errno = SLOW5_ERR_NOIDX; return NULL; }
    const struct idx_ent *e = idx_find(s->index, read_id);
    if (!e) {
        slow5_errno = SLOW5_ERR_NOTFOUND;
        if (!slow5_compat_skip_rid) slow5_compat_error("read id '%s' is not in the index", read_id);
        return NULL;
    }
    const size_t sz = (size_t)(e->size - 8);
    void *mem = malloc(sz ? sz : 1);
    if (!mem) { slow5_errno = SLOW5_ERR_MEM; return NULL; }
    if (pread(fileno(s->fp), mem, sz, (off_t)(e->offset + 8)) != (ssize_t)sz) { free(mem); slow5_errno = SLOW5_ERR_IO; return NULL; }   /* thread-safe like slow5_get */
    if (n) *n = sz;
    return mem;
}

int slow5_get(const char *read_id, struct slow5_rec **read, slow5_file_t *s) {
    size_t n = 0;
    char *mem = (char *)slow5_get_mem(read_id, &n, s);
    if (!mem) return slow5_errno;
    int rc = slow5_rec_depress_parse(&mem, &n, read_id, read, s);
    free(mem);
    return rc == 0 ? 0 : (slow5_errno ? slow5_errno : SLOW5_ERR_RECPARSE);
}
