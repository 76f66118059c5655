/*
 * slow5_compat.h — the slice of slow5lib's C API that sits on the record press path, implemented
 * on the MI355X kernels (libslow5gpu.so).  Same names, argument meaning, ownership and error
 * behaviour as the calls slow5tools makes:
 *
 *   slow5_press_init / slow5_press_free      /root/reference/src/view.c:43,54  merge.c:52,67  get.c:54,61
 *   slow5_rec_to_mem                         /root/reference/src/view.c:49     merge.c:62     get.c:59
 *   slow5_rec_depress_parse                  /root/reference/src/view.c:38     merge.c:46     split.c:84
 *   slow5_rec_free                           /root/reference/src/view.c:56
 *   slow5_ptr_compress_solo / _depress_solo  (slow5lib public press API named by BASELINE north_star;
 *                                             no call site inside slow5tools — SURVEY.md §8b last row)
 * plus the batch hooks that replace work_db() (src/thread.c:114) for view / merge / get.
 *
 * slow5lib itself is an absent submodule of the reference (/root/reference/.gitmodules:1-3), so the
 * struct layouts below follow the field uses visible at the call sites, not
 * slow5lib's private headers.  Two deliberate differences, both outside the press path:
 *   - aux fields travel as their already-serialised BLOW5 bytes (aux_blob / aux_len) instead of
 *     slow5lib's khash map: the aux/header attribute API is out of scope;
 *   - SLOW5 ASCII is handled at the same level: slow5_open / slow5_get_next_mem / slow5_hdr_fwrite
 *     understand .slow5 files and slow5_gpu_convert_batch parses / prints whole batches of record lines.
 * A maintainer integrating into the real slow5lib keeps slow5lib's structs and calls the s5gpu_*
 * functions from slow5lib's own slow5_rec_to_mem / slow5_rec_depress_parse — see INTEGRATION.md.
 */
#ifndef SLOW5_COMPAT_H
#define SLOW5_COMPAT_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* names from /root/reference/src/misc.c:253-263 (API enum; the on-disk codes of Appendix A.1 are mapped in blow5_file.c) */
enum slow5_press_method {
    SLOW5_COMPRESS_NONE = 0,
    SLOW5_COMPRESS_ZLIB = 1,
    SLOW5_COMPRESS_SVB_ZD = 2,
    SLOW5_COMPRESS_ZSTD = 3,
    SLOW5_COMPRESS_EX_ZD = 4   /* signal press only (the `degrade` default, src/degrade.c:302); layout pinned on the reference's fixtures */
};
typedef struct {
    enum slow5_press_method record_method;   /* core->press_method.record_method, src/demux.c:1072 */
    enum slow5_press_method signal_method;
} slow5_press_method_t;

struct __slow5_press { enum slow5_press_method method; void *stream; };
struct slow5_press {                          /* fields read at src/stats.c:114,128; src/cat.c:227-228 */
    struct __slow5_press *record_press;
    struct __slow5_press *signal_press;
};
typedef struct slow5_press slow5_press_t;

enum slow5_fmt { SLOW5_FORMAT_UNKNOWN = 0, SLOW5_FORMAT_ASCII = 1, SLOW5_FORMAT_BINARY = 2 };

struct slow5_aux_meta {                       /* slow5tools only tests it against NULL (src/merge.c:58-62); here it carries */
    uint32_t num;                             /* the aux column types of the header's types line, S5GPU_AUX_* codes        */
    uint8_t *types;
};
typedef struct slow5_aux_meta slow5_aux_meta_t;

struct slow5_rec {                            /* field uses: src/read_fast5.c:665-666,736-763,1136-1138; src/merge.c:51 */
    uint16_t read_id_len;
    char *read_id;
    uint32_t read_group;
    double digitisation;
    double offset;
    double range;
    double sampling_rate;
    uint64_t len_raw_signal;
    int16_t *raw_signal;
    uint8_t *aux_blob;                        /* serialised aux fields (see header comment) */
    uint64_t aux_len;
};
typedef struct slow5_rec slow5_rec_t;

struct slow5_version { uint8_t major, minor, patch; };   /* src/stats.c:98 prints header->version.{major,minor,patch} */
struct slow5_hdr {                            /* framing view of the header: the attribute API is out of scope */
    struct slow5_version version;
    uint32_t num_read_groups;                 /* src/stats.c:109 */
    char *data;                               /* the header text exactly as stored after the u32 length (Appendix A.1) */
    uint32_t data_len;
    struct slow5_aux_meta *aux_meta;          /* NULL when the file has no aux columns (src/merge.c:58) */
};
typedef struct slow5_hdr slow5_hdr_t;
struct slow5_idx;                             /* read_id -> (offset, size), Appendix A.5 */

struct slow5_file {                           /* fields read at src/stats.c:98-114, src/quickcheck.c:88-89 */
    FILE *fp;
    enum slow5_fmt format;
    struct slow5_press *compress;
    struct slow5_hdr *header;
    struct slow5_idx *index;
    struct { const char *pathname; uint64_t start_rec_offset; } meta;
};
typedef struct slow5_file slow5_file_t;

extern __thread int slow5_errno;              /* thread-local like slow5lib's (workers run concurrently) */
/* slow5lib's numbering (slow5_error.h), so that a caller comparing against slow5lib's codes reads failures right */
enum { SLOW5_ERR_OK = 0, SLOW5_ERR_EOF = -1, SLOW5_ERR_ARG = -2, SLOW5_ERR_TRUNC = -3, SLOW5_ERR_RECPARSE = -4, SLOW5_ERR_IO = -5,
       SLOW5_ERR_NOIDX = -6, SLOW5_ERR_NOTFOUND = -7, SLOW5_ERR_OTH = -8, SLOW5_ERR_UNK = -9, SLOW5_ERR_MEM = -10, SLOW5_ERR_NOAUX = -11,
       SLOW5_ERR_NOFLD = -12, SLOW5_ERR_PRESS = -13, SLOW5_ERR_MAGIC = -14, SLOW5_ERR_VERSION = -15, SLOW5_ERR_HDRPARSE = -16,
       SLOW5_ERR_TYPE = -17 };

/* process-wide switches slow5tools sets once at start-up: /root/reference/src/main.c:246-247, src/get.c:194 */
enum slow5_log_level_opt { SLOW5_LOG_OFF = 0, SLOW5_LOG_ERR, SLOW5_LOG_WARN, SLOW5_LOG_INFO, SLOW5_LOG_VERB, SLOW5_LOG_DBUG };
enum slow5_exit_condition_opt { SLOW5_EXIT_OFF = 0, SLOW5_EXIT_ON_ERR, SLOW5_EXIT_ON_WARN };
void slow5_set_log_level(enum slow5_log_level_opt log_level);          /* what this layer prints to stderr (default: INFO) */
void slow5_set_exit_condition(enum slow5_exit_condition_opt exit_condition);   /* exit(EXIT_FAILURE) after an error / a warning (default: off) */
void slow5_set_skip_rid(void);            /* slow5_get of an unknown read id is no longer an error message, only SLOW5_ERR_NOTFOUND */

struct slow5_press *slow5_press_init(slow5_press_method_t method);
void slow5_press_free(struct slow5_press *comp);

/* one-shot (de)compression of a byte range; returns a malloc'd buffer, NULL on error.  Methods: zlib, zstd (bytes),
 * svb-zd, ex-zd (ptr = int16 samples, count in bytes). */
void *slow5_ptr_compress_solo(enum slow5_press_method method, const void *ptr, size_t count, size_t *n);
void *slow5_ptr_depress_solo(enum slow5_press_method method, const void *ptr, size_t count, size_t *n);
/* the stateful forms (SURVEY 8b last row): the method comes from one half of a slow5_press_t (comp->record_press or
 * comp->signal_press); codec state lives on the device, so `comp` only names the method */
void *slow5_ptr_compress(struct __slow5_press *comp, const void *ptr, size_t count, size_t *n);
void *slow5_ptr_depress(struct __slow5_press *comp, const void *ptr, size_t count, size_t *n);

/* BLOW5: returns malloc'd [u64 size][press_record(payload)], *n = total length; NULL on error.
 * aux_meta == NULL drops the aux fields (lossy, src/merge.c:58-62). */
void *slow5_rec_to_mem(struct slow5_rec *read, struct slow5_aux_meta *aux_meta, enum slow5_fmt format,
                       struct slow5_press *compress, size_t *n);
/* *mem = record bytes without the size prefix (as slow5_get_next_mem returns them); may replace *mem with
 * the uncompressed record; allocates *read if NULL.  0 on success. */
int slow5_rec_depress_parse(char **mem, size_t *bytes, const char *read_id, struct slow5_rec **read,
                            struct slow5_file *s5p);
/* the same without the read id argument (/root/reference/src/skim.c:320); < 0 on error */
int slow5_decode(char **mem, size_t *bytes, struct slow5_rec **read, struct slow5_file *s5p);
/* slow5_rec_to_mem + fwrite (/root/reference/src/get.c:89, src/read_fast5.c:177): bytes written, -1 on error.  format ASCII
 * prints the record line (raw_signal column formatted on the GPU), aux_meta then names the aux column types. */
int slow5_rec_fwrite(FILE *fp, struct slow5_rec *read, struct slow5_aux_meta *aux_meta, enum slow5_fmt format,
                     struct slow5_press *compress);
struct slow5_rec *slow5_rec_init(void);
void slow5_rec_free(struct slow5_rec *read);

/* ---- BLOW5 file framing (layouts: Appendix A.1/A.2/A.4/A.5, test/misc/make_blow5.c:11-101) ----
 * Only what view / merge / get need around the press path: open + header, sequential record framing, header and
 * EOF writers, and the read_id index.  slow5_open tells BLOW5 from SLOW5 ASCII by the file's first bytes; for ASCII,
 * slow5_get_next_mem returns one record line (without its newline) and slow5_hdr_fwrite prints the two '#' version lines
 * followed by the same header text.  The index calls are BLOW5 only. */
slow5_file_t *slow5_open(const char *pathname, const char *mode);                 /* "r" only */
/* the same with the format the caller parsed from --from / the extension (/root/reference/src/view.c:192, src/degrade.c:423):
 * SLOW5_FORMAT_UNKNOWN = look at the file; a named format must match what the file's first bytes say */
slow5_file_t *slow5_open_with(const char *pathname, const char *mode, enum slow5_fmt format);
int slow5_close(slow5_file_t *s5p);
/* next record's bytes without the u64 size prefix, malloc'd; NULL + slow5_errno = SLOW5_ERR_EOF at the end
 * marker (src/view.c:265-278) */
void *slow5_get_next_mem(size_t *n, const slow5_file_t *s5p);
/* the same through out-parameters (/root/reference/src/skim.c:385): 0 and *mem / *bytes set, or slow5_errno (< 0) */
int slow5_get_next_bytes(char **mem, size_t *bytes, slow5_file_t *s5p);
/* 64-byte binary header + u32 length + header text; the version is raised to 0.2.0 when a signal press is set
 * (fixture exp_1_lossless_zlib_svb_v0.2.0.blow5 vs exp_1_lossless_zlib.blow5).  Returns bytes written or -1. */
int slow5_hdr_fwrite(FILE *fp, struct slow5_hdr *header, enum slow5_fmt format, slow5_press_method_t comp);
long slow5_eof_fwrite(FILE *fp);                                                  /* "5WOLB", src/view.c:313 */
int slow5_idx_create(slow5_file_t *s5p);   /* writes <pathname>.idx (slow5tools index, src/index.c) */
int slow5_idx_load(slow5_file_t *s5p);     /* loads <pathname>.idx, building it first if absent (src/get.c:286) */
void slow5_idx_unload(slow5_file_t *s5p);
/* where read_id's record sits in the file: offset of its u64 size prefix, size = 8 + record bytes (slow5lib's slow5_idx_get [RECALLED]); 0, or
 * -1 when the id is not in the index.  For loops that pread many records into one buffer (examples/s5get.c). */
struct slow5_rec_idx { uint64_t offset, size; };
int slow5_idx_get(struct slow5_idx *index, const char *read_id, struct slow5_rec_idx *read_index);
/* the ids of the index in file order (slow5_get_rids, /root/reference/src/skim.c): *n of them; the array and the strings stay the index's */
char **slow5_get_rids(const slow5_file_t *s5p, uint64_t *n);
/* raw record bytes of read_id (pread by index), malloc'd; the decode half of slow5_get goes through the batch hooks */
void *slow5_get_mem(const char *read_id, size_t *n, const slow5_file_t *s5p);
int slow5_get(const char *read_id, struct slow5_rec **read, slow5_file_t *s5p);   /* src/get.c:45 */

/* ---- batch hooks: one call per db_t batch instead of work_db(core, db, callback) ----
 * view / merge worker (src/view.c:35-57, src/merge.c:43-70): decode n input records, optionally rewrite
 * read_group (merge), re-encode with `to`.  mem[i] are freed like the reference's worker does; out[i] are
 * malloc'd buffers for the ordered fwrite loop (src/view.c:296-299). new_read_group may be NULL. */
int slow5_gpu_recompress_batch(int64_t n, char **mem, size_t *bytes, slow5_press_method_t from,
                               slow5_press_method_t to, const uint32_t *new_read_group, int drop_aux, void **out,
                               size_t *out_len);
/* The same worker when either side may be SLOW5 ASCII (src/view.c:35-57 with s5p->format / core->format_out ASCII):
 * mem[i] are record lines or BLOW5 records according to from_fmt; out[i] likewise for to_fmt (lines end in '\n').
 * aux_meta = the input header's (NULL: no aux columns).  Press methods are ignored on an ASCII side. */
int slow5_gpu_convert_batch(int64_t n, char **mem, size_t *bytes, enum slow5_fmt from_fmt, slow5_press_method_t from,
                            const struct slow5_aux_meta *aux_meta, enum slow5_fmt to_fmt, slow5_press_method_t to,
                            const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len);
/* get worker (src/get.c:37-66) after the pread: decode n records into slow5_rec_t's */
int slow5_gpu_depress_parse_batch(int64_t n, char **mem, size_t *bytes, slow5_press_method_t from,
                                  struct slow5_rec **reads);
/* encode n in-memory reads (f2s-style producers, src/read_fast5.c:176-181) */
int slow5_gpu_rec_to_mem_batch(int64_t n, struct slow5_rec **reads, int drop_aux, slow5_press_method_t to, void **out,
                               size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
