/*
 * s5oracle.h — CPU ORACLE for the BLOW5 per-read press path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm that slow5tools reaches through
 * slow5lib (an un-vendored submodule: /root/reference/.gitmodules:1-3, slow5lib/ is empty, so
 * the reference's own implementation cannot be compiled here).  Third-party code restated:
 *   - hasindu2008/slow5lib >= v1.3.0 (src/slow5.c, src/slow5_press.c) — record layout + press API
 *   - its thirdparty/streamvbyte (namespaced streamvbyte_slow5) — 32-bit StreamVByte + zigzag delta
 *   - system zlib 1.2.11 (used as-is through libz, not restated)
 * Call sites that define the contract: src/view.c:35-57, src/merge.c:43-70, src/get.c:37-66;
 * the uncompressed layout is stated literally by the reference in test/misc/make_blow5.c:11-101.
 *
 * PARITY STATUS: pinned.  tests/test_oracle_golden.py checks every function here against the
 * reference's own golden .blow5 fixtures (tests/golden/, copied from /root/reference/test/data,
 * see SURVEY.md Appendix B): every svb-zd blob re-encodes bit-for-bit, every zlib record
 * re-compresses byte-for-byte, and decoded signals equal the ASCII twin exp_1_lossless.slow5.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (slow5tools_amd/, include/) never links or calls it.
 */
#ifndef S5ORACLE_H
#define S5ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* on-disk method codes (names from src/misc.c:253-263) */
enum { S5O_REC_NONE = 0, S5O_REC_ZLIB = 1 };
enum { S5O_SIG_NONE = 0, S5O_SIG_SVB_ZD = 1, S5O_SIG_EX_ZD = 2 };

/* ---- a5 / a8: svb-zd signal codec (StreamVByte 32-bit, 2-bit keys, zigzag delta) ---- */
size_t s5o_svbzd_bound(uint64_t n);                       /* 4 + ceil(n/4) + 4n            */
size_t s5o_svbzd_encode(const int16_t *x, uint64_t n, uint8_t *out);      /* returns bytes */
/* returns 0 on success; *n_out = sample count; out must hold n samples (query with out=NULL) */
int s5o_svbzd_decode(const uint8_t *in, size_t in_len, int16_t *out, uint64_t *n_out);

/* ---- a6 / a9: zlib record codec (system libz; level 6, wbits 15, memLevel 8) ---- */
size_t s5o_zlib_bound(size_t n);
/* per-record deflateInit2/deflate/deflateEnd, as slow5_press_init per record (src/view.c:43-54) */
int s5o_zlib_compress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);
int s5o_zlib_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len /* in: cap */);
/* the same bytes from a per-thread deflate state that is reset, not re-allocated, per record: NOT the reference's shape; the
 * cpu_baseline's "pooled_zstream" point only (s5o_pool_zstream != 0 makes s5o_rec_to_mem use it) */
extern int s5o_pool_zstream;
int s5o_zlib_compress_pooled(const uint8_t *in, size_t n, uint8_t *out, size_t *out_len);
void s5o_zlib_pool_release(void);
uint32_t s5o_adler32(const uint8_t *p, size_t n);

/* ---- a1 / a4 / a7: the in-memory read and its BLOW5 record ---- */
typedef struct {
    uint16_t read_id_len;
    const char *read_id;
    uint32_t read_group;
    double digitisation, offset, range, sampling_rate;
    uint64_t len_raw_signal;          /* samples */
    const int16_t *raw_signal;
    const uint8_t *aux;               /* already-serialised aux fields (may be NULL) */
    size_t aux_len;
} s5o_rec_t;

size_t s5o_payload_bound(const s5o_rec_t *r, int sig_method);
/* packs the uncompressed record payload (Appendix A.3); returns its length */
size_t s5o_rec_pack(const s5o_rec_t *r, int sig_method, uint8_t *out);
size_t s5o_rec_to_mem_bound(const s5o_rec_t *r, int sig_method);
/* slow5_rec_to_mem (src/view.c:49): [u64 size][press_record(payload)]; returns total length, 0 on error.
 * scratch must hold s5o_payload_bound() bytes. */
size_t s5o_rec_to_mem(const s5o_rec_t *r, int rec_method, int sig_method, uint8_t *scratch, uint8_t *out);

/* parse an uncompressed payload: fills r (pointers into payload / into sig_out). sig_out must hold
 * the samples (call once with sig_out=NULL to learn len_raw_signal). returns 0 ok. */
int s5o_rec_parse(const uint8_t *payload, size_t len, int sig_method, s5o_rec_t *r, int16_t *sig_out);

/* ---- synthetic reads (SURVEY.md §8(d) generator, integer-only so CPU == GPU bit-for-bit) ---- */
void s5o_synth_read(uint64_t seed, uint64_t read_idx, uint64_t n, int16_t *out);
void s5o_synth_read_id(uint64_t read_idx, char out[37]);

/* ---- a12-a14: reference-shaped CPU batch encode (the cpu_baseline) ----
 * pthread static split + 1-item work stealing as src/thread.c:19-111; per record:
 * svb-zd encode -> pack -> deflateInit2/deflate/deflateEnd -> malloc'd buffer (src/view.c:35-57).
 * sig: n_reads signals of n_samples each, contiguous. Returns total output bytes; *secs = compute wall time. */
uint64_t s5o_encode_batch_mt(const int16_t *sig, uint64_t n_reads, uint64_t n_samples, uint64_t first_read_idx,
                             int rec_method, int sig_method, int n_threads, int batch_size, double *secs,
                             uint64_t *checksum);

/* the decode twin: compute phase of `get --benchmark -t T -K B` (src/get.c:52) — per id inflate (per-record inflateInit) +
 * parse + svb-zd decode into a malloc'd buffer; stream = BLOW5 records [u64 size][zlib], rec_off[i] their offsets.
 * Returns samples decoded (0 if any record failed); *secs = compute wall time. */
uint64_t s5o_decode_batch_mt(const uint8_t *stream, const uint64_t *rec_off, const uint32_t *ids, uint64_t n_ids,
                             int rec_method, int sig_method, int n_threads, int batch_size, double *secs,
                             uint64_t *checksum);

/* the same worker fed with SLOW5 text (BASELINE configs[0]: view in.slow5 -o out.blow5): per record parse of the ASCII line (ascii.c) +
 * svb-zd + zlib; line i = text[line_off[i], line_off[i + 1]).  Returns total output bytes (0 if a line failed). */
uint64_t s5o_convert_ascii_batch_mt(const char *text, const uint64_t *line_off, uint64_t n_lines, int rec_method, int sig_method,
                                    int n_threads, int batch_size, double *secs, uint64_t *checksum);

/* END-TO-END twins on files (bench.py `e2e`): the whole batch loop of `view` (serial read of K records, work_db, serial ordered write:
 * src/view.c:241-323) from a .slow5 or a zlib + svb-zd BLOW5 file to a zlib + svb-zd BLOW5 file; phases = {read, compute, write, first read
 * to last write} seconds.  Returns records written (0 on failure). */
uint64_t s5o_view_file(const char *in_path, const char *out_path, int n_threads, int batch_size, uint64_t max_reads, double phases[4]);
/* ... and of `get --benchmark` (src/get.c:52,321-386): per id pread + inflate + parse + svb-zd decode inside the worker, nothing written;
 * pos / len = file extents of the records ([u64 size][bytes]).  Returns samples decoded (0 on failure). */
uint64_t s5o_get_file(const char *path, const uint64_t *pos, const uint32_t *len, uint64_t n_ids, int n_threads, int batch_size, double *secs);

/* ---- §8f row 2: SLOW5 ASCII record lines <-> uncompressed payloads (ascii.c) ---- */
/* aux type codes: low 4 bits 0..11 = int8,int16,int32,int64,uint8,uint16,uint32,uint64,float,double,char,enum; 0x80 = array */
int s5o_aux_types(const char *types_line, size_t len, uint8_t *types, unsigned cap);
size_t s5o_double_to_text(double v, char *out);
size_t s5o_signal_to_text(const int16_t *sig, uint64_t n, char *out);                      /* out >= 7n bytes */
int64_t s5o_text_to_signal(const char *txt, size_t len, int16_t *out, uint64_t cap);       /* -1 on malformed text */
size_t s5o_ascii_line_to_payload(const char *line, size_t len, const uint8_t *types, unsigned n_aux, uint8_t *out);
size_t s5o_payload_to_ascii_line(const uint8_t *pay, size_t len, const uint8_t *types, unsigned n_aux, char *out);

/* ---- §8f row 4: ex-zd signal codec (exzd.c; layout pinned on the reference's ex-zd fixtures) ---- */
size_t s5o_exzd_bound(uint64_t n);
size_t s5o_exzd_encode(const int16_t *x, uint64_t n, uint8_t *out);                        /* returns bytes, 0 on allocation failure */
int s5o_exzd_decode(const uint8_t *in, size_t len, int16_t *out, uint64_t *n_out);          /* out = NULL: query n */

/* ---- §8f row 4: zstd record press — restated frame decoder (zstd_dec.c; pinned against libzstd itself) ---- */
size_t s5o_zstd_restated_decompress(const uint8_t *in, size_t len, uint8_t *out, size_t cap);   /* (size_t)-1 on error */
/* the frame layout of the device encoder (zstd_enc.c): literals-only blocks of <= 16 KiB */
int s5o_zstd_seq_ctable(int which, uint16_t *next, int32_t *dnb, int32_t *dfs);
int s5o_zstd_seq_dtable(int which, uint32_t *cells);   /* decode cells of a predefined distribution, packed as csrc/zstd_seq_tables.h has them */
size_t s5o_zstd_literals_bound(size_t n);
size_t s5o_zstd_literals_compress(const uint8_t *in, size_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
