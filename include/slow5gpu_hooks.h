/*
 * slow5gpu_hooks.h — the batch hooks of the MI355X press path, in a header that can sit NEXT TO slow5lib's own
 * <slow5/slow5.h>: it declares no slow5lib name (no struct slow5_rec, no enum slow5_press_method, no slow5_press_method_t),
 * only plain integers, pointers and one struct of its own.  This is what a patched slow5tools includes; slow5_compat.h
 * (which restates slow5lib's types because the submodule is absent here) is for programs built WITHOUT slow5lib.
 *
 * A hook stands where work_db(core, db, callback) stands (/root/reference/src/thread.c:114):
 *     /root/reference/src/view.c:292    work_db(&core, &db, depress_parse_rec_to_mem)
 *     /root/reference/src/merge.c:440   work_db(&core, &db, parallel_reads_model)
 *     /root/reference/src/get.c:364     work_db(&core, &db, work_per_single_read_get)
 *     /root/reference/src/split.c:506   (read-group split: the same decode half)
 * and takes the whole db_t batch.  Press methods and formats travel as ints holding slow5lib's OWN enum values —
 * enum slow5_press_method: NONE 0, ZLIB 1, SVB_ZD 2, ZSTD 3, EX_ZD 4 (names at /root/reference/src/misc.c:253-263);
 * enum slow5_fmt: ASCII 1, BINARY 2 — so the call site passes core.press_method.record_method etc. unchanged.
 * Every hook returns 0 on success, -1 on failure (slow5_gpu_hook_error() says why); out[i] are malloc'd buffers the
 * ordered fwrite loop frees (/root/reference/src/view.c:296-299).  INTEGRATION.md shows the patch.
 */
#ifndef SLOW5GPU_HOOKS_H
#define SLOW5GPU_HOOKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The integer values above are what this library maps to codecs.  They are restated from memory of slow5lib >= 1.3.0 (the submodule
 * is absent from the reference snapshot this was built against), so a patched source file puts
 *     SLOW5_GPU_HOOK_CHECK_ENUMS;
 * once at file scope behind #include <slow5/slow5.h>: if upstream's enums ever differ, the build fails there instead of the
 * patch silently mis-mapping codecs.  (A macro: this header itself names no slow5lib identifier.) */
#ifdef __cplusplus
#define SLOW5_GPU_HOOK_STATIC_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define SLOW5_GPU_HOOK_STATIC_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
#define SLOW5_GPU_HOOK_CHECK_ENUMS                                                                                                   \
    SLOW5_GPU_HOOK_STATIC_ASSERT((int)SLOW5_COMPRESS_NONE == 0 && (int)SLOW5_COMPRESS_ZLIB == 1 && (int)SLOW5_COMPRESS_SVB_ZD == 2 &&      \
                                 (int)SLOW5_COMPRESS_ZSTD == 3 && (int)SLOW5_COMPRESS_EX_ZD == 4 && (int)SLOW5_FORMAT_ASCII == 1 &&      \
                                 (int)SLOW5_FORMAT_BINARY == 2,                                                                       \
                                 "slow5lib's enum values differ from the ones slow5gpu_hooks.h documents: the hooks would mis-map codecs")

/* Optional: name the GPUs (bit d = HIP device d); without it the first hook call takes device 0.  A batch is cut into one
 * contiguous index range per device, as work_db cuts it per thread (/root/reference/src/thread.c:76-90). */
int slow5_gpu_hook_init(uint64_t dev_mask);
void slow5_gpu_hook_shutdown(void);
const char *slow5_gpu_hook_error(void);

/* view / merge worker (src/view.c:35-57, src/merge.c:43-70) for BLOW5 -> BLOW5: decode the n records mem[i] (bytes as
 * slow5_get_next_mem returns them), optionally rewrite read_group (merge.c:51; NULL keeps it) and drop the aux fields
 * (lossy, merge.c:58-62), re-encode.  mem[i] are freed and set NULL like the reference's worker does (view.c:41). */
int slow5_gpu_hook_recompress(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method,
                              int to_record_method, int to_signal_method, const uint32_t *new_read_group, int drop_aux,
                              void **out, size_t *out_len);

/* The same worker with ONE release per batch instead of one free per record (round 5): out[i] point into pinned buffers the library
 * owns (no malloc, no memcpy, no page fault per record — what holds the malloc form to 5 GB/s at a million records per call), and
 * the ordered write loop ends with slow5_gpu_hook_release(batch) where it had free(db.read_record[i].buffer) per record
 * (/root/reference/src/view.c:296-299; INTEGRATION.md shows the three changed lines).  *batch is NULL on failure. */
int slow5_gpu_hook_recompress_arena(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method,
                                    int to_record_method, int to_signal_method, const uint32_t *new_read_group, int drop_aux,
                                    void **out, size_t *out_len, void **batch);
void slow5_gpu_hook_release(void *batch);

/* ... and with the batch IN FLIGHT while the loop reads the next one (round 6): _submit returns a ticket at once (NULL: failure), the work
 * runs on a thread of the library's; _wait(ticket, &batch) blocks until that batch is done and returns what the synchronous hook would
 * have (0 / -1; *batch = the arena to release after the write loop).  mem[i] are freed and set NULL by the time _wait returns; mem, bytes,
 * out, out_len (and new_read_group) must stay valid and untouched until then — use two sets of them and alternate.  Two tickets in
 * flight keep both of the library's contexts busy: batch k + 1 uploads and decodes under batch k's encode and download.
 * INTEGRATION.md section 2 shows the dozen lines /root/reference/src/view.c:254-300 needs. */
void *slow5_gpu_hook_recompress_submit(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method,
                                       int to_record_method, int to_signal_method, const uint32_t *new_read_group, int drop_aux,
                                       void **out, size_t *out_len);
int slow5_gpu_hook_recompress_wait(void *ticket, void **batch);

/* The same worker on a CHUNK of a BLOW5 file, for a loop that reads the file in large pieces instead of one record at a time
 * (examples/s5view.c; 17 GB/s of raw signal end to end against 2.9 with one fread + malloc per record): the n records sit framed
 * — [u64 size][bytes] — in `chunk` exactly as read from disk, rec_pos[i] / rec_len[i] = offset and length of record i's bytes
 * behind its size prefix; the re-encoded records come back as ONE contiguous stream in out_buf, the bytes the ordered write loop
 * emits (out_off[i] = offset of record i, out_off[n] = total).  Buffers from slow5_gpu_hook_alloc (pinned) move at PCIe speed.
 * -1 with out_off[0] > out_cap: the output needs that much room. */
void *slow5_gpu_hook_alloc(size_t bytes);
void slow5_gpu_hook_free(void *p);
int slow5_gpu_hook_recompress_chunk(int64_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len,
                                    int from_record_method, int from_signal_method, int to_record_method, int to_signal_method,
                                    const uint32_t *new_read_group, int drop_aux, void *out_buf, size_t out_cap, uint64_t *out_off);

/* The same worker when either side is SLOW5 ASCII.  aux_types_line: the header's column-types line ("#char*\tuint32_t\t...",
 * with or without the newline; NULL or no aux columns: none).  ASCII output lines end in '\n'. */
int slow5_gpu_hook_convert(int64_t n, char **mem, size_t *bytes, int from_fmt, int from_record_method, int from_signal_method,
                           const char *aux_types_line, int to_fmt, int to_record_method, int to_signal_method,
                           const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len);

/* One decoded read, for callers that fill their own slow5_rec_t (get --benchmark src/get.c:52, skim src/skim.c:320, split).
 * read_id and aux point INTO the uncompressed record that replaced mem[i] (the caller frees mem[i], as after
 * slow5_rec_depress_parse, src/view.c:41); raw_signal is a malloc'd buffer the caller owns. */
typedef struct slow5_gpu_read {
    const char *read_id;          /* not NUL-terminated */
    uint16_t read_id_len;
    uint32_t read_group;
    double digitisation, offset, range, sampling_rate;
    uint64_t len_raw_signal;
    int16_t *raw_signal;
    const uint8_t *aux;           /* serialised aux fields in header order (BLOW5 layout), slow5lib's aux parser reads them */
    uint64_t aux_len;
} slow5_gpu_read_t;
int slow5_gpu_hook_depress_parse(int64_t n, char **mem, size_t *bytes, int from_record_method, int from_signal_method,
                                 slow5_gpu_read_t *reads);
/* encode n in-memory reads (f2s-style producers, src/read_fast5.c:176-181): out[i] = [u64 size][record] */
int slow5_gpu_hook_rec_to_mem(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method,
                              void **out, size_t *out_len);
/* ... into an arena (slow5_gpu_hook_release(batch) instead of n free()s) */
int slow5_gpu_hook_rec_to_mem_arena(int64_t n, const slow5_gpu_read_t *reads, int drop_aux, int to_record_method, int to_signal_method,
                                    void **out, size_t *out_len, void **batch);

#ifdef __cplusplus
}
#endif
#endif
