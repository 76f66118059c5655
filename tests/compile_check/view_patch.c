/* Compile-only check (tests/test_abi.py): the hunk INTEGRATION.md section 2 puts at /root/reference/src/view.c:292 in place of
 * work_db(&core, &db, depress_parse_rec_to_mem), type-checked next to slow5lib's names.  core_t / db_t restate the fields of
 * slow5tools' own /root/reference/src/thread.h:29-66 that the hunk touches.  Builds as C and as C++11 (the reference compiles
 * its .c files as C++, /root/reference/Makefile:9). */
#include <stdio.h>
#include <stdlib.h>
#include <slow5/slow5.h>        /* the stand-in of tests/compile_check/slow5/ (slow5lib is an absent submodule) */
#include <slow5gpu_hooks.h>     /* include/ of this repo */
SLOW5_GPU_HOOK_CHECK_ENUMS;     /* the build fails here if slow5lib's enum values are not the ones the hooks map to codecs */

typedef struct { int32_t num_thread; slow5_file_t *fp; slow5_fmt format_out; slow5_press_method_t press_method; int lossy; } core_t;
typedef struct { int len; void *buffer; } raw_record_t;
typedef struct { int64_t n_batch; int64_t n_err; raw_record_t *read_record; char **mem_records; size_t *mem_bytes; } db_t;
#define ERROR(fmt, ...) fprintf(stderr, "[%s::ERROR] " fmt "\n", __func__, __VA_ARGS__)

void view_batch_on_gpu(core_t *core_p, db_t *db_p) {
    core_t core = *core_p;
    db_t db = *db_p;
    /* ---- begin hunk ---- */
    {
        size_t *lens = (size_t *) malloc(db.n_batch * sizeof *lens);
        void  **bufs = (void **)  malloc(db.n_batch * sizeof *bufs);
        if (slow5_gpu_hook_recompress(db.n_batch, db.mem_records, db.mem_bytes,
                                      core.fp->compress->record_press->method, core.fp->compress->signal_press->method,
                                      core.press_method.record_method, core.press_method.signal_method,
                                      NULL /* keep read_group */, 0 /* keep aux */, bufs, lens) != 0) {
            ERROR("GPU press path failed: %s", slow5_gpu_hook_error());
            exit(EXIT_FAILURE);                     /* same fail-fast policy as src/view.c:39,46,52 */
        }
        for (int64_t i = 0; i < db.n_batch; i++) {  /* hand results to the unchanged write loop (src/view.c:296-299) */
            db.read_record[i].buffer = bufs[i];
            db.read_record[i].len    = (int) lens[i];
        }
        free(bufs); free(lens);
    }
    /* ---- end hunk ---- */
    /* ---- the ASCII-side variant of the same spot ---- */
    if (core.fp->format == SLOW5_FORMAT_ASCII || core.format_out == SLOW5_FORMAT_ASCII) {
        const char *types_line = "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n";
        size_t lens1[1]; void *bufs1[1];
        (void) slow5_gpu_hook_convert(0, db.mem_records, db.mem_bytes, core.fp->format, core.fp->compress->record_press->method,
                                      core.fp->compress->signal_press->method, types_line, core.format_out,
                                      core.press_method.record_method, core.press_method.signal_method, NULL, core.lossy, bufs1, lens1);
    }
}
