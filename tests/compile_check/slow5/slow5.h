/*
 * TEST-ONLY stand-in for slow5lib's <slow5/slow5.h>, written from the call-site contract in SURVEY.md 8(b) — slow5lib is an
 * absent submodule of the reference, so its real header is not available here.  It exists for ONE purpose: to prove that
 * include/slow5gpu_hooks.h can be included next to a header that defines slow5lib's names (struct slow5_rec, enum
 * slow5_press_method, slow5_press_method_t, struct slow5_file, ...) without a clash, and that the INTEGRATION.md hunk
 * type-checks against those names.  Nothing is built from it, nothing in the product or the oracle includes it.
 */
#ifndef MOCK_SLOW5_H
#define MOCK_SLOW5_H
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
enum slow5_press_method { SLOW5_COMPRESS_NONE, SLOW5_COMPRESS_ZLIB, SLOW5_COMPRESS_SVB_ZD, SLOW5_COMPRESS_ZSTD, SLOW5_COMPRESS_EX_ZD };
typedef struct { enum slow5_press_method record_method; enum slow5_press_method signal_method; } slow5_press_method_t;
struct __slow5_press { enum slow5_press_method method; void *stream; };
struct slow5_press { struct __slow5_press *record_press; struct __slow5_press *signal_press; };
typedef struct slow5_press slow5_press_t;
enum slow5_fmt { SLOW5_FORMAT_UNKNOWN, SLOW5_FORMAT_ASCII, SLOW5_FORMAT_BINARY };
typedef enum slow5_fmt slow5_fmt;
struct slow5_aux_meta { uint32_t num; char **attrs; void *types; };
typedef struct slow5_aux_meta slow5_aux_meta_t;
struct slow5_rec { uint16_t read_id_len; char *read_id; uint32_t read_group; double digitisation, offset, range, sampling_rate;
                   uint64_t len_raw_signal; int16_t *raw_signal; void *aux_map; };
typedef struct slow5_rec slow5_rec_t;
struct slow5_hdr { struct { uint8_t major, minor, patch; } version; uint32_t num_read_groups; slow5_aux_meta_t *aux_meta; };
struct slow5_file { FILE *fp; enum slow5_fmt format; slow5_press_t *compress; struct slow5_hdr *header; };
typedef struct slow5_file slow5_file_t;
extern int slow5_errno;
void *slow5_get_next_mem(size_t *n, const slow5_file_t *s5p);
#ifdef __cplusplus
}
#endif
#endif
