/*
 * slow5gpu.h — C ABI of the MI355X-native BLOW5 record press path (libslow5gpu.so).
 *
 * Plain C, plain pointers and sizes.  This is the drop-in boundary for the one data-parallel hot
 * path of slow5tools: the per-record worker that `work_db` fans out over pthreads
 *     /root/reference/src/view.c:35-57   depress_parse_rec_to_mem
 *     /root/reference/src/merge.c:43-70  parallel_reads_model
 *     /root/reference/src/get.c:37-66    work_per_single_read_get
 * i.e. slow5_rec_to_mem() = svb-zd(raw_signal) -> pack -> zlib, and its inverse
 * slow5_rec_depress_parse().  A whole batch (db_t, /root/reference/src/thread.h:50-66) is handed
 * over in one call; results come back in the same per-record slots so the ordered fwrite loops
 * (/root/reference/src/view.c:296-299) stay untouched.  See INTEGRATION.md for the patch.
 *
 * Two levels:
 *   s5gpu_*_dev   : device-resident buffers + a HIP stream (what bench.py times; kernels only)
 *   s5gpu_*_batch : host buffers in, malloc'd host buffers out (what slow5tools would call)
 * The slow5lib-compatible per-record API (slow5_press_*, slow5_rec_to_mem, ...) is declared in
 * slow5_compat.h and implemented on top of the batch calls.
 */
#ifndef SLOW5GPU_H
#define SLOW5GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* on-disk method codes, identical to slow5lib's enum slow5_press_method values used by
 * /root/reference/src/misc.c:253-263 (SLOW5_COMPRESS_NONE/ZLIB, SLOW5_COMPRESS_NONE/SVB_ZD) */
enum { S5GPU_REC_NONE = 0, S5GPU_REC_ZLIB = 1, S5GPU_REC_ZSTD = 2 };   /* zstd: any frame libzstd writes is decoded (an optional content
                                                                         * checksum is verified: status 4 on a mismatch); encoded frames are literals-only
                                                                         *  */
enum { S5GPU_SIG_NONE = 0, S5GPU_SIG_SVB_ZD = 1, S5GPU_SIG_EX_ZD = 2 };

enum {
    S5GPU_OK = 0,
    S5GPU_ERR_ARG = -1,       /* bad argument / unsupported method        */
    S5GPU_ERR_HIP = -2,       /* HIP runtime error (s5gpu_last_error())   */
    S5GPU_ERR_NOMEM = -3,
    S5GPU_ERR_NODEV = -4,     /* no gfx950 device: the library never falls back to a CPU path */
    S5GPU_ERR_DATA = -5       /* corrupt record (per-record status says which) */
};

/* One read of an encode batch.  All offsets index the batch-wide device buffers. */
typedef struct s5gpu_read_desc {
    uint64_t sig_off;    /* first sample of the read in `sig` (sample index, multiple of 8)          */
    uint64_t hdr_off;    /* byte offset in `hdr` of the record bytes that precede the u64 length:    */
                         /*   u16 read_id_len | read_id | u32 read_group | 4 x f64                   */
    uint64_t aux_off;    /* byte offset in `aux` of the already-serialised aux fields                */
    uint64_t out_off;    /* byte offset of this read's slot in `slots` (multiple of 16)              */
    uint32_t n_samples;  /* len_raw_signal                                                          */
    uint32_t hdr_len;    /* 2 + read_id_len + 4 + 32                                                 */
    uint32_t aux_len;
    uint32_t slot_cap;   /* bytes available at out_off, >= s5gpu_slot_bound()                        */
} s5gpu_read_desc_t;

/* Worst-case bytes one encoded record can occupy in its slot (incl. the u64 size prefix). */
uint64_t s5gpu_slot_bound(uint32_t n_samples, uint32_t hdr_len, uint32_t aux_len, int rec_method, int sig_method);
/* Uncompressed payload upper bound (what `max_payload` below must cover). */
uint64_t s5gpu_payload_bound(uint32_t n_samples, uint32_t hdr_len, uint32_t aux_len, int sig_method);

typedef struct s5gpu_encode_args {
    uint32_t n_reads;
    int32_t rec_method, sig_method;
    const s5gpu_read_desc_t *desc;   /* device, n_reads entries                                      */
    const int16_t *sig;              /* device, raw_signal of all reads                              */
    const uint8_t *hdr;              /* device                                                       */
    const uint8_t *aux;              /* device (may be NULL when every aux_len == 0)                 */
    uint8_t *slots;                  /* device, out: [u64 size][record bytes] per read at out_off    */
    uint32_t *out_len;               /* device, out: bytes written per read, incl. the 8-byte prefix */
    uint32_t max_payload;            /* max over reads of s5gpu_payload_bound()                      */
    uint32_t lds_payload_cap;        /* 0 = auto.  LDS bytes the one-workgroup-per-read kernel keeps */
                                     /*   for a payload; reads that need more (long or incompressible*/
                                     /*   signals) are re-run through the HBM-staged kernels.  With 0 and */
                                     /*   a longest read far over the budget the whole batch is staged;   */
                                     /*   name a budget (8192 is a good one) for batches that mix short   */
                                     /*   and long reads: the host batch calls do so from the lengths     */
    uint32_t *ovf;                   /* device, n_reads + 1 words of scratch (list of such reads)    */
} s5gpu_encode_args_t;

/* One record of a decode batch. */
typedef struct s5gpu_rec_desc {
    uint64_t in_off;     /* byte offset in `in` of the record bytes (without the u64 size prefix)    */
    uint64_t pay_off;    /* byte offset of this record's payload slot in `payload` (multiple of 16)  */
    uint64_t sig_off;    /* sample offset of this record's output in `sig_out` (multiple of 8)       */
    uint32_t in_len;
    uint32_t pay_cap;    /* bytes available at pay_off                                               */
    uint32_t sig_cap;    /* samples available at sig_off                                             */
    uint32_t reserved;
} s5gpu_rec_desc_t;

/* Parsed primary fields of one decoded record (slow5_rec_t minus the pointers). */
typedef struct s5gpu_rec_fields {
    int32_t status;        /* 0 ok; 1 bad zlib header; 2 corrupt data; 3 truncated; 4 adler mismatch;  */
                           /* 5 payload slot too small (payload_len = needed); 6 signal slot too small */
                           /* (n_samples = needed); 7 malformed record                                 */
    uint32_t payload_len;  /* uncompressed record length                                               */
    uint32_t n_samples;
    uint32_t read_id_len;  /* read_id sits at payload + 2                                              */
    uint32_t read_group;
    uint32_t aux_off;      /* offset of the aux bytes within the payload                               */
    uint32_t aux_len;
    uint32_t reserved;
    double digitisation, offset, range, sampling_rate;
} s5gpu_rec_fields_t;

/* s5gpu_decode_args.flags */
enum {
    /* s5gpu_decode_dev, zlib or zstd records with svb-zd signals and zlib records with ex-zd signals: the caller wants fields + signals only (what `get` and the
     * decode half of a signal consumer need; read_id / aux bytes live in the uncompressed record and are NOT kept).  `payload` is then
     * SCRATCH of payload_bytes bytes (s5gpu_decode_scratch_bytes() says how much is useful): the kernel runs as persistent
     * workgroups, each with one scratch slot of max_pay_cap bytes that it reuses record after record, so an uncompressed record
     * needs no slot of its own (n x pay_cap bytes saved; the HBM traffic is that of the full form: DESIGN.md 4.8).  desc[i].pay_off / pay_cap are ignored; a record whose payload
     * exceeds max_pay_cap reports status 5 with the size needed.  fields[i].aux_off / aux_len are still reported. */
    S5GPU_DEC_NO_PAYLOAD = 1
};

typedef struct s5gpu_decode_args {
    uint32_t n_recs;
    int32_t rec_method, sig_method;
    uint32_t flags;                  /* 0, or S5GPU_DEC_* (this word was padding before: zeroed structs keep their meaning)  */
    const s5gpu_rec_desc_t *desc;    /* device                                                       */
    const uint8_t *in;               /* device, compressed records; 16 readable bytes must follow the */
                                     /*   last record (the inflate kernels fetch aligned dwords ahead) */
    uint8_t *payload;                /* device, out: uncompressed record per slot (scratch with S5GPU_DEC_NO_PAYLOAD) */
    int16_t *sig_out;                /* device, out: raw_signal per record                           */
    s5gpu_rec_fields_t *fields;      /* device, out                                                  */
    uint64_t payload_bytes;          /* S5GPU_DEC_NO_PAYLOAD: bytes of scratch at `payload`          */
    uint32_t max_pay_cap;            /* largest uncompressed record to expect (0 = not known).  S5GPU_DEC_NO_PAYLOAD: the scratch slot
                                      * size.  Any form: a batch of short records (<= 32 KiB of slot, i.e. reads of up to ~10 k samples) with
                                      * svb-zd / ex-zd signals runs the inflate kernel in its 24-waves-per-CU shape (a 256-entry list
                                      * of waiting matches); longer or unknown ones in the 21-wave shape with 768 entries, which long
                                      * reads written by stock zlib need in their key bytes */
    uint32_t max_in_len;             /* largest COMPRESSED record of the batch, bytes (0 = not known; this word was `reserved`: zeroed structs keep
                                      * their meaning).  S5GPU_DEC_NO_PAYLOAD on zlib + svb-zd records: when every record fits one 4 KiB window
                                      * of the inflate (max_in_len <= 4000: reads of up to ~4500 samples) the uncompressed record is kept in
                                      * LDS and never reaches HBM (round 6: k_inflate_par_np_lp; a record whose payload outgrows the 5.3 KiB
                                      * of LDS after all is redone by the slot decoder).  A hint: any value is safe */
} s5gpu_decode_args_t;
/* scratch worth bringing for S5GPU_DEC_NO_PAYLOAD (one slot per workgroup the device can hold + the fallback decoder's);
 * anything from 64 + 2 * (max_pay_cap + 32) bytes up works, less only means fewer workgroups */
uint64_t s5gpu_decode_scratch_bytes(uint32_t max_pay_cap);

/* ---- lifetime ---- */
int s5gpu_init(int device);              /* select device; S5GPU_ERR_NODEV if it is not a gfx950 GPU */
/* Several GPUs of one node behind the host-buffer batch calls: bit d of dev_mask = HIP device d.  A host batch is
 * then cut into one contiguous index range per device — device g of G takes records [g*n/G, (g+1)*n/G), exactly how work_db
 * cuts a batch per thread (/root/reference/src/thread.c:76-90) — each range goes through its own pinned H2D -> kernels -> D2H on
 * its device's stream, concurrently, and every result lands in the caller's out[i]: the ordered write loop
 * (/root/reference/src/view.c:296-299) does not change.  No collective, no peer traffic.  The *_dev entry points are not
 * affected: their caller picks the device (hipSetDevice) and passes buffers of that device. */
int s5gpu_init_mask(uint64_t dev_mask);
int s5gpu_devices_in_use(void);          /* devices the batch calls run on (0 before initialisation) */
/* Brings up, on the first device in use, what the first batch call would otherwise pay for: the HIP runtime and context, the kernels'
 * code objects (loaded on first launch) and one set of streams.  A tool calls it from a helper thread at start-up, under its own file
 * opening / index loading (examples/s5view.c, s5get.c: the first GPU call of a 100 k-id `get` was 175 ms of a 230 ms job).  Optional.
 * If the library is not initialised yet it initialises it as s5gpu_init(0) does — a caller that wants other devices calls
 * s5gpu_init_mask FIRST (afterwards it reports "already initialised"). */
int s5gpu_warmup(void);
void s5gpu_shutdown(void);
const char *s5gpu_last_error(void);
int s5gpu_device_count(void);
/* tuning knobs.  "inflate_par" (0/1, default 1): zlib records are inflated by the decoder that is parallel inside a record (one record
 * per wave, 64 self-synchronising segment decoders; whatever it declines is redone by the wave-per-record decoder).  With 0 the two
 * older kernels are used, chosen by batch size:  "inflate_simt_min": batches with at least this many zlib records use the lane-per-record
 * inflate kernel (throughput), smaller ones the wave-per-record kernel (latency); default 24576.
 * "inflate_route" (0/1, default 1): such batches are first counting-sorted by compressed length on the device, and records
 * of >= 32 KiB go to the wave-per-record kernel beside the lane kernel (real runs have read lengths spread over two decades).
 * In inflate-only calls fields[i].reserved, fields[0..128].read_group and fields[128].aux_len are then left holding routing
 * scratch (s5gpu_decode_dev overwrites all of them with the parsed fields).
 * "multi_min_per_device" (default 1024): a host batch of fewer than this many records per device stays on the first device.
 * "unpack_fused" (0/1, default 1): s5gpu_decode_dev on zlib / zstd + svb-zd records lets the wave that decompressed a record parse it and decode
 * its signal as well (fields.reserved is scratch on the way and 0 at the end); 0 = always the separate unpack kernel.
 * "zstd_sequences" (0/1, default 1): the zstd encoder sends runs of >= 5 equal bytes as one literal + one match at the repeat
 * offset (predefined FSE tables); 0 = literals-only frames cut at the record's seams (round 1; ~2 % larger records).
 * "fused_tier2" (bytes, 0 .. 16384, default 0 = off): batches of mixed lengths (s5gpu_encode_args.lds_payload_cap named) get a SECOND
 * one-workgroup-per-read launch with this LDS budget for the reads between the named budget and one 16 KiB DEFLATE block, before
 * the HBM-staged kernels take the rest; measured on real-run read lengths it gains nothing (profiles/r04_mixed_tier2.txt).
 * "order_min" (default 8192; 0 = never): batches of at least this many zlib / zstd records are DECODED longest record first (a counting sort
 * by compressed length on the device builds the launch order), and the overflow list of a mixed ENCODE batch (the reads the staged kernels
 * redo) is taken longest read first.  Costs 4 bytes per record of scratch per (device, stream), kept until s5gpu_shutdown.
 * "zstd_pre_min" (default 256; 0 = never): zstd batches of at least this many frames run a first pass that reads every frame's first tree
 * description, a frame per lane, in front of the decoder.  Costs 144 bytes per record in the same per-(device, stream) scratch
 * (144 MB for a million frames); a process that decodes on many short-lived streams should reuse streams or set this to 0. */
int s5gpu_set_option(const char *key, long value);

/* ---- device-resident entry points (asynchronous on `hip_stream`, a hipStream_t; NULL = default) ---- */
int s5gpu_encode_dev(const s5gpu_encode_args_t *args, void *hip_stream);
int s5gpu_decode_dev(const s5gpu_decode_args_t *args, void *hip_stream);
/* svb-zd only (BASELINE config 2): blob per read written at slots+out_off, out_len = blob bytes */
int s5gpu_svbzd_encode_dev(const s5gpu_encode_args_t *args, void *hip_stream);
/* Encode with ORDERED SINGLE-PASS OUTPUT (zlib record press): records go straight into the contiguous BLOW5 record
 * stream, rec_off[i] / rec_off[n] as s5gpu_compact_dev would produce them; no slots, no second pass.  args->slots and
 * args->ovf are not used.  state: n_reads u64 of scratch; ctl: 4 u32 of scratch, read them back after completion:
 * ctl[0] != 0 (some read did not fit the LDS budget) or ctl[2] != 0 (look-back timed out) => the stream is invalid,
 * fall back to s5gpu_encode_dev + s5gpu_compact_dev.  stream_out must hold the sum of the slot bounds. */
int s5gpu_encode_stream_dev(const s5gpu_encode_args_t *args, uint8_t *stream_out, uint64_t *rec_off, uint64_t *state,
                            uint32_t *ctl, void *hip_stream);
/* The same for the svb-zd stage alone (s5gpu_svbzd_encode_dev + s5gpu_compact_dev in one pass): blobs straight into the contiguous
 * blob stream; ctl as above (ctl[0]: a blob did not fit the LDS budget).  args->hdr / aux / slots / ovf are not used. */
int s5gpu_svbzd_encode_stream_dev(const s5gpu_encode_args_t *args, uint8_t *stream_out, uint64_t *rec_off, uint64_t *state,
                                  uint32_t *ctl, void *hip_stream);
/* single stages, for the solo press calls (slow5_ptr_compress_solo / slow5_ptr_depress_solo):
 *  deflate_parked: zlib-compress byte ranges already parked in their slots — read i's bytes sit at
 *    slots + out_off + ((slot_cap - s5gpu_payload_bound(desc i)) & ~15), out_len[i] = their length on entry
 *    and the framed length on return; works in place (see DESIGN.md "staged path").
 *  inflate: zlib streams -> payload slots, fields[i].status / payload_len (Adler-32 verified)
 *  svbzd_decode: svb-zd blobs (desc.in_off/in_len) -> sig_out, fields[i].status / n_samples */
int s5gpu_deflate_parked_dev(const s5gpu_encode_args_t *args, void *hip_stream);   /* rec_method zstd: the zstd twin */
/* pack_parked: the step in front of it for whole reads — signal press + record layout, every payload parked where deflate_parked
 *    expects it, out_len[i] = its length.  pack_parked + deflate_parked = s5gpu_encode_dev on a batch of long reads, as two calls a
 *    caller may put on different streams (bench.py's configs[3] leg: the next chunk's pack and the previous chunk's compaction run
 *    beside a chunk's deflate). */
int s5gpu_pack_parked_dev(const s5gpu_encode_args_t *args, void *hip_stream);
int s5gpu_inflate_dev(const s5gpu_decode_args_t *args, void *hip_stream);
int s5gpu_svbzd_decode_dev(const s5gpu_decode_args_t *args, void *hip_stream);
/* inflate_head: only the first desc[i].pay_cap bytes of every zlib record (a record's head: u16 read_id_len | read_id | ...; what
 * slow5_idx_create needs of a record), decoding stops there: fields[i].payload_len = bytes written, no Adler-32; desc[i].in_len may
 * cover just the front of the record (status 3 if it ends before pay_cap bytes are out). */
int s5gpu_inflate_head_dev(const s5gpu_decode_args_t *args, void *hip_stream);
/* Gather the slots into one contiguous BLOW5 record stream (what the ordered fwrite loop emits):
 * rec_off[i] = byte offset of record i in `stream`, rec_off[n] = total bytes.  tmp: >= 8*(n/1024+2) bytes. */
int s5gpu_compact_dev(uint32_t n_reads, const s5gpu_read_desc_t *desc, const uint8_t *slots, const uint32_t *out_len,
                      uint64_t *rec_off, uint8_t *stream, uint64_t *tmp, void *hip_stream);
/* write val[i] (little-endian u32) at base + off[i]: the read_group rewrite of merge (src/merge.c:51) on device */
int s5gpu_patch_u32_dev(uint8_t *base, const uint64_t *off, const uint32_t *val, uint32_t n, void *hip_stream);
/* synthetic reads on device (bench/test workload; bit-identical to oracle/synth.c) */
int s5gpu_synth_dev(int16_t *sig, uint64_t n_reads, uint64_t n_samples, uint64_t stride_samples, uint64_t seed,
                    uint64_t first_read_idx, void *hip_stream);
/* hdr bytes for synthetic reads: 74 bytes per read (36-char id, read_group 0, 8192/23/1467.61/4000) */
int s5gpu_synth_hdr_dev(uint8_t *hdr, uint64_t n_reads, uint64_t first_read_idx, void *hip_stream);

/* timing hook: runs fn-equivalent `iters` times between two hipEvents on the stream, returns ms (bench.py) */
int s5gpu_event_create(void **ev);
int s5gpu_event_record(void *ev, void *hip_stream);
int s5gpu_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);   /* synchronises on ev_stop */
int s5gpu_event_destroy(void *ev);

/* ---- host-buffer batch entry points: the work_db replacement ---- */
/* Encode n reads.  hdr[i]/aux[i] as in s5gpu_read_desc_t.  out[i] receives a malloc'd buffer the caller
 * frees (ownership as slow5_rec_to_mem, /root/reference/src/view.c:49,298); out_len[i] its length. */
int s5gpu_encode_batch(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                       const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                       int sig_method, void **out, size_t *out_len);
/* Decode n records (bytes without the u64 prefix).  payload[i] and sig[i] receive malloc'd buffers. */
int s5gpu_decode_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int rec_method, int sig_method,
                       void **payload, int16_t **sig, s5gpu_rec_fields_t *fields);

/* The whole view / merge worker for a batch of BLOW5 records (src/view.c:35-57, src/merge.c:43-70): decode with
 * (from_rec, from_sig), optionally rewrite read_group and drop the aux fields, re-encode with (to_rec, to_sig).
 * Only compressed bytes cross PCIe: decoded signals and payloads stay in HBM between the two halves.
 * out[i] malloc'd [u64 size][record]; a corrupt input record fails the call (status[i], may be NULL, says which). */
int s5gpu_recompress_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                           int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                           int32_t *status);

/* The ARENA form of the two calls above (round 5).  One malloc per record is what slow5_rec_to_mem promises (src/view.c:49,298), and at
 * a million records per call it is what bounds the call: a million fresh buffers the process has never touched, page-faulted in one by
 * one.  Here out[i] point INTO a few pinned buffers the device-to-host copies landed in — no copy, no allocation per record — and the
 * caller gives the whole batch back with ONE s5gpu_arena_release(*arena) once its ordered write loop is through (the free() of
 * src/view.c:298 goes).  The buffers return to a process-wide pool (up to 6 GB kept; drained by s5gpu_shutdown), so a loop of
 * similar batches neither allocates nor faults after its first ones.  *arena is NULL when the call fails.  Never free() an out[i]. */
int s5gpu_encode_batch_arena(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                             const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                             int sig_method, void **out, size_t *out_len, void **arena);
int s5gpu_recompress_batch_arena(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                 int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                 int32_t *status, void **arena);
void s5gpu_arena_release(void *arena);   /* NULL is fine */

/* The two calls as SUBMIT / WAIT pairs (round 6): the reference's loop reads K records, works on them, writes them, and overlaps nothing
 * (/root/reference/src/view.c:254-300; /root/reference/README.md:197 names the overlap as future work).  A submitted batch runs on a
 * thread of the library's and takes one of its contexts (S5GPU_CONTEXTS, default 2): with two tickets in flight one batch's PCIe copies
 * run under the other's kernels while the caller reads the next batch.  want_arena != 0: out[i] point into an arena that
 * s5gpu_batch_wait hands over (release it with s5gpu_arena_release); 0: one malloc per record.  Every array named here belongs to the
 * library until the ticket is waited for; every ticket is waited for exactly once.  NULL ticket: s5gpu_last_error() says why.
 * s5gpu_batch_wait returns the batch call's own code, its message in the waiting thread's s5gpu_last_error(). */
void *s5gpu_recompress_batch_submit(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                    int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                    int32_t *status, int want_arena);
void *s5gpu_encode_batch_submit(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                                const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                                int sig_method, void **out, size_t *out_len, int want_arena);
int s5gpu_batch_wait(void *ticket, void **arena);

/* The same worker on a CHUNK of a BLOW5 file (what bounds `view` end to end is the read and write phases around
 * work_db, /root/reference/src/view.c:265-278,296-299, not the compute).  The n records sit framed — [u64 size][bytes] — in one
 * host buffer `chunk` exactly as read from disk: rec_pos[i] = offset of record i's bytes (behind its size prefix), rec_len[i]
 * their length.  The re-encoded records come back as ONE contiguous stream in out_buf, exactly the bytes the ordered write loop
 * emits: out_off[i] = offset of record i (its u64 prefix), out_off[n] = total.  No per-record malloc or memcpy on either side.
 * chunk / out_buf from s5gpu_host_alloc (pinned) move at PCIe speed; any host memory works.  If out_cap is too small the call
 * fails with S5GPU_ERR_NOMEM and out_off[0] = the capacity needed.
 * s5gpu_host_alloc never pins less than 2 MiB at a time (meant for chunk-sized buffers, not for small objects): round 5 met small pinned
 * buffers, allocated by one host thread while another was inside its first batch, that device copies did not reach. */
void *s5gpu_host_alloc(size_t bytes);
void s5gpu_host_free(void *p);
int s5gpu_recompress_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int from_rec,
                            int from_sig, int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux, void *out_buf, size_t out_cap,
                            uint64_t *out_off, int32_t *status);

/* The decode half alone on a chunk of framed records (`get --benchmark`, /root/reference/src/get.c:52, and any consumer of signals): the
 * decoded signals come back as ONE contiguous int16 block — sig_off[i] = first sample of record i, sig_off[n] = total — and the parsed
 * fields in fields[i]; no malloc per record.  sig_cap in samples; too little: S5GPU_ERR_NOMEM and sig_off[0] = samples needed.  A corrupt
 * record fails the call with S5GPU_ERR_DATA (fields[i].status says which; with several devices, records of a share that gave up because
 * another share failed read S5GPU_STATUS_NOT_DECODED, never 0). */
#define S5GPU_STATUS_NOT_DECODED 15
int s5gpu_decode_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int rec_method,
                        int sig_method, int16_t *sig_out, size_t sig_cap, uint64_t *sig_off, s5gpu_rec_fields_t *fields);

/* The read ids of the n records of a file chunk (framed as for s5gpu_recompress_stream), for the index builder (slow5_idx_create): only the
 * front of every zlib record crosses PCIe and only its first 2 + id_pitch bytes are inflated; uncompressed records are read on the host.
 * ids: n * id_pitch bytes (id i at ids + i * id_pitch, id_len[i] bytes, not terminated).  status[i] != 0: this record needs the general
 * decode (id longer than id_pitch: 5; corrupt: 1-4, 7).  rec_method zstd is refused (S5GPU_ERR_ARG). */
int s5gpu_record_ids_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int rec_method,
                            uint32_t id_pitch, char *ids, uint16_t *id_len, int32_t *status);

/* ---- one-stage host-buffer calls behind slow5_ptr_compress_solo / slow5_ptr_depress_solo ----
 * stage: 0 zlib compress, 1 zlib inflate, 2 svb-zd encode (in = int16 samples, in_len in bytes),
 * 3 svb-zd decode, 4 zstd decompress (whole frames), 5 zstd compress, 6 ex-zd encode (in = int16 samples), 7 ex-zd decode.
 * out[i] malloc'd, caller frees.  status[i] per record (0 ok), may be NULL. */
int s5gpu_solo_batch(int stage, uint32_t n, const void *const *in, const size_t *in_len, void **out, size_t *out_len,
                     int32_t *status);

/* ---- SLOW5 ASCII <-> BLOW5 (the parse/format half of slow5_rec_depress_parse / slow5_rec_to_mem
 * when one side of `view` is a .slow5 file, /root/reference/src/view.c:35-57) ----
 * The raw_signal column (comma-separated decimal int16) is ~95 % of an ASCII record: it is parsed / formatted on the
 * device, one read per workgroup.  The handful of scalar columns and the aux columns are converted on the host. */
typedef struct {             /* 32 B: one read's raw_signal text on the device */
    uint64_t txt_off;        /* byte offset of the text in `text` (any alignment) */
    uint64_t sig_off;        /* int16 index of the first sample in `sig` */
    uint32_t txt_len;        /* parse: length of the text; format: capacity of the text slot (7 bytes/sample is enough) */
    uint32_t n_samples;      /* parse: the count the len_raw_signal column promised; format: samples to print */
    uint32_t reserved[2];
} s5gpu_txt_desc_t;
/* status[i]: 0 ok, 1 bad character, 2 value out of int16 range / too many digits, 3 empty number, 4 count mismatch,
 * 5 text slot too small.  `text` needs 32 readable bytes after the last read. */
int s5gpu_ascii_parse_dev(uint32_t n, const s5gpu_txt_desc_t *desc, const uint8_t *text, int16_t *sig, int32_t *status, void *stream);
int s5gpu_ascii_format_dev(uint32_t n, const s5gpu_txt_desc_t *desc, const int16_t *sig, uint8_t *text, uint32_t *txt_len,
                           int32_t *status, void *stream);
/* copy n byte ranges src[src_off[i] .. +len[i]) -> dst[dst_off[i] ..) on the device */
/* dst[0, bytes) = src[0, bytes) by a kernel on `stream`: both 16-byte aligned, with room for `bytes` rounded up to 16.  dst may be pinned host
 * memory (s5gpu_host_alloc): results then travel as the kernel's own stores, not through the copy engines every stream of the process shares. */
int s5gpu_copy_dev(void *dst, const void *src, uint64_t bytes, void *stream);
int s5gpu_gather_dev(uint32_t n, const uint64_t *src_off, const uint32_t *len, const uint64_t *dst_off, const uint8_t *src, uint8_t *dst,
                     void *stream);

/* aux column types of a SLOW5 header, in column order: low 4 bits = element kind, bit 7 = array ("type*"; char* = string) */
enum { S5GPU_AUX_INT8 = 0, S5GPU_AUX_INT16, S5GPU_AUX_INT32, S5GPU_AUX_INT64, S5GPU_AUX_UINT8, S5GPU_AUX_UINT16, S5GPU_AUX_UINT32,
       S5GPU_AUX_UINT64, S5GPU_AUX_FLOAT, S5GPU_AUX_DOUBLE, S5GPU_AUX_CHAR, S5GPU_AUX_ENUM, S5GPU_AUX_ARRAY = 0x80 };
/* Types line of a SLOW5 header ("#char*\tuint32_t\t...") -> aux type codes of the columns after raw_signal.
 * Returns the number of aux columns (<= cap), or a negative S5GPU_ERR_*. */
int s5gpu_aux_types_parse(const char *types_line, size_t len, uint8_t *aux_type, uint32_t cap);

/* ASCII records (one line each, with or without the trailing newline) -> BLOW5 records [u64 size][press(payload)].
 * new_read_group / drop_aux as in s5gpu_recompress_batch.  status[i] (may be NULL): 0 ok, 1-5 as above, 16 malformed line. */
int s5gpu_ascii_to_blow5_batch(uint32_t n, const char *const *line, const size_t *line_len, uint32_t n_aux, const uint8_t *aux_type,
                               int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                               int32_t *status);
/* The same on a CHUNK of a .slow5 file, for a loop that reads the file in large pieces (examples/s5view.c; what s5gpu_recompress_stream
 * is for BLOW5 input): the n record lines sit in one host buffer `chunk` exactly as read from disk, line_pos[i] / line_len[i] = offset and
 * length of line i (with or without its newline).  The chunk is uploaded as it is and the raw_signal columns are parsed where they lie;
 * the BLOW5 records come back as ONE contiguous stream in out_buf (out_off[i] = offset of record i's u64 prefix, out_off[n] = total):
 * no per-line malloc or memcpy on either side.  Too little room: S5GPU_ERR_NOMEM and out_off[0] = the capacity needed.  `chunk` needs
 * 32 readable bytes behind the last line. */
int s5gpu_ascii_to_blow5_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *line_pos, const uint32_t *line_len,
                                uint32_t n_aux, const uint8_t *aux_type, int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux,
                                void *out_buf, size_t out_cap, uint64_t *out_off, int32_t *status);
/* slot r (desc[r].out_off in `slots`, len[r] bytes) -> dst + off[r], on the device (the copy of s5gpu_compact_dev with the destinations given) */
int s5gpu_scatter_slots_dev(uint32_t n, const s5gpu_read_desc_t *desc, const uint8_t *slots, const uint32_t *len, const uint64_t *off, uint8_t *dst,
                            void *hip_stream);
/* ... and the other way round on a CHUNK of a BLOW5 file (records framed as for s5gpu_recompress_stream): the SLOW5 text lines of the n records
 * come back as ONE contiguous block in out_buf (out_off[i] = start of line i, out_off[n] = total; every line ends in a newline) — the signal
 * columns printed on the device, prefix | signal | suffix of every line put in place there, one D2H.  Too little room: S5GPU_ERR_NOMEM and
 * out_off[0] = the capacity needed. */
int s5gpu_blow5_to_ascii_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int from_rec,
                                int from_sig, uint32_t n_aux, const uint8_t *aux_type, const uint32_t *new_read_group, int drop_aux,
                                void *out_buf, size_t out_cap, uint64_t *out_off, int32_t *status);
/* BLOW5 records (bytes without the u64 prefix) -> ASCII lines ending in a newline; out[i] malloc'd. */
int s5gpu_blow5_to_ascii_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, uint32_t n_aux,
                               const uint8_t *aux_type, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                               int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
