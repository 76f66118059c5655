// host_ctx.h — shared by the host-side translation units of libslow5gpu.so (host_api.hip, ascii_api.hip):
// error macro, grow-only workspaces, the per-process context and the threaded host loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/slow5gpu.h"

extern "C" void s5gpu_set_error(const char *fmt, ...);
extern "C" const char *s5gpu_last_error(void);

#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            s5gpu_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return S5GPU_ERR_HIP;                                                                    \
        }                                                                                            \
    } while (0)

// Pinned host memory: never less than S5_PIN_MIN bytes at a time, one host thread inside the allocator at a time (host_api.hip).
// Round 5: a SMALL pinned buffer (4.9 KB: the stream encoder's offsets) that a worker thread allocated while another worker was in its
// own first batch came back, one process in a hundred, as a buffer the stream's D2H copies never reached — hipMemcpyAsync + 
// hipStreamSynchronize succeed, a synchronous hipMemcpy into the same buffer lands, the stream's copies do not, however often they are
// repeated (tools/view_flake.sh: 8-12 of 900 runs of `s5view` with two workers; none with one worker, none with one context).  Buffers
// of 2 MiB and more — past the runtime's sub-allocator for small host allocations — do not show it: 0 of 900, and 0 of 600 with the
// small buffers allocated before the workers start.  Serialising the allocator calls alone did not help (9 of 900).
#ifndef S5_PIN_MIN
#define S5_PIN_MIN ((size_t)2 << 20)
#endif
hipError_t s5_pinned_alloc(void **p, size_t bytes, size_t *got);   // at least S5_PIN_MIN bytes; *got (may be NULL) = what was allocated
hipError_t s5_pinned_free(void *p);

// ---- grow-only device / pinned workspaces for the host-buffer batch calls ----
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    int reserve(size_t n) {
        if (n <= cap) return S5GPU_OK;
        if (p) { if (pinned) (void)s5_pinned_free(p); else (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 4096;
        hipError_t e = pinned ? s5_pinned_alloc(&p, want, &want) : hipMalloc(&p, want);
        if (e != hipSuccess) { s5gpu_set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e)); p = nullptr; return S5GPU_ERR_NOMEM; }
        cap = want;
        return S5GPU_OK;
    }
    void release() { if (p) { if (pinned) (void)s5_pinned_free(p); else (void)hipFree(p); } p = nullptr; cap = 0; }
};
struct Ctx {
    Buf d_sig, d_hdr, d_aux, d_desc, d_slots, d_len, d_ovf, d_in, d_pay, d_fields, d_stream, d_scan, d_sig2, d_desc2, d_patch;
    Buf d_txt, d_tdesc, d_gather;   // SLOW5 ASCII path (ascii_api.hip)
    Buf h_in, h_out;   // pinned staging
    hipStream_t st = nullptr;
    hipEvent_t ev_up = nullptr;   // "this context's uploads have landed" (upload_landed, host_api.hip); made on first use
    int slot = 0;                 // the device slot this context belongs to
    std::mutex mu;      // held by the batch call that owns this context
    Ctx() { h_in.pinned = true; h_out.pinned = true; }
};

namespace s5host {
extern std::mutex g_mu;
// A batch call owns one of a few contexts (workspaces + stream) for its duration, so host threads can run batches
// concurrently: one batch's PCIe copies overlap another's kernels.  S5GPU_CONTEXTS (default 2, max 4).
// slot = index into the devices the library was initialised on (s5gpu_init_mask); a host thread that holds a context has
// that device current.
struct CtxHold {
    Ctx *c = nullptr;
    int slot = 0;
    std::unique_lock<std::mutex> lk;
    int acquire(int slot = 0);
};
// devices in use (>= 1 once the library is initialised; initialises on device 0 if nobody has)
int n_devices();
// The reference splits a batch into one contiguous index range per worker thread (/root/reference/src/thread.c:76-90);
// here per DEVICE: device g of G takes [g*n/G, (g+1)*n/G), one host thread each, results land in the caller's own
// per-record slots, so the order is the caller's.  fn(slot, lo, hi) -> S5GPU_* code.  Batches of fewer than
// multi_min_per_device * G records stay on device 0.
int for_each_device_range(uint32_t n, const std::function<int(int, uint32_t, uint32_t)> &fn);
// The batch calls' ARENA form (round 5): the records of a call are not handed out as one malloc each but as pointers into a few
// pinned host buffers — the D2H lands in them directly — that the caller gives back with ONE s5gpu_arena_release.  The buffers come
// from a process-wide pool and return to it: a steady caller (view's loop: batch after batch of similar size) neither allocates nor
// page-faults after its first batches.  A call's shares (devices, pieces) each add the buffers they filled.
struct Arena {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> bufs;
    uint32_t generation = 0;
    uint32_t pool_gen = 0;      // the pool's generation when the arena was made (arena_pool_generation())
    void add(void *p, size_t cap) { std::lock_guard<std::mutex> g(mu); bufs.emplace_back(p, cap); }
};
void *arena_pool_take(size_t bytes, size_t *cap);   // pinned; NULL + error message when the allocation fails
void arena_pool_give(void *p, size_t cap, uint32_t gen);
void arena_pool_drain();                            // s5gpu_shutdown
uint32_t arena_pool_generation();
int arena_pool_set_keep(size_t bytes);              // option "arena_pool_keep_mb"
// encode descriptors already on the device -> one malloc per record on the host, or (ar != nullptr) pointers into arena buffers (host_api.hip)
int encode_and_collect(Ctx *c, uint32_t n, const std::vector<s5gpu_read_desc_t> &desc, s5gpu_encode_args_t a, uint64_t slots_bytes,
                       void **out, size_t *out_len, Arena *ar = nullptr);
// decode host records, results resident in c->d_pay / c->d_sig2 (host_api.hip)
int decode_resident(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                    std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status);
// ... the records sitting framed in one host buffer (a file chunk): [base, base + bytes) is uploaded as it is, rec[i] point into it
int decode_resident_framed(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                           std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status, const uint8_t *base, size_t bytes);
// encode descriptors already on the device -> the contiguous BLOW5 record stream in c->d_stream (what the ordered fwrite loop
// emits); off[i] / off[n] = record offsets / total, on the host (host_api.hip)
int encode_stream_resident(Ctx *c, uint32_t n, const std::vector<s5gpu_read_desc_t> &desc, s5gpu_encode_args_t a, uint64_t slots_bytes,
                           std::vector<uint64_t> &off);

// One output stream made of the shares of several devices (the chunk calls: s5gpu_recompress_stream, s5gpu_ascii_to_blow5_stream).
// Every device thread publishes the size of its share, waits for the shares in front of it (so it knows where its bytes go) and
// fetches them itself while it still owns its context.  If the whole does not fit the caller's buffer nobody copies, but every
// share still publishes its size: the caller learns the room the WHOLE output needs, whatever the number of devices.
struct ShareGather {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int64_t> totals;
    bool failed = false, overflow = false;
    int fail_rc = S5GPU_OK, fail_slot = -1;   // the FIRST failure: what the call reports, whichever share's thread returns first
    std::string fail_msg;
    explicit ShareGather(int G) : totals((size_t)G, -1) {}
    // a share gives up with rc (its message is this thread's s5gpu_last_error()); shares waiting behind it are released
    int fail(int rc, int slot = -1) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) { fail_rc = rc; fail_slot = slot; fail_msg = s5gpu_last_error(); }
        failed = true;
        cv.notify_all();
        return rc;
    }
    // 0 and *base (where this share starts) / *copy (false: the output does not fit, skip the copy).  A share whose predecessors have
    // all published goes on whatever happened behind it; one that waits on a failed share returns that share's code (report() below
    // names the share that failed, not the one that waited)
    int place(int slot, uint64_t total, size_t out_cap, uint64_t *base, bool *copy) {
        std::unique_lock<std::mutex> g(mu);
        totals[(size_t)slot] = (int64_t)total;
        cv.notify_all();
        uint64_t b = 0;
        bool ready = false;
        for (;;) {
            ready = true;
            b = 0;
            for (int q = 0; q < slot; q++) { if (totals[(size_t)q] < 0) ready = false; else b += (uint64_t)totals[(size_t)q]; }
            if (ready || failed) break;
            cv.wait(g);
        }
        if (!ready) return fail_rc;
        if (b + total > out_cap) overflow = true;
        *base = b;
        *copy = !overflow;
        return S5GPU_OK;
    }
    // after the device threads have joined: the code and message of the share that failed first (rc = what for_each_device_range
    // returned: the lowest slot's code, which may be a share that merely waited on the failing one)
    int report(int rc) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed) return rc;
        if (fail_slot >= 0 && totals.size() > 1) s5gpu_set_error("device slot %d: %s", fail_slot, fail_msg.c_str());
        else s5gpu_set_error("%s", fail_msg.c_str());
        return fail_rc;
    }
    uint64_t need() const {
        uint64_t s = 0;
        for (int64_t t : totals) if (t > 0) s += (uint64_t)t;
        return s;
    }
};
}  // namespace s5host

static inline uint64_t up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// S5GPU_TRACE=1 (tools): milliseconds since the calling thread's previous trace point, to stderr — where a chunk call's time goes
#include <time.h>
static inline void s5_trace(const char *what) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("S5GPU_TRACE"); on = e && atoi(e) ? 1 : 0; }
    if (!on) return;
    static thread_local double last = 0;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    const double now = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
    fprintf(stderr, "s5gpu[trace] %p +%8.3f ms  %s\n", (void *)&last, last ? 1e3 * (now - last) : 0.0, what);
    last = now;
}

// host-side packing / unpacking of a batch is plain memcpy work: spread it over a few threads
template <class F>
static void parallel_for(uint32_t n, uint64_t bytes, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = hw ? (hw < 16 ? hw : 16) : 4;
    if (bytes < (8u << 20) || n < 64 || nt < 2) { fn(0u, n); return; }
    std::vector<std::thread> th;
    const uint32_t step = (n + nt - 1) / nt;
    for (uint32_t lo = 0; lo < n; lo += step) th.emplace_back(fn, lo, lo + step < n ? lo + step : n);
    for (auto &t : th) t.join();
}

