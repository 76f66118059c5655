// host_api.hip — host side of libslow5gpu.so: lifetime, error reporting, events, and the
// host-buffer batch calls that stand where slow5tools' work_db() stands
// (/root/reference/src/thread.c:114, called at src/view.c:292, src/merge.c:440, src/get.c:364).
//
// No CPU fallback exists on purpose: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/slow5gpu.h"
#include <deque>
#include "host_ctx.h"

static thread_local char g_err[512] = "";
namespace s5host { std::mutex g_mu; }
using s5host::g_mu;

extern "C" void s5gpu_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *s5gpu_last_error(void) { return g_err; }

extern "C" int s5gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- devices and contexts ----
// The library runs on the devices named at initialisation (one for s5gpu_init, several for s5gpu_init_mask).  Each device
// has a few contexts (workspaces + stream); a batch call owns one for its duration.
static const int MAX_DEV = 16, MAX_CTX = 4;
struct DevState {
    int phys = -1;                 // HIP device ordinal
    Ctx *ctx[MAX_CTX] = {nullptr, nullptr, nullptr, nullptr};
    int nctx = 0;
};
static DevState g_dev[MAX_DEV];
static int g_ndev = 0;
static uint32_t g_multi_min = 1024;   // records per device below which a host batch is not split over devices
uint32_t s5host_generation = 1;       // bumped by s5gpu_shutdown: per-thread helper streams of kernels.hip are stale after it

static int init_devices_locked(const int *phys, int count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        s5gpu_set_error("s5gpu_init: no HIP device visible (this library has no CPU path)");
        return S5GPU_ERR_NODEV;
    }
    // S5GPU_ALIAS_DEVICES=1 (tests on a one-GPU box): an ordinal past the last device wraps around, so several logical
    // devices share a physical one and the multi-device split runs as it would on a node
    const char *al = getenv("S5GPU_ALIAS_DEVICES");
    const bool alias = al && atoi(al) != 0;
    int resolved[MAX_DEV];
    for (int i = 0; i < count; i++) {
        int d = phys[i];
        if (d >= n && alias) d %= n;
        if (d < 0 || d >= n) {
            s5gpu_set_error("s5gpu_init: device %d out of range (0..%d)", phys[i], n - 1);
            return S5GPU_ERR_ARG;
        }
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, d));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            s5gpu_set_error("s5gpu_init: device %d is %s; kernels are built for gfx950 only", d, prop.gcnArchName);
            return S5GPU_ERR_NODEV;
        }
        resolved[i] = d;
    }
    if (g_ndev) {   // re-initialisation on the same devices is a no-op; on other devices it needs s5gpu_shutdown first
        bool same = g_ndev == count;
        for (int i = 0; same && i < count; i++) same = g_dev[i].phys == resolved[i];
        if (same) return hipSetDevice(resolved[0]) == hipSuccess ? S5GPU_OK : S5GPU_ERR_HIP;
        for (int i = 0; i < g_ndev; i++)
            if (g_dev[i].nctx) { s5gpu_set_error("s5gpu_init: already initialised on other devices with live contexts; call s5gpu_shutdown first"); return S5GPU_ERR_ARG; }
    }
    for (int i = 0; i < count; i++) g_dev[i].phys = resolved[i];
    g_ndev = count;
    { const char *mm = getenv("S5GPU_MULTI_MIN"); if (mm && atoi(mm) >= 1) g_multi_min = (uint32_t)atoi(mm); }   // = option "multi_min_per_device"
    HIP_TRY(hipSetDevice(resolved[0]));
    return S5GPU_OK;
}

extern "C" int s5gpu_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    return init_devices_locked(&device, 1);
}

extern "C" int s5gpu_init_mask(uint64_t dev_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    int phys[MAX_DEV], count = 0;
    for (int d = 0; d < 64 && count < MAX_DEV; d++)
        if ((dev_mask >> d) & 1) phys[count++] = d;
    if (count == 0) { s5gpu_set_error("s5gpu_init_mask: empty device mask"); return S5GPU_ERR_ARG; }
    return init_devices_locked(phys, count);
}

extern "C" int s5gpu_devices_in_use(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_ndev;
}

static uint32_t g_host_turns = 1;      // option "host_turns": packing and uploading of concurrent batches take turns
static uint32_t g_d2h_kernel = 1;      // option "d2h_kernel_copy": results return to pinned host buffers by a kernel's stores, not by the copy engines
int s5host_set_option(const char *key, long value) {
    if (key && strcmp(key, "multi_min_per_device") == 0 && value >= 1) { g_multi_min = (uint32_t)value; return S5GPU_OK; }
    // pinned memory the arena pool keeps between batch calls, MiB (default 6144; 0: every released batch is unpinned at once)
    if (key && strcmp(key, "host_turns") == 0 && (value == 0 || value == 1)) { g_host_turns = (uint32_t)value; return S5GPU_OK; }
    if (key && strcmp(key, "d2h_kernel_copy") == 0 && (value == 0 || value == 1)) { g_d2h_kernel = (uint32_t)value; return S5GPU_OK; }
    if (key && strcmp(key, "arena_pool_keep_mb") == 0 && value >= 0) return s5host::arena_pool_set_keep((size_t)value << 20);
    return S5GPU_ERR_ARG;
}

int s5host::n_devices() {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_ndev) return g_ndev;
    }
    if (s5gpu_init(0)) return 0;
    return 1;
}

static std::mutex g_pin_mu;
hipError_t s5_pinned_alloc(void **p, size_t bytes, size_t *got) {
    if (bytes < S5_PIN_MIN) bytes = S5_PIN_MIN;
    if (got) *got = bytes;
    std::lock_guard<std::mutex> g(g_pin_mu);
    return hipHostMalloc(p, bytes, hipHostMallocPortable);
}
hipError_t s5_pinned_free(void *p) {
    std::lock_guard<std::mutex> g(g_pin_mu);
    return hipHostFree(p);
}

// Batches that are in flight TOGETHER (two tickets of s5gpu_*_batch_submit, the two halves of a big batch, several host threads) fall into
// step if nothing keeps them apart: both pack at once (twice the threads on the same memory), both upload at once (half the link each),
// both download at once — and nothing overlaps (round 6, tools/hook_trace.py: K = 4096, two in flight, 1.68 ms per batch against 1.70
// alone).  So the two phases that share a resource take TURNS: packing into pinned staging (the host's memory system), and the upload
// (the link's host-to-device direction), which is held until the copies have LANDED — an event behind them, waited for before the turn is
// given up.  The batch behind then packs while this one uploads and uploads while this one computes and downloads.
static int d2h(Ctx *c, void *dst, const void *src, size_t bytes) {      // dst: pinned, 16-byte aligned, room for bytes rounded up to 16; src: device, 16-byte aligned
    if (g_d2h_kernel && !(((uintptr_t)dst | (uintptr_t)src) & 15)) return s5gpu_copy_dev(dst, src, bytes, c->st);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->st));
    return S5GPU_OK;
}
static std::mutex g_pack_turn;
static std::mutex g_upload_turn[64];
static int upload_landed(Ctx *c) {
    if (!c->ev_up) HIP_TRY(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_up, c->st));
    HIP_TRY(hipEventSynchronize(c->ev_up));
    return S5GPU_OK;
}

int s5host::CtxHold::acquire(int want_slot) {
    if (s5host::n_devices() == 0) return S5GPU_ERR_NODEV;
    DevState *D;
    {
        std::lock_guard<std::mutex> g(g_mu);
        if (want_slot < 0 || want_slot >= g_ndev) { s5gpu_set_error("no device slot %d (library runs on %d device(s))", want_slot, g_ndev); return S5GPU_ERR_ARG; }
        D = &g_dev[want_slot];
        if (hipSetDevice(D->phys) != hipSuccess) { s5gpu_set_error("hipSetDevice(%d) failed", D->phys); return S5GPU_ERR_HIP; }   // per host thread
        if (D->nctx == 0) {
            const char *e = getenv("S5GPU_CONTEXTS");
            int want = e ? atoi(e) : 2;
            want = want < 1 ? 1 : want > MAX_CTX ? MAX_CTX : want;
            for (int i = 0; i < want; i++) {
                Ctx *c = new Ctx();
                c->slot = want_slot;
                if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) {
                    delete c;
                    s5gpu_set_error("hipStreamCreate failed");
                    if (D->nctx == 0) return S5GPU_ERR_HIP;
                    break;
                }
                D->ctx[D->nctx++] = c;
            }
        }
    }
    slot = want_slot;
    for (int i = 0; i < D->nctx; i++) {
        std::unique_lock<std::mutex> t(D->ctx[i]->mu, std::try_to_lock);
        if (t.owns_lock()) { lk = std::move(t); c = D->ctx[i]; return S5GPU_OK; }
    }
    const size_t pick = std::hash<std::thread::id>()(std::this_thread::get_id()) % (size_t)D->nctx;
    lk = std::unique_lock<std::mutex>(D->ctx[pick]->mu);
    c = D->ctx[pick];
    return S5GPU_OK;
}

// s5gpu_warmup: the first device's contexts (streams) exist before the first batch call
int s5host_warm_contexts() {
    s5host::CtxHold hold;
    return hold.acquire(0);
}

int s5host::for_each_device_range(uint32_t n, const std::function<int(int, uint32_t, uint32_t)> &fn) {
    int G = s5host::n_devices();
    if (G == 0) return S5GPU_ERR_NODEV;
    if ((uint64_t)n < (uint64_t)g_multi_min * (uint64_t)G) G = 1;
    if (G == 1) return fn(0, 0u, n);
    std::vector<int> rcs(G, S5GPU_OK);
    std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    auto body = [&](int g) {
        const uint32_t lo = (uint32_t)((uint64_t)n * g / G), hi = (uint32_t)((uint64_t)n * (g + 1) / G);
        s5gpu_set_error("%s", "");                  // a share that gives up with a bare code must not report an older call's message of this thread
        rcs[g] = lo < hi ? fn(g, lo, hi) : S5GPU_OK;
        if (rcs[g]) errs[g] = s5gpu_last_error();   // the message lives in the thread that failed
    };
    for (int g = 1; g < G; g++) th.emplace_back(body, g);
    body(0);
    for (auto &t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[g]) { s5gpu_set_error("device slot %d: %s", g, errs[g].c_str()); return rcs[g]; }
    return S5GPU_OK;
}

using s5host::encode_and_collect;
void s5kern_release_aux();   // kernels.hip
void s5kern_release_order();

void s5host_stop_ticket_workers();
extern "C" void s5gpu_shutdown(void) {
    s5host_stop_ticket_workers();
    std::lock_guard<std::mutex> lk(g_mu);
    for (int d = 0; d < g_ndev; d++) {
        DevState &D = g_dev[d];
        if (D.nctx) (void)hipSetDevice(D.phys);
        for (int i = 0; i < D.nctx; i++) {
            Ctx *c = D.ctx[i];
            {
                std::lock_guard<std::mutex> own(c->mu);   // wait for a batch still running on it
                Buf *bs[] = {&c->d_sig, &c->d_hdr, &c->d_aux, &c->d_desc, &c->d_slots, &c->d_len, &c->d_ovf, &c->d_in, &c->d_pay, &c->d_fields,
                             &c->d_stream, &c->d_scan, &c->d_sig2, &c->d_desc2, &c->d_patch, &c->d_txt, &c->d_tdesc, &c->d_gather, &c->h_in, &c->h_out};
                for (Buf *b : bs) b->release();
                if (c->ev_up) (void)hipEventDestroy(c->ev_up);
                if (c->st) (void)hipStreamDestroy(c->st);
            }
            delete c;
            D.ctx[i] = nullptr;
        }
        D.nctx = 0;
        D.phys = -1;
    }
    g_ndev = 0;
    s5kern_release_aux();
    s5kern_release_order();
    s5host::arena_pool_drain();
    s5host_generation++;
}

// ---- events (bench.py times the kernels on the stream they run on) ----
extern "C" int s5gpu_event_create(void **ev) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (void *)e;
    return S5GPU_OK;
}
extern "C" int s5gpu_event_record(void *ev, void *stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return S5GPU_OK;
}
extern "C" int s5gpu_event_elapsed_ms(void *a, void *b, float *ms) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return S5GPU_OK;
}
extern "C" int s5gpu_event_destroy(void *ev) {
    HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return S5GPU_OK;
}

// Run the encode for descriptors already on the device and leave the contiguous BLOW5 record stream (the bytes the ordered
// fwrite loop emits) in c->d_stream; off[i] / off[n] = record offsets / total, on the host.
int s5host::encode_stream_resident(Ctx *c, uint32_t n, const std::vector<s5gpu_read_desc_t> &desc, s5gpu_encode_args_t a, uint64_t slots_bytes,
                                   std::vector<uint64_t> &off) {
    int rc;
    off.resize((size_t)n + 1);
    if ((rc = c->d_len.reserve(4ull * n))) return rc;
    a.out_len = (uint32_t *)c->d_len.p;
    // First choice: ordered single-pass output — records land in the contiguous stream directly.  It needs every read
    // to fit the LDS budget of the fused kernel; if the device reports otherwise, the slot path below redoes the batch.
    // (signal press none goes through the LZ77 matcher of the slot path instead: see s5gpu_encode_dev)
    if (a.rec_method == S5GPU_REC_ZLIB && a.sig_method != S5GPU_SIG_NONE && (uint64_t)a.max_payload * 100 / (a.sig_method == S5GPU_SIG_EX_ZD ? 950 : 325) <= 16384) {
        const size_t scan_bytes = 8ull * (n + 1) + 8ull * n + 16;
        if ((rc = c->d_stream.reserve(slots_bytes + 64)) || (rc = c->d_scan.reserve(scan_bytes)) ||
            (rc = c->h_out.reserve(up(scan_bytes, 64) + 64)))
            return rc;
        uint64_t *d_off = (uint64_t *)c->d_scan.p, *d_state = d_off + (n + 1);
        uint32_t *d_ctl = (uint32_t *)(d_state + n);
        if ((rc = s5gpu_encode_stream_dev(&a, (uint8_t *)c->d_stream.p, d_off, d_state, d_ctl, c->st))) return rc;
        uint8_t *h = (uint8_t *)c->h_out.p;
        if ((rc = d2h(c, h, d_off, 8ull * (n + 1) + 8ull * n + 16))) return rc;      // offsets | states | control words: one piece (d_scan's layout)
        HIP_TRY(hipStreamSynchronize(c->st));
        s5_trace("encode_stream_resident: uploads + kernel done, offsets back");
        const uint32_t *ctl = (const uint32_t *)(h + 8ull * (n + 1) + 8ull * n);
        if (ctl[0] == 0 && ctl[2] == 0) {
            memcpy(off.data(), h, 8ull * (n + 1));
            for (uint32_t i = 0; i < n; i++)
                if (off[i + 1] < off[i] + 8 || off[i + 1] - off[i] > desc[i].slot_cap) {
                    s5gpu_set_error("read %u of %u: device produced an impossible record extent (offsets %llu .. %llu, slot %u; next %llu)", i, n, (unsigned long long)off[i],
                                    (unsigned long long)off[i + 1], desc[i].slot_cap, (unsigned long long)(i + 2 <= n ? off[i + 2 <= n ? i + 2 : n] : 0));
                    return S5GPU_ERR_HIP;
                }
            return S5GPU_OK;
        }
    }
    if ((rc = c->d_slots.reserve(slots_bytes + 64)) || (rc = c->d_ovf.reserve(4ull * n + 64)) ||
        (rc = c->h_out.reserve(up(4ull * n, 64) + 64)))
        return rc;
    a.slots = (uint8_t *)c->d_slots.p; a.out_len = (uint32_t *)c->d_len.p;
    a.lds_payload_cap = 0;
    a.ovf = (uint32_t *)c->d_ovf.p;
    if (a.rec_method != S5GPU_REC_NONE) {
        // A batch of a real run mixes read lengths over two decades.  The device entry point only knows the longest read and
        // would send such a batch through the staged kernels as a whole; the lengths are known here: when a tenth of the
        // reads or more fit an 8 KiB payload, those take the fused kernel and the rest its overflow list (measured on
        // log-normal lengths, median 6000 samples: 233 -> 285 GB/s; a batch of long reads only loses 3 % to the attempt)
        const double per = a.sig_method == S5GPU_SIG_SVB_ZD ? 1.55 : a.sig_method == S5GPU_SIG_EX_ZD ? 1.30 : 2.0;
        uint32_t fit = 0;
        for (uint32_t i = 0; i < n; i++) fit += desc[i].hdr_len + 8.0 + desc[i].aux_len + per * desc[i].n_samples + 64 <= 8192.0;
        const uint32_t div = a.sig_method == S5GPU_SIG_SVB_ZD ? 325 : a.sig_method == S5GPU_SIG_EX_ZD ? 950 : 100;
        if ((uint64_t)a.max_payload * 100 / div > 4ull * 16384 && fit >= n / 10 && fit > 0) a.lds_payload_cap = 8192;
    }
    if ((rc = s5gpu_encode_dev(&a, c->st))) return rc;
    uint8_t *ho_len = (uint8_t *)c->h_out.p;
    HIP_TRY(hipMemcpyAsync(ho_len, c->d_len.p, 4ull * n, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    const uint32_t *lens = (const uint32_t *)ho_len;
    off[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (lens[i] < 8 || lens[i] > desc[i].slot_cap) { s5gpu_set_error("read %u: device produced an impossible length %u", i, lens[i]); return S5GPU_ERR_HIP; }
        off[i + 1] = off[i] + lens[i];
    }
    const uint64_t produced = off[n];
    if ((rc = c->d_stream.reserve(produced + 64)) || (rc = c->d_scan.reserve(8ull * (n + 1) + 8ull * (n / 1024 + 8)))) return rc;
    uint64_t *d_off = (uint64_t *)c->d_scan.p, *d_tmp = d_off + (n + 1);
    return s5gpu_compact_dev(n, a.desc, (const uint8_t *)c->d_slots.p, (const uint32_t *)c->d_len.p, d_off, (uint8_t *)c->d_stream.p, d_tmp, c->st);
}

// ---- pool of pinned buffers behind the arena form of the batch calls ----
namespace {
struct PoolBuf { void *p; size_t cap; };
std::mutex g_pool_mu;
std::vector<PoolBuf> g_pool;
size_t g_pool_bytes = 0;
size_t g_pool_keep = 6ull << 30;   // what the pool holds on to between calls (a 1 M-read batch gives back ~3.6 GB); option "arena_pool_keep_mb"
uint32_t g_pool_gen = 0;           // the library generation the pooled buffers belong to (a shutdown drains the pool and moves on)
}  // namespace
void *s5host::arena_pool_take(size_t bytes, size_t *cap) {
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); i++)   // smallest buffer that fits and is not wastefully large
            if (g_pool[i].cap >= bytes && g_pool[i].cap <= 2 * bytes + (8u << 20) && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
        if (best >= 0) {
            const PoolBuf b = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
            g_pool_bytes -= b.cap;
            *cap = b.cap;
            return b.p;
        }
    }
    size_t want = (size_t)up(bytes + bytes / 8 + 4096, 1u << 20);
    void *p = nullptr;
    const hipError_t e = s5_pinned_alloc(&p, want, &want);
    if (e != hipSuccess) { s5gpu_set_error("arena allocation of %zu pinned bytes failed: %s", want, hipGetErrorString(e)); return nullptr; }
    *cap = want;
    return p;
}
// gen: the library generation the buffer was taken in.  The comparison with the pool's own generation happens UNDER the pool's lock, so a
// release that races a shutdown either gets in before the drain (and is drained with the rest) or sees the new generation and frees its
// buffer itself — never parks a buffer of the old generation in the new pool (round-5 advisor finding).
void s5host::arena_pool_give(void *p, size_t cap, uint32_t gen) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (gen == g_pool_gen && g_pool_bytes + cap <= g_pool_keep && g_pool.size() < 64) { g_pool.push_back({p, cap}); g_pool_bytes += cap; return; }
    }
    (void)s5_pinned_free(p);
}
void s5host::arena_pool_drain() {
    std::lock_guard<std::mutex> g(g_pool_mu);
    for (const PoolBuf &b : g_pool) (void)s5_pinned_free(b.p);
    g_pool.clear();
    g_pool_bytes = 0;
    g_pool_gen++;                  // (s5gpu_shutdown bumps s5host_generation right behind this call: the two move together)
}
uint32_t s5host::arena_pool_generation() {
    std::lock_guard<std::mutex> g(g_pool_mu);
    return g_pool_gen;
}
int s5host::arena_pool_set_keep(size_t bytes) {
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_pool_keep = bytes;
    while (g_pool_bytes > g_pool_keep && !g_pool.empty()) {      // a lower limit takes effect at once
        (void)s5_pinned_free(g_pool.back().p);
        g_pool_bytes -= g_pool.back().cap;
        g_pool.pop_back();
    }
    return S5GPU_OK;
}
extern "C" void s5gpu_arena_release(void *arena) {
    s5host::Arena *ar = (s5host::Arena *)arena;
    if (!ar) return;
    for (auto &b : ar->bufs) s5host::arena_pool_give(b.first, b.second, ar->pool_gen);   // (a batch that outlived a shutdown: its buffers are freed, not pooled)
    delete ar;
}

// ... then bring back only what was produced and hand out one malloc per record — or, with an arena, pointers into the buffer the D2H filled.
int s5host::encode_and_collect(Ctx *c, uint32_t n, const std::vector<s5gpu_read_desc_t> &desc, s5gpu_encode_args_t a, uint64_t slots_bytes,
                              void **out, size_t *out_len, Arena *ar) {
    std::vector<uint64_t> off;
    int rc = s5host::encode_stream_resident(c, n, desc, a, slots_bytes, off);
    if (rc) return rc;
    const uint64_t produced = off[n];
    if (ar) {
        size_t cap = 0;
        uint8_t *buf = (uint8_t *)arena_pool_take(produced + 64, &cap);
        if (!buf) return S5GPU_ERR_NOMEM;
        ar->add(buf, cap);                                  // (from here on the arena owns it, whatever happens)
        if ((rc = d2h(c, buf, c->d_stream.p, produced))) return rc;                  // (the pool's buffers hold produced + 64 bytes at least)
        HIP_TRY(hipStreamSynchronize(c->st));
        s5_trace("encode_and_collect: records back in the arena buffer");
        for (uint32_t i = 0; i < n; i++) { out[i] = buf + off[i]; out_len[i] = (size_t)(off[i + 1] - off[i]); }
        return S5GPU_OK;
    }
    if ((rc = c->h_out.reserve(produced + 64))) return rc;
    uint8_t *ho_stream = (uint8_t *)c->h_out.p;
    HIP_TRY(hipMemcpyAsync(ho_stream, c->d_stream.p, produced, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    int oom = 0;
    parallel_for(n, produced, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) {
            const size_t len = (size_t)(off[i + 1] - off[i]);
            void *b = malloc(len);
            if (!b) { oom = 1; out[i] = NULL; continue; }
            memcpy(b, ho_stream + off[i], len);
            out[i] = b;
            out_len[i] = len;
        }
    });
    if (oom) { for (uint32_t j = 0; j < n; j++) { free(out[j]); out[j] = NULL; } return S5GPU_ERR_NOMEM; }
    return S5GPU_OK;
}

// The whole batch in one call: H2D of signals/headers, one launch, D2H of the slots, one malloc per
// record (the ownership contract of slow5_rec_to_mem: caller frees each buffer, src/view.c:298).
static int encode_batch_one(int slot, uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                            const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                            int sig_method, void **out, size_t *out_len, s5host::Arena *ar);

// one device's share of a batch
static int encode_batch_dev(int slot, uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                            const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                            int sig_method, void **out, size_t *out_len, s5host::Arena *ar) {
    // A big batch is cut in pieces that run on two contexts at once: one piece's H2D overlaps the other's kernels and D2H (PCIe is
    // full duplex, and the host-side packing of one piece hides behind the copies of the other).  Two halves up to 131072 reads; beyond that
    // pieces of about 65536 reads, two host threads taking them in turn — the pinned staging stays at a few hundred MB whatever the batch
    // (round 4: 1 M reads in one call went through two 4 GB halves at 4.1 GB/s; 65536-read pieces run at 20+).
    const char *e = getenv("S5GPU_SPLIT");
    const bool split = n >= 16384 && (!e || atoi(e) != 0);
    if (!split) return encode_batch_one(slot, n, sig, n_samples, hdr, hdr_len, aux, aux_len, rec_method, sig_method, out, out_len, ar);
    const char *pe = getenv("S5GPU_PIECE");
    const uint32_t piece = pe && atoi(pe) >= 1024 ? (uint32_t)atoi(pe) : 65536u;
    const uint32_t P = n <= 2 * piece ? 2u : (n + piece - 1) / piece;
    std::atomic<uint32_t> next{0};
    std::atomic<int> rcs[2] = {{S5GPU_OK}, {S5GPU_OK}};            // (each written by its own thread, read by both)
    char errs[2][512] = {"", ""};
    auto body = [&](int w) {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= P || rcs[0].load() || rcs[1].load()) return;
            const uint32_t lo = (uint32_t)((uint64_t)n * k / P), hi = (uint32_t)((uint64_t)n * (k + 1) / P);
            const int rc = encode_batch_one(slot, hi - lo, sig + lo, n_samples + lo, hdr + lo, hdr_len + lo, aux ? aux + lo : nullptr, aux_len ? aux_len + lo : nullptr,
                                            rec_method, sig_method, out + lo, out_len + lo, ar);
            if (rc) { snprintf(errs[w], sizeof errs[w], "%s", s5gpu_last_error()); rcs[w].store(rc); return; }   // the message lives in that thread
        }
    };
    std::thread t(body, 1);
    body(0);
    t.join();
    if (rcs[0].load() || rcs[1].load()) {
        const int w = rcs[0].load() ? 0 : 1;
        s5gpu_set_error("%s", errs[w]);
        return rcs[w].load();
    }
    return S5GPU_OK;
}

static int encode_batch_any(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                            const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                            int sig_method, void **out, size_t *out_len, void **arena) {
    if (arena) *arena = NULL;
    if (n == 0) return S5GPU_OK;
    if (!sig || !n_samples || !hdr || !hdr_len || !out || !out_len) { s5gpu_set_error("s5gpu_encode_batch: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) out[i] = NULL;
    s5host::Arena *ar = nullptr;
    if (arena) {
        if (s5host::n_devices() == 0) return S5GPU_ERR_NODEV;   // (initialises the library: the arena remembers its generation)
        ar = new s5host::Arena;
        ar->generation = s5host_generation;
        ar->pool_gen = s5host::arena_pool_generation();
    }
    // contiguous index range per device (src/thread.c:76-90 does the same per thread); every record's result lands in the
    // caller's out[i], so the ordered fwrite loop of src/view.c:296-299 is untouched
    const int rc = s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) {
        return encode_batch_dev(slot, hi - lo, sig + lo, n_samples + lo, hdr + lo, hdr_len + lo, aux ? aux + lo : nullptr,
                                aux_len ? aux_len + lo : nullptr, rec_method, sig_method, out + lo, out_len + lo, ar);
    });
    if (rc) {
        for (uint32_t i = 0; i < n; i++) { if (!ar) free(out[i]); out[i] = NULL; }
        if (ar) s5gpu_arena_release(ar);
        return rc;
    }
    if (arena) *arena = ar;
    return rc;
}
extern "C" int s5gpu_encode_batch(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                                  const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                                  int sig_method, void **out, size_t *out_len) {
    return encode_batch_any(n, sig, n_samples, hdr, hdr_len, aux, aux_len, rec_method, sig_method, out, out_len, nullptr);
}
extern "C" int s5gpu_encode_batch_arena(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                                        const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                                        int sig_method, void **out, size_t *out_len, void **arena) {
    if (!arena) { s5gpu_set_error("s5gpu_encode_batch_arena: NULL arena"); return S5GPU_ERR_ARG; }
    return encode_batch_any(n, sig, n_samples, hdr, hdr_len, aux, aux_len, rec_method, sig_method, out, out_len, arena);
}

static int encode_batch_one(int slot, uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                            const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                            int sig_method, void **out, size_t *out_len, s5host::Arena *ar) {
    s5host::CtxHold hold;
    int rc = hold.acquire(slot);
    if (rc) return rc;
    Ctx *c = hold.c;
    s5_trace("encode_batch: context held");
    std::vector<s5gpu_read_desc_t> desc(n);
    uint64_t so = 0, ho = 0, ao = 0, oo = 0;
    uint32_t max_payload = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (n_samples[i] > 0xFFFFFFF0ull) { s5gpu_set_error("read %u: %llu samples exceed the u32 svb-zd count", i, (unsigned long long)n_samples[i]); return S5GPU_ERR_ARG; }
        s5gpu_read_desc_t &d = desc[i];
        d.sig_off = so; d.hdr_off = ho; d.aux_off = ao; d.out_off = oo;
        d.n_samples = (uint32_t)n_samples[i];
        d.hdr_len = hdr_len[i];
        d.aux_len = (aux && aux_len) ? aux_len[i] : 0;
        const uint64_t pb = s5gpu_payload_bound(d.n_samples, d.hdr_len, d.aux_len, sig_method);
        const uint64_t sb = s5gpu_slot_bound(d.n_samples, d.hdr_len, d.aux_len, rec_method, sig_method);
        if (pb > 0xFFFFFF00ull) { s5gpu_set_error("read %u: record larger than 4 GiB", i); return S5GPU_ERR_ARG; }
        d.slot_cap = (uint32_t)sb;
        if (pb > max_payload) max_payload = (uint32_t)pb;
        so += up(d.n_samples, 8);
        ho += d.hdr_len;
        ao += d.aux_len;
        oo += sb;
    }
    const size_t sig_bytes = (size_t)so * 2 + 64, in_bytes = up(sig_bytes, 64) + up(ho + 64, 64) + up(ao + 64, 64) + sizeof(s5gpu_read_desc_t) * n;
    if ((rc = c->h_in.reserve(in_bytes)) || (rc = c->d_sig.reserve(sig_bytes)) || (rc = c->d_hdr.reserve(ho + 64)) ||
        (rc = c->d_aux.reserve(ao + 64)) || (rc = c->d_desc.reserve(sizeof(s5gpu_read_desc_t) * n)))
        return rc;
    s5_trace("encode_batch: descriptors made, workspaces reserved");
    // pack into pinned staging
    uint8_t *hs = (uint8_t *)c->h_in.p;
    uint8_t *hh = hs + up(sig_bytes, 64), *ha = hh + up(ho + 64, 64), *hd = ha + up(ao + 64, 64);
    {
        std::unique_lock<std::mutex> turn(g_pack_turn, std::defer_lock);
        if (g_host_turns) turn.lock();
        parallel_for(n, (uint64_t)so * 2, [&](uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; i++) {
                const s5gpu_read_desc_t &d = desc[i];
                if (d.n_samples) memcpy(hs + 2 * d.sig_off, sig[i], 2ull * d.n_samples);
                memcpy(hh + d.hdr_off, hdr[i], d.hdr_len);
                if (d.aux_len) memcpy(ha + d.aux_off, aux[i], d.aux_len);
            }
        });
        memcpy(hd, desc.data(), sizeof(s5gpu_read_desc_t) * n);
    }
    s5_trace("encode_batch: packed into pinned staging");
    {
        std::unique_lock<std::mutex> turn(g_upload_turn[c->slot & 63], std::defer_lock);
        if (g_host_turns) turn.lock();
        HIP_TRY(hipMemcpyAsync(c->d_sig.p, hs, (size_t)so * 2, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_hdr.p, hh, ho, hipMemcpyHostToDevice, c->st));
        if (ao) HIP_TRY(hipMemcpyAsync(c->d_aux.p, ha, ao, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_read_desc_t) * n, hipMemcpyHostToDevice, c->st));
        if (g_host_turns && (rc = upload_landed(c))) return rc;
    }
    s5_trace("encode_batch: uploads landed");
    s5gpu_encode_args_t a;
    memset(&a, 0, sizeof a);
    a.n_reads = n; a.rec_method = rec_method; a.sig_method = sig_method;
    a.desc = (const s5gpu_read_desc_t *)c->d_desc.p;
    a.sig = (const int16_t *)c->d_sig.p; a.hdr = (const uint8_t *)c->d_hdr.p; a.aux = (const uint8_t *)c->d_aux.p;
    a.max_payload = max_payload;
    return encode_and_collect(c, n, desc, a, oo, out, out_len, ar);
}

// first guess at a record's uncompressed size: a zstd frame says it in its header, zlib does not
static uint64_t payload_guess(int rec_method, const void *rec, size_t len) {
    if (rec_method == S5GPU_REC_ZLIB) return 4ull * len + 4096;
    if (rec_method == S5GPU_REC_ZSTD) {
        const uint8_t *p = (const uint8_t *)rec;
        if (len >= 6 && p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD) {
            const unsigned fhd = p[4], flag = fhd >> 6, single = (fhd >> 5) & 1;
            const unsigned nb = flag == 0 ? single : flag == 1 ? 2 : flag == 2 ? 4 : 8;
            const size_t at = 5 + (single ? 0 : 1);
            if (nb && at + nb <= len) {
                uint64_t v = 0;
                for (unsigned i = 0; i < nb; i++) v |= (uint64_t)p[at + i] << (8 * i);
                return flag == 1 ? v + 256 : v;
            }
            if (!nb && at <= len) {
                // no content size in the header (streaming writers): walk the block headers — a raw / RLE block says its size,
                // a compressed one regenerates at most 128 KiB (Block_Maximum_Size)
                uint64_t bound = 0;
                size_t q = at;
                for (;;) {
                    if (q + 3 > len) break;
                    const uint32_t bh = p[q] | (p[q + 1] << 8) | ((uint32_t)p[q + 2] << 16);
                    const uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
                    q += 3;
                    if (type == 0) { bound += bs; q += bs; }
                    else if (type == 1) { bound += bs; q += 1; }
                    else { bound += 128 * 1024; q += bs; }
                    if ((bh & 1) || type == 3) break;
                }
                return bound + 16;
            }
        }
        return 8ull * len + 4096;
    }
    return len;
}

static int decode_batch_dev(int slot, uint32_t n, const void *const *rec, const size_t *rec_len, int rec_method, int sig_method,
                            void **payload, int16_t **sig, s5gpu_rec_fields_t *fields);

extern "C" int s5gpu_decode_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int rec_method, int sig_method,
                                  void **payload, int16_t **sig, s5gpu_rec_fields_t *fields) {
    if (n == 0) return S5GPU_OK;
    if (!rec || !rec_len || !payload || !sig || !fields) { s5gpu_set_error("s5gpu_decode_batch: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) { payload[i] = NULL; sig[i] = NULL; }
    return s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) {
        return decode_batch_dev(slot, hi - lo, rec + lo, rec_len + lo, rec_method, sig_method, payload + lo, sig + lo, fields + lo);
    });
}

static int decode_batch_dev(int slot, uint32_t n, const void *const *rec, const size_t *rec_len, int rec_method, int sig_method,
                            void **payload, int16_t **sig, s5gpu_rec_fields_t *fields) {
    s5host::CtxHold hold;
    int rc = hold.acquire(slot);
    if (rc) return rc;
    Ctx *c = hold.c;
    std::vector<s5gpu_rec_desc_t> desc(n);
    std::vector<uint8_t> done(n, 0);
    // capacity guesses; records that overflow report the size they need and are retried once
    std::vector<uint32_t> pcap(n), scap(n);
    for (uint32_t i = 0; i < n; i++) {
        if (rec_len[i] > 0xFFFFFF00ull) { s5gpu_set_error("record %u larger than 4 GiB", i); return S5GPU_ERR_ARG; }
        const uint64_t g = payload_guess(rec_method, rec[i], rec_len[i]);
        pcap[i] = (uint32_t)(g > 0xFFFFFF00ull ? 0xFFFFFF00ull : g);
        scap[i] = pcap[i];   // >= 1 byte per sample in either signal format... refined below
    }
    int overall = S5GPU_OK;
    for (int attempt = 0; attempt < 3; attempt++) {
        std::vector<uint32_t> idx;
        for (uint32_t i = 0; i < n; i++) if (!done[i]) idx.push_back(i);
        if (idx.empty()) break;
        const uint32_t m = (uint32_t)idx.size();
        uint64_t io = 0, po = 0, so = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = idx[k];
            s5gpu_rec_desc_t &d = desc[k];
            d.in_off = io; d.pay_off = po; d.sig_off = so;
            d.in_len = (uint32_t)rec_len[i]; d.pay_cap = pcap[i]; d.sig_cap = scap[i]; d.reserved = 0;
            io += up(rec_len[i], 16); po += up((uint64_t)pcap[i] + 16, 16); so += up((uint64_t)scap[i] + 8, 8);
        }
        const size_t hin = up(io + 64, 64) + sizeof(s5gpu_rec_desc_t) * m;
        const size_t hout = up(po + 64, 64) + up(so * 2 + 64, 64) + sizeof(s5gpu_rec_fields_t) * m;
        if ((rc = c->h_in.reserve(hin)) || (rc = c->d_in.reserve(io + 64)) || (rc = c->d_desc.reserve(sizeof(s5gpu_rec_desc_t) * m)) ||
            (rc = c->d_pay.reserve(po + 64)) || (rc = c->d_sig.reserve(so * 2 + 64)) || (rc = c->d_fields.reserve(sizeof(s5gpu_rec_fields_t) * m)) ||
            (rc = c->h_out.reserve(hout)))
            return rc;
        uint8_t *hi = (uint8_t *)c->h_in.p, *hd = hi + up(io + 64, 64);
        for (uint32_t k = 0; k < m; k++) memcpy(hi + desc[k].in_off, rec[idx[k]], rec_len[idx[k]]);
        memcpy(hd, desc.data(), sizeof(s5gpu_rec_desc_t) * m);
        HIP_TRY(hipMemcpyAsync(c->d_in.p, hi, io, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_rec_desc_t) * m, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemsetAsync(c->d_fields.p, 0, sizeof(s5gpu_rec_fields_t) * m, c->st));
        s5gpu_decode_args_t a;
        memset(&a, 0, sizeof a);
        a.n_recs = m; a.rec_method = rec_method; a.sig_method = sig_method;
        a.desc = (const s5gpu_rec_desc_t *)c->d_desc.p; a.in = (const uint8_t *)c->d_in.p;
        a.payload = (uint8_t *)c->d_pay.p; a.sig_out = (int16_t *)c->d_sig.p; a.fields = (s5gpu_rec_fields_t *)c->d_fields.p;
        for (uint32_t k = 0; k < m; k++) if (desc[k].pay_cap > a.max_pay_cap) a.max_pay_cap = desc[k].pay_cap;   // (short records: the inflate kernel's 24-wave shape)
        if ((rc = s5gpu_decode_dev(&a, c->st))) return rc;
        uint8_t *hp = (uint8_t *)c->h_out.p, *hsg = hp + up(po + 64, 64), *hf = hsg + up(so * 2 + 64, 64);
        HIP_TRY(hipMemcpyAsync(hf, c->d_fields.p, sizeof(s5gpu_rec_fields_t) * m, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipMemcpyAsync(hp, c->d_pay.p, po, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipMemcpyAsync(hsg, c->d_sig.p, so * 2, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        const s5gpu_rec_fields_t *ff = (const s5gpu_rec_fields_t *)hf;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = idx[k];
            const s5gpu_rec_fields_t &f = ff[k];
            if (f.status == 5 && attempt < 2) { pcap[i] = f.payload_len ? f.payload_len : (pcap[i] < 0x10000000u ? 8 * pcap[i] + 65536 : 0xFFFFFF00u); if (scap[i] < f.payload_len) scap[i] = f.payload_len; continue; }
            if (f.status == 6 && attempt < 2) { scap[i] = f.n_samples; continue; }
            fields[i] = f;
            done[i] = 1;
            if (f.status != 0) { overall = S5GPU_ERR_DATA; continue; }
            payload[i] = malloc(f.payload_len ? f.payload_len : 1);
            sig[i] = (int16_t *)malloc(f.n_samples ? 2ull * f.n_samples : 2);
            if (!payload[i] || !sig[i]) return S5GPU_ERR_NOMEM;
            memcpy(payload[i], hp + desc[k].pay_off, f.payload_len);
            memcpy(sig[i], hsg + 2 * desc[k].sig_off, 2ull * f.n_samples);
        }
    }
    if (overall) s5gpu_set_error("s5gpu_decode_batch: at least one record is corrupt (see fields[i].status)");
    return overall;
}

// ---- single-stage host-buffer batches (slow5_ptr_compress_solo / slow5_ptr_depress_solo) ----
extern "C" int s5gpu_solo_batch(int stage, uint32_t n, const void *const *in, const size_t *in_len, void **out, size_t *out_len,
                                int32_t *status) {
    if (n == 0) return S5GPU_OK;
    if (!in || !in_len || !out || !out_len || stage < 0 || stage > 7) { s5gpu_set_error("s5gpu_solo_batch: bad argument"); return S5GPU_ERR_ARG; }
    s5host::CtxHold hold;
    int rc = hold.acquire();
    if (rc) return rc;
    Ctx *c = hold.c;
    for (uint32_t i = 0; i < n; i++) { out[i] = NULL; out_len[i] = 0; if (status) status[i] = 0; }
    for (uint32_t i = 0; i < n; i++)
        if (in_len[i] > 0xFFFFFF00ull / 4) { s5gpu_set_error("item %u too large", i); return S5GPU_ERR_ARG; }
    int overall = S5GPU_OK;
    const bool park_in = stage == 0 || stage == 5;   // whole buffers through a record press (5: zstd)
    const int park_rec = stage == 5 ? S5GPU_REC_ZSTD : S5GPU_REC_ZLIB;
    const bool exzd_enc = stage == 6, exzd_dec = stage == 7;
    if (park_in || stage == 2 || exzd_enc) {
        // encode side: READ_DESC slots.  ex-zd has no kernel of its own: the blob is what the record builder writes behind the
        // u64 length of a record without head and aux (record press none): slot = [u64 size][u64 L][blob]
        std::vector<s5gpu_read_desc_t> desc(n);
        std::vector<uint32_t> park(n, 0), lens(n);
        uint64_t so = 0, oo = 0;
        for (uint32_t i = 0; i < n; i++) {
            s5gpu_read_desc_t &d = desc[i];
            memset(&d, 0, sizeof d);
            d.out_off = oo;
            if (park_in) {
                const uint32_t len = (uint32_t)in_len[i];
                d.hdr_len = (len > 8 ? len : 8) - 8;
                d.slot_cap = (uint32_t)s5gpu_slot_bound(0, d.hdr_len, 0, park_rec, S5GPU_SIG_NONE);
                park[i] = (d.slot_cap - (d.hdr_len + 8)) & ~15u;
                lens[i] = len;
            } else {
                if (in_len[i] & 1) { s5gpu_set_error("svb-zd input %u is not a whole number of int16 samples", i); return S5GPU_ERR_ARG; }
                d.n_samples = (uint32_t)(in_len[i] / 2);
                d.sig_off = so;
                so += up(d.n_samples, 8);
                d.slot_cap = exzd_enc ? (uint32_t)s5gpu_slot_bound(d.n_samples, 0, 0, S5GPU_REC_NONE, S5GPU_SIG_EX_ZD)
                                      : (uint32_t)up(4ull + (d.n_samples + 3ull) / 4 + 3ull * d.n_samples + 16, 16);
            }
            oo += d.slot_cap;
        }
        const size_t hin = up(park_in ? oo : so * 2, 64) + 64 + sizeof(s5gpu_read_desc_t) * n + 4ull * n;
        if ((rc = c->h_in.reserve(hin)) || (rc = c->d_slots.reserve(oo + 64)) || (rc = c->d_sig.reserve(so * 2 + 64)) ||
            (rc = c->d_desc.reserve(sizeof(s5gpu_read_desc_t) * n)) || (rc = c->d_len.reserve(4ull * n)) ||
            (rc = c->d_hdr.reserve(64)) || (rc = c->h_out.reserve(oo + 64 + 4ull * n)))
            return rc;
        uint8_t *h0 = (uint8_t *)c->h_in.p;
        uint8_t *hd = h0 + up(park_in ? oo : so * 2, 64) + 64, *hl = hd + sizeof(s5gpu_read_desc_t) * n;
        for (uint32_t i = 0; i < n; i++) {
            if (!in_len[i]) continue;
            if (park_in) memcpy(h0 + desc[i].out_off + park[i], in[i], in_len[i]);
            else memcpy(h0 + 2 * desc[i].sig_off, in[i], in_len[i]);
        }
        memcpy(hd, desc.data(), sizeof(s5gpu_read_desc_t) * n);
        memcpy(hl, lens.data(), 4ull * n);
        if (park_in) HIP_TRY(hipMemcpyAsync(c->d_slots.p, h0, oo, hipMemcpyHostToDevice, c->st));
        else if (so) HIP_TRY(hipMemcpyAsync(c->d_sig.p, h0, so * 2, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_read_desc_t) * n, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_len.p, hl, 4ull * n, hipMemcpyHostToDevice, c->st));
        s5gpu_encode_args_t a;
        memset(&a, 0, sizeof a);
        a.n_reads = n;
        a.rec_method = park_in ? park_rec : S5GPU_REC_NONE;
        a.sig_method = park_in ? S5GPU_SIG_NONE : exzd_enc ? S5GPU_SIG_EX_ZD : S5GPU_SIG_SVB_ZD;
        a.desc = (const s5gpu_read_desc_t *)c->d_desc.p;
        a.sig = (const int16_t *)c->d_sig.p; a.hdr = (const uint8_t *)c->d_hdr.p;
        a.slots = (uint8_t *)c->d_slots.p; a.out_len = (uint32_t *)c->d_len.p;
        if ((rc = park_in ? s5gpu_deflate_parked_dev(&a, c->st) : exzd_enc ? s5gpu_encode_dev(&a, c->st) : s5gpu_svbzd_encode_dev(&a, c->st))) return rc;
        uint8_t *ho_len = (uint8_t *)c->h_out.p, *ho_slots = ho_len + up(4ull * n, 64);
        if ((rc = c->h_out.reserve(up(4ull * n, 64) + oo + 64))) return rc;
        ho_len = (uint8_t *)c->h_out.p; ho_slots = ho_len + up(4ull * n, 64);
        HIP_TRY(hipMemcpyAsync(ho_len, c->d_len.p, 4ull * n, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipMemcpyAsync(ho_slots, c->d_slots.p, oo, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        const uint32_t *ol = (const uint32_t *)ho_len;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t skip = park_in ? 8 : exzd_enc ? 16 : 0;   // the solo call returns the bare zlib stream / zstd frame / ex-zd blob
            if (ol[i] < skip || ol[i] > desc[i].slot_cap) { s5gpu_set_error("item %u: impossible device length %u", i, ol[i]); return S5GPU_ERR_HIP; }
            out_len[i] = ol[i] - skip;
            out[i] = malloc(out_len[i] ? out_len[i] : 1);
            if (!out[i]) return S5GPU_ERR_NOMEM;
            memcpy(out[i], ho_slots + desc[i].out_off + skip, out_len[i]);
        }
        return S5GPU_OK;
    }
    // decode side: REC_DESC
    const bool infl = stage == 1 || stage == 4;   // 4: zstd frames
    std::vector<uint32_t> pcap(n), scap(n);
    std::vector<uint8_t> done(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (infl) { const uint64_t g = payload_guess(stage == 4 ? S5GPU_REC_ZSTD : S5GPU_REC_ZLIB, in[i], in_len[i]); pcap[i] = (uint32_t)(g > 0xFFFFFF00ull ? 0xFFFFFF00ull : g); scap[i] = 0; }
        else if (exzd_dec) {
            // the blob goes in as the signal of a record without id and aux (EXZD_HEAD zero bytes of head + u64 L); N sits at blob + 1
            uint64_t ns = 0;
            if (in_len[i] >= 9) memcpy(&ns, (const uint8_t *)in[i] + 1, 8);
            if (ns > 2ull * in_len[i] + 64) ns = 2ull * in_len[i] + 64;   // an impossible count: the kernel reports what it needs
            scap[i] = (uint32_t)ns; pcap[i] = (uint32_t)(in_len[i] + 64);
        } else {
            uint32_t ns = 0;
            if (in_len[i] >= 4) memcpy(&ns, in[i], 4);
            if ((uint64_t)ns > 4ull * in_len[i]) ns = 0;   // impossible count: let the kernel report it
            scap[i] = ns; pcap[i] = 0;
        }
    }
    for (int attempt = 0; attempt < 3; attempt++) {
        std::vector<uint32_t> idx;
        for (uint32_t i = 0; i < n; i++) if (!done[i]) idx.push_back(i);
        if (idx.empty()) break;
        const uint32_t m = (uint32_t)idx.size();
        std::vector<s5gpu_rec_desc_t> desc(m);
        uint64_t io = 0, po = 0, so = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = idx[k];
            s5gpu_rec_desc_t &d = desc[k];
            d.in_off = io; d.pay_off = po; d.sig_off = so;
            const uint32_t EXZD_HEAD = 2 + 4 + 32 + 8;   // u16 id_len 0 | u32 read_group | 4 x f64 | u64 L
            d.in_len = (uint32_t)in_len[i] + (exzd_dec ? EXZD_HEAD : 0u); d.pay_cap = pcap[i]; d.sig_cap = scap[i]; d.reserved = 0;
            io += up((uint64_t)d.in_len + 16, 16); po += up((uint64_t)pcap[i] + 16, 16); so += up((uint64_t)scap[i] + 8, 8);
        }
        const size_t hin = up(io + 64, 64) + sizeof(s5gpu_rec_desc_t) * m;
        const size_t hout = up(po + 64, 64) + up(so * 2 + 64, 64) + sizeof(s5gpu_rec_fields_t) * m;
        if ((rc = c->h_in.reserve(hin)) || (rc = c->d_in.reserve(io + 64)) || (rc = c->d_desc.reserve(sizeof(s5gpu_rec_desc_t) * m)) ||
            (rc = c->d_pay.reserve(po + 64)) || (rc = c->d_sig.reserve(so * 2 + 64)) || (rc = c->d_fields.reserve(sizeof(s5gpu_rec_fields_t) * m)) ||
            (rc = c->h_out.reserve(hout)))
            return rc;
        uint8_t *hi = (uint8_t *)c->h_in.p, *hd = hi + up(io + 64, 64);
        for (uint32_t k = 0; k < m; k++) {
            uint8_t *dst = hi + desc[k].in_off;
            if (exzd_dec) {
                memset(dst, 0, 38);
                const uint64_t L = in_len[idx[k]];
                memcpy(dst + 38, &L, 8);
                dst += 46;
            }
            if (in_len[idx[k]]) memcpy(dst, in[idx[k]], in_len[idx[k]]);
        }
        memcpy(hd, desc.data(), sizeof(s5gpu_rec_desc_t) * m);
        HIP_TRY(hipMemcpyAsync(c->d_in.p, hi, io, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_rec_desc_t) * m, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemsetAsync(c->d_fields.p, 0, sizeof(s5gpu_rec_fields_t) * m, c->st));
        s5gpu_decode_args_t a;
        memset(&a, 0, sizeof a);
        a.n_recs = m; a.rec_method = exzd_dec ? S5GPU_REC_NONE : stage == 4 ? S5GPU_REC_ZSTD : S5GPU_REC_ZLIB; a.sig_method = exzd_dec ? S5GPU_SIG_EX_ZD : S5GPU_SIG_SVB_ZD;
        a.desc = (const s5gpu_rec_desc_t *)c->d_desc.p; a.in = (const uint8_t *)c->d_in.p;
        a.payload = (uint8_t *)c->d_pay.p; a.sig_out = (int16_t *)c->d_sig.p; a.fields = (s5gpu_rec_fields_t *)c->d_fields.p;
        if ((rc = infl ? s5gpu_inflate_dev(&a, c->st) : exzd_dec ? s5gpu_decode_dev(&a, c->st) : s5gpu_svbzd_decode_dev(&a, c->st))) return rc;
        uint8_t *hp = (uint8_t *)c->h_out.p, *hsg = hp + up(po + 64, 64), *hf = hsg + up(so * 2 + 64, 64);
        HIP_TRY(hipMemcpyAsync(hf, c->d_fields.p, sizeof(s5gpu_rec_fields_t) * m, hipMemcpyDeviceToHost, c->st));
        if (infl) HIP_TRY(hipMemcpyAsync(hp, c->d_pay.p, po, hipMemcpyDeviceToHost, c->st));
        else HIP_TRY(hipMemcpyAsync(hsg, c->d_sig.p, so * 2, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        const s5gpu_rec_fields_t *ff = (const s5gpu_rec_fields_t *)hf;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = idx[k];
            const s5gpu_rec_fields_t &f = ff[k];
            if (f.status == 5 && attempt < 2) { pcap[i] = f.payload_len ? f.payload_len : (pcap[i] < 0x10000000u ? 8 * pcap[i] + 65536 : 0xFFFFFF00u); continue; }
            if (f.status == 6 && attempt < 2) { scap[i] = f.n_samples; continue; }
            done[i] = 1;
            if (status) status[i] = f.status;
            if (f.status != 0) { overall = S5GPU_ERR_DATA; continue; }
            out_len[i] = infl ? f.payload_len : 2ull * f.n_samples;
            out[i] = malloc(out_len[i] ? out_len[i] : 1);
            if (!out[i]) return S5GPU_ERR_NOMEM;
            memcpy(out[i], infl ? hp + desc[k].pay_off : hsg + 2 * desc[k].sig_off, out_len[i]);
        }
    }
    if (overall) s5gpu_set_error("s5gpu_solo_batch: at least one input is corrupt (see status[i])");
    return overall;
}


// Decode n host records on the device and leave payloads (c->d_pay) and signals (c->d_sig2) resident; rd[i] says where,
// ff[i] what was found.  Records that overflow their guessed slots are redone once with exact sizes.
// framed: the records sit in ONE host buffer (a chunk of a BLOW5 file as read from disk): it is uploaded as it is, no per-record
// packing; rec[i] then point into it.
struct FramedSrc { const uint8_t *base; size_t bytes; };
static int decode_resident_impl(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                                std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status, const FramedSrc *framed);
int s5host::decode_resident(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                            std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status) {
    return decode_resident_impl(c, n, rec, rec_len, from_rec, from_sig, rd, ff, status, nullptr);
}
int s5host::decode_resident_framed(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                                   std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status, const uint8_t *base, size_t bytes) {
    FramedSrc fs = {base, bytes};
    return decode_resident_impl(c, n, rec, rec_len, from_rec, from_sig, rd, ff, status, &fs);
}
static int decode_resident_impl(Ctx *c, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig,
                                std::vector<s5gpu_rec_desc_t> &rd, std::vector<s5gpu_rec_fields_t> &ff, int32_t *status, const FramedSrc *framed) {
    int rc;
    std::vector<uint32_t> pcap(n), scap(n);
    for (uint32_t i = 0; i < n; i++) {
        if (rec_len[i] > 0xFFFFFF00ull / 8) { s5gpu_set_error("record %u too large", i); return S5GPU_ERR_ARG; }
        { const uint64_t g = payload_guess(from_rec, rec[i], rec_len[i]); pcap[i] = (uint32_t)(g > 0xFFFFFF00ull ? 0xFFFFFF00ull : g); }
        scap[i] = pcap[i];   // a sample takes at least one payload byte in either signal format
    }
    rd.resize(n);
    ff.resize(n);
    for (int attempt = 0;; attempt++) {
        uint64_t io = 0, po = 0, so = 0;
        for (uint32_t i = 0; i < n; i++) {
            s5gpu_rec_desc_t &d = rd[i];
            d.in_off = io; d.pay_off = po; d.sig_off = so;
            d.in_len = (uint32_t)rec_len[i]; d.pay_cap = pcap[i]; d.sig_cap = scap[i]; d.reserved = 0;
            if (framed) d.in_off = (uint64_t)((const uint8_t *)rec[i] - framed->base);
            io += up(rec_len[i] + 16, 16); po += up((uint64_t)pcap[i] + 16, 16); so += up((uint64_t)scap[i] + 8, 8);
        }
        if (framed) io = framed->bytes;
        if ((rc = c->h_in.reserve((framed ? 0 : up(io + 64, 64)) + sizeof(s5gpu_rec_desc_t) * n)) || (rc = c->d_in.reserve(io + 64)) ||
            (rc = c->d_desc2.reserve(sizeof(s5gpu_rec_desc_t) * n)) || (rc = c->d_pay.reserve(po + 64)) ||
            (rc = c->d_sig2.reserve(so * 2 + 64)) || (rc = c->d_fields.reserve(sizeof(s5gpu_rec_fields_t) * n)))
            return rc;
        s5_trace("decode_resident: workspaces reserved");
        uint8_t *hi = (uint8_t *)c->h_in.p, *hd = hi + (framed ? 0 : up(io + 64, 64));
        if (!framed) {
            std::unique_lock<std::mutex> turn(g_pack_turn, std::defer_lock);
        if (g_host_turns) turn.lock();
            parallel_for(n, io, [&](uint32_t lo, uint32_t hi_) {
                for (uint32_t i = lo; i < hi_; i++) memcpy(hi + rd[i].in_off, rec[i], rec_len[i]);
            });
        }
        memcpy(hd, rd.data(), sizeof(s5gpu_rec_desc_t) * n);
        {
            std::unique_lock<std::mutex> turn(g_upload_turn[c->slot & 63], std::defer_lock);
        if (g_host_turns) turn.lock();
            if (!framed || attempt == 0)   // a framed chunk is already on the device when overflowing records are redone
                HIP_TRY(hipMemcpyAsync(c->d_in.p, framed ? framed->base : hi, io, hipMemcpyHostToDevice, c->st));
            HIP_TRY(hipMemcpyAsync(c->d_desc2.p, hd, sizeof(s5gpu_rec_desc_t) * n, hipMemcpyHostToDevice, c->st));
            if (g_host_turns && io >= (4u << 20)) { if ((rc = upload_landed(c))) return rc; }     // (small uploads: not worth a wait)
        }
        HIP_TRY(hipMemsetAsync(c->d_fields.p, 0, sizeof(s5gpu_rec_fields_t) * n, c->st));
        s5gpu_decode_args_t da;
        memset(&da, 0, sizeof da);
        da.n_recs = n; da.rec_method = from_rec; da.sig_method = from_sig;
        da.desc = (const s5gpu_rec_desc_t *)c->d_desc2.p; da.in = (const uint8_t *)c->d_in.p;
        da.payload = (uint8_t *)c->d_pay.p; da.sig_out = (int16_t *)c->d_sig2.p; da.fields = (s5gpu_rec_fields_t *)c->d_fields.p;
        for (uint32_t i = 0; i < n; i++) if (rd[i].pay_cap > da.max_pay_cap) da.max_pay_cap = rd[i].pay_cap;
        s5_trace("decode_resident: uploads enqueued");
        if ((rc = s5gpu_decode_dev(&da, c->st))) return rc;
        s5_trace("decode_resident: kernels enqueued");
        HIP_TRY(hipMemcpyAsync(ff.data(), c->d_fields.p, sizeof(s5gpu_rec_fields_t) * n, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        s5_trace("decode_resident: fields back");
        bool retry = false, bad = false;
        for (uint32_t i = 0; i < n; i++) {
            if (ff[i].status == 5 && attempt < 2) { pcap[i] = ff[i].payload_len ? ff[i].payload_len : (pcap[i] < 0x10000000u ? 8 * pcap[i] + 65536 : 0xFFFFFF00u); if (scap[i] < pcap[i]) scap[i] = pcap[i]; retry = true; }
            else if (ff[i].status == 6 && attempt < 2) { scap[i] = ff[i].n_samples; retry = true; }
            else if (ff[i].status != 0) { bad = true; if (status) status[i] = ff[i].status; }
        }
        if (bad) { s5gpu_set_error("at least one input record is corrupt (see status[i])"); return S5GPU_ERR_DATA; }
        if (!retry) break;   // rare: a record inflated to more than 4x its size; the whole batch is decoded again with exact slots
    }
    return S5GPU_OK;
}

// ---- view / merge worker for a whole batch, device-resident between decode and encode ----
static int recompress_batch_dev(int slot, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len, int32_t *status, s5host::Arena *ar);

static int recompress_batch_any(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                int32_t *status, void **arena) {
    if (arena) *arena = NULL;
    if (n == 0) return S5GPU_OK;
    if (!rec || !rec_len || !out || !out_len) { s5gpu_set_error("s5gpu_recompress_batch: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) { out[i] = NULL; out_len[i] = 0; if (status) status[i] = 0; }
    s5host::Arena *ar = nullptr;
    if (arena) {
        if (s5host::n_devices() == 0) return S5GPU_ERR_NODEV;
        ar = new s5host::Arena;
        ar->generation = s5host_generation;
        ar->pool_gen = s5host::arena_pool_generation();
    }
    const int rc = s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) {
        return recompress_batch_dev(slot, hi - lo, rec + lo, rec_len + lo, from_rec, from_sig, to_rec, to_sig,
                                    new_read_group ? new_read_group + lo : nullptr, drop_aux, out + lo, out_len + lo, status ? status + lo : nullptr, ar);
    });
    if (rc) {
        for (uint32_t i = 0; i < n; i++) { if (!ar) free(out[i]); out[i] = NULL; }
        if (ar) s5gpu_arena_release(ar);
        return rc;
    }
    if (arena) *arena = ar;
    return rc;
}
extern "C" int s5gpu_recompress_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                      int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                      int32_t *status) {
    return recompress_batch_any(n, rec, rec_len, from_rec, from_sig, to_rec, to_sig, new_read_group, drop_aux, out, out_len, status, nullptr);
}
extern "C" int s5gpu_recompress_batch_arena(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                            int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                            int32_t *status, void **arena) {
    if (!arena) { s5gpu_set_error("s5gpu_recompress_batch_arena: NULL arena"); return S5GPU_ERR_ARG; }
    return recompress_batch_any(n, rec, rec_len, from_rec, from_sig, to_rec, to_sig, new_read_group, drop_aux, out, out_len, status, arena);
}

// ---- the batch calls as SUBMIT / WAIT pairs (round 6) ----
// The reference's loop is read K -> work_db -> write K with nothing overlapped (/root/reference/src/view.c:254-300; overlapping them is
// the authors' own to-do, /root/reference/README.md:197), and a synchronous batch call keeps it that way: at K = 4096 a call is one
// H2D -> kernels -> D2H round of 1.7 ms, latency the device spends mostly idle.  A submitted batch runs on a thread of its own and takes one
// of the library's contexts (S5GPU_CONTEXTS, two by default) like any batch call of a caller's thread: two tickets in flight = one batch's
// copies under the other's kernels, and the caller reads batch k + 1 from the file in the meantime.  The buffers named at submit time
// (rec / rec_len / out / out_len / status, new_read_group) belong to the library until the ticket is waited for.  Every ticket must be
// waited for exactly once; s5gpu_batch_wait returns the call's code and puts its message where s5gpu_last_error() of the WAITING thread
// finds it.
namespace {
struct BatchTicket {
    std::function<int(void **)> work;     // runs the batch call; the void ** is where an arena goes (nullptr: malloc form)
    bool want_arena = false, done = false;
    int rc = S5GPU_OK;
    std::string err;
    void *arena = nullptr;
};
// The submitted batches run on a few long-lived threads (as many as there can be contexts to take): a thread's first HIP call sets up the
// runtime's per-thread state, which a thread per ticket would pay for every batch.
struct TicketWorkers {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<BatchTicket *> q;
    std::vector<std::thread> th;
    bool stop = false;
    size_t idle = 0;                      // workers waiting for a ticket
    void run() {
        for (;;) {
            BatchTicket *t;
            {
                std::unique_lock<std::mutex> lk(mu);
                idle++;
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                idle--;
                if (q.empty()) return;
                t = q.front();
                q.pop_front();
            }
            t->rc = t->work(t->want_arena ? &t->arena : nullptr);
            if (t->rc != S5GPU_OK) t->err = s5gpu_last_error();
            {
                std::lock_guard<std::mutex> lk(mu);
                t->done = true;
            }
            cv_done.notify_all();
        }
    }
    bool submit(BatchTicket *t) {
        std::lock_guard<std::mutex> lk(mu);
        if (stop) return false;
        if (th.size() < 4 && idle <= q.size()) {           // one more worker when none is free to take this ticket (four at most: the contexts)
            try { th.emplace_back([this] { run(); }); } catch (...) { if (th.empty()) return false; }
        }
        q.push_back(t);
        cv_work.notify_one();
        return true;
    }
    void wait(BatchTicket *t) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return t->done; });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) if (t.joinable()) t.join();     // (queued tickets are run to the end first: run() only returns on an empty queue)
        std::lock_guard<std::mutex> lk(mu);
        th.clear();
        stop = false;
    }
};
TicketWorkers *g_tickets_p = new TicketWorkers();      // (never destroyed: a process that exits without s5gpu_shutdown must not meet joinable threads in a static destructor)
TicketWorkers &g_tickets = *g_tickets_p;
void *submit_ticket(std::function<int(void **)> work, int want_arena, const char *who) {
    BatchTicket *t = new (std::nothrow) BatchTicket();
    if (!t) { s5gpu_set_error("%s: out of memory", who); return nullptr; }
    t->work = std::move(work);
    t->want_arena = want_arena != 0;
    if (!g_tickets.submit(t)) { delete t; s5gpu_set_error("%s: no thread for the batch", who); return nullptr; }
    return t;
}
}  // namespace
void s5host_stop_ticket_workers() { g_tickets.shutdown(); }      // s5gpu_shutdown: every submitted batch is through before the contexts go
extern "C" void *s5gpu_recompress_batch_submit(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                               int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                               int32_t *status, int want_arena) {
    return submit_ticket([=](void **arena) {
        return recompress_batch_any(n, rec, rec_len, from_rec, from_sig, to_rec, to_sig, new_read_group, drop_aux, out, out_len, status, arena);
    }, want_arena, "s5gpu_recompress_batch_submit");
}
extern "C" void *s5gpu_encode_batch_submit(uint32_t n, const int16_t *const *sig, const uint64_t *n_samples, const void *const *hdr,
                                           const uint32_t *hdr_len, const void *const *aux, const uint32_t *aux_len, int rec_method,
                                           int sig_method, void **out, size_t *out_len, int want_arena) {
    return submit_ticket([=](void **arena) {
        return arena ? s5gpu_encode_batch_arena(n, sig, n_samples, hdr, hdr_len, aux, aux_len, rec_method, sig_method, out, out_len, arena)
                     : s5gpu_encode_batch(n, sig, n_samples, hdr, hdr_len, aux, aux_len, rec_method, sig_method, out, out_len);
    }, want_arena, "s5gpu_encode_batch_submit");
}
extern "C" int s5gpu_batch_wait(void *ticket, void **arena) {
    if (arena) *arena = nullptr;
    if (!ticket) { s5gpu_set_error("s5gpu_batch_wait: NULL ticket"); return S5GPU_ERR_ARG; }
    BatchTicket *t = static_cast<BatchTicket *>(ticket);
    g_tickets.wait(t);
    const int rc = t->rc;
    if (rc != S5GPU_OK) s5gpu_set_error("%s", t->err.c_str());
    if (arena) *arena = t->arena;
    else if (t->arena) s5gpu_arena_release(t->arena);      // (submitted with want_arena, waited for without a place for it: nothing leaks)
    delete t;
    return rc;
}

static int recompress_encode_half(Ctx *c, uint32_t n, const std::vector<s5gpu_rec_desc_t> &rd, const std::vector<s5gpu_rec_fields_t> &ff, int to_rec, int to_sig,
                                  const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len, std::vector<uint64_t> *stream_off, s5host::Arena *ar = nullptr);

static int recompress_batch_dev(int slot, uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, int to_rec,
                                int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len, int32_t *status, s5host::Arena *ar) {
    s5host::CtxHold hold;
    int rc = hold.acquire(slot);
    if (rc) return rc;
    Ctx *c = hold.c;
    std::vector<s5gpu_rec_desc_t> rd;
    std::vector<s5gpu_rec_fields_t> ff;
    if ((rc = s5host::decode_resident(c, n, rec, rec_len, from_rec, from_sig, rd, ff, status))) return rc;
    return recompress_encode_half(c, n, rd, ff, to_rec, to_sig, new_read_group, drop_aux, out, out_len, nullptr, ar);
}

// second half of the worker: encode descriptors straight from the decoded fields (payloads in c->d_pay, signals in c->d_sig2)
static int recompress_encode_half(Ctx *c, uint32_t n, const std::vector<s5gpu_rec_desc_t> &rd, const std::vector<s5gpu_rec_fields_t> &ff, int to_rec, int to_sig,
                                  const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len, std::vector<uint64_t> *stream_off, s5host::Arena *ar) {
    int rc;
    // encode descriptors straight from the decoded fields: heads and aux tails are read out of the decoded payloads
    std::vector<s5gpu_read_desc_t> ed(n);
    uint64_t oo = 0;
    uint32_t max_payload = 0;
    for (uint32_t i = 0; i < n; i++) {
        s5gpu_read_desc_t &d = ed[i];
        const s5gpu_rec_fields_t &f = ff[i];
        d.sig_off = rd[i].sig_off;
        d.hdr_off = rd[i].pay_off;
        d.aux_off = rd[i].pay_off + f.aux_off;
        d.out_off = oo;
        d.n_samples = f.n_samples;
        d.hdr_len = 2 + f.read_id_len + 4 + 32;
        d.aux_len = drop_aux ? 0 : f.aux_len;
        const uint64_t pb = s5gpu_payload_bound(d.n_samples, d.hdr_len, d.aux_len, to_sig);
        const uint64_t sb = s5gpu_slot_bound(d.n_samples, d.hdr_len, d.aux_len, to_rec, to_sig);
        if (pb > 0xFFFFFF00ull) { s5gpu_set_error("read %u: record larger than 4 GiB", i); return S5GPU_ERR_ARG; }
        d.slot_cap = (uint32_t)sb;
        if (pb > max_payload) max_payload = (uint32_t)pb;
        oo += sb;
    }
    if ((rc = c->d_desc.reserve(sizeof(s5gpu_read_desc_t) * n))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_desc.p, ed.data(), sizeof(s5gpu_read_desc_t) * n, hipMemcpyHostToDevice, c->st));
    if (new_read_group) {   // src/merge.c:51
        if ((rc = c->d_patch.reserve(12ull * n + 64))) return rc;
        std::vector<uint64_t> po(n);
        for (uint32_t i = 0; i < n; i++) po[i] = rd[i].pay_off + 2 + ff[i].read_id_len;
        uint64_t *d_po = (uint64_t *)c->d_patch.p;
        uint32_t *d_val = (uint32_t *)(d_po + n);
        HIP_TRY(hipMemcpyAsync(d_po, po.data(), 8ull * n, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(d_val, new_read_group, 4ull * n, hipMemcpyHostToDevice, c->st));
        if ((rc = s5gpu_patch_u32_dev((uint8_t *)c->d_pay.p, d_po, d_val, n, c->st))) return rc;
        HIP_TRY(hipStreamSynchronize(c->st));   // po / new_read_group are pageable host memory
    }
    s5gpu_encode_args_t a;
    memset(&a, 0, sizeof a);
    a.n_reads = n; a.rec_method = to_rec; a.sig_method = to_sig;
    a.desc = (const s5gpu_read_desc_t *)c->d_desc.p;
    a.sig = (const int16_t *)c->d_sig2.p; a.hdr = (const uint8_t *)c->d_pay.p; a.aux = (const uint8_t *)c->d_pay.p;
    a.max_payload = max_payload;
    if (stream_off) return s5host::encode_stream_resident(c, n, ed, a, oo, *stream_off);   // the caller fetches c->d_stream itself
    return encode_and_collect(c, n, ed, a, oo, out, out_len, ar);
}

// ---- the decode half alone on a chunk of framed records: `get --benchmark` (/root/reference/src/get.c:52: slow5_get per id, nothing
// written) and any consumer of signals.  Records framed in one host buffer (as s5gpu_recompress_stream takes them: a file chunk, or the
// records a get batch pread next to each other); the decoded signals come back as ONE contiguous int16 block, sig_off[i] = first sample
// of record i, sig_off[n] = total — one D2H, no malloc per record.  fields[i] as s5gpu_decode_batch.  Too little room: S5GPU_ERR_NOMEM
// and sig_off[0] = samples needed.  Device g of G takes the g-th contiguous share of the records. ----
extern "C" int s5gpu_decode_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int rec_method,
                                   int sig_method, int16_t *sig_out, size_t sig_cap, uint64_t *sig_off, s5gpu_rec_fields_t *fields) {
    if (n == 0) { if (sig_off) sig_off[0] = 0; return S5GPU_OK; }
    if (!chunk || !rec_pos || !rec_len || !sig_out || !sig_off || !fields) { s5gpu_set_error("s5gpu_decode_stream: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++)
        if (rec_pos[i] > chunk_bytes || rec_len[i] > chunk_bytes - rec_pos[i]) { s5gpu_set_error("record %u lies outside the chunk", i); return S5GPU_ERR_ARG; }
    const int G = s5host::n_devices();
    if (G == 0) return S5GPU_ERR_NODEV;
    // a share that waits behind a failing one gives up before it has looked at its records: their statuses must not read as "ok"
    for (uint32_t i = 0; i < n; i++) { memset(&fields[i], 0, sizeof fields[i]); fields[i].status = S5GPU_STATUS_NOT_DECODED; }
    s5host::ShareGather sg(G);
    auto share = [&](int slot, uint32_t lo, uint32_t hi) -> int {
        s5_trace("decode_stream: share starts");
        s5host::CtxHold hold;
        int r = hold.acquire(slot);
        if (r) return sg.fail(r, slot);
        s5_trace("decode_stream: context held");
        Ctx *c = hold.c;
        const uint32_t m = hi - lo;
        uint64_t b0 = UINT64_MAX, e1 = 0;
        for (uint32_t i = lo; i < hi; i++) {
            b0 = b0 < rec_pos[i] ? b0 : rec_pos[i];
            e1 = e1 > rec_pos[i] + rec_len[i] ? e1 : rec_pos[i] + rec_len[i];
        }
        b0 &= ~15ull;
        std::vector<const void *> rec(m);
        std::vector<size_t> len(m);
        for (uint32_t i = 0; i < m; i++) { rec[i] = (const uint8_t *)chunk + rec_pos[lo + i]; len[i] = rec_len[lo + i]; }
        FramedSrc fs = {(const uint8_t *)chunk + b0, (size_t)(e1 - b0)};
        std::vector<s5gpu_rec_desc_t> rd;
        std::vector<s5gpu_rec_fields_t> ff;
        std::vector<int32_t> st(m, 0);
        r = decode_resident_impl(c, m, rec.data(), len.data(), rec_method, sig_method, rd, ff, st.data(), &fs);
        if (r == S5GPU_ERR_DATA) {                     // corrupt records are the caller's to look at: fields[i].status says which
            for (uint32_t i = 0; i < m; i++) { fields[lo + i] = ff[i]; if (st[i]) { fields[lo + i].status = st[i]; fields[lo + i].n_samples = 0; } }
        }
        if (r) return sg.fail(r, slot);
        // compact the signals (each sits in its own guessed slot) into one block on the device, then one D2H
        std::vector<uint64_t> src(m), dst(m), off(m + 1);
        std::vector<uint32_t> gl(m);
        off[0] = 0;
        for (uint32_t i = 0; i < m; i++) {
            src[i] = 2 * rd[i].sig_off; dst[i] = 2 * off[i]; gl[i] = 2 * ff[i].n_samples;
            off[i + 1] = off[i] + ff[i].n_samples;
            fields[lo + i] = ff[i];
        }
        uint64_t base = 0;
        bool copy = false;
        if ((r = sg.place(slot, off[m], sig_cap, &base, &copy))) return r;
        if (!copy) return S5GPU_OK;
        s5_trace("decode_stream: placed");
        const size_t b8 = up(8ull * m, 64);
        if ((r = c->d_gather.reserve(2 * off[m] + 64)) || (r = c->d_patch.reserve(2 * b8 + 4ull * m + 64)) || (r = c->h_in.reserve(2 * b8 + 4ull * m + 64))) return r;
        uint8_t *h = (uint8_t *)c->h_in.p, *dv = (uint8_t *)c->d_patch.p;
        memcpy(h, src.data(), 8ull * m); memcpy(h + b8, dst.data(), 8ull * m); memcpy(h + 2 * b8, gl.data(), 4ull * m);
        HIP_TRY(hipMemcpyAsync(dv, h, 2 * b8 + 4ull * m, hipMemcpyHostToDevice, c->st));
        if ((r = s5gpu_gather_dev(m, (const uint64_t *)dv, (const uint32_t *)(dv + 2 * b8), (const uint64_t *)(dv + b8), (const uint8_t *)c->d_sig2.p,
                                  (uint8_t *)c->d_gather.p, c->st)))
            return r;
        if (off[m]) HIP_TRY(hipMemcpyAsync(sig_out + base, c->d_gather.p, 2 * off[m], hipMemcpyDeviceToHost, c->st));
        for (uint32_t i = 0; i < m; i++) sig_off[lo + i] = base + off[i];
        if (hi == n) sig_off[n] = base + off[m];
        HIP_TRY(hipStreamSynchronize(c->st));
        s5_trace("decode_stream: signals back");
        return S5GPU_OK;
    };
    // any way a share gives up releases the shares waiting behind it (ShareGather::place)
    const int rc = s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) -> int { const int r = share(slot, lo, hi); if (r) sg.fail(r, slot); return r; });
    if (rc) return sg.report(rc);   // the share that failed first, not the lowest slot that noticed
    if (sg.overflow) {
        sig_off[0] = sg.need();
        s5gpu_set_error("s5gpu_decode_stream: signal buffer too small (%llu samples needed)", (unsigned long long)sig_off[0]);
        return S5GPU_ERR_NOMEM;
    }
    return S5GPU_OK;
}

// ---- read ids of the records of a file chunk: what slow5_idx_create needs of a record (csrc/blow5_file.c) ----
// The reference's index builder (slow5_idx_create -> slow5_idx_build, reached from /root/reference/src/index.c and get.c:286) reads every
// record to learn its read_id; decoding the whole record for that — 8 KB of signal for a 36-byte id — is what round 2 did.  Here only the
// FRONT of every zlib record crosses PCIe (the dynamic-Huffman header + a few dozen symbols: 1 KiB covers it), the wave-per-record
// decoder stops after the first 2 + id_pitch bytes (k_inflate_head), and uncompressed records are read on the host.  A record whose front
// was too short, or whose id is longer than id_pitch, reports status != 0: the caller takes the general path for it.
extern "C" int s5gpu_record_ids_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int rec_method,
                                       uint32_t id_pitch, char *ids, uint16_t *id_len, int32_t *status) {
    if (n == 0) return S5GPU_OK;
    if (!chunk || !rec_pos || !rec_len || !ids || !id_len || !status || id_pitch == 0 || id_pitch > 65535) { s5gpu_set_error("s5gpu_record_ids_stream: bad argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++)
        if (rec_pos[i] > chunk_bytes || rec_len[i] > chunk_bytes - rec_pos[i]) { s5gpu_set_error("record %u lies outside the chunk", i); return S5GPU_ERR_ARG; }
    const uint8_t *base = (const uint8_t *)chunk;
    auto take = [&](uint32_t i, const uint8_t *head, uint32_t have) {        // head = the record's first uncompressed bytes
        status[i] = 7; id_len[i] = 0;
        if (have < 2) return;
        const uint32_t l = head[0] | ((uint32_t)head[1] << 8);
        id_len[i] = (uint16_t)l;
        if (l > id_pitch) { status[i] = 5; return; }
        if (2 + l > have) return;
        memcpy(ids + (size_t)i * id_pitch, head + 2, l);
        status[i] = 0;
    };
    if (rec_method == S5GPU_REC_NONE) {
        for (uint32_t i = 0; i < n; i++) take(i, base + rec_pos[i], rec_len[i]);
        return S5GPU_OK;
    }
    if (rec_method != S5GPU_REC_ZLIB) { s5gpu_set_error("s5gpu_record_ids_stream: zlib or uncompressed records (zstd frames take the general decode)"); return S5GPU_ERR_ARG; }
    s5host::CtxHold hold;
    int rc = hold.acquire();
    if (rc) return rc;
    Ctx *c = hold.c;
    const uint32_t FRONT = 1024, hp = (uint32_t)up(2ull + id_pitch, 16);
    std::vector<uint32_t> todo(n);
    for (uint32_t i = 0; i < n; i++) todo[i] = i;
    for (int attempt = 0; attempt < 2 && !todo.empty(); attempt++) {
        const uint32_t m = (uint32_t)todo.size();
        std::vector<s5gpu_rec_desc_t> desc(m);
        uint64_t io = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = todo[k], f = attempt == 0 && rec_len[i] > FRONT ? FRONT : rec_len[i];
            s5gpu_rec_desc_t &d = desc[k];
            memset(&d, 0, sizeof d);
            d.in_off = io; d.in_len = f; d.pay_off = (uint64_t)k * hp; d.pay_cap = 2 + id_pitch;
            io += up((uint64_t)f + 16, 16);
        }
        const size_t hin = up(io + 64, 64) + sizeof(s5gpu_rec_desc_t) * m, hout = (size_t)m * hp + sizeof(s5gpu_rec_fields_t) * m + 64;
        if ((rc = c->h_in.reserve(hin)) || (rc = c->d_in.reserve(io + 64)) || (rc = c->d_desc.reserve(sizeof(s5gpu_rec_desc_t) * m)) ||
            (rc = c->d_pay.reserve((uint64_t)m * hp + 64)) || (rc = c->d_fields.reserve(sizeof(s5gpu_rec_fields_t) * m)) || (rc = c->h_out.reserve(hout)))
            return rc;
        uint8_t *hi = (uint8_t *)c->h_in.p, *hd = hi + up(io + 64, 64);
        parallel_for(m, io, [&](uint32_t lo, uint32_t hi_) {
            for (uint32_t k = lo; k < hi_; k++) memcpy(hi + desc[k].in_off, base + rec_pos[todo[k]], desc[k].in_len);
        });
        memcpy(hd, desc.data(), sizeof(s5gpu_rec_desc_t) * m);
        HIP_TRY(hipMemcpyAsync(c->d_in.p, hi, io, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_rec_desc_t) * m, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemsetAsync(c->d_fields.p, 0, sizeof(s5gpu_rec_fields_t) * m, c->st));
        s5gpu_decode_args_t a;
        memset(&a, 0, sizeof a);
        a.n_recs = m; a.rec_method = S5GPU_REC_ZLIB; a.sig_method = S5GPU_SIG_NONE;
        a.desc = (const s5gpu_rec_desc_t *)c->d_desc.p; a.in = (const uint8_t *)c->d_in.p;
        a.payload = (uint8_t *)c->d_pay.p; a.fields = (s5gpu_rec_fields_t *)c->d_fields.p;
        if ((rc = s5gpu_inflate_head_dev(&a, c->st))) return rc;
        uint8_t *hp_ = (uint8_t *)c->h_out.p, *hf = hp_ + (size_t)m * hp;
        HIP_TRY(hipMemcpyAsync(hp_, c->d_pay.p, (size_t)m * hp, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipMemcpyAsync(hf, c->d_fields.p, sizeof(s5gpu_rec_fields_t) * m, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        const s5gpu_rec_fields_t *ff = (const s5gpu_rec_fields_t *)hf;
        std::vector<uint32_t> again;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t i = todo[k];
            if (ff[k].status == 0) take(i, hp_ + (size_t)k * hp, ff[k].payload_len);
            else { status[i] = ff[k].status; id_len[i] = 0; }
            // the front ended before the id was out (a record with a very long code-length header, or stored blocks): the whole record next
            if (status[i] != 0 && status[i] != 5 && attempt == 0 && rec_len[i] > FRONT) again.push_back(i);
        }
        todo.swap(again);
    }
    return S5GPU_OK;
}

// ---- pinned host memory for callers that stream a file through the library (examples/s5view.c) ----
extern "C" void *s5gpu_host_alloc(size_t bytes) {
    if (s5host::n_devices() == 0) return NULL;
    void *p = NULL;
    if (s5_pinned_alloc(&p, bytes ? bytes : 1, nullptr) != hipSuccess) { s5gpu_set_error("pinned allocation of %zu bytes failed", bytes); return NULL; }
    return p;
}
extern "C" void s5gpu_host_free(void *p) { if (p) (void)s5_pinned_free(p); }

// The view / merge worker on a CHUNK of a BLOW5 file: the records sit framed ([u64 size][bytes]) in one host buffer, exactly as
// read from disk, and the re-encoded records come back as one contiguous stream, exactly as the ordered write loop would emit
// them — no per-record malloc or memcpy on either side (the reader thread of the reference's loop, src/view.c:265-278, spends
// its time in exactly those).  Device g of G takes the g-th contiguous share of the records.
extern "C" int s5gpu_recompress_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int from_rec,
                                       int from_sig, int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux, void *out_buf,
                                       size_t out_cap, uint64_t *out_off, int32_t *status) {
    if (n == 0) { if (out_off) out_off[0] = 0; return S5GPU_OK; }
    if (!chunk || !rec_pos || !rec_len || !out_buf || !out_off) { s5gpu_set_error("s5gpu_recompress_stream: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) {
        if (status) status[i] = 0;
        if (rec_pos[i] > chunk_bytes || rec_len[i] > chunk_bytes - rec_pos[i]) { s5gpu_set_error("record %u lies outside the chunk", i); return S5GPU_ERR_ARG; }
    }
    const int G = s5host::n_devices();
    if (G == 0) return S5GPU_ERR_NODEV;
    s5host::ShareGather sg(G);
    auto share = [&](int slot, uint32_t lo, uint32_t hi) -> int {
        s5host::CtxHold hold;
        int r = hold.acquire(slot);
        if (r) return sg.fail(r, slot);
        Ctx *c = hold.c;
        const uint32_t m = hi - lo;
        // the extent of this share of the chunk (the records of a share are contiguous in a file chunk)
        const uint64_t e0 = rec_pos[lo] & ~15ull;
        uint64_t e1 = 0;
        for (uint32_t i = lo; i < hi; i++) e1 = e1 > rec_pos[i] + rec_len[i] ? e1 : rec_pos[i] + rec_len[i];
        uint64_t b0 = e0;
        for (uint32_t i = lo; i < hi; i++) b0 = b0 < rec_pos[i] ? b0 : rec_pos[i] & ~15ull;
        std::vector<const void *> rec(m);
        std::vector<size_t> len(m);
        for (uint32_t i = 0; i < m; i++) { rec[i] = (const uint8_t *)chunk + rec_pos[lo + i]; len[i] = rec_len[lo + i]; }
        FramedSrc fs = {(const uint8_t *)chunk + b0, (size_t)(e1 - b0)};
        std::vector<s5gpu_rec_desc_t> rd;
        std::vector<s5gpu_rec_fields_t> ff;
        std::vector<uint64_t> off;
        if ((r = decode_resident_impl(c, m, rec.data(), len.data(), from_rec, from_sig, rd, ff, status ? status + lo : nullptr, &fs))) return sg.fail(r, slot);
        if ((r = recompress_encode_half(c, m, rd, ff, to_rec, to_sig, new_read_group ? new_read_group + lo : nullptr, drop_aux, nullptr, nullptr, &off))) return sg.fail(r, slot);
        uint64_t base = 0;
        bool copy = false;
        if ((r = sg.place(slot, off[m], out_cap, &base, &copy))) return r;
        if (!copy) return S5GPU_OK;
        HIP_TRY(hipMemcpyAsync((uint8_t *)out_buf + base, c->d_stream.p, off[m], hipMemcpyDeviceToHost, c->st));
        for (uint32_t i = 0; i < m; i++) out_off[lo + i] = base + off[i];
        if (hi == n) out_off[n] = base + off[m];
        HIP_TRY(hipStreamSynchronize(c->st));
        return S5GPU_OK;
    };
    // any way a share gives up releases the shares waiting behind it (ShareGather::place)
    const int rc = s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) -> int { const int r = share(slot, lo, hi); if (r) sg.fail(r, slot); return r; });
    if (rc) return sg.report(rc);   // the share that failed first, not the lowest slot that noticed
    if (sg.overflow) {
        const uint64_t need = sg.need();
        out_off[0] = need;   // the size the caller has to bring (the sum over ALL shares)
        s5gpu_set_error("s5gpu_recompress_stream: output buffer too small (%llu bytes needed)", (unsigned long long)need);
        return S5GPU_ERR_NOMEM;
    }
    return S5GPU_OK;
}
