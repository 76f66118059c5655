// ascii_api.hip — SLOW5 ASCII records <-> BLOW5 records for a whole batch.
//
// What slow5tools' view worker does per record when one side is a .slow5 file
// (/root/reference/src/view.c:35-57: slow5_rec_depress_parse parses the line, slow5_rec_to_mem prints it), split the way the
// bytes are: the raw_signal column goes through the device kernels of ascii_kernels.hip, the few scalar columns and the aux
// columns are converted here on the host, spread over a few threads.
//
// Text conventions, pinned on the reference's fixture pairs (tests/test_ascii.py):
//   - doubles print as "%f" with trailing zeros (and a bare '.') trimmed — exp/index/example_multi_rg_v0.1.0.{slow5,blow5}:
//     195.77062844206847 <-> "195.770628";
//   - enum columns print their label index, strings / arrays print "." when empty
//     (raw/split/multi_group_enum/with_and_without_enum.slow5).
//   - [RECALLED, no fixture pair] a missing scalar "." is the type's maximum (signed and unsigned ints, enum = 0xFF) or NaN.
#include <errno.h>
#include <math.h>
#include <stdio.h>

#include <atomic>
#include <string>

#include "host_ctx.h"


namespace {

constexpr uint8_t KIND_MASK = 0x0F;
const uint32_t kind_size[12] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 1, 1};

bool name_is(const char *p, size_t n, const char *lit) { return strlen(lit) == n && memcmp(p, lit, n) == 0; }

// one column of the types line -> code, or -1
int type_code(const char *p, size_t n) {
    int arr = 0;
    if (n && p[n - 1] == '*') { arr = S5GPU_AUX_ARRAY; n--; }
    static const char *const names[] = {"int8_t", "int16_t", "int32_t", "int64_t", "uint8_t", "uint16_t", "uint32_t", "uint64_t", "float", "double", "char"};
    for (int k = 0; k < 11; k++)
        if (name_is(p, n, names[k])) return k | arr;
    if (n >= 6 && memcmp(p, "enum{", 5) == 0 && p[n - 1] == '}') return S5GPU_AUX_ENUM | arr;
    return -1;
}

struct Line {
    const char *sig = nullptr;   // raw_signal column
    uint32_t sig_len = 0;
    uint32_t n_samples = 0;
    std::string head;            // u16 id_len | id | u32 rg | 4 x f64
    std::string aux;             // BLOW5 aux bytes
    int status = 0;
};

template <class T>
void put(std::string &s, T v) { s.append((const char *)&v, sizeof v); }

bool is_dot(const char *p, size_t n) { return n == 1 && p[0] == '.'; }

bool parse_u64(const char *p, size_t n, uint64_t max, uint64_t *out) {
    if (n == 0 || n > 20) return false;
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        const unsigned d = (unsigned)(p[i] - '0');
        if (d > 9) return false;
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    if (v > max) return false;
    *out = v;
    return true;
}
bool parse_i64(const char *p, size_t n, int64_t min, int64_t max, int64_t *out) {
    bool neg = n && p[0] == '-';
    if (neg || (n && p[0] == '+')) { p++; n--; }
    uint64_t m;
    if (!parse_u64(p, n, neg ? (uint64_t)INT64_MAX + 1 : (uint64_t)INT64_MAX, &m)) return false;
    const int64_t v = neg ? (int64_t)(0 - m) : (int64_t)m;
    if (v < min || v > max) return false;
    *out = v;
    return true;
}
bool parse_f64(const char *p, size_t n, double *out) {
    char tmp[96];
    if (n == 0 || n >= sizeof tmp) return false;
    memcpy(tmp, p, n);
    tmp[n] = 0;
    char *end;
    errno = 0;
    const double v = strtod(tmp, &end);
    if (end != tmp + n) return false;
    *out = v;
    return true;
}

// one scalar element: text -> bytes appended to `o`
bool elem_from_text(int kind, const char *p, size_t n, bool allow_missing, std::string &o) {
    const bool dot = allow_missing && is_dot(p, n);
    switch (kind) {
    case S5GPU_AUX_INT8: case S5GPU_AUX_INT16: case S5GPU_AUX_INT32: case S5GPU_AUX_INT64: {
        static const int64_t lo[4] = {INT8_MIN, INT16_MIN, INT32_MIN, INT64_MIN}, hi[4] = {INT8_MAX, INT16_MAX, INT32_MAX, INT64_MAX};
        int64_t v = hi[kind];
        if (!dot && !parse_i64(p, n, lo[kind], hi[kind], &v)) return false;
        o.append((const char *)&v, kind_size[kind]);     // little-endian host
        return true;
    }
    case S5GPU_AUX_UINT8: case S5GPU_AUX_UINT16: case S5GPU_AUX_UINT32: case S5GPU_AUX_UINT64: case S5GPU_AUX_ENUM: {
        static const uint64_t hi[4] = {UINT8_MAX, UINT16_MAX, UINT32_MAX, UINT64_MAX};
        const uint64_t mx = kind == S5GPU_AUX_ENUM ? UINT8_MAX : hi[kind - S5GPU_AUX_UINT8];
        uint64_t v = mx;
        if (!dot && !parse_u64(p, n, mx, &v)) return false;
        o.append((const char *)&v, kind_size[kind]);
        return true;
    }
    case S5GPU_AUX_FLOAT: case S5GPU_AUX_DOUBLE: {
        double v = NAN;
        if (!dot && !parse_f64(p, n, &v)) return false;
        if (kind == S5GPU_AUX_FLOAT) put(o, (float)v); else put(o, v);
        return true;
    }
    case S5GPU_AUX_CHAR:
        if (n != 1) return false;
        o.push_back(dot ? 0 : p[0]);
        return true;
    }
    return false;
}

bool aux_from_text(uint8_t type, const char *p, size_t n, std::string &o) {
    const int kind = type & KIND_MASK;
    if (!(type & S5GPU_AUX_ARRAY)) return elem_from_text(kind, p, n, true, o);
    if (is_dot(p, n)) { put(o, (uint64_t)0); return true; }
    if (kind == S5GPU_AUX_CHAR) { put(o, (uint64_t)n); o.append(p, n); return true; }
    const size_t at = o.size();
    put(o, (uint64_t)0);
    uint64_t cnt = 0;
    size_t b = 0;
    while (b <= n) {
        const char *c = (const char *)memchr(p + b, ',', n - b);
        const size_t e = c ? (size_t)(c - p) : n;
        if (!elem_from_text(kind, p + b, e - b, false, o)) return false;
        cnt++;
        b = e + 1;
    }
    memcpy(&o[at], &cnt, 8);
    return true;
}

void parse_line(const char *line, size_t len, uint32_t n_aux, const uint8_t *aux_type, const uint32_t *new_rg, int drop_aux, Line &L) {
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
    const char *f[8];
    size_t fl[8];
    size_t b = 0;
    int k = 0;
    for (; k < 8; k++) {
        const char *t = (const char *)memchr(line + b, '\t', len - b);
        f[k] = line + b;
        fl[k] = t ? (size_t)(t - line) - b : len - b;
        b += fl[k] + 1;
        if (!t) { k++; break; }
    }
    L.status = 16;
    if (k < 8) return;
    uint64_t rg, ns;
    double dv[4];
    if (fl[0] == 0 || fl[0] > 0xFFFF || !parse_u64(f[1], fl[1], UINT32_MAX, &rg) || !parse_u64(f[6], fl[6], 0xFFFFFFF0ull, &ns)) return;
    for (int j = 0; j < 4; j++)
        if (!parse_f64(f[2 + j], fl[2 + j], &dv[j])) return;
    if (new_rg) rg = *new_rg;
    L.head.reserve(2 + fl[0] + 36);
    put(L.head, (uint16_t)fl[0]);
    L.head.append(f[0], fl[0]);
    put(L.head, (uint32_t)rg);
    for (int j = 0; j < 4; j++) put(L.head, dv[j]);
    L.sig = f[7];
    L.sig_len = (uint32_t)fl[7];
    L.n_samples = (uint32_t)ns;
    if (fl[7] > 0xFFFFFF00ull) return;
    // aux columns
    for (uint32_t a = 0; a < n_aux; a++) {
        if (b > len) return;                        // column missing
        const char *t = (const char *)memchr(line + b, '\t', len - b);
        const size_t l = t ? (size_t)(t - line) - b : len - b;
        if (!drop_aux && !aux_from_text(aux_type[a], line + b, l, L.aux)) return;
        b += l + 1;
    }
    if (b <= len) return;                           // more columns than the header declares
    L.status = 0;
}

// ---- formatting ----
void fmt_f64(std::string &o, double v) {
    if (isnan(v)) { o.push_back('.'); return; }
    char tmp[352];
    int n = snprintf(tmp, sizeof tmp, "%f", v);
    if (n < 0) n = 0;
    if (memchr(tmp, '.', (size_t)n)) {
        while (n && tmp[n - 1] == '0') n--;
        if (n && tmp[n - 1] == '.') n--;
    }
    o.append(tmp, (size_t)n);
}
void fmt_u64(std::string &o, uint64_t v) {
    char tmp[24];
    o.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v));
}
void fmt_i64(std::string &o, int64_t v) {
    char tmp[24];
    o.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%lld", (long long)v));
}

bool elem_to_text(int kind, const uint8_t *p, bool allow_missing, std::string &o) {
    switch (kind) {
    case S5GPU_AUX_INT8: { int8_t v; memcpy(&v, p, 1); if (allow_missing && v == INT8_MAX) o.push_back('.'); else fmt_i64(o, v); return true; }
    case S5GPU_AUX_INT16: { int16_t v; memcpy(&v, p, 2); if (allow_missing && v == INT16_MAX) o.push_back('.'); else fmt_i64(o, v); return true; }
    case S5GPU_AUX_INT32: { int32_t v; memcpy(&v, p, 4); if (allow_missing && v == INT32_MAX) o.push_back('.'); else fmt_i64(o, v); return true; }
    case S5GPU_AUX_INT64: { int64_t v; memcpy(&v, p, 8); if (allow_missing && v == INT64_MAX) o.push_back('.'); else fmt_i64(o, v); return true; }
    case S5GPU_AUX_UINT8: case S5GPU_AUX_ENUM: { uint8_t v; memcpy(&v, p, 1); if (allow_missing && v == UINT8_MAX) o.push_back('.'); else fmt_u64(o, v); return true; }
    case S5GPU_AUX_UINT16: { uint16_t v; memcpy(&v, p, 2); if (allow_missing && v == UINT16_MAX) o.push_back('.'); else fmt_u64(o, v); return true; }
    case S5GPU_AUX_UINT32: { uint32_t v; memcpy(&v, p, 4); if (allow_missing && v == UINT32_MAX) o.push_back('.'); else fmt_u64(o, v); return true; }
    case S5GPU_AUX_UINT64: { uint64_t v; memcpy(&v, p, 8); if (allow_missing && v == UINT64_MAX) o.push_back('.'); else fmt_u64(o, v); return true; }
    case S5GPU_AUX_FLOAT: { float v; memcpy(&v, p, 4); fmt_f64(o, (double)v); return true; }
    case S5GPU_AUX_DOUBLE: { double v; memcpy(&v, p, 8); fmt_f64(o, v); return true; }
    case S5GPU_AUX_CHAR: o.push_back(allow_missing && p[0] == 0 ? '.' : (char)p[0]); return true;
    }
    return false;
}

// aux bytes -> "\tcol\tcol..."; false when the bytes do not match the declared types
bool aux_to_text(const uint8_t *p, size_t n, uint32_t n_aux, const uint8_t *aux_type, std::string &o) {
    size_t at = 0;
    for (uint32_t a = 0; a < n_aux; a++) {
        const int kind = aux_type[a] & KIND_MASK;
        if (kind > S5GPU_AUX_ENUM) return false;
        const uint32_t es = kind_size[kind];
        o.push_back('\t');
        if (!(aux_type[a] & S5GPU_AUX_ARRAY)) {
            if (at + es > n) return false;
            elem_to_text(kind, p + at, true, o);
            at += es;
            continue;
        }
        uint64_t cnt;
        if (at + 8 > n) return false;
        memcpy(&cnt, p + at, 8);
        at += 8;
        if (cnt > (n - at) / es) return false;
        if (cnt == 0) { o.push_back('.'); continue; }
        if (kind == S5GPU_AUX_CHAR) { o.append((const char *)p + at, (size_t)cnt); at += cnt; continue; }
        for (uint64_t e = 0; e < cnt; e++) {
            if (e) o.push_back(',');
            elem_to_text(kind, p + at, false, o);
            at += es;
        }
    }
    return at == n;
}

}  // namespace

extern "C" int s5gpu_aux_types_parse(const char *types_line, size_t len, uint8_t *aux_type, uint32_t cap) {
    if (!types_line) { s5gpu_set_error("s5gpu_aux_types_parse: NULL argument"); return S5GPU_ERR_ARG; }
    while (len && (types_line[len - 1] == '\n' || types_line[len - 1] == '\r')) len--;
    size_t b = 0;
    if (len && types_line[0] == '#') b = 1;
    uint32_t col = 0, n_aux = 0;
    while (b <= len) {
        const char *t = (const char *)memchr(types_line + b, '\t', len - b);
        const size_t l = t ? (size_t)(t - types_line) - b : len - b;
        if (col >= 8) {
            const int code = type_code(types_line + b, l);
            if (code < 0) { s5gpu_set_error("aux column %u: unknown type '%.*s'", n_aux, (int)(l > 40 ? 40 : l), types_line + b); return S5GPU_ERR_DATA; }
            if (n_aux >= cap || !aux_type) { s5gpu_set_error("more than %u aux columns", cap); return S5GPU_ERR_ARG; }
            aux_type[n_aux++] = (uint8_t)code;
        }
        col++;
        b += l + 1;
    }
    if (col < 8) { s5gpu_set_error("types line has %u columns, 8 primary ones expected", col); return S5GPU_ERR_DATA; }
    return (int)n_aux;
}

extern "C" int s5gpu_ascii_to_blow5_batch(uint32_t n, const char *const *line, const size_t *line_len, uint32_t n_aux, const uint8_t *aux_type,
                                          int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                          int32_t *status) {
    if (n == 0) return S5GPU_OK;
    if (!line || !line_len || !out || !out_len || (n_aux && !aux_type)) { s5gpu_set_error("s5gpu_ascii_to_blow5_batch: NULL argument"); return S5GPU_ERR_ARG; }
    s5host::CtxHold hold;
    int rc = hold.acquire();
    if (rc) return rc;
    Ctx *c = hold.c;
    for (uint32_t i = 0; i < n; i++) { out[i] = NULL; out_len[i] = 0; if (status) status[i] = 0; }
    std::vector<Line> L(n);
    uint64_t text_bytes = 0;
    for (uint32_t i = 0; i < n; i++) text_bytes += line_len[i];
    parallel_for(n, text_bytes, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) parse_line(line[i], line_len[i], n_aux, aux_type, new_read_group ? new_read_group + i : nullptr, drop_aux, L[i]);
    });
    bool bad = false;
    for (uint32_t i = 0; i < n; i++)
        if (L[i].status) { bad = true; if (status) status[i] = L[i].status; }
    if (bad) { s5gpu_set_error("s5gpu_ascii_to_blow5_batch: at least one line is malformed (see status[i])"); return S5GPU_ERR_DATA; }
    // layout
    std::vector<s5gpu_read_desc_t> desc(n);
    std::vector<s5gpu_txt_desc_t> td(n);
    uint64_t so = 0, ho = 0, ao = 0, oo = 0, to = 0;
    uint32_t max_payload = 0;
    for (uint32_t i = 0; i < n; i++) {
        s5gpu_read_desc_t &d = desc[i];
        d.sig_off = so; d.hdr_off = ho; d.aux_off = ao; d.out_off = oo;
        d.n_samples = L[i].n_samples;
        d.hdr_len = (uint32_t)L[i].head.size();
        d.aux_len = (uint32_t)L[i].aux.size();
        const uint64_t pb = s5gpu_payload_bound(d.n_samples, d.hdr_len, d.aux_len, to_sig);
        const uint64_t sb = s5gpu_slot_bound(d.n_samples, d.hdr_len, d.aux_len, to_rec, to_sig);
        if (pb > 0xFFFFFF00ull) { s5gpu_set_error("read %u: record larger than 4 GiB", i); return S5GPU_ERR_ARG; }
        d.slot_cap = (uint32_t)sb;
        if (pb > max_payload) max_payload = (uint32_t)pb;
        s5gpu_txt_desc_t &t = td[i];
        memset(&t, 0, sizeof t);
        t.txt_off = to; t.sig_off = so; t.txt_len = L[i].sig_len; t.n_samples = d.n_samples;
        so += up(d.n_samples, 8); ho += d.hdr_len; ao += d.aux_len; oo += sb;
        to += up(L[i].sig_len, 16) + 16;
    }
    const size_t h_txt = up(to + 64, 64), h_hdr = up(ho + 64, 64), h_aux = up(ao + 64, 64), h_desc = up(sizeof(s5gpu_read_desc_t) * n, 64),
                 h_td = sizeof(s5gpu_txt_desc_t) * n;
    if ((rc = c->h_in.reserve(h_txt + h_hdr + h_aux + h_desc + h_td)) || (rc = c->d_txt.reserve(to + 64)) || (rc = c->d_sig.reserve(so * 2 + 64)) ||
        (rc = c->d_hdr.reserve(ho + 64)) || (rc = c->d_aux.reserve(ao + 64)) || (rc = c->d_desc.reserve(sizeof(s5gpu_read_desc_t) * n)) ||
        (rc = c->d_tdesc.reserve(h_td + 4ull * n)) || (rc = c->h_out.reserve(4ull * n + 64)))
        return rc;
    uint8_t *ht = (uint8_t *)c->h_in.p, *hh = ht + h_txt, *ha = hh + h_hdr, *hd = ha + h_aux, *htd = hd + h_desc;
    parallel_for(n, to, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) {
            if (L[i].sig_len) memcpy(ht + td[i].txt_off, L[i].sig, L[i].sig_len);
            memcpy(hh + desc[i].hdr_off, L[i].head.data(), desc[i].hdr_len);
            if (desc[i].aux_len) memcpy(ha + desc[i].aux_off, L[i].aux.data(), desc[i].aux_len);
        }
    });
    memcpy(hd, desc.data(), sizeof(s5gpu_read_desc_t) * n);
    memcpy(htd, td.data(), h_td);
    HIP_TRY(hipMemcpyAsync(c->d_txt.p, ht, to, hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(c->d_hdr.p, hh, ho, hipMemcpyHostToDevice, c->st));
    if (ao) HIP_TRY(hipMemcpyAsync(c->d_aux.p, ha, ao, hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_read_desc_t) * n, hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(c->d_tdesc.p, htd, h_td, hipMemcpyHostToDevice, c->st));
    int32_t *d_status = (int32_t *)((uint8_t *)c->d_tdesc.p + h_td);
    if ((rc = s5gpu_ascii_parse_dev(n, (const s5gpu_txt_desc_t *)c->d_tdesc.p, (const uint8_t *)c->d_txt.p, (int16_t *)c->d_sig.p, d_status, c->st)))
        return rc;
    HIP_TRY(hipMemcpyAsync(c->h_out.p, d_status, 4ull * n, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    const int32_t *hs = (const int32_t *)c->h_out.p;
    for (uint32_t i = 0; i < n; i++)
        if (hs[i]) { bad = true; if (status) status[i] = hs[i]; }
    if (bad) { s5gpu_set_error("s5gpu_ascii_to_blow5_batch: raw_signal text of at least one line is malformed (see status[i])"); return S5GPU_ERR_DATA; }
    s5gpu_encode_args_t a;
    memset(&a, 0, sizeof a);
    a.n_reads = n; a.rec_method = to_rec; a.sig_method = to_sig;
    a.desc = (const s5gpu_read_desc_t *)c->d_desc.p;
    a.sig = (const int16_t *)c->d_sig.p; a.hdr = (const uint8_t *)c->d_hdr.p; a.aux = (const uint8_t *)c->d_aux.p;
    a.max_payload = max_payload;
    return s5host::encode_and_collect(c, n, desc, a, oo, out, out_len);
}

// The same for a CHUNK of a .slow5 file: lines framed in place, the chunk uploaded as it is, one contiguous record stream back
// (the ASCII twin of s5gpu_recompress_stream; /root/reference/src/view.c:35-57 with a .slow5 input, configs[0] of BASELINE.json).
extern "C" int s5gpu_ascii_to_blow5_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *line_pos, const uint32_t *line_len,
                                           uint32_t n_aux, const uint8_t *aux_type, int to_rec, int to_sig, const uint32_t *new_read_group, int drop_aux,
                                           void *out_buf, size_t out_cap, uint64_t *out_off, int32_t *status) {
    if (n == 0) { if (out_off) out_off[0] = 0; return S5GPU_OK; }
    if (!chunk || !line_pos || !line_len || !out_buf || !out_off || (n_aux && !aux_type)) { s5gpu_set_error("s5gpu_ascii_to_blow5_stream: NULL argument"); return S5GPU_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) {
        if (status) status[i] = 0;
        if (line_pos[i] > chunk_bytes || line_len[i] > chunk_bytes - line_pos[i]) { s5gpu_set_error("line %u lies outside the chunk", i); return S5GPU_ERR_ARG; }
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
        const char *base_p = (const char *)chunk;
        s5_trace("ascii_to_blow5_stream: context acquired");
        // scalar and aux columns on the host; the raw_signal column stays where it is
        std::vector<Line> L(m);
        uint64_t text_bytes = 0, b0 = UINT64_MAX, e1 = 0;
        for (uint32_t i = lo; i < hi; i++) {
            text_bytes += line_len[i];
            b0 = b0 < line_pos[i] ? b0 : line_pos[i];
            e1 = e1 > line_pos[i] + line_len[i] ? e1 : line_pos[i] + line_len[i];
        }
        b0 &= ~15ull;
        parallel_for(m, text_bytes, [&](uint32_t a, uint32_t b) {
            for (uint32_t i = a; i < b; i++)
                parse_line(base_p + line_pos[lo + i], line_len[lo + i], n_aux, aux_type, new_read_group ? new_read_group + lo + i : nullptr, drop_aux, L[i]);
        });
        bool bad = false;
        for (uint32_t i = 0; i < m; i++)
            if (L[i].status) { bad = true; if (status) status[lo + i] = L[i].status; }
        if (bad) { s5gpu_set_error("s5gpu_ascii_to_blow5_stream: at least one line is malformed (see status[i])"); return sg.fail(S5GPU_ERR_DATA, slot); }
        s5_trace("scalar / aux columns parsed on the host");
        std::vector<s5gpu_read_desc_t> desc(m);
        std::vector<s5gpu_txt_desc_t> td(m);
        uint64_t so = 0, ho = 0, ao = 0, oo = 0;
        uint32_t max_payload = 0;
        for (uint32_t i = 0; i < m; i++) {
            s5gpu_read_desc_t &d = desc[i];
            d.sig_off = so; d.hdr_off = ho; d.aux_off = ao; d.out_off = oo;
            d.n_samples = L[i].n_samples;
            d.hdr_len = (uint32_t)L[i].head.size();
            d.aux_len = (uint32_t)L[i].aux.size();
            const uint64_t pb = s5gpu_payload_bound(d.n_samples, d.hdr_len, d.aux_len, to_sig);
            const uint64_t sb = s5gpu_slot_bound(d.n_samples, d.hdr_len, d.aux_len, to_rec, to_sig);
            if (pb > 0xFFFFFF00ull) { s5gpu_set_error("read %u: record larger than 4 GiB", lo + i); return sg.fail(S5GPU_ERR_ARG, slot); }
            d.slot_cap = (uint32_t)sb;
            if (pb > max_payload) max_payload = (uint32_t)pb;
            s5gpu_txt_desc_t &t = td[i];
            memset(&t, 0, sizeof t);
            t.txt_off = (uint64_t)(L[i].sig - base_p) - b0; t.sig_off = so; t.txt_len = L[i].sig_len; t.n_samples = d.n_samples;
            so += up(d.n_samples, 8); ho += d.hdr_len; ao += d.aux_len; oo += sb;
        }
        const size_t h_hdr = up(ho + 64, 64), h_aux = up(ao + 64, 64), h_desc = up(sizeof(s5gpu_read_desc_t) * m, 64), h_td = sizeof(s5gpu_txt_desc_t) * m;
        const uint64_t tbytes = e1 - b0;
        if ((r = c->h_in.reserve(h_hdr + h_aux + h_desc + h_td)) || (r = c->d_txt.reserve(tbytes + 64)) || (r = c->d_sig.reserve(so * 2 + 64)) ||
            (r = c->d_hdr.reserve(ho + 64)) || (r = c->d_aux.reserve(ao + 64)) || (r = c->d_desc.reserve(sizeof(s5gpu_read_desc_t) * m)) ||
            (r = c->d_tdesc.reserve(h_td + 4ull * m)) || (r = c->h_out.reserve(4ull * m + 64)))
            return sg.fail(r, slot);
        s5_trace("workspaces reserved");
        uint8_t *hh = (uint8_t *)c->h_in.p, *ha = hh + h_hdr, *hd = ha + h_aux, *htd = hd + h_desc;
        for (uint32_t i = 0; i < m; i++) {
            memcpy(hh + desc[i].hdr_off, L[i].head.data(), desc[i].hdr_len);
            if (desc[i].aux_len) memcpy(ha + desc[i].aux_off, L[i].aux.data(), desc[i].aux_len);
        }
        memcpy(hd, desc.data(), sizeof(s5gpu_read_desc_t) * m);
        memcpy(htd, td.data(), h_td);
        auto hip = [&](hipError_t e, const char *what) -> int {
            if (e == hipSuccess) return 0;
            s5gpu_set_error("%s failed: %s", what, hipGetErrorString(e));
            return sg.fail(S5GPU_ERR_HIP, slot);
        };
        if ((r = hip(hipMemcpyAsync(c->d_txt.p, (const uint8_t *)chunk + b0, tbytes, hipMemcpyHostToDevice, c->st), "upload of the chunk"))) return r;
        if ((r = hip(hipMemcpyAsync(c->d_hdr.p, hh, ho, hipMemcpyHostToDevice, c->st), "upload"))) return r;
        if (ao && (r = hip(hipMemcpyAsync(c->d_aux.p, ha, ao, hipMemcpyHostToDevice, c->st), "upload"))) return r;
        if ((r = hip(hipMemcpyAsync(c->d_desc.p, hd, sizeof(s5gpu_read_desc_t) * m, hipMemcpyHostToDevice, c->st), "upload"))) return r;
        if ((r = hip(hipMemcpyAsync(c->d_tdesc.p, htd, h_td, hipMemcpyHostToDevice, c->st), "upload"))) return r;
        int32_t *d_status = (int32_t *)((uint8_t *)c->d_tdesc.p + h_td);
        if ((r = s5gpu_ascii_parse_dev(m, (const s5gpu_txt_desc_t *)c->d_tdesc.p, (const uint8_t *)c->d_txt.p, (int16_t *)c->d_sig.p, d_status, c->st))) return sg.fail(r, slot);
        if ((r = hip(hipMemcpyAsync(c->h_out.p, d_status, 4ull * m, hipMemcpyDeviceToHost, c->st), "status download"))) return r;
        if ((r = hip(hipStreamSynchronize(c->st), "synchronise"))) return r;
        s5_trace("chunk uploaded, raw_signal text parsed on the device");
        const int32_t *hs = (const int32_t *)c->h_out.p;
        for (uint32_t i = 0; i < m; i++)
            if (hs[i]) { bad = true; if (status) status[lo + i] = hs[i]; }
        if (bad) { s5gpu_set_error("s5gpu_ascii_to_blow5_stream: raw_signal text of at least one line is malformed (see status[i])"); return sg.fail(S5GPU_ERR_DATA, slot); }
        s5gpu_encode_args_t a;
        memset(&a, 0, sizeof a);
        a.n_reads = m; a.rec_method = to_rec; a.sig_method = to_sig;
        a.desc = (const s5gpu_read_desc_t *)c->d_desc.p;
        a.sig = (const int16_t *)c->d_sig.p; a.hdr = (const uint8_t *)c->d_hdr.p; a.aux = (const uint8_t *)c->d_aux.p;
        a.max_payload = max_payload;
        std::vector<uint64_t> off;
        if ((r = s5host::encode_stream_resident(c, m, desc, a, oo, off))) return sg.fail(r, slot);
        s5_trace("records encoded");
        uint64_t base = 0;
        bool copy = false;
        if ((r = sg.place(slot, off[m], out_cap, &base, &copy))) return r;
        if (!copy) return S5GPU_OK;
        HIP_TRY(hipMemcpyAsync((uint8_t *)out_buf + base, c->d_stream.p, off[m], hipMemcpyDeviceToHost, c->st));
        for (uint32_t i = 0; i < m; i++) out_off[lo + i] = base + off[i];
        if (hi == n) out_off[n] = base + off[m];
        HIP_TRY(hipStreamSynchronize(c->st));
        s5_trace("record stream downloaded");
        return S5GPU_OK;
    };
    // any way a share gives up releases the shares waiting behind it (ShareGather::place)
    const int rc = s5host::for_each_device_range(n, [&](int slot, uint32_t lo, uint32_t hi) -> int { const int r = share(slot, lo, hi); if (r) sg.fail(r, slot); return r; });
    if (rc) return sg.report(rc);   // the share that failed first, not the lowest slot that noticed
    if (sg.overflow) {
        const uint64_t need = sg.need();
        out_off[0] = need;
        s5gpu_set_error("s5gpu_ascii_to_blow5_stream: output buffer too small (%llu bytes needed)", (unsigned long long)need);
        return S5GPU_ERR_NOMEM;
    }
    return S5GPU_OK;
}

extern "C" int s5gpu_blow5_to_ascii_batch(uint32_t n, const void *const *rec, const size_t *rec_len, int from_rec, int from_sig, uint32_t n_aux,
                                          const uint8_t *aux_type, const uint32_t *new_read_group, int drop_aux, void **out, size_t *out_len,
                                          int32_t *status) {
    if (n == 0) return S5GPU_OK;
    if (!rec || !rec_len || !out || !out_len || (n_aux && !aux_type)) { s5gpu_set_error("s5gpu_blow5_to_ascii_batch: NULL argument"); return S5GPU_ERR_ARG; }
    s5host::CtxHold hold;
    int rc = hold.acquire();
    if (rc) return rc;
    Ctx *c = hold.c;
    for (uint32_t i = 0; i < n; i++) { out[i] = NULL; out_len[i] = 0; if (status) status[i] = 0; }
    std::vector<s5gpu_rec_desc_t> rd;
    std::vector<s5gpu_rec_fields_t> ff;
    if ((rc = s5host::decode_resident(c, n, rec, rec_len, from_rec, from_sig, rd, ff, status))) return rc;
    // text slots (worst case 7 bytes / sample) + the read_id and aux ranges the host needs
    std::vector<s5gpu_read_desc_t> td_slots(n);      // reused by the compaction: out_off / slot_cap
    std::vector<s5gpu_txt_desc_t> td(n);
    std::vector<uint64_t> g_src(2ull * n), g_dst(2ull * n);
    std::vector<uint32_t> g_len(2ull * n);
    uint64_t to = 0, go = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t cap = up(7ull * ff[i].n_samples + 16, 16);
        if (cap > 0xFFFFFF00ull) { s5gpu_set_error("read %u: signal text larger than 4 GiB", i); return S5GPU_ERR_ARG; }
        s5gpu_txt_desc_t &t = td[i];
        memset(&t, 0, sizeof t);
        t.txt_off = to; t.sig_off = rd[i].sig_off; t.txt_len = (uint32_t)cap; t.n_samples = ff[i].n_samples;
        memset(&td_slots[i], 0, sizeof td_slots[i]);
        td_slots[i].out_off = to; td_slots[i].slot_cap = (uint32_t)cap;
        to += cap;
        g_src[2 * i] = rd[i].pay_off + 2; g_len[2 * i] = ff[i].read_id_len; g_dst[2 * i] = go; go += ff[i].read_id_len;
        const uint32_t al = drop_aux ? 0 : ff[i].aux_len;
        g_src[2 * i + 1] = rd[i].pay_off + ff[i].aux_off; g_len[2 * i + 1] = al; g_dst[2 * i + 1] = go; go += al;
    }
    const size_t b_td = up(sizeof(s5gpu_txt_desc_t) * n, 64), b_rd = up(sizeof(s5gpu_read_desc_t) * n, 64), b_g8 = up(16ull * n, 64), b_g4 = up(8ull * n, 64);
    // d_tdesc: txt desc | slot desc | gather src | gather dst | gather len | txt_len (u32 n) | status (i32 n)
    const size_t b_all = b_td + b_rd + 2 * b_g8 + b_g4 + up(4ull * n, 64) * 2;
    if ((rc = c->d_tdesc.reserve(b_all)) || (rc = c->d_txt.reserve(to + 64)) || (rc = c->d_gather.reserve(go + 64)) ||
        (rc = c->h_in.reserve(b_all)) || (rc = c->d_scan.reserve(8ull * (n + 1) + 8ull * (n / 1024 + 8))))
        return rc;
    uint8_t *h = (uint8_t *)c->h_in.p, *dv = (uint8_t *)c->d_tdesc.p;
    const size_t o_rd = b_td, o_src = o_rd + b_rd, o_dst = o_src + b_g8, o_len = o_dst + b_g8, o_tl = o_len + b_g4, o_st = o_tl + up(4ull * n, 64);
    memcpy(h, td.data(), sizeof(s5gpu_txt_desc_t) * n);
    memcpy(h + o_rd, td_slots.data(), sizeof(s5gpu_read_desc_t) * n);
    memcpy(h + o_src, g_src.data(), 16ull * n);
    memcpy(h + o_dst, g_dst.data(), 16ull * n);
    memcpy(h + o_len, g_len.data(), 8ull * n);
    HIP_TRY(hipMemcpyAsync(dv, h, o_tl, hipMemcpyHostToDevice, c->st));
    uint32_t *d_tl = (uint32_t *)(dv + o_tl);
    int32_t *d_st = (int32_t *)(dv + o_st);
    if ((rc = s5gpu_ascii_format_dev(n, (const s5gpu_txt_desc_t *)dv, (const int16_t *)c->d_sig2.p, (uint8_t *)c->d_txt.p, d_tl, d_st, c->st)))
        return rc;
    if ((rc = s5gpu_gather_dev(2 * n, (const uint64_t *)(dv + o_src), (const uint32_t *)(dv + o_len), (const uint64_t *)(dv + o_dst),
                               (const uint8_t *)c->d_pay.p, (uint8_t *)c->d_gather.p, c->st)))
        return rc;
    // compact the worst-case text slots into one stream, then one D2H
    std::vector<uint32_t> tl(n);
    std::vector<int32_t> fs(n);
    HIP_TRY(hipMemcpyAsync(tl.data(), d_tl, 4ull * n, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipMemcpyAsync(fs.data(), d_st, 4ull * n, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    std::vector<uint64_t> off(n + 1);
    off[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (fs[i] || tl[i] > td[i].txt_len) { s5gpu_set_error("read %u: signal formatting failed (status %d)", i, fs[i]); return S5GPU_ERR_HIP; }
        off[i + 1] = off[i] + tl[i];
    }
    const uint64_t produced = off[n];
    if ((rc = c->d_stream.reserve(produced + 64)) || (rc = c->h_out.reserve(up(produced + 64, 64) + go + 64))) return rc;
    uint64_t *d_off = (uint64_t *)c->d_scan.p, *d_tmp = d_off + (n + 1);
    if ((rc = s5gpu_compact_dev(n, (const s5gpu_read_desc_t *)(dv + o_rd), (const uint8_t *)c->d_txt.p, d_tl, d_off, (uint8_t *)c->d_stream.p, d_tmp, c->st)))
        return rc;
    uint8_t *h_text = (uint8_t *)c->h_out.p, *h_g = h_text + up(produced + 64, 64);
    if (produced) HIP_TRY(hipMemcpyAsync(h_text, c->d_stream.p, produced, hipMemcpyDeviceToHost, c->st));
    if (go) HIP_TRY(hipMemcpyAsync(h_g, c->d_gather.p, go, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    int fail = 0;
    parallel_for(n, produced, [&](uint32_t lo, uint32_t hi) {
        std::string s;
        for (uint32_t i = lo; i < hi; i++) {
            const s5gpu_rec_fields_t &f = ff[i];
            s.clear();
            s.reserve(tl[i] + 256);
            s.append((const char *)h_g + g_dst[2 * i], f.read_id_len);
            s.push_back('\t'); fmt_u64(s, new_read_group ? new_read_group[i] : f.read_group);
            s.push_back('\t'); fmt_f64(s, f.digitisation);
            s.push_back('\t'); fmt_f64(s, f.offset);
            s.push_back('\t'); fmt_f64(s, f.range);
            s.push_back('\t'); fmt_f64(s, f.sampling_rate);
            s.push_back('\t'); fmt_u64(s, f.n_samples);
            s.push_back('\t'); s.append((const char *)h_text + off[i], tl[i]);
            if (!drop_aux && n_aux && !aux_to_text(h_g + g_dst[2 * i + 1], g_len[2 * i + 1], n_aux, aux_type, s)) {
                if (status) status[i] = 16;
                fail = 1;
                continue;
            }
            s.push_back('\n');
            void *b = malloc(s.size());
            if (!b) { fail = 2; continue; }
            memcpy(b, s.data(), s.size());
            out[i] = b;
            out_len[i] = s.size();
        }
    });
    if (fail) {
        for (uint32_t j = 0; j < n; j++) { free(out[j]); out[j] = NULL; out_len[j] = 0; }
        if (fail == 2) return S5GPU_ERR_NOMEM;
        s5gpu_set_error("s5gpu_blow5_to_ascii_batch: aux bytes of at least one record do not match the header's aux types");
        return S5GPU_ERR_DATA;
    }
    return S5GPU_OK;
}

// The same for a CHUNK of a BLOW5 file -> SLOW5 text lines as ONE contiguous block (the chunk twin of the call above, and the reverse of
// s5gpu_ascii_to_blow5_stream; /root/reference/src/view.c:35-57 with a .slow5 output): records framed in one host buffer exactly as read
// from disk; decode device-resident, raw_signal printed on the device into worst-case slots; the few scalar columns and the aux columns
// are printed on the host from the decoded fields (ids and aux bytes gathered on the device, one small D2H) into a prefix
// ("id \t rg \t ... \t n \t") and a suffix ("\t aux ... \n") per line; the device then puts prefix | signal text | suffix of every
// line at its place in the output block (offsets = prefix sums of the three lengths), and ONE D2H brings the block back.
extern "C" int s5gpu_blow5_to_ascii_stream(uint32_t n, const void *chunk, size_t chunk_bytes, const uint64_t *rec_pos, const uint32_t *rec_len, int from_rec,
                                           int from_sig, uint32_t n_aux, const uint8_t *aux_type, const uint32_t *new_read_group, int drop_aux,
                                           void *out_buf, size_t out_cap, uint64_t *out_off, int32_t *status) {
    if (n == 0) { if (out_off) out_off[0] = 0; return S5GPU_OK; }
    if (!chunk || !rec_pos || !rec_len || !out_buf || !out_off || (n_aux && !aux_type)) { s5gpu_set_error("s5gpu_blow5_to_ascii_stream: NULL argument"); return S5GPU_ERR_ARG; }
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
        uint64_t b0 = UINT64_MAX, e1 = 0;
        for (uint32_t i = lo; i < hi; i++) {
            b0 = b0 < rec_pos[i] ? b0 : rec_pos[i];
            e1 = e1 > rec_pos[i] + rec_len[i] ? e1 : rec_pos[i] + rec_len[i];
        }
        b0 &= ~15ull;
        std::vector<const void *> rec(m);
        std::vector<size_t> len(m);
        for (uint32_t i = 0; i < m; i++) { rec[i] = (const uint8_t *)chunk + rec_pos[lo + i]; len[i] = rec_len[lo + i]; }
        std::vector<s5gpu_rec_desc_t> rd;
        std::vector<s5gpu_rec_fields_t> ff;
        if ((r = s5host::decode_resident_framed(c, m, rec.data(), len.data(), from_rec, from_sig, rd, ff, status ? status + lo : nullptr,
                                                (const uint8_t *)chunk + b0, (size_t)(e1 - b0))))
            return sg.fail(r, slot);
        auto hip = [&](hipError_t e, const char *what) -> int {
            if (e == hipSuccess) return 0;
            s5gpu_set_error("%s failed: %s", what, hipGetErrorString(e));
            return sg.fail(S5GPU_ERR_HIP, slot);
        };
        // 1. signal text into worst-case slots; ids and aux bytes gathered for the host
        std::vector<s5gpu_read_desc_t> slots(m);
        std::vector<s5gpu_txt_desc_t> td(m);
        std::vector<uint64_t> g_src(2ull * m), g_dst(2ull * m);
        std::vector<uint32_t> g_len(2ull * m);
        uint64_t to = 0, go = 0;
        for (uint32_t i = 0; i < m; i++) {
            const uint64_t cap = up(7ull * ff[i].n_samples + 16, 16);
            if (cap > 0xFFFFFF00ull) { s5gpu_set_error("read %u: signal text larger than 4 GiB", lo + i); return sg.fail(S5GPU_ERR_ARG, slot); }
            memset(&td[i], 0, sizeof td[i]);
            td[i].txt_off = to; td[i].sig_off = rd[i].sig_off; td[i].txt_len = (uint32_t)cap; td[i].n_samples = ff[i].n_samples;
            memset(&slots[i], 0, sizeof slots[i]);
            slots[i].out_off = to; slots[i].slot_cap = (uint32_t)cap;
            to += cap;
            g_src[2 * i] = rd[i].pay_off + 2; g_len[2 * i] = ff[i].read_id_len; g_dst[2 * i] = go; go += ff[i].read_id_len;
            const uint32_t al = drop_aux ? 0 : ff[i].aux_len;
            g_src[2 * i + 1] = rd[i].pay_off + ff[i].aux_off; g_len[2 * i + 1] = al; g_dst[2 * i + 1] = go; go += al;
        }
        const size_t b_td = up(sizeof(s5gpu_txt_desc_t) * m, 64), b_rd = up(sizeof(s5gpu_read_desc_t) * m, 64), b_g8 = up(16ull * m, 64), b_g4 = up(8ull * m, 64), b_4 = up(4ull * m, 64);
        // d_tdesc: txt desc | slot desc | gather src | gather dst | gather len | txt_len | status | (stage 2) piece src | piece dst | piece len | sig dst
        const size_t o_rd = b_td, o_src = o_rd + b_rd, o_dst = o_src + b_g8, o_len = o_dst + b_g8, o_tl = o_len + b_g4, o_st = o_tl + b_4, o_p = o_st + b_4;
        const size_t b_all = o_p + 2 * b_g8 + b_g4 + up(8ull * m, 64);
        if ((r = c->d_tdesc.reserve(b_all)) || (r = c->d_txt.reserve(to + 64)) || (r = c->d_gather.reserve(go + 64)) || (r = c->h_in.reserve(b_all)) ||
            (r = c->h_out.reserve(go + 64)))
            return sg.fail(r, slot);
        uint8_t *h = (uint8_t *)c->h_in.p, *dv = (uint8_t *)c->d_tdesc.p;
        memcpy(h, td.data(), sizeof(s5gpu_txt_desc_t) * m);
        memcpy(h + o_rd, slots.data(), sizeof(s5gpu_read_desc_t) * m);
        memcpy(h + o_src, g_src.data(), 16ull * m);
        memcpy(h + o_dst, g_dst.data(), 16ull * m);
        memcpy(h + o_len, g_len.data(), 8ull * m);
        if ((r = hip(hipMemcpyAsync(dv, h, o_tl, hipMemcpyHostToDevice, c->st), "upload"))) return r;
        uint32_t *d_tl = (uint32_t *)(dv + o_tl);
        int32_t *d_st = (int32_t *)(dv + o_st);
        if ((r = s5gpu_ascii_format_dev(m, (const s5gpu_txt_desc_t *)dv, (const int16_t *)c->d_sig2.p, (uint8_t *)c->d_txt.p, d_tl, d_st, c->st))) return sg.fail(r, slot);
        if ((r = s5gpu_gather_dev(2 * m, (const uint64_t *)(dv + o_src), (const uint32_t *)(dv + o_len), (const uint64_t *)(dv + o_dst),
                                  (const uint8_t *)c->d_pay.p, (uint8_t *)c->d_gather.p, c->st)))
            return sg.fail(r, slot);
        std::vector<uint32_t> tl(m);
        std::vector<int32_t> fs(m);
        uint8_t *h_g = (uint8_t *)c->h_out.p;
        if ((r = hip(hipMemcpyAsync(tl.data(), d_tl, 4ull * m, hipMemcpyDeviceToHost, c->st), "download"))) return r;
        if ((r = hip(hipMemcpyAsync(fs.data(), d_st, 4ull * m, hipMemcpyDeviceToHost, c->st), "download"))) return r;
        if (go && (r = hip(hipMemcpyAsync(h_g, c->d_gather.p, go, hipMemcpyDeviceToHost, c->st), "download"))) return r;
        if ((r = hip(hipStreamSynchronize(c->st), "synchronise"))) return r;
        // 2. the other columns on the host: prefix and suffix of every line, back to back in one blob
        std::vector<std::string> pre(m), suf(m);
        std::atomic<int> fail{0};   // written by the parallel_for workers
        parallel_for(m, to / 4, [&](uint32_t a, uint32_t b) {
            for (uint32_t i = a; i < b; i++) {
                const s5gpu_rec_fields_t &f = ff[i];
                if (fs[i] || tl[i] > td[i].txt_len) { fail = 1; continue; }
                std::string &s = pre[i];
                s.reserve(f.read_id_len + 96);
                s.append((const char *)h_g + g_dst[2 * i], f.read_id_len);
                s.push_back('\t'); fmt_u64(s, new_read_group ? new_read_group[lo + i] : f.read_group);
                s.push_back('\t'); fmt_f64(s, f.digitisation);
                s.push_back('\t'); fmt_f64(s, f.offset);
                s.push_back('\t'); fmt_f64(s, f.range);
                s.push_back('\t'); fmt_f64(s, f.sampling_rate);
                s.push_back('\t'); fmt_u64(s, f.n_samples);
                s.push_back('\t');
                if (!drop_aux && n_aux && !aux_to_text(h_g + g_dst[2 * i + 1], g_len[2 * i + 1], n_aux, aux_type, suf[i])) { if (status) status[lo + i] = 16; fail = 2; continue; }
                suf[i].push_back('\n');
            }
        });
        if (fail == 1) { s5gpu_set_error("s5gpu_blow5_to_ascii_stream: signal formatting failed"); return sg.fail(S5GPU_ERR_HIP, slot); }
        if (fail == 2) { s5gpu_set_error("s5gpu_blow5_to_ascii_stream: aux bytes of at least one record do not match the header's aux types"); return sg.fail(S5GPU_ERR_DATA, slot); }
        std::vector<uint64_t> off(m + 1), p_src(2ull * m), p_dst(2ull * m), s_dst(m);
        std::vector<uint32_t> p_len(2ull * m);
        uint64_t bo = 0;
        off[0] = 0;
        for (uint32_t i = 0; i < m; i++) {
            p_src[2 * i] = bo; p_len[2 * i] = (uint32_t)pre[i].size(); p_dst[2 * i] = off[i]; bo += pre[i].size();
            s_dst[i] = off[i] + pre[i].size();
            p_src[2 * i + 1] = bo; p_len[2 * i + 1] = (uint32_t)suf[i].size(); p_dst[2 * i + 1] = s_dst[i] + tl[i]; bo += suf[i].size();
            off[i + 1] = p_dst[2 * i + 1] + suf[i].size();
        }
        uint64_t base = 0;
        bool copy = false;
        if ((r = sg.place(slot, off[m], out_cap, &base, &copy))) return r;
        if (!copy) return S5GPU_OK;
        // 3. prefix | signal text | suffix of every line to its place on the device, one D2H
        if ((r = c->d_stream.reserve(off[m] + 64)) || (r = c->d_aux.reserve(bo + 64)) || (r = c->h_in.reserve(b_all + bo + 64))) return r;
        h = (uint8_t *)c->h_in.p;
        uint8_t *hb = h + b_all;
        for (uint32_t i = 0; i < m; i++) {
            memcpy(hb + p_src[2 * i], pre[i].data(), pre[i].size());
            memcpy(hb + p_src[2 * i + 1], suf[i].data(), suf[i].size());
        }
        memcpy(h + o_p, p_src.data(), 16ull * m);
        memcpy(h + o_p + b_g8, p_dst.data(), 16ull * m);
        memcpy(h + o_p + 2 * b_g8, p_len.data(), 8ull * m);
        memcpy(h + o_p + 2 * b_g8 + b_g4, s_dst.data(), 8ull * m);
        HIP_TRY(hipMemcpyAsync(dv + o_p, h + o_p, b_all - o_p, hipMemcpyHostToDevice, c->st));
        HIP_TRY(hipMemcpyAsync(c->d_aux.p, hb, bo, hipMemcpyHostToDevice, c->st));
        if ((r = s5gpu_gather_dev(2 * m, (const uint64_t *)(dv + o_p), (const uint32_t *)(dv + o_p + 2 * b_g8), (const uint64_t *)(dv + o_p + b_g8),
                                  (const uint8_t *)c->d_aux.p, (uint8_t *)c->d_stream.p, c->st)))
            return r;
        // the signal text: slot i -> stream + s_dst[i] (the compaction kernel's copy: dword-wise with a byte shift)
        if ((r = s5gpu_scatter_slots_dev(m, (const s5gpu_read_desc_t *)(dv + o_rd), (const uint8_t *)c->d_txt.p, d_tl, (const uint64_t *)(dv + o_p + 2 * b_g8 + b_g4),
                                         (uint8_t *)c->d_stream.p, c->st)))
            return r;
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
        out_off[0] = need;
        s5gpu_set_error("s5gpu_blow5_to_ascii_stream: output buffer too small (%llu bytes needed)", (unsigned long long)need);
        return S5GPU_ERR_NOMEM;
    }
    return S5GPU_OK;
}
