/*
 * ascii.c — CPU ORACLE for the SLOW5 ASCII record line <-> uncompressed BLOW5 payload conversion
 * (SURVEY.md §8f row 2).  TEST INFRASTRUCTURE ONLY — see s5oracle.h.
 *
 * Restates what slow5lib does for an ASCII file inside slow5_rec_depress_parse (line -> slow5_rec_t) and
 * slow5_rec_to_mem(..., SLOW5_FORMAT_ASCII, ...) (slow5_rec_t -> line); call sites /root/reference/src/view.c:38,49.
 * slow5lib is an absent submodule, so the text conventions are pinned on the reference's own ASCII/binary fixture
 * pairs by tests/test_oracle_golden.py::test_ascii_*:
 *   test/data/exp/one_fast5/exp_1_lossless.{slow5,blow5}, exp/aux_array/exp_lossless.{slow5,blow5},
 *   exp/index/example_multi_rg_v0.1.0.{slow5,blow5}  (doubles: "%f", trailing zeros trimmed — 195.77062844206847 <-> "195.770628")
 * Missing values ("."): strings / arrays are pinned by the enum fixture's text; scalar sentinels (type maximum, NaN) are
 * [RECALLED] from slow5lib's slow5_defs.h and not covered by a fixture pair.
 *
 * Deliberately scalar and strtol/snprintf based, one value at a time, like the code it restates.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "s5oracle.h"

static const unsigned kind_size[12] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 1, 1};

int s5o_aux_types(const char *line, size_t len, uint8_t *types, unsigned cap) {
    static const char *const names[] = {"int8_t", "int16_t", "int32_t", "int64_t", "uint8_t", "uint16_t", "uint32_t", "uint64_t", "float", "double", "char"};
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
    size_t b = (len && line[0] == '#') ? 1 : 0;
    unsigned col = 0, n = 0;
    while (b <= len) {
        const char *t = memchr(line + b, '\t', len - b);
        size_t l = t ? (size_t)(t - line) - b : len - b;
        if (col >= 8) {
            size_t m = l;
            int code = -1, arr = 0;
            if (m && line[b + m - 1] == '*') { arr = 0x80; m--; }
            for (int k = 0; k < 11; k++)
                if (strlen(names[k]) == m && memcmp(line + b, names[k], m) == 0) code = k;
            if (code < 0 && m > 5 && memcmp(line + b, "enum{", 5) == 0 && line[b + m - 1] == '}') code = 11;
            if (code < 0 || n >= cap) return -1;
            types[n++] = (uint8_t)(code | arr);
        }
        col++;
        b += l + 1;
    }
    return col < 8 ? -1 : (int)n;
}

size_t s5o_double_to_text(double v, char *out) {
    if (isnan(v)) { out[0] = '.'; return 1; }
    int n = sprintf(out, "%f", v);
    if (memchr(out, '.', (size_t)n)) {
        while (n && out[n - 1] == '0') n--;
        if (n && out[n - 1] == '.') n--;
    }
    return (size_t)n;
}

size_t s5o_signal_to_text(const int16_t *sig, uint64_t n, char *out) {
    size_t o = 0;
    for (uint64_t i = 0; i < n; i++) o += (size_t)sprintf(out + o, i ? ",%d" : "%d", sig[i]);
    return o;
}

/* returns the number of samples, or -1 when the text is not a comma-separated list of int16 values */
int64_t s5o_text_to_signal(const char *txt, size_t len, int16_t *out, uint64_t cap) {
    uint64_t n = 0;
    size_t b = 0;
    if (len == 0) return 0;
    while (b <= len) {
        const char *c = memchr(txt + b, ',', len - b);
        size_t e = c ? (size_t)(c - txt) : len;
        char tmp[16];
        if (e == b || e - b >= sizeof tmp) return -1;
        memcpy(tmp, txt + b, e - b);
        tmp[e - b] = 0;
        for (size_t k = 0; k < e - b; k++)
            if (!((tmp[k] >= '0' && tmp[k] <= '9') || (k == 0 && tmp[k] == '-'))) return -1;
        char *end;
        long v = strtol(tmp, &end, 10);
        if (*end || end == tmp || v < INT16_MIN || v > INT16_MAX) return -1;
        if (n >= cap) return -1;
        out[n++] = (int16_t)v;
        b = e + 1;
    }
    return (int64_t)n;
}

static int field_num(const char *p, size_t l, char *tmp, size_t cap) {
    if (l == 0 || l >= cap) return -1;
    memcpy(tmp, p, l);
    tmp[l] = 0;
    return 0;
}

static int elem_from_text(int kind, const char *p, size_t l, int allow_missing, uint8_t *out) {
    char tmp[96], *end;
    int dot = allow_missing && l == 1 && p[0] == '.';
    if (kind == 10) { if (l != 1) return -1; out[0] = dot ? 0 : (uint8_t)p[0]; return 0; }
    if (!dot && field_num(p, l, tmp, sizeof tmp)) return -1;
    errno = 0;
    if (kind <= 3) {
        static const long long lo[4] = {INT8_MIN, INT16_MIN, INT32_MIN, INT64_MIN}, hi[4] = {INT8_MAX, INT16_MAX, INT32_MAX, INT64_MAX};
        long long v = hi[kind];
        if (!dot) { v = strtoll(tmp, &end, 10); if (*end || errno || v < lo[kind] || v > hi[kind]) return -1; }
        memcpy(out, &v, kind_size[kind]);
    } else if (kind <= 7 || kind == 11) {
        static const unsigned long long hi[4] = {UINT8_MAX, UINT16_MAX, UINT32_MAX, UINT64_MAX};
        unsigned long long mx = kind == 11 ? UINT8_MAX : hi[kind - 4], v = mx;
        if (!dot) { if (tmp[0] == '-') return -1; v = strtoull(tmp, &end, 10); if (*end || errno || v > mx) return -1; }
        memcpy(out, &v, kind_size[kind]);
    } else {
        double v = NAN;
        if (!dot) { v = strtod(tmp, &end); if (*end) return -1; }
        if (kind == 8) { float f = (float)v; memcpy(out, &f, 4); } else memcpy(out, &v, 8);
    }
    return 0;
}

/* line (with or without '\n') -> uncompressed BLOW5 payload with sig_method none (Appendix A.3).  Returns payload length,
 * 0 on a malformed line.  out must hold 2*len bytes (binary is never larger than that). */
size_t s5o_ascii_line_to_payload(const char *line, size_t len, const uint8_t *types, unsigned n_aux, uint8_t *out) {
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
    const char *f[8];
    size_t fl[8], b = 0;
    for (int k = 0; k < 8; k++) {
        if (b > len) return 0;
        const char *t = memchr(line + b, '\t', len - b);
        f[k] = line + b;
        fl[k] = t ? (size_t)(t - line) - b : len - b;
        b += fl[k] + 1;
    }
    char tmp[96], *end;
    size_t o = 0;
    if (fl[0] == 0 || fl[0] > 0xFFFF) return 0;
    uint16_t idl = (uint16_t)fl[0];
    memcpy(out + o, &idl, 2); o += 2;
    memcpy(out + o, f[0], idl); o += idl;
    if (field_num(f[1], fl[1], tmp, sizeof tmp)) return 0;
    unsigned long long rg = strtoull(tmp, &end, 10);
    if (*end || rg > UINT32_MAX) return 0;
    uint32_t rg32 = (uint32_t)rg;
    memcpy(out + o, &rg32, 4); o += 4;
    for (int k = 2; k < 6; k++) {
        if (field_num(f[k], fl[k], tmp, sizeof tmp)) return 0;
        double v = strtod(tmp, &end);
        if (*end) return 0;
        memcpy(out + o, &v, 8); o += 8;
    }
    if (field_num(f[6], fl[6], tmp, sizeof tmp)) return 0;
    uint64_t ns = strtoull(tmp, &end, 10);
    if (*end) return 0;
    memcpy(out + o, &ns, 8); o += 8;
    {
        int16_t *sig = (int16_t *)malloc(ns ? 2 * ns : 2);
        if (!sig) return 0;
        int64_t got = s5o_text_to_signal(f[7], fl[7], sig, ns);
        if (got < 0 || (uint64_t)got != ns) { free(sig); return 0; }
        memcpy(out + o, sig, 2 * ns);
        free(sig);
    }
    o += 2 * ns;
    for (unsigned a = 0; a < n_aux; a++) {
        if (b > len) return 0;
        const char *t = memchr(line + b, '\t', len - b);
        size_t l = t ? (size_t)(t - line) - b : len - b;
        const char *p = line + b;
        int kind = types[a] & 15;
        if (!(types[a] & 0x80)) {
            if (elem_from_text(kind, p, l, 1, out + o)) return 0;
            o += kind_size[kind];
        } else if (l == 1 && p[0] == '.') {
            uint64_t z = 0; memcpy(out + o, &z, 8); o += 8;
        } else if (kind == 10) {
            uint64_t c = l; memcpy(out + o, &c, 8); o += 8; memcpy(out + o, p, l); o += l;
        } else {
            size_t at = o, bb = 0; uint64_t cnt = 0;
            o += 8;
            while (bb <= l) {
                const char *c = memchr(p + bb, ',', l - bb);
                size_t e = c ? (size_t)(c - p) : l;
                if (elem_from_text(kind, p + bb, e - bb, 0, out + o)) return 0;
                o += kind_size[kind]; cnt++; bb = e + 1;
            }
            memcpy(out + at, &cnt, 8);
        }
        b += l + 1;
    }
    if (b <= len) return 0;
    return o;
}

static size_t elem_to_text(int kind, const uint8_t *p, int allow_missing, char *out) {
    long long sv = 0; unsigned long long uv = 0;
    switch (kind) {
    case 0: { int8_t v; memcpy(&v, p, 1); sv = v; if (allow_missing && v == INT8_MAX) goto dot; return (size_t)sprintf(out, "%lld", sv); }
    case 1: { int16_t v; memcpy(&v, p, 2); sv = v; if (allow_missing && v == INT16_MAX) goto dot; return (size_t)sprintf(out, "%lld", sv); }
    case 2: { int32_t v; memcpy(&v, p, 4); sv = v; if (allow_missing && v == INT32_MAX) goto dot; return (size_t)sprintf(out, "%lld", sv); }
    case 3: { int64_t v; memcpy(&v, p, 8); sv = v; if (allow_missing && v == INT64_MAX) goto dot; return (size_t)sprintf(out, "%lld", sv); }
    case 4: case 11: { uint8_t v; memcpy(&v, p, 1); uv = v; if (allow_missing && v == UINT8_MAX) goto dot; return (size_t)sprintf(out, "%llu", uv); }
    case 5: { uint16_t v; memcpy(&v, p, 2); uv = v; if (allow_missing && v == UINT16_MAX) goto dot; return (size_t)sprintf(out, "%llu", uv); }
    case 6: { uint32_t v; memcpy(&v, p, 4); uv = v; if (allow_missing && v == UINT32_MAX) goto dot; return (size_t)sprintf(out, "%llu", uv); }
    case 7: { uint64_t v; memcpy(&v, p, 8); uv = v; if (allow_missing && v == UINT64_MAX) goto dot; return (size_t)sprintf(out, "%llu", uv); }
    case 8: { float v; memcpy(&v, p, 4); return s5o_double_to_text((double)v, out); }
    case 9: { double v; memcpy(&v, p, 8); return s5o_double_to_text(v, out); }
    case 10: out[0] = (allow_missing && p[0] == 0) ? '.' : (char)p[0]; return 1;
    }
    return 0;
dot:
    out[0] = '.';
    return 1;
}

/* uncompressed payload (sig_method none) -> line ending in '\n'.  Returns the length, 0 when the aux bytes do not match the types.
 * out must hold 8*payload_len + 512 bytes. */
size_t s5o_payload_to_ascii_line(const uint8_t *pay, size_t len, const uint8_t *types, unsigned n_aux, char *out) {
    size_t p = 0, o = 0;
    uint16_t idl;
    if (len < 2) return 0;
    memcpy(&idl, pay, 2); p = 2;
    if (p + idl + 4 + 32 + 8 > len) return 0;
    memcpy(out, pay + p, idl); o = idl; p += idl;
    uint32_t rg; memcpy(&rg, pay + p, 4); p += 4;
    o += (size_t)sprintf(out + o, "\t%u", rg);
    for (int k = 0; k < 4; k++) { double v; memcpy(&v, pay + p, 8); p += 8; out[o++] = '\t'; o += s5o_double_to_text(v, out + o); }
    uint64_t ns; memcpy(&ns, pay + p, 8); p += 8;
    if (ns > (len - p) / 2) return 0;
    o += (size_t)sprintf(out + o, "\t%llu\t", (unsigned long long)ns);
    for (uint64_t i = 0; i < ns; i++) { int16_t v; memcpy(&v, pay + p + 2 * i, 2); o += (size_t)sprintf(out + o, i ? ",%d" : "%d", v); }
    p += 2 * ns;
    for (unsigned a = 0; a < n_aux; a++) {
        int kind = types[a] & 15;
        unsigned es = kind_size[kind];
        out[o++] = '\t';
        if (!(types[a] & 0x80)) {
            if (p + es > len) return 0;
            o += elem_to_text(kind, pay + p, 1, out + o); p += es;
            continue;
        }
        uint64_t cnt;
        if (p + 8 > len) return 0;
        memcpy(&cnt, pay + p, 8); p += 8;
        if (cnt > (len - p) / es) return 0;
        if (cnt == 0) { out[o++] = '.'; continue; }
        if (kind == 10) { memcpy(out + o, pay + p, cnt); o += cnt; p += cnt; continue; }
        for (uint64_t e = 0; e < cnt; e++) { if (e) out[o++] = ','; o += elem_to_text(kind, pay + p, 0, out + o); p += es; }
    }
    if (p != len) return 0;
    out[o++] = '\n';
    return o;
}
