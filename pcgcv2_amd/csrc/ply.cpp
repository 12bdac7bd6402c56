// Native ASCII-PLY geometry reader / writer (host).  Replaces the pure-Python per-line / per-point loops of the
// reference's read_ply_ascii_geo / write_ply_ascii_geo (data_utils.py:19-48), which dominate its (untimed) wall clock at
// ~10^6 points.  Acceptance rule of the reference reader, kept exactly: a line is a data row iff every ' '-separated token
// (a lone "\n" token is skipped) parses as a float; header lines drop out because they do not; the first three columns
// are kept and truncated toward zero like numpy's astype('int').
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/pcgc_hip.h"

namespace {

// parse one token [b,e) as Python's float() would (surrounding whitespace allowed); returns false if it does not parse
bool parse_float_token(const char* b, const char* e, double& out) {
    while (b < e && (*b == ' ' || *b == '\t' || *b == '\r' || *b == '\n' || *b == '\f' || *b == '\v')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r' || e[-1] == '\n' || e[-1] == '\f' || e[-1] == '\v')) --e;
    if (b == e) return false;
    // fast path: [+-]digits[.digits]
    const char* p = b;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
    const char* d0 = p;
    uint64_t ip = 0; int nd = 0;
    while (p < e && *p >= '0' && *p <= '9' && nd < 18) { ip = ip * 10 + (uint64_t)(*p - '0'); ++p; ++nd; }
    if (p == e && nd > 0) { out = neg ? -(double)ip : (double)ip; return true; }
    if (p < e && *p == '.' && nd < 18) {
        const char* q = p + 1; uint64_t fp = 0; int nf = 0;
        while (q < e && *q >= '0' && *q <= '9' && nf < 18) { fp = fp * 10 + (uint64_t)(*q - '0'); ++q; ++nf; }
        if (q == e && (nd > 0 || nf > 0)) {
            static const double p10[19] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18};
            double v = (double)ip + (double)fp / p10[nf];
            out = neg ? -v : v; return true;
        }
    }
    (void)d0;
    // general path (exponents, inf, nan, long mantissas): strtod on a NUL-terminated copy, must consume everything
    std::string tmp(b, e);
    for (char c : tmp) if (c == '_') return false;                    // Python accepts 1_000; PLY files never contain it: reject
    errno = 0; char* end = nullptr;
    double v = std::strtod(tmp.c_str(), &end);
    if (end == tmp.c_str() || *end != '\0') return false;
    if (tmp.find('x') != std::string::npos || tmp.find('X') != std::string::npos) return false;     // no hex floats in Python
    out = v; return true;
}

}  // namespace

// Returns the number of data rows.  If xyz != NULL, fills up to cap rows (int32 x,y,z).  <0 on I/O error.
extern "C" int64_t pcgc_ply_read_ascii_geo(const char* path, int32_t* xyz, int64_t cap) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return -1;
    std::fseek(f, 0, SEEK_END); long size = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)size + 1);
    if (size > 0 && std::fread(buf.data(), 1, (size_t)size, f) != (size_t)size) { std::fclose(f); return -1; }
    std::fclose(f);
    buf[(size_t)size] = '\n';
    const char* p = buf.data(); const char* end = p + size;
    int64_t rows = 0;
    while (p < end) {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        const char* line_end = eol ? eol + 1 : end;                    // the line INCLUDING its '\n', as Python iterates
        double v[3] = {0, 0, 0}; int ncol = 0; bool ok = true;
        const char* t = p;
        while (t <= line_end && ok) {
            const char* sp = (const char*)std::memchr(t, ' ', (size_t)(line_end - t));
            const char* te = sp ? sp : line_end;
            if (!(te - t == 1 && *t == '\n')) {                       // `if v == '\n': continue`
                double x;
                if (!parse_float_token(t, te, x)) ok = false;
                else { if (ncol < 3) v[ncol] = x; ++ncol; }
            }
            if (!sp) break;
            t = sp + 1;
        }
        if (ok && ncol > 0) {
            if (ncol < 3) return -3;                                  // the reference would fail building its [N,3] slice
            if (xyz && rows < cap) { xyz[3 * rows] = (int32_t)v[0]; xyz[3 * rows + 1] = (int32_t)v[1]; xyz[3 * rows + 2] = (int32_t)v[2]; }
            ++rows;
        }
        p = line_end;
    }
    return rows;
}

extern "C" int pcgc_ply_write_ascii_geo(const char* path, const int32_t* xyz, int64_t n) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    std::vector<char> out; out.reserve((size_t)n * 16 + 256);
    char head[256];
    int h = std::snprintf(head, sizeof(head), "ply\nformat ascii 1.0\nelement vertex %lld\nproperty float x\nproperty float y\nproperty float z\nend_header\n", (long long)n);
    out.insert(out.end(), head, head + h);
    char tmp[16];
    auto put_int = [&](int32_t v) {
        int64_t a = v; if (a < 0) { out.push_back('-'); a = -a; }
        int k = 0; do { tmp[k++] = (char)('0' + a % 10); a /= 10; } while (a);
        while (k) out.push_back(tmp[--k]);
    };
    for (int64_t i = 0; i < n; ++i) {
        put_int(xyz[3 * i]); out.push_back(' '); put_int(xyz[3 * i + 1]); out.push_back(' '); put_int(xyz[3 * i + 2]); out.push_back('\n');
    }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok ? 0 : -1;
}
