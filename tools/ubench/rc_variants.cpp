// scratch variants of the range coder for A/B timing against the library build (tools/ubench/rc_ab.cpp)
#include <cstdint>
#include <cstring>
#include <vector>
#include <immintrin.h>
namespace {
inline int lz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
struct Sink {
    uint8_t* out; int64_t cap; int64_t len = 0; uint64_t acc = 0; int nbits = 0;
    inline void put(uint32_t v, int n) {
        acc = (acc << n) | (uint64_t)v; nbits += n;
        const int full = nbits >= 32; const int keep = nbits - (full << 5);
        const uint32_t w = __builtin_bswap32((uint32_t)(acc >> keep));
        if (len + 4 <= cap) std::memcpy(out + len, &w, 4);
        len += full << 2; nbits = keep;
    }
    inline void put_run(uint32_t bit, uint64_t count) { const uint32_t word = bit ? 0xFFFFFFFFu : 0u; while (count >= 32) { put(word, 32); count -= 32; } put(count ? (word >> (32 - count)) : 0u, (int)count); }
    inline void flush() { while (nbits > 0) { int take = nbits >= 8 ? 8 : nbits; uint8_t b = (uint8_t)((nbits >= 8 ? (acc >> (nbits - 8)) : (acc << (8 - nbits))) & 0xff); if (len < cap) out[len] = b; ++len; nbits -= take; } }
};
struct SourceBF {
    const uint8_t* in; int64_t pos = 0; uint64_t acc = 0; int nbits = 0;
    inline void refill() { uint64_t w; std::memcpy(&w, in + pos, 8); acc |= __builtin_bswap64(w) >> nbits; pos += (63 - nbits) >> 3; nbits |= 56; }
    inline uint32_t take(int n) { const uint32_t v = (uint32_t)((acc >> 1) >> (63 - n)); acc <<= n; nbits -= n; return v; }
};
}
// state (low, span): t = nshare + m from one xor/lzcnt + one andn/lzcnt
extern "C" __attribute__((target("lzcnt,bmi,bmi2"))) int64_t rc4_encode(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap) {
    Sink sink{out, cap};
    uint32_t low = 0; uint64_t span = 1ull << 32; uint64_t pending = 0;
    const int top_symbol = Lp - 2;
    std::vector<uint32_t> rows((size_t)C * Lp);
    for (int c = 0; c < C; ++c) { for (int j = 0; j < Lp - 1; ++j) rows[(size_t)c * Lp + j] = cdf[(size_t)c * Lp + j]; rows[(size_t)c * Lp + Lp - 1] = 0x10000u; }
    int ch = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* row = rows.data() + (size_t)ch * Lp;
        if (++ch == C) ch = 0;
        const int s = sym[i];
        if ((unsigned)s > (unsigned)top_symbol) return INT64_MIN;
        const uint32_t c_lo = (uint32_t)((span * row[s]) >> 16), c_hi = (uint32_t)((span * row[s + 1]) >> 16);
        const uint32_t lo = low + c_lo, hi = low + c_hi - 1;
        const int nshare = lz32(lo ^ hi);
        const uint32_t z = ~(lo & ~hi) & (uint32_t)(0x7FFFFFFFull >> nshare);
        const int t = lz32(z) - 1;                      // nshare + m
        const int m = t - nshare;
        if (nshare) {
            const uint32_t bits = (uint32_t)(((uint64_t)lo << nshare) >> 32);
            const uint32_t first = bits >> (nshare - 1);
            if (pending > 31) { sink.put(first, 1); sink.put_run(first ^ 1u, pending); }
            else sink.put(((1u << pending) - 1u) + first, (int)pending + 1);
            sink.put(bits & ((1u << (nshare - 1)) - 1u), nshare - 1);
            pending = 0;
        }
        pending += (uint64_t)m;
        low = (uint32_t)((uint64_t)lo << t) & 0x7FFFFFFFu;
        span = (uint64_t)(c_hi - c_lo) << t;
    }
    ++pending;
    const uint32_t last = low < 0x40000000u ? 0u : 1u;
    sink.put(last, 1); sink.put_run(last ^ 1u, pending); sink.flush();
    return sink.len <= cap ? sink.len : -sink.len;
}

// decoder state (low, span, off = value - low): no `value`, no `high`
extern "C" __attribute__((target("avx512f,avx512bw,avx512dq,popcnt,lzcnt,bmi,bmi2"))) int rc4_decode(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n) {
    std::vector<uint8_t> padded((size_t)nbytes + 64 + (size_t)n * 3, 0);
    std::memcpy(padded.data(), in, (size_t)nbytes);
    SourceBF src{padded.data()};
    src.refill();
    uint32_t low = 0; uint64_t span = 1ull << 32; uint32_t off = src.take(32);
    const int nvec = (Lp + 7) / 8, W = nvec * 8, RS = W + 8;
    // 32-bit rows with one guard entry in front and guards behind: any boundary count 0..W indexes inside the row
    std::vector<uint32_t> rows_store((size_t)C * RS, 0x10000u);
    for (int c = 0; c < C; ++c) { uint32_t* r = rows_store.data() + (size_t)c * RS; r[0] = 0; for (int j = 0; j < Lp - 1; ++j) r[1 + j] = cdf[(size_t)c * Lp + j]; }
    std::vector<uint64_t> wide_store((size_t)C * W + 8);
    uint64_t* wide = (uint64_t*)(((uintptr_t)wide_store.data() + 63) & ~(uintptr_t)63);
    for (int c = 0; c < C; ++c) for (int j = 0; j < W; ++j) wide[(size_t)c * W + j] = j < Lp - 1 ? cdf[(size_t)c * Lp + j] : 0x10000u;
    int ch = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* row = rows_store.data() + (size_t)ch * RS; const uint64_t* wr = wide + (size_t)ch * W;
        if (++ch == C) ch = 0;
        src.refill();
        const __m512i vs = _mm512_set1_epi64((long long)(span - 1)), voff = _mm512_set1_epi64((long long)(uint64_t)off);
        unsigned cnt = 0;
        for (int v = 0; v < nvec; ++v) {
            const __m512i r = _mm512_load_si512((const void*)(wr + 8 * v));
            const __m512i cum = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(vs, r), r), 16);
            cnt += (unsigned)__builtin_popcount((unsigned)_mm512_cmple_epu64_mask(cum, voff));
        }
        sym[i] = (int16_t)((int)cnt - 1);
        const uint32_t c_lo = (uint32_t)((span * row[cnt]) >> 16), c_hi = (uint32_t)((span * row[cnt + 1]) >> 16);   // row[1 + s], row[2 + s]
        const uint32_t lo = low + c_lo, hi = low + c_hi - 1;
        const int nshare = lz32(lo ^ hi);
        const uint32_t z = ~(lo & ~hi) & (uint32_t)(0x7FFFFFFFull >> nshare);
        const int t = lz32(z) - 1;
        low = (uint32_t)((uint64_t)lo << t) & 0x7FFFFFFFu;
        span = (uint64_t)(c_hi - c_lo) << t;
        off = (uint32_t)((uint64_t)(off - c_lo) << t) | src.take(t);
    }
    return 0;
}
