// Host-side (sequential) entropy coding of the PCGCv2 bitstream:
//   * pcgc_rc_encode / pcgc_rc_decode — the 32-bit range coder behind torchac 0.9.3 ‡ encode_float_cdf /
//     decode_float_cdf (reference call sites entropy_model.py:174,192).  One 16-bit CDF row per channel instead of
//     the reference's [N8, C, L+1] expansion (entropy_model.py:173).
//   * pcgc_oct_encode / pcgc_oct_decode — native lossless codec for the stride-8 coordinates (`_C.bin`), used when
//     the external G-PCC `tmc3` binary of gpcc.py:6-41 is not installed.  Not G-PCC interoperable (magic "PCGO").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>
#include <memory>
#include <string>
#include <zlib.h>
#include "../../include/pcgc_hip.h"
void pcgc_set_error(const char* fmt, ...);           // coords.hip

// ------------------------------------------------------------------------------------------------ torchac-compatible
namespace {
inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
#ifndef PCGC_SINK_BYTES
// MSB-first bit writer: fields are appended to a 64-bit word (shift, or, add, compare: the whole common path); a full word is stored
// big-endian when the next field does not fit — once per ~19 symbols of a latent stream.  (The reference appends one bit at a time.)
struct Sink {
    uint8_t* out; int64_t cap; int64_t len = 0; uint64_t acc = 0; int nbits = 0;   // nbits < 64 bits of acc (right-aligned) not yet stored
    __attribute__((always_inline)) inline void put(uint64_t v, int n) {            // n <= 56, v < 2^n
        const int room = 64 - nbits;                                               // >= 1
        if (__builtin_expect(n < room, 1)) { acc = (acc << n) | v; nbits += n; return; }
        const int spill = n - room;                                                // bits of v that start the next word
        const uint64_t word = __builtin_bswap64((acc << (room & 63)) | (v >> spill));   // (room = 64 never gets here: n <= 56)
        if (__builtin_expect(len + 8 <= cap, 1)) std::memcpy(out + len, &word, 8);
        else for (int b = 0; b < 8; ++b) if (len + b < cap) out[len + b] = (uint8_t)(word >> (8 * b));
        len += 8;
        acc = v & ((1ull << spill) - 1ull); nbits = spill;
    }
    uint64_t bits_written() const { return (uint64_t)len * 8 + (uint64_t)nbits; }
    // (not inlined, and called on a COPY in the coder's loop: a call that takes the writer's address would pin its fields — and
    // with them the loop's whole state — to the stack)
    __attribute__((noinline)) void put_run(uint32_t bit, uint64_t count) { const uint32_t word = bit ? 0xFFFFFFFFu : 0u; while (count >= 32) { put(word, 32); count -= 32; } put(count ? (word >> (32 - count)) : 0u, (int)count); }
    inline void flush() {                                                           // the last bits, zero-padded to a byte
        const int bytes = (nbits + 7) >> 3;
        const uint64_t word = nbits ? acc << (64 - nbits) : 0;
        for (int b = 0; b < bytes; ++b) if (len + b < cap) out[len + b] = (uint8_t)(word >> (56 - 8 * b));
        len += bytes; nbits = 0; acc = 0;
    }
};
#else
struct Sink {
    uint8_t* out; int64_t cap; int64_t len = 0; uint64_t acc = 0; int nbits = 0;   // nbits < 8 between calls
    __attribute__((always_inline)) inline void put(uint64_t v, int n) {            // n <= 56, v < 2^n
        acc |= (v << 1) << (63 - nbits - n);
        nbits += n;
        const int adv = nbits >> 3;
        if (__builtin_expect(len + 8 <= cap, 1)) { const uint64_t w = __builtin_bswap64(acc); std::memcpy(out + len, &w, 8); }
        else for (int b = 0; b < adv; ++b) if (len + b < cap) out[len + b] = (uint8_t)(acc >> (56 - 8 * b));
        len += adv;
        acc <<= adv << 3;
        nbits &= 7;
    }
    uint64_t bits_written() const { return (uint64_t)len * 8 + (uint64_t)nbits; }
    __attribute__((noinline)) void put_run(uint32_t bit, uint64_t count) { const uint32_t word = bit ? 0xFFFFFFFFu : 0u; while (count >= 32) { put(word, 32); count -= 32; } put(count ? (word >> (32 - count)) : 0u, (int)count); }
    inline void flush() { if (nbits > 0) { if (len < cap) out[len] = (uint8_t)(acc >> 56); ++len; nbits = 0; acc = 0; } }    // zero-padded to a byte
};
#endif
// MSB-first bit readers over a copy of the stream with 64 zero bytes behind it: past the end the stream reads as zeros (the
// reference's reader too).  The read position is clamped to `limit` = stream bytes + 8, inside the zeros: a corrupt stream can ask
// for any number of bits and keeps getting zeros, without the decoder ever leaving the buffer (the copy used to be padded by four
// bytes per symbol instead: 700 KB of page faults and zero fill per frame).
struct Source {                                      // refills 32 bits at a time
    const uint8_t* in; int64_t limit; int64_t pos = 0; uint64_t acc = 0; int nbits = 0;
    inline uint32_t take(int n) {
        if (nbits < n) { if (pos > limit) pos = limit; uint32_t w; std::memcpy(&w, in + pos, 4); pos += 4; acc = (acc << 32) | __builtin_bswap32(w); nbits += 32; }
        nbits -= n;
        return n == 0 ? 0u : (uint32_t)((acc >> nbits) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1ull)));
    }
};
struct SourceBF {                                    // branch-free: refill() leaves >= 56 valid bits at the top of acc
    const uint8_t* in; int64_t limit; int64_t pos = 0; uint64_t acc = 0; int nbits = 0;
    inline void refill() {
        pos = pos > limit ? limit : pos;
        uint64_t w; std::memcpy(&w, in + pos, 8); acc |= __builtin_bswap64(w) >> nbits; pos += (63 - nbits) >> 3; nbits |= 56;
    }
    inline uint32_t take(int n) { const uint32_t v = (uint32_t)((acc >> 1) >> (63 - n)); acc <<= n; nbits -= n; return v; }   // n <= 32
};
// One 16-bit CDF row per channel, widened to 32 bits with the last boundary pinned to 2^16 (torchac hard-codes
// c_high = 0x10000 for the top symbol): removes the per-symbol special case.
std::vector<uint32_t> widen_rows(const uint16_t* cdf, int C, int Lp) {
    std::vector<uint32_t> rows((size_t)C * Lp);
    for (int c = 0; c < C; ++c) { for (int j = 0; j < Lp - 1; ++j) rows[(size_t)c * Lp + j] = cdf[(size_t)c * Lp + j]; rows[(size_t)c * Lp + Lp - 1] = 0x10000u; }
    return rows;
}
int g_rc_impl = 0;                                   // 0 automatic, 1 portable scalar decoder (A/B and tests)
}
extern "C" int pcgc_set_rc_impl(int impl) { if (impl < 0 || impl > 1) return -1; g_rc_impl = impl; return 0; }

// Renormalisation in runs instead of single bits, without data-dependent branches.  The coder state is kept as
// (low, span = high - low + 1): every E1/E2/E3 step of torchac ‡ doubles the span, so after t steps span' = span << t and
// `high` never has to be formed.  After narrowing to [lo, hi] for a symbol:
//   (1) lo and hi share n = clz(lo ^ hi) leading bits -> they are emitted; the first one, b, resolves the pending
//       E3 bits: "b followed by `pending` copies of !b" is the number (2^pending - 1) + b in pending+1 bits, so it goes out
//       as ONE field whatever b and pending are (pending = 0 included);
//   (2) below the first differing bit (lo 0, hi 1) the E3 "near convergence" case repeats for every further position
//       where lo has a 1 and hi a 0: with y = lo & ~hi, the run ends at the first zero of y below bit 31 - n, i.e.
//       t = n + m = clz(~y & (0x7FFFFFFF >> n)) - 1 (an all-ones tail gives clz = 32: the run reaches bit 0);
//   (3) low' = (lo << t) with the MSB cleared (E3 pins it to 0; after E1/E2 alone it already is 0), pending += m.
// After that neither case applies again, exactly as in the bit-serial loop of torchac.
namespace {
// Decoder state at a symbol boundary (see pcgc_rc_encode_indexed): everything a decoder needs to start there.
struct RcCkpt { uint32_t sym, bitpos_lo, bitpos_hi, low, span_m1, off; };
static_assert(sizeof(RcCkpt) == 4 * PCGC_RC_CKPT_WORDS, "checkpoint layout");

// n_ck checkpoints at the ascending symbol indices ck[].sym (filled in by the caller); the other fields are written here.
template <class LZ>
__attribute__((always_inline)) inline int64_t rc_encode_body(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, LZ lz,
                                                             RcCkpt* ck = nullptr, int n_ck = 0) {
    Sink sink{out, cap};
    uint32_t low = 0; uint64_t span = 1ull << 32; uint64_t pending = 0;
    int next_ck = 0; int64_t ck_at = n_ck > 0 ? (int64_t)ck[0].sym : -1;
    const int top_symbol = Lp - 2;
    const std::vector<uint32_t> rows = widen_rows(cdf, C, Lp);
    {                                                           // symbols are checked up front (a vectorised pass), not one by one in the chain below
        unsigned bad = 0;
        for (int64_t i = 0; i < n; ++i) bad |= (unsigned)((unsigned)sym[i] > (unsigned)top_symbol);
        if (bad) return INT64_MIN;
    }
    // (state passed BY VALUE: a by-reference capture in a lambda that is not inlined would keep low / span / pending in memory)
    auto checkpoint = [ck, n_ck](int at, uint64_t bits_written, uint64_t pend, uint32_t lo_now, uint64_t span_now) -> int64_t {
        // renormalisation shifts so far = bits written + pending E3 bits (every E1/E2 shift writes one, every E3 shift defers one);
        // pending > 0 exactly when the last shift run ended in E3 steps (an E1/E2 shift resolves all of them)
        const uint64_t shifts = bits_written + pend;
        RcCkpt& k = ck[at];
        k.bitpos_lo = (uint32_t)shifts; k.bitpos_hi = (uint32_t)(shifts >> 32); k.low = lo_now; k.span_m1 = (uint32_t)(span_now - 1);
        k.off = pend ? 0x80000000u : 0u;
        return at + 1 < n_ck ? (int64_t)ck[at + 1].sym : -1;           // where the next checkpoint is due
    };
    auto step = [&](const uint32_t* row, const int s) __attribute__((always_inline)) {
        const uint32_t c_lo = (uint32_t)((span * row[s]) >> 16), c_hi = (uint32_t)((span * row[s + 1]) >> 16);
        const uint32_t lo = low + c_lo, hi = low + c_hi - 1;
        const int nshare = lz(lo ^ hi);
        // (2) in the domain shifted up by one, with the lowest bit set: the "- 1" and the all-ones special case disappear, and the
        // shifted operand does not depend on nshare (one instruction less on the loop-carried chain; nshare < 32: hi > lo always)
        const int t = lz((((~lo | hi) << 1) | 1u) & (0xFFFFFFFFu >> nshare));
        // the coder state first: this is the loop-carried dependency chain; the bit emission below hangs off it
        low = (uint32_t)((uint64_t)lo << t) & 0x7FFFFFFFu;
        span = (uint64_t)(c_hi - c_lo) << t;
        // "first shared bit, `pending` copies of its complement, the other shared bits" as ONE field of pending + nshare bits: the
        // first two parts are the number (2^pending - 1) + first, so the field is bits + (2^pending - 1) 2^(nshare - 1).  No shared
        // bit (a symbol of probability > 1/2 may not settle one): a field of zero bits, by masks instead of a branch — whether a
        // symbol settles a bit is what the stream encodes, i.e. unpredictable.
        const uint64_t emit = (uint64_t)0 - (uint64_t)(nshare != 0);
        if (__builtin_expect(pending + (uint64_t)nshare > 56, 0)) {                // (a run of > 24 E3 steps: never seen, handled)
            if (nshare) {
                const uint32_t bits = (uint32_t)(((uint64_t)lo << nshare) >> 32), first = bits >> (nshare - 1);
                Sink far = sink;
                far.put(first, 1); far.put_run(first ^ 1u, pending);
                far.put(bits & ((1u << (nshare - 1)) - 1u), nshare - 1);
                sink = far;
            }
        } else {
            const uint64_t bits = ((uint64_t)lo << nshare) >> 32;                    // the nshare shared bits
            sink.put((bits + (((1ull << pending) - 1ull) << ((nshare - 1) & 63))) & emit, (int)((pending + (uint64_t)nshare) & emit));
        }
        pending = (pending & ~emit) + (uint64_t)(t - nshare);
    };
    // whole points (C symbols, one table row each) with the checkpoint test once per point — checkpoints sit on point boundaries
    // (pcgc_rc_encode_indexed) — then the ragged tail, if any, symbol by symbol
    bool on_points = true;
    for (int c = 0; c < n_ck; ++c) on_points = on_points && ck[c].sym % (uint32_t)C == 0;
    const int64_t points = on_points ? n / C : 0;
    const size_t stride = (size_t)Lp;
    for (int64_t r = 0; r < points; ++r) {
        if (__builtin_expect(r * C == ck_at, 0)) { ck_at = checkpoint(next_ck, sink.bits_written(), pending, low, span); ++next_ck; }
        const int16_t* sp = sym + r * C;
        const uint32_t* row = rows.data();
        for (int ch = 0; ch < C; ++ch, row += stride) step(row, sp[ch]);
    }
    {
        int ch = 0;
        for (int64_t i = points * C; i < n; ++i) {
            if (__builtin_expect(i == ck_at, 0)) { ck_at = checkpoint(next_ck, sink.bits_written(), pending, low, span); ++next_ck; }
            step(rows.data() + (size_t)ch * Lp, sym[i]);
            if (++ch == C) ch = 0;
        }
    }
    ++pending;
    const uint32_t last = low < 0x40000000u ? 0u : 1u;
    { Sink far = sink; far.put(last, 1); far.put_run(last ^ 1u, pending); far.flush(); sink = far; }
    if (sink.len > cap) return -sink.len;
    // The decoder keeps off = value - low, and value is the 32-bit window of the stream at its read position, minus 2^31 while
    // the last renormalisation ended in E3 steps (torchac flips the top bit there; the flip is shifted out by the next shift):
    // with the stream complete, off at a checkpoint is window - low - e3 (mod 2^32).  Past the end the stream reads as zeros.
    for (int c = 0; c < n_ck; ++c) {
        RcCkpt& k = ck[c];
        const uint64_t pos = ((uint64_t)k.bitpos_hi << 32) | k.bitpos_lo;
        uint64_t w = 0;
        for (int b = 0; b < 5; ++b) { const uint64_t at = (pos >> 3) + b; w = (w << 8) | (at < (uint64_t)sink.len ? out[at] : 0u); }
        const uint32_t window = (uint32_t)(w >> (8 - (pos & 7)));
        k.off = window - k.low - k.off;
    }
    return sink.len;
}
__attribute__((target("lzcnt,bmi,bmi2")))
int64_t rc_encode_bmi(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, RcCkpt* ck, int n_ck) {
    return rc_encode_body(cdf, C, Lp, sym, n, out, cap, [](uint32_t v) __attribute__((target("lzcnt"))) { return (int)_lzcnt_u32(v); }, ck, n_ck);
}
int64_t rc_encode_generic(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, RcCkpt* ck, int n_ck) {
    return rc_encode_body(cdf, C, Lp, sym, n, out, cap, [](uint32_t v) { return clz32(v); }, ck, n_ck);
}
int64_t rc_encode_any(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, RcCkpt* ck, int n_ck) {
    if (g_rc_impl == 0 && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("lzcnt")) return rc_encode_bmi(cdf, C, Lp, sym, n, out, cap, ck, n_ck);
    return rc_encode_generic(cdf, C, Lp, sym, n, out, cap, ck, n_ck);
}
}
extern "C" int64_t pcgc_rc_encode(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap) {
    return rc_encode_any(cdf, C, Lp, sym, n, out, cap, nullptr, 0);
}
// The same stream, plus a decoding index: the decoder state at n_ckpt symbol boundaries (multiples of C, evenly spread), from which
// independent threads can decode the segments in between (pcgc_rc_decode_indexed).  The stream itself does not change by a bit.
extern "C" int64_t pcgc_rc_encode_indexed(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap,
                                          int n_ckpt, uint32_t* ckpt /*[n_ckpt][PCGC_RC_CKPT_WORDS]*/) {
    if (n_ckpt < 0 || (n_ckpt > 0 && !ckpt) || C < 1) return INT64_MIN;
    RcCkpt* ck = (RcCkpt*)ckpt;
    const int64_t rows = n / C;
    for (int c = 0; c < n_ckpt; ++c) {
        const int64_t at = (rows * c / n_ckpt) * C;        // checkpoint c opens segment c; checkpoint 0 is the start of the stream
        if (at > 0xFFFFFFFFll) return INT64_MIN;
        ck[c].sym = (uint32_t)at;
    }
    int uniq = 0;                                          // (tiny inputs: drop repeated positions)
    for (int c = 0; c < n_ckpt; ++c) if (c == 0 || ck[c].sym != ck[uniq - 1].sym) ck[uniq++].sym = ck[c].sym;
    for (int c = uniq; c < n_ckpt; ++c) ck[c] = RcCkpt{0xFFFFFFFFu, 0, 0, 0, 0, 0};      // unused entries
    return rc_encode_any(cdf, C, Lp, sym, n, out, cap, ck, n > 0 ? uniq : 0);
}

// Where a decoder starts: the beginning of the stream, or a checkpoint of pcgc_rc_encode_indexed.
struct RcStart { uint64_t bitpos; uint32_t low; uint64_t span; uint32_t off; int64_t first, count; };

// Portable decoder: torchac's target = ((value - low + 1) * 2^16 - 1) / span, symbol search seeded from the target's high
// byte.
struct RcScalarTables {
    std::vector<uint32_t> rows; std::vector<int16_t> seed;
    RcScalarTables(const uint16_t* cdf, int C, int Lp) : rows(widen_rows(cdf, C, Lp)), seed((size_t)C * 256) {
        const int top_symbol = Lp - 2;
        for (int c = 0; c < C; ++c) { const uint32_t* row = rows.data() + (size_t)c * Lp; int m = 0; for (int b = 0; b < 256; ++b) { const uint32_t t = (uint32_t)b << 8; while (m < top_symbol && row[m + 1] <= t) ++m; seed[(size_t)c * 256 + b] = (int16_t)m; } }
    }
};
static void rc_decode_scalar_seg(const RcScalarTables& tb, int C, int Lp, const uint8_t* padded, int64_t limit, const RcStart& st, int16_t* sym) {
    const uint64_t start = st.bitpos + 32;                       // the decoder has read 32 bits beyond the bits it has shifted out
    Source src{padded, limit, (int64_t)(start >> 3)};
    (void)src.take((int)(start & 7));
    uint32_t low = st.low, high = (uint32_t)(st.low + st.span - 1); uint32_t value = st.low + st.off;
    int ch = (int)(st.first % C);
    for (int64_t i = st.first; i < st.first + st.count; ++i) {
        const uint32_t* row = tb.rows.data() + (size_t)ch * Lp; const int16_t* sd = tb.seed.data() + (size_t)ch * 256;
        if (++ch == C) ch = 0;
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        const uint32_t target = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000ull - 1) / span);
        int s = sd[target >> 8];
        while (row[s + 1] <= target) ++s;              // row[top+1] = 0x10000 > target: no bound check needed
        sym[i] = (int16_t)s;
        if (i == st.first + st.count - 1) break;
        high = (low - 1) + (uint32_t)((span * row[s + 1]) >> 16);
        low = low + (uint32_t)((span * row[s]) >> 16);
        const int nshare = clz32(low ^ high);
        if (nshare) { low <<= nshare; high = (high << nshare) | ((1u << nshare) - 1u); value = (value << nshare) | src.take(nshare); }
        while (low >= 0x40000000u && high < 0xC0000000u) {
            int m = clz32(~(low << 1)); const int mz = clz32(high << 1); if (mz < m) m = mz; if (m > 31) m = 31;
            low = (low << m) & 0x7FFFFFFFu; high = (high << m) | 0x80000000u | ((1u << m) - 1u);
            value = ((value << m) | src.take(m)) ^ 0x80000000u;
        }
    }
}

// AVX-512 decoder for alphabets of up to 63 symbols (the PCGCv2 latents use ~20): no division and no search loop.
// torchac picks the largest s with cdf[s] <= target; since  cdf[j] <= floor(((off + 1) * 2^16 - 1) / span)  <=>
// (span * cdf[j]) >> 16 <= off  (off = value - low), s + 1 is the number of boundaries whose scaled position
// cum(j) = (span * cdf[j]) >> 16 is <= off.  All cum(j) of the row are evaluated at once in 64-bit lanes
// (vpmuludq on span - 1 <= 2^32 - 1, plus cdf[j], keeps the product exact for span = 2^32) and counted with a mask
// popcount; lanes past the row are padded with 2^16 (cum = span > off).  The state is (low, span, off): E1/E2/E3 shift
// value and low alike, so off' = ((off - cum_lo) << t) | next t bits, and neither `value` nor `high` is ever formed.
// One branch-free bit fetch of t = nshare + m bits per symbol (a symbol has probability >= 2^-16: t <= 18).
// The 32-bit rows carry one guard entry in front and guards behind, so any boundary count 0..W (only a corrupt stream
// produces the extremes) indexes inside the row.
struct RcWideTables {
    int nvec, W, RS; std::vector<uint32_t> rows; std::vector<uint64_t> wide_store; uint64_t* wide;
    RcWideTables(const uint16_t* cdf, int C, int Lp) : nvec((Lp + 7) / 8), W(nvec * 8), RS(W + 8), rows((size_t)C * RS, 0x10000u), wide_store((size_t)C * W + 8) {
        for (int c = 0; c < C; ++c) { uint32_t* r = rows.data() + (size_t)c * RS; r[0] = 0; for (int j = 0; j < Lp - 1; ++j) r[1 + j] = cdf[(size_t)c * Lp + j]; }
        wide = (uint64_t*)(((uintptr_t)wide_store.data() + 63) & ~(uintptr_t)63);
        for (int c = 0; c < C; ++c) for (int j = 0; j < W; ++j) wide[(size_t)c * W + j] = j < Lp - 1 ? cdf[(size_t)c * Lp + j] : 0x10000u;
    }
};
// K = 1, 2 or 4 segments per call: a symbol's decode is one dependency chain of ~40 cycles (broadcast, multiply, compare, mask count,
// table row, two leading-zero counts, shifts) that leaves most of the core idle; K segments advanced in lock step are K independent
// chains in the same loop (two: 1.8x the symbols per thread and second; four, measured: 1.34x — the dispatcher uses one or two).
struct RcWideState { SourceBF src; uint32_t low; uint64_t span; uint32_t off; int ch; int64_t i; };
template <int K>
__attribute__((target("avx512f,avx512bw,avx512dq,popcnt,lzcnt,bmi,bmi2")))
static void rc_decode_avx512_segs(const RcWideTables& tb, int C, const uint8_t* padded, int64_t limit, const RcStart* st /*[n_st <= K]*/, int n_st, int16_t* sym) {
    const int nvec = tb.nvec, W = tb.W, RS = tb.RS;
    const uint32_t* const rows = tb.rows.data(); const uint64_t* const wide = tb.wide;
    auto open = [&](const RcStart& s0) __attribute__((always_inline, target("avx512f,avx512bw,avx512dq,popcnt,lzcnt,bmi,bmi2"))) {
        const uint64_t start = s0.bitpos + 32;
        RcWideState S{SourceBF{padded, limit, (int64_t)(start >> 3)}, s0.low, s0.span, s0.off, (int)(s0.first % C), s0.first};
        S.src.refill();
        (void)S.src.take((int)(start & 7));
        return S;
    };
    auto step = [&](RcWideState& S) __attribute__((always_inline, target("avx512f,avx512bw,avx512dq,popcnt,lzcnt,bmi,bmi2"))) {
        const uint32_t* row = rows + (size_t)S.ch * RS; const uint64_t* wr = wide + (size_t)S.ch * W;
        if (++S.ch == C) S.ch = 0;
        S.src.refill();
        const __m512i vs = _mm512_set1_epi64((long long)(S.span - 1)), voff = _mm512_set1_epi64((long long)(uint64_t)S.off);
        unsigned cnt = 0;                                                          // = s + 1
        for (int v = 0; v < nvec; ++v) {
            const __m512i r = _mm512_load_si512((const void*)(wr + 8 * v));
            const __m512i cum = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(vs, r), r), 16);
            cnt += (unsigned)__builtin_popcount((unsigned)_mm512_cmple_epu64_mask(cum, voff));
        }
        sym[S.i++] = (int16_t)((int)cnt - 1);
        const uint32_t c_lo = (uint32_t)((S.span * row[cnt]) >> 16), c_hi = (uint32_t)((S.span * row[cnt + 1]) >> 16);   // cdf[s], cdf[s + 1]
        const uint32_t lo = S.low + c_lo, hi = S.low + c_hi - 1;
        const int nshare = (int)_lzcnt_u32(lo ^ hi);
        const int t = (int)_lzcnt_u32((((~lo | hi) << 1) | 1u) & (0xFFFFFFFFu >> nshare));       // (as in rc_encode_body)
        S.low = (uint32_t)((uint64_t)lo << t) & 0x7FFFFFFFu;
        S.span = (uint64_t)(c_hi - c_lo) << t;
        S.off = (uint32_t)((uint64_t)(S.off - c_lo) << t) | S.src.take(t);
    };
    static_assert(K == 1 || K == 2 || K == 4, "chains per call");
    if (n_st < K) {                                            // (the last task of an odd division)
        if constexpr (K > 1) { rc_decode_avx512_segs<K / 2>(tb, C, padded, limit, st, n_st < K / 2 ? n_st : K / 2, sym);
                               if (n_st > K / 2) rc_decode_avx512_segs<K / 2>(tb, C, padded, limit, st + K / 2, n_st - K / 2, sym); }
        return;
    }
    RcWideState S[K]; int64_t left[K]; int64_t both = st[0].count;
    for (int k = 0; k < K; ++k) { S[k] = open(st[k]); left[k] = st[k].count; both = std::min(both, left[k]); }
    for (int64_t j = 0; j < both; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) step(S[k]);
    }
    for (int k = 0; k < K; ++k)                                // (segments differ by a point or two)
        for (int64_t j = both; j < left[k]; ++j) step(S[k]);
}

// Lane-parallel form (round 4): EIGHT segments of one stream advance together, one per 64-bit lane of a zmm register, on ONE thread.
// The per-symbol dependency chain of the form above (~40 cycles, of which the core's vector units are busy for a few) becomes a
// throughput problem: every boundary of the channel's row is broadcast and compared against all eight lanes' offsets at once
// (segments start at row boundaries, so all lanes decode the same channel in the same step), the two table entries of the decoded
// symbols come from one gather each, the renormalisation is the scalar formula in vplzcntq / variable shifts, and each lane reads its
// next t bits from its own position of the stream through one 8-byte gather.  A rank whose CPU budget is one or two threads (eight
// ranks on a 16-CPU quota: pcgcv2_amd.configure_host_threads) decodes the 150 k symbols of a vox10 frame in ~0.2 ms instead of the
// 1.3 ms of the serial form, with no helper threads to wake.  G = 1 or 2 groups of eight lanes per loop (two: independent chains).
// Same arithmetic per lane as rc_decode_avx512_segs (tested against it and against the bit-serial oracle).
struct RcLaneGroup { __m512i low, span, off, pos, out; };      // out: index of the next symbol of each lane
template <int G>
__attribute__((target("avx512f,avx512bw,avx512dq,avx512cd,avx512vl,popcnt,lzcnt,bmi,bmi2")))
static void rc_decode_avx512_lanes(const RcWideTables& tb, int C, int Lp, const uint8_t* padded, int64_t limit, const RcStart* st /*[8 G]*/, int16_t* sym) {
    const int W = tb.W, RS = tb.RS, NB = Lp - 1;               // NB real boundaries per row (the last one is pinned to 2^16: never counted)
    const uint32_t* const rows = tb.rows.data(); const uint64_t* const wide = tb.wide;
    const __m512i m32 = _mm512_set1_epi64(0xFFFFFFFFll), m31 = _mm512_set1_epi64(0x7FFFFFFFll), one = _mm512_set1_epi64(1), c32 = _mm512_set1_epi64(32),
                  c63 = _mm512_set1_epi64(63), c7 = _mm512_set1_epi64(7), vlimit = _mm512_set1_epi64(limit);
    const __m512i bswap = _mm512_set_epi8(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7,
                                          8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
    RcLaneGroup L[G];
    int64_t steps = st[0].count;
    for (int g = 0; g < G; ++g) {
        alignas(64) uint64_t lo[8], sp[8], of[8], ps[8], ou[8];
        for (int k = 0; k < 8; ++k) {
            const RcStart& s0 = st[8 * g + k];
            lo[k] = s0.low; sp[k] = s0.span; of[k] = s0.off; ps[k] = s0.bitpos + 32; ou[k] = (uint64_t)s0.first;
            steps = std::min(steps, s0.count);
        }
        L[g] = RcLaneGroup{_mm512_load_si512(lo), _mm512_load_si512(sp), _mm512_load_si512(of), _mm512_load_si512(ps), _mm512_load_si512(ou)};
    }
    int ch = 0;                                                // every segment starts at a row boundary
    for (int64_t j = 0; j < steps; ++j) {
        const uint64_t* wr = wide + (size_t)ch * W;
        const uint32_t* row = rows + (size_t)ch * RS;
        if (++ch == C) ch = 0;
        __m512i vs[G], cnt[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { vs[g] = _mm512_sub_epi64(L[g].span, one); cnt[g] = _mm512_setzero_si512(); }
        for (int b = 0; b < NB; ++b) {                         // cnt = s + 1 = boundaries whose scaled position is <= off
            const __m512i r = _mm512_set1_epi64((long long)wr[b]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const __m512i cum = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(vs[g], r), r), 16);
                cnt[g] = _mm512_mask_add_epi64(cnt[g], _mm512_cmple_epu64_mask(cum, L[g].off), cnt[g], one);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            alignas(64) uint64_t cv[8], ov[8];
            _mm512_store_si512(cv, cnt[g]); _mm512_store_si512(ov, L[g].out);
            for (int k = 0; k < 8; ++k) sym[ov[k]] = (int16_t)((int)cv[k] - 1);
            L[g].out = _mm512_add_epi64(L[g].out, one);
            // cdf[s], cdf[s + 1] of every lane (rows carry a guard entry in front and guards behind: any count 0..W indexes inside)
            const __m512i rlo = _mm512_cvtepu32_epi64(_mm512_i64gather_epi32(cnt[g], (const void*)row, 4));
            const __m512i rhi = _mm512_cvtepu32_epi64(_mm512_i64gather_epi32(_mm512_add_epi64(cnt[g], one), (const void*)row, 4));
            const __m512i c_lo = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(vs[g], rlo), rlo), 16);
            const __m512i c_hi = _mm512_srli_epi64(_mm512_add_epi64(_mm512_mul_epu32(vs[g], rhi), rhi), 16);
            const __m512i lo = _mm512_and_si512(_mm512_add_epi64(L[g].low, c_lo), m32);
            const __m512i hi = _mm512_and_si512(_mm512_sub_epi64(_mm512_add_epi64(L[g].low, c_hi), one), m32);
            const __m512i nshare = _mm512_sub_epi64(_mm512_lzcnt_epi64(_mm512_xor_si512(lo, hi)), c32);
            __m512i a = _mm512_or_si512(_mm512_andnot_si512(lo, m32), hi);                       // ~lo | hi  (32 bits)
            a = _mm512_and_si512(_mm512_or_si512(_mm512_slli_epi64(a, 1), one), m32);
            a = _mm512_and_si512(a, _mm512_srlv_epi64(m32, nshare));
            const __m512i t = _mm512_sub_epi64(_mm512_lzcnt_epi64(a), c32);
            // the next t bits of every lane's own stream position (big-endian bit order, as SourceBF)
            const __m512i byte = _mm512_min_epu64(_mm512_srli_epi64(L[g].pos, 3), vlimit);
            __m512i w = _mm512_shuffle_epi8(_mm512_i64gather_epi64(byte, (const void*)padded, 1), bswap);
            w = _mm512_sllv_epi64(w, _mm512_and_si512(L[g].pos, c7));
            const __m512i bits = _mm512_srlv_epi64(_mm512_srli_epi64(w, 1), _mm512_sub_epi64(c63, t));
            L[g].low = _mm512_and_si512(_mm512_sllv_epi64(lo, t), m31);
            L[g].span = _mm512_sllv_epi64(_mm512_sub_epi64(c_hi, c_lo), t);
            L[g].off = _mm512_or_si512(_mm512_and_si512(_mm512_sllv_epi64(_mm512_and_si512(_mm512_sub_epi64(L[g].off, c_lo), m32), t), m32), bits);
            L[g].pos = _mm512_add_epi64(L[g].pos, t);
        }
    }
    // segments differ by a row or two: every lane finishes its own tail on the one-segment form, from the lane's state
    for (int g = 0; g < G; ++g) {
        alignas(64) uint64_t lo[8], sp[8], of[8], ps[8];
        _mm512_store_si512(lo, L[g].low); _mm512_store_si512(sp, L[g].span); _mm512_store_si512(of, L[g].off); _mm512_store_si512(ps, L[g].pos);
        for (int k = 0; k < 8; ++k) {
            const RcStart& s0 = st[8 * g + k];
            if (s0.count > steps) {
                const RcStart rest{ps[k] - 32, (uint32_t)lo[k], sp[k], (uint32_t)of[k], s0.first + steps, s0.count - steps};
                rc_decode_avx512_segs<1>(tb, C, padded, limit, &rest, 1, sym);
            }
        }
    }
}

// ---- a small persistent pool for the indexed decoder (segments of one stream decoded side by side)
namespace {
class SegmentPool {
    std::vector<std::thread> workers; std::mutex m; std::condition_variable work, done;
    std::mutex serial;                                 // one run() at a time per pool
    const std::function<void(int)>* job = nullptr; int n_tasks = 0; std::atomic<int> next{0}; int active = 0; bool stop = false;
    std::atomic<uint64_t> gen{0};                      // bumped by run() and prewake(); changes under the mutex, read by spinning workers without it
    std::atomic<int64_t> warm_until{0};                // steady-clock ns until which idle workers spin instead of sleeping (prewake)
    int wanted = 0, joined = 0;                        // helpers this run asked for / workers that have joined it (tickets)
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void drain(const std::function<void(int)>& fn, std::atomic<int>& next, int n) { for (int i; (i = next.fetch_add(1)) < n;) fn(i); }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            work.wait(lk, [&] { return stop || gen.load(std::memory_order_relaxed) != seen; });
            if (stop) return;
            seen = gen.load(std::memory_order_relaxed);
            const std::function<void(int)>* fn = job;  // null: a prewake, or a run() that has finished already (a worker that wakes late has
            if (!fn || joined >= wanted) {             // nothing to do, and run() does not wait for it) / the run asked for fewer helpers
                lk.unlock();
                // stay awake for the work that was announced: a thread asleep on the condition variable takes 50-150 us to run again
                // (measured: the caller had decoded two or three segments before its helpers arrived); a spinning one takes none
                while (now_ns() < warm_until.load(std::memory_order_relaxed) && gen.load(std::memory_order_acquire) == seen) _mm_pause();
                continue;
            }
            ++joined;
            const int n = n_tasks;
            ++active;
            lk.unlock();
            drain(*fn, next, n);
            lk.lock();
            if (--active == 0) done.notify_one();
        }
    }
    void grow(int helpers) { while ((int)workers.size() < helpers) workers.emplace_back([this] { loop(); }); }
public:
    ~SegmentPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } work.notify_all(); for (auto& t : workers) t.join(); }
    // run fn(0..tasks-1) on the calling thread plus up to `helpers` pool threads
    void run(int tasks, int helpers, const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> one(serial);
        {
            std::lock_guard<std::mutex> lk(m);
            grow(helpers);
            job = &fn; n_tasks = tasks; next.store(0); wanted = helpers; joined = 0; gen.fetch_add(1, std::memory_order_release);
        }
        work.notify_all();
        drain(fn, next, tasks);                        // returns once every task has been taken
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return active == 0; });    // ... and the ones taken by workers are finished
        job = nullptr;
    }
    // work for up to `helpers` pool threads is about to arrive (within `us` microseconds): wake them now and let them spin for it
    void prewake(int helpers, int us) {
        if (helpers <= 0) return;
        if (!serial.try_lock()) return;                // a run is in progress: the workers are awake anyway
        {
            std::lock_guard<std::mutex> lk(m);
            grow(helpers);
            warm_until.store(now_ns() + (int64_t)us * 1000, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
        }
        serial.unlock();
        work.notify_all();
    }
};
SegmentPool& segment_pool() { static SegmentPool p; return p; }     // range decoder
SegmentPool& octree_pool() { static SegmentPool p; return p; }      // coordinate codec: its own threads, the two decode side by side
int g_rc_threads = 0;                                  // 0 = automatic: min(8, hardware threads)
// CPUs this process may really use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota/period) and the affinity mask, not the
// advertised core count (a 16-CPU container on a 256-thread host)
int effective_cpus() {
    static int cached = 0;
    if (cached) return cached;
    unsigned hw = std::thread::hardware_concurrency();
    long n = hw ? (long)hw : 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
    long quota = -1, period = -1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atol(q);
        std::fclose(f);
    } else {
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%ld", &quota) != 1) quota = -1; std::fclose(g); }
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%ld", &period) != 1) period = -1; std::fclose(g); }
    }
    if (quota > 0 && period > 0) { const long c = (quota + period - 1) / period; if (c >= 1 && c < n) n = c; }
    cached = (int)(n < 1 ? 1 : n);
    return cached;
}
thread_local bool tl_item_worker = false;                // set on the threads that code the items of a batch side by side (pcgc_items_*)
int rc_threads() {
    if (tl_item_worker) return 1;                          // the parallelism is across items there: no nested pools
    if (g_rc_threads > 0) return g_rc_threads;
    return std::min(8, effective_cpus());
}
int g_rc_lanes = -1;                                   // -1 automatic (thread budget <= 2), 0 never, 1 always: the lane-parallel decoder
bool rc_lanes_wanted(int threads) { return g_rc_lanes > 0 || (g_rc_lanes < 0 && threads <= 2); }
bool rc_use_avx512(int Lp) {
    return g_rc_impl == 0 && Lp <= 64 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
           __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("lzcnt");
}
// decode the given starts (disjoint symbol ranges of one stream) with up to `threads` threads
int rc_decode_starts(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n, std::vector<RcStart> starts,
                     int threads) {
    static thread_local std::vector<uint8_t> padded_store;                    // kept per calling thread: no allocation per frame
    if (padded_store.size() < (size_t)nbytes + 64) padded_store.resize((size_t)nbytes + 64);
    uint8_t* const padded = padded_store.data();
    std::memcpy(padded, in, (size_t)nbytes);
    std::memset(padded + nbytes, 0, 64);
    const int64_t limit = nbytes + 8;
    for (RcStart& st : starts)                                                // at the start of the stream, off = value = its first 32 bits
        if (st.first == 0) st.off = ((uint32_t)padded[0] << 24) | ((uint32_t)padded[1] << 16) | ((uint32_t)padded[2] << 8) | padded[3];
    const bool wide = rc_use_avx512(Lp);
    RcWideTables* wt = wide ? new RcWideTables(cdf, C, Lp) : nullptr;
    RcScalarTables* stb = wide ? nullptr : new RcScalarTables(cdf, C, Lp);
    // a thread budget of one or two (pcgc_set_rc_threads / a rank's share of a small CPU quota), or pcgc_set_rc_lanes(1): eight
    // segments per zmm register on the calling thread (+ one helper for a second group of eight), no pool of segment threads
    if (wide && rc_lanes_wanted(threads) && (int)starts.size() >= 8 && __builtin_cpu_supports("avx512cd") && __builtin_cpu_supports("avx512vl")) {
        bool rows_ok = true;
        for (const RcStart& st : starts) rows_ok = rows_ok && st.first % C == 0 && st.count > 0;
        if (rows_ok) {
            const int ngroups = (int)starts.size() / 8;                      // whole groups of eight; the remaining segments one by one
            const std::function<void(int)> grp = [&](int k) {
                if (k < ngroups / 2) rc_decode_avx512_lanes<2>(*wt, C, Lp, padded, limit, &starts[(size_t)16 * k], sym);
                else if (k == ngroups / 2 && (ngroups & 1)) rc_decode_avx512_lanes<1>(*wt, C, Lp, padded, limit, &starts[(size_t)16 * (ngroups / 2)], sym);
                else { const int first = 8 * ngroups + (k - (ngroups + 1) / 2); rc_decode_avx512_segs<1>(*wt, C, padded, limit, &starts[(size_t)first], 1, sym); }
            };
            const int tasks = (ngroups + 1) / 2 + ((int)starts.size() - 8 * ngroups);
            if (threads >= 2 && ngroups >= 2) {
                // two threads: one group of eight lanes each
                const std::function<void(int)> half = [&](int k) {
                    if (k < ngroups) rc_decode_avx512_lanes<1>(*wt, C, Lp, padded, limit, &starts[(size_t)8 * k], sym);
                    else rc_decode_avx512_segs<1>(*wt, C, padded, limit, &starts[(size_t)(8 * ngroups + (k - ngroups))], 1, sym);
                };
                segment_pool().run(ngroups + ((int)starts.size() - 8 * ngroups), 1, half);
            } else {
                for (int k = 0; k < tasks; ++k) grp(k);
            }
            delete wt; delete stb;
            return 0;
        }
    }
    // more segments than threads (and the vector decoder): two or four segments per task, advanced in lock step
    const int nseg = (int)starts.size();
    const int th = std::max(threads, 1);
    const int per = !wide || nseg <= th ? 1 : 2;           // (four chains per loop are SLOWER than two: 0.53 vs 0.46 ms on two threads)
    const std::function<void(int)> one = [&](int k) {
        const int first = per * k, n_here = std::min(per, nseg - first);
        if (!wide) rc_decode_scalar_seg(*stb, C, Lp, padded, limit, starts[(size_t)k], sym);
        else if (per == 1) rc_decode_avx512_segs<1>(*wt, C, padded, limit, &starts[(size_t)first], n_here, sym);
        else if (per == 2) rc_decode_avx512_segs<2>(*wt, C, padded, limit, &starts[(size_t)first], n_here, sym);
        else rc_decode_avx512_segs<4>(*wt, C, padded, limit, &starts[(size_t)first], n_here, sym);
    };
    const int tasks = (nseg + per - 1) / per;
    if (tasks <= 1 || threads <= 1) { for (int k = 0; k < tasks; ++k) one(k); }
    else segment_pool().run(tasks, std::min(threads, tasks) - 1, one);
    delete wt; delete stb;
    return 0;
}
}
extern "C" int pcgc_set_rc_threads(int threads) { if (threads < 0) return -1; g_rc_threads = threads; return 0; }
extern "C" int pcgc_set_rc_lanes(int mode) { if (mode < -1 || mode > 1) return -1; g_rc_lanes = mode; return 0; }

extern "C" int pcgc_rc_decode(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n) {
    if (n <= 0) return 0;
    return rc_decode_starts(cdf, C, Lp, in, nbytes, sym, n, {RcStart{0, 0, 1ull << 32, 0, 0, n}}, 1);
}
// Decode with the index of pcgc_rc_encode_indexed: the segments between checkpoints are independent once the decoder state at
// their first symbol is known, so they are decoded side by side (pcgc_set_rc_threads; default min(8, hardware threads)).
// The result is the same as pcgc_rc_decode's on the same stream; a malformed index is rejected (-2), never trusted blindly
// with respect to memory: every segment writes only its own symbol range.
extern "C" int pcgc_rc_decode_indexed(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n,
                                      int n_ckpt, const uint32_t* ckpt) {
    if (n <= 0) return 0;
    if (n_ckpt < 0 || (n_ckpt > 0 && !ckpt) || C < 1) { pcgc_set_error("rc_decode_indexed: bad arguments"); return -2; }
    const RcCkpt* ck = (const RcCkpt*)ckpt;
    std::vector<RcStart> starts;
    for (int c = 0; c < n_ckpt; ++c) {
        if (ck[c].sym == 0xFFFFFFFFu) break;                                   // unused tail entries
        const int64_t first = ck[c].sym;
        const uint64_t bitpos = ((uint64_t)ck[c].bitpos_hi << 32) | ck[c].bitpos_lo;
        if (first >= n || first % C != 0 || (starts.empty() ? first != 0 : first <= starts.back().first) || bitpos > (uint64_t)nbytes * 8 + 64) {
            pcgc_set_error("rc_decode_indexed: checkpoint %d does not fit this stream (first symbol %lld, bit %llu)", c, (long long)first, (unsigned long long)bitpos);
            return -2;
        }
        if (!starts.empty()) starts.back().count = first - starts.back().first;
        starts.push_back(RcStart{bitpos, ck[c].low, (uint64_t)ck[c].span_m1 + 1, ck[c].off, first, n - first});
    }
    if (starts.empty()) starts.push_back(RcStart{0, 0, 1ull << 32, 0, 0, n});
    if (starts[0].bitpos != 0 || starts[0].low != 0 || starts[0].span != (1ull << 32)) {
        pcgc_set_error("rc_decode_indexed: the first checkpoint is not the initial coder state");
        return -2;
    }
    return rc_decode_starts(cdf, C, Lp, in, nbytes, sym, n, starts, rc_threads());
}

// ------------------------------------------------------------------------------------------------ octree codec
// Breadth-first occupancy octree over the Morton-sorted points; each node's 8-bit child occupancy is coded as 8 binary
// decisions with an adaptive binary range coder (12-bit probabilities, carry-propagating 32-bit range, LZMA-style);
// the contexts (struct OctCoder) are neighbour-occupancy based, which is also what G-PCC exploits.
namespace {

struct BinEnc {
    std::vector<uint8_t> out; uint64_t low = 0; uint32_t range = 0xFFFFFFFFu; uint8_t cache = 0; int64_t cache_size = 1;
    void shift_low() {
        if ((uint32_t)low < 0xFF000000u || (low >> 32) != 0) {
            uint8_t carry = (uint8_t)(low >> 32);
            uint8_t temp = cache;
            do { out.push_back((uint8_t)(temp + carry)); temp = 0xFF; } while (--cache_size != 0);
            cache = (uint8_t)((uint32_t)low >> 24);
        }
        ++cache_size;
        low = (uint32_t)low << 8;
    }
    void encode(uint16_t& p, int bit, int shift = 4) {
        uint32_t bound = (range >> 12) * p;
        if (bit == 0) { range = bound; p += (4096 - p) >> shift; }
        else { low += bound; range -= bound; p -= p >> shift; }
        while (range < (1u << 24)) { range <<= 8; shift_low(); }
    }
    void finish() { for (int i = 0; i < 5; ++i) shift_low(); }
};
struct BinDec {
    const uint8_t* in; int64_t len; int64_t pos = 0; uint32_t range = 0xFFFFFFFFu, code = 0;
    uint8_t next() { return pos < len ? in[pos++] : 0; }
    void init() { next(); for (int i = 0; i < 4; ++i) code = (code << 8) | next(); }
    int decode(uint16_t& p, int shift = 4) {
        uint32_t bound = (range >> 12) * p;
        int bit;
        if (code < bound) { range = bound; p += (4096 - p) >> shift; bit = 0; }
        else { code -= bound; range -= bound; p -= p >> shift; bit = 1; }
        while (range < (1u << 24)) { range <<= 8; code = (code << 8) | next(); }
        return bit;
    }
};

inline uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    auto spread = [](uint64_t v) {
        v &= 0x1FFFFF;
        v = (v | v << 32) & 0x1F00000000FFFFull; v = (v | v << 16) & 0x1F0000FF0000FFull;
        v = (v | v << 8) & 0x100F00F00F00F00Full; v = (v | v << 4) & 0x10C30C30C30C30C3ull;
        v = (v | v << 2) & 0x1249249249249249ull; return v;
    };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}
inline void demorton3(uint64_t m, int32_t& x, int32_t& y, int32_t& z) {
    auto compact = [](uint64_t v) {
        v &= 0x1249249249249249ull;
        v = (v ^ (v >> 2)) & 0x10C30C30C30C30C3ull; v = (v ^ (v >> 4)) & 0x100F00F00F00F00Full;
        v = (v ^ (v >> 8)) & 0x1F0000FF0000FFull; v = (v ^ (v >> 16)) & 0x1F00000000FFFFull;
        v = (v ^ (v >> 32)) & 0x1FFFFF; return (int32_t)v;
    };
    x = compact(m); y = compact(m >> 1); z = compact(m >> 2);
}

// ascending sort of keys below 2^bits: LSD radix passes of 11 bits (std::sort is 60 % of the time to code the 18.7 k stride-8 voxels
// of a vox10 frame: 21-bit keys, two passes here)
void sort_codes(std::vector<uint64_t>& v, int bits) {
    if (v.size() < 512 || bits > 44) { std::sort(v.begin(), v.end()); return; }
    static thread_local std::vector<uint64_t> tmp;               // (kept per thread: a fresh 150 KB vector is an mmap and its page faults)
    if (tmp.size() < v.size()) tmp.resize(v.size());
    uint64_t* src = v.data(); uint64_t* dst = tmp.data();
    for (int shift = 0; shift < bits; shift += 11) {
        uint32_t count[2048] = {0};
        for (size_t i = 0; i < v.size(); ++i) ++count[(src[i] >> shift) & 2047];
        uint32_t at = 0;
        for (int b = 0; b < 2048; ++b) { const uint32_t c = count[b]; count[b] = at; at += c; }
        for (size_t i = 0; i < v.size(); ++i) dst[count[(src[i] >> shift) & 2047]++] = src[i];
        std::swap(src, dst);
    }
    if (src != v.data()) std::memcpy(v.data(), src, v.size() * sizeof(uint64_t));
}

// Occupancy of one octree level: a Morton-indexed bitmap while the level has at most 2^24 cells (2 MiB), binary search in
// the sorted node list beyond that.  The bitmaps are two per-thread buffers that are ALL ZERO whenever no stream is being coded:
// a level's bits are cleared again by walking its node list, never by a fill (a fresh 256 KB vector per level and group — the mmap,
// its page faults and the zero fill — was a third of the time to code the 18.7 k stride-8 voxels of a vox10 frame, and the page
// faults of eight threads serialise in the kernel).
struct OccBuffers { std::vector<uint64_t> buf[2]; };
inline OccBuffers& occ_buffers() { static thread_local OccBuffers b; return b; }
struct LevelOcc {
    const std::vector<uint64_t>* nodes = nullptr; int level_bits = 0; bool use_bitmap = false;
    uint64_t* bits = nullptr;
    void begin(const std::vector<uint64_t>& n, int lb, std::vector<uint64_t>& storage) {
        nodes = &n; level_bits = lb; use_bitmap = 3 * lb <= 24;
        if (use_bitmap) {
            const size_t words = ((size_t)1 << (3 * lb)) / 64 + 1;
            if (storage.size() < words) storage.resize(words, 0);        // (grows with zeros; what was there is zero by the invariant)
            bits = storage.data();
        }
    }
    inline void mark(uint64_t c) { if (use_bitmap) bits[c >> 6] |= 1ull << (c & 63); }
    inline void fill(const std::vector<uint64_t>& n) { if (use_bitmap) for (uint64_t c : n) bits[c >> 6] |= 1ull << (c & 63); }
    inline void wipe(const std::vector<uint64_t>& n) { if (use_bitmap) for (uint64_t c : n) bits[c >> 6] = 0; }
    inline bool has(uint64_t c) const {
        return use_bitmap ? ((bits[c >> 6] >> (c & 63)) & 1ull) != 0 : std::binary_search(nodes->begin(), nodes->end(), c);
    }
    inline int lim() const { return level_bits >= 21 ? INT32_MAX : (1 << level_bits); }
};

// Context model shared by encoder and decoder.  For child j = (jx,jy,jz) of node P the context is
//   * the level bucket (distance from the leaves, 0..3),
//   * per axis, whether the child's OUTWARD neighbour is occupied: on the low side (j_axis = 0) that neighbour is a child of
//     P - e_axis, which precedes P in Morton order and is therefore already coded (exact child-level knowledge); on the high
//     side (j_axis = 1) it belongs to P + e_axis whose children are not known yet, so the parent-level occupancy is used,
//   * the position in the node's byte: child index and how many of the earlier siblings are occupied,
//   * the number of occupied face neighbours of P (0..6), coarsened to 0 / 1-2 / 3-4 / 5-6.
// Surfaces are locally connected, so "is there something right next to this cell" is by far the strongest predictor.
constexpr uint8_t kOctVersion = 2;
struct OctCoder {
    static constexpr int kBuckets = 4, kAxis = 8, kPos = 8 * 9, kNb = 4;
    std::vector<uint16_t> prob;
    LevelOcc parent, child;
    OctCoder() : prob((size_t)kBuckets * kAxis * kPos * kNb, 2048) {}
    int bucket = 0, nb_class = 0;
    bool has_hi[3] = {false, false, false};          // parent-level neighbour on the +x / +y / +z side exists
    bool lo_ok[3] = {false, false, false};           // parent has a neighbour slot on the -x / -y / -z side (inside the cube)
    uint64_t lo_code[3] = {0, 0, 0};                 // Morton code of P - e_axis (its children are already coded)
    // `nodes`: the level's nodes (their bits are set here); `next`: the level under construction (marked as its nodes are coded).
    // end_level() leaves both buffers all zero again.
    void begin_level(const std::vector<uint64_t>& nodes, const std::vector<uint64_t>& next, int lvl, int depth) {
        bucket = std::min(kBuckets - 1, depth - 1 - lvl);
        OccBuffers& ob = occ_buffers();
        parent.begin(nodes, lvl, ob.buf[0]);
        parent.fill(nodes);
        child.begin(next, lvl + 1, ob.buf[1]);
    }
    void end_level() { parent.wipe(*parent.nodes); child.wipe(*child.nodes); }
    // The six face neighbours of P in Morton space: a step along one axis is an add / subtract on that axis' dilated bits (the
    // carry runs through the other axes' positions when they are filled with ones / zeros) — no de- and re-interleaving.
    void begin_node(uint64_t node) {
        constexpr uint64_t kX = 0x1249249249249249ull;
        const uint64_t level_mask = parent.level_bits >= 21 ? ~0ull >> 1 : ((1ull << (3 * parent.level_bits)) - 1);
        int cnt = 0;
        for (int a = 0; a < 3; ++a) {
            const uint64_t M = (kX << a) & level_mask, own = node & M, rest = node & ~M;
            lo_ok[a] = own != 0;
            if (lo_ok[a]) { lo_code[a] = ((own - 1) & M) | rest; cnt += parent.has(lo_code[a]); }
            has_hi[a] = false;
            if (own != M) { has_hi[a] = parent.has((((own | ~M) + 1) & M) | rest); cnt += has_hi[a]; }
        }
        nb_class = (cnt + 1) / 2;                                   // 0, 1-2, 3-4, 5-6
        // the three axis bits of every child's context at once: the level under construction, indexed by child code, IS a byte per
        // node (bit j = child j), so the low-side neighbours' children come as one byte each
        if (child.use_bitmap) {
            const uint8_t* occ = (const uint8_t*)child.bits;
            for (int j = 0; j < 8; ++j) axis_of[j] = 0;
            for (int a = 0; a < 3; ++a) {
                const int bit = 1 << a;
                const unsigned lo = lo_ok[a] ? occ[lo_code[a]] : 0u;
                for (int j = 0; j < 8; ++j) axis_of[j] |= (uint8_t)(((j & bit) ? (unsigned)has_hi[a] : ((lo >> (j | bit)) & 1u)) << a);
            }
        }
    }
    uint8_t axis_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // outward neighbour of child j along axis a: high side -> parent-level knowledge; low side -> the child (j | bit_a) of
    // P - e_a, looked up in the level under construction
    inline bool outward(int j, int a) const {
        const int bit = 1 << a;
        if (j & bit) return has_hi[a];
        return lo_ok[a] && child.has((lo_code[a] << 3) | (uint64_t)(j | bit));
    }
    inline uint16_t& ctx(int j, int occupied_before) {
        const int axis = child.use_bitmap ? (int)axis_of[j] : ((int)outward(j, 0) | ((int)outward(j, 1) << 1) | ((int)outward(j, 2) << 2));
        last = (((size_t)bucket * kAxis + axis) * kPos + (j * 9 + occupied_before)) * kNb + nb_class;
        return prob[last];
    }
    // Model 1 (stream versions 4 / 5, round 5): every stream starts from contexts trained on a MIX of shapes (oct_prior_mixed: none of them a
    // bench cloud) and a context adapts FAST on its first visits within the stream — update shifts 2, 3, 3, then the usual 4 — so that a
    // group of a few thousand points re-centres a prior that does not fit its cloud in three visits instead of sixteen.
    bool fast = false;
    size_t last = 0;
    std::vector<uint8_t> seen;
    void enable_fast_start() { fast = true; seen.assign(prob.size(), 0); }
    inline int shift_of_last() {                                  // update shift of the context ctx() returned last
        if (!fast) return 4;
        uint8_t& c = seen[last];
        if (c >= 3) return 4;
        return c++ == 0 ? 2 : 3;
    }
    inline void clamp_last() { if (fast) { uint16_t& p = prob[last]; p = p < 32 ? 32 : (p > 4064 ? 4064 : p); } }
};

constexpr uint8_t kMagic[4] = {'P', 'C', 'G', 'O'};

}  // namespace

// ---- one range-coded occupancy stream: levels [lvl0, depth) below the given start nodes (all of level lvl0, Morton-sorted)
namespace {
// `leaves`: the Morton-sorted leaf codes that lie below `roots` (a contiguous run of the cloud's sorted leaves)
std::vector<uint8_t> oct_encode_part(const std::vector<uint64_t>& roots, int lvl0, int depth, const uint64_t* leaves, size_t n_leaves,
                                     const std::vector<uint16_t>* prior = nullptr, std::vector<uint16_t>* trained = nullptr, bool fast = false) {
    BinEnc enc; OctCoder oc;
    if (prior) oc.prob = *prior;
    if (fast) oc.enable_fast_start();
    std::vector<uint64_t> level_nodes = roots, next;
    for (int lvl = lvl0; lvl < depth && !level_nodes.empty(); ++lvl) {
        const int shift = 3 * (depth - 1 - lvl);                     // leaves >> shift = child code at level lvl+1
        next.clear();
        next.reserve(level_nodes.size() * 2);
        oc.begin_level(level_nodes, next, lvl, depth);
        size_t cursor = 0;
        for (uint64_t node : level_nodes) {
            unsigned occ = 0;
            while (cursor < n_leaves && ((leaves[cursor] >> shift) >> 3) == node) { occ |= 1u << ((leaves[cursor] >> shift) & 7); ++cursor; }
            oc.begin_node(node);
            int before = 0;
            for (int j = 0; j < 8; ++j) {
                const int bit = (occ >> j) & 1;
                uint16_t& p = oc.ctx(j, before);
                enc.encode(p, bit, oc.shift_of_last());
                oc.clamp_last();
                if (bit) { const uint64_t c = (node << 3) | (uint64_t)j; next.push_back(c); oc.child.mark(c); ++before; }
            }
        }
        oc.end_level();
        level_nodes.swap(next);
    }
    enc.finish();
    if (trained) *trained = oc.prob;
    return std::move(enc.out);
}
// Version 3 starts every stream from TRAINED contexts instead of p = 1/2: the state the context model reaches after coding a fixed
// integer-defined training surface (a sphere shell of radius 45 in a 128^3 grid, 25 k voxels), computed once per process by the
// coder itself — encoder and decoder derive the same table, nothing is transmitted.  Groups of a few thousand points cannot
// afford to learn 9216 contexts from scratch (1.47 -> 2.22 bit per point on 18.7 k points without this).
const std::vector<uint16_t>& oct_prior() {
    static const std::vector<uint16_t> prior = [] {
        std::vector<uint64_t> leaves;
        for (int z = 16; z < 112; ++z) for (int y = 16; y < 112; ++y) for (int x = 16; x < 112; ++x) {
            const int d2 = (x - 64) * (x - 64) + (y - 64) * (y - 64) + (z - 64) * (z - 64);
            if (d2 >= 45 * 45 && d2 < 46 * 46) leaves.push_back(morton3((uint32_t)x, (uint32_t)y, (uint32_t)z));
        }
        std::sort(leaves.begin(), leaves.end());
        std::vector<uint16_t> trained;
        (void)oct_encode_part(std::vector<uint64_t>(1, 0), 0, 7, leaves.data(), leaves.size(), nullptr, &trained);
        return trained;
    }();
    return prior;
}
// Model 1: the prior is the state the context model reaches after coding FOUR integer-defined training shapes in a 128^3 grid one after
// the other (plain shift-4 updates) — a sphere shell, an ellipsoid shell (semi-axes 54 / 36 / 27), a tilted plane slab and a sphere shell
// with hashed drop-outs and volume salt: curved, flat and ragged neighbourhoods, none of them one of the clouds the bench or the tests
// code ("train / adapt on something that is not the benchmark shape").  tools/experiments/oct/ctx_probe.py is the offline
// evaluation behind the choice (profiles/r05_coord_codec.md).
const std::vector<uint16_t>& oct_prior_mixed() {
    static const std::vector<uint16_t> prior = [] {
        std::vector<uint16_t> state;
        for (int shape = 0; shape < 4; ++shape) {
            std::vector<uint64_t> leaves;
            for (int z = 0; z < 128; ++z) for (int y = 0; y < 128; ++y) for (int x = 0; x < 128; ++x) {
                bool in = false;
                if (shape == 0) { const int d2 = (x - 64) * (x - 64) + (y - 64) * (y - 64) + (z - 64) * (z - 64); in = d2 >= 45 * 45 && d2 < 46 * 46; }
                else if (shape == 1) { const int e = 4 * (x - 64) * (x - 64) + 9 * (y - 64) * (y - 64) + 16 * (z - 64) * (z - 64); in = e >= 108 * 108 && e < 112 * 112; }
                else if (shape == 2) { const int pl = 3 * x + 5 * y + 7 * z; in = pl >= 960 && pl < 969 && x > 8 && x < 120 && y > 8 && y < 120 && z > 8 && z < 120; }
                else {
                    const uint64_t h = (((uint64_t)x * 73856093ull) ^ ((uint64_t)y * 19349663ull) ^ ((uint64_t)z * 83492791ull)) & 1023ull;
                    const int d3 = (x - 60) * (x - 60) + (y - 66) * (y - 66) + (z - 62) * (z - 62);
                    in = (d3 >= 38 * 38 && d3 < 39 * 39 && h >= 100) || (h < 2 && d3 < 50 * 50);
                }
                if (in) leaves.push_back(morton3((uint32_t)x, (uint32_t)y, (uint32_t)z));
            }
            std::sort(leaves.begin(), leaves.end());
            std::vector<uint16_t> next;
            (void)oct_encode_part(std::vector<uint64_t>(1, 0), 0, 7, leaves.data(), leaves.size(), state.empty() ? nullptr : &state, &next);
            state.swap(next);
        }
        return state;
    }();
    return prior;
}
// -> 0, or -2 on a corrupt stream; `out` = the nodes of level `depth_to` below `roots`, Morton-sorted
int oct_decode_part(const uint8_t* in, int64_t nbytes, const std::vector<uint64_t>& roots, int lvl0, int depth_to, int depth, int64_t max_nodes,
                    std::vector<uint64_t>& out, const std::vector<uint16_t>* prior = nullptr, bool fast = false) {
    BinDec dec{in, nbytes}; dec.init();
    OctCoder oc;
    if (prior) oc.prob = *prior;
    if (fast) oc.enable_fast_start();
    std::vector<uint64_t> level_nodes = roots, next;
    for (int lvl = lvl0; lvl < depth_to && !level_nodes.empty(); ++lvl) {
        next.clear();
        next.reserve(level_nodes.size() * 2);
        oc.begin_level(level_nodes, next, lvl, depth);
        for (uint64_t node : level_nodes) {
            oc.begin_node(node);
            int before = 0;
            for (int j = 0; j < 8; ++j) {
                uint16_t& p = oc.ctx(j, before);
                const int bit = dec.decode(p, oc.shift_of_last());
                oc.clamp_last();
                if (bit) { const uint64_t c = (node << 3) | (uint64_t)j; next.push_back(c); oc.child.mark(c); ++before; }
            }
            if ((int64_t)next.size() > max_nodes) { oc.end_level(); return -2; }          // corrupt stream
        }
        oc.end_level();
        level_nodes.swap(next);
    }
    out.swap(level_nodes);
    return 0;
}
constexpr uint8_t kOctTiled = 3;
constexpr int64_t kOctTiledMin = 8192;                 // smaller clouds: one stream (version 2)
std::atomic<int> kOctGroups{8};
#ifndef PCGC_OCT_NODES_PER_GROUP
#define PCGC_OCT_NODES_PER_GROUP 2
#endif
std::atomic<int> g_oct_tiled{1};                                 // 0 = always one stream (A/B tests)
std::atomic<int> g_oct_model{1};                                 // what the ENCODER writes: 1 = versions 4 / 5 (mixed prior + fast start), 0 = the round-3 versions 2 / 3
constexpr uint8_t kOctVersion1 = 4, kOctTiled1 = 5;    // model 1: one stream / groups of subtrees (the decoder reads all four versions)
}  // namespace
// (pcgc_set_oct_model / pcgc_set_oct_tiled are A/B and test knobs: process-wide, read once per encode call — a caller that flips them while
//  another thread encodes gets one or the other container, never a mixed one; the decoder reads every version whatever they say)
extern "C" int pcgc_set_oct_model(int model) { if (model != 0 && model != 1) return -1; g_oct_model = model; return 0; }
// Builds the coder's trained priors now instead of inside the first encode / decode call of the process (4 x 128^3 voxel tests + four training
// passes: tens of milliseconds a cold decoder would otherwise pay on its first frame).  Idempotent, thread-safe (function-local statics).
extern "C" int pcgc_oct_warm(void) { (void)oct_prior(); (void)oct_prior_mixed(); return 0; }
// 0 = one stream, 1 = the default 8 groups, n > 1 = n groups (clamped to the 255 the one-byte group count of the stream can hold)
extern "C" int pcgc_set_oct_tiled(int on) { g_oct_tiled = on ? 1 : 0; kOctGroups = on > 1 ? (on > 255 ? 255 : on) : 8; return 0; }

// stream versions 4 / 5 (round 5, what the encoder writes): the layouts of versions 2 / 3 below with context MODEL 1 — every stream starts from
// the mixed-shape prior (oct_prior_mixed) and contexts adapt fast on their first visits; 2 / 3 (p = 1/2 or the sphere-trained prior, shift-4
// updates) are still decoded, and written when pcgc_set_oct_model(0) asks for them (format tests).
// stream, version 2: "PCGO" | 2 | depth u8 | n u32 | range-coded occupancy bits of the whole tree
// stream, version 3 (clouds of >= 8192 points): "PCGO" | 3 | depth u8 | n u32 | split level d u8 | groups G u8 | top bytes u32 |
//     G x (roots u32, points u32, bytes u32) | the occupancy stream of levels [0, d) | G occupancy streams of levels [d, depth)
// A group is a run of consecutive level-d nodes with about n / 8 points below them, coded from the trained contexts above with
// only its own nodes as neighbours: the groups are independent, so they are decoded (and encoded) side by side on the segment
// pool — the coordinate decode is on the decoder's critical path (0.63 -> 0.2 ms for the 18.7 k stride-8 voxels of a vox10 frame).
// Price: every group adapts the contexts to the cloud on its own, 1.47 -> 1.69 bit per point there (+0.6 % of the whole bitstream).
extern "C" int64_t pcgc_oct_encode(const int32_t* xyz, int64_t n, uint8_t* out, int64_t cap) {
    uint32_t maxc = 0;
    for (int64_t i = 0; i < 3 * n; ++i) { if (xyz[i] < 0 || xyz[i] >= (1 << 21)) return INT64_MIN; maxc = std::max(maxc, (uint32_t)xyz[i]); }
    int depth = 1; while ((1u << depth) <= maxc) ++depth;
    std::vector<uint64_t> leaves((size_t)n);
    for (int64_t i = 0; i < n; ++i) leaves[(size_t)i] = morton3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    sort_codes(leaves, 3 * depth);
    leaves.erase(std::unique(leaves.begin(), leaves.end()), leaves.end());
    const int64_t n_unique = (int64_t)leaves.size();
    const uint32_t n32 = (uint32_t)n_unique;

    // split level: the first one with at least 4 nodes per group (or the last but one)
    int d = 0; std::vector<uint64_t> split_nodes; std::vector<int64_t> split_first;      // level-d nodes and the index of their first leaf
    if (g_oct_tiled && n_unique >= kOctTiledMin && depth >= 3) {
        for (d = 1; d < depth - 1; ++d) {
            const int shift = 3 * (depth - d);
            split_nodes.clear(); split_first.clear();
            for (int64_t i = 0; i < n_unique; ++i) {
                const uint64_t c = leaves[(size_t)i] >> shift;
                if (split_nodes.empty() || split_nodes.back() != c) { split_nodes.push_back(c); split_first.push_back(i); }
            }
            if ((int)split_nodes.size() >= PCGC_OCT_NODES_PER_GROUP * kOctGroups) break;
        }
        if (d >= depth - 1 || (int)split_nodes.size() < 2) d = 0;
    }
    if (d == 0) {
        std::vector<uint64_t> root; if (n_unique > 0) root.push_back(0);
        const std::vector<uint8_t> body = g_oct_model ? oct_encode_part(root, 0, depth, leaves.data(), leaves.size(), &oct_prior_mixed(), nullptr, true)
                                                      : oct_encode_part(root, 0, depth, leaves.data(), leaves.size());
        const int64_t total = 4 + 2 + 4 + (int64_t)body.size();
        if (total > cap) return -total;
        std::memcpy(out, kMagic, 4);
        out[4] = g_oct_model ? kOctVersion1 : kOctVersion; out[5] = (uint8_t)depth;
        std::memcpy(out + 6, &n32, 4);
        std::memcpy(out + 10, body.data(), body.size());
        return total;
    }
    // groups: consecutive level-d nodes, cut when a group has reached its share of the points
    split_first.push_back(n_unique);
    std::vector<int> group_begin;                                       // index into split_nodes
    {
        const int64_t share = (n_unique + kOctGroups - 1) / kOctGroups;
        int64_t acc = 0;
        for (size_t k = 0; k < split_nodes.size(); ++k) {
            if (group_begin.empty() || acc >= share) { group_begin.push_back((int)k); acc = 0; }
            acc += split_first[k + 1] - split_first[k];
        }
    }
    const int G = (int)group_begin.size();
    group_begin.push_back((int)split_nodes.size());
    std::vector<std::vector<uint8_t>> bodies((size_t)G);
    // the top of the tree: levels [0, d), with the level-d nodes as its "leaves"
    std::vector<uint64_t> root(1, 0);
    const bool m1 = g_oct_model != 0;
    const std::vector<uint16_t>& prior = m1 ? oct_prior_mixed() : oct_prior();
    const std::vector<uint8_t> top = oct_encode_part(root, 0, d, split_nodes.data(), split_nodes.size(), &prior, nullptr, m1);
    const std::function<void(int)> one = [&](int g) {
        const std::vector<uint64_t> roots(split_nodes.begin() + group_begin[(size_t)g], split_nodes.begin() + group_begin[(size_t)g + 1]);
        const int64_t lo = split_first[(size_t)group_begin[(size_t)g]], hi = split_first[(size_t)group_begin[(size_t)g + 1]];
        bodies[(size_t)g] = oct_encode_part(roots, d, depth, leaves.data() + lo, (size_t)(hi - lo), &prior, nullptr, m1);
    };
    const int threads = rc_threads();
    if (threads <= 1) { for (int g = 0; g < G; ++g) one(g); } else octree_pool().run(G, std::min(threads, G) - 1, one);
    int64_t total = 4 + 2 + 4 + 2 + 4 + 12 * (int64_t)G + (int64_t)top.size();
    for (const auto& b : bodies) total += (int64_t)b.size();
    if (total > cap) return -total;
    uint8_t* p = out;
    std::memcpy(p, kMagic, 4); p += 4;
    *p++ = m1 ? kOctTiled1 : kOctTiled; *p++ = (uint8_t)depth;
    std::memcpy(p, &n32, 4); p += 4;
    *p++ = (uint8_t)d; *p++ = (uint8_t)G;
    { const uint32_t tb = (uint32_t)top.size(); std::memcpy(p, &tb, 4); p += 4; }
    for (int g = 0; g < G; ++g) {
        const uint32_t rec[3] = {(uint32_t)(group_begin[(size_t)g + 1] - group_begin[(size_t)g]),
                                 (uint32_t)(split_first[(size_t)group_begin[(size_t)g + 1]] - split_first[(size_t)group_begin[(size_t)g]]),
                                 (uint32_t)bodies[(size_t)g].size()};
        std::memcpy(p, rec, 12); p += 12;
    }
    std::memcpy(p, top.data(), top.size()); p += top.size();
    for (const auto& b : bodies) { std::memcpy(p, b.data(), b.size()); p += b.size(); }
    return total;
}

extern "C" int64_t pcgc_oct_decode_count(const uint8_t* in, int64_t nbytes) {
    if (nbytes < 10 || std::memcmp(in, kMagic, 4) != 0 || (in[4] != kOctVersion && in[4] != kOctTiled && in[4] != kOctVersion1 && in[4] != kOctTiled1)) return -1;
    uint32_t n32; std::memcpy(&n32, in + 6, 4);
    return (int64_t)n32;
}

namespace {
// the stream's voxels as Morton codes, ascending
int oct_decode_leaves(const uint8_t* in, int64_t nbytes, int64_t n, std::vector<uint64_t>& leaves) {
    if (pcgc_oct_decode_count(in, nbytes) != n) { pcgc_set_error("oct_decode: not a PCGO stream of %lld points", (long long)n); return -1; }
    const int depth = in[5];
    if (depth < 1 || depth > 21) { pcgc_set_error("oct_decode: bad depth %d", depth); return -1; }
    leaves.clear();
    const bool m1 = in[4] == kOctVersion1 || in[4] == kOctTiled1;                  // model 1: mixed prior + fast start (versions 4 / 5)
    if (in[4] == kOctVersion || in[4] == kOctVersion1) {
        std::vector<uint64_t> root; if (n > 0) root.push_back(0);
        if (oct_decode_part(in + 10, nbytes - 10, root, 0, depth, depth, n, leaves, m1 ? &oct_prior_mixed() : nullptr, m1)) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
    } else {
        if (nbytes < 16) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
        const int d = in[10], G = in[11];
        uint32_t top_bytes; std::memcpy(&top_bytes, in + 12, 4);
        const int64_t table = 16, payload = table + 12 * (int64_t)G;
        if (d < 1 || d >= depth || G < 1 || payload + (int64_t)top_bytes > nbytes) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
        std::vector<uint64_t> split_nodes, root(1, 0);
        const std::vector<uint16_t>& prior = m1 ? oct_prior_mixed() : oct_prior();
        if (oct_decode_part(in + payload, top_bytes, root, 0, d, d, n, split_nodes, &prior, m1)) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }      // (the top is a tree of depth d of its own)
        std::vector<int64_t> root_at((size_t)G + 1, 0), leaf_at((size_t)G + 1, 0), byte_at((size_t)G + 1, payload + top_bytes);
        for (int g = 0; g < G; ++g) {
            uint32_t rec[3]; std::memcpy(rec, in + table + 12 * g, 12);
            root_at[(size_t)g + 1] = root_at[(size_t)g] + rec[0]; leaf_at[(size_t)g + 1] = leaf_at[(size_t)g] + rec[1]; byte_at[(size_t)g + 1] = byte_at[(size_t)g] + rec[2];
            if (rec[0] == 0) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
        }
        if (root_at[(size_t)G] != (int64_t)split_nodes.size() || leaf_at[(size_t)G] != n || byte_at[(size_t)G] > nbytes) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
        leaves.assign((size_t)n, 0);
        std::vector<int> status((size_t)G, 0);
        const std::function<void(int)> one = [&](int g) {
            const std::vector<uint64_t> roots(split_nodes.begin() + root_at[(size_t)g], split_nodes.begin() + root_at[(size_t)g + 1]);
            const int64_t want = leaf_at[(size_t)g + 1] - leaf_at[(size_t)g];
            std::vector<uint64_t> got;
            if (oct_decode_part(in + byte_at[(size_t)g], byte_at[(size_t)g + 1] - byte_at[(size_t)g], roots, d, depth, depth, want, got, &prior, m1) || (int64_t)got.size() != want) { status[(size_t)g] = -2; return; }
            std::memcpy(leaves.data() + leaf_at[(size_t)g], got.data(), (size_t)want * sizeof(uint64_t));
        };
        const int threads = rc_threads();
        if (threads <= 1) { for (int g = 0; g < G; ++g) one(g); } else octree_pool().run(G, std::min(threads, G) - 1, one);
        for (int g = 0; g < G; ++g) if (status[(size_t)g]) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
    }
    if ((int64_t)leaves.size() != n) { pcgc_set_error("oct_decode: corrupt or truncated stream (line %d)", __LINE__); return -2; }
    return 0;
}
// Morton code -> (z, y, x) sort key with d bits per coordinate: three bit extractions where the CPU has them
constexpr uint64_t kMortonX = 0x1249249249249249ull;
__attribute__((target("bmi2")))
void zyx_keys_bmi2(const uint64_t* leaves, int64_t n, int d, uint64_t* keys) {
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t m = leaves[i];
        keys[i] = (_pext_u64(m, kMortonX << 2) << (2 * d)) | (_pext_u64(m, kMortonX << 1) << d) | _pext_u64(m, kMortonX);
    }
}
void zyx_keys(const uint64_t* leaves, int64_t n, int d, uint64_t* keys) {
    static const bool bmi2 = __builtin_cpu_supports("bmi2");
    if (bmi2) { zyx_keys_bmi2(leaves, n, d, keys); return; }
    for (int64_t i = 0; i < n; ++i) {
        int32_t x, y, z; demorton3(leaves[i], x, y, z);
        keys[i] = ((uint64_t)(uint32_t)z << (2 * d)) | ((uint64_t)(uint32_t)y << d) | (uint64_t)(uint32_t)x;
    }
}
}  // namespace

extern "C" int pcgc_oct_decode(const uint8_t* in, int64_t nbytes, int32_t* xyz, int64_t n) {
    std::vector<uint64_t> leaves;
    if (int rc = oct_decode_leaves(in, nbytes, n, leaves)) return rc;
    for (int64_t i = 0; i < n; ++i) demorton3(leaves[(size_t)i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return 0;
}


// ------------------------------------------------------------------------------------------------ bitstream files of several items
// The host half of Coder.encode / Coder.decode for the items of a collated batch (or one cloud), in native threads: per item the CDF
// table, the range coder, the sidecar, the octree coordinate stream and the four files.  In Python these stages are a few hundred
// microseconds of interpreter time per item that eight threads serialise on the GIL (2.7 ms for the 8 octant blocks of config 5,
// each direction); here the items really run side by side.
//   <stem>_F.bin             range-coded latent (torchac-compatible stream)                                   coder.py:49-55
//   <stem>_H.bin             int32 rows | int32 C | int8 1 | float32 min_v | float32 max_v  (17 bytes)          coder.py:51-55
//   <stem>_num_points.bin    int32[3] = N4, N2, N1                                                              coder.py:85-87
//   <stem>_C.bin             native "PCGO" octree stream of the stride-8 coordinates (gpcc.py's tmc3 stream is written by the caller)
//   <stem>_F.idx             sidecar: "PCG2" | stream bytes | stream CRC-32 | checkpoints | table CRC-32 | CRC-32(head + body) | checkpoints
namespace {
// CRC-32 (zlib's: IEEE 802.3, reflected) of the bitstream with carry-less multiplies: four 128-bit lanes folded per 64 bytes, then
// to 128, 64 and (Barrett) 32 bits — the folding constants are x^n mod P for the lane distances, from Gopal et al., "Fast CRC
// Computation for Generic Polynomials Using PCLMULQDQ" (Intel, 2009).  ~10x zlib 1.2.11's table walk (a 97 KB latent stream: 50 -> 5 us,
// once per direction and frame); the first / last bytes that do not fill 16-byte blocks go through zlib.  Same value by construction
// (tested against zlib over random lengths and seeds).
__attribute__((target("pclmul,sse4.1")))
uint32_t crc32_clmul_blocks(uint32_t crc /*pre-inverted register value*/, const uint8_t* buf, size_t len /*>= 64, multiple of 16*/) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);                                 // four lanes -> one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                                        // remaining whole blocks
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                                   // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);                                 // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
}  // namespace
// zlib's crc32(crc, buf, len), faster on long buffers
extern "C" uint32_t pcgc_crc32(uint32_t crc, const uint8_t* buf, int64_t len) {
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (len <= 0 || !buf) return crc;
    if (have && len >= 256) {
        const size_t blocks = (size_t)len & ~(size_t)15;
        crc = ~crc32_clmul_blocks(~crc, buf, blocks);
        buf += blocks; len -= (int64_t)blocks;
    }
    while (len > 0) { const uInt part = (uInt)std::min<int64_t>(len, 1 << 30); crc = (uint32_t)crc32(crc, buf, part); buf += part; len -= part; }
    return crc;
}
namespace {
// PCGC_ITEMS_TRACE=1: per-stage microseconds of every item task on stderr (diagnostics; tools/host_timeline.py shows the calls' totals)
bool items_trace() { static const bool on = [] { const char* e = std::getenv("PCGC_ITEMS_TRACE"); return e && *e && *e != '0'; }(); return on; }
struct StageClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0; std::string line;
    void mark(const char* what) {
        if (!items_trace()) return;
        const auto now = std::chrono::steady_clock::now();
        char b[64]; std::snprintf(b, sizeof b, " %s %.0f", what, std::chrono::duration<double, std::micro>(now - last).count());
        line += b; last = now;
    }
    void done(const char* task, int item) {
        if (!items_trace()) return;
        std::fprintf(stderr, "[pcgc items] %s %d:%s | total %.0f us\n", task, item, line.c_str(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
};
SegmentPool& items_pool() { static SegmentPool p; return p; }
// plain system calls: a stdio stream per file is a buffer allocation and two more calls on each of the ten small files of a frame
bool write_file(const std::string& path, const void* data, size_t n) {
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return false;
    const uint8_t* p = (const uint8_t*)data;
    bool ok = true;
    while (n > 0) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
        p += w; n -= (size_t)w;
    }
    return ::close(fd) == 0 && ok;
}
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
    out.resize((size_t)st.st_size);
    size_t got = 0;
    bool ok = true;
    while (got < out.size()) {
        const ssize_t r = ::read(fd, out.data() + got, out.size() - got);
        if (r < 0) { if (errno == EINTR) continue; ok = false; break; }
        if (r == 0) break;                                       // (shrank under us)
        got += (size_t)r;
    }
    ::close(fd);
    out.resize(got);
    return ok;
}
void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
constexpr size_t kSidecarHead = 24;
// run fn(item) for every item on up to `threads` threads; -> first failing item's code, its message set as the caller's error
int for_items(int n_items, int threads, const std::function<int(int, std::string&)>& fn, int tasks_per_item = 1) {
    std::vector<int> rc((size_t)n_items, 0);
    std::vector<std::string> msg((size_t)n_items);
    if (threads <= 0) threads = effective_cpus();
    const bool side_by_side = n_items > 1 && threads > 1;
    const bool nested_serial = n_items > 2 && threads > 1;          // more than one item: the parallelism is across items, inner pools stay idle
    const std::function<void(int)> one = [&](int i) {
        const bool was = tl_item_worker;
        tl_item_worker = nested_serial;
        rc[(size_t)i] = fn(i, msg[(size_t)i]);
        tl_item_worker = was;
    };
    if (!side_by_side) { for (int i = 0; i < n_items; ++i) one(i); }
    else items_pool().run(n_items, std::min(threads, n_items) - 1, one);
    for (int i = 0; i < n_items; ++i)
        if (rc[(size_t)i]) { pcgc_set_error("item %d: %s", i / tasks_per_item, msg[(size_t)i].c_str()); return rc[(size_t)i]; }
    return 0;
}
}  // namespace

typedef int (*pcgc_table_fn)(const float* params, int C, float min_v, float max_v, uint16_t* table_u16, float* cdf_f32);

namespace {
// The decode of n_items is about to need its pools: wake their threads now (they spin for the announced work for up to `us`).
// One item: its two tasks run on the caller + one helper, each with the segment / group pool; more items: one thread per task.
thread_local bool tl_frame_path = false;               // set by pcgc_frame_decode_begin around its probe
void prewake_for_decode(int n_items, int threads, int us) {
    if (n_items <= 0) return;
    if (threads <= 0) threads = effective_cpus();
    if (n_items > 1 && threads > 1) { items_pool().prewake(std::min(threads, 2 * n_items) - 1, us); return; }      // (for_items: nested pools idle)
    if (threads > 1 && !tl_frame_path) items_pool().prewake(1, us);        // (the frame path has its own worker for the second task)
    const int inner = rc_threads();
    if (inner > 1) { segment_pool().prewake(inner - 1, us); octree_pool().prewake(inner - 1, us); }
}

// The table is a pure function of (parameters, range): the last few are kept (the decode of a batch this process has just encoded,
// repeated frames, sequences whose latent range repeats), keyed by the CRC-32 of the parameter bytes + the range.  Same values as a
// fresh evaluation by construction.
struct TableKey { uint32_t pcrc; int C; float lo, hi; pcgc_table_fn fn; bool operator==(const TableKey& o) const { return pcrc == o.pcrc && C == o.C && lo == o.lo && hi == o.hi && fn == o.fn; } };
struct TableEntry { TableKey key; std::shared_ptr<const std::vector<uint16_t>> table; uint32_t crc; };
std::mutex g_table_mu;
std::vector<TableEntry> g_tables;                        // most recent last, at most 32
bool g_table_cache_on = true;                            // pcgc_table_cache(): off = every call evaluates its table, as the reference does
int cached_table(pcgc_table_fn fn, const float* params, int C, float lo, float hi, std::shared_ptr<const std::vector<uint16_t>>& out, uint32_t& crc) {
    const TableKey key{(uint32_t)crc32(0L, (const Bytef*)params, (uInt)(44 * C * 4)), C, lo, hi, fn};
    {
        std::lock_guard<std::mutex> lk(g_table_mu);
        for (auto it = g_tables.rbegin(); it != g_tables.rend(); ++it)
            if (it->key == key) { out = it->table; crc = it->crc; return 0; }
    }
    const int L = (int)(hi - lo) + 1;
    auto t = std::make_shared<std::vector<uint16_t>>((size_t)C * (L + 1));
    if (fn(params, C, lo, hi, t->data(), nullptr) != 0) return -1;
    crc = (uint32_t)crc32(0L, (const Bytef*)t->data(), (uInt)(t->size() * 2));
    out = t;
    std::lock_guard<std::mutex> lk(g_table_mu);
    if (!g_table_cache_on) return 0;
    if (g_tables.size() >= 32) g_tables.erase(g_tables.begin());
    g_tables.push_back(TableEntry{key, out, crc});
    return 0;
}
}  // namespace

// The CDF-table cache of pcgc_items_encode / _decode / pcgc_frame_decode.  mode 0: drop every cached table (the next call of each
// (parameters, range) evaluates it again: what a codec process sees on its first frame, and what the reference does on EVERY call,
// entropy_model.py:165-171,185-190); 1: keep tables (default); -1: keep nothing from now on.  -> number of tables dropped.
extern "C" int pcgc_table_cache(int mode) {
    std::lock_guard<std::mutex> lk(g_table_mu);
    const int n = (int)g_tables.size();
    if (mode <= 0) g_tables.clear();
    if (mode != 0) g_table_cache_on = mode > 0;
    return mode <= 0 ? n : 0;
}

extern "C" int pcgc_items_encode(int n_items, const char* const* stems, const int16_t* sym, const int32_t* xyz, const int64_t* rows,
                                 const float* ranges, int C, const int32_t* counts, const float* eb_params, pcgc_table_fn table_fn,
                                 int index_segments, int write_coords, int threads) {
    if (n_items < 0 || (n_items > 0 && (!stems || !sym || !rows || !ranges || !counts || !eb_params || !table_fn)) || C < 1 || (write_coords && !xyz)) {
        pcgc_set_error("items_encode: bad arguments"); return -2;
    }
    std::vector<int64_t> off((size_t)n_items + 1, 0);
    for (int i = 0; i < n_items; ++i) { if (rows[i] <= 0) { pcgc_set_error("items_encode: item %d is empty", i); return -2; } off[(size_t)i + 1] = off[(size_t)i] + rows[i]; }
    return for_items(n_items, threads, [&](int i, std::string& err) -> int {
        const std::string stem = stems[i];
        const int64_t n = rows[i];
        const float min_v = ranges[2 * i], max_v = ranges[2 * i + 1];
        // (a header's range must be two integral values, as compress() writes them: the table callback sizes its output from the same
        //  two numbers in ITS arithmetic — a fractional bound from a damaged `_H.bin` made it write one row entry past this allocation)
        if (!(min_v <= max_v) || min_v != std::floor(min_v) || max_v != std::floor(max_v) || max_v - min_v > 65000.0f) { err = "symbol range"; return -2; }
        const int L = (int)(max_v - min_v) + 1, Lp = L + 1;
        std::shared_ptr<const std::vector<uint16_t>> tptr;
        uint32_t table_crc = 0;
        StageClock clk;
        if (cached_table(table_fn, eb_params, C, min_v, max_v, tptr, table_crc) != 0) { err = "CDF table evaluation failed"; return -1; }
        const std::vector<uint16_t>& table = *tptr;
        clk.mark("table");
        int segs = (int)std::min<int64_t>(index_segments, n / 1024);
        if (segs < 2) segs = 0;
        std::vector<uint32_t> ckpt((size_t)segs * PCGC_RC_CKPT_WORDS);
        const int16_t* s0 = sym + off[(size_t)i] * C;
        // the stream buffer is kept per thread: a fresh 300 KB vector per frame is an mmap, its page faults and a zero fill (~50 us)
        static thread_local std::vector<uint8_t> stream;
        if (stream.size() < (size_t)(n * C) * 2 + 64) stream.resize((size_t)(n * C) * 2 + 64);
        int64_t nb = segs ? pcgc_rc_encode_indexed(table.data(), C, Lp, s0, n * C, stream.data(), (int64_t)stream.size(), segs, ckpt.data())
                          : pcgc_rc_encode(table.data(), C, Lp, s0, n * C, stream.data(), (int64_t)stream.size());
        if (nb < 0 && nb != INT64_MIN) {
            stream.resize((size_t)(-nb));
            nb = segs ? pcgc_rc_encode_indexed(table.data(), C, Lp, s0, n * C, stream.data(), (int64_t)stream.size(), segs, ckpt.data())
                      : pcgc_rc_encode(table.data(), C, Lp, s0, n * C, stream.data(), (int64_t)stream.size());
        }
        if (nb < 0) { err = "symbol outside the CDF table"; return -2; }
        clk.mark("range");
        if (index_segments > 0) {
            std::vector<uint8_t> side;
            side.insert(side.end(), {'P', 'C', 'G', '2'});
            put32(side, (uint32_t)nb); put32(side, pcgc_crc32(0, stream.data(), nb)); put32(side, (uint32_t)segs); put32(side, table_crc);
            uLong c = crc32(0L, side.data(), (uInt)side.size());
            if (!ckpt.empty()) c = crc32(c, (const Bytef*)ckpt.data(), (uInt)(ckpt.size() * 4));     // (crc32(c, NULL, 0) would RESET the value)
            put32(side, (uint32_t)c);
            side.insert(side.end(), (const uint8_t*)ckpt.data(), (const uint8_t*)ckpt.data() + ckpt.size() * 4);
            if (!write_file(stem + "_F.idx", side.data(), side.size())) { err = "cannot write " + stem + "_F.idx"; return -1; }
        } else {
            std::remove((stem + "_F.idx").c_str());
        }
        if (!write_file(stem + "_F.bin", stream.data(), (size_t)nb)) { err = "cannot write " + stem + "_F.bin"; return -1; }
        uint8_t head[17];
        const int32_t n32 = (int32_t)n, c32 = C;
        std::memcpy(head, &n32, 4); std::memcpy(head + 4, &c32, 4); head[8] = 1; std::memcpy(head + 9, &min_v, 4); std::memcpy(head + 13, &max_v, 4);
        if (!write_file(stem + "_H.bin", head, 17)) { err = "cannot write " + stem + "_H.bin"; return -1; }
        if (!write_file(stem + "_num_points.bin", counts + 3 * i, 12)) { err = "cannot write " + stem + "_num_points.bin"; return -1; }
        clk.mark("files");
        if (write_coords) {
            const int32_t* p = xyz + off[(size_t)i] * 3;
            std::vector<uint8_t> cbin((size_t)n * 4 + 64);
            int64_t cb = pcgc_oct_encode(p, n, cbin.data(), (int64_t)cbin.size());
            if (cb < 0 && cb != INT64_MIN) { cbin.resize((size_t)(-cb)); cb = pcgc_oct_encode(p, n, cbin.data(), (int64_t)cbin.size()); }
            if (cb < 0) { err = "coordinates out of the octree codec's range"; return -2; }
            if (!write_file(stem + "_C.bin", cbin.data(), (size_t)cb)) { err = "cannot write " + stem + "_C.bin"; return -1; }
            clk.mark("coords");
        }
        clk.done("encode", i);
        return 0;
    });
}

// sizes of every item's streams: rows[i] (latent rows = stride-8 voxels), channels, ranges, budgets; native_coords[i] = 1 if <stem>_C.bin
// is a native octree stream of exactly rows[i] points (else the caller decodes it: tmc3)
extern "C" int pcgc_items_probe(int n_items, const char* const* stems, int64_t* rows, int32_t* channels, float* ranges, int32_t* counts,
                                int32_t* native_coords) {
    if (n_items < 0 || (n_items > 0 && (!stems || !rows || !channels || !ranges || !counts || !native_coords))) { pcgc_set_error("items_probe: bad arguments"); return -2; }
    prewake_for_decode(n_items, 0, 500);               // pcgc_items_decode follows within ~0.1 ms: its threads wake up meanwhile
    for (int i = 0; i < n_items; ++i) {
        const std::string stem = stems[i];
        std::vector<uint8_t> h, c;
        if (!read_file(stem + "_H.bin", h) || h.size() < 17 || h[8] != 1) { pcgc_set_error("items_probe: %s_H.bin missing or malformed", stem.c_str()); return -1; }
        int32_t n32, c32;
        std::memcpy(&n32, h.data(), 4); std::memcpy(&c32, h.data() + 4, 4);
        if (n32 < 0 || c32 < 1 || (i > 0 && c32 != channels[0])) { pcgc_set_error("items_probe: %s_H.bin: bad shape", stem.c_str()); return -1; }
        if (n32 > (1 << 27) || c32 > 4096) {               // (a damaged header must not make the caller allocate gigabytes of pinned memory)
            pcgc_set_error("items_probe: %s_H.bin: implausible shape %d x %d", stem.c_str(), (int)n32, (int)c32); return -1;
        }
        rows[i] = n32; channels[0] = c32;
        std::memcpy(ranges + 2 * i, h.data() + 9, 4); std::memcpy(ranges + 2 * i + 1, h.data() + 13, 4);
        if (!read_file(stem + "_num_points.bin", c) || c.size() < 12) { pcgc_set_error("items_probe: %s_num_points.bin missing", stem.c_str()); return -1; }
        std::memcpy(counts + 3 * i, c.data(), 12);
        std::vector<uint8_t> cb;
        native_coords[i] = (read_file(stem + "_C.bin", cb) && pcgc_oct_decode_count(cb.data(), (int64_t)cb.size()) == rows[i]) ? 1 : 0;
    }
    return 0;
}

// The two independent tasks of one item's decode: its coordinate stream (-> xyz: [n, 3] voxels in stream order for coord_layout 0, the sorted
// coordinate level [n, 4] for coord_layout 1) and its feature stream (-> out [n, C] symbols).  -> 0 or an error code with `err` set.
namespace {
int decode_item_coords(const std::string& stem, int i, int64_t n, bool native, int32_t* xyz, int coord_layout, int coord_scale, std::string& err) {
    StageClock clk;
    if (native) {
        std::vector<uint8_t> cb;
        if (!read_file(stem + "_C.bin", cb)) { err = "cannot read " + stem + "_C.bin"; return -1; }
        clk.mark("read");
        if (coord_layout == 0) {
            if (pcgc_oct_decode(cb.data(), (int64_t)cb.size(), xyz, n) != 0) { err = "corrupt " + stem + "_C.bin"; return -2; }
            clk.mark("octree");
        } else {
            // the coordinate LEVEL the decoder starts from (coder.py:97-102): rows (item, scale x, scale y, scale z) in (z, y, x) order —
            // sorted here, on the thread that has the voxels in cache, instead of by a dozen launches after an upload
            static thread_local std::vector<uint64_t> leaves, keys;
            if (oct_decode_leaves(cb.data(), (int64_t)cb.size(), n, leaves) != 0) { err = "corrupt " + stem + "_C.bin"; return -2; }
            clk.mark("octree");
            const int d = cb[5];                                          // bits per coordinate (the tree's depth; checked by the decoder)
            keys.resize((size_t)n);
            zyx_keys(leaves.data(), n, d, keys.data());
            const uint64_t m = (1ull << d) - 1;
            int32_t* L = xyz;
            auto row = [&](int64_t r, uint64_t k) {
                L[4 * r] = i; L[4 * r + 1] = (int32_t)(k & m) * coord_scale;
                L[4 * r + 2] = (int32_t)((k >> d) & m) * coord_scale; L[4 * r + 3] = (int32_t)(k >> (2 * d)) * coord_scale;
            };
            if (3 * d <= 22 && n >= 512) {
                // two 11-bit radix passes with both histograms from one sweep, the second pass scattering the finished rows
                static thread_local std::vector<uint64_t> tmp;
                if (tmp.size() < (size_t)n) tmp.resize((size_t)n);
                uint32_t c0[2048] = {0}, c1[2048] = {0};
                for (int64_t r = 0; r < n; ++r) { const uint64_t k = keys[(size_t)r]; ++c0[k & 2047]; ++c1[(k >> 11) & 2047]; }
                uint32_t a0 = 0, a1 = 0;
                for (int b = 0; b < 2048; ++b) { const uint32_t x0 = c0[b], x1 = c1[b]; c0[b] = a0; c1[b] = a1; a0 += x0; a1 += x1; }
                for (int64_t r = 0; r < n; ++r) { const uint64_t k = keys[(size_t)r]; tmp[c0[k & 2047]++] = k; }
                for (int64_t r = 0; r < n; ++r) { const uint64_t k = tmp[(size_t)r]; row((int64_t)c1[(k >> 11) & 2047]++, k); }
            } else {
                sort_codes(keys, 3 * d);
                for (int64_t r = 0; r < n; ++r) row(r, keys[(size_t)r]);
            }
            clk.mark("level");
        }
    }
    clk.done("decode coords", i);
    return 0;
}
int decode_item_features(const std::string& stem, int i, int64_t n, int C, float min_v, float max_v, const float* eb_params, pcgc_table_fn table_fn,
                         int use_sidecar, int16_t* out, std::string& err) {
    StageClock clk;
    if (n == 0) return 0;
    // (a header's range must be two integral values, as compress() writes them: the table callback sizes its output from the same
    //  two numbers in ITS arithmetic — a fractional bound from a damaged `_H.bin` made it write one row entry past this allocation)
    if (!(min_v <= max_v) || min_v != std::floor(min_v) || max_v != std::floor(max_v) || max_v - min_v > 65000.0f) { err = "symbol range"; return -2; }
    const int L = (int)(max_v - min_v) + 1, Lp = L + 1;
    std::shared_ptr<const std::vector<uint16_t>> tptr;
    uint32_t mine = 0;
    if (cached_table(table_fn, eb_params, C, min_v, max_v, tptr, mine) != 0) { err = "CDF table evaluation failed"; return -1; }
    const std::vector<uint16_t>& table = *tptr;
    clk.mark("table");
    std::vector<uint8_t> stream, side;
    if (!read_file(stem + "_F.bin", stream)) { err = "cannot read " + stem + "_F.bin"; return -1; }
    clk.mark("read");
    int n_ck = 0; const uint32_t* ck = nullptr;
    if (use_sidecar && read_file(stem + "_F.idx", side) && side.size() >= kSidecarHead && std::memcmp(side.data(), "PCG2", 4) == 0) {
        const uint32_t nbytes = get32(side.data() + 4), scrc = get32(side.data() + 8), count = get32(side.data() + 12), tcrc = get32(side.data() + 16), self = get32(side.data() + 20);
        uLong c = crc32(0L, side.data(), 20);
        if (side.size() > kSidecarHead) c = crc32(c, side.data() + kSidecarHead, (uInt)(side.size() - kSidecarHead));
        if (nbytes == stream.size() && side.size() == kSidecarHead + (size_t)count * 4 * PCGC_RC_CKPT_WORDS && self == (uint32_t)c &&
            scrc == pcgc_crc32(0, stream.data(), (int64_t)stream.size())) {
            if (tcrc != mine) {
                char b[160]; std::snprintf(b, sizeof b, "the CDF table derived on this host (CRC-32 %08x) is not the one the stream was coded with (%08x)", mine, tcrc);
                err = b; return -5;
            }
            if (count >= 2) { n_ck = (int)count; ck = (const uint32_t*)(side.data() + kSidecarHead); }
        }
    }
    clk.mark("sidecar");
        int rc;
    if (n_ck) rc = pcgc_rc_decode_indexed(table.data(), C, Lp, stream.data(), (int64_t)stream.size(), out, n * C, n_ck, ck);
    else rc = pcgc_rc_decode(table.data(), C, Lp, stream.data(), (int64_t)stream.size(), out, n * C);
    if (rc != 0) { err = "range decoder refused " + stem + "_F.bin"; return rc; }
    clk.mark("range");
    clk.done("decode features", i);
    return 0;
}
}  // namespace

// -> sym [sum rows, C] and (for items with native_coords) xyz [sum rows, 3].  use_sidecar = 0: never read <stem>_F.idx.
// Returns -5 if a sidecar says the stream was coded with another CDF table than this host derives (see coder.py).
extern "C" int pcgc_items_decode(int n_items, const char* const* stems, const int64_t* rows, int C, const float* ranges, const int32_t* native_coords,
                                 const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int16_t* sym, int32_t* xyz, int coord_layout,
                                 int coord_scale, int threads) {
    if (n_items < 0 || (n_items > 0 && (!stems || !rows || !ranges || !native_coords || !eb_params || !table_fn || !sym || !xyz)) || C < 1 ||
        coord_layout < 0 || coord_layout > 1 || coord_scale < 1) {
        pcgc_set_error("items_decode: bad arguments"); return -2;
    }
    prewake_for_decode(n_items, threads, 300);
    std::vector<int64_t> off((size_t)n_items + 1, 0);
    for (int i = 0; i < n_items; ++i) off[(size_t)i + 1] = off[(size_t)i] + rows[i];
    // two tasks per item — its coordinate stream and its feature stream are independent — so that ONE cloud decodes both side by side as
    // well, each on its own pool of segment / group threads (the coordinate stream is the longer task by now — 0.18 against 0.16 ms — so
    // the calling thread starts it at once and the feature stream pays the helper's wake-up)
    return for_items(2 * n_items, threads, [&](int task, std::string& err) -> int {
        const int i = task >> 1;
        if ((task & 1) == 0)
            return decode_item_coords(stems[i], i, rows[i], native_coords[i] != 0, xyz + off[(size_t)i] * (coord_layout == 0 ? 3 : 4), coord_layout, coord_scale, err);
        return decode_item_features(stems[i], i, rows[i], C, ranges[2 * i], ranges[2 * i + 1], eb_params, table_fn, use_sidecar, sym + off[(size_t)i] * C, err);
    }, 2);
}

// One cloud in ONE call (the single-frame path of Coder.decode, coder.py:93-104): what pcgc_items_probe + pcgc_items_decode do for it,
// into buffers the caller keeps (pinned memory: both uploads are asynchronous copies).  sym [cap_rows, C] int16, level [cap_rows, 4] int32
// (coord_layout 1 of pcgc_items_decode).  info[6] = rows, channels, N4, N2, N1, native_coords; range[2] = min_v, max_v.
// -> 0; 1 = cap_rows is too small (info[0] rows are needed: grow the buffers and call again; nothing was decoded); < 0 error.
// A `_C.bin` that is not a native octree stream (tmc3): info[5] = 0, the features are decoded and `level` is left untouched.
extern "C" int pcgc_frame_decode(const char* stem, int C, const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int coord_scale,
                                 int64_t cap_rows, int16_t* sym, int32_t* level, int64_t* info, float* range, int threads) {
    if (!stem || !eb_params || !table_fn || !sym || !level || !info || !range || C < 1 || cap_rows < 0) { pcgc_set_error("frame_decode: bad arguments"); return -2; }
    int64_t rows = 0; int32_t channels = 0, counts[3] = {0, 0, 0}, native = 0;
    const int rc = pcgc_items_probe(1, &stem, &rows, &channels, range, counts, &native);
    if (rc != 0) return rc;
    info[0] = rows; info[1] = channels; info[2] = counts[0]; info[3] = counts[1]; info[4] = counts[2]; info[5] = native;
    if (channels != C) { pcgc_set_error("frame_decode: %s_H.bin has %d channels, the model %d", stem, (int)channels, C); return -2; }
    if (rows > cap_rows) return 1;
    return pcgc_items_decode(1, &stem, &rows, C, range, &native, eb_params, table_fn, use_sidecar, sym, level, 1, coord_scale > 0 ? coord_scale : 1, threads);
}

// pcgc_frame_decode in two halves (round 4).  The coordinate stream of a vox10 frame is decoded in ~0.17 ms, the feature stream (CDF table
// + range decoder) in ~0.3 ms, side by side; the first decoder kernels — hash of the stride-8 level, its kernel map, the children level and its
// map — need the coordinates only.  `_begin` returns as soon as the coordinate level is in `level` (the feature stream keeps decoding on the
// library's threads), the caller uploads the level and enqueues those kernels, `_end` waits for the symbols.  Same arguments, same results
// and error codes as pcgc_frame_decode; 1 from `_begin` = buffers too small (nothing pending).  One frame at a time per process in this
// form: a second caller that arrives while a frame is pending is served synchronously (its `_begin` does everything, its `_end` nothing).
namespace {
struct FrameAsync {
    std::mutex owner;                                    // held from a successful asynchronous _begin to its _end
    std::thread worker; std::mutex m; std::condition_variable cv;
    std::function<void()> job; bool has_job = false, stop = false;
    std::atomic<int> finished{0};
    std::atomic<int64_t> spin_until{0};
    std::atomic<int> test_delay_us{0}, stolen{0};
    int rc = 0, coords_rc = 0; std::string err, coords_err;
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void loop() {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                while (!has_job && !stop) {
                    if (now_ns() < spin_until.load(std::memory_order_relaxed)) { lk.unlock(); _mm_pause(); lk.lock(); continue; }
                    cv.wait(lk, [&] { return has_job || stop || now_ns() < spin_until.load(std::memory_order_relaxed); });
                }
                if (stop) return;
                if (const int d = test_delay_us.load(std::memory_order_relaxed)) {      // (tests: a worker that is late to its job)
                    lk.unlock();
                    std::this_thread::sleep_for(std::chrono::microseconds(d));
                    lk.lock();
                    if (!has_job) continue;                                             // the caller took it back (finish())
                }
                j = std::move(job); has_job = false;
            }
            j();
            finished.store(1, std::memory_order_release);
        }
    }
    // the caller's side of the hand-off: spin for the ~0.1-0.3 ms a warm decode takes, then stop burning the core (a cold table, slow files or a
    // small cgroup CPU quota shared with the segment pool): yield, then sleep in 50 us steps
    static void wait(const std::atomic<int>& flag) {
        const int64_t t0 = now_ns();
        for (int spins = 0;; ++spins) {
            if (flag.load(std::memory_order_acquire) != 0) return;
            if (spins < 256) { _mm_pause(); continue; }
            const int64_t dt = now_ns() - t0;
            if (dt < 400 * 1000) _mm_pause();
            else if (dt < 2 * 1000 * 1000) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    // the posted job is complete when this returns.  A worker that has not even TAKEN the job yet (its wake-up is late: a busy host — 2-8 ms
    // were measured on a shared box, profiles/r05_step_outliers.md) loses it to the caller, which runs it on its own thread instead of waiting.
    void finish() {
        std::function<void()> j;
        { std::lock_guard<std::mutex> lk(m); if (has_job) { j = std::move(job); has_job = false; } }
        if (j) { stolen.fetch_add(1, std::memory_order_relaxed); j(); finished.store(1, std::memory_order_release); }
        else wait(finished);
    }
    void ensure() { if (!worker.joinable()) worker = std::thread([this] { loop(); }); }
    void prewake(int us) { ensure(); spin_until.store(now_ns() + (int64_t)us * 1000, std::memory_order_relaxed); cv.notify_one(); }
    void post(std::function<void()> j) { ensure(); { std::lock_guard<std::mutex> lk(m); job = std::move(j); has_job = true; } cv.notify_one(); }
    ~FrameAsync() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); if (worker.joinable()) worker.join(); }
};
FrameAsync& frame_async() { static FrameAsync* f = new FrameAsync; return *f; }      // (never destroyed: its thread may outlive static destructors)
thread_local bool tl_frame_pending = false;              // this thread owns frame_async().owner
}  // namespace

extern "C" int pcgc_frame_decode_begin(const char* stem, int C, const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int coord_scale,
                                       int64_t cap_rows, int16_t* sym, int32_t* level, int64_t* info, float* range, int threads) {
    if (!stem || !eb_params || !table_fn || !sym || !level || !info || !range || C < 1 || cap_rows < 0) { pcgc_set_error("frame_decode: bad arguments"); return -2; }
    FrameAsync& fa = frame_async();
    if (tl_frame_pending) {
        // a frame this thread began and never finished (an exception between _begin and _end on the caller's side): finish it here — wait for the
        // worker, which may still be writing into the previous call's buffers, and release the slot — instead of refusing every later frame
        fa.finish();
        tl_frame_pending = false;
        fa.owner.unlock();
    }
    const bool async = (threads <= 0 ? effective_cpus() : threads) >= 2 && effective_cpus() >= 3 && fa.owner.try_lock();
    if (!async) return pcgc_frame_decode(stem, C, eb_params, table_fn, use_sidecar, coord_scale, cap_rows, sym, level, info, range, threads);
    fa.prewake(400);                                      // (the probe reads four small files meanwhile)
    int64_t rows = 0; int32_t channels = 0, counts[3] = {0, 0, 0}, native = 0;
    tl_frame_path = true;
    int rc = pcgc_items_probe(1, &stem, &rows, &channels, range, counts, &native);
    tl_frame_path = false;
    if (rc == 0) {
        info[0] = rows; info[1] = channels; info[2] = counts[0]; info[3] = counts[1]; info[4] = counts[2]; info[5] = native;
        if (channels != C) { pcgc_set_error("frame_decode: %s_H.bin has %d channels, the model %d", stem, (int)channels, C); rc = -2; }
        else if (rows > cap_rows) rc = 1;
    }
    if (rc != 0) { fa.owner.unlock(); return rc; }
    fa.finished.store(0); fa.rc = fa.coords_rc = 0; fa.err.clear(); fa.coords_err.clear();
    const std::string stem_copy = stem;
    const float r0 = range[0], r1 = range[1];
    // the feature stream goes to the frame worker (which spins for it since the prewake above) with the segment pool under it.  The
    // parameter vector is COPIED into the job: the caller's array need not outlive this call (the table is evaluated, and its cache key
    // hashed, after _begin has returned)
    const std::vector<float> params_copy(eb_params, eb_params + (size_t)44 * C);
    fa.post([=, &fa] {
        std::string e;
        const int r = decode_item_features(stem_copy, 0, rows, C, r0, r1, params_copy.data(), table_fn, use_sidecar, sym, e);
        if (r != 0) { fa.rc = r; fa.err = e; }
    });
    tl_frame_pending = true;
    // the coordinate stream is decoded HERE, on the calling thread (round 5; round 4 handed both tasks to the worker and spun for a flag): the
    // caller needs this result before it can do anything else, so no other thread's wake-up is on its path — only the octree pool's helpers,
    // which the calling thread never waits for before it has run out of groups itself
    fa.coords_rc = decode_item_coords(stem_copy, 0, rows, native != 0, level, 1, coord_scale > 0 ? coord_scale : 1, fa.coords_err);
    if (fa.coords_rc != 0) {
        // `level` is stale or uninitialised now: the caller must not upload it and enqueue map kernels on garbage coordinates, so the error
        // surfaces HERE (the feature task is drained first: it writes into the caller's symbol buffer)
        fa.finish();
        const int rc = fa.coords_rc;
        pcgc_set_error("item 0: %s", fa.coords_err.c_str());
        tl_frame_pending = false;
        fa.owner.unlock();
        return rc;
    }
    return 0;
}
// Test hook: make the frame worker `delay_us` late to every job (0 = off) -> the number of jobs callers have taken back so far.
extern "C" int pcgc_frame_worker_test(int delay_us) {
    FrameAsync& fa = frame_async();
    if (delay_us >= 0) fa.test_delay_us.store(delay_us, std::memory_order_relaxed);
    return fa.stolen.load(std::memory_order_relaxed);
}
extern "C" int pcgc_frame_decode_end(void) {
    if (!tl_frame_pending) return 0;                      // (the synchronous form: everything happened in _begin)
    FrameAsync& fa = frame_async();
    fa.finish();
    const int rc = fa.coords_rc ? fa.coords_rc : fa.rc;   // (the order of pcgc_items_decode: the first failing task's code and message)
    if (rc != 0) pcgc_set_error("item 0: %s", (fa.coords_rc ? fa.coords_err : fa.err).c_str());
    tl_frame_pending = false;
    fa.owner.unlock();
    return rc;
}
