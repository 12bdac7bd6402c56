#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
#include "../../include/pcgc_hip.h"
extern "C" int64_t pcgc_rc_encode_indexed(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, int n_ckpt, uint32_t* ckpt);
extern "C" int pcgc_rc_decode_indexed(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n, int n_ckpt, const uint32_t* ckpt);
extern "C" int pcgc_rc_decode(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n);
extern "C" int pcgc_set_rc_threads(int);
extern "C" int64_t pcgc_oct_encode(const int32_t* xyz, int64_t n, uint8_t* out, int64_t cap);
extern "C" int pcgc_oct_decode(const uint8_t* in, int64_t nbytes, int32_t* xyz, int64_t n);
void pcgc_set_error(const char* fmt, ...) { }
extern "C" const char* pcgc_last_error(void) { return ""; }                      // (coords.hip in the library)
int main() {
    std::mt19937 rng(5);
    int bad = 0;
    for (int it = 0; it < 300; ++it) {
        const int C = 1 + rng() % 9, L = 1 + rng() % (it % 7 == 0 ? 200 : 30), Lp = L + 1;
        std::vector<uint16_t> cdf((size_t)C * Lp);
        std::vector<std::vector<double>> pm(C, std::vector<double>(L));
        for (int c = 0; c < C; ++c) {
            double sum = 0; const int mode = rng() % 4;
            for (int j = 0; j < L; ++j) { double p = mode == 0 ? 1.0 : (mode == 1 ? std::exp(-0.5 * (j - L / 2.0) * (j - L / 2.0)) + 1e-9 : (mode == 2 ? (j == L / 2 ? 1e-5 : 1.0) : (j == 0 ? 1000.0 : 1e-3))); pm[c][j] = p; sum += p; }
            double acc = 0;
            for (int j = 0; j < L; ++j) { cdf[c * Lp + j] = (uint16_t)(std::lround(acc / sum * (65536 - L)) + j); acc += pm[c][j]; }
            cdf[c * Lp + L] = 0;
        }
        const int64_t rows = 1 + rng() % 3000, n = rows * C;
        std::vector<int16_t> sym((size_t)n);
        for (int64_t i = 0; i < n; ++i) { const int c = i % C; const int mode = rng() % 3; sym[i] = mode == 0 ? (int16_t)(rng() % L) : (mode == 1 ? (int16_t)(L / 2) : (int16_t)std::discrete_distribution<int>(pm[c].begin(), pm[c].end())(rng)); }
        std::vector<uint8_t> out((size_t)n * 3 + 64);
        const int nck = rng() % 20; std::vector<uint32_t> ck((size_t)std::max(nck, 1) * 6);
        int64_t nb = pcgc_rc_encode_indexed(cdf.data(), C, Lp, sym.data(), n, out.data(), (int64_t)out.size(), nck, ck.data());
        if (nb < 0) { out.resize((size_t)(-nb)); nb = pcgc_rc_encode_indexed(cdf.data(), C, Lp, sym.data(), n, out.data(), (int64_t)out.size(), nck, ck.data()); }
        if (nb < 0) { printf("encode failed %lld\n", (long long)nb); ++bad; continue; }
        std::vector<int16_t> back((size_t)n);
        pcgc_set_rc_threads(1 + rng() % 4);
        int rc = nck ? pcgc_rc_decode_indexed(cdf.data(), C, Lp, out.data(), nb, back.data(), n, nck, ck.data()) : pcgc_rc_decode(cdf.data(), C, Lp, out.data(), nb, back.data(), n);
        if (rc != 0 || memcmp(back.data(), sym.data(), (size_t)n * 2) != 0) { printf("MISMATCH it %d C %d L %d rows %lld nck %d rc %d\n", it, C, L, (long long)rows, nck, rc); ++bad; }
        // corrupt stream: must not crash
        std::vector<uint8_t> junk(out.begin(), out.begin() + nb); for (int k = 0; k < 8 && !junk.empty(); ++k) junk[rng() % junk.size()] = (uint8_t)rng();
        pcgc_rc_decode(cdf.data(), C, Lp, junk.data(), (int64_t)junk.size(), back.data(), n);
    }
    // octree
    for (int it = 0; it < 60; ++it) {
        const int ext = 2 + rng() % (it % 5 == 0 ? 2000 : 200); const int64_t n = 1 + rng() % 20000;
        std::vector<int32_t> xyz((size_t)n * 3); for (auto& v : xyz) v = rng() % ext;
        std::vector<uint8_t> out((size_t)n * 8 + 64);
        int64_t nb = pcgc_oct_encode(xyz.data(), n, out.data(), (int64_t)out.size());
        if (nb < 0) { printf("oct encode failed\n"); ++bad; continue; }
        int64_t cnt; memcpy(&cnt, out.data() + 6, 4); cnt &= 0xFFFFFFFF;
        std::vector<int32_t> back((size_t)cnt * 3);
        if (pcgc_oct_decode(out.data(), nb, back.data(), cnt) != 0) { printf("oct decode failed\n"); ++bad; }
        std::vector<uint8_t> junk(out.begin(), out.begin() + nb); for (int k = 0; k < 6; ++k) junk[10 + rng() % (junk.size() - 10)] ^= (uint8_t)(1 + rng() % 255);
        pcgc_oct_decode(junk.data(), (int64_t)junk.size(), back.data(), cnt);
    }
    printf("done, failures %d\n", bad);
}
