#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <random>
#include <fstream>
#include "../../include/pcgc_hip.h"
void pcgc_set_error(const char* fmt, ...) { }
extern "C" const char* pcgc_last_error(void) { return ""; }                      // (coords.hip in the library)
static int table_fn(const float* params, int C, float min_v, float max_v, uint16_t* t, float* cdf) {
    if (min_v != std::floor(min_v) || max_v != std::floor(max_v)) return -2;
    const int L = (int)(max_v - min_v) + 1, Lp = L + 1;
    for (int c = 0; c < C; ++c) {
        std::vector<double> pm(L); double sum = 0;
        for (int j = 0; j < L; ++j) { pm[j] = std::exp(-0.5 * (j - L / 2.0) * (j - L / 2.0) / (4.0 + c)) + 1e-6; sum += pm[j]; }
        double acc = 0;
        for (int j = 0; j < L; ++j) { t[c * Lp + j] = (uint16_t)(std::lround(acc / sum * (65536 - L)) + j); acc += pm[j]; }
        t[c * Lp + L] = 0;
    }
    return 0;
}
static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
static void dump(const std::string& p, const std::vector<uint8_t>& b) { std::ofstream f(p, std::ios::binary); f.write((const char*)b.data(), b.size()); }
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 500;
    const int C = 8; const int64_t r = 9000;
    std::mt19937 rng(11);
    std::vector<int16_t> sym(r * C); for (auto& v : sym) v = (int16_t)std::min(16, std::max(0, (int)std::lround(8 + 2.5 * std::normal_distribution<double>()(rng))));
    sym[0] = 0; sym.back() = 16;
    std::vector<int32_t> xyz; { std::vector<uint8_t> used(100 * 100 * 100, 0); while ((int64_t)xyz.size() < 3 * r) { int x = rng() % 100, y = rng() % 100, z = rng() % 100; auto& u = used[(x * 100 + y) * 100 + z]; if (!u) { u = 1; xyz.push_back(x); xyz.push_back(y); xyz.push_back(z); } } }
    std::vector<float> params(44 * C, 0.1f);
    std::string stem = "/tmp/pcgc_fuzz_g"; const char* st = stem.c_str();
    float ranges[2] = {-8.f, 8.f}; int32_t counts[3] = {1, 2, 3}; int64_t rows = r;
    int rc = pcgc_items_encode(1, &st, sym.data(), xyz.data(), &rows, ranges, C, counts, params.data(), table_fn, 16, 1, 1);
    printf("encode rc %d\n", rc);
    const char* sfx[] = {"_C.bin", "_F.bin", "_H.bin", "_num_points.bin", "_F.idx"};
    std::vector<std::vector<uint8_t>> good; for (auto s : sfx) good.push_back(slurp(stem + s));
    std::vector<int16_t> so((r + 8) * C); std::vector<int32_t> lo((r + 8) * 4);
    int64_t info[6]; float rg[2]; int hist[8] = {0};
    for (int it = 0; it < iters; ++it) {
        const int f = rng() % 5; auto bad = good[f];
        const int kind = rng() % 4;
        if (kind == 0 && !bad.empty()) bad.resize(rng() % bad.size());
        else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 4); ++k) if (!bad.empty()) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
        else if (kind == 2) for (int k = 0; k < 8; ++k) if (!bad.empty()) bad[rng() % bad.size()] = (uint8_t)rng();
        else if (!bad.empty()) { size_t a = rng() % bad.size(); for (size_t k = a; k < bad.size() && k < a + 64; ++k) bad[k] = (uint8_t)rng(); }
        dump(stem + sfx[f], bad);
        const int q = pcgc_frame_decode(st, C, params.data(), table_fn, 1, 8, r + 4, so.data(), lo.data(), info, rg, (int)(rng() % 3));
        ++hist[q == 0 ? 0 : (q == 1 ? 1 : (q == -1 ? 2 : (q == -2 ? 3 : (q == -5 ? 4 : 5))))];
        dump(stem + sfx[f], good[f]);
    }
    printf("rc histogram: ok %d, grow %d, -1 %d, -2 %d, -5 %d, other %d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
}
