#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <random>
#include <thread>
#include "../../include/pcgc_hip.h"
void pcgc_set_error(const char* fmt, ...) { }
extern "C" const char* pcgc_last_error(void) { return ""; }                      // (coords.hip in the library)
static int table_fn(const float* params, int C, float min_v, float max_v, uint16_t* t, float* cdf) {
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
int main() {
    const int C = 8, T = 4;
    std::vector<float> params(44 * C, 0.1f);
    std::vector<std::string> stems; std::vector<int64_t> rs; std::vector<std::vector<int16_t>> syms;
    for (int t = 0; t < T; ++t) {
        std::mt19937 rng(3 + t); const int64_t r = 9000 + 2000 * t;
        std::vector<int16_t> sym(r * C); for (auto& v : sym) v = (int16_t)std::min(16, std::max(0, (int)std::lround(8 + 2.5 * std::normal_distribution<double>()(rng))));
        sym[0] = 0; sym.back() = 16;
        std::vector<int32_t> xyz; { std::vector<uint8_t> used(100 * 100 * 100, 0); while ((int64_t)xyz.size() < 3 * r) { int x = rng() % 100, y = rng() % 100, z = rng() % 100; auto& u = used[(x * 100 + y) * 100 + z]; if (!u) { u = 1; xyz.push_back(x); xyz.push_back(y); xyz.push_back(z); } } }
        std::string stem = "/tmp/pcgc_tsan_t" + std::to_string(t); const char* st = stem.c_str();
        float ranges[2] = {-8.f, 8.f}; int32_t counts[3] = {1, 2, 3}; int64_t rows = r;
        if (pcgc_items_encode(1, &st, sym.data(), xyz.data(), &rows, ranges, C, counts, params.data(), table_fn, 16, 1, 0)) { printf("encode failed\n"); return 1; }
        stems.push_back(stem); rs.push_back(r); syms.push_back(sym);
    }
    int bad = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        std::vector<int16_t> so(rs[t] * C); std::vector<int32_t> lo(rs[t] * 4); int64_t info[6]; float rg[2];
        for (int it = 0; it < 8; ++it) {
            if (pcgc_frame_decode(stems[t].c_str(), C, params.data(), table_fn, 1, 8, rs[t], so.data(), lo.data(), info, rg, 2) != 0 || memcmp(so.data(), syms[t].data(), so.size() * 2) != 0) __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED);
        }
    });
    for (auto& x : th) x.join();
    // the two-halves form from all four threads at once (one is served by the frame worker, the others synchronously), with the worker on time
    // and 0.3 s late to every job (the caller takes the job back in _end)
    for (int delay : {0, 300000}) {
        pcgc_frame_worker_test(delay);
        th.clear();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            std::vector<int16_t> so(rs[t] * C); std::vector<int32_t> lo(rs[t] * 4); int64_t info[6]; float rg[2];
            for (int it = 0; it < 6; ++it) {
                std::fill(so.begin(), so.end(), (int16_t)-1);
                int rc = pcgc_frame_decode_begin(stems[t].c_str(), C, params.data(), table_fn, 1, 8, rs[t], so.data(), lo.data(), info, rg, 0);
                if (rc == 0) rc = pcgc_frame_decode_end();
                if (rc != 0 || memcmp(so.data(), syms[t].data(), so.size() * 2) != 0) __atomic_add_fetch(&bad, 1, __ATOMIC_RELAXED);
            }
        });
        for (auto& x : th) x.join();
    }
    printf("threads done, failures %d, jobs taken back by callers %d\n", bad, pcgc_frame_worker_test(0));
    return bad != 0;
}
