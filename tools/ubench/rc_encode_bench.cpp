#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <chrono>
#include <cmath>
#include <random>
#include <cstring>
extern "C" int64_t pcgc_rc_encode_indexed(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, int n_ckpt, uint32_t* ckpt);
extern "C" int64_t pcgc_rc_encode(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap);
void pcgc_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); }
int main(int argc, char** argv) {
    const int C = 8, L = 21, Lp = L + 1; const int64_t rows = 18732, n = rows * C;
    std::mt19937 rng(7);
    std::vector<uint16_t> cdf((size_t)C * Lp);
    std::vector<std::vector<double>> pm(C, std::vector<double>(L));
    for (int c = 0; c < C; ++c) {
        double sig = 1.0 + 0.5 * c, sum = 0;
        for (int j = 0; j < L; ++j) { pm[c][j] = std::exp(-0.5 * (j - 10) * (j - 10) / (sig * sig)) + 1e-6; sum += pm[c][j]; }
        double acc = 0; 
        for (int j = 0; j < L; ++j) { cdf[c * Lp + j] = (uint16_t)(std::lround(acc / sum * (65536 - L)) + j); acc += pm[c][j]; }
        cdf[c * Lp + L] = 0;   // 2^16 wraps to 0 in uint16 (torchac)
    }
    std::vector<int16_t> sym((size_t)n);
    for (int64_t i = 0; i < n; ++i) { int c = i % C; std::discrete_distribution<int> d(pm[c].begin(), pm[c].end()); sym[i] = (int16_t)d(rng); }
    std::vector<uint8_t> out((size_t)n * 4 + 64);
    std::vector<uint32_t> ck(8 * 6);
    int64_t nb = pcgc_rc_encode_indexed(cdf.data(), C, Lp, sym.data(), n, out.data(), out.size(), 8, ck.data());
    uint64_t h = 1469598103934665603ull; for (int64_t i = 0; i < nb; ++i) h = (h ^ out[i]) * 1099511628211ull;
    for (uint32_t v : ck) h = (h ^ v) * 1099511628211ull;
    double best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; ++i) pcgc_rc_encode_indexed(cdf.data(), C, Lp, sym.data(), n, out.data(), out.size(), 8, ck.data());
        auto t1 = std::chrono::steady_clock::now();
        best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count() / 20);
    }
    printf("bytes %lld (%.2f bit/sym) hash %016llx  encode %.3f ms = %.2f ns/sym\n", (long long)nb, 8.0 * nb / n, (unsigned long long)h, best, best * 1e6 / n);
}
