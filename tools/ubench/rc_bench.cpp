// standalone timing harness for the range coder variants
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include <cmath>
extern "C" int64_t pcgc_rc_encode(const uint16_t*, int, int, const int16_t*, int64_t, uint8_t*, int64_t);
extern "C" int pcgc_rc_decode(const uint16_t*, int, int, const uint8_t*, int64_t, int16_t*, int64_t);
int main() {
    const int C = 8, L = 21, Lp = L + 1; const int64_t n = 18732 * 8;
    std::vector<uint16_t> cdf(C * Lp);
    for (int c = 0; c < C; ++c) {   // gaussian-ish pmf, sigma 2.5
        double p[64], tot = 0; for (int s = 0; s < L; ++s) { p[s] = std::exp(-0.5 * std::pow((s - 10.0 - 0.1 * c) / 2.5, 2)) + 1e-6; tot += p[s]; }
        double run = 0; cdf[c * Lp] = 0;
        for (int s = 0; s < L; ++s) { run += p[s] / tot; double v = std::min(run, 1.0); cdf[c * Lp + s + 1] = (uint16_t)((int)std::nearbyint(v * (65536 - L)) + s + 1); }
    }
    std::mt19937 g(1); std::normal_distribution<double> nd(10.0, 2.5);
    std::vector<int16_t> sym(n), back(n);
    for (auto& s : sym) { int v = (int)std::lround(nd(g)); s = (int16_t)std::min(std::max(v, 0), L - 1); }
    std::vector<uint8_t> out(n * 2 + 4096);
    int64_t nb = 0; double te = 1e9, td = 1e9;
    for (int it = 0; it < 40; ++it) {
        auto a = std::chrono::steady_clock::now();
        nb = pcgc_rc_encode(cdf.data(), C, Lp, sym.data(), n, out.data(), (int64_t)out.size());
        auto b = std::chrono::steady_clock::now();
        pcgc_rc_decode(cdf.data(), C, Lp, out.data(), nb, back.data(), n);
        auto c = std::chrono::steady_clock::now();
        te = std::min(te, std::chrono::duration<double, std::milli>(b - a).count());
        td = std::min(td, std::chrono::duration<double, std::milli>(c - b).count());
    }
    uint64_t h = 1469598103934665603ull; for (int64_t i = 0; i < nb; ++i) h = (h ^ out[i]) * 1099511628211ull;
    printf("bytes %ld hash %016lx roundtrip %s  enc %.3f ms (%.1f ns/sym)  dec %.3f ms (%.1f ns/sym)\n", (long)nb, (unsigned long)h,
           memcmp(sym.data(), back.data(), n * 2) == 0 ? "OK" : "FAIL", te, te * 1e6 / n, td, td * 1e6 / n);
}
