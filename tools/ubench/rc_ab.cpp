// A/B timing + equality of tools/ubench/rc_variants.cpp against the library's range coder
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include <cmath>
#include <algorithm>
extern "C" int64_t pcgc_rc_encode(const uint16_t*, int, int, const int16_t*, int64_t, uint8_t*, int64_t);
extern "C" int pcgc_rc_decode(const uint16_t*, int, int, const uint8_t*, int64_t, int16_t*, int64_t);
extern "C" int64_t rc4_encode(const uint16_t*, int, int, const int16_t*, int64_t, uint8_t*, int64_t);
extern "C" int rc4_decode(const uint16_t*, int, int, const uint8_t*, int64_t, int16_t*, int64_t);
template <class F> double best(F f, int it = 30) { double t = 1e9; for (int i = 0; i < it; ++i) { auto a = std::chrono::steady_clock::now(); f(); auto b = std::chrono::steady_clock::now(); t = std::min(t, std::chrono::duration<double, std::milli>(b - a).count()); } return t; }
int run(int L, double sigma, int64_t n, unsigned seedv) {
    const int C = 8, Lp = L + 1;
    std::vector<uint16_t> cdf(C * Lp);
    double mid = (L - 1) / 2.0;
    for (int c = 0; c < C; ++c) {
        std::vector<double> p(L); double tot = 0; for (int s = 0; s < L; ++s) { p[s] = std::exp(-0.5 * std::pow((s - mid - 0.1 * c) / sigma, 2)) + 1e-9; tot += p[s]; }
        double runp = 0; cdf[c * Lp] = 0;
        for (int s = 0; s < L; ++s) { runp += p[s] / tot; double v = std::min(runp, 1.0); cdf[c * Lp + s + 1] = (uint16_t)((int)std::nearbyint(v * (65536 - L)) + s + 1); }
    }
    std::mt19937 g(seedv); std::normal_distribution<double> nd(mid, sigma);
    std::vector<int16_t> sym(n), b1(n), b2(n);
    for (auto& s : sym) { int v = (int)std::lround(nd(g)); s = (int16_t)std::min(std::max(v, 0), L - 1); }
    if (seedv % 3 == 0) for (auto& s : sym) if (g() % 50 == 0) s = (int16_t)(g() % L);
    std::vector<uint8_t> o1(n * 4 + 4096), o2(n * 4 + 4096);
    int64_t n1 = 0, n2 = 0;
    double t1 = best([&] { n1 = pcgc_rc_encode(cdf.data(), C, Lp, sym.data(), n, o1.data(), (int64_t)o1.size()); });
    double t2 = best([&] { n2 = rc4_encode(cdf.data(), C, Lp, sym.data(), n, o2.data(), (int64_t)o2.size()); });
    bool same = n1 == n2 && memcmp(o1.data(), o2.data(), n1) == 0;
    double d1 = best([&] { pcgc_rc_decode(cdf.data(), C, Lp, o1.data(), n1, b1.data(), n); });
    double d2 = Lp <= 64 ? best([&] { rc4_decode(cdf.data(), C, Lp, o1.data(), n1, b2.data(), n); }) : 0;
    bool ok1 = memcmp(sym.data(), b1.data(), n * 2) == 0, ok2 = Lp > 64 || memcmp(sym.data(), b2.data(), n * 2) == 0;
    printf("L %4d sigma %6.2f n %7ld %.2f b/sym | enc %s lib %.2f new %.2f ns/sym | dec %s%s lib %.2f new %.2f ns/sym\n", L, sigma, (long)n, n1 * 8.0 / n,
           same ? "SAME" : "DIFF", t1 * 1e6 / n, t2 * 1e6 / n, ok1 ? "ok" : "FAIL", ok2 ? "ok" : "FAIL", d1 * 1e6 / n, d2 * 1e6 / n);
    return (same && ok1 && ok2) ? 0 : 1;
}
int main() {
    int bad = 0;
    bad += run(21, 2.5, 149856, 1); bad += run(21, 8.0, 149856, 2); bad += run(21, 0.3, 149856, 3); bad += run(5, 0.2, 100000, 4);
    bad += run(2, 0.5, 100000, 5); bad += run(63, 12.0, 100000, 6); bad += run(40, 9.0, 100000, 7); bad += run(300, 40.0, 100000, 9);
    bad += run(21, 2.5, 1, 13); bad += run(21, 2.5, 7, 14); bad += run(21, 0.05, 50000, 15); bad += run(1, 1.0, 1000, 16);
    printf(bad ? "FAILURES %d\n" : "all good\n", bad);
    return bad;
}
