// Probe of v_mfma_f32_4x4x1_16b_f32: operand/result lane layout and whether the accumulate is a single-rounding fma.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float float4_ __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, const float* c, float* d, int steps) {
    const int lane = threadIdx.x;
    float4_ acc; for (int i = 0; i < 4; ++i) acc[i] = c[lane * 4 + i];
    for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s * 64 + lane], b[s * 64 + lane], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}
int main() {
    const int steps = 16;
    std::vector<float> a(64 * steps), b(64 * steps), c(256), d(256);
    std::mt19937 g(3); std::uniform_real_distribution<float> u(-2.f, 2.f);
    for (auto& v : a) v = u(g) * 1.2345f; for (auto& v : b) v = u(g); for (auto& v : c) v = u(g) * 1e-3f;
    float *da, *db, *dc, *dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dc, dd, steps);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: block = lane / 4; A row i = lane % 4; B col j = lane % 4; D[i][j] in vgpr i of lane 4*block + j
    int bad_fma = 0, bad_unfused = 0;
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float f = c[(4 * blk + j) * 4 + i], m = f;
        for (int s = 0; s < steps; ++s) { const float av = a[s * 64 + 4 * blk + i], bv = b[s * 64 + 4 * blk + j]; f = fmaf(av, bv, f); volatile float p = av * bv; m = m + p; }
        const float got = d[(4 * blk + j) * 4 + i];
        bad_fma += got != f; bad_unfused += got != m;
    }
    printf("layout+fma mismatches: %d of 256 (unfused-model mismatches: %d)\n", bad_fma, bad_unfused);
    return 0;
}
