// microbenchmark: issue rate of v_fma_f32 (VGPR and SGPR operand) vs v_pk_fma_f32 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* wsrc, int iters) {
    float a[16]; f2 p[8];
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < 16; ++i) a[i] = i;
    for (int i = 0; i < 8; ++i) p[i] = (f2){(float)i, (float)-i};
    const float s0 = wsrc[0], s1 = wsrc[1];           // wave-uniform -> SGPRs
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(a[(i + 1) & 15]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(x));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        } else {
            f2 ss = {s0, s1};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "s"(ss));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, float* out, float* w) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)blocks * 256 * iters * 16;    // 16 scalar FMAs per iteration per lane in every mode
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, 2 * fma / ms / 1e9);
}
int main() {
    float *out, *w; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&w, 64); hipMemset(w, 0, 64);
    run<0>("v_fma_f32 vgpr", out, w); run<1>("v_fmac_f32 sgpr operand", out, w);
    run<2>("v_pk_fma_f32 vgpr", out, w); run<3>("v_pk_fma_f32 sgpr+op_sel", out, w);
    return 0;
}
