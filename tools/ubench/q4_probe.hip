// Probes behind the round-5 "quad-block" children-level kernels (v_mfma_f32_4x4x1_16b_f32):
//   1. issue rate / dependent-accumulator latency: CH independent accumulator chains, one or two waves per SIMD
//   2. straight-line code vs a loop of the same instruction count (instruction-cache behaviour of a long unrolled body)
//   3. the same MFMA stream next to the LDS traffic of the real kernel: per 32 MFMAs 4 broadcast ds_read_b128 (B operand) and, per
//      108 MFMAs, 8 ds_read_b128 (A operand) + 8 `buffer_load_dwordx4 ... lds` (row gather of one cell for 128 parents)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/q4_probe tools/ubench/q4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int CH, int UNROLL>
__global__ void __launch_bounds__(512) k_rate(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_nop 0" : "+v"(acc[0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int c = 0; c < CH; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the real kernel's operand traffic: table (7 KB) + a ring per wave; per "cell": wait the gather, 8 A reads, refill, then G groups of
// (4 B reads + 32 MFMAs)
template <int G>
__global__ void __launch_bounds__(512) k_mix(const float* __restrict__ in, int64_t in_bytes, float* out, int cells, int do_dma, int do_lds,
                                            unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* tab = (float4*)lds_raw;                                                // 7 KB table
    for (int i = threadIdx.x; i < 448; i += blockDim.x) tab[i] = make_float4(1e-3f * i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4* ring = (float4*)(lds_raw + 7168) + wave * 1024;                        // 16 KB per wave: 2 slots x 2 tiles x 4 KB
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    f32x4 acc[2][8];
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 8; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + (lane & 3) * 16);
    const unsigned a_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + lane * 16);
    unsigned voff = (unsigned)((blockIdx.x * 8 + wave) * 65536u + (lane >> 2) * 512u + (lane & 3) * 16u);
    auto issue = [&](int slot) {
        if (!do_dma) return;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(ring + slot * 512 + q * 64), 16, (int)(voff + q * 8192), 0, 0, 0);
        voff = (voff + 64u * 37u) % (unsigned)(in_bytes - (1 << 20));
        voff &= ~15u; voff |= (lane & 3) * 16u;
    };
    issue(0); issue(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < cells; ++c) {
        if (do_dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        f32x4 a[2][4];
        if (do_lds) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[m][e]) : "v"(a_lane + ((c & 1) * 8192 + m * 4096)), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            for (int m = 0; m < 2; ++m) for (int e = 0; e < 4; ++e) a[m][e] = (f32x4){1.f, 2.f, 3.f, 4.f};
        }
        issue(c & 1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 b[4];
            if (do_lds) {
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[e]) : "v"(tab_lane), "n"(256 * 3) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                for (int e = 0; e < 4; ++e) b[e] = (f32x4){1.f, 2.f, 3.f, 4.f};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][e][u], b[e][u], acc[m][g], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 8; ++j) r += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}


#include <type_traits>
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_ld128_off(unsigned base) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void tie(f32x4& v) { asm volatile("" : "+v"(v)); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// the pipelined form: B of group n+1 requested before the MFMAs of group n (two register sets); the A operand of cell c+1 requested
// INSIDE the last group of cell c, each quarter (4 channels, both M tiles) right behind the MFMAs that consumed that quarter's
// registers; counted lgkmcnt waits (LDS operations return in order).  A body = two cells (static register-set parity).
template <int G>
__global__ void __launch_bounds__(512) k_mix2(const float* __restrict__ in, int64_t in_bytes, float* out, int cells, int do_dma,
                                             unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* tab = (float4*)lds_raw;
    for (int i = threadIdx.x; i < 448; i += blockDim.x) tab[i] = make_float4(1e-3f * i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4* ring = (float4*)(lds_raw + 7168) + wave * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
    f32x4 acc[2][8];
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 8; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + (lane & 3) * 16);
    const unsigned a_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + lane * 16);
    unsigned voff = (unsigned)((blockIdx.x * 8 + wave) * 65536u + (lane >> 2) * 512u + (lane & 3) * 16u);
    auto issue = [&](int slot) {
        if (!do_dma) return;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(ring + slot * 512 + q * 64), 16, (int)(voff + q * 8192), 0, 0, 0);
        voff = (voff + 64u * 37u) % (unsigned)(in_bytes - (1 << 20));
        voff &= ~15u; voff |= (lane & 3) * 16u;
        asm volatile("" ::: "memory");
    };
    f32x4 a[2][4], b[2][4];
    auto load_a = [&](auto ie, int slot) {
        constexpr int e = decltype(ie)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
            a[m][e] = lds_ld128_off<e * 16>(a_lane + (slot * 8192 + m * 4096));
    };
    auto load_b = [&](auto ibuf) {
        constexpr int buf = decltype(ibuf)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[buf][e] = lds_ld128_off<256 * 3>(tab_lane);
    };
    issue(0); issue(1);
    if (do_dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    static_for<0, 4>([&](auto ie) { load_a(ie, 0); });
    load_b(std::integral_constant<int, 0>{});
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < cells; c += 2) {
        static_for<0, 2 * G>([&](auto in_) {
            constexpr int n = decltype(in_)::value, ci = n / G, g = n % G, cur = n & 1;
            constexpr bool first = g == 0, last = g == G - 1;
            // at entry outstanding (oldest first): B(n) x4 [+ A quarters 0..3 x2 each if first]
            if constexpr (first) wait_lgkm<6>(); else wait_lgkm<0>();
            static_for<0, 4>([&](auto ie) { tie(b[cur][decltype(ie)::value]); });
            if constexpr (first) { tie(a[0][0]); tie(a[1][0]); }
            load_b(std::integral_constant<int, cur ^ 1>{});
            if constexpr (last) { if (do_dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // the next cell's rows have landed
            static_for<0, 4>([&](auto ie) {
                constexpr int e = decltype(ie)::value;
                if constexpr (first && e > 0) {                                    // younger than quarter e: quarters e+1.. (2 each) + B(n+1) (4)
                    __builtin_amdgcn_sched_barrier(0);
                    wait_lgkm<(3 - e) * 2 + 4>();
                    tie(a[0][e]); tie(a[1][e]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][e][u], b[cur][e][u], acc[m][g], 0, 0, 0);
                if constexpr (last) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(ie, (c + ci + 1) & 1);
                }
                if constexpr (e == 3) __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (first) {                                                  // this cell's ring slot has been read (by the previous cell's last group): refill it
                // (placed after the first group's MFMAs: the A reads of this cell were waited for above)
                issue((c + ci) & 1);
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 8; ++j) r += acc[m][j][0] + acc[m][j][1] + acc[m][j][2] + acc[m][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float* out; static unsigned long long* cyc;
template <int CH, int UNROLL> void rate(const char* name, int waves_per_simd, int total_per_chain) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    const int iters = total_per_chain / UNROLL;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<CH, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, 4, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<CH, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * UNROLL * CH;                                  // MFMAs per wave
    printf("%-44s %2d waves/SIMD  %8.1f us  %6.2f cyc/MFMA/SIMD (wave 0: %6.2f cyc per own MFMA)  %6.1f TFLOP/s\n", name, waves_per_simd, ms * 1e3,
           (double)c / (n * waves_per_simd), (double)c / n, n * waves_per_simd * 1024 * 512 / ms / 1e9);
}
template <int G> void mix(const char* name, const float* in, int64_t in_bytes, int do_dma, int do_lds) {
    const int cells = 256, blocks = 256, threads = 512;
    const size_t lds = 7168 + 8 * 16384;
    hipFuncSetAttribute((const void*)k_mix<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<G>), dim3(blocks), dim3(threads), lds, 0, in, in_bytes, out, 4, do_dma, do_lds, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<G>), dim3(blocks), dim3(threads), lds, 0, in, in_bytes, out, cells, do_dma, do_lds, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)cells * G * 32;
    printf("%-44s dma %d lds %d  %8.1f us  %6.2f cyc/MFMA/SIMD (2 waves per SIMD)  %6.1f TFLOP/s\n", name, do_dma, do_lds, ms * 1e3, (double)c / (2 * n),
           n * 2 * 1024 * 512 / ms / 1e9);
}
template <int G> void mix2(const char* name, const float* in, int64_t in_bytes, int do_dma, int wgs = 256) {
    const int cells = 256, blocks = wgs, threads = 512;
    const size_t lds = 7168 + 8 * 16384;
    hipFuncSetAttribute((const void*)k_mix2<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix2<G>), dim3(blocks), dim3(threads), lds, 0, in, in_bytes, out, 4, do_dma, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix2<G>), dim3(blocks), dim3(threads), lds, 0, in, in_bytes, out, cells, do_dma, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)cells * G * 32;
    printf("pipelined %-34s dma %d        %8.1f us  %6.1f TFLOP/s (%.0f of them issued per us and SIMD)\n", name, do_dma, ms * 1e3, n * 2 * 4 * blocks * 512 / ms / 1e9, n * 2 / (ms * 1e3));
}
int main() {
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int64_t in_bytes = 128ll << 20;
    float* in; hipMalloc(&in, in_bytes); hipMemset(in, 0, in_bytes);
    printf("# 1. chains (looped body of 64 MFMAs per chain-set)\n");
    rate<1, 64>("1 chain", 1, 16384); rate<2, 32>("2 chains", 1, 16384); rate<4, 16>("4 chains", 1, 16384); rate<8, 8>("8 chains", 1, 16384);
    rate<1, 64>("1 chain", 2, 16384); rate<2, 32>("2 chains", 2, 16384); rate<4, 16>("4 chains", 2, 16384);
    printf("# 2. code size: 4 chains, straight-line bodies of 64 / 1024 / 4096 / 8192 MFMAs (0.5 / 8 / 32 / 64 KB), ONE pass over 8192 MFMAs per wave\n");
    rate<4, 16>("body 64 MFMAs, looped 128x", 2, 2048); rate<4, 256>("body 1024, looped 8x", 2, 2048); rate<4, 1024>("body 4096, looped 2x", 2, 2048);
    rate<4, 2048>("body 8192, once", 2, 2048);
    printf("# 3. MFMA stream + operand traffic of the quad-block kernel (G groups of 2 x 16 MFMAs per cell, 64 cells)\n");
    mix<3>("3 groups per cell", in, in_bytes, 0, 0); mix<3>("3 groups per cell", in, in_bytes, 0, 1); mix<3>("3 groups per cell", in, in_bytes, 1, 1);
    mix<4>("4 groups per cell", in, in_bytes, 0, 0); mix<4>("4 groups per cell", in, in_bytes, 0, 1); mix<4>("4 groups per cell", in, in_bytes, 1, 1);
    mix<1>("1 group per cell (corner cells)", in, in_bytes, 1, 1); mix<8>("8 groups per cell (interior)", in, in_bytes, 1, 1);
    printf("# 4. the same with software pipelining (B one group ahead, next cell's A inside the last group, counted waits)\n");
    mix2<2>("2 groups per cell", in, in_bytes, 0); mix2<2>("2 groups per cell", in, in_bytes, 1);
    mix2<3>("3 groups per cell", in, in_bytes, 0); mix2<3>("3 groups per cell", in, in_bytes, 1);
    mix2<4>("4 groups per cell", in, in_bytes, 0); mix2<4>("4 groups per cell", in, in_bytes, 1);
    mix2<1>("1 group per cell", in, in_bytes, 0); mix2<1>("1 group per cell", in, in_bytes, 1);
    return 0;
}
