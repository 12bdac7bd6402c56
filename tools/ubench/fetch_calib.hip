// FETCH_SIZE calibration for gather patterns: rocprofv3's FETCH_SIZE on gfx950 is documented to report half the
// bytes of wide coalesced streaming reads; is the x2 correction also right for the 64 / 128 / 256-byte row-segment gathers of the
// sparse-conv kernels (buffer_load_dwordx4 ... lds, 4 / 8 / 16 adjacent lanes per row)?  Each kernel touches a KNOWN byte count:
// every row of a 2 GiB buffer exactly once (far beyond the 256 MiB Infinity Cache), rows visited in a pseudo-random order.
//   k_stream      64 lanes x 16 B contiguous per instruction                    bytes = rows * W
//   k_gather<W>   rows of W bytes at permuted positions, W/16 adjacent lanes per row, LDS-DMA like the conv kernels
// run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib      then compare FETCH_SIZE (KB) x 1024 with the printed bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ in, int64_t n16, float* sink) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) { float4 v = in[i]; acc += v.x + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
// permuted row index: multiply by an odd constant modulo a power-of-two row count (bijective)
__device__ static inline uint32_t perm(uint32_t r, uint32_t mask) { return (r * 2654435761u + 12345u) & mask; }

template <int W>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ in, uint32_t rows_mask, int64_t rows, float* sink) {
    constexpr int LPR = W / 16;                    // lanes per row
    constexpr int RPI = 64 / LPR;                  // rows per DMA instruction
    __shared__ __attribute__((aligned(16))) float4 buf[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 0xFFFFFFF0u > (uint64_t)rows * W ? (int)(rows * W) : (int)0xFFFFFFF0u, 0x00020000);
    float acc = 0.f;
    const int64_t instrs = rows / RPI;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < instrs; t += (int64_t)gridDim.x * 4) {
        const uint32_t r = perm((uint32_t)(t * RPI + lane / LPR), rows_mask);
        const unsigned voff = (unsigned)((uint64_t)r * W + (lane % LPR) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(&buf[wave][0]), 16, (int)voff, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[wave][lane].x;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;           // 2 GiB
    float *buf, *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const float4*)buf, (int64_t)(bytes / 16), sink);
        hipLaunchKernelGGL((k_gather<32>), dim3(4096), dim3(256), 0, 0, buf, (uint32_t)(bytes / 32 - 1), (int64_t)(bytes / 32), sink);
        hipLaunchKernelGGL((k_gather<64>), dim3(4096), dim3(256), 0, 0, buf, (uint32_t)(bytes / 64 - 1), (int64_t)(bytes / 64), sink);
        hipLaunchKernelGGL((k_gather<128>), dim3(4096), dim3(256), 0, 0, buf, (uint32_t)(bytes / 128 - 1), (int64_t)(bytes / 128), sink);
        hipLaunchKernelGGL((k_gather<256>), dim3(4096), dim3(256), 0, 0, buf, (uint32_t)(bytes / 256 - 1), (int64_t)(bytes / 256), sink);
    }
    hipDeviceSynchronize();
    printf("each kernel reads %zu bytes (every row of the 2 GiB buffer once)\n", bytes);
    return 0;
}
