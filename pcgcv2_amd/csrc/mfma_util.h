// Shared device helpers of the LDS-DMA gather / fp32-MFMA kernels (conv.hip, child.hip).
#pragma once
#include "pcgc_common.h"

typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int N>
__device__ static inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ static inline void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ static inline void lane_transpose4(float4& v) {
    // in: lane-quarter q holds components (c = 0..3) = element [q][c]; out: component c of quarter q = element [c][q]
    unsigned x = __float_as_uint(v.x), y = __float_as_uint(v.y), z = __float_as_uint(v.z), w = __float_as_uint(v.w);
    auto r0 = __builtin_amdgcn_permlane32_swap(x, z, false, false); x = r0[0]; z = r0[1];   // quarter bit 1 <-> component bit 1
    auto r1 = __builtin_amdgcn_permlane32_swap(y, w, false, false); y = r1[0]; w = r1[1];
    auto r2 = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r2[0]; y = r2[1];   // quarter bit 0 <-> component bit 0
    auto r3 = __builtin_amdgcn_permlane16_swap(z, w, false, false); z = r3[0]; w = r3[1];
    v = make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
}

__device__ static inline f32x4 lds_ld128_raw(const float4* p) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_void_ptr)p) : "memory");
    return v;
}
template <int BYTE_OFF>
__device__ static inline float lds_ld32_raw(const float* p) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_void_ptr)p), "n"(BYTE_OFF) : "memory");
    return v;
}
__device__ static inline void lds_tie(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ static inline void lds_tie(float& v) { asm volatile("" : "+v"(v)); }
