// Shared helpers for libpcgc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pcgc_hip.h"

void pcgc_set_error(const char* fmt, ...);

#define PCGC_CHECK_LAUNCH(name)                                                            \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) { pcgc_set_error("%s: %s", name, hipGetErrorString(e__)); return -1; } \
    } while (0)

#define PCGC_REQUIRE(cond, msg)                                                            \
    do { if (!(cond)) { pcgc_set_error("%s: %s", __func__, msg); return -2; } } while (0)

static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// ---- coordinate key: 4-bit batch | 20-bit z | 20-bit y | 20-bit x ------------------------------------------
#define PCGC_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
__host__ __device__ static inline bool coord_in_range(int32_t b, int32_t x, int32_t y, int32_t z) {
    return ((uint32_t)x < (1u << 20)) && ((uint32_t)y < (1u << 20)) && ((uint32_t)z < (1u << 20)) && ((uint32_t)b < 16u);
}
__host__ __device__ static inline uint64_t coord_key(int32_t b, int32_t x, int32_t y, int32_t z) {
    return ((uint64_t)b << 60) | ((uint64_t)z << 40) | ((uint64_t)y << 20) | (uint64_t)x;
}
// Spatially blocked hash: a 4x4x4 voxel block (in units of the level's own lattice: callers pre-divide by stride)
// maps to 64 consecutive slots, so the 27 probes of neighbouring outputs share cache lines.
__device__ static inline uint64_t mix64(uint64_t v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33; return v;
}
__device__ static inline uint64_t hash_slot(int32_t b, int32_t x, int32_t y, int32_t z, int sh, uint64_t cap_mask) {
    // sh = log2(stride) of the level; lattice coordinates are (x>>sh) etc.
    uint32_t lx = (uint32_t)x >> sh, ly = (uint32_t)y >> sh, lz = (uint32_t)z >> sh;
    uint64_t blk = ((uint64_t)b << 54) | ((uint64_t)(lz >> 2) << 36) | ((uint64_t)(ly >> 2) << 18) | (uint64_t)(lx >> 2);
    uint64_t local = ((lz & 3) << 4) | ((ly & 3) << 2) | (lx & 3);
    return ((mix64(blk) << 6) | local) & cap_mask;
}
__device__ static inline int32_t hash_lookup(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                             uint64_t cap_mask, int sh, int32_t b, int32_t x, int32_t y, int32_t z) {
    if (!coord_in_range(b, x, y, z)) return -1;
    uint64_t key = coord_key(b, x, y, z);
    uint64_t h = hash_slot(b, x, y, z, sh, cap_mask);
    for (;;) {
        uint64_t k = keys[h];
        if (k == key) return vals[h];
        if (k == PCGC_EMPTY_KEY) return -1;
        h = (h + 1) & cap_mask;
    }
}
