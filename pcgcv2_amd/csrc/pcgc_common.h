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

// XCD-aware tile order: workgroup b is observed to run on XCD b % 8 (each XCD has its own 4 MiB L2).  Remap so every XCD
// walks a contiguous slab of tiles: neighbouring tiles gather overlapping rows, which then hit the same L2.  Bijective
// for any grid size; placement only affects speed, never results.
__device__ static inline unsigned xcd_tile(unsigned b, unsigned nb) {
    const unsigned q = nb >> 3, r = nb & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---- coordinate key: 4-bit batch | 20-bit z | 20-bit y | 20-bit x ------------------------------------------
#define PCGC_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
__host__ __device__ static inline bool coord_in_range(int32_t b, int32_t x, int32_t y, int32_t z) {
    // (the one key that equals PCGC_EMPTY_KEY — batch 15 at the far corner of the 2^20 cube — is out of range too)
    return ((uint32_t)x < (1u << 20)) && ((uint32_t)y < (1u << 20)) && ((uint32_t)z < (1u << 20)) && ((uint32_t)b < 16u) &&
           !(b == 15 && (x & y & z) == 0xFFFFF);
}
__host__ __device__ static inline uint64_t coord_key(int32_t b, int32_t x, int32_t y, int32_t z) {
    return ((uint64_t)b << 60) | ((uint64_t)z << 40) | ((uint64_t)y << 20) | (uint64_t)x;
}
// Open addressing with linear probing on a mixed key.  (Round 1 mapped each 4x4x4 block of cells to 64 consecutive slots so that the 27
// probes of a row would share cache lines.  Measured on MI355X it lost everywhere: surfaces fill a block to ~15 %, several blocks overlay
// one slot group and probe chains — above all those of ABSENT neighbours, which run to the next empty slot — get long, and the inserts of
// neighbouring rows queue up on the same few lines: 18.7 k keys inserted in 63 us vs 12 us, their 27-neighbour probe 34 us vs 8 us.)
__device__ static inline uint64_t mix64(uint64_t v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33; return v;
}
__device__ static inline uint64_t hash_slot(uint64_t key, uint64_t cap_mask) { return mix64(key) & cap_mask; }
__device__ static inline int32_t hash_lookup(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                             uint64_t cap_mask, int32_t b, int32_t x, int32_t y, int32_t z) {
    if (!coord_in_range(b, x, y, z)) return -1;
    uint64_t key = coord_key(b, x, y, z);
    uint64_t h = hash_slot(key, cap_mask);
    for (;;) {
        uint64_t k = keys[h];
        if (k == key) return vals[h];
        if (k == PCGC_EMPTY_KEY) return -1;
        h = (h + 1) & cap_mask;
    }
}

// ---- rank bitmap of a pruned level (pcgc_topk_select): bit m of `bits` = candidate row m survives; wprefix[w] = survivors before row
// 64 w.  -> the survivor's row on the pruned level, or -1
__device__ static inline int32_t sel_rank(const uint64_t* __restrict__ bits, const int32_t* __restrict__ wprefix, int64_t m) {
    const uint64_t w = bits[m >> 6], bit = 1ull << (m & 63);
    return (w & bit) ? wprefix[m >> 6] + (int32_t)__popcll(w & (bit - 1)) : -1;
}

// ---- hierarchical kernel maps: a fine voxel at child slot j = (jx, jy, jz) of its parent, displaced by kernel offset k, lands in the parent
// displaced by kp at child slot jn (coords.hip; conv.hip's unit-input conv derives presence the same way)
__device__ static inline void child_offset(int j, int k, int& kp, int& jn) {
    int tx = (j & 1) + (k % 3 - 1), ty = ((j >> 1) & 1) + ((k / 3) % 3 - 1), tz = (j >> 2) + (k / 9 - 1);
    int px = tx < 0 ? -1 : (tx > 1 ? 1 : 0), py = ty < 0 ? -1 : (ty > 1 ? 1 : 0), pz = tz < 0 ? -1 : (tz > 1 ? 1 : 0);
    kp = (px + 1) + 3 * (py + 1) + 9 * (pz + 1);
    jn = (tx & 1) | ((ty & 1) << 1) | ((tz & 1) << 2);
}

