// Stream compaction (MinkowskiPruning, autoencoder.py:237,247), top-k mask (istopk, data_utils.py:77-89) and the
// canonical z-major ordering (sort_spare_tensor / array2vector, data_utils.py:55-61,91-101; coder.py:97-99).
#include <cstring>
#include "pcgc_common.h"
#include <rocprim/rocprim.hpp>

// ------------------------------------------------------------------------------------------- mask scan
// Three launches: per-tile popcounts -> single-block scan of tile sums -> per-tile exclusive scan.
// Tile = 2048 mask bytes per 256-thread block (8 per thread, loaded as one 8-byte word).
constexpr int SCAN_TILE = 2048;

__device__ static inline int block_exclusive_scan_256(int v, int* total_out) {
    // wave scan (64 lanes) + 4-wave combine through LDS
    __shared__ int wave_sums[4];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int s = wave_sums[i]; if (i < w) base += s; tot += s; }
    __syncthreads();
    if (total_out) *total_out = tot;
    return base + incl - v;
}

__device__ static inline int load_mask8(const uint8_t* mask, int64_t n, int64_t base, uint8_t m[8]) {
    int cnt = 0;
    if (base + 8 <= n && ((uintptr_t)(mask + base) & 7) == 0) {
        uint64_t w = *(const uint64_t*)(mask + base);
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = ((w >> (8 * j)) & 0xff) ? 1 : 0; cnt += m[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = (base + j < n && mask[base + j]) ? 1 : 0; cnt += m[j]; }
    }
    return cnt;
}

__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint8_t* __restrict__ mask, int64_t n, int32_t* tile_sums) {
    uint8_t m[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int cnt = load_mask8(mask, n, base, m);
    int tot;
    block_exclusive_scan_256(cnt, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_tile_offsets(int32_t* tile_sums, int64_t n_tiles, int32_t* total) {
    // single block, sequential over chunks of 256 tiles
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t c = 0; c < n_tiles; c += 256) {
        int64_t i = c + threadIdx.x;
        int v = i < n_tiles ? tile_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan_256(v, &tot);
        int carry = carry_s;
        if (i < n_tiles) tile_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint8_t* __restrict__ mask, int64_t n,
                                                    const int32_t* __restrict__ tile_offsets, int32_t* __restrict__ prefix) {
    uint8_t m[8];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    int cnt = load_mask8(mask, n, base, m);
    int ex = block_exclusive_scan_256(cnt, nullptr) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (base + j < n) prefix[base + j] = ex; ex += m[j]; }
}

extern "C" size_t pcgc_scan_workspace_bytes(int64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(int32_t); }

extern "C" int pcgc_mask_scan(const uint8_t* mask, int64_t n, int32_t* prefix, int32_t* total, void* workspace,
                              size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_scan_workspace_bytes(n), "workspace too small");
    if (n == 0) { (void)hipMemsetAsync(total, 0, 4, S(stream)); return 0; }
    int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int32_t* ts = (int32_t*)workspace;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)tiles), dim3(256), 0, S(stream), mask, n, ts);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(256), 0, S(stream), ts, tiles, total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)tiles), dim3(256), 0, S(stream), mask, n, ts, prefix);
    PCGC_CHECK_LAUNCH("mask_scan");
    return 0;
}

__global__ void k_compact_coords(const int4* __restrict__ in, const uint8_t* __restrict__ mask,
                                 const int32_t* __restrict__ prefix, int64_t n, int4* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) out[prefix[i]] = in[i];
}
// one thread per (row, 4-float chunk); C % 4 == 0 fast path, scalar otherwise
__global__ void k_compact_feats4(const float* __restrict__ in, int C4, int in_ld, const uint8_t* __restrict__ mask,
                                 const int32_t* __restrict__ prefix, int64_t n, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / C4; int c = (int)(t % C4);
    if (i < n && mask[i])
        *(float4*)(out + (int64_t)prefix[i] * (C4 * 4) + 4 * c) = *(const float4*)(in + i * in_ld + 4 * c);
}
__global__ void k_compact_feats1(const float* __restrict__ in, int C, int in_ld, const uint8_t* __restrict__ mask,
                                 const int32_t* __restrict__ prefix, int64_t n, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / C; int c = (int)(t % C);
    if (i < n && mask[i]) out[(int64_t)prefix[i] * C + c] = in[i * in_ld + c];
}
extern "C" int pcgc_compact_coords(const int32_t* coords, const uint8_t* mask, const int32_t* prefix, int64_t n,
                                   int32_t* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_compact_coords, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, mask, prefix,
                       n, (int4*)out);
    PCGC_CHECK_LAUNCH("compact_coords");
    return 0;
}
extern "C" int pcgc_compact_feats(const float* in, int C, int in_ld, const uint8_t* mask, const int32_t* prefix, int64_t n,
                                  float* out, void* stream) {
    if (n == 0) return 0;
    if (C % 4 == 0 && in_ld % 4 == 0)
        hipLaunchKernelGGL(k_compact_feats4, dim3(grid_for(n * (C / 4), 256)), dim3(256), 0, S(stream), in, C / 4, in_ld, mask,
                           prefix, n, out);
    else
        hipLaunchKernelGGL(k_compact_feats1, dim3(grid_for(n * C, 256)), dim3(256), 0, S(stream), in, C, in_ld, mask, prefix,
                           n, out);
    PCGC_CHECK_LAUNCH("compact_feats");
    return 0;
}

// ------------------------------------------------------------------------------------------- top-k mask
// MSB-first radix select on the order-preserving integer image of the fp32 logits: 4 passes of 8 bits, each a
// block-privatised LDS histogram + a one-block digit pick.  Then mask = key > T, plus the first `need` rows (by
// index) among key == T, ranked by an exclusive scan of the equality flags (canonical tie rule: lower row wins).
struct TopkState { uint32_t prefix; uint32_t pad; int64_t k_remaining; };   // lives at workspace[0]

__device__ static inline uint32_t order_key(float f) {
    f = f + 0.0f;                                      // -0.0 -> +0.0
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u); // ascending in float order
}
__global__ void k_topk_init(TopkState* st, uint32_t* hist, int64_t k) {
    if (threadIdx.x == 0) { st->prefix = 0; st->k_remaining = k; }
    hist[threadIdx.x] = 0;
}
// Block-local LDS histogram, flushed with one global atomic per non-empty bin.  The grid is kept SMALL (<= 256 blocks):
// the flush is up to 256 same-address atomics per block, and with 2048 blocks those serialised in L2 for ~30 us per pass
// on the 2 M-candidate level (the element loop itself is ~3 us).
__global__ void __launch_bounds__(256) k_topk_hist(const float* __restrict__ v, int ld, int64_t n, const TopkState* st,
                                                   int pass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int shift = 24 - 8 * pass;
    uint32_t prefix = st->prefix;
    uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t key = order_key(v[i * ld]);
        if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_topk_pick(TopkState* st, uint32_t* hist, int pass) {
    // digit d = the largest one whose inclusive suffix count S[d] = sum_{e >= d} hist[e] reaches k_remaining (d = 0 if none):
    // parallel suffix scan over the 256 bins instead of a serial walk of dependent global loads
    __shared__ int64_t S[257];
    const int t = threadIdx.x;
    const int64_t mine = hist[t];
    S[t] = mine;
    if (t == 0) S[256] = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int64_t add = t + off < 256 ? S[t + off] : 0;
        __syncthreads();
        S[t] += add;
        __syncthreads();
    }
    const int64_t need = st->k_remaining;
    __syncthreads();                                   // every thread has read k_remaining before it is rewritten
    const bool hit = t == 0 ? (S[1] < need) : (S[t] >= need && (S[t + 1] < need || t == 255));     // (t = 255 also covers k = 0)
    if (hit) {
        st->prefix |= (uint32_t)t << (24 - 8 * pass);
        st->k_remaining = need - S[t + 1];             // how many to take among keys sharing the new prefix
    }
    hist[t] = 0;                                       // ready for the next pass
}
__global__ void k_topk_flags(const float* __restrict__ v, int ld, int64_t n, const TopkState* st, uint8_t* eq) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eq[i] = order_key(v[i * ld]) == st->prefix;
}
// tie_high = 0: among logits equal to the threshold the LOWER row indices are kept (canonical); 1: the HIGHER ones
__global__ void k_topk_mask(const float* __restrict__ v, int ld, int64_t n, const TopkState* st,
                            const int32_t* __restrict__ eq_rank, const int32_t* __restrict__ eq_total, int tie_high, uint8_t* mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t key = order_key(v[i * ld]), T = st->prefix;
    const int64_t need = st->k_remaining, r = eq_rank[i];
    mask[i] = (key > T) || (key == T && (tie_high ? r >= (int64_t)eq_total[0] - need : r < need));
}

static int g_topk_tie_high = 0;
// ‡ conventions of the un-vendored dependencies that the reference's results depend on but its sources do not pin:
//   what = 0  top-k tie rule (data_utils.py:85-87, torch.topk on ME's row order): value 0 = lower row wins (default), 1 = higher row wins
// (the dedup policy is an argument of pcgc_hash_insert_policy; the kernel-offset order is a weight permutation done by the host).
extern "C" int pcgc_set_convention(int what, int value) {
    if (what == 0) { g_topk_tie_high = value ? 1 : 0; return 0; }
    pcgc_set_error("set_convention: unknown convention %d", what);
    return -2;
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t pcgc_topk_workspace_bytes(int64_t n) {
    return 256 + 1024 + align256((size_t)n) + align256((size_t)n * 4) + 256 + align256(pcgc_scan_workspace_bytes(n));
}
extern "C" int pcgc_topk_mask(const float* logits, int ld, int64_t n, int64_t k, uint8_t* mask, void* workspace,
                              size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_topk_workspace_bytes(n), "workspace too small");
    if (n == 0) return 0;
    if (k >= n) { (void)hipMemsetAsync(mask, 1, (size_t)n, S(stream)); return 0; }
    if (k <= 0) { (void)hipMemsetAsync(mask, 0, (size_t)n, S(stream)); return 0; }
    char* ws = (char*)workspace;
    TopkState* st = (TopkState*)ws; ws += 256;
    uint32_t* hist = (uint32_t*)ws; ws += 1024;
    uint8_t* eq = (uint8_t*)ws; ws += align256((size_t)n);
    int32_t* rank = (int32_t*)ws; ws += align256((size_t)n * 4);
    int32_t* total = (int32_t*)ws; ws += 256;
    void* scan_ws = ws;
    unsigned g = grid_for(n, 256 * 8); if (g > 256) g = 256; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_topk_init, dim3(1), dim3(256), 0, S(stream), st, hist, k);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(k_topk_hist, dim3(g), dim3(256), 0, S(stream), logits, ld, n, st, pass, hist);
        hipLaunchKernelGGL(k_topk_pick, dim3(1), dim3(256), 0, S(stream), st, hist, pass);
    }
    hipLaunchKernelGGL(k_topk_flags, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), logits, ld, n, st, eq);
    int rc = pcgc_mask_scan(eq, n, rank, total, scan_ws, pcgc_scan_workspace_bytes(n), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_topk_mask, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), logits, ld, n, st, rank, total, g_topk_tie_high, mask);
    PCGC_CHECK_LAUNCH("topk_mask");
    return 0;
}

// ------------------------------------------------------------------------------------------- z-major sort
// array2vector(C, C.max()+1) orders rows by (z, y, x, batch), z most significant (data_utils.py:55-61); every field
// is < step, so sorting the packed 64-bit key (z<<44 | y<<24 | x<<4 | batch) gives the same permutation.
__global__ void k_zyx_keys(const int4* __restrict__ coords, int64_t n, uint64_t* keys, int32_t* idx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    keys[i] = ((uint64_t)(uint32_t)c.w << 44) | ((uint64_t)(uint32_t)c.z << 24) | ((uint64_t)(uint32_t)c.y << 4) | (uint64_t)(uint32_t)c.x;
    idx[i] = (int32_t)i;
}
static size_t sort_temp_bytes(int64_t n) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs((void*)nullptr, tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                              (size_t)n, 0, 64, (hipStream_t)0);
    return tmp;
}
extern "C" size_t pcgc_sort_workspace_bytes(int64_t n) {
    if (n < 1) n = 1;
    return align256((size_t)n * 8) * 2 + align256((size_t)n * 4) + align256(sort_temp_bytes(n));
}
extern "C" int pcgc_sort_zyx(const int32_t* coords, int64_t n, int32_t* perm, void* workspace, size_t workspace_bytes,
                             void* stream) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_sort_workspace_bytes(n), "workspace too small");
    if (n == 0) return 0;
    char* ws = (char*)workspace;
    uint64_t* kin = (uint64_t*)ws; ws += align256((size_t)n * 8);
    uint64_t* kout = (uint64_t*)ws; ws += align256((size_t)n * 8);
    int32_t* idx = (int32_t*)ws; ws += align256((size_t)n * 4);
    size_t tmp = sort_temp_bytes(n);
    hipLaunchKernelGGL(k_zyx_keys, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n, kin, idx);
    hipError_t e = rocprim::radix_sort_pairs((void*)ws, tmp, kin, kout, idx, perm, (size_t)n, 0, 64, S(stream));
    if (e != hipSuccess) { pcgc_set_error("sort_zyx: %s", hipGetErrorString(e)); return -1; }
    PCGC_CHECK_LAUNCH("sort_zyx");
    return 0;
}

__global__ void k_gather_i32x4(const int4* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, int4* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}
__global__ void k_gather_f32(const float* __restrict__ in, int C, const int32_t* __restrict__ perm, int64_t n, float* out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = t / C; int c = (int)(t % C);
    if (i < n) out[i * C + c] = in[(int64_t)perm[i] * C + c];
}
extern "C" int pcgc_gather_rows_i32x4(const int32_t* in, const int32_t* perm, int64_t n, int32_t* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gather_i32x4, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)in, perm, n, (int4*)out);
    PCGC_CHECK_LAUNCH("gather_rows_i32x4");
    return 0;
}
extern "C" int pcgc_gather_rows_f32(const float* in, int C, const int32_t* perm, int64_t n, float* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gather_f32, dim3(grid_for(n * C, 256)), dim3(256), 0, S(stream), in, C, perm, n, out);
    PCGC_CHECK_LAUNCH("gather_rows_f32");
    return 0;
}
