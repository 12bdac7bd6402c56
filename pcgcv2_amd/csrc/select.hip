// Stream compaction (MinkowskiPruning, autoencoder.py:237,247), top-k mask (istopk, data_utils.py:77-89) and the
// canonical z-major ordering (sort_spare_tensor / array2vector, data_utils.py:55-61,91-101; coder.py:97-99).
#include <algorithm>
#include <cstring>
#include "pcgc_common.h"
#include <rocprim/rocprim.hpp>

// ------------------------------------------------------------------------------------------- mask scan
// One launch: a single-pass scan with decoupled look-back.  Tile = 2048 mask bytes per 256-thread block (8 per thread, loaded
// as one 8-byte word); tiles take their index from a ticket counter (a tile only ever waits for tiles that are already
// running), publish (status, value) descriptors — 1 = the tile's own count, 2 = inclusive prefix — and wave 0 of each tile
// walks back 64 descriptors at a time until it meets an inclusive prefix.  Descriptors and ticket must be zero at launch.
// A descriptor is ONE 64-bit word carrying status and value, read and written with relaxed device-scope atomics: nothing else
// is communicated between tiles, so no fences (on this multi-XCD part an acquire / release at device scope is an L2
// invalidate / write-back: 24 us per scan instead of 6).
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

constexpr uint64_t SCAN_AGG = 1ull << 62, SCAN_INCL = 2ull << 62;
// `enable` (may be null): a device flag; the launch is a no-op when it reads 0 (the top-k tie path below)
__global__ void __launch_bounds__(256) k_scan_lookback(const uint8_t* __restrict__ mask, int64_t n, int64_t n_tiles,
                                                       unsigned long long* desc, int32_t* ticket, int32_t* __restrict__ prefix,
                                                       int32_t* total, const uint32_t* enable) {
    if (enable && *enable == 0) return;
    __shared__ int tile_s, excl_s;
    if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1);
    __syncthreads();
    const int64_t tile = tile_s;
    uint8_t m[8];
    const int64_t base = tile * SCAN_TILE + threadIdx.x * 8;
    const int cnt = load_mask8(mask, n, base, m);
    int tot;
    int ex = block_exclusive_scan_256(cnt, &tot);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&desc[0], SCAN_INCL | (uint64_t)(uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], SCAN_AGG | (uint64_t)(uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t j = tile - 1;; j -= 64) {
                const int64_t idx = j - lane;
                unsigned long long d = SCAN_INCL;                        // before tile 0: an inclusive prefix of 0
                if (idx >= 0) {
                    do { d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((d >> 62) == 0);
                }
                const unsigned long long incl = __ballot((d >> 62) == 2);
                const int first = incl ? (int)__ffsll((long long)incl) - 1 : 63;     // nearest inclusive prefix, or the whole window
                int v = lane <= first ? (int)(uint32_t)d : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                excl += v;
                if (incl) break;
            }
            if (lane == 0) __hip_atomic_store(&desc[tile], SCAN_INCL | (uint64_t)(uint32_t)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { excl_s = excl; if (tile == n_tiles - 1) *total = excl + tot; }
    }
    __syncthreads();
    ex += excl_s;
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (base + j < n) prefix[base + j] = ex; ex += m[j]; }
}

static int64_t scan_tiles(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
extern "C" size_t pcgc_scan_workspace_bytes(int64_t n) { return (size_t)(scan_tiles(n) + 2) * sizeof(uint64_t); }   // descriptors | ticket

// workspace already zeroed by the caller
static void launch_scan(const uint8_t* mask, int64_t n, int32_t* prefix, int32_t* total, void* workspace, const uint32_t* enable, hipStream_t s) {
    const int64_t tiles = scan_tiles(n);
    unsigned long long* desc = (unsigned long long*)workspace;
    hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)tiles), dim3(256), 0, s, mask, n, tiles, desc, (int32_t*)(desc + tiles), prefix, total, enable);
}
// as pcgc_mask_scan, for a workspace the caller has zeroed already (several scans behind one memset)
extern "C" int pcgc_mask_scan_zeroed(const uint8_t* mask, int64_t n, int32_t* prefix, int32_t* total, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_scan_workspace_bytes(n), "workspace too small");
    PCGC_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (n == 0) return 0;                              // (total stays as the caller zeroed it)
    launch_scan(mask, n, prefix, total, workspace, nullptr, S(stream));
    PCGC_CHECK_LAUNCH("mask_scan");
    return 0;
}
extern "C" int pcgc_mask_scan(const uint8_t* mask, int64_t n, int32_t* prefix, int32_t* total, void* workspace,
                              size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_scan_workspace_bytes(n), "workspace too small");
    PCGC_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (n == 0) { (void)hipMemsetAsync(total, 0, 4, S(stream)); return 0; }
    (void)hipMemsetAsync(workspace, 0, pcgc_scan_workspace_bytes(n), S(stream));
    launch_scan(mask, n, prefix, total, workspace, nullptr, S(stream));
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
// MSB-first radix select on the order-preserving integer image of the fp32 logits: 3 passes of 11 / 11 / 10 bits, each a
// block-privatised LDS histogram whose LAST block to finish picks the digit (no separate pick launch).  Then
// mask = key > T, plus the first `need` rows (by index) among key == T (canonical tie rule: lower row wins).  The final
// pick knows how many keys equal T: unless fewer than all of them are needed (a genuine tie at the threshold) every one is
// kept and the ranking launches (equality flags scanned, mask fixed up) return at once.
struct TopkState { uint32_t prefix; uint32_t tie; int64_t k_remaining; uint32_t done; uint32_t count_eq; };   // one per segment, at workspace[0]
// Row segments of a collated batch (data_utils.py:77-89: istopk loops over the batch items, each with its own budget): item b =
// rows [off[b], off[b + 1]).  A single cloud is one segment.  Passed by value: up to 16 items (the coordinate key's 4 batch bits).
constexpr int TOPK_MAX_SEGS = 16;
struct TopkSegs { int n; long long off[TOPK_MAX_SEGS + 1]; long long k[TOPK_MAX_SEGS]; };

__device__ static inline uint32_t order_key(float f) {
    f = f + 0.0f;                                      // -0.0 -> +0.0
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u); // ascending in float order
}
// Digits of the radix select: 3 passes of 11 / 11 / 10 bits (round 4; four passes of 8 bits before: one launch and one sweep over the
// logits less per decoder stage).  TOPK_BINS = the widest digit.
constexpr int TOPK_PASSES = 3, TOPK_BINS = 2048;
__host__ __device__ static inline int topk_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__host__ __device__ static inline int topk_bits(int pass) { return pass == 2 ? 10 : 11; }
// one block per segment: zeroes its state and histogram; block 0 also the tie path's scan workspace and the any-tie flag
__global__ void __launch_bounds__(256) k_topk_init(TopkState* st, uint32_t* hist, TopkSegs segs, unsigned long long* scan_ws, int64_t scan_words, uint32_t* any_tie) {
    const int y = blockIdx.x;
    if (threadIdx.x == 0) { st[y].prefix = 0; st[y].tie = 0; st[y].k_remaining = segs.k[y]; st[y].done = 0; st[y].count_eq = 0; }
    for (int b = threadIdx.x; b < TOPK_BINS; b += blockDim.x) hist[TOPK_BINS * y + b] = 0;
    if (y == 0) {
        if (threadIdx.x == 0) *any_tie = 0;
        for (int64_t i = threadIdx.x; i < scan_words; i += blockDim.x) scan_ws[i] = 0;
    }
}
// Block-local LDS histogram, flushed with one global atomic per non-empty bin.  The grid is kept SMALL (<= 256 blocks per segment):
// the flush is one same-address atomic per non-empty bin and block, and with 2048 blocks those serialised in L2 for ~30 us per pass
// on the 2 M-candidate level (the element loop itself is ~3 us).  blockIdx.y = segment.
__global__ void __launch_bounds__(256) k_topk_hist(const float* __restrict__ v_all, int ld, TopkSegs segs, TopkState* st_all,
                                                   int pass, uint32_t* hist_all, uint32_t* any_tie) {
    __shared__ uint32_t h[TOPK_BINS];
    __shared__ int64_t S[257];
    __shared__ bool last_s;
    __shared__ int pick_s;
    const int t = threadIdx.x, y = blockIdx.y;
    const float* __restrict__ v = v_all + segs.off[y] * ld;
    const int64_t n = segs.off[y + 1] - segs.off[y];
    TopkState* st = st_all + y;
    uint32_t* hist = hist_all + TOPK_BINS * y;
    const int shift = topk_shift(pass), nb = 1 << topk_bits(pass);
    const uint32_t dmask = (uint32_t)nb - 1u;
    for (int b = t; b < nb; b += 256) h[b] = 0;
    __syncthreads();
    const uint32_t prefix = st->prefix;
    const uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + topk_bits(pass)));
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + t;
    for (; i + 7 * step < n; i += 8 * step) {          // eight independent loads in flight per thread (one block per CU: nothing else hides them)
        float f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = v[(i + u * step) * ld];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t key = order_key(f[u]);
            if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & dmask], 1u);
        }
    }
    for (; i < n; i += step) {
        uint32_t key = order_key(v[i * ld]);
        if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & dmask], 1u);
    }
    __syncthreads();
    // the last block of the segment to arrive picks the digit.  No fences (a device-scope release is a whole-L2 write-back here): the
    // bin updates are device-scope atomics, i.e. performed at the memory side; each thread waits for the RETURN of its own updates
    // before the block's arrival is counted, and the last block reads the bins with device-scope atomic loads.
    uint32_t seen = 0;
    for (int b = t; b < nb; b += 256) if (h[b]) seen += atomicAdd(&hist[b], h[b]);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen) :: "memory");
    __syncthreads();
    if (t == 0) last_s = atomicAdd(&st->done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last_s) return;
    // digit d = the largest one whose inclusive suffix count sum_{e >= d} hist[e] reaches k_remaining (d = 0 if none).  Thread t owns the
    // PER = nb / 256 consecutive bins [t PER, (t + 1) PER): S[t] = suffix count from its first bin on (parallel suffix scan over the 256
    // group sums), then the owner of the hit walks its own bins from the top.
    const int PER = nb >> 8;
    uint32_t mine[TOPK_BINS / 256];
    int64_t gsum = 0;
#pragma unroll
    for (int e = 0; e < TOPK_BINS / 256; ++e) {
        mine[e] = e < PER ? __hip_atomic_load(&hist[t * PER + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        gsum += mine[e];
    }
    S[t] = gsum;
    if (t == 0) { S[256] = 0; pick_s = -1; }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int64_t add = t + off < 256 ? S[t + off] : 0;
        __syncthreads();
        S[t] += add;
        __syncthreads();
    }
    const int64_t need = st->k_remaining;
    __syncthreads();                                   // every thread has read k_remaining before it is rewritten
    // the group that holds the digit: S[t] >= need > S[t + 1]; if no digit reaches need (need > total: cannot happen for k <= n; or
    // need == 0: every suffix count reaches it) the conventions of the 8-bit form are kept: need == 0 -> the top digit, none -> digit 0
    const bool group_hit = t == 0 ? (S[1] < need) : (S[t] >= need && (S[t + 1] < need || t == 255));
    if (group_hit) {
        int64_t above = S[t + 1];                      // keys with a larger digit
        int d = t * PER;                               // (t = 0 and nothing reaches need: digit 0)
        uint32_t cnt = mine[0];
        bool found = false;
#pragma unroll
        for (int e = TOPK_BINS / 256 - 1; e >= 0; --e) {
            if (e < PER && !found) {
                if (above + mine[e] >= need || (t == 0 && e == 0)) { d = t * PER + e; cnt = mine[e]; found = true; }
                else above += mine[e];
            }
        }
        const int64_t rem = need - above;              // how many to take among keys sharing the new prefix
        st->prefix = prefix | ((uint32_t)d << shift);
        st->k_remaining = rem;
        st->done = 0;
        if (pass == TOPK_PASSES - 1) {
            st->count_eq = cnt;
            st->tie = rem < (int64_t)cnt ? 1u : 0u;
            if (rem < (int64_t)cnt) atomicOr(any_tie, 1u);     // some segment needs the ranking launches below
        }
    }
    for (int b = t; b < nb; b += 256) hist[b] = 0;     // ready for the next pass
}
// mask = key > T, or key == T when every such row is kept; on a genuine tie the equal rows are flagged for the ranking below
// (eq is written for every row: the ranking scans the flags of ALL segments in one pass)
__global__ void k_topk_mask(const float* __restrict__ v, int ld, TopkSegs segs, const TopkState* st_all, uint8_t* mask, uint8_t* eq) {
    const int y = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = segs.off[y + 1] - segs.off[y];
    if (i >= n) return;
    const int64_t g = segs.off[y] + i;
    const TopkState* st = st_all + y;
    const uint32_t key = order_key(v[g * ld]), T = st->prefix;
    const bool tie = st->tie != 0;
    mask[g] = (key > T) || (key == T && !tie);
    eq[g] = tie && key == T;
}
// tie_high = 0: among logits equal to the threshold the LOWER row indices are kept (canonical); 1: the HIGHER ones.
// eq_rank = exclusive count of flagged rows over the whole batch: the rank inside the segment subtracts the count at its first row
__global__ void k_topk_tie_fix(const TopkState* st_all, TopkSegs segs, const uint8_t* __restrict__ eq, const int32_t* __restrict__ eq_rank,
                               int tie_high, uint8_t* mask) {
    const int y = blockIdx.y;
    const TopkState* st = st_all + y;
    if (st->tie == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = segs.off[y + 1] - segs.off[y];
    if (i >= n) return;
    const int64_t g = segs.off[y] + i;
    if (!eq[g]) return;
    const int64_t need = st->k_remaining, r = (int64_t)eq_rank[g] - (int64_t)eq_rank[segs.off[y]];
    mask[g] = tie_high ? r >= (int64_t)st->count_eq - need : r < need;
}
static int g_topk_tie_high = 0;
// result-changing ‡ conventions (see include/pcgc_hip.h); 0 = top-k tie rule
extern "C" int pcgc_set_convention(int what, int value) {
    if (what == 0) { g_topk_tie_high = value ? 1 : 0; return 0; }
    pcgc_set_error("set_convention: unknown convention %d", what);
    return -2;
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// workspace: states (16 x 32 B) | any-tie flag | histograms (16 x 2048 x 4 B) | eq flags | ranks | total | scan workspace
extern "C" size_t pcgc_topk_workspace_bytes(int64_t n) {
    return 1024 + (size_t)TOPK_MAX_SEGS * TOPK_BINS * 4 + align256((size_t)n) + align256((size_t)n * 4) + 256 + align256(pcgc_scan_workspace_bytes(n));
}
static int topk_segments(const float* logits, int ld, const TopkSegs& segs, uint8_t* mask, void* workspace, size_t workspace_bytes, void* stream) {
    const int64_t n = segs.off[segs.n];
    PCGC_REQUIRE(workspace_bytes >= pcgc_topk_workspace_bytes(n), "workspace too small");
    PCGC_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (n == 0) return 0;
    char* ws = (char*)workspace;
    TopkState* st = (TopkState*)ws;
    uint32_t* any_tie = (uint32_t*)(ws + 768); ws += 1024;
    uint32_t* hist = (uint32_t*)ws; ws += (size_t)TOPK_MAX_SEGS * TOPK_BINS * 4;
    uint8_t* eq = (uint8_t*)ws; ws += align256((size_t)n);
    int32_t* rank = (int32_t*)ws; ws += align256((size_t)n * 4);
    int32_t* total = (int32_t*)ws; ws += 256;
    void* scan_ws = ws;
    int64_t nmax = 0;
    for (int b = 0; b < segs.n; ++b) nmax = std::max<int64_t>(nmax, segs.off[b + 1] - segs.off[b]);
    unsigned g = grid_for(nmax, 256 * 8); if (g > 256) g = 256; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_topk_init, dim3(segs.n), dim3(256), 0, S(stream), st, hist, segs, (unsigned long long*)scan_ws,
                       (int64_t)(pcgc_scan_workspace_bytes(n) / 8), any_tie);
    for (int pass = 0; pass < TOPK_PASSES; ++pass)
        hipLaunchKernelGGL(k_topk_hist, dim3(g, segs.n), dim3(256), 0, S(stream), logits, ld, segs, st, pass, hist, any_tie);
    hipLaunchKernelGGL(k_topk_mask, dim3(grid_for(nmax, 256), segs.n), dim3(256), 0, S(stream), logits, ld, segs, st, mask, eq);
    // genuine tie at a segment's threshold only (any_tie): rank the equal rows and keep `need` of them per segment
    launch_scan(eq, n, rank, total, scan_ws, any_tie, S(stream));
    hipLaunchKernelGGL(k_topk_tie_fix, dim3(grid_for(nmax, 256), segs.n), dim3(256), 0, S(stream), st, segs, eq, rank, g_topk_tie_high, mask);
    PCGC_CHECK_LAUNCH("topk_mask");
    return 0;
}
extern "C" int pcgc_topk_mask(const float* logits, int ld, int64_t n, int64_t k, uint8_t* mask, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (n == 0) return 0;
    if (k >= n) { (void)hipMemsetAsync(mask, 1, (size_t)n, S(stream)); return 0; }
    if (k <= 0) { (void)hipMemsetAsync(mask, 0, (size_t)n, S(stream)); return 0; }
    TopkSegs segs{};
    segs.n = 1; segs.off[0] = 0; segs.off[1] = n; segs.k[0] = k;
    return topk_segments(logits, ld, segs, mask, workspace, workspace_bytes, stream);
}

// Batched form: ONE launch sequence for all items (blockIdx.y = item).  seg_rows / seg_k are HOST arrays; the workspace is sized for
// the TOTAL row count (pcgc_topk_workspace_bytes(sum of seg_rows)).  k is clamped to [0, rows] per item.
extern "C" int pcgc_topk_mask_segments(const float* logits, int ld, int nseg, const int64_t* seg_rows, const int64_t* seg_k,
                                       uint8_t* mask, void* workspace, size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(nseg >= 0 && nseg <= TOPK_MAX_SEGS && (nseg == 0 || (seg_rows && seg_k)), "at most 16 segments");
    if (nseg == 0) return 0;
    TopkSegs segs{};
    segs.n = nseg; segs.off[0] = 0;
    for (int b = 0; b < nseg; ++b) {
        PCGC_REQUIRE(seg_rows[b] >= 0, "negative segment");
        segs.off[b + 1] = segs.off[b] + seg_rows[b];
        segs.k[b] = seg_k[b] < 0 ? 0 : (seg_k[b] > seg_rows[b] ? seg_rows[b] : seg_k[b]);
    }
    return topk_segments(logits, ld, segs, mask, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------- top-k + pruning in one sweep (round 4)
// prune_voxel (autoencoder.py:239-249) = istopk (data_utils.py:77-89) + MinkowskiPruning (autoencoder.py:237,247).  After the radix passes
// have fixed every segment's threshold T, ONE single-pass scan (decoupled look-back, two counters per tile: keys above T, keys equal to T)
// decides each row, ranks the equal keys of a genuine tie on the way (canonical rule: the lower rows; `tie_high`: the higher rows), and
// writes everything the pruned level needs: its coordinates, the candidate row of every surviving row (`orig`) and a RANK BITMAP of the
// candidate level — one bit per row + the exclusive count of survivors at every 64th row — instead of a byte mask and an int32 prefix
// per row (10 bytes per candidate row, 20 MB at 2 M candidates; the bitmap is 0.19 bytes per row and stays in L2 for the 27 random rank
// queries per surviving row of the kernel-map derivation).  Replaces k_topk_mask + (no-op) tie scan + k_topk_tie_fix + memset +
// k_scan_lookback + k_compact_coords + k_compact_index: seven launches, four sweeps over per-row arrays.
// The exclusive survivor count at row g of segment b is  K0[b] + (#above before g in b) + (kept equal keys before g in b), and the counts
// "before g in b" are global running counts minus their value at the segment's first row — G0[b] = sum_{b' < b} (k[b'] - need[b']),
// E0[b] = sum_{b' < b} count_eq[b'] — both known from the segments' final states, so one global scan serves all segments.
struct SelSeg { uint32_t T; int64_t need, thr, K0, G0, E0; };   // thr = count_eq - need (tie_high: the equal keys from rank thr on are kept)
__device__ static inline int4 child_coords(const int4* __restrict__ parent, int64_t g, int32_t h) {
    const int4 c = parent[g >> 3]; const int k = (int)(g & 7);
    return make_int4(c.x, c.y + (k & 1) * h, c.z + ((k >> 1) & 1) * h, c.w + (k >> 2) * h);
}
__global__ void __launch_bounds__(256) k_topk_select(const float* __restrict__ v, int ld, TopkSegs segs, const TopkState* __restrict__ st_all,
                                                     const int4* __restrict__ coords, const int4* __restrict__ parent, int32_t half,
                                                     int tie_high, int64_t n, int64_t n_tiles, unsigned long long* desc, int32_t* ticket,
                                                     uint8_t* __restrict__ bits, int32_t* __restrict__ wprefix,
                                                     int32_t* __restrict__ orig, int4* __restrict__ out_coords) {
    __shared__ SelSeg seg[TOPK_MAX_SEGS];
    __shared__ int tile_s, exg_s, exe_s;
    const int t = threadIdx.x;
    if (t == 0) {
        tile_s = atomicAdd(ticket, 1);
        int64_t K0 = 0, G0 = 0, E0 = 0;
        for (int b = 0; b < segs.n; ++b) {
            const TopkState s = st_all[b];
            const int64_t need = s.k_remaining < (int64_t)s.count_eq ? s.k_remaining : (int64_t)s.count_eq;   // equal keys kept
            seg[b].T = s.prefix; seg[b].need = need; seg[b].thr = (int64_t)s.count_eq - need;
            seg[b].K0 = K0; seg[b].G0 = G0; seg[b].E0 = E0;
            K0 += segs.k[b]; G0 += segs.k[b] - need; E0 += s.count_eq;
        }
    }
    __syncthreads();
    const int64_t tile = tile_s;
    const int64_t base = tile * SCAN_TILE + t * 8;
    // the thread's eight consecutive rows: keys, segment of the first row (rows are consecutive: the segment only ever advances)
    uint32_t key[8];
    if (ld == 1 && base + 8 <= n && ((uintptr_t)(v + base) & 15) == 0) {
        const float4 a = *(const float4*)(v + base), b4 = *(const float4*)(v + base + 4);
        key[0] = order_key(a.x); key[1] = order_key(a.y); key[2] = order_key(a.z); key[3] = order_key(a.w);
        key[4] = order_key(b4.x); key[5] = order_key(b4.y); key[6] = order_key(b4.z); key[7] = order_key(b4.w);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) key[j] = base + j < n ? order_key(v[(base + j) * ld]) : 0u;
    }
    int b0 = 0;
    while (b0 + 1 < segs.n && base >= segs.off[b0 + 1]) ++b0;
    uint32_t gt = 0, eq = 0;                            // bit j = row base + j
    {
        int b = b0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t g = base + j;
            while (b + 1 < segs.n && g >= segs.off[b + 1]) ++b;
            if (g < n) {
                const uint32_t T = seg[b].T;
                gt |= (uint32_t)(key[j] > T) << j;
                eq |= (uint32_t)(key[j] == T) << j;
            }
        }
    }
    // block scan of both counts at once: low half = rows above their threshold, high half = rows equal to it (<= 2048 each)
    int tot;
    const int packed = __popc(gt) | (__popc(eq) << 16);
    const int ex = block_exclusive_scan_256(packed, &tot);
    if (t < 64) {
        const int lane = t;
        int exg = 0, exe = 0;
        const unsigned long long own = ((unsigned long long)(uint32_t)(tot & 0xFFFF) << 31) | (unsigned long long)(uint32_t)(tot >> 16);
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&desc[0], SCAN_INCL | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], SCAN_AGG | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t j = tile - 1;; j -= 64) {
                const int64_t idx = j - lane;
                unsigned long long d = SCAN_INCL;                        // before tile 0: an inclusive prefix of 0
                if (idx >= 0) {
                    do { d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((d >> 62) == 0);
                }
                const unsigned long long incl = __ballot((d >> 62) == 2);
                const int first = incl ? (int)__ffsll((long long)incl) - 1 : 63;     // nearest inclusive prefix, or the whole window
                int vg = lane <= first ? (int)((d >> 31) & 0x7FFFFFFFull) : 0, ve = lane <= first ? (int)(d & 0x7FFFFFFFull) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { vg += __shfl_xor(vg, o, 64); ve += __shfl_xor(ve, o, 64); }
                exg += vg; exe += ve;
                if (incl) break;
            }
            if (lane == 0) {
                const unsigned long long inc = ((unsigned long long)(uint32_t)(exg + (tot & 0xFFFF)) << 31) | (unsigned long long)(uint32_t)(exe + (tot >> 16));
                __hip_atomic_store(&desc[tile], SCAN_INCL | inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) { exg_s = exg; exe_s = exe; }
    }
    __syncthreads();
    int64_t G = (int64_t)exg_s + (ex & 0xFFFF), E = (int64_t)exe_s + (ex >> 16);      // global running counts at the thread's first row
    uint32_t keep = 0;
    int b = b0;
    int64_t first_out = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t g = base + j;
        while (b + 1 < segs.n && g >= segs.off[b + 1]) ++b;
        const SelSeg sg = seg[b];
        const int64_t r = E - sg.E0;                                          // equal keys of this segment before row g
        const int64_t eq_kept = tie_high ? (r > sg.thr ? r - sg.thr : 0) : (r < sg.need ? r : sg.need);
        const int64_t out = sg.K0 + (G - sg.G0) + eq_kept;                    // survivors before row g (all segments)
        if (j == 0) first_out = out;
        const bool is_gt = (gt >> j) & 1u, is_eq = (eq >> j) & 1u;
        const bool kept = is_gt || (is_eq && (tie_high ? r >= sg.thr : r < sg.need));
        if (kept) {                                                           // (g < n: rows past the end carry no flag)
            keep |= 1u << j;
            orig[out] = (int32_t)g;
            out_coords[out] = coords ? coords[g] : child_coords(parent, g, half);
        }
        G += is_gt; E += is_eq;
    }
    const int64_t n64 = (n + 63) & ~(int64_t)63;
    if (base < n64) {
        bits[base >> 3] = (uint8_t)keep;
        if ((t & 7) == 0) wprefix[base >> 6] = (int32_t)first_out;
    }
}
// Rank bitmap sizes for n candidate rows: bits = ((n + 63) / 64) * 8 bytes (8-byte aligned: the consumers read 64-bit words), wprefix =
// (n + 63) / 64 int32.  Workspace: pcgc_topk_workspace_bytes(n) (states, histograms, scan descriptors).
// workspace of pcgc_topk_select: states (16 x 32 B) | any-tie flag | histograms (16 x 2048 x 4 B) | scan descriptors + ticket
extern "C" size_t pcgc_topk_select_workspace_bytes(int64_t n) {
    return 1024 + (size_t)TOPK_MAX_SEGS * TOPK_BINS * 4 + align256(pcgc_scan_workspace_bytes(n));
}
extern "C" int pcgc_topk_select(const float* logits, int ld, int nseg, const int64_t* seg_rows, const int64_t* seg_k,
                                const int32_t* coords, const int32_t* parent_coords, int32_t parent_stride,
                                uint8_t* bits, int32_t* wprefix, int32_t* orig, int32_t* out_coords,
                                void* workspace, size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(nseg >= 1 && nseg <= TOPK_MAX_SEGS && seg_rows && seg_k, "1 to 16 segments");
    PCGC_REQUIRE((coords != nullptr) != (parent_coords != nullptr), "exactly one of coords / parent_coords");
    PCGC_REQUIRE(!parent_coords || (parent_stride >= 2 && (parent_stride & 1) == 0), "parent stride must be even");
    PCGC_REQUIRE(((uintptr_t)bits & 7) == 0 && ((uintptr_t)workspace & 7) == 0, "bits and workspace must be 8-byte aligned");
    TopkSegs segs{};
    segs.n = nseg; segs.off[0] = 0;
    for (int b = 0; b < nseg; ++b) {
        PCGC_REQUIRE(seg_rows[b] >= 0, "negative segment");
        segs.off[b + 1] = segs.off[b] + seg_rows[b];
        segs.k[b] = seg_k[b] < 0 ? 0 : (seg_k[b] > seg_rows[b] ? seg_rows[b] : seg_k[b]);
    }
    const int64_t n = segs.off[nseg];
    PCGC_REQUIRE(n < (1ll << 31), "too many rows");
    PCGC_REQUIRE(!parent_coords || (n & 7) == 0, "a children level has 8 rows per parent");
    PCGC_REQUIRE(workspace_bytes >= pcgc_topk_select_workspace_bytes(n), "workspace too small");
    if (n == 0) return 0;
    char* ws = (char*)workspace;
    TopkState* st = (TopkState*)ws;
    uint32_t* any_tie = (uint32_t*)(ws + 768); ws += 1024;
    uint32_t* hist = (uint32_t*)ws; ws += (size_t)TOPK_MAX_SEGS * TOPK_BINS * 4;
    unsigned long long* desc = (unsigned long long*)ws;
    const int64_t tiles = scan_tiles(n);
    int64_t nmax = 0;
    for (int b = 0; b < nseg; ++b) nmax = std::max<int64_t>(nmax, segs.off[b + 1] - segs.off[b]);
    unsigned g = grid_for(nmax, 256 * 8); if (g > 256) g = 256; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_topk_init, dim3(nseg), dim3(256), 0, S(stream), st, hist, segs, desc, (int64_t)(pcgc_scan_workspace_bytes(n) / 8), any_tie);
    for (int pass = 0; pass < TOPK_PASSES; ++pass)
        hipLaunchKernelGGL(k_topk_hist, dim3(g, nseg), dim3(256), 0, S(stream), logits, ld, segs, st, pass, hist, any_tie);
    hipLaunchKernelGGL(k_topk_select, dim3((unsigned)tiles), dim3(256), 0, S(stream), logits, ld, segs, st, (const int4*)coords,
                       (const int4*)parent_coords, parent_stride / 2, g_topk_tie_high, n, tiles, desc, (int32_t*)(desc + tiles),
                       bits, wprefix, orig, (int4*)out_coords);
    PCGC_CHECK_LAUNCH("topk_select");
    return 0;
}
// out[r] = in[orig[r]] for rows of C floats (C % 4 == 0): the surviving feature rows of a pruned level, through `orig`
__global__ void k_gather_rows_f32x4(const float* __restrict__ in, int C4, int in_ld, const int32_t* __restrict__ orig, int64_t n_out,
                                    float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t / C4; const int c = (int)(t % C4);
    if (r < n_out) *(float4*)(out + r * (C4 * 4) + 4 * c) = *(const float4*)(in + (int64_t)orig[r] * in_ld + 4 * c);
}
extern "C" int pcgc_gather_rows_f32_ld(const float* in, int C, int in_ld, const int32_t* orig, int64_t n_out, float* out, void* stream) {
    PCGC_REQUIRE(C > 0 && (C & 3) == 0 && (in_ld & 3) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "rows of 4-float chunks, 16-byte aligned");
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(k_gather_rows_f32x4, dim3(grid_for(n_out * (C / 4), 256)), dim3(256), 0, S(stream), in, C / 4, in_ld, orig, n_out, out);
    PCGC_CHECK_LAUNCH("gather_rows_f32_ld");
    return 0;
}

// rows per batch item (column 0 of the coordinates): counts[b] for b < 16 (the coordinate key holds 4 batch bits)
__global__ void k_batch_counts(const int4* __restrict__ coords, int64_t n, int32_t* __restrict__ counts) {
    __shared__ int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = coords[i].x;
        if ((unsigned)b < 16u) atomicAdd(&h[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
extern "C" int pcgc_batch_counts(const int32_t* coords, int64_t n, int32_t* counts, void* stream) {
    PCGC_REQUIRE(counts != nullptr, "null argument");
    hipError_t e = hipMemsetAsync(counts, 0, 16 * sizeof(int32_t), S(stream));
    if (e != hipSuccess) { pcgc_set_error("batch_counts: %s", hipGetErrorString(e)); return -1; }
    if (n == 0) return 0;
    unsigned g = grid_for(n, 256 * 16); if (g > 512) g = 512; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batch_counts, dim3(g), dim3(256), 0, S(stream), (const int4*)coords, n, counts);
    PCGC_CHECK_LAUNCH("batch_counts");
    return 0;
}

// ------------------------------------------------------------------------------------------- z-major sort
// array2vector(C, C.max()+1) orders rows by (z, y, x, batch), z most significant (data_utils.py:55-61); every field
// is < step, so sorting the packed 64-bit key (z<<44 | y<<24 | x<<4 | batch) gives the same permutation.
// batch_major: (batch, z, y, x) instead — the items of a collated batch stay contiguous, each in its own (z, y, x) order: the order
// sort_spare_tensor gives every item when the clouds are coded one by one
__global__ void k_zyx_keys(const int4* __restrict__ coords, int64_t n, uint64_t* keys, int32_t* idx, int batch_major) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    keys[i] = batch_major ? (((uint64_t)(uint32_t)c.x << 60) | ((uint64_t)(uint32_t)c.w << 40) | ((uint64_t)(uint32_t)c.z << 20) | (uint64_t)(uint32_t)c.y)
                          : (((uint64_t)(uint32_t)c.w << 44) | ((uint64_t)(uint32_t)c.z << 24) | ((uint64_t)(uint32_t)c.y << 4) | (uint64_t)(uint32_t)c.x);
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
static int sort_coords(const int32_t* coords, int64_t n, int32_t* perm, void* workspace, size_t workspace_bytes, void* stream, int batch_major);
extern "C" int pcgc_sort_zyx(const int32_t* coords, int64_t n, int32_t* perm, void* workspace, size_t workspace_bytes,
                             void* stream) {
    return sort_coords(coords, n, perm, workspace, workspace_bytes, stream, 0);
}
extern "C" int pcgc_sort_bzyx(const int32_t* coords, int64_t n, int32_t* perm, void* workspace, size_t workspace_bytes,
                              void* stream) {
    return sort_coords(coords, n, perm, workspace, workspace_bytes, stream, 1);
}
static int sort_coords(const int32_t* coords, int64_t n, int32_t* perm, void* workspace, size_t workspace_bytes, void* stream, int batch_major) {
    PCGC_REQUIRE(workspace_bytes >= pcgc_sort_workspace_bytes(n), "workspace too small");
    if (n == 0) return 0;
    char* ws = (char*)workspace;
    uint64_t* kin = (uint64_t*)ws; ws += align256((size_t)n * 8);
    uint64_t* kout = (uint64_t*)ws; ws += align256((size_t)n * 8);
    int32_t* idx = (int32_t*)ws; ws += align256((size_t)n * 4);
    size_t tmp = sort_temp_bytes(n);
    hipLaunchKernelGGL(k_zyx_keys, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n, kin, idx, batch_major);
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
