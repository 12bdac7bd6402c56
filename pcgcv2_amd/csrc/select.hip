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
// MSB-first radix select on the order-preserving integer image of the fp32 logits: 4 passes of 8 bits, each a
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
// one block per segment: zeroes its state and histogram; block 0 also the tie path's scan workspace and the any-tie flag
__global__ void k_topk_init(TopkState* st, uint32_t* hist, TopkSegs segs, unsigned long long* scan_ws, int64_t scan_words, uint32_t* any_tie) {
    const int y = blockIdx.x;
    if (threadIdx.x == 0) { st[y].prefix = 0; st[y].tie = 0; st[y].k_remaining = segs.k[y]; st[y].done = 0; st[y].count_eq = 0; }
    hist[256 * y + threadIdx.x] = 0;
    if (y == 0) {
        if (threadIdx.x == 0) *any_tie = 0;
        for (int64_t i = threadIdx.x; i < scan_words; i += blockDim.x) scan_ws[i] = 0;
    }
}
// Block-local LDS histogram, flushed with one global atomic per non-empty bin.  The grid is kept SMALL (<= 256 blocks per segment):
// the flush is up to 256 same-address atomics per block, and with 2048 blocks those serialised in L2 for ~30 us per pass
// on the 2 M-candidate level (the element loop itself is ~3 us).  blockIdx.y = segment.
__global__ void __launch_bounds__(256) k_topk_hist(const float* __restrict__ v_all, int ld, TopkSegs segs, TopkState* st_all,
                                                   int pass, uint32_t* hist_all, uint32_t* any_tie) {
    __shared__ uint32_t h[256];
    __shared__ int64_t S[257];
    __shared__ bool last_s;
    const int t = threadIdx.x, y = blockIdx.y;
    const float* __restrict__ v = v_all + segs.off[y] * ld;
    const int64_t n = segs.off[y + 1] - segs.off[y];
    TopkState* st = st_all + y;
    uint32_t* hist = hist_all + 256 * y;
    h[t] = 0;
    __syncthreads();
    int shift = 24 - 8 * pass;
    uint32_t prefix = st->prefix;
    uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + t;
    for (; i + 3 * step < n; i += 4 * step) {          // four independent loads in flight per thread (one block per CU: nothing else hides them)
        const float f0 = v[i * ld], f1 = v[(i + step) * ld], f2 = v[(i + 2 * step) * ld], f3 = v[(i + 3 * step) * ld];
        const uint32_t k0 = order_key(f0), k1 = order_key(f1), k2 = order_key(f2), k3 = order_key(f3);
        if ((k0 & pmask) == prefix) atomicAdd(&h[(k0 >> shift) & 0xff], 1u);
        if ((k1 & pmask) == prefix) atomicAdd(&h[(k1 >> shift) & 0xff], 1u);
        if ((k2 & pmask) == prefix) atomicAdd(&h[(k2 >> shift) & 0xff], 1u);
        if ((k3 & pmask) == prefix) atomicAdd(&h[(k3 >> shift) & 0xff], 1u);
    }
    for (; i < n; i += step) {
        uint32_t key = order_key(v[i * ld]);
        if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    // the last block of the segment to arrive picks the digit.  No fences (a device-scope release is a whole-L2 write-back here): the
    // bin updates are device-scope atomics, i.e. performed at the memory side; each thread waits for the RETURN of its own update
    // before the block's arrival is counted, and the last block reads the bins with device-scope atomic loads.
    uint32_t seen = 0;
    if (h[t]) seen = atomicAdd(&hist[t], h[t]);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen) :: "memory");
    __syncthreads();
    if (t == 0) last_s = atomicAdd(&st->done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last_s) return;
    // digit d = the largest one whose inclusive suffix count S[d] = sum_{e >= d} hist[e] reaches k_remaining (d = 0 if none):
    // parallel suffix scan over the 256 bins
    const int64_t mine = __hip_atomic_load(&hist[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const int64_t rem = need - S[t + 1];           // how many to take among keys sharing the new prefix
        st->prefix = prefix | ((uint32_t)t << shift);
        st->k_remaining = rem;
        st->done = 0;
        if (pass == 3) {
            st->count_eq = (uint32_t)mine;
            st->tie = rem < mine ? 1u : 0u;
            if (rem < mine) atomicOr(any_tie, 1u);     // some segment needs the ranking launches below
        }
    }
    hist[t] = 0;                                       // ready for the next pass
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
// workspace: states (16 x 32 B) | any-tie flag | histograms (16 x 256 x 4 B) | eq flags | ranks | total | scan workspace
extern "C" size_t pcgc_topk_workspace_bytes(int64_t n) {
    return 1024 + (size_t)TOPK_MAX_SEGS * 1024 + align256((size_t)n) + align256((size_t)n * 4) + 256 + align256(pcgc_scan_workspace_bytes(n));
}
static int topk_segments(const float* logits, int ld, const TopkSegs& segs, uint8_t* mask, void* workspace, size_t workspace_bytes, void* stream) {
    const int64_t n = segs.off[segs.n];
    PCGC_REQUIRE(workspace_bytes >= pcgc_topk_workspace_bytes(n), "workspace too small");
    PCGC_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (n == 0) return 0;
    char* ws = (char*)workspace;
    TopkState* st = (TopkState*)ws;
    uint32_t* any_tie = (uint32_t*)(ws + 768); ws += 1024;
    uint32_t* hist = (uint32_t*)ws; ws += (size_t)TOPK_MAX_SEGS * 1024;
    uint8_t* eq = (uint8_t*)ws; ws += align256((size_t)n);
    int32_t* rank = (int32_t*)ws; ws += align256((size_t)n * 4);
    int32_t* total = (int32_t*)ws; ws += 256;
    void* scan_ws = ws;
    int64_t nmax = 0;
    for (int b = 0; b < segs.n; ++b) nmax = std::max<int64_t>(nmax, segs.off[b + 1] - segs.off[b]);
    unsigned g = grid_for(nmax, 256 * 8); if (g > 256) g = 256; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_topk_init, dim3(segs.n), dim3(256), 0, S(stream), st, hist, segs, (unsigned long long*)scan_ws,
                       (int64_t)(pcgc_scan_workspace_bytes(n) / 8), any_tie);
    for (int pass = 0; pass < 4; ++pass)
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
