// Coordinate hash map, coordinate transforms and kernel-map construction.
// Replaces MinkowskiEngine's CoordinateManager (reference call sites: data_utils.py:96,108,116; coder.py:102;
// every MinkowskiConvolution in autoencoder.py builds/looks up a kernel map through it).
#include <stdarg.h>
#include "pcgc_common.h"

static thread_local char g_err[512] = "";
void pcgc_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* pcgc_last_error(void) { return g_err; }
extern "C" int pcgc_version(void) { return 1; }

extern "C" int64_t pcgc_hash_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

__global__ void k_hash_clear(uint64_t* keys, int32_t* vals, int64_t cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = PCGC_EMPTY_KEY; vals[i] = 0x7fffffff; }
}

__global__ void k_hash_insert(const int4* __restrict__ coords, int64_t n, uint64_t* keys, int32_t* vals, uint64_t cap_mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int4 c = make_int4(-1, -1, -1, -1);               // (b, x, y, z)
    if (i < n) c = coords[i];
    // rows outside the key range are never found again; the host rejects such input up front (pcgc_coords_check_order)
    const bool ok = i < n && coord_in_range(c.x, c.y, c.z, c.w);
    const uint64_t key = ok ? coord_key(c.x, c.y, c.z, c.w) : PCGC_EMPTY_KEY;
    // a run of equal keys in consecutive lanes (quantised x-neighbours of a raster-ordered cloud) is inserted by its first lane only:
    // that lane has the smallest row of the run.  (Shuffle before any lane leaves: every lane holds a defined key.)
    const uint64_t left = __shfl_up((unsigned long long)key, 1, 64);
    if (!ok || ((threadIdx.x & 63) != 0 && left == key)) return;
    uint64_t h = hash_slot(key, cap_mask);
    // Quantised coordinates arrive up to 8 times each.  Test before the atomics: a slot only ever goes EMPTY -> key and its
    // row value only ever decreases, so a plain (possibly stale) load that already shows this key / a smaller row lets the
    // thread skip the CAS / the atomicMin; a stale load merely falls through to the atomic path.
    for (;;) {
        unsigned long long prev = __builtin_nontemporal_load((const unsigned long long*)&keys[h]);
        if (prev == PCGC_EMPTY_KEY)                     // (a slot holding another key keeps it for good: just probe on)
            prev = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)PCGC_EMPTY_KEY, (unsigned long long)key);
        if (prev == PCGC_EMPTY_KEY || prev == key) {
            if (__builtin_nontemporal_load(&vals[h]) > (int32_t)i) atomicMin(&vals[h], (int32_t)i);
            return;
        }
        h = (h + 1) & cap_mask;
    }
}
// dedup policy "keep the LAST occurrence" (the default everywhere is the first): same table, vals[slot] = LARGEST row with the key
__global__ void k_hash_insert_last(const int4* __restrict__ coords, int64_t n, uint64_t* keys, int32_t* vals, uint64_t cap_mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    if (!coord_in_range(c.x, c.y, c.z, c.w)) return;
    uint64_t key = coord_key(c.x, c.y, c.z, c.w);
    uint64_t h = hash_slot(key, cap_mask);
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)PCGC_EMPTY_KEY, (unsigned long long)key);
        if (prev == PCGC_EMPTY_KEY || prev == key) {
            // slots are cleared to INT_MAX (the minimum-keeping insert's neutral element): it counts as "no row" when maximising
            int32_t old = __builtin_nontemporal_load(&vals[h]);
            while (old == 0x7fffffff || old < (int32_t)i) {
                int32_t seen = atomicCAS(&vals[h], old, (int32_t)i);
                if (seen == old) break;
                old = seen;
            }
            return;
        }
        h = (h + 1) & cap_mask;
    }
}
__global__ void k_hash_first_mask(const int4* __restrict__ coords, int64_t n, const uint64_t* __restrict__ keys,
                                  const int32_t* __restrict__ vals, uint64_t cap_mask, uint8_t* keep, int32_t* first_row) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = coords[i];
    int32_t f = hash_lookup(keys, vals, cap_mask, c.x, c.y, c.z, c.w);
    keep[i] = f == (int32_t)i;
    if (first_row) first_row[i] = f;
}

// rows whose coordinates the 4+20+20+20-bit key cannot hold (negative, >= 2^20, batch >= 16): the hash kernels skip such rows, so the host
// validates every externally supplied coordinate tensor with this pass before building a level on it; plus the number of DESCENTS of the
// (batch, z, y, x) key along the rows (0 = the rows are in sort_spare_tensor's order,
// data_utils.py:91-101; about n / 2 = no order at all): the encoder sorts an unordered cloud once at ingest (coder.Coder._ingest) instead
// of dragging every gather of every encoder level through a random row order
__global__ void k_coords_check_order(const int4* __restrict__ coords, int64_t n, int32_t* __restrict__ out2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool oob = false, desc = false;
    if (i < n) {
        const int4 c = coords[i];
        oob = !coord_in_range(c.x, c.y, c.z, c.w);
        if (i > 0) {
            const int4 p = coords[i - 1];
            desc = coord_key(c.x & 15, c.y & 0xFFFFF, c.z & 0xFFFFF, c.w & 0xFFFFF) < coord_key(p.x & 15, p.y & 0xFFFFF, p.z & 0xFFFFF, p.w & 0xFFFFF);
        }
    }
    const unsigned long long b = __ballot(oob), d = __ballot(desc);
    if ((threadIdx.x & 63) == 0) {
        if (b) atomicAdd(out2, (int32_t)__popcll(b));
        if (d) atomicAdd(out2 + 1, (int32_t)__popcll(d));
    }
}
extern "C" int pcgc_coords_check_order(const int32_t* coords, int64_t n, int32_t* out2, void* stream) {
    PCGC_REQUIRE(out2 != nullptr, "null counters");
    hipMemsetAsync(out2, 0, 2 * sizeof(int32_t), S(stream));
    if (n > 0) hipLaunchKernelGGL(k_coords_check_order, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n, out2);
    PCGC_CHECK_LAUNCH("coords_check_order");
    return 0;
}
extern "C" int pcgc_hash_clear(uint64_t* keys, int32_t* vals, int64_t cap, void* stream) {
    PCGC_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "capacity must be a power of two");
    hipLaunchKernelGGL(k_hash_clear, dim3(grid_for(cap, 256)), dim3(256), 0, S(stream), keys, vals, cap);
    PCGC_CHECK_LAUNCH("hash_clear");
    return 0;
}
extern "C" int pcgc_hash_insert(const int32_t* coords, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals, int64_t cap,
                                void* stream) {
    (void)stride;                                     // (the slot no longer depends on the level's lattice: see pcgc_common.h)
    PCGC_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "capacity must be a power of two >= 2n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_hash_insert, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n,
                       keys, vals, (uint64_t)(cap - 1));
    PCGC_CHECK_LAUNCH("hash_insert");
    return 0;
}
extern "C" int pcgc_hash_insert_policy(const int32_t* coords, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals, int64_t cap,
                                       int keep_last, void* stream) {
    if (!keep_last) return pcgc_hash_insert(coords, n, stride, keys, vals, cap, stream);
    PCGC_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "capacity must be a power of two >= 2n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_hash_insert_last, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n,
                       keys, vals, (uint64_t)(cap - 1));
    PCGC_CHECK_LAUNCH("hash_insert_policy");
    return 0;
}
extern "C" int pcgc_hash_first_mask(const int32_t* coords, int64_t n, int32_t stride, const uint64_t* keys,
                                    const int32_t* vals, int64_t cap, uint8_t* keep, int32_t* first_row, void* stream) {
    (void)stride;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_hash_first_mask, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n,
                       keys, vals, (uint64_t)(cap - 1), keep, first_row);
    PCGC_CHECK_LAUNCH("hash_first_mask");
    return 0;
}

// ---- coordinate transforms ------------------------------------------------------------------------------------
__global__ void k_coords_quantize(const int4* __restrict__ in, int64_t n, int32_t s, int4* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = in[i];
    auto fl = [s](int32_t v) { return (v >= 0 ? v / s : -((-v + s - 1) / s)) * s; };
    out[i] = make_int4(c.x, fl(c.y), fl(c.z), fl(c.w));
}
__global__ void k_coords_children(const int4* __restrict__ in, int64_t n, int32_t h, int4* out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per child row
    if (t >= 8 * n) return;
    int4 c = in[t >> 3]; int k = (int)(t & 7);
    out[t] = make_int4(c.x, c.y + (k & 1) * h, c.z + ((k >> 1) & 1) * h, c.w + (k >> 2) * h);
}
__global__ void k_coords_scale(const int4* __restrict__ in, int64_t n, float f, int4* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = in[i];
    out[i] = make_int4(c.x, (int32_t)rintf((float)c.y * f), (int32_t)rintf((float)c.z * f), (int32_t)rintf((float)c.w * f));
}
extern "C" int pcgc_coords_quantize(const int32_t* coords, int64_t n, int32_t stride_out, int32_t* out, void* stream) {
    PCGC_REQUIRE(stride_out > 0, "stride must be positive");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_coords_quantize, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n,
                       stride_out, (int4*)out);
    PCGC_CHECK_LAUNCH("coords_quantize");
    return 0;
}
extern "C" int pcgc_coords_children(const int32_t* coords, int64_t n, int32_t stride_in, int32_t* out, void* stream) {
    PCGC_REQUIRE(stride_in >= 2 && (stride_in & 1) == 0, "input stride must be even");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_coords_children, dim3(grid_for(8 * n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n,
                       stride_in / 2, (int4*)out);
    PCGC_CHECK_LAUNCH("coords_children");
    return 0;
}
extern "C" int pcgc_coords_scale(const int32_t* coords, int64_t n, float factor, int32_t* out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_coords_scale, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n, factor,
                       (int4*)out);
    PCGC_CHECK_LAUNCH("coords_scale");
    return 0;
}

// ---- kernel maps ----------------------------------------------------------------------------------------------
// One thread per (offset, site) pair, site fastest: 27x more threads than sites — this kernel only runs on the coarsest
// levels (<= 32k sites), where a thread-per-site version is latency-bound (27 serial probe chains on ~70 workgroups).
// nbr is offset-major [27][n] so the conv kernels read it coalesced.
__global__ void __launch_bounds__(256) k_kmap_k3(const int4* __restrict__ coords, int64_t n, int32_t s,
                                                 const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                 uint64_t cap_mask, int32_t* __restrict__ nbr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 27 * n) return;
    int k = (int)(t / n); int64_t o = t - (int64_t)k * n;
    int4 c = coords[o];
    int dx = (k % 3 - 1) * s, dy = ((k / 3) % 3 - 1) * s, dz = (k / 9 - 1) * s;
    nbr[t] = (k == 13) ? (int32_t)o : hash_lookup(keys, vals, cap_mask, c.x, c.y + dx, c.z + dy, c.w + dz);
}
extern "C" int pcgc_kmap_k3(const int32_t* coords, int64_t n, int32_t stride, const uint64_t* keys, const int32_t* vals,
                            int64_t cap, int32_t* nbr, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_kmap_k3, dim3(grid_for(27 * n, 256)), dim3(256), 0, S(stream), (const int4*)coords, n, stride,
                       keys, vals, (uint64_t)(cap - 1), nbr);
    PCGC_CHECK_LAUNCH("kmap_k3");
    return 0;
}
// ---- hierarchical kernel maps ---------------------------------------------------------------------------------
// A fine voxel at child slot j = (jx,jy,jz) of its parent, displaced by d in {-1,0,1}^3, lands in the parent displaced
// by p = floor((j+d)/2) at child slot (j+d)&1 — so the fine level's 27-neighbourhood is a pure gather through the
// coarse level's kernel map (8x smaller, cache resident).  Only the coarsest level of a pyramid probes the hash.
// (child_offset: pcgc_common.h)

// generative-transpose children (all 8 exist, rows 8*i+j): nbr[k][8i+j] = 8 * pnbr[kp][i] + j'
__global__ void __launch_bounds__(256) k_kmap_children(const int32_t* __restrict__ pnbr, int64_t np, int32_t* __restrict__ nbr) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = 8 * np;
    if (c >= n) return;
    int64_t i = c >> 3; int j = (int)(c & 7);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int kp, jn; child_offset(j, k, kp, jn);
        int32_t pn = pnbr[(int64_t)kp * np + i];
        nbr[(int64_t)k * n + c] = pn < 0 ? -1 : 8 * pn + jn;
    }
}
// pruned level: rows orig[r] of the candidate level survive; neighbours are renumbered through mask/prefix
__global__ void __launch_bounds__(256) k_kmap_prune(const int32_t* __restrict__ cand, int64_t n_cand,
                                                    const uint8_t* __restrict__ mask, const int32_t* __restrict__ prefix,
                                                    const int32_t* __restrict__ orig, int64_t n_out, int32_t* __restrict__ nbr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    int64_t o = orig[r];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int32_t m = cand[(int64_t)k * n_cand + o];
        nbr[(int64_t)k * n_out + r] = (m >= 0 && mask[m]) ? prefix[m] : -1;
    }
}
// pruned level straight from the PARENT level's map: the candidate (children) level's own [27][8 n_p] map is never built.
// candidate row o = 8 i + j; its neighbour through offset k is row 8 pnbr[kp][i] + j' (child_offset), kept iff its mask is set.
__global__ void __launch_bounds__(256) k_kmap_prune_parent(const int32_t* __restrict__ pnbr, int64_t np,
                                                           const uint8_t* __restrict__ mask, const int32_t* __restrict__ prefix,
                                                           const int32_t* __restrict__ orig, int64_t n_out, int32_t* __restrict__ nbr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    const int64_t o = orig[r];
    const int64_t i = o >> 3; const int j = (int)(o & 7);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int kp, jn; child_offset(j, k, kp, jn);
        const int32_t pn = pnbr[(int64_t)kp * np + i];
        const int64_t m = pn < 0 ? -1 : 8 * (int64_t)pn + jn;
        nbr[(int64_t)k * n_out + r] = (m >= 0 && mask[m]) ? prefix[m] : -1;
    }
}
// the same two derivations through the RANK BITMAP pcgc_topk_select writes (12 bytes per query from a structure of 0.19 bytes per
// candidate row instead of 5 bytes from one of 5 bytes per row)
__global__ void __launch_bounds__(256) k_kmap_prune_sel(const int32_t* __restrict__ cand, int64_t n_cand, const uint64_t* __restrict__ bits,
                                                        const int32_t* __restrict__ wprefix, const int32_t* __restrict__ orig,
                                                        int64_t n_out, int32_t* __restrict__ nbr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    int64_t o = orig[r];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int32_t m = cand[(int64_t)k * n_cand + o];
        nbr[(int64_t)k * n_out + r] = m >= 0 ? sel_rank(bits, wprefix, m) : -1;
    }
}
__global__ void __launch_bounds__(256) k_kmap_prune_parent_sel(const int32_t* __restrict__ pnbr, int64_t np, const uint64_t* __restrict__ bits,
                                                               const int32_t* __restrict__ wprefix, const int32_t* __restrict__ orig,
                                                               int64_t n_out, int32_t* __restrict__ nbr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    const int64_t o = orig[r];
    const int64_t i = o >> 3; const int j = (int)(o & 7);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int kp, jn; child_offset(j, k, kp, jn);
        const int32_t pn = pnbr[(int64_t)kp * np + i];
        nbr[(int64_t)k * n_out + r] = pn < 0 ? -1 : sel_rank(bits, wprefix, 8 * (int64_t)pn + jn);
    }
}
// strided pyramid (encoder): fine row c has parent row parent_of[c]; down[j][p] = fine row at slot j of coarse row p
__global__ void __launch_bounds__(256) k_kmap_from_coarse(const int4* __restrict__ fine, int64_t nf, int32_t stride_f,
                                                          const int32_t* __restrict__ parent_of,
                                                          const int32_t* __restrict__ pnbr, const int32_t* __restrict__ down,
                                                          int64_t nc, int32_t* __restrict__ nbr) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nf) return;
    int4 q = fine[c];
    int j = ((q.y / stride_f) & 1) | (((q.z / stride_f) & 1) << 1) | (((q.w / stride_f) & 1) << 2);
    int64_t p = parent_of[c];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        int32_t r;
        if (k == 13) r = (int32_t)c;
        else {
            int kp, jn; child_offset(j, k, kp, jn);
            int32_t pn = pnbr[(int64_t)kp * nc + p];
            r = pn < 0 ? -1 : down[(int64_t)jn * nc + pn];
        }
        nbr[(int64_t)k * nf + c] = r;
    }
}
// parent_of[c] = prefix[first_row[c]] ; down[slot(c)][parent_of[c]] = c   (down pre-filled with -1)
__global__ void k_down_maps(const int4* __restrict__ fine, const int32_t* __restrict__ first_row,
                            const int32_t* __restrict__ prefix, int64_t nf, int32_t stride_f, int64_t nc,
                            int32_t* __restrict__ parent_of, int32_t* __restrict__ down) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nf) return;
    int4 q = fine[c];
    int j = ((q.y / stride_f) & 1) | (((q.z / stride_f) & 1) << 1) | (((q.w / stride_f) & 1) << 2);
    int32_t p = prefix[first_row[c]];
    parent_of[c] = p;
    down[(int64_t)j * nc + p] = (int32_t)c;
}
__global__ void k_compact_index(const uint8_t* __restrict__ mask, const int32_t* __restrict__ prefix, int64_t n, int32_t* orig) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) orig[prefix[i]] = (int32_t)i;
}

extern "C" int pcgc_kmap_k3_children(const int32_t* parent_nbr, int64_t n_parent, int32_t* nbr, void* stream) {
    if (n_parent == 0) return 0;
    hipLaunchKernelGGL(k_kmap_children, dim3(grid_for(8 * n_parent, 256)), dim3(256), 0, S(stream), parent_nbr, n_parent, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_children");
    return 0;
}
extern "C" int pcgc_kmap_k3_prune(const int32_t* cand_nbr, int64_t n_cand, const uint8_t* mask, const int32_t* prefix,
                                  const int32_t* orig, int64_t n_out, int32_t* nbr, void* stream) {
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(k_kmap_prune, dim3(grid_for(n_out, 256)), dim3(256), 0, S(stream), cand_nbr, n_cand, mask, prefix, orig,
                       n_out, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_prune");
    return 0;
}
extern "C" int pcgc_kmap_k3_prune_parent(const int32_t* parent_nbr, int64_t n_parent, const uint8_t* mask, const int32_t* prefix,
                                         const int32_t* orig, int64_t n_out, int32_t* nbr, void* stream) {
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(k_kmap_prune_parent, dim3(grid_for(n_out, 256)), dim3(256), 0, S(stream), parent_nbr, n_parent, mask, prefix,
                       orig, n_out, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_prune_parent");
    return 0;
}
extern "C" int pcgc_kmap_k3_prune_sel(const int32_t* cand_nbr, int64_t n_cand, const uint8_t* bits, const int32_t* wprefix,
                                      const int32_t* orig, int64_t n_out, int32_t* nbr, void* stream) {
    PCGC_REQUIRE(((uintptr_t)bits & 7) == 0, "bits must be 8-byte aligned");
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(k_kmap_prune_sel, dim3(grid_for(n_out, 256)), dim3(256), 0, S(stream), cand_nbr, n_cand, (const uint64_t*)bits, wprefix,
                       orig, n_out, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_prune_sel");
    return 0;
}
extern "C" int pcgc_kmap_k3_prune_parent_sel(const int32_t* parent_nbr, int64_t n_parent, const uint8_t* bits, const int32_t* wprefix,
                                             const int32_t* orig, int64_t n_out, int32_t* nbr, void* stream) {
    PCGC_REQUIRE(((uintptr_t)bits & 7) == 0, "bits must be 8-byte aligned");
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(k_kmap_prune_parent_sel, dim3(grid_for(n_out, 256)), dim3(256), 0, S(stream), parent_nbr, n_parent, (const uint64_t*)bits,
                       wprefix, orig, n_out, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_prune_parent_sel");
    return 0;
}
extern "C" int pcgc_kmap_k3_from_coarse(const int32_t* fine, int64_t n_fine, int32_t stride_fine, const int32_t* parent_of,
                                        const int32_t* coarse_nbr, const int32_t* down, int64_t n_coarse, int32_t* nbr,
                                        void* stream) {
    if (n_fine == 0) return 0;
    hipLaunchKernelGGL(k_kmap_from_coarse, dim3(grid_for(n_fine, 256)), dim3(256), 0, S(stream), (const int4*)fine, n_fine,
                       stride_fine, parent_of, coarse_nbr, down, n_coarse, nbr);
    PCGC_CHECK_LAUNCH("kmap_k3_from_coarse");
    return 0;
}
extern "C" int pcgc_down_maps(const int32_t* fine, const int32_t* first_row, const int32_t* prefix, int64_t n_fine,
                              int32_t stride_fine, int64_t n_coarse, int32_t* parent_of, int32_t* down, void* stream) {
    if (n_coarse > 0) {
        hipError_t e = hipMemsetAsync(down, 0xFF, (size_t)n_coarse * 8 * sizeof(int32_t), S(stream));
        if (e != hipSuccess) { pcgc_set_error("down_maps: %s", hipGetErrorString(e)); return -1; }
    }
    if (n_fine == 0) return 0;
    hipLaunchKernelGGL(k_down_maps, dim3(grid_for(n_fine, 256)), dim3(256), 0, S(stream), (const int4*)fine, first_row, prefix,
                       n_fine, stride_fine, n_coarse, parent_of, down);
    PCGC_CHECK_LAUNCH("down_maps");
    return 0;
}
// One strided level of the encoder pyramid (MinkowskiConvolution k=2 s=2 coordinate side) as two calls around the single
// host read-back of the coarse count: the host-side launch glue between the seven small dependent kernels of a level was
// as long as the kernels themselves (this phase starts behind a synchronisation, so the host cannot run ahead).
extern "C" int pcgc_down_prepare(const int32_t* fine, int64_t n, int32_t stride_fine, int32_t* q, uint64_t* keys, int32_t* vals,
                                 int64_t cap, uint8_t* keep, int32_t* first_row, int32_t* prefix, int32_t* total, void* scan_ws,
                                 size_t scan_ws_bytes, void* stream) {
    int rc;
    if ((rc = pcgc_coords_quantize(fine, n, 2 * stride_fine, q, stream))) return rc;
    if ((rc = pcgc_hash_clear(keys, vals, cap, stream))) return rc;
    if ((rc = pcgc_hash_insert(q, n, 2 * stride_fine, keys, vals, cap, stream))) return rc;
    if ((rc = pcgc_hash_first_mask(q, n, 2 * stride_fine, keys, vals, cap, keep, first_row, stream))) return rc;
    return pcgc_mask_scan(keep, n, prefix, total, scan_ws, scan_ws_bytes, stream);
}
extern "C" int pcgc_down_finish(const int32_t* fine, const int32_t* q, const uint8_t* keep, const int32_t* first_row,
                                const int32_t* prefix, int64_t n, int32_t stride_fine, int64_t n_coarse, int32_t* coarse,
                                int32_t* parent_of, int32_t* down, void* stream) {
    int rc;
    if ((rc = pcgc_compact_coords(q, keep, prefix, n, coarse, stream))) return rc;
    return pcgc_down_maps(fine, first_row, prefix, n, stride_fine, n_coarse, parent_of, down, stream);
}
// The same level in ONE call: prepare, read the coarse count back (the stream is synchronised here), finish into caller-provided
// buffers of upper-bound size (a coarse level has at most n rows; `down` is written as [8][n_coarse] at the front of its buffer).
// Between the read-back and the next launch the host does nothing but this function: the two-call form left the GPU idle for
// ~30 us per level while the interpreter sized and allocated the outputs.
extern "C" int pcgc_down_level(const int32_t* fine, int64_t n, int32_t stride_fine, int32_t* q, uint64_t* keys, int32_t* vals,
                               int64_t cap, uint8_t* keep, int32_t* first_row, int32_t* prefix, int32_t* total, void* scan_ws,
                               size_t scan_ws_bytes, int32_t* coarse, int32_t* parent_of, int32_t* down, int64_t* n_coarse_out,
                               void* stream) {
    PCGC_REQUIRE(n_coarse_out != nullptr, "null count");
    int rc = pcgc_down_prepare(fine, n, stride_fine, q, keys, vals, cap, keep, first_row, prefix, total, scan_ws, scan_ws_bytes, stream);
    if (rc) return rc;
    static thread_local int32_t* host_total = nullptr;         // pinned: the 4-byte copy does not go through a staging buffer
    if (!host_total && hipHostMalloc((void**)&host_total, 64, hipHostMallocDefault) != hipSuccess) {
        host_total = nullptr; pcgc_set_error("down_level: cannot allocate pinned memory"); return -1;
    }
    hipError_t e = hipMemcpyAsync(host_total, total, sizeof(int32_t), hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) { pcgc_set_error("down_level: %s", hipGetErrorString(e)); return -1; }
    const int64_t n_coarse = *host_total;
    *n_coarse_out = n_coarse;
    return pcgc_down_finish(fine, q, keep, first_row, prefix, n, stride_fine, n_coarse, coarse, parent_of, down, stream);
}
// ---- the whole strided pyramid behind ONE read-back ------------------------------------------------------------------------------
// Level l+1 in canonical order is "the distinct keys floor(c / 2^(l+1) s) of level l, by first occurrence in level-l order"; level l
// is itself in first-occurrence order of the input, so level l+1 is equally "the distinct keys of the INPUT rows, by first occurrence
// in input order".  Hence every level can be deduplicated straight from the input rows, all levels in the same launches, and the
// sizes of all of them come back in one synchronising copy (the level-by-level form waits for the GPU once per level, with ~50 us
// of idle device around each wait).  Costs two more passes over the n input rows than the nested form; wins ~0.15 ms per vox10 frame.
constexpr int PYR_MAX = 4;
struct PyrArgs {
    uint64_t* keys[PYR_MAX]; int32_t* vals[PYR_MAX]; int32_t* first[PYR_MAX]; uint8_t* keep[PYR_MAX];
    int32_t s[PYR_MAX]; uint64_t cap_mask; int levels;
};
__device__ static inline int4 pyr_quantise(int4 c, int32_t s) { return make_int4(c.x, c.y / s * s, c.z / s * s, c.w / s * s); }   // (coordinates are >= 0 here)
// Level by level (round 4).  The first input row of a level-l cell is also the first row of its level-(l-1) cell (an earlier row with the
// same finer cell would map to the same coarser cell), so level l only needs the rows that level l - 1 KEPT: 786 k + 256 k + 71 k
// hash insertions for a vox10 frame instead of 3 x 786 k, same tables, same firsts — for two short launches more per level.
__global__ void k_pyr_insert_level(const int4* __restrict__ fine, int64_t n, PyrArgs a, int l) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int4 c = make_int4(-1, -1, -1, -1);
    bool ok = false;
    if (i < n && (l == 0 || a.keep[l - 1][i])) { c = fine[i]; ok = coord_in_range(c.x, c.y, c.z, c.w); }
    const int4 q = pyr_quantise(c, a.s[l]);
    const uint64_t key = ok ? coord_key(q.x, q.y, q.z, q.w) : PCGC_EMPTY_KEY;
    const uint64_t left = __shfl_up((unsigned long long)key, 1, 64);              // runs of equal keys in consecutive lanes: first lane inserts
    if (!ok || ((threadIdx.x & 63) != 0 && left == key)) return;
    uint64_t* keys = a.keys[l]; int32_t* vals = a.vals[l];
    uint64_t h = hash_slot(key, a.cap_mask);
    for (;;) {
        unsigned long long prev = __builtin_nontemporal_load((const unsigned long long*)&keys[h]);
        if (prev == PCGC_EMPTY_KEY) prev = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)PCGC_EMPTY_KEY, (unsigned long long)key);
        if (prev == PCGC_EMPTY_KEY || prev == key) { if (__builtin_nontemporal_load(&vals[h]) > (int32_t)i) atomicMin(&vals[h], (int32_t)i); return; }
        h = (h + 1) & a.cap_mask;
    }
}
__global__ void k_pyr_first_level(const int4* __restrict__ fine, int64_t n, PyrArgs a, int l) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (l > 0 && !a.keep[l - 1][i]) { a.keep[l][i] = 0; return; }                  // (first[l][i] is only read for rows kept below)
    const int4 q = pyr_quantise(fine[i], a.s[l]);
    const int32_t f = hash_lookup(a.keys[l], a.vals[l], a.cap_mask, q.x, q.y, q.z, q.w);
    a.first[l][i] = f;
    a.keep[l][i] = f == (int32_t)i;
}

// After the read-back: for every level at once, the compacted coarse coordinates, and for the level below each (rows = the input rows kept
// by the previous level's mask, all of them for the first) parent_of and the 8-slot down map (pre-filled with -1).
struct PyrOut { int4* coarse[PYR_MAX]; int32_t* parent_of[PYR_MAX]; int32_t* down[PYR_MAX]; const int32_t* prefix[PYR_MAX]; int64_t count[PYR_MAX]; int32_t stride; };
__global__ void k_pyr_finish(const int4* __restrict__ fine, int64_t n, PyrArgs a, PyrOut o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = fine[i];
    bool kept_below = true; int32_t j = (int32_t)i;                 // row i of the input is row j of the level below coarse level l
#pragma unroll
    for (int l = 0; l < PYR_MAX; ++l) {
        if (l >= a.levels || !kept_below) break;
        const int32_t s_cur = o.stride << l;
        const int k = ((c.y / s_cur) & 1) | (((c.z / s_cur) & 1) << 1) | (((c.w / s_cur) & 1) << 2);
        const int32_t p = o.prefix[l][a.first[l][i]];
        o.parent_of[l][j] = p;
        o.down[l][(int64_t)k * o.count[l] + p] = j;
        kept_below = a.keep[l][i] != 0;
        if (kept_below) { o.coarse[l][p] = pyr_quantise(c, a.s[l]); j = p; }
    }
}
extern "C" size_t pcgc_pyramid_scratch_bytes(int64_t n, int levels) {
    const size_t cap = (size_t)pcgc_hash_capacity(n), scan = (pcgc_scan_workspace_bytes(n) + 15) & ~(size_t)15, nn = ((size_t)n + 15) & ~(size_t)15;
    return 64 + (size_t)levels * (scan + cap * 12 + nn * 4 * 2 + nn) + 64;
}
extern "C" int pcgc_pyramid(const int32_t* fine, int64_t n, int32_t stride, int levels, void* scratch, size_t scratch_bytes,
                            int32_t* const* coarse, int32_t* const* parent_of, int32_t* const* down, int64_t* counts, void* stream) {
    PCGC_REQUIRE(levels >= 1 && levels <= PYR_MAX, "1 to 4 levels");
    PCGC_REQUIRE(fine && scratch && coarse && parent_of && down && counts, "null argument");
    PCGC_REQUIRE(scratch_bytes >= pcgc_pyramid_scratch_bytes(n, levels) && ((uintptr_t)scratch & 15) == 0, "scratch too small or misaligned");
    PCGC_REQUIRE(stride >= 1 && (int64_t)stride << levels < ((int64_t)1 << 30), "bad stride");
    if (n == 0) { for (int l = 0; l < levels; ++l) counts[l] = 0; return 0; }
    hipStream_t s = S(stream);
    const size_t cap = (size_t)pcgc_hash_capacity(n), scan = (pcgc_scan_workspace_bytes(n) + 15) & ~(size_t)15, nn = ((size_t)n + 15) & ~(size_t)15;
    PyrArgs a; a.levels = levels; a.cap_mask = (uint64_t)cap - 1;
    int32_t* prefix[PYR_MAX]; void* scan_ws[PYR_MAX];
    // scratch: totals[4] (64 B) | scan workspaces (zeroed together with the totals by ONE memset) | per level keys, vals, first, prefix, keep
    char* p = (char*)scratch;
    int32_t* totals = (int32_t*)p; p += 64;
    for (int l = 0; l < levels; ++l) { scan_ws[l] = p; p += scan; }
    const size_t zero_bytes = (size_t)(p - (char*)scratch);
    uint64_t* keys0 = (uint64_t*)p;                                  // all levels' keys are contiguous, then all vals: one clear
    for (int l = 0; l < levels; ++l) { a.keys[l] = (uint64_t*)p; p += cap * 8; }
    int32_t* vals0 = (int32_t*)p;
    for (int l = 0; l < levels; ++l) { a.vals[l] = (int32_t*)p; p += cap * 4; }
    for (int l = 0; l < levels; ++l) {
        a.first[l] = (int32_t*)p; p += nn * 4;
        prefix[l] = (int32_t*)p; p += nn * 4;
        a.keep[l] = (uint8_t*)p; p += nn;
        a.s[l] = stride << (l + 1);
    }
    int rc;
    hipError_t e = hipMemsetAsync(scratch, 0, zero_bytes, s);
    if (e != hipSuccess) { pcgc_set_error("pyramid: %s", hipGetErrorString(e)); return -1; }
    hipLaunchKernelGGL(k_hash_clear, dim3(grid_for((int64_t)cap * levels, 256)), dim3(256), 0, s, keys0, vals0, (int64_t)cap * levels);
    const dim3 g(grid_for(n, 256)), b(256);
    for (int l = 0; l < levels; ++l) {
        hipLaunchKernelGGL(k_pyr_insert_level, g, b, 0, s, (const int4*)fine, n, a, l);
        hipLaunchKernelGGL(k_pyr_first_level, g, b, 0, s, (const int4*)fine, n, a, l);
    }
    for (int l = 0; l < levels; ++l)
        if ((rc = pcgc_mask_scan_zeroed(a.keep[l], n, prefix[l], totals + l, scan_ws[l], pcgc_scan_workspace_bytes(n), stream))) return rc;
    static thread_local int32_t* host_total = nullptr;
    if (!host_total && hipHostMalloc((void**)&host_total, 64, hipHostMallocDefault) != hipSuccess) { host_total = nullptr; pcgc_set_error("pyramid: cannot allocate pinned memory"); return -1; }
    e = hipMemcpyAsync(host_total, totals, (size_t)levels * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { pcgc_set_error("pyramid: %s", hipGetErrorString(e)); return -1; }
    PyrOut o; o.stride = stride;
    for (int l = 0; l < levels; ++l) {
        counts[l] = host_total[l];
        o.coarse[l] = (int4*)coarse[l]; o.parent_of[l] = parent_of[l]; o.down[l] = down[l]; o.prefix[l] = prefix[l]; o.count[l] = counts[l];
        if (counts[l] > 0 && (e = hipMemsetAsync(down[l], 0xFF, (size_t)counts[l] * 8 * sizeof(int32_t), s)) != hipSuccess) { pcgc_set_error("pyramid: %s", hipGetErrorString(e)); return -1; }
    }
    hipLaunchKernelGGL(k_pyr_finish, g, b, 0, s, (const int4*)fine, n, a, o);
    PCGC_CHECK_LAUNCH("pyramid");
    return 0;
}
extern "C" int pcgc_compact_index(const uint8_t* mask, const int32_t* prefix, int64_t n, int32_t* orig, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_compact_index, dim3(grid_for(n, 256)), dim3(256), 0, S(stream), mask, prefix, n, orig);
    PCGC_CHECK_LAUNCH("compact_index");
    return 0;
}

// ---- D1 (point-to-point) geometry distortion on device -------------------------------------------------------------
// mpeg-pcc-dmetric's D1 (pc_error.py:27-74) is the mean squared distance from every point of A to its nearest neighbour in
// B (and back).  On voxelised clouds the nearest neighbour is found exactly by probing B's coordinate hash at lattice
// offsets visited in ASCENDING squared distance: the first occupied offset is the nearest neighbour.  `offsets` is that
// table (dx,dy,dz,d2 as int4, sorted by d2, all offsets with max|d| <= R).  Squared distances are integers, so the
// float64 sum is exact and independent of the reduction order.  Points with no neighbour inside the table are counted in
// `unresolved` (the host finishes those with a KD-tree; it does not happen for codec outputs).
__global__ void __launch_bounds__(256) k_d1_nn(const int4* __restrict__ a, int64_t na, const uint64_t* __restrict__ keys,
                                               const int32_t* __restrict__ vals, uint64_t cap_mask,
                                               const int4* __restrict__ offsets, int n_off, double* __restrict__ sum,
                                               unsigned long long* __restrict__ max_d2, int32_t* __restrict__ unresolved) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0; unsigned long long mymax = 0; int miss = 0;
    if (i < na) {
        const int4 c = a[i];
        int found = -1;
        for (int t = 0; t < n_off; ++t) {
            const int4 o = offsets[t];
            if (hash_lookup(keys, vals, cap_mask, c.x, c.y + o.x, c.z + o.y, c.w + o.z) >= 0) { found = o.w; break; }
        }
        if (found >= 0) { mine = (double)found; mymax = (unsigned long long)found; } else miss = 1;
    }
    // wave reduction, then one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mine += __shfl_xor(mine, d, 64);
        const unsigned long long om = __shfl_xor(mymax, d, 64); mymax = om > mymax ? om : mymax;
        miss += __shfl_xor(miss, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mine != 0.0) atomicAdd(sum, mine);
        if (mymax) atomicMax(max_d2, mymax);
        if (miss) atomicAdd(unresolved, miss);
    }
}
extern "C" int pcgc_d1_nn(const int32_t* a, int64_t na, const uint64_t* b_keys, const int32_t* b_vals, int64_t b_cap,
                          const int32_t* offsets, int n_offsets, double* sum, uint64_t* max_d2, int32_t* unresolved,
                          void* stream) {
    PCGC_REQUIRE(b_cap > 0 && (b_cap & (b_cap - 1)) == 0, "bad hash capacity");
    hipError_t e = hipMemsetAsync(sum, 0, 8, S(stream));
    if (e == hipSuccess) e = hipMemsetAsync(max_d2, 0, 8, S(stream));
    if (e == hipSuccess) e = hipMemsetAsync(unresolved, 0, 4, S(stream));
    if (e != hipSuccess) { pcgc_set_error("d1_nn: %s", hipGetErrorString(e)); return -1; }
    if (na == 0) return 0;
    hipLaunchKernelGGL(k_d1_nn, dim3(grid_for(na, 256)), dim3(256), 0, S(stream), (const int4*)a, na, b_keys, b_vals,
                       (uint64_t)(b_cap - 1), (const int4*)offsets, n_offsets, sum, (unsigned long long*)max_d2, unresolved);
    PCGC_CHECK_LAUNCH("d1_nn");
    return 0;
}

// ---- the same metric through 4 x 4 x 4 cells (round 4).  The probe-per-lattice-offset form above visits ~4/3 pi d^3 offsets for a point whose
// nearest neighbour is d voxels away: fine for a codec's output (d <= 2), 21 ms on a cloud that is 5-10 voxels off (the random-weight stand-in of
// the bench).  Here cloud B is held as its stride-4 cells (coordinate hash of the cells -> row) with a 64-bit occupancy mask per cell; a query point
// walks the CELL offsets in ascending lower bound of the distance to any voxel of that cell (per axis max(0, 4 |o| - 3)) and stops at the first
// bound that is not below the best distance found: at most a few dozen hash probes, each settling up to 64 voxels by bit scans.  Exact (integer
// squared distances, the same float64 sum); a nearest neighbour farther than the offset table reaches counts as unresolved, as before.
__global__ void k_d1_cell_masks(const int4* __restrict__ b, int64_t nb, const uint64_t* __restrict__ ckeys, const int32_t* __restrict__ cvals,
                                uint64_t cmask, unsigned long long* __restrict__ masks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int4 c = b[i];
    const int32_t row = hash_lookup(ckeys, cvals, cmask, c.x, c.y & ~3, c.z & ~3, c.w & ~3);
    if (row >= 0) atomicOr(&masks[row], 1ull << ((c.y & 3) | ((c.z & 3) << 2) | ((c.w & 3) << 4)));
}
__global__ void __launch_bounds__(256) k_d1_nn_cells(const int4* __restrict__ a, int64_t na, const uint64_t* __restrict__ ckeys,
                                                     const int32_t* __restrict__ cvals, uint64_t cmask,
                                                     const unsigned long long* __restrict__ masks, const int4* __restrict__ offsets, int n_off,
                                                     int32_t reach2, double* __restrict__ sum, unsigned long long* __restrict__ max_d2,
                                                     int32_t* __restrict__ unresolved) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0; unsigned long long mymax = 0; int miss = 0;
    if (i < na) {
        const int4 c = a[i];
        const int lx = c.y & 3, ly = c.z & 3, lz = c.w & 3;
        const int X = c.y & ~3, Y = c.z & ~3, Z = c.w & ~3;
        int best = 0x7FFFFFFF;
        for (int t = 0; t < n_off; ++t) {
            const int4 o = offsets[t];                               // (cell offset x, y, z; lower bound of the squared distance)
            if (o.w >= best) break;
            const int32_t row = hash_lookup(ckeys, cvals, cmask, c.x, X + 4 * o.x, Y + 4 * o.y, Z + 4 * o.z);
            if (row < 0) continue;
            unsigned long long m = masks[row];
            const int bx = 4 * o.x - lx, by = 4 * o.y - ly, bz = 4 * o.z - lz;
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int dx = bx + (bit & 3), dy = by + ((bit >> 2) & 3), dz = bz + (bit >> 4);
                const int d2 = dx * dx + dy * dy + dz * dz;
                best = d2 < best ? d2 : best;
            }
        }
        if (best < reach2) { mine = (double)best; mymax = (unsigned long long)best; } else miss = 1;   // (beyond the table's reach a closer voxel could hide)
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mine += __shfl_xor(mine, d, 64);
        const unsigned long long om = __shfl_xor(mymax, d, 64); mymax = om > mymax ? om : mymax;
        miss += __shfl_xor(miss, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mine != 0.0) atomicAdd(sum, mine);
        if (mymax) atomicMax(max_d2, mymax);
        if (miss) atomicAdd(unresolved, miss);
    }
}
extern "C" int pcgc_d1_cell_masks(const int32_t* b, int64_t nb, const uint64_t* cell_keys, const int32_t* cell_vals, int64_t cell_cap,
                                  uint64_t* masks, int64_t n_cells, void* stream) {
    PCGC_REQUIRE(cell_cap > 0 && (cell_cap & (cell_cap - 1)) == 0, "bad hash capacity");
    hipError_t e = hipMemsetAsync(masks, 0, (size_t)n_cells * 8, S(stream));
    if (e != hipSuccess) { pcgc_set_error("d1_cell_masks: %s", hipGetErrorString(e)); return -1; }
    if (nb == 0) return 0;
    hipLaunchKernelGGL(k_d1_cell_masks, dim3(grid_for(nb, 256)), dim3(256), 0, S(stream), (const int4*)b, nb, cell_keys, cell_vals,
                       (uint64_t)(cell_cap - 1), (unsigned long long*)masks);
    PCGC_CHECK_LAUNCH("d1_cell_masks");
    return 0;
}
extern "C" int pcgc_d1_nn_cells(const int32_t* a, int64_t na, const uint64_t* cell_keys, const int32_t* cell_vals, int64_t cell_cap,
                                const uint64_t* masks, const int32_t* offsets, int n_offsets, int32_t reach2, double* sum, uint64_t* max_d2,
                                int32_t* unresolved, void* stream) {
    PCGC_REQUIRE(cell_cap > 0 && (cell_cap & (cell_cap - 1)) == 0, "bad hash capacity");
    hipError_t e = hipMemsetAsync(sum, 0, 8, S(stream));
    if (e == hipSuccess) e = hipMemsetAsync(max_d2, 0, 8, S(stream));
    if (e == hipSuccess) e = hipMemsetAsync(unresolved, 0, 4, S(stream));
    if (e != hipSuccess) { pcgc_set_error("d1_nn_cells: %s", hipGetErrorString(e)); return -1; }
    if (na == 0) return 0;
    hipLaunchKernelGGL(k_d1_nn_cells, dim3(grid_for(na, 256)), dim3(256), 0, S(stream), (const int4*)a, na, cell_keys, cell_vals,
                       (uint64_t)(cell_cap - 1), (const unsigned long long*)masks, (const int4*)offsets, n_offsets, reach2, sum,
                       (unsigned long long*)max_d2, unresolved);
    PCGC_CHECK_LAUNCH("d1_nn_cells");
    return 0;
}

// The coordinate-only part of a decoder stage on a freshly decoded level, in ONE call (five launches: the host round trips between them
// are what the GPU waits for at the head of a decode): hash of the level, its k3 map, the children level
// (MinkowskiGenerativeConvolutionTranspose's output coordinates, autoencoder.py:155-161) and the children level's k3 map.
// keys / vals: cap = pcgc_hash_capacity(n) entries; nbr [27][n]; children [8 n][4]; nbr_children [27][8 n].
extern "C" int pcgc_level_prepare_children(const int32_t* coords, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals, int64_t cap, int32_t* nbr,
                                           int32_t* children, int32_t* nbr_children, void* stream) {
    PCGC_REQUIRE(coords && keys && vals && nbr && children && nbr_children, "null argument");
    if (int rc = pcgc_hash_clear(keys, vals, cap, stream)) return rc;
    if (int rc = pcgc_hash_insert(coords, n, stride, keys, vals, cap, stream)) return rc;
    if (int rc = pcgc_kmap_k3(coords, n, stride, keys, vals, cap, nbr, stream)) return rc;
    if (int rc = pcgc_coords_children(coords, n, stride, children, stream)) return rc;
    return pcgc_kmap_k3_children(nbr, n, nbr_children, stream);
}

