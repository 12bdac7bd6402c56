// Tile order of the children-level kernels (child_kernels.h): parents sorted by their neighbour-occupancy pattern.
//
// A children-level tile is 16 parents; the halo cells that belong to a neighbour parent are dead work for a parent that does not
// have that neighbour — absent rows arrive as zeros and are multiplied anyway, because an output-stationary MFMA tile cannot skip
// a ROW.  It can skip a CELL when none of its 16 parents has the neighbour.  With parents in their canonical (input) order that
// almost never happens (0.3-5 % of the cell work); with parents grouped by pattern it is 35-49 % on a thin surface
// (tools/pattern_probe.py), close to the bound where every tile holds one exact pattern.  The key ranks the neighbour parents by
// how many halo cells they carry — 6 faces (4 cells each), 12 edges (2), 8 corners (1); the centre is always present — so the
// sort groups by what matters most first.  Optionally the order is local (chunks of consecutive parents keep their place) so
// that neighbouring tiles still gather from the same region.
#include <cstring>
#include "pcgc_common.h"
#include <rocprim/rocprim.hpp>

namespace {
// neighbour-parent index kp = (dz+1)*9 + (dy+1)*3 + (dx+1); rank by cells carried: faces, edges, corners
struct PlanOrder { int k[26]; };
constexpr PlanOrder plan_order() {
    PlanOrder o{};
    int n = 0;
    for (int want = 1; want <= 3; ++want)                      // number of non-zero axes: 1 = face, 2 = edge, 3 = corner
        for (int kp = 0; kp < 27; ++kp) {
            const int dx = kp % 3 - 1, dy = (kp / 3) % 3 - 1, dz = kp / 9 - 1;
            const int nz = (dx != 0) + (dy != 0) + (dz != 0);
            if (nz == want) o.k[n++] = kp;
        }
    return o;
}
__constant__ PlanOrder c_plan_order = plan_order();

__global__ void k_plan_keys(const int32_t* __restrict__ nbr, int64_t n, int chunk_shift, int pattern_bits, uint32_t* __restrict__ keys,
                            int32_t* __restrict__ idx) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t pat = 0;
#pragma unroll
    for (int r = 0; r < 26; ++r) pat = (pat << 1) | (nbr[(int64_t)c_plan_order.k[r] * n + p] >= 0 ? 1u : 0u);
    const uint32_t chunk = chunk_shift >= 0 ? (uint32_t)(p >> chunk_shift) : 0u;
    keys[p] = (chunk << pattern_bits) | (pat >> (26 - pattern_bits));
    idx[p] = (int32_t)p;
}
__global__ void k_plan_apply(const int32_t* __restrict__ nbr, int64_t n, int64_t n_slots, const int32_t* __restrict__ order,
                             int32_t* __restrict__ perm, int32_t* __restrict__ nbr_sorted) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 28 * n_slots) return;
    const int64_t i = t % n_slots;
    const int row = (int)(t / n_slots);
    const int32_t p = i < n ? order[i] : -1;
    if (row == 27) perm[i] = p;
    else nbr_sorted[(int64_t)row * n_slots + i] = p >= 0 ? nbr[(int64_t)row * n + p] : -1;
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t plan_sort_temp(int64_t n) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs((void*)nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (size_t)n, 0, 32, (hipStream_t)0);
    return tmp;
}
}  // namespace

extern "C" size_t pcgc_child_plan_workspace_bytes(int64_t n_parent) {
    if (n_parent < 1) n_parent = 1;
    return 3 * align256((size_t)n_parent * 4) + align256((size_t)n_parent * 4) + align256(plan_sort_temp(n_parent));
}

extern "C" int pcgc_child_plan(const int32_t* parent_nbr, int64_t n_parent, int64_t chunk_rows, int32_t* perm, int32_t* nbr_sorted,
                               void* workspace, size_t workspace_bytes, void* stream) {
    PCGC_REQUIRE(parent_nbr && perm && nbr_sorted && workspace, "null argument");
    PCGC_REQUIRE(workspace_bytes >= pcgc_child_plan_workspace_bytes(n_parent), "workspace too small");
    PCGC_REQUIRE(n_parent < ((int64_t)1 << 31) - 16, "too many parents");
    if (n_parent == 0) return 0;
    const int64_t n_slots = (n_parent + 15) & ~(int64_t)15;
    hipStream_t s = S(stream);
    // key = chunk index (high) | leading pattern bits: both must fit 32 bits; a level of up to 2^(32-19) chunks keeps 19 pattern bits
    // (faces + edges + one corner), beyond that the pattern is truncated further
    int chunk_shift = -1, chunk_bits = 0;
    if (chunk_rows > 0) {
        chunk_shift = 0;
        while (((int64_t)1 << chunk_shift) < chunk_rows) ++chunk_shift;
        while ((((n_parent - 1) >> chunk_shift) >> chunk_bits) != 0) ++chunk_bits;
    }
    int pattern_bits = 32 - chunk_bits;
    if (pattern_bits > 26) pattern_bits = 26;
    if (pattern_bits < 6) { pattern_bits = 6; chunk_bits = 26; }
    char* ws = (char*)workspace;
    uint32_t* kin = (uint32_t*)ws; ws += align256((size_t)n_parent * 4);
    uint32_t* kout = (uint32_t*)ws; ws += align256((size_t)n_parent * 4);
    int32_t* idx = (int32_t*)ws; ws += align256((size_t)n_parent * 4);
    int32_t* order = (int32_t*)ws; ws += align256((size_t)n_parent * 4);
    size_t tmp = plan_sort_temp(n_parent);
    hipLaunchKernelGGL(k_plan_keys, dim3(grid_for(n_parent, 256)), dim3(256), 0, s, parent_nbr, n_parent, chunk_shift, pattern_bits, kin, idx);
    hipError_t e = rocprim::radix_sort_pairs((void*)ws, tmp, kin, kout, idx, order, (size_t)n_parent, 0, (unsigned)(pattern_bits + chunk_bits > 32 ? 32 : pattern_bits + chunk_bits), s);
    if (e != hipSuccess) { pcgc_set_error("child_plan: %s", hipGetErrorString(e)); return -1; }
    hipLaunchKernelGGL(k_plan_apply, dim3(grid_for(28 * n_slots, 256)), dim3(256), 0, s, parent_nbr, n_parent, n_slots, order, perm, nbr_sorted);
    PCGC_CHECK_LAUNCH("child_plan");
    return 0;
}
