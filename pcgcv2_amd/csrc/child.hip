// Sparse k3 convolutions on CHILDREN levels (the decoder's levels: every row 8p+j is child j of parent p, produced by
// MinkowskiGenerativeConvolutionTranspose, autoencoder.py:155-161,182-188,209-215) as fp32-MFMA kernels that gather each
// input row ONCE per parent tile instead of once per output row.
//
// Geometry.  The 8 children of a parent occupy a 2x2x2 block; their 3x3x3 neighbourhoods together cover the 4x4x4 "halo" of
// cells c = (cx,cy,cz), c* in {-1,0,1,2} (child units, relative to the parent's origin).  Cell c belongs to the neighbour
// parent at offset P = floor(c/2) in {-1,0,0,1} per axis and is its child j' = c & 1 per axis.  Child j reaches cell c iff
// |c - j| <= 1 on every axis, through kernel offset k = (c-j+1) (x fastest).  So per parent: 64 gathered rows feed
// 216 = 8 x 27 (row, offset) pairs — 3.4x fewer gathered rows than the per-output-row gather kernels of conv.hip — and the
// kernel map needed is the PARENT level's [27][n_p] (8x smaller than the children level's own map, which these kernels never
// read).
//
// One wave = 16 parents (one MFMA M-tile; 128 output rows).  For each cell, in ascending (cz,cy,cx) order:
//   A operand  = rows 8*pnbr[kp(c)][p] + j'(c) of the 16 parents, fetched by one `buffer_load_dwordx4 ... lds` per 16-channel
//                block (4 adjacent lanes per 64-byte row segment; absent neighbours use an out-of-range offset and land as
//                zeros) into a per-wave ring of D cells, so D-1 cells of gather are in flight behind the MFMAs;
//   B operands = one lane-linear 1 KB fragment per (cell, accumulator tile), read from an LDS-resident table with one
//                conflict-free ds_read_b128 per lane (4 K-steps at once);
//   MFMA       = v_mfma_f32_16x16x4_f32 into the accumulator tiles the cell reaches.
// The 64 cells are unrolled at compile time (which tiles a cell feeds is static geometry: no branches around the MFMAs);
// WHERE a (cell, tile) pair's B fragment sits in the table is data (the "plan": one row of byte offsets per cell, read by
// scalar loads), so one kernel serves every layer shape: plain convs (tile = (child j, 16 output columns), fragment = the
// offset's weight slice) and the narrow layers whose N dimension packs (child, output channel) pairs with zero columns
// where a child does not reach the cell (host-built tables, pcgcv2_amd/ops.py).
//
// Numerics: per output element the products arrive in ascending cell order = ascending kernel offset k, and inside a cell in
// ascending input channel (16-channel block, K-step, K index) — the canonical fmaf chain of DESIGN.md §3.  A zero B column
// or an absent (zero) row adds fma(x, 0, acc) = acc.  Bit-identical to the per-row kernels and the oracle (tests).
#include "pcgc_common.h"
#include "mfma_util.h"
#include <type_traits>

namespace {

struct ChildEpi {
    const float* bias;      // [cols]
    const float* res;       // residual rows (children level) or nullptr
    int res_ld;
    int relu;
    float* out;             // children-level rows [8 n_p][out_ld]
    int out_ld;
    int nt;                 // EPI 0: column tiles per child
};

// ---- static halo geometry (cell index c = (cz'*4 + cy')*4 + cx', c' = c + 1 in 0..3) -------------------------------------
constexpr int halo_p1(int c) { return c == 0 ? 0 : (c == 3 ? 2 : 1); }           // neighbour-parent offset + 1
constexpr int halo_bit(int c) { return (c == 0 || c == 2) ? 1 : 0; }              // which child of that parent (per axis)
constexpr int cell_kp(int c) { return halo_p1(c >> 4) * 9 + halo_p1((c >> 2) & 3) * 3 + halo_p1(c & 3); }
constexpr int cell_child(int c) { return halo_bit(c & 3) + 2 * halo_bit((c >> 2) & 3) + 4 * halo_bit(c >> 4); }
constexpr bool axis_reach(int c, int jb) { return c - jb >= 0 && c - jb <= 2; }
constexpr unsigned cell_reach(int c) {                                             // bit j = child j's window contains the cell
    unsigned m = 0;
    for (int j = 0; j < 8; ++j)
        if (axis_reach(c & 3, j & 1) && axis_reach((c >> 2) & 3, (j >> 1) & 1) && axis_reach(c >> 4, j >> 2)) m |= 1u << j;
    return m;
}
constexpr int cell_k(int c, int j) {                                               // kernel offset through which child j sees cell c
    return ((c >> 4) - (j >> 2)) * 9 + (((c >> 2) & 3) - ((j >> 1) & 1)) * 3 + ((c & 3) - (j & 1));
}
constexpr bool cell_is_own(int c, int j) { return cell_k(c, j) == 13 && (cell_reach(c) >> j & 1); }   // the cell IS child j

// ---- layer variants: which children have columns in accumulator tile t, which K-steps of a 16-channel block the tile
//      consumes, and which B fragment (index into the table, units of NB KB) a (cell, tile) pair multiplies by ------------
template <int NB_, int NT>
struct PlainConv {                     // k3 conv Cin = 16 NB -> Cout = 16 NT: tile t = (child j, column tile n), fragment = slice of offset k
    static constexpr int NB = NB_, T = 8 * NT;
    static constexpr unsigned children(int t) { return 1u << (t / NT); }
    static constexpr unsigned ksteps(int) { return 0xF; }
    static constexpr bool active(int c, int t) { return (cell_reach(c) & children(t)) != 0; }
    static constexpr int frag(int c, int t) { return cell_k(c, t / NT) * NT + t % NT; }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The gather + MFMA main loop shared by every variant: leaves acc[t] (t < V::T) for the epilogue.
template <class V, int NW, int D, int ROWCHUNKS = 4>
__device__ __forceinline__ bool child_mainloop(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
                                               const float* __restrict__ table, int table_bytes, unsigned char* lds_raw,
                                               f32x4 (&acc)[V::T], int64_t& p0_out) {
    constexpr int NB = V::NB, T = V::T;
    static_assert((D & (D - 1)) == 0, "ring depth must be a power of two");
    static_assert((D - 1) * NB < 64, "vmcnt is 6 bits");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* tab = (float4*)lds_raw;
    float4* ring = (float4*)(lds_raw + table_bytes) + wave * (D * NB * 64);

    for (int i = threadIdx.x; i < table_bytes / 16; i += NW * 64) tab[i] = ((const float4*)table)[i];
    const int64_t p0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * NW + wave) * 16;
    p0_out = p0;
    const bool active = p0 < n_p;                              // idle waves still help staging and take the barrier
    const int mi = lane & 15, mq = lane >> 4;
    const int dma_r = lane >> 2;                               // tile row (parent) this lane fetches for
    const bool row_ok = active && p0 + dma_r < n_p;
    int pn[27];                                                // its 27 neighbour parents (-1 = absent)
#pragma unroll
    for (int kp = 0; kp < 27; ++kp) pn[kp] = pnbr[(int64_t)kp * n_p + (row_ok ? p0 + dma_r : 0)];
#pragma unroll
    for (int kp = 0; kp < 27; ++kp) pn[kp] = row_ok ? pn[kp] : -1;
    __syncthreads();
    if (!active) return false;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(8 * n_p * in_ld * 4), 0x00020000);
    const int f_a = (0x78 >> (2 * (mi >> 2))) & 3;             // read-side swizzle of the A image (see conv.hip v2)
    const int dma_chunk = (lane & 3) ^ ((0x78 >> (2 * ((dma_r >> 2) & 3))) & 3);
    const bool chunk_ok = dma_chunk < ROWCHUNKS;               // rows narrower than 64 bytes: the other lanes fetch nothing (zeros)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // pn[] loaded: from here on vmcnt counts the gather DMAs only

    auto issue = [&](auto ic) {
        constexpr int c = decltype(ic)::value;
        const int pr = pn[cell_kp(c)];
        float4* dst = ring + (c & (D - 1)) * (NB * 64);
        const int64_t rowoff = (int64_t)(8 * pr + cell_child(c)) * in_ld + dma_chunk * 4;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const unsigned voff = (pr >= 0 && chunk_ok) ? (unsigned)((rowoff + cb * 16) * 4) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + cb * 64), 16, (int)voff, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    };

#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4* tab_lane = (const float4*)lds_raw + lane;

    static_for<0, D>(issue);
    static_for<0, 64>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        constexpr int younger = (63 - c) < (D - 1) ? (63 - c) : (D - 1);       // cells issued after c that may stay in flight
        wait_vmcnt<younger * NB>();
        const float4* abase = ring + (c & (D - 1)) * (NB * 64);
        f32x4 araw[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) araw[cb] = lds_ld128_raw(abase + cb * 64 + mi * 4 + (mq ^ f_a));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) lds_tie(araw[cb]);
        if constexpr (c + D < 64) issue(std::integral_constant<int, c + D>{});    // refill this ring slot
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            float4 a = make_float4(araw[cb][0], araw[cb][1], araw[cb][2], araw[cb][3]);
            lane_transpose4(a);
            f32x4 b[T];
            static_for<0, T>([&](auto it) {
                constexpr int t = decltype(it)::value;
                if constexpr (V::active(c, t)) b[t] = lds_ld128_raw(tab_lane + (V::frag(c, t) * NB + cb) * 64);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<0, T>([&](auto it) {
                constexpr int t = decltype(it)::value;
                if constexpr (V::active(c, t)) lds_tie(b[t]);
            });
            static_for<0, 4>([&](auto ij) {
                constexpr int jj = decltype(ij)::value;
                const float av = jj == 0 ? a.x : (jj == 1 ? a.y : (jj == 2 ? a.z : a.w));
                static_for<0, T>([&](auto it) {
                    constexpr int t = decltype(it)::value;
                    if constexpr (V::active(c, t) && ((V::ksteps(t) >> jj) & 1))
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t][jj], acc[t], 0, 0, 0);
                });
            });
        }
    });
    return true;
}

// plain conv:  acc[t][r] = out[8 (p0 + 4 mq + r) + j][16 n + mi],  t = j * NT + n
template <int NB, int NT, int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_child_conv(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
             const float* __restrict__ table, int table_bytes, ChildEpi ep) {
    using V = PlainConv<NB, NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    f32x4 acc[V::T];
    int64_t p0;
    if (!child_mainloop<V, NW, D>(pnbr, n_p, in, in_ld, table, table_bytes, lds_raw, acc, p0)) return;
    const int lane = threadIdx.x & 63, mi = lane & 15, mq = lane >> 4;
#pragma unroll
    for (int t = 0; t < V::T; ++t) {
        const int j = t / NT, n = t % NT;
        const int col = 16 * n + mi;
        const float bv = ep.bias ? ep.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = p0 + 4 * mq + r;
            if (p >= n_p) continue;
            const int64_t row = 8 * p + j;
            float v = acc[t][r];
            if (ep.bias) v = v + bv;
            if (ep.res) v = v + ep.res[row * ep.res_ld + col];
            if (ep.relu) v = fmaxf(v, 0.0f);
            ep.out[row * ep.out_ld + col] = v;
        }
    }
}

template <typename K>
int child_lds_limit(K kern, size_t lds, size_t& granted) {
    if (lds > 160 * 1024) { pcgc_set_error("conv_child: %zu bytes of LDS needed", lds); return -2; }
    if (lds > 48 * 1024 && lds > granted) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pcgc_set_error("conv_child: cannot raise the LDS limit to %zu: %s", lds, hipGetErrorString(e)); return -1; }
        granted = lds;
    }
    return 0;
}

template <int NB, int NT, int NW, int D>
int launch_child_conv(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                      const ChildEpi& ep, hipStream_t s) {
    const size_t lds = (size_t)table_bytes + (size_t)NW * D * NB * 1024;
    auto kern = k_child_conv<NB, NT, NW, D>;
    static size_t granted = 0;
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(n_p, 16 * NW)), dim3(NW * 64), lds, s, pnbr, n_p, in, in_ld, table, table_bytes, ep);
    return 0;
}

}  // namespace

static int g_child_nw = 0, g_child_depth = 0;                  // 0 = default; A/B switches
extern "C" int pcgc_set_child_tuning(int waves, int depth) { g_child_nw = waves; g_child_depth = depth; return 0; }

// Plain k3 conv on a children level.  parent_nbr: [27][n_parent] k3 map of the PARENT level; in/out/residual: children-level
// rows (8 n_parent).  table: the layer's `kernel` re-laid-out as B fragments (ops.child_conv_table).
extern "C" int pcgc_conv_child(const int32_t* parent_nbr, int64_t n_parent, const float* in, int Cin, int in_ld,
                               const float* table, int64_t table_bytes, const float* bias,
                               const float* residual, int res_ld, int relu, float* out, int Cout, int out_ld, void* stream) {
    PCGC_REQUIRE(parent_nbr && in && table && out, "null argument");
    PCGC_REQUIRE((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");
    PCGC_REQUIRE(8 * n_parent * (int64_t)in_ld * 4 < (int64_t)0xFFFFFFF0, "tensor too large for 32-bit buffer offsets");
    PCGC_REQUIRE(table_bytes % 16 == 0 && table_bytes == (int64_t)27 * Cin * Cout * 4, "table size");
    if (n_parent == 0) return 0;
    hipStream_t s = S(stream);
    ChildEpi ep{bias, residual, res_ld, relu, out, out_ld, Cout / 16};
    int rc = -2;
    const int nw = g_child_nw, d = g_child_depth;
    const int tb = (int)table_bytes;
    if (Cin == 16 && Cout == 16) {              // measured on 2.05 M rows (us): (4 waves, ring 4) 308, (4,2) 280, (4,8) 308, (8,4) 257
        if (nw == 4 && d == 2) rc = launch_child_conv<1, 1, 4, 2>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else if (nw == 4) rc = launch_child_conv<1, 1, 4, 4>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else if (d == 2) rc = launch_child_conv<1, 1, 8, 2>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else if (nw == 16) rc = launch_child_conv<1, 1, 16, 2>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else rc = launch_child_conv<1, 1, 8, 4>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
    } else if (Cin == 32 && Cout == 32) {       // 108 KB of weights: one workgroup per CU
        if (nw == 4) rc = launch_child_conv<2, 2, 4, 4>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else if (d == 1) rc = launch_child_conv<2, 2, 8, 1>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else rc = launch_child_conv<2, 2, 8, 2>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
    } else {
        pcgc_set_error("conv_child: unsupported shape %d -> %d (16->16, 32->32)", Cin, Cout);
        return -2;
    }
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("conv_child");
    return 0;
}
