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
#pragma once
#include "pcgc_common.h"
#include "mfma_util.h"
#include <type_traits>

#ifdef PCGC_CHILD_TIMING
// per-phase shader-clock cycles summed over waves: [0] tile prologue (neighbour-parent loads + drain), [1] gather/MFMA loop,
// [3] tiles, [4] whole tile iterations (prologue + loop + epilogue), [5] the wait for the tile's stores; experiments only
// (tools/child_phase_times.py).  Accumulated in registers, ONE atomic per slot and wave at the end of the kernel (atomics inside the
// loop would sit in the very vmcnt queue the kernel's waits count).
static __device__ unsigned long long g_child_dbg[8];
#define CHILD_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define CHILD_TADD(slot, a, b) do { child_dbg[slot] += (b) - (a); } while (0)
#define CHILD_DBG_DECL unsigned long long child_dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CHILD_DBG_PARAM , unsigned long long (&child_dbg)[8]
#define CHILD_DBG_ARG , child_dbg
#define CHILD_TFLUSH do { if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 8; ++q_) if (child_dbg[q_]) atomicAdd(&g_child_dbg[q_], child_dbg[q_]); } } while (0)
// the counters are per translation unit (static __device__): every unit with kernels exports its own reader
#define CHILD_TIMING_READER(NAME)                                                                                      \
    extern "C" int NAME(unsigned long long* out8, int reset) {                                                         \
        (void)hipDeviceSynchronize();                                                                                  \
        if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_child_dbg), 8 * sizeof(unsigned long long));           \
        if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_child_dbg), z, sizeof(z)); }  \
        return 0;                                                                                                      \
    }
#else
#define CHILD_T(var)
#define CHILD_TADD(slot, a, b)
#define CHILD_DBG_DECL
#define CHILD_DBG_PARAM
#define CHILD_DBG_ARG
#define CHILD_TFLUSH
#define CHILD_TIMING_READER(NAME)
#endif

struct ChildEpi {
    const float* bias;      // [cols]
    const float* res;       // residual rows (children level) or nullptr
    int res_ld;
    int relu;
    float* out;             // children-level rows [8 n_p][out_ld]
    int out_ld;
    int nt;                 // EPI 0: column tiles per child
};
struct IrnEpi {
    const float* b0;        // pass A: b00 ; pass B: b01
    const float* b1;        // pass A: b10 ; pass B: b11
    const float* b2;        // pass B: b12
    const float* x;         // pass B: the block input (residual), children rows [.., x_ld]
    int x_ld;
    float* out;             // pass A: t [.., 2Q] ; pass B: out [.., out_ld]
    int out_ld;
};
namespace {


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
constexpr int cz_of(int c) { return c >> 4; }
constexpr int cy_of(int c) { return (c >> 2) & 3; }
constexpr int cx_of(int c) { return c & 3; }
constexpr bool in02(int v) { return v >= 0 && v <= 2; }

// Where a variant's gathered rows come from.  Children levels (the default): the 64 halo cells of a 16-parent tile — cell c is child
// cell_child(c) of the neighbour parent at map offset cell_kp(c), rows 8 p + j.  A variant may override these (rows_irn.hip: plain
// levels, "cell" k = kernel offset k of the tile's own 16 rows, one map row per offset).
struct HaloGeometry {
    static constexpr int NCELLS = 64, ROW_MUL = 8, NMAP = 27;      // NMAP: rows of the kernel map ([NMAP][n])
    static constexpr int kp(int c) { return cell_kp(c); }
    static constexpr int child(int c) { return cell_child(c); }
    static constexpr int byte_off(int) { return 0; }            // a cell may be a PART of a row: byte offset of its first channel (rows_irn.hip)
};

// ---- layer variants.  A variant says: how wide the gathered rows are (NB 16-channel blocks, ROWCHUNKS 16-byte chunks present
//      per block), how many accumulator tiles there are (T), how many K-steps of a block a tile consumes (KS, starting at
//      kfirst(t)), which cells feed a tile (active) and which B fragment of the table a (cell, tile) pair multiplies by (frag).
//      A fragment is lane-linear: lane l holds its KS K-step values contiguously (one ds_read_b32/b64/b128).  With HALF only
//      8 of a tile's 16 columns are meaningful: lanes of columns 8-15 alias columns 0-7 (their results are never stored).
// HZ: -1 = all eight children; 0 / 1 = the four children with that z bit (half units: see k_child_irn_a)
template <int NB_, int NT, int HZ = -1>
struct PlainConv : HaloGeometry {                     // k3 conv Cin = 16 NB -> Cout = 16 NT: tile t = (child j, column tile n), fragment = slice of offset k
    static constexpr int NB = NB_, ROWCHUNKS = 4, T = 8 * NT, KS = 4;
    static constexpr int Z_HALF = HZ;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int c, int t) { return (HZ < 0 || (((t / NT) >> 2) & 1) == HZ) && ((cell_reach(c) >> (t / NT)) & 1); }
    static constexpr int frag(int c, int t) { return cell_k(c, t / NT) * NT + t % NT; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 64 * KS * 4; }      // byte offset in the table
};
// k3 conv C -> 1 (classification head): one tile, column j (0..7) = child j.  B fragments are NOT stored per cell (round 4: 64 cells x NB x
// 512 B = 128 KB at C = 64, which left seven waves one ring slot each — every cell's gather latency exposed, 57 us for 1171 tiles): the
// weights sit in LDS once, in a 5 x 5 x 5 offset space k' = (kz + 1, ky + 1, kx + 1) with kz, ky, kx in -1 .. 3 whose rows outside 0 .. 2 are
// ZERO, row k' = [cb][mq][jj] -> W[k][16 cb + 4 jj + mq] (+ 16 bytes of padding: rows start 17 / 9 / 5 sixteen-byte groups apart, and the
// eight children's rows of one cell — k' offsets {0, 1, 5, 6, 25, 26, 30, 31} — fall on eight different groups).  Child j of the tile sees
// cell c through k' = cellterm(c) + 31 - j', j' = 25 jz + 5 jy + jx: a lane's fragment of (cell, block) is ONE ds_read_b128 at a per-lane
// base + a compile-time offset, and a child that does not reach the cell reads a zero row — the same products as the stored fragments.
template <int NB_>
struct ClsHead : HaloGeometry {
    static constexpr int NB = NB_, ROWCHUNKS = 4, T = 1, KS = 4;
    static constexpr bool HALF = true;
    static constexpr int ROWB = NB * 64 + 16;                   // bytes per k' row
    static constexpr int TABLE_BYTES = (125 * ROWB + 1023) / 1024 * 1024;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int, int) { return true; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int, int cb) { return (cz_of(c) * 25 + cy_of(c) * 5 + cx_of(c)) * ROWB + cb * 64; }
    __device__ static inline int lane_table_off(int mi, int mq) {       // byte offset of the lane's base in the table
        const int j = mi & 7;
        return (31 - (25 * (j >> 2) + 5 * ((j >> 1) & 1) + (j & 1))) * ROWB + mq * 16;
    }
};
// InceptionResNet pass A (autoencoder.py:52-57 first half): conv0_0 (k3 C -> Q) and conv1_0 (k1 C -> Q), Q = C/4.
// Columns pack (child, output channel): 16/Q children per tile.  Tiles [0, T/2) = conv0_0, [T/2, T) = conv1_0 (fed only by the
// cell that IS the child: offset k = 13).
// HZ: -1 = all eight children; 0 / 1 = only the four children with that z bit (a half unit: see k_child_irn_a)
template <int C, int HZ = -1>
struct PassA : HaloGeometry {
    static_assert(C == 16 || C == 32, "children-level passes: C = 16, 32 (C = 64 runs on the rows kernels through the level's own map)");
    static constexpr int Q = C / 4, NB = C / 16, ROWCHUNKS = 4, CPT = 16 / Q /*children per tile: 4, 2 or 1*/, TH = 8 / CPT, T = 2 * TH, KS = 4;
    static_assert(HZ < 0 || CPT <= 2, "half units: tiles of one or two children (a z-half tile holds both halves' columns)");
    static constexpr int Z_HALF = HZ;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    // tile geometry: CPT == 4: tile = z-half (children 4 jz + {0..3});  CPT == 2: tile = (jz, jy) quarter (children 4 jz + 2 jy + {0,1})
    static constexpr int tz(int t) { return CPT == 4 ? (t % TH) : (t % TH) >> 1; }
    static constexpr int ty(int t) { return (t % TH) & 1; }
    static constexpr bool active(int c, int t) {
        const int kz = cz_of(c) - tz(t);
        if (HZ >= 0 && tz(t) != HZ) return false;               // (CPT == 2: tile = (jz, jy) quarter)
        if (CPT == 4) {
            if (t < TH) return in02(kz);
            return kz == 1 && (cy_of(c) == 1 || cy_of(c) == 2) && (cx_of(c) == 1 || cx_of(c) == 2);
        }
        const int ky = cy_of(c) - ty(t);
        if (t < TH) return in02(kz) && in02(ky);
        return kz == 1 && ky == 1 && (cx_of(c) == 1 || cx_of(c) == 2);
    }
    static constexpr int N0 = CPT == 4 ? 48 : (CPT == 2 ? 36 : 27);                 // conv0_0 fragments
    static constexpr int frag(int c, int t) {
        const int kz = cz_of(c) - tz(t);
        if (CPT == 4) return t < TH ? kz * 16 + (c & 15) : N0 + (cy_of(c) - 1) * 2 + (cx_of(c) - 1);
        const int ky = cy_of(c) - ty(t);
        return t < TH ? (kz * 3 + ky) * 4 + cx_of(c) : N0 + (cx_of(c) - 1);
    }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 64 * KS * 4; }
};
// InceptionResNet pass B: the gathered rows are t = [relu(conv0_0) | relu(conv1_0)] (2Q wide).  conv0_1 (k3 Q -> 2Q) reads the
// first Q channels, conv1_1 (k3 Q -> Q) the last Q: with Q = 8 (C = 32) those are K-steps {0,1} / {2,3} of the one 16-channel
// block, with Q = 4 (C = 16) K-step 0 / 1 of a half-width block.
// T2 (C = 16, round 5): the gathered tensor t was written by the quad-block pass A (child_q4.h) in its store-friendly layout — per parent
// 256 bytes = [z half h][conv 0 / 1][child 4 h + s][4 channels], i.e. chunk c (conv0_0 / conv1_0 half) of row 8 p + j sits at
// 256 p + 128 (j >> 2) + 64 c + 16 (j & 3) instead of 256 p + 32 j + 16 c.  Only the gather addresses change.
template <int C, int HZ = -1, bool T2_ = false>
struct PassB : HaloGeometry {
    static constexpr bool T2 = T2_;
    static_assert(!T2_ || C == 16, "the split layout exists for C = 16");
    static constexpr int Q = C / 4, NB = 1, ROWCHUNKS = Q / 2 /*2Q floats = Q/2 chunks*/, KS = Q / 4;
    static constexpr int CPT0 = 16 / (2 * Q) /*children per conv0_1 tile: 1 or 2*/, T0 = 8 / CPT0, CPT1 = 16 / Q, T1 = 8 / CPT1, T = T0 + T1;
    static constexpr int Z_HALF = HZ;
    static_assert(HZ < 0 || CPT1 <= 2, "half units: no tile may hold children of both z halves");
    static constexpr bool HALF = false;
    static constexpr int kfirst(int t) { return t < T0 ? 0 : KS; }
    static constexpr int tile_z(int t) { return t < T0 ? ((t * CPT0) >> 2) & 1 : (((t - T0) * CPT1) >> 2) & 1; }   // z bit of the tile's children
    static constexpr bool active(int c, int t) {
        if (HZ >= 0 && tile_z(t) != HZ) return false;
        if (t < T0) {
            if (CPT0 == 1) return (cell_reach(c) >> t) & 1;
            return in02(cz_of(c) - (t >> 1)) && in02(cy_of(c) - (t & 1));           // (jz, jy) quarters
        }
        const int u = t - T0;
        if (CPT1 == 2) return in02(cz_of(c) - (u >> 1)) && in02(cy_of(c) - (u & 1));
        return in02(cz_of(c) - u);                                                    // z halves
    }
    static constexpr int N0 = CPT0 == 1 ? 27 : 36;
    static constexpr int N1 = CPT1 == 2 ? 36 : 48;
    static constexpr int FRAG_W12 = N0 + N1;                                          // the k1 conv1_2 weights ride behind the table
    static constexpr int frag(int c, int t) {
        if (t < T0) {
            if (CPT0 == 1) return cell_k(c, t);
            return ((cz_of(c) - (t >> 1)) * 3 + (cy_of(c) - (t & 1))) * 4 + cx_of(c);
        }
        const int u = t - T0;
        if (CPT1 == 2) return N0 + ((cz_of(c) - (u >> 1)) * 3 + (cy_of(c) - (u & 1))) * 4 + cx_of(c);
        return N0 + (cz_of(c) - u) * 16 + (c & 15);
    }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 64 * KS * 4; }
};

template <class V, class = void> struct child_t2_layout : std::false_type {};
template <class V> struct child_t2_layout<V, std::enable_if_t<V::T2>> : std::true_type {};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// B-fragment reads with the (compile-time) table offset in the instruction's 16-bit offset field: an address register per
// fragment would cost ~one VGPR per distinct fragment (hipcc hoists the additions out of the tile loop).
template <int KS> struct BFrag;
template <> struct BFrag<4> {
    f32x4 v;
    template <int OFF> __device__ __forceinline__ void load(unsigned base) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory"); }
    __device__ __forceinline__ void tie() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ float get(int i) const { return v[i]; }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <> struct BFrag<2> {
    f32x2 v;
    template <int OFF> __device__ __forceinline__ void load(unsigned base) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory"); }
    __device__ __forceinline__ void tie() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ float get(int i) const { return v[i]; }
};
template <> struct BFrag<1> {
    float v;
    template <int OFF> __device__ __forceinline__ void load(unsigned base) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory"); }
    __device__ __forceinline__ void tie() { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ float get(int) const { return v; }
};
template <int OFF>
__device__ __forceinline__ float lds_ld32_off(unsigned base) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_ld128_off(unsigned base) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}

template <class V> constexpr int frag_floats() { return (V::HALF ? 32 : 64) * V::KS; }

// Persistent tile order: XCD x (workgroups b = x mod 8; each XCD has its own L2) walks one contiguous eighth of the tiles, and
// inside it consecutive tiles go to different workgroups first, then to the next wave slot: the i-th tile of (workgroup b, wave w).
// A level's tile count is rarely a multiple of the resident waves (e.g. 4451 tiles on 2048 waves), so waves loop over tiles
// instead of one-tile workgroups whose last round would run nearly empty; the table is staged once per workgroup.
template <int NW, bool PERSIST = true>
__device__ __forceinline__ int64_t child_tile(int i, int wave, int64_t ntiles) {
    if constexpr (!PERSIST) {                                  // one tile per wave (levels with many more tiles than resident waves)
        const int64_t tile = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * NW + wave;
        return (i == 0 && tile < ntiles) ? tile : -1;
    }
    const int64_t S = (ntiles + 7) >> 3;
    const int64_t l = (blockIdx.x >> 3) + (int64_t)(gridDim.x >> 3) * (wave + NW * i);
    const int64_t tile = (blockIdx.x & 7) * S + l;
    return (l < S && tile < ntiles) ? tile : -1;
}

// Stage the B-fragment table into LDS (all waves), once per workgroup.
template <int NW>
__device__ __forceinline__ void child_stage_table(const float* __restrict__ table, int table_bytes, unsigned char* lds_raw) {
    float4* tab = (float4*)lds_raw;
    for (int i = threadIdx.x; i < table_bytes / 16; i += NW * 64) tab[i] = ((const float4*)table)[i];
    __syncthreads();
}

// The cells a variant uses, ascending (= every cell for whole tiles; a variant that computes a subset of the children — the z-half
// units of the C = 64 kernels — skips the cells none of its children reaches: they are neither gathered nor read).
struct ChildCells { int n; int c[64]; };
template <class V> constexpr ChildCells child_cells() {
    ChildCells L{};
    for (int c = 0; c < V::NCELLS; ++c) {
        bool used = false;
        for (int t = 0; t < V::T; ++t) used = used || V::active(c, t);
        if (used) L.c[L.n++] = c;
    }
    return L;
}

// The gather + MFMA main loop of one tile of 16 MT parents, shared by every variant: leaves acc[m][t] (t < V::T) for the epilogue.
// MT = 2 (round 4): a wave owns TWO MFMA M tiles (parents p0 .. p0 + 15 and p0 + 16 .. p0 + 31).  Every B fragment read from the LDS
// table, every `s_waitcnt` and every address computation of a cell serves both; the per-tile prologue (map loads and their latency, the
// first D cells' gather latency) and the epilogue's hand-offs are paid once per 32 parents.  A ring slot holds the cell's rows of both M
// tiles (M tile m at + m NB KB).  Per output element nothing changes: same products, same order.
template <class V, int D, int MT>
__device__ __forceinline__ void child_tile_mainloop_mt(const int32_t* __restrict__ pnbr, int64_t n_p, int64_t p0,
                                                       const __amdgpu_buffer_rsrc_t& rs_in, int in_ld, const unsigned char* lds_raw,
                                                       float4* ring, f32x4 (&acc)[MT][V::T] CHILD_DBG_PARAM) {
    constexpr int NB = V::NB, T = V::T, KS = V::KS;
    static_assert(MT == 1 || MT == 2, "one or two M tiles per wave");
    static_assert((D & (D - 1)) == 0, "ring depth must be a power of two");
    static_assert((D - 1) * NB * MT < 64, "vmcnt is 6 bits");
    CHILD_T(t_tile0);
    const int lane = threadIdx.x & 63;
    const int mi = lane & 15, mq = lane >> 4;
    const int dma_r = lane >> 2;                               // tile row (parent) this lane fetches for
    bool row_ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) row_ok[m] = p0 + 16 * m + dma_r < n_p;
    // byte offset of each neighbour parent's first child row; absent -> a value no in-row offset can bring back into range (the
    // entry point checks the tensor is smaller than ABSENT), so the per-cell address is ONE add and needs no select
    constexpr unsigned ABSENT = 0xF0000000u;
    unsigned rowb[MT][V::NMAP];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int kp = 0; kp < V::NMAP; ++kp) rowb[m][kp] = (unsigned)pnbr[(int64_t)kp * n_p + (row_ok[m] ? p0 + 16 * m + dma_r : 0)];
    const int f_a = (0x78 >> (2 * (mi >> 2))) & 3;             // read-side swizzle of the A image (see conv.hip v2)
    const int dma_chunk = (lane & 3) ^ ((0x78 >> (2 * ((dma_r >> 2) & 3))) & 3);
    const bool chunk_ok = dma_chunk < V::ROWCHUNKS;            // rows narrower than 64 bytes: the other lanes fetch nothing (zeros)
    const unsigned row_bytes = (unsigned)in_ld * 4u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // map entries loaded (and the previous tile's stores retired): vmcnt now counts DMAs only
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int kp = 0; kp < V::NMAP; ++kp)
            rowb[m][kp] = (row_ok[m] && (int)rowb[m][kp] >= 0) ? rowb[m][kp] * ((unsigned)V::ROW_MUL * row_bytes) : ABSENT;
    constexpr bool T2 = child_t2_layout<V>::value;             // (PassB<16> behind the quad-block pass A: chunks of a row are 64 bytes apart)
    const unsigned lane_off = chunk_ok ? (unsigned)dma_chunk * (T2 ? 64u : 16u) : ABSENT;
    CHILD_T(t_loop0);

    constexpr ChildCells CL = child_cells<V>();
    constexpr int NC = CL.n;                                   // cells of this variant; position i in the list uses ring slot i mod D
    auto issue = [&](auto ii) {
        constexpr int i = decltype(ii)::value, c = CL.c[i];
        float4* dst = ring + (i & (D - 1)) * (MT * NB * 64);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            unsigned voff = rowb[m][V::kp(c)] + (T2 ? (unsigned)((V::child(c) >> 2) * 128 + (V::child(c) & 3) * 16) : (unsigned)V::child(c) * row_bytes) + lane_off +
                            (unsigned)V::byte_off(c);
            if constexpr (V::ROWCHUNKS < 4) voff = chunk_ok ? voff : 0xFFFFFFF0u;   // (ABSENT + ABSENT would wrap)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + (m * NB + cb) * 64), 16, (int)(voff + cb * 64), 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    };

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned tab_lane;
    if constexpr (V::HALF) tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)(lds_raw + V::lane_table_off(mi, mq));
    else tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + lane * KS);
    // A operand straight in MFMA layout: lane (mi, mq) needs channel 4 jj + mq of row mi for K-step jj; chunk jj of row mi sits at
    // slot position jj ^ f(mi >> 2) of the (source-swizzled) image, so four ds_read_b32 — conflict-free: bank = 16 (mi & 3) +
    // 4 (jj ^ f(mi >> 2)) + mq covers all 64 banks — replace the ds_read_b128 + 4x4 lane transpose (4 permlane swaps + moves per
    // block and cell): the VALU issue slots those took are what the MFMA pipe was waiting on (2.4-2.8 VALU per MFMA before).
    unsigned a_addr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) a_addr[jj] = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + (mi * 4 + (jj ^ f_a)) * 4 + mq);
    constexpr unsigned ksteps_used = [] { unsigned m = 0; for (int t = 0; t < T; ++t) for (int j = 0; j < KS; ++j) m |= 1u << (V::kfirst(t) + j); return m; }();

    static_assert(NB * T <= 32, "all B fragments of a cell are held in registers at once");
    // (A two-deep register pipeline — LDS reads of cell c+1 issued before the MFMAs of cell c — was measured and dropped on these
    // variants: no gain, 8-30 more registers; with 3-4 waves per SIMD the other waves already cover the LDS latency.)
    static_for<0, (D < NC ? D : NC)>(issue);
    static_for<0, NC>([&](auto ii) {
        constexpr int i = decltype(ii)::value, c = CL.c[i];
        constexpr int younger = (NC - 1 - i) < (D - 1) ? (NC - 1 - i) : (D - 1);       // cells issued after this one that may stay in flight
        wait_vmcnt<younger * NB * MT>();
        float a[MT][NB][4];
        BFrag<KS> b[NB][T];
        static_for<0, NB>([&](auto icb) {                                      // all LDS reads of the cell, one wait
            constexpr int cb = decltype(icb)::value;
            static_for<0, MT>([&](auto im) {
                constexpr int m = decltype(im)::value;
                static_for<0, 4>([&](auto ij) {
                    constexpr int jj = decltype(ij)::value;
                    if constexpr ((ksteps_used >> jj) & 1)
                        a[m][cb][jj] = lds_ld32_off<(((i & (D - 1)) * MT + m) * NB + cb) * 1024>(a_addr[jj]);
                });
            });
            static_for<0, T>([&](auto it) {
                constexpr int t = decltype(it)::value;
                if constexpr (V::active(c, t) && V::uses_block(t, cb)) {
                    constexpr int off = V::frag_off(c, t, cb);                                   // byte offset of the fragment in the table
                    if constexpr (off < 65536) b[cb][t].template load<off>(tab_lane);
                    else b[cb][t].template load<off - 65536>(tab_lane + 65536);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<0, NB>([&](auto icb) {
            constexpr int cb = decltype(icb)::value;
            static_for<0, MT>([&](auto im) {
                static_for<0, 4>([&](auto ij) {
                    constexpr int jj = decltype(ij)::value;
                    if constexpr ((ksteps_used >> jj) & 1) lds_tie(a[decltype(im)::value][cb][jj]);
                });
            });
            static_for<0, T>([&](auto it) {
                constexpr int t = decltype(it)::value;
                if constexpr (V::active(c, t) && V::uses_block(t, cb)) b[cb][t].tie();
            });
        });
        if constexpr (i + D < NC) issue(std::integral_constant<int, i + D>{});    // refill the ring slot this cell was read from
        static_for<0, NB>([&](auto icb) {
            constexpr int cb = decltype(icb)::value;
            static_for<0, 4>([&](auto ij) {
                constexpr int jj = decltype(ij)::value;
                static_for<0, T>([&](auto it) {
                    constexpr int t = decltype(it)::value;
                    if constexpr (V::active(c, t) && V::uses_block(t, cb) && jj >= V::kfirst(t) && jj < V::kfirst(t) + KS) {
                        static_for<0, MT>([&](auto im) {                       // both M tiles against the one B fragment, back to back
                            constexpr int m = decltype(im)::value;
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][cb][jj], b[cb][t].get(jj - V::kfirst(t)), acc[m][t], 0, 0, 0);
                        });
                    }
                });
            });
        });
    });
#ifdef PCGC_CHILD_TIMING
    asm volatile("s_nop 0" : "+v"(acc[0][0]));                  // (keeps the stamp behind the last MFMA's issue)
    CHILD_T(t_loop1);
    CHILD_TADD(0, t_tile0, t_loop0);
    CHILD_TADD(1, t_loop0, t_loop1);
    CHILD_TADD(3, 0ull, 1ull);
#endif
}

template <class V, int D>
__device__ __forceinline__ void child_tile_mainloop(const int32_t* __restrict__ pnbr, int64_t n_p, int64_t p0,
                                                    const __amdgpu_buffer_rsrc_t& rs_in, int in_ld, const unsigned char* lds_raw,
                                                    float4* ring, f32x4 (&acc)[V::T] CHILD_DBG_PARAM) {
    child_tile_mainloop_mt<V, D, 1>(pnbr, n_p, p0, rs_in, in_ld, lds_raw, ring, reinterpret_cast<f32x4 (&)[1][V::T]>(acc) CHILD_DBG_ARG);
}

// Lanes of one wave exchange data through LDS in the epilogues (one group of lanes writes, all lanes read).  The hardware runs a wave's
// LDS operations in order, but the compiler reasons per thread: this fence + wave barrier tells it other lanes' stores become visible
// here (no instruction is emitted beyond the waits the reordering restriction implies).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Epilogue staging.  The accumulators hold the tile in MFMA layout (lane = column, 4 rows per lane); written from there,
// a row reaches memory as 4-byte pieces, 16 lanes to a 64-byte segment — measured at 47 % (plain conv) to 66 % (pass B) of a
// tile's time.  A tile's 128 output rows are contiguous in memory, so the values go through a per-wave LDS scratch (the gather
// ring, idle by then) in row-major order, CH rows at a time, and leave as 16-byte-per-lane stores: whole rows, fully coalesced;
// residual rows are read the same way.  Arithmetic order per element is unchanged: (acc + bias) [+ residual] [relu].
// `half` (0 / 1, or -1 = all rows): only the rows of the children with that z bit (rows 4 half .. 4 half + 3 of every parent's eight).
// Round 4: the trip count is static (ROWS x W / 4 sixteen-byte pieces over 64 lanes) and the residual rows are REQUESTED before anything
// else — `child_flush_prefetch` at the start of a chunk, ahead of the staging writes and the hand-off — so that one memory latency is paid
// per chunk instead of one per piece (the loop form waited `vmcnt(0)` for every residual piece, and with it for the previous piece's
// store: 44 k of pass B's 86 k cycles per tile and wave, tools/child_phase_times.py).
template <int W, int ROWS>
struct ChildResidual { float4 x[(ROWS * (W / 4) + 63) / 64]; };
template <int W, int ROWS>
__device__ __forceinline__ void child_flush_prefetch(ChildResidual<W, ROWS>& rr, int64_t row0, int64_t rows_total, const float* __restrict__ res,
                                                     int res_ld, int lane, int half = -1) {
    constexpr int C4 = W / 4, N = (ROWS * C4 + 63) / 64;
    if (!res) return;                                          // (rr is never read then)
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int i = lane + 64 * it, lr = i / C4, c4 = i % C4;
        const int64_t row = row0 + lr;
        const bool ok = i < ROWS * C4 && row < rows_total && !(half >= 0 && ((lr >> 2) & 1) != half);
        rr.x[it] = ok ? *(const float4*)(res + row * res_ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// (rr by reference + a flag: a pointer that may be null put the struct on the stack — scratch memory — instead of in registers)
template <int W, int ROWS>
__device__ __forceinline__ void child_flush(const float* scratch, const ChildResidual<W, ROWS>& rr, bool has_res, int64_t row0, int64_t rows_total,
                                            float* __restrict__ out, int out_ld, int relu, int lane, int half = -1) {
    constexpr int C4 = W / 4, N = (ROWS * C4 + 63) / 64;
    ChildResidual<W, ROWS> xr = rr;
    if (has_res) {
        // ONE wait for every residual piece, here, before the first store is issued.  (vmcnt retires in order and counts stores too: left to
        // the compiler, each piece's use inside the loop below waited `vmcnt(0)` — i.e. also for the PREVIOUS piece's store to be
        // acknowledged by memory, four round trips per chunk.  The empty asm makes the registers' readiness a fact the compiler knows.)
#pragma unroll
        for (int it = 0; it < N; ++it) asm volatile("" : "+v"(xr.x[it].x), "+v"(xr.x[it].y), "+v"(xr.x[it].z), "+v"(xr.x[it].w));
    }
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int i = lane + 64 * it, lr = i / C4, c4 = i % C4;
        const int64_t row = row0 + lr;
        if (i >= ROWS * C4 || row >= rows_total || (half >= 0 && ((lr >> 2) & 1) != half)) continue;
        float4 v = ((const float4*)scratch)[i];
        if (has_res) {
            const float4 x = xr.x[it];
            v.x = v.x + x.x; v.y = v.y + x.y; v.z = v.z + x.z; v.w = v.w + x.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        *(float4*)(out + row * out_ld + 4 * c4) = v;
    }
}
template <int W, int ROWS>
__device__ __forceinline__ void child_flush(const float* scratch, int64_t row0, int64_t rows_total, float* __restrict__ out, int out_ld,
                                            int relu, int lane, int half = -1) {
    constexpr int C4 = W / 4, N = (ROWS * C4 + 63) / 64;
#pragma unroll
    for (int it = 0; it < N; ++it) {
        const int i = lane + 64 * it, lr = i / C4, c4 = i % C4;
        const int64_t row = row0 + lr;
        if (i >= ROWS * C4 || row >= rows_total || (half >= 0 && ((lr >> 2) & 1) != half)) continue;
        float4 v = ((const float4*)scratch)[i];
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        *(float4*)(out + row * out_ld + 4 * c4) = v;
    }
}
// (runtime row count, no residual: the plain-rows kernels of rows_irn.hip whose chunk is 16 rows)
template <int W>
__device__ __forceinline__ void child_flush(const float* scratch, int rows, int64_t row0, int64_t rows_total, float* __restrict__ out,
                                            int out_ld, const float* __restrict__ res, int res_ld, int relu, int lane, int half = -1) {
    constexpr int C4 = W / 4;
    for (int i = lane; i < rows * C4; i += 64) {
        const int lr = i / C4, c4 = i % C4;
        const int64_t row = row0 + lr;
        if (row >= rows_total || (half >= 0 && ((lr >> 2) & 1) != half)) continue;
        float4 v = ((const float4*)scratch)[i];
        if (res) {
            const float4 x = *(const float4*)(res + row * res_ld + 4 * c4);
            v.x = v.x + x.x; v.y = v.y + x.y; v.z = v.z + x.z; v.w = v.w + x.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        *(float4*)(out + row * out_ld + 4 * c4) = v;
    }
}

// rows of the gathered tensor: the children of the n_p parents, or the level itself (a kernel whose input is ANOTHER level redefines it)
#define CHILD_IN_ROWS(V, n_p) (V::ROW_MUL * (n_p))
#define CHILD_KERNEL_PROLOGUE(V, NW, D, RING_FLOAT4)                                                                          \
    CHILD_DBG_DECL                                                                                                            \
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];                                                   \
    const int lane = threadIdx.x & 63, mi = lane & 15, mq = lane >> 4;                                                        \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                                        \
    float4* ring = (float4*)(lds_raw + table_bytes) + wave * (RING_FLOAT4);                                                   \
    child_stage_table<NW>(table, table_bytes, lds_raw);                                                                        \
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(CHILD_IN_ROWS(V, n_p) * in_ld * 4), 0x00020000); \
    const int64_t ntiles = (n_p + 15) >> 4;

template <class T_> struct child_type_tag { using type = T_; };
// plain conv:  acc[t][r] = out[8 (p0 + 4 mq + r) + j][16 n + mi],  t = j * NT + n.   SPLIT: half units (see k_child_irn_a)
// RES: the launch has residual rows (the no-residual instantiation — every conv of the decoder — carries no residual registers at all:
// with a runtime flag their 16-32 VGPRs cost the 16 -> 16 kernel its third wave per SIMD, 241 -> 268 us)
template <int NB, int NT, int NW, int D, bool SPLIT = false, int MT = 1, bool RES = false>
__global__ void __launch_bounds__(NW * 64)
k_child_conv(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
             const float* __restrict__ table, int table_bytes, ChildEpi ep) {
    using V = PlainConv<NB, NT>;
    CHILD_KERNEL_PROLOGUE(V, NW, D, D * NB * 64 * MT)
    float bvs[NT];                                             // (loaded once: inside the tile loop every use was a global load + vmcnt(0))
#pragma unroll
    for (int n = 0; n < NT; ++n) bvs[n] = ep.bias ? ep.bias[16 * n + mi] : 0.0f;
    auto unit = [&](auto tag, const int64_t pu) {
        using VV = typename decltype(tag)::type;
        constexpr int HZ = VV::Z_HALF;
        f32x4 acc[MT][VV::T];
        child_tile_mainloop_mt<VV, D, MT>(pnbr, n_p, pu, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        // staged epilogue: CH rows (CH / 8 parents) at a time through the ring's LDS
        constexpr int W = 16 * NT, CH = (64 * W * 4 <= D * NB * 1024 * MT) ? 64 : 32, PPC = CH / 8, MQC = PPC / 4;   // MQC: lane quarters per chunk
        float* scratch = (float*)ring;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t p0 = pu + 16 * m;
            if (m > 0 && p0 >= n_p) break;
#pragma unroll
            for (int h = 0; h < 128 / CH; ++h) {
                ChildResidual<W, RES ? CH : 1> rr;
                if constexpr (RES) child_flush_prefetch<W, CH>(rr, 8 * p0 + h * CH, 8 * n_p, ep.res, ep.res_ld, lane, HZ);
                if (mq / MQC == h) {
#pragma unroll
                    for (int t = 0; t < VV::T; ++t) {
                        const int j = t / NT, n = t % NT;
                        if (HZ >= 0 && ((j >> 2) & 1) != HZ) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[m][t][r];
                            if (ep.bias) v = v + bvs[n];
                            scratch[(8 * (4 * (mq % MQC) + r) + j) * W + 16 * n + mi] = v;
                        }
                    }
                }
                wave_lds_sync();
                if constexpr (RES) child_flush<W, CH>(scratch, rr, true, 8 * p0 + h * CH, 8 * n_p, ep.out, ep.out_ld, ep.relu, lane, HZ);
                else child_flush<W, CH>(scratch, 8 * p0 + h * CH, 8 * n_p, ep.out, ep.out_ld, ep.relu, lane, HZ);
                wave_lds_sync();
            }
        }
    };
    const int64_t nunits = (SPLIT ? 2 : 1) * ((n_p + 16 * MT - 1) / (16 * MT));
    for (int i = 0;; ++i) {
        const int64_t u = child_tile<NW>(i, wave, nunits);
        if (u < 0) break;
        CHILD_T(t_it0);
        if constexpr (SPLIT) {
            if (u & 1) unit(child_type_tag<PlainConv<NB, NT, 1>>{}, (u >> 1) * (16 * MT));
            else unit(child_type_tag<PlainConv<NB, NT, 0>>{}, (u >> 1) * (16 * MT));
        } else {
            unit(child_type_tag<V>{}, u * (16 * MT));
        }
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// classification head C -> 1:  acc[0][r] column j = out[8 (p0 + 4 mq + r) + j]
template <int NB, int NW, int D, int MT = 1>
__global__ void __launch_bounds__(NW * 64)
k_child_cls(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
            const float* __restrict__ table, int table_bytes, ChildEpi ep) {
    using V = ClsHead<NB>;
    CHILD_KERNEL_PROLOGUE(V, NW, D, D * NB * 64 * MT)
    const float bv = ep.bias ? ep.bias[0] : 0.0f;
    const int64_t nunits = (n_p + 16 * MT - 1) / (16 * MT);
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, nunits);
        if (tile < 0) break;
        const int64_t pu = tile * (16 * MT);
        CHILD_T(t_it0);
        f32x4 acc[MT][1];
        child_tile_mainloop_mt<V, D, MT>(pnbr, n_p, pu, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        if (mi < 8) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t p = pu + 16 * m + 4 * mq + r;
                    if (p >= n_p) continue;
                    float v = acc[m][0][r];
                    if (ep.bias) v = v + bv;
                    ep.out[(8 * p + mi) * ep.out_ld] = v;
                }
            }
        }
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}



// pass A:  t[row][0:Q] = relu(conv0_0 + b00), t[row][Q:2Q] = relu(conv1_0 + b10)
// SPLIT (C = 64): the level is ~1.14 tiles per SIMD, and a tile is 48 us of one SIMD's MFMA pipe — whole tiles quantise to TWO tile times
// per launch.  Half units (the four children with z bit 0 / 1 of 16 parents: half the MFMAs, 48 of the 64 cells) quantise to three
// half-tile times.  Both halves are separate instantiations of the statically unrolled body; a wave picks one per unit.
template <int C, int NW, int D, bool SPLIT = false, int MT = 1>
__global__ void __launch_bounds__(NW * 64)
k_child_irn_a(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
              const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    using V = PassA<C>;
    constexpr int Q = V::Q, CPT = V::CPT, TH = V::TH;
    static_assert(!SPLIT || CPT <= 2, "half units: tiles of one or two children");
    CHILD_KERNEL_PROLOGUE(V, NW, D, D * V::NB * 64 * MT)
    const int co = mi % Q, sub = mi / Q;                       // column -> (child within the tile, output channel)
    const float b00 = ep.b0[co], b10 = ep.b1[co];
    auto unit = [&](auto tag, const int64_t pu) {
        using VV = typename decltype(tag)::type;
        constexpr int HZ = VV::Z_HALF;
        f32x4 acc[MT][VV::T];
        child_tile_mainloop_mt<VV, D, MT>(pnbr, n_p, pu, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        constexpr int W = 2 * Q, CH = (64 * W * 4 <= D * V::NB * 1024 * MT) ? 64 : 32, MQC = CH / 32;
        float* scratch = (float*)ring;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t p0 = pu + 16 * m;
            if (m > 0 && p0 >= n_p) break;
#pragma unroll
            for (int h = 0; h < 128 / CH; ++h) {
                if (mq / MQC == h) {
#pragma unroll
                    for (int t = 0; t < TH; ++t) {
                        if (HZ >= 0 && (((t * CPT) >> 2) & 1) != HZ) continue;     // (tile t holds children t CPT ...)
                        const int j = t * CPT + sub;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* y = scratch + (8 * (4 * (mq % MQC) + r) + j) * W;
                            y[co] = fmaxf(acc[m][t][r] + b00, 0.0f);
                            y[Q + co] = fmaxf(acc[m][TH + t][r] + b10, 0.0f);
                        }
                    }
                }
                wave_lds_sync();
                child_flush<W, CH>(scratch, 8 * p0 + h * CH, 8 * n_p, ep.out, W, 0, lane, HZ);
                wave_lds_sync();
            }
        }
    };
    const int64_t nunits = (SPLIT ? 2 : 1) * ((n_p + 16 * MT - 1) / (16 * MT));
    for (int i = 0;; ++i) {
        const int64_t u = child_tile<NW>(i, wave, nunits);
        if (u < 0) break;
        CHILD_T(t_it0);
        if constexpr (SPLIT) {
            if (u & 1) unit(child_type_tag<PassA<C, 1>>{}, (u >> 1) * (16 * MT));
            else unit(child_type_tag<PassA<C, 0>>{}, (u >> 1) * (16 * MT));
        } else {
            unit(child_type_tag<V>{}, u * (16 * MT));
        }
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// pass B:  out[row][0:2Q]  = (conv0_1(t[:, :Q]) + b01) + x[row][0:2Q]
//          out[row][2Q:4Q] = (conv1_2(relu(conv1_1(t[:, Q:]) + b11)) + b12) + x[row][2Q:4Q]
// conv1_2 (k1, Q -> 2Q) is a second, tiny MFMA product: u = relu(conv1_1 + b11) goes through a per-wave LDS scratch (the gather
// ring, idle by then) from the accumulator layout (lane = column) into A fragments (lane = row), 16 output rows per product.
template <int C, int NW, int D, bool SPLIT = false, int MT = 1, bool T2 = false>
__global__ void __launch_bounds__(NW * 64)
k_child_irn_b(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in /* t */, int in_ld,
              const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    using V = PassB<C, -1, T2>;
    constexpr int Q = V::Q, H = 2 * Q, KQ = Q / 4, T0 = V::T0, T1 = V::T1, CPT0 = V::CPT0, CPT1 = V::CPT1;
    // epilogue scratch inside the ring's LDS: us [128 rows][Q] (u = relu(conv1_1)), then the CH-row output staging [CH][C]
    constexpr int CH = ((128 * Q + 64 * C) * 4 <= D * 1024 * MT) ? 64 : 32, MQC = CH / 32;
    constexpr int NEEDF4 = (128 * Q + CH * C) / 4;
    constexpr int RINGF4 = (D * 64 * MT > NEEDF4) ? D * 64 * MT : NEEDF4;
    CHILD_KERNEL_PROLOGUE(V, NW, D, RINGF4)
    float* us = (float*)ring;
    float* stage = us + 128 * Q;
    // conv1_2 B fragment: W12[4 jj + mq][mi]  (columns >= 2Q unused)
    float w12[KQ];
#pragma unroll
    for (int jj = 0; jj < KQ; ++jj) w12[jj] = ((const float*)lds_raw)[V::FRAG_W12 * frag_floats<V>() + lane * KQ + jj];
    // one unit: a whole tile, or (SPLIT) the four children with one z bit of its 16 parents — see k_child_irn_a.  In a half unit the other
    // half's rows of the scratch hold stale values: they go through conv1_2 like the rest and are never stored.
    // (per-lane biases, loaded once: inside the tile loop each was a global load + vmcnt(0) per tile)
    const int c1 = mi % Q, sub1 = mi / Q, c0 = mi % H, sub0 = mi / H;
    const float b11 = ep.b1[c1], b01 = ep.b0[c0], b12 = mi < H ? ep.b2[mi] : 0.0f;
    auto unit = [&](auto tag, const int64_t pu) {
        using VV = typename decltype(tag)::type;
        constexpr int HZ = VV::Z_HALF;
        f32x4 acc[MT][VV::T];
        child_tile_mainloop_mt<VV, D, MT>(pnbr, n_p, pu, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
        const int64_t p0 = pu + 16 * m;
        if (m > 0 && p0 >= n_p) break;
        ChildResidual<C, CH> rr;                                // the first chunk's residual rows: requested before the hand-offs below
        child_flush_prefetch<C, CH>(rr, 8 * p0, 8 * n_p, ep.x, ep.x_ld, lane, HZ);
        // ---- u = relu(conv1_1 + b11) -> scratch us [local row = 8 (4 mq + r) + child][Q]
        {
#pragma unroll
            for (int u = 0; u < T1; ++u) {
                if (HZ >= 0 && VV::tile_z(T0 + u) != HZ) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) us[(8 * (4 * mq + r) + u * CPT1 + sub1) * Q + c1] = fmaxf(acc[m][T0 + u][r] + b11, 0.0f);
            }
        }
        wave_lds_sync();
        // ---- CH rows at a time: [conv0_1 + b01 | conv1_2(u) + b12] staged row-major, then flushed with the residual x
        {
#pragma unroll
            for (int h = 0; h < 128 / CH; ++h) {
                if (h > 0) child_flush_prefetch<C, CH>(rr, 8 * p0 + h * CH, 8 * n_p, ep.x, ep.x_ld, lane, HZ);
                if (mq / MQC == h) {
#pragma unroll
                    for (int t = 0; t < T0; ++t) {
                        if (HZ >= 0 && VV::tile_z(t) != HZ) continue;
                        const int j = t * CPT0 + sub0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[(8 * (4 * (mq % MQC) + r) + j) * C + c0] = acc[m][t][r] + b01;
                    }
                }
                // conv1_2 on u, 16 rows per MFMA product: every A operand is read before any result is written (the compiler cannot move an
                // LDS read across an LDS write of the same array: written group by group, each group was read -> wait -> MFMA -> wait -> write)
                float au[CH / 16][KQ];
#pragma unroll
                for (int gg = 0; gg < CH / 16; ++gg)
#pragma unroll
                    for (int jj = 0; jj < KQ; ++jj) au[gg][jj] = us[(16 * (h * (CH / 16) + gg) + mi) * Q + 4 * jj + mq];
                f32x4 d[CH / 16];
#pragma unroll
                for (int gg = 0; gg < CH / 16; ++gg) {
                    d[gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jj = 0; jj < KQ; ++jj) d[gg] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[gg][jj], w12[jj], d[gg], 0, 0, 0);
                }
                if (mi < H) {
#pragma unroll
                    for (int gg = 0; gg < CH / 16; ++gg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[(16 * gg + 4 * mq + r) * C + H + mi] = d[gg][r] + b12;
                }
                wave_lds_sync();
                child_flush<C, CH>(stage, rr, ep.x != nullptr, 8 * p0 + h * CH, 8 * n_p, ep.out, ep.out_ld, 0, lane, HZ);
                wave_lds_sync();
            }
        }
        }
    };
    const int64_t nunits = (SPLIT ? 2 : 1) * ((n_p + 16 * MT - 1) / (16 * MT));
    for (int i = 0;; ++i) {
        const int64_t u = child_tile<NW>(i, wave, nunits);
        if (u < 0) break;
        CHILD_T(t_it0);
        if constexpr (SPLIT) {
            if (u & 1) unit(child_type_tag<PassB<C, 1, T2>>{}, (u >> 1) * (16 * MT));
            else unit(child_type_tag<PassB<C, 0, T2>>{}, (u >> 1) * (16 * MT));
        } else {
            unit(child_type_tag<V>{}, u * (16 * MT));
        }
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// kernels above the default dynamic-LDS limit need the attribute raised once per (kernel, device)
struct ChildLdsGrant { size_t bytes[16] = {0}; };
template <typename K>
int child_lds_limit(K kern, size_t lds, ChildLdsGrant& granted) {
    if (lds > 160 * 1024) { pcgc_set_error("child kernel: %zu bytes of LDS needed", lds); return -2; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& g = granted.bytes[dev & 15];
    if (lds > 48 * 1024 && lds > g) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pcgc_set_error("child kernel: cannot raise the LDS limit to %zu: %s", lds, hipGetErrorString(e)); return -1; }
        g = lds;
    }
    return 0;
}
// persistent grid: as many workgroups as stay resident (LDS-limited, at most 16 waves per CU), a multiple of 8 (one share per XCD).
// n_units: work items of the launch (tiles of 16 MT parents, x 2 for half units)
static unsigned child_grid_units(int64_t n_units, int nw, size_t lds, int max_waves_per_cu = 16) {
    static int cus = 0;
    if (!cus) { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > max_waves_per_cu / nw) per_cu = max_waves_per_cu / nw;
    if (per_cu < 1) per_cu = 1;
    int64_t want = (n_units + nw - 1) / nw;
    int64_t g = (int64_t)cus * per_cu;
    if (g > want) g = want;
    g = (g + 7) / 8 * 8;
    return (unsigned)g;
}
static unsigned child_grid(int64_t n_p, int nw, size_t lds, int units_per_tile = 1) {
    return child_grid_units(((n_p + 15) / 16) * units_per_tile, nw, lds);
}
// KERN, waves per workgroup, ring bytes per wave, epilogue struct, M tiles per wave (MT), units per tile (2 = half units)
#define CHILD_LAUNCH_EX(KERN, NW, RINGBYTES, EP, MT_, UPT)                                                                     \
    do {                                                                                                                       \
        const size_t lds = (size_t)table_bytes + (size_t)(NW) * (RINGBYTES);                                                   \
        auto kern = KERN;                                                                                                      \
        static ChildLdsGrant granted;                                                                                          \
        if (int rc = child_lds_limit(kern, lds, granted)) return rc;                                                           \
        const int64_t units = ((n_p + 16 * (MT_) - 1) / (16 * (MT_))) * (UPT);                                                 \
        hipLaunchKernelGGL(kern, dim3(child_grid_units(units, NW, lds)), dim3((NW) * 64), lds, s, pnbr, n_p, in, in_ld, table, table_bytes, EP); \
        return 0;                                                                                                              \
    } while (0)
#define CHILD_LAUNCH(KERN, NW, RINGBYTES, EP) CHILD_LAUNCH_EX(KERN, NW, RINGBYTES, EP, 1, 1)
#define CHILD_LAUNCH_SPLIT(KERN, NW, RINGBYTES, EP) CHILD_LAUNCH_EX(KERN, NW, RINGBYTES, EP, 1, 2)      // half units: twice the work items per tile

template <int NB, int NT, int NW, int D, int MT = 1>
int launch_child_conv(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                      const ChildEpi& ep, hipStream_t s) {
    CHILD_LAUNCH_EX((k_child_conv<NB, NT, NW, D, false, MT, false>), NW, D * NB * 1024 * MT, ep, MT, 1);
}
template <int NB, int NT, int NW, int D>
int launch_child_conv_split(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                            const ChildEpi& ep, hipStream_t s) {
    CHILD_LAUNCH_EX((k_child_conv<NB, NT, NW, D, true, 1, false>), NW, D * NB * 1024, ep, 1, 2);
}
// WPC: waves per CU the persistent grid may count on
template <int NB, int NW, int D, int MT = 1, int WPC = 16>
int launch_child_cls(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                     const ChildEpi& ep, hipStream_t s) {
    const size_t lds = (size_t)table_bytes + (size_t)NW * (D * NB * 1024 * MT);
    auto kern = k_child_cls<NB, NW, D, MT>;
    static ChildLdsGrant granted;
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    const int64_t units = (n_p + 16 * MT - 1) / (16 * MT);
    hipLaunchKernelGGL(kern, dim3(child_grid_units(units, NW, lds, WPC)), dim3(NW * 64), lds, s, pnbr, n_p, in, in_ld, table, table_bytes, ep);
    return 0;
}
template <int C, int NW, int D, int MT = 1>
int launch_child_irn_a(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                       const IrnEpi& ep, hipStream_t s) {
    CHILD_LAUNCH_EX((k_child_irn_a<C, NW, D, false, MT>), NW, D * (C / 16) * 1024 * MT, ep, MT, 1);
}
template <int C, int NW, int D>
int launch_child_irn_a_split(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                             const IrnEpi& ep, hipStream_t s) {
    CHILD_LAUNCH_SPLIT((k_child_irn_a<C, NW, D, true>), NW, D * (C / 16) * 1024, ep);
}
template <int C, int NW, int D, int MT = 1, bool T2 = false>
int launch_child_irn_b(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                       const IrnEpi& ep, hipStream_t s) {
    constexpr int Q = C / 4, CH = ((128 * Q + 64 * C) * 4 <= D * 1024 * MT) ? 64 : 32;
    constexpr int need = (128 * Q + CH * C) * 4;
    constexpr int ringb = (D * 1024 * MT > need) ? D * 1024 * MT : need;
    CHILD_LAUNCH_EX((k_child_irn_b<C, NW, D, false, MT, T2>), NW, ringb, ep, MT, 1);
}
template <int C, int NW, int D>
int launch_child_irn_b_split(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                             const IrnEpi& ep, hipStream_t s) {
    constexpr int Q = C / 4, CH = ((128 * Q + 64 * C) * 4 <= D * 1024) ? 64 : 32;
    constexpr int need = (128 * Q + CH * C) * 4;
    constexpr int ringb = (D * 1024 > need) ? D * 1024 : need;
    CHILD_LAUNCH_SPLIT((k_child_irn_b<C, NW, D, true>), NW, ringb, ep);
}

}  // namespace

#define CHILD_COMMON_CHECKS(ROWS_LD)                                                                                           \
    PCGC_REQUIRE(parent_nbr && in && table, "null argument");                                                                  \
    PCGC_REQUIRE(((ROWS_LD) & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");                 \
    PCGC_REQUIRE(8 * n_parent * (int64_t)(ROWS_LD) * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets"); \
    PCGC_REQUIRE(table_bytes % 16 == 0, "table size");                                                                         \
    if (n_parent == 0) return 0;
