// Children-level kernels in QUAD-BLOCK form (round 5): the narrow layers of the stride-1 decoder level (C = 16: InceptionResNet
// pass A, k3 16 -> 4 + k1 16 -> 4; autoencoder.py:7-57 on the level of :209-237) on `v_mfma_f32_4x4x1_16b_f32`.
//
// Why.  The packed-N kernels of child_kernels.h put (child, output channel) pairs into the 16 columns of a 16x16x4 MFMA tile; a halo
// cell is reached by 4 / 2 / 1 of a tile's four children, so 44 % of the columns multiply zeros — and the launch is bound by the
// matrix pipe (6.65 M issued MFMAs = 13.6 GFLOP at the ~120 TFLOP/s the fp32 pipe sustains at its loaded clock of ~1.95 GHz = the
// 115 us it takes; profiles/r05_child_q4.md).  The 4x4x1 instruction is sixteen independent 4 x 4 outer products per issue, each block
// with its OWN A column and B row, at the same 64 FLOP/clk/SIMD: a block = 4 parents x the 4 output channels of ONE (cell, child)
// pair, so only pairs that exist are issued — no structural zero column at all (216 pairs x 16 channels per parent instead of
// 96 tiles x 16 columns x 16 channels / 4): 1.86x fewer pipe cycles.
//
// Geometry.  One wave = MT M tiles of 64 parents (lane l of M tile m = parent p0 + 64 m + l; block = 4 consecutive parents).  Per
// halo cell (ascending, as in child_kernels.h) the cell's rows of the 64 MT parents are gathered by 4 MT `buffer_load_dwordx4 ... lds`
// (4 adjacent lanes per 64-byte row, absent neighbours out of range -> zeros) into a two-slot ring; lane l reads ITS row (16 channels
// = four conflict-free ds_read_b128, chunks XOR-swizzled on the source side by (row >> 2) & 3) and supplies channel ci as the A
// operand of instruction ci.  The B operand of a (cell, child) pair is W[k(cell, child)][ci][lane & 3]: 16 values per lane = four
// broadcast ds_read_b128 from a [k][co][ci] table (7 KB), shared by the MT M tiles.  MFMA ci of pair (c, j) accumulates into
// acc[m][j] (4 registers: parents 4 b + {0..3}, column co = lane & 3) — per output element the products arrive in ascending cell =
// ascending kernel offset k and ascending channel: the canonical chain of DESIGN.md section 3, one fma per product (the instruction is
// a single-rounding fma: tools/ubench/mfma4x4_probe.hip).  Bit-identical to k_child_irn_a<16> and the oracle (tests).
//
// Schedule.  The 224 groups (216 pairs + the 8 conv1_0 groups, 16 MT MFMAs each) of a tile are straight-line code (the instruction
// cache streams a body executed once per tile at full rate: tools/ubench/q4_probe.hip).  B of group n + 1 is requested before the MFMAs of
// group n (two register sets); the A rows of cell c + 1 are requested INSIDE the last group of cell c, quarter by quarter, each right
// behind the MFMAs that consumed the registers it overwrites; every wait is a counted lgkmcnt / vmcnt (both queues return in order)
// computed from a compile-time simulation of the instruction order (q4_sched).  Finished outputs leave straight from the accumulators
// as soon as their last product is in (conv1_0 after its one group, child j after cell j + (2, 2, 2)): the stores of a tile are spread
// over its second half instead of forming a tail, and the 16 MT KB of LDS a staged epilogue would need stay with the gather ring.
#pragma once
#include "child_kernels.h"

namespace {

struct Q4Group {
    int cell, child, kind;        // kind 0: k3 pair (cell, child) -> acc0[child];  1: the k1 conv (conv1_0) of `child` at its own cell -> acc1
    int k;                        // fragment index in the table: kernel offset (kind 0) or 27 (kind 1)
    bool first, last;             // first / last group of its cell
    bool fin;                     // kind 0: the child's last product (k = 26): its outputs are final
};
struct Q4Sched {
    int n;
    Q4Group g[232];
    int vm_wait[232];             // last group of cell c (c < 63): vmcnt that guarantees cell c + 1's rows have landed
    int pend[232];                // store instructions issued at the end of group n (a flush unit completed by group n - 1: see q4_flush)
};
constexpr int Q4_PASS_A = 0, Q4_CLS = 1;
// KIND = Q4_CLS (k3 conv 16 -> 1, the classification head): a group = (cell, z half h): the block's four columns are the children
// 4 h + {0..3}, column s multiplying by kernel[k(cell, 4 h + s)][ci] — zero where that child does not reach the cell (host-built table, one
// fragment per group) — into acc[h]; `child` = h, `k` = the group's number (= its fragment).
template <int KIND, int MT, int D = 2>
constexpr Q4Sched q4_sched() {
    Q4Sched S{};
    for (int c = 0; c < 64; ++c) {
        const int n0 = S.n;
        if (KIND == Q4_CLS) {
            for (int h = 0; h < 2; ++h)
                if (in02(cz_of(c) - h)) { S.g[S.n] = Q4Group{c, h, 0, S.n, false, false, false}; S.n++; }
        } else {
            for (int j = 0; j < 8; ++j) {
                if (!((cell_reach(c) >> j) & 1)) continue;
                const int k = cell_k(c, j);
                S.g[S.n++] = Q4Group{c, j, 0, k, false, false, k == 26};
                if (k == 13) S.g[S.n++] = Q4Group{c, j, 1, 27, false, false, false};
            }
        }
        S.g[n0].first = true;
        S.g[S.n - 1].last = true;
    }
    // VMEM instruction order: [gathers 0 .. D - 1] then per group: (last: wait for cell + 1) ... (first: gather cell + D) (stores of group n - 1)
    int ops = 0, pending = 0;
    int gather_end[64 + 16] = {};
    for (int c = 0; c < D; ++c) { ops += 4 * MT; gather_end[c] = ops; }
    for (int n = 0; n < S.n; ++n) {
        const Q4Group& G = S.g[n];
        if (G.last && G.cell + 1 < 64) S.vm_wait[n] = ops - gather_end[G.cell + 1];
        if (G.first && G.cell + D < 64) { ops += 4 * MT; gather_end[G.cell + D] = ops; }
        S.pend[n] = pending;
        ops += pending;
        pending = (KIND == Q4_PASS_A && ((G.kind == 1) || G.fin) && (G.child & 3) == 3) ? 4 * MT : 0;      // a flush unit: 4 parents-steps x MT stores
    }
    return S;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void q4_store(const f32x4& v, const __amdgpu_buffer_rsrc_t& rs, unsigned voff) {
    u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
#ifndef Q4_STORE_AUX
#define Q4_STORE_AUX 0
#endif
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)voff, 0, Q4_STORE_AUX);
}

// InceptionResNet pass A at C = 16:  t[row][0:4] = relu(conv0_0 x + b00), t[row][4:8] = relu(conv1_0 x + b10)   (row = 8 p + j)
// table: [k = 0..26][co = 0..3][ci = 0..15] = W00[k][ci][co], then [co][ci] = W10[ci][co]   (ops.child_q4_tables)
//
// Classification head at C = 16 (KIND = Q4_CLS; autoencoder.py:228-234 conv2_cls):  out[8 p + j] = conv(x)[.., 0] + bias, dense [8 n_p, 1].
// table: one fragment per (cell, z half) group in schedule order: [s = 0..3][ci = 0..15] = kernel[k(cell, 4 h + s)][ci][0] or 0
// (ops.child_q4_cls_table).  The lane holds its parent's eight logits (two accumulators): 32 contiguous bytes per lane, 2 KB per store pair.
template <int KIND, int NW, int MT, int D = 2>
__global__ void __launch_bounds__(NW * 64)
k_child_q4(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
           const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    constexpr int TP = 64 * MT, SLOT_F4 = MT * 256;            // parents per tile; float4 per ring slot (MT x 4 KB)
    constexpr Q4Sched S = q4_sched<KIND, MT, D>();
    constexpr int NACC = KIND == Q4_CLS ? 2 : 8, OUT_ROW_BYTES = KIND == Q4_CLS ? 4 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* ring = (float4*)(lds_raw + table_bytes) + wave * (D * SLOT_F4);
    child_stage_table<NW>(table, table_bytes, lds_raw);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(8 * n_p * in_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)ep.out, 0, (int)(8 * n_p * OUT_ROW_BYTES), 0x00020000);
    const int co = lane & 3;
    float b00[4], b10[4];                                      // (wave-uniform: scalar registers)
#pragma unroll
    for (int r = 0; r < 4; ++r) { b00[r] = KIND == Q4_CLS ? (ep.b0 ? ep.b0[0] : 0.0f) : ep.b0[r]; b10[r] = KIND == Q4_CLS ? 0.0f : ep.b1[r]; }
    const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + co * 16);
    // chunk e of row `lane` sits at slot position e ^ ((row >> 2) & 3): address = a_row ^ (e << 4)  (a_row has the swizzle in bits 4-5)
    const unsigned a_row = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + lane * 16 + (((lane >> 2) & 3) << 2));
    const unsigned row_bytes = (unsigned)in_ld * 4u;
    // gather instruction d of M tile m: lane L fetches for row 16 d + (L >> 2) the chunk that belongs at slot position L & 3
    const unsigned lane_off = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    constexpr unsigned ABSENT = 0xF0000000u;
    const int64_t ntiles = (n_p + TP - 1) / TP;

    for (int it = 0;; ++it) {
#ifdef Q4_EXP_TILE_ORDER
        // (experiment) the waves of a workgroup take CONSECUTIVE tiles of their XCD's slab instead of tiles a workgroup-count apart
        int64_t tile;
        {
            const int64_t S = (ntiles + 7) >> 3;
            const int64_t l = ((int64_t)(blockIdx.x >> 3) + (int64_t)(gridDim.x >> 3) * it) * NW + wave;
            tile = (blockIdx.x & 7) * S + l;
            if (!(l < S && tile < ntiles)) tile = -1;
        }
#else
        const int64_t tile = child_tile<NW>(it, wave, ntiles);
#endif
        if (tile < 0) break;
        const int64_t p_base = tile * TP;
        // the parent map entries of row 16 (L & 3) + (L >> 2) of each M tile: quad L >> 2 holds rows (L >> 2) + 16 {0, 1, 2, 3}, and gather
        // instruction d takes its entry from quad lane d (a quad_perm broadcast folded into the address add)
        unsigned rowb[MT][27];
        bool ok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t rho = p_base + 64 * m + 16 * (lane & 3) + (lane >> 2);
            ok[m] = rho < n_p;
#pragma unroll
            for (int kp = 0; kp < 27; ++kp) rowb[m][kp] = (unsigned)pnbr[(int64_t)kp * n_p + (ok[m] ? rho : 0)];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // map entries here, the previous tile's stores retired: vmcnt counts from zero
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int kp = 0; kp < 27; ++kp)
#ifdef Q4_KO_GATHER
                rowb[m][kp] = ((int)rowb[m][kp] == 0x7FFFFFF1) ? rowb[m][kp] * (8u * row_bytes) : ABSENT;     // (timing experiment: no row is fetched)
#else
                rowb[m][kp] = (ok[m] && (int)rowb[m][kp] >= 0) ? rowb[m][kp] * (8u * row_bytes) : ABSENT;
#endif
        // byte offset of the lane's 16-byte piece in the 256 bytes of the first parent of its quad: t is written in the T2 layout
        // (child_kernels.h PassB) [parent][z half][conv][child & 3][4 channels], so that a quad's store is 64 contiguous bytes; rows past the
        // tensor are dropped by the bounds check
        auto ovq = [&](int m) { return (unsigned)((p_base + 64 * m + (lane & ~3)) * 256 + (lane & 3) * 16); };
#ifdef Q4_KO_STORE
        f32x4 sink = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif

        f32x4 acc0[MT][NACC], c1[MT][KIND == Q4_CLS ? 1 : 4];   // c1 (pass A): conv1_0 of the four children of a z half (accumulators, then outputs)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc0[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 a[MT][4], b[2][4];

        auto gather = [&](auto ic) {
            constexpr int c = decltype(ic)::value, kp = cell_kp(c), ch = cell_child(c), slot = c % D;
            const unsigned cell_off = (unsigned)ch * row_bytes + lane_off;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int rb = (int)rowb[m][kp];
                const unsigned v0 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0x00, 0xF, 0xF, true) + cell_off;
                const unsigned v1 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0x55, 0xF, 0xF, true) + cell_off;
                const unsigned v2 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xAA, 0xF, 0xF, true) + cell_off;
                const unsigned v3 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xFF, 0xF, 0xF, true) + cell_off;
                float4* dst = ring + slot * SLOT_F4 + m * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst), 16, (int)v0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 64), 16, (int)v1, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 128), 16, (int)v2, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 192), 16, (int)v3, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        };
        auto load_a = [&](auto ic, auto ie) {                  // quarter e (channels 4 e .. 4 e + 3) of the lane's row of cell c, every M tile
            constexpr int c = decltype(ic)::value, e = decltype(ie)::value, slot = c % D;
            static_for<0, MT>([&](auto im) {
                constexpr int m = decltype(im)::value;
                a[m][e] = lds_ld128_off<(slot * MT + m) * 4096>(a_row ^ (unsigned)(e << 4));
            });
        };
        auto load_b = [&](auto in_, auto ibuf) {
            constexpr int n = decltype(in_)::value, buf = decltype(ibuf)::value;
            static_for<0, 4>([&](auto ie) {
                constexpr int e = decltype(ie)::value;
                b[buf][e] = lds_ld128_off<S.g[n].k * 256 + e * 16>(tab_lane);
            });
        };
        // Output.  A lane holds, per child, its parent's four channels (16 bytes of a 32-byte row [conv0_0 | conv1_0]); rows of consecutive
        // lanes are 256 bytes apart, and stores of one 16-byte piece per lane cost the launch 35 us (4.2 M partial-line requests next to the
        // gathers: profiles/r05_child_q4.md).  So outputs leave per FLUSH UNIT = (M tile, z half, conv): the four children 4 h + {0..3} of
        // the lane's parent are transposed across the lanes of each quad — by the matrix pipe itself: D[i][k] += V_k(lane i) * [lane == k]
        // puts child k's value of parent 4 b + i into lane k, register i; x * 1 + 0 is exact for the post-ReLU values (>= +0) — so that
        // in store step t the quad writes the four children of ONE parent: 4 pieces of one 128-byte line per quad instead of 4 lines.
        auto epi = [&](auto in_) {                              // what group n completed (issued at the end of group n + 1: its results have landed)
            constexpr int n = decltype(in_)::value;
            constexpr Q4Group G = S.g[n];
            constexpr int sidx = G.child & 3, h = G.child >> 2;
            if constexpr (G.kind == 1) {
                static_for<0, MT>([&](auto im) {
                    constexpr int m = decltype(im)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c1[m][sidx][r] = fmaxf(c1[m][sidx][r] + b10[r], 0.0f);
                });
            }
            if constexpr ((G.kind == 1 || G.fin) && sidx == 3) {
                float onehot[4];                               // [lane & 3 == k]: the B operand of the quad transposes
#pragma unroll
                for (int k = 0; k < 4; ++k) onehot[k] = (co == k) ? 1.0f : 0.0f;
                static_for<0, MT>([&](auto im) {               // one M tile at a time: 16 + 16 transient registers
                    constexpr int m = decltype(im)::value;
                    f32x4 V[4], W[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (G.kind == 1) V[k] = c1[m][k];
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) V[k][r] = fmaxf(acc0[m][4 * h + k][r] + b00[r], 0.0f);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) W[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int r = 0; r < 4; ++r) W[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(V[k][r], onehot[k], W[r], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 v = {W[0][t], W[1][t], W[2][t], W[3][t]};
#ifdef Q4_KO_STORE
                        sink = sink + v;                       // (timing experiment: no output traffic)
#else
                        q4_store(v, rs_out, ovq(m) + (unsigned)(t * 256 + h * 128 + (G.kind == 1 ? 64 : 0)));
#endif
                    }
                });
            }
        };

        using I0 = std::integral_constant<int, 0>;
        static_for<0, D>(gather);
        wait_vmcnt<4 * MT * (D - 1)>();
        load_b(I0{}, I0{});
        static_for<0, 4>([&](auto ie) { load_a(I0{}, ie); });

        static_for<0, S.n>([&](auto in_) {
            constexpr int n = decltype(in_)::value, cur = n & 1;
            constexpr Q4Group G = S.g[n];
            constexpr bool has_next = n + 1 < S.n, next_cell = G.cell + 1 < 64;
            // LDS operations outstanding at entry, oldest first: B(n) x 4, then — first group of a cell — the cell's A quarters, MT each
            if constexpr (G.first) wait_lgkmcnt<3 * MT>(); else wait_lgkmcnt<0>();
            static_for<0, 4>([&](auto ie) { lds_tie(b[cur][decltype(ie)::value]); });
            if constexpr (G.first) static_for<0, MT>([&](auto im) { lds_tie(a[decltype(im)::value][0]); });
            if constexpr (has_next) load_b(std::integral_constant<int, has_next ? n + 1 : n>{}, std::integral_constant<int, cur ^ 1>{});
            if constexpr (G.last && next_cell) wait_vmcnt<S.vm_wait[n]>();        // the next cell's rows have landed in the other ring slot
            if constexpr (G.kind == 1) static_for<0, MT>([&](auto im) { c1[decltype(im)::value][G.child & 3] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
            static_for<0, 4>([&](auto ie) {
                constexpr int e = decltype(ie)::value;
                if constexpr (G.first && e > 0) {
                    // younger than quarter e: the later quarters, B(n + 1), and — a group that is also its cell's last — the next cell's quarters so far
                    __builtin_amdgcn_sched_barrier(0);
                    wait_lgkmcnt<(3 - e) * MT + (has_next ? 4 : 0) + ((G.last && next_cell) ? e * MT : 0)>();
                    static_for<0, MT>([&](auto im) { lds_tie(a[decltype(im)::value][e]); });
                }
                static_for<0, 4>([&](auto iu) {
                    constexpr int u = decltype(iu)::value;
                    static_for<0, MT>([&](auto im) {
                        constexpr int m = decltype(im)::value;
#ifdef Q4_KO_MFMA
                        if constexpr (u == 0) { if constexpr (G.kind == 1) c1[m][G.child & 3][0] += a[m][e][0] * b[cur][e][0]; else acc0[m][G.child][0] += a[m][e][0] * b[cur][e][0]; }   // (timing experiment)
#else
                        if constexpr (G.kind == 1) c1[m][G.child & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(b[cur][e][u], a[m][e][u], c1[m][G.child & 3], 0, 0, 0);
                        else acc0[m][G.child] = __builtin_amdgcn_mfma_f32_4x4x1f32(b[cur][e][u], a[m][e][u], acc0[m][G.child], 0, 0, 0);
#endif
                    });
                });
                if constexpr (G.last && next_cell) {           // this quarter's registers are free: the next cell's quarter e
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(std::integral_constant<int, next_cell ? G.cell + 1 : G.cell>{}, ie);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G.first && G.cell + D < 64) gather(std::integral_constant<int, (G.cell + D < 64) ? G.cell + D : 0>{});   // this cell's slot has been read
            if constexpr (KIND == Q4_PASS_A && n > 0) epi(std::integral_constant<int, (n > 0 ? n - 1 : 0)>{});
        });
        if constexpr (KIND == Q4_PASS_A) {
            epi(std::integral_constant<int, S.n - 1>{});       // (child 7's last product is the tile's last group)
        } else {                                               // the lane's parent: eight logits = 32 contiguous bytes
            static_for<0, MT>([&](auto im) {
                constexpr int m = decltype(im)::value;
                const unsigned ov = (unsigned)((p_base + 64 * m + lane) * 32);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 v = acc0[m][h];
                    if (ep.b0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] + b00[r];
                    }
#ifdef Q4_KO_STORE
                    sink = sink + v;
#else
                    q4_store(v, rs_out, ov + (unsigned)(h * 16));
#endif
                }
            });
        }
#ifdef Q4_KO_STORE
        q4_store(sink, rs_out, ovq(0));
#endif
    }
}

template <int KIND, int NW, int MT, int D = 2>
int launch_child_q4(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                    const IrnEpi& ep, hipStream_t s) {
    const size_t lds = (size_t)table_bytes + (size_t)NW * (D * MT * 4096);
    auto kern = k_child_q4<KIND, NW, MT, D>;
    static ChildLdsGrant granted;
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    const int64_t units = (n_p + 64 * MT - 1) / (64 * MT);
    hipLaunchKernelGGL(kern, dim3(child_grid_units(units, NW, lds)), dim3(NW * 64), lds, s, pnbr, n_p, in, in_ld, table, table_bytes, ep);
    return 0;
}

}  // namespace
