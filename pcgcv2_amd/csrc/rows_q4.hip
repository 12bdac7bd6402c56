// Fused InceptionResNet passes (autoencoder.py:52-57) at C = 32 on a PLAIN level (the encoder's stride-2 level, autoencoder.py:85-89, and
// its stride-8 level, :123-127) in QUAD-BLOCK form on the policy-driven engine of q4x.h: `v_mfma_f32_4x4x1_16b_f32`, lane = row, one
// instruction per (input channel, group of four output channels) — no zero column.  The packed-N kernels these replace on large levels
// (k_rows_irn_a32 / _b32, rows_irn.hip) put Cout = 8 into 16-column tiles: pass A issues 3.9x its algorithmic flops and the matrix pipe is
// what the launch waits for (profiles/r06_rows_irn32_pmc.txt); here only the absent neighbour rows are padding (2.08x).
//
// Pass A  t[row][0:8] = relu(conv0_0 x + b00), t[row][8:16] = relu(conv1_0 x + b10): cells = (offset k, channel half h) — 64 bytes of
//         the neighbour's 128-byte row — two groups per cell (output channels 0-3 / 4-7), plus conv1_0's two at k = 13.
// Pass B  out[row][0:16] = (conv0_1(t[:, :8]) + b01) + x[row][0:16], out[row][16:32] = (conv1_2(relu(conv1_1(t[:, 8:]) + b11)) + b12) +
//         x[row][16:32]: cells = offsets (t rows are 64 bytes), three groups per cell — conv0_1 output channels 0-7, 8-15 from t's first two
//         quarters, conv1_1's 0-7 from the last two; conv1_2 (k1 8 -> 16) as 32 more instructions on the lane's own u.
// Per output element: ascending offset, ascending input channel, one fma per product — the canonical chain; bit-identical to
// k_rows_irn_*32, the VALU pair and the oracle (tests).  Outputs leave through the idle ring in row-major order: 1 KB contiguous per store.
#include "q4x.h"

namespace {

#include "rows_q4_policy.h"

#define ROWS_Q4_PROLOGUE                                                                                                       \
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];                                                    \
    const int lane = threadIdx.x & 63;                                                                                         \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                                         \
    float4* ring = (float4*)(lds_raw + table_bytes) + wave * (D * MT * 256);                                                   \
    unsigned char* stage = (unsigned char*)ring;                                                                               \
    child_stage_table<NW>(table, table_bytes, lds_raw);                                                                        \
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(n * in_ld * 4), 0x00020000);   \
    const int64_t ntiles = (n + 64 * MT - 1) / (64 * MT);

template <int NW, int MT, int D, bool PAIRED>
__global__ void __launch_bounds__(NW * 64)
k_rows_q4_a32(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ in, int in_ld, const float* __restrict__ table,
              int table_bytes, IrnEpi ep) {
    ROWS_Q4_PROLOGUE
    float bias[4][4];                                          // (wave-uniform: scalar registers) chunk c of a t row: conv0_0 0-3, 4-7, conv1_0 0-3, 4-7
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[c][r] = c < 2 ? ep.b0[4 * c + r] : ep.b1[4 * (c - 2) + r];
    for (int it = 0;; ++it) {
        const int64_t tile = child_tile<NW>(it, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * (64 * MT);
        f32x4 acc[MT][4];
        Q4XTile<RowsQ4A32<PAIRED>, MT, D>::run(nbr, n, row0, rs_in, (unsigned)in_ld * 4u, lds_raw, ring, acc);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[m][c][r] + bias[c][r], 0.0f);
                *(f32x4*)(stage + m * 4096 + q4x_stage_addr<4>(lane, c)) = v;
            }
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = lane + 64 * i, r = q >> 2, c = q & 3;
                const int64_t row = row0 + 64 * m + r;
                if (row < n) *(f32x4*)(ep.out + row * 16 + 4 * c) = *(const f32x4*)(stage + m * 4096 + q4x_stage_addr<4>(r, c));
            }
        wave_lds_sync();
    }
}

template <int NW, int MT, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_q4_b32(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ in /* t [n, 16] */, int in_ld,
              const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    static_assert(D >= 2, "the output staging needs two ring slots");
    ROWS_Q4_PROLOGUE
    float b01[4][4], b11[2][4], b12[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) { b01[g][r] = ep.b0[4 * g + r]; b12[g][r] = ep.b2[4 * g + r]; }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) b11[g][r] = ep.b1[4 * g + r];
    const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + (lane & 3) * 16);
    for (int it = 0;; ++it) {
        const int64_t tile = child_tile<NW>(it, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * (64 * MT);
        f32x4 acc[MT][6];
        Q4XTile<RowsQ4B32, MT, D>::run(nbr, n, row0, rs_in, (unsigned)in_ld * 4u, lds_raw, ring, acc);
        // residual rows, lane-linear (piece q = lane + 64 i: chunk q & 7 of tile row q >> 3): requested before the conv1_2 products and the staging
        f32x4 xr[MT][8];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = lane + 64 * i;
                const int64_t row = row0 + 64 * m + (q >> 3);
                xr[m][i] = row < n ? *(const f32x4*)(ep.x + row * ep.x_ld + 4 * (q & 7)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        // conv1_2 (k1 8 -> 16) on u = relu(conv1_1 + b11): weights from the table's last two fragments, the lane's own u as the row operand
        f32x4 w12[2][4];
        static_for<0, 2>([&](auto ip) {
            static_for<0, 4>([&](auto ie) {
                constexpr int p = decltype(ip)::value, e = decltype(ie)::value;
                w12[p][e] = lds_ld128_off<(RowsQ4B32::FRAG_W12 + p) * 256 + e * 16>(tab_lane);
            });
        });
        wait_lgkmcnt<0>();                                     // (inline-asm LDS reads: the compiler does not wait for them)
        static_for<0, 2>([&](auto ip) { static_for<0, 4>([&](auto ie) { lds_tie(w12[decltype(ip)::value][decltype(ie)::value]); }); });
        f32x4 o2[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 u[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) u[g][r] = fmaxf(acc[m][4 + g][r] + b11[g][r], 0.0f);
#pragma unroll
            for (int g = 0; g < 4; ++g) o2[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu)
                        o2[m][2 * p + (e >> 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(w12[p][e][uu], u[e & 1][uu], o2[m][2 * p + (e >> 1)], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = c < 4 ? acc[m][c][r] + b01[c][r] : o2[m][c - 4][r] + b12[c - 4][r];
                *(f32x4*)(stage + m * 8192 + q4x_stage_addr<8>(lane, c)) = v;
            }
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = lane + 64 * i, r = q >> 3, c = q & 7;
                const int64_t row = row0 + 64 * m + r;
                f32x4 v = *(const f32x4*)(stage + m * 8192 + q4x_stage_addr<8>(r, c));
                v = v + xr[m][i];
                if (row < n) *(f32x4*)(ep.out + row * ep.out_ld + 4 * c) = v;
            }
        wave_lds_sync();
    }
}

template <typename K>
int launch_rows_q4(K kern, int nw, int mt, int d, const int32_t* nbr, int64_t n, const float* in, int in_ld, const float* table, int table_bytes,
                   const IrnEpi& ep, hipStream_t s, ChildLdsGrant& granted) {
    const size_t lds = (size_t)table_bytes + (size_t)nw * d * mt * 4096;
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    const int64_t units = (n + 64 * mt - 1) / (64 * mt);
    hipLaunchKernelGGL(kern, dim3(child_grid_units(units, nw, lds)), dim3(nw * 64), lds, s, nbr, n, in, in_ld, table, table_bytes, ep);
    return 0;
}

int g_rows_q4_variant = 0;                                     // 0 = by level size; 1.. = a fixed instantiation (A/B tools)

}  // namespace

// which (waves per workgroup, M tiles per wave, ring depth) instantiation pcgc_irn_rows_q4_pass launches: 0 = the default — 3; 1 = (8, 2, 2):
// two M tiles per wave share every weight operand; 2 = (8, 1, 4), pass A with PAIRED half-row gathers; 3 = pass A (16, 1, 2), pass B (12, 1, 2: its
// 142 registers allow three waves per SIMD).  Measured on the 255 692-row level (tools/rows32_ab.py, us per block): 1: 92-99, 2: 94, 3: 87 — pass A
// takes ~50 us in every geometry tried (4 x 2 x 4, 12 x 1 x 2, paired or not: matrix pipe 53 % busy, texture-addresser FIFOs full 45 % of the
// time, LDS ~60 %: three co-limiters; profiles/r06_rows_q4.md).  An A/B knob; results do not depend on it.
extern "C" int pcgc_set_rows_q4_variant(int v) {
    if (v < 0 || v > 3) return -1;
    g_rows_q4_variant = v;
    return 0;
}

// Fused InceptionResNet passes at C = 32 on a plain level through its own k3 map nbr [27][n], quad-block form.  pass 1 (A): in = x [n, 32]
// -> out = t [n, 16]; pass 2 (B): in = t -> out [n, 32] with the residual x.  tables: ops.rows_q4_tables(params) (28 672 / 21 248 bytes).
extern "C" int pcgc_irn_rows_q4_pass(const int32_t* nbr, int64_t n, int C, int pass, const float* in, int in_ld, const float* table,
                                     int64_t table_bytes, const float* b0, const float* b1, const float* b2, const float* x, int x_ld,
                                     float* out, int out_ld, void* stream) {
    PCGC_REQUIRE(nbr && in && table, "null argument");
    PCGC_REQUIRE((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");
    PCGC_REQUIRE(n * (int64_t)in_ld * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets");
    PCGC_REQUIRE(C == 32, "the quad-block rows kernels serve C = 32");
    PCGC_REQUIRE(pass == 1 || pass == 2, "pass must be 1 (A) or 2 (B)");
    PCGC_REQUIRE(out && b0 && b1 && (pass == 1 || (b2 && x)), "null argument");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (pass == 1 || ((x_ld & 3) == 0 && (((uintptr_t)x) & 15) == 0)),
                 "rows must be 16-byte aligned");
    PCGC_REQUIRE(pass == 2 || out_ld == 16, "pass A writes a dense [rows, 16] tensor");
    PCGC_REQUIRE(pass == 1 ? in_ld >= 32 : in_ld == 16, "input rows: x at least 32 wide, t dense [rows, 16]");
    PCGC_REQUIRE(pass == 1 || (out_ld >= 32 && x_ld >= 32), "pass B rows narrower than the layer");
    PCGC_REQUIRE(table_bytes == (pass == 1 ? RowsQ4A32Base::NFRAG : RowsQ4B32::NFRAG) * 256, "table size");
    if (n == 0) return 0;
    hipStream_t s = S(stream);
    IrnEpi ep{b0, b1, b2, x, x_ld, out, out_ld};
    const int v = g_rows_q4_variant ? g_rows_q4_variant : 3;
    static ChildLdsGrant granted[6];
    int rc;
#define Q4_GO(SLOT, KERN, NW_, MT_, D_) launch_rows_q4(KERN, NW_, MT_, D_, nbr, n, in, in_ld, table, (int)table_bytes, ep, s, granted[SLOT])
    if (pass == 1)
        rc = v == 1 ? Q4_GO(0, (k_rows_q4_a32<8, 2, 2, false>), 8, 2, 2) : v == 2 ? Q4_GO(1, (k_rows_q4_a32<8, 1, 4, true>), 8, 1, 4)
           : Q4_GO(2, (k_rows_q4_a32<16, 1, 2, false>), 16, 1, 2);
    else
        rc = v == 1 ? Q4_GO(3, (k_rows_q4_b32<8, 2, 2>), 8, 2, 2) : v == 2 ? Q4_GO(4, (k_rows_q4_b32<8, 1, 4>), 8, 1, 4) : Q4_GO(5, (k_rows_q4_b32<12, 1, 2>), 12, 1, 2);
#undef Q4_GO
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("irn_rows_q4_pass");
    return 0;
}
