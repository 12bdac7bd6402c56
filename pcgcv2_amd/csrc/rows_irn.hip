// Fused InceptionResNet passes (autoencoder.py:52-57) at C = 64 on a PLAIN level — the encoder's stride-4 level — with the machinery of the
// children-level kernels (child_kernels.h): the whole B-fragment table resident in LDS, one wave = one 16-row M tile that walks the 27
// kernel offsets on its own (rows fetched by `buffer_load ... lds` into a per-wave ring, D - 1 offsets of gather in flight behind the
// MFMAs), no workgroup barrier and no weight staging per offset.  The block-sparse gather kernels these replace on that level
// (k_conv_gather_mfma_pipe<64,32,2> / <32,48,2> + k_irn_tail) stage one weight slice per offset through LDS for four waves at a
// time: 54 barrier-separated steps of 1.85 us carrying 0.2-0.5 us of MFMA work each (profiles/r03_child_skip_experiment.md §5).
//
// Geometry: "cell" k of a tile = kernel offset k of its own 16 rows: row nbr[k][r] of the level itself (ROW_MUL = 1, no child index), fed
// to the accumulator tiles in ascending k — the same fmaf chain per output element as every other kernel of this library (ascending
// offset, then ascending input channel: 16-channel block, K-step, K index).  Tables: ops.child_irn_tables(params) at C = 64 (one
// fragment per (offset, 16-channel block): the layout the children-level C = 64 kernels use).
#include "child_kernels.h"

namespace {

struct RowsGeometry {
    static constexpr int NCELLS = 27, ROW_MUL = 1, NMAP = 27;
    static constexpr int kp(int c) { return c; }
    static constexpr int child(int) { return 0; }
    static constexpr int byte_off(int) { return 0; }
};
// pass A (C = 64): tile 0 = conv0_0 (k3 64 -> 16), tile 1 = conv1_0 (k1 64 -> 16: the centre offset only), in HALF-row cells: cell 2 k + h = channels [32 h, 32 h + 32) of offset k's rows (two 16-channel blocks, 2 KB per 16 rows), so that
// twelve waves can each keep TWO gathers in flight in the 48 KB the 112 KB table leaves (whole-row cells: one 4 KB slot per wave — nothing in
// flight behind a wave's own MFMAs): 54.5 -> 51 us on 71 k rows, 103 -> 93 on 150 k.  (Quarter-row cells, four in flight: no further gain.)
// Same chain: ascending offset, then ascending channel.
struct RowsPassA64H {
    static constexpr int NCELLS = 54, ROW_MUL = 1, NMAP = 27;
    static constexpr int kp(int c) { return c >> 1; }
    static constexpr int child(int) { return 0; }
    static constexpr int byte_off(int c) { return (c & 1) * 128; }
    static constexpr int NB = 2, ROWCHUNKS = 4, T = 2, KS = 4, Z_HALF = -1;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int c, int t) { return t == 0 || (c >> 1) == 13; }
    static constexpr int frag(int c, int t) { return t == 0 ? (c >> 1) : 27; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * 4 + 2 * (c & 1) + cb) * 1024; }
};
// pass B: t rows (32 wide: block 0 = relu(conv0_0), block 1 = relu(conv1_0)) -> tiles 0, 1 = conv0_1 (k3 16 -> 32) from block 0,
// tile 2 = conv1_1 (k3 16 -> 16) from block 1.  Fragments: conv0_1 (k, n) = 2 k + n, conv1_1 k = 54 + k, conv1_2 (k1 16 -> 32) n = 81 + n.
struct RowsPassB64 : RowsGeometry {
    static constexpr int NB = 2, ROWCHUNKS = 4, T = 3, KS = 4, Z_HALF = -1, FRAG_W12 = 81;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int, int) { return true; }
    static constexpr int frag(int c, int t) { return t < 2 ? 2 * c + t : 54 + c; }
    static constexpr bool uses_block(int t, int cb) { return t < 2 ? cb == 0 : cb == 1; }
    static constexpr int frag_off(int c, int t, int) { return frag(c, t) * 1024; }
};

// pass A:  t[row][0:16] = relu(conv0_0 + b00), t[row][16:32] = relu(conv1_0 + b10)        acc[t][r] = row 4 mq + r of the tile, column mi
template <int NW, int D, class V = RowsPassA64H>
__global__ void __launch_bounds__(NW * 64)
k_rows_irn_a64(const int32_t* __restrict__ pnbr, int64_t n_p /* rows of the level */, const float* __restrict__ in, int in_ld,
               const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    CHILD_KERNEL_PROLOGUE(V, NW, D, D * V::NB * 64)
    const float b00 = ep.b0[mi], b10 = ep.b1[mi];
    float* scratch = (float*)ring;                             // [16 rows][32]: 2 KB of the (idle) gather ring
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            scratch[(4 * mq + r) * 32 + mi] = fmaxf(acc[0][r] + b00, 0.0f);
            scratch[(4 * mq + r) * 32 + 16 + mi] = fmaxf(acc[1][r] + b10, 0.0f);
        }
        wave_lds_sync();
        child_flush<32, 16>(scratch, row0, n_p, ep.out, 32, 0, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// pass B:  out[row][0:32]  = (conv0_1(t[:, :16]) + b01) + x[row][0:32]
//          out[row][32:64] = (conv1_2(relu(conv1_1(t[:, 16:]) + b11)) + b12) + x[row][32:64]      (conv1_2: 8 MFMAs on u through the scratch)
template <int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_irn_b64(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in /* t [.., 32] */, int in_ld,
               const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    using V = RowsPassB64;
    constexpr int NEEDF4 = (16 * 16 + 16 * 64) / 4;                                        // us [16][16] + stage [16][64]
    constexpr int RINGF4 = (D * V::NB * 64 > NEEDF4) ? D * V::NB * 64 : NEEDF4;
    CHILD_KERNEL_PROLOGUE(V, NW, D, RINGF4)
    float* us = (float*)ring;
    float* stage = us + 16 * 16;
    f32x4 w12[2];                                                                          // conv1_2 B fragments: W12[4 jj + mq][16 n + mi]
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) w12[n2] = *((const f32x4*)(lds_raw + (V::FRAG_W12 + n2) * 1024) + lane);
    const float b01a = ep.b0[mi], b01b = ep.b0[16 + mi], b11 = ep.b1[mi], b12a = ep.b2[mi], b12b = ep.b2[16 + mi];
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        ChildResidual<64, 16> rr;                                  // residual rows: requested now, consumed after the staging hand-off
        child_flush_prefetch<64, 16>(rr, row0, n_p, ep.x, ep.x_ld, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            us[(4 * mq + r) * 16 + mi] = fmaxf(acc[2][r] + b11, 0.0f);
            stage[(4 * mq + r) * 64 + mi] = acc[0][r] + b01a;
            stage[(4 * mq + r) * 64 + 16 + mi] = acc[1][r] + b01b;
        }
        wave_lds_sync();
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) d = __builtin_amdgcn_mfma_f32_16x16x4f32(us[mi * 16 + 4 * jj + mq], w12[n2][jj], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(4 * mq + r) * 64 + 32 + 16 * n2 + mi] = d[r] + (n2 ? b12b : b12a);
        }
        wave_lds_sync();
        child_flush<64, 16>(stage, rr, ep.x != nullptr, row0, n_p, ep.out, ep.out_ld, 0, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// ---- C = 32 (Q = 8) on a plain level: the small levels (the encoder's stride-8 level: 18.7 k rows), where the row-split VALU kernels are
// chains of launch-sized latencies.  Pass A: ONE tile per offset — columns 0-7 = conv0_0 (k3 32 -> 8), columns 8-15 = conv1_0 (k1 32 -> 8:
// zero columns except at the centre offset; fma(x, 0, acc) = acc).  Pass B: t rows are 16 wide = one block: K-steps {0, 1} (channels 0-7 =
// relu(conv0_0)) feed tile 0 = conv0_1 (k3 8 -> 16), K-steps {2, 3} (relu(conv1_0)) feed tile 1 = conv1_1 (k3 8 -> 8, columns 8-15 zero).
// Tables: ops.rows_irn32_tables.
struct RowsPassA32 : RowsGeometry {
    static constexpr int NB = 2, ROWCHUNKS = 4, T = 1, KS = 4, Z_HALF = -1;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int, int) { return true; }
    static constexpr int frag(int c, int) { return c; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 1024; }
};
struct RowsPassB32 : RowsGeometry {
    static constexpr int NB = 1, ROWCHUNKS = 4, T = 2, KS = 2, Z_HALF = -1, FRAG_W12 = 54;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int t) { return t == 0 ? 0 : 2; }
    static constexpr bool active(int, int) { return true; }
    static constexpr int frag(int c, int t) { return t == 0 ? c : 27 + c; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int) { return frag(c, t) * 512; }
};
// pass A:  t[row][0:8] = relu(conv0_0 + b00), t[row][8:16] = relu(conv1_0 + b10)
template <int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_irn_a32(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
               const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    using V = RowsPassA32;
    CHILD_KERNEL_PROLOGUE(V, NW, D, D * V::NB * 64)
    const float bcol = mi < 8 ? ep.b0[mi] : ep.b1[mi - 8];
    float* scratch = (float*)ring;                             // [16 rows][16]: 1 KB
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(4 * mq + r) * 16 + mi] = fmaxf(acc[0][r] + bcol, 0.0f);
        wave_lds_sync();
        child_flush<16, 16>(scratch, row0, n_p, ep.out, 16, 0, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}
// pass B:  out[row][0:16]  = (conv0_1(t[:, :8]) + b01) + x[row][0:16]
//          out[row][16:32] = (conv1_2(relu(conv1_1(t[:, 8:]) + b11)) + b12) + x[row][16:32]       (conv1_2 k1 8 -> 16: two MFMAs on u)
template <int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_irn_b32(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in /* t [.., 16] */, int in_ld,
               const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    using V = RowsPassB32;
    constexpr int NEEDF4 = (16 * 8 + 16 * 32) / 4;                                         // us [16][8] + stage [16][32]
    constexpr int RINGF4 = (D * V::NB * 64 > NEEDF4) ? D * V::NB * 64 : NEEDF4;
    CHILD_KERNEL_PROLOGUE(V, NW, D, RINGF4)
    float* us = (float*)ring;
    float* stage = us + 16 * 8;
    float w12[2];                                                                          // conv1_2 B fragment: W12[4 jj + mq][mi]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) w12[jj] = ((const float*)(lds_raw + V::FRAG_W12 * 512))[lane * 2 + jj];
    const float b01 = ep.b0[mi], b11 = mi < 8 ? ep.b1[mi] : 0.0f, b12 = ep.b2[mi];
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        ChildResidual<32, 16> rr;                                  // residual rows: requested now, consumed after the staging hand-off
        child_flush_prefetch<32, 16>(rr, row0, n_p, ep.x, ep.x_ld, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mi < 8) us[(4 * mq + r) * 8 + mi] = fmaxf(acc[1][r] + b11, 0.0f);
            stage[(4 * mq + r) * 32 + mi] = acc[0][r] + b01;
        }
        wave_lds_sync();
        f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) d = __builtin_amdgcn_mfma_f32_16x16x4f32(us[mi * 8 + 4 * jj + mq], w12[jj], d, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(4 * mq + r) * 32 + 16 + mi] = d[r] + b12;
        wave_lds_sync();
        child_flush<32, 16>(stage, rr, ep.x != nullptr, row0, n_p, ep.out, ep.out_ld, 0, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// plain k3 conv Cin = 16 NB -> Cout = 16 NT on a plain level: tile n = output columns [16 n, 16 n + 16), fragment (k, n) = the offset's
// weight slice (ops.child_conv_table: [k][n][cb])
template <int NB_, int NT>
struct RowsConv : RowsGeometry {
    static constexpr int NB = NB_, ROWCHUNKS = 4, T = NT, KS = 4, Z_HALF = -1;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int, int) { return true; }
    static constexpr int frag(int c, int t) { return c * NT + t; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 1024; }
};
template <int NB, int NT, int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_conv(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in, int in_ld,
            const float* __restrict__ table, int table_bytes, ChildEpi ep) {
    using V = RowsConv<NB, NT>;
    constexpr int W = 16 * NT;
    constexpr int NEEDF4 = 16 * W / 4;                         // epilogue staging [16 rows][W]
    constexpr int RINGF4 = (D * NB * 64 > NEEDF4) ? D * NB * 64 : NEEDF4;
    CHILD_KERNEL_PROLOGUE(V, NW, D, RINGF4)
    float* scratch = (float*)ring;
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = ep.bias ? ep.bias[16 * t + mi] : 0.0f;
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_in, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        ChildResidual<W, 16> rr;                                  // residual rows: requested now, consumed after the staging hand-off
        child_flush_prefetch<W, 16>(rr, row0, n_p, ep.res, ep.res_ld, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][r];
                if (ep.bias) v = v + bv[t];
                scratch[(4 * mq + r) * W + 16 * t + mi] = v;
            }
        }
        wave_lds_sync();
        child_flush<W, 16>(scratch, rr, ep.res != nullptr, row0, n_p, ep.out, ep.out_ld, ep.relu, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// k2 s2 down conv (autoencoder.py:78-84,97-103,116-122): tile = 16 COARSE rows, "cell" k = child offset k of the coarse row: row
// down[k][r] of the FINE level (absent children: zero rows), fragment (k, n) = W[k][:, 16 n : 16 n + 16].  Same chain as the gather
// kernels: ascending k, ascending input channel.
struct DownGeometry {
    static constexpr int NCELLS = 8, ROW_MUL = 1, NMAP = 8;
    static constexpr int kp(int c) { return c; }
    static constexpr int child(int) { return 0; }
    static constexpr int byte_off(int) { return 0; }
};
template <int NB_, int NT>
struct RowsDown : DownGeometry {
    static constexpr int NB = NB_, ROWCHUNKS = 4, T = NT, KS = 4, Z_HALF = -1;
    static constexpr bool HALF = false;
    static constexpr int kfirst(int) { return 0; }
    static constexpr bool active(int, int) { return true; }
    static constexpr int frag(int c, int t) { return c * NT + t; }
    static constexpr bool uses_block(int, int) { return true; }
    static constexpr int frag_off(int c, int t, int cb) { return (frag(c, t) * NB + cb) * 1024; }
};
template <int NB, int NT, int NW, int D>
__global__ void __launch_bounds__(NW * 64)
k_rows_down(const int32_t* __restrict__ pnbr /* down [8][n_p] */, int64_t n_p /* coarse rows */, const float* __restrict__ in, int in_ld,
            const float* __restrict__ table, int table_bytes, ChildEpi ep, int64_t n_in /* fine rows */) {
    using V = RowsDown<NB, NT>;
    constexpr int W = 16 * NT;
    constexpr int NEEDF4 = 16 * W / 4;                         // epilogue staging [16 rows][W]
    constexpr int RINGF4 = (D * NB * 64 > NEEDF4) ? D * NB * 64 : NEEDF4;
    CHILD_KERNEL_PROLOGUE(V, NW, D, RINGF4)
    (void)rs_in;                                               // (sized for the coarse level: the gathered rows are the fine level's)
    const __amdgpu_buffer_rsrc_t rs_fine = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(n_in * in_ld * 4), 0x00020000);
    float* scratch = (float*)ring;
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = ep.bias ? ep.bias[16 * t + mi] : 0.0f;
    for (int i = 0;; ++i) {
        const int64_t tile = child_tile<NW>(i, wave, ntiles);
        if (tile < 0) break;
        const int64_t row0 = tile * 16;
        CHILD_T(t_it0);
        f32x4 acc[V::T];
        child_tile_mainloop<V, D>(pnbr, n_p, row0, rs_fine, in_ld, lds_raw, ring, acc CHILD_DBG_ARG);
        ChildResidual<W, 16> rr;                                  // residual rows: requested now, consumed after the staging hand-off
        child_flush_prefetch<W, 16>(rr, row0, n_p, ep.res, ep.res_ld, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][r];
                if (ep.bias) v = v + bv[t];
                scratch[(4 * mq + r) * W + 16 * t + mi] = v;
            }
        }
        wave_lds_sync();
        child_flush<W, 16>(scratch, rr, ep.res != nullptr, row0, n_p, ep.out, ep.out_ld, ep.relu, lane);
        wave_lds_sync();
#ifdef PCGC_CHILD_TIMING
        { CHILD_T(t_dr0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CHILD_T(t_it1); CHILD_TADD(5, t_dr0, t_it1); CHILD_TADD(4, t_it0, t_it1); }
#endif
    }
    CHILD_TFLUSH;
}

// persistent grid over 16-row tiles (child_grid counts 16-PARENT tiles: the same number here)
template <typename K, typename EPI>
int launch_rows(K kern, int nw, size_t lds, const int32_t* nbr, int64_t n, const float* in, int in_ld, const float* table, int table_bytes,
                const EPI& ep, hipStream_t s, ChildLdsGrant& granted) {
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    hipLaunchKernelGGL(kern, dim3(child_grid(n, nw, lds)), dim3(nw * 64), lds, s, nbr, n, in, in_ld, table, table_bytes, ep);
    return 0;
}

}  // namespace

// C = 32 levels of up to this many rows (one round of 8-wave workgroups on 256 CUs) take the deep-ring instantiations: 18.7 k rows 31 -> 20 us per
// block, 4.6 k rows 25 -> 19; at 55 k rows (two rounds) the 16-wave form is ahead, 34 vs 36 (tools/rows32_small_ab.py)
constexpr int64_t ROWS32_DEEP_MAX = 32768;

// Fused InceptionResNet passes at C = 64 on a plain level through its own k3 map nbr [27][n].  pass 1 (A): in = x [n, 64] -> out = t [n, 32];
// pass 2 (B): in = t -> out [n, 64] with the residual x.  tables: ops.child_irn_tables(params) (112 KB / 83 KB).
extern "C" int pcgc_irn_rows_pass(const int32_t* nbr, int64_t n, int C, int pass, const float* in, int in_ld, const float* table,
                                  int64_t table_bytes, const float* b0, const float* b1, const float* b2, const float* x, int x_ld, float* out,
                                  int out_ld, void* stream) {
    PCGC_REQUIRE(nbr && in && table, "null argument");
    PCGC_REQUIRE((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");
    PCGC_REQUIRE(n * (int64_t)in_ld * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets");
    PCGC_REQUIRE(C == 64 || C == 32, "channels must be 32 or 64");
    PCGC_REQUIRE(pass == 1 || pass == 2, "pass must be 1 (A) or 2 (B)");
    PCGC_REQUIRE(out && b0 && b1 && (pass == 1 || (b2 && x)), "null argument");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (pass == 1 || ((x_ld & 3) == 0 && (((uintptr_t)x) & 15) == 0)),
                 "rows must be 16-byte aligned");
    PCGC_REQUIRE(pass == 2 || out_ld == C / 2, "pass A writes a dense [rows, C/2] tensor");
    PCGC_REQUIRE(pass == 1 ? in_ld >= C : in_ld >= C / 2, "input rows narrower than the pass reads");
    PCGC_REQUIRE(table_bytes == (C == 64 ? (pass == 1 ? 28 * 4 * 1024 : 83 * 1024) : (pass == 1 ? 27 * 2 * 1024 : 55 * 512)), "table size");
    if (n == 0) return 0;
    hipStream_t s = S(stream);
    IrnEpi ep{b0, b1, b2, x, x_ld, out, out_ld};
    int rc;
    static ChildLdsGrant granted[8];
#define ROWS_GO(SLOT, KERN, NW_, RINGBYTES) launch_rows(KERN, NW_, (size_t)table_bytes + (size_t)(NW_) * (RINGBYTES), nbr, n, in, in_ld, table, (int)table_bytes, ep, s, granted[SLOT])
    if (C == 32 && n <= ROWS32_DEEP_MAX) {
        // small levels (the encoder's stride-8 level: 1 171 sixteen-row tiles for 4 096 wave slots — every wave runs ONE tile, and the launch takes a
        // tile's chain of 27 gather latencies): half the waves, rings two to four times as deep (LDS is plentiful when every CU holds one workgroup)
        rc = pass == 1 ? ROWS_GO(4, (k_rows_irn_a32<8, 4>), 8, 4 * 2048) : ROWS_GO(5, (k_rows_irn_b32<8, 8>), 8, 8 * 1024);
    } else if (C == 32) {                                      // 54 KB / 27.5 KB tables; 16 waves, ring slots of 2 KB / 1 KB (pass B: scratch 2.5 KB)
        rc = pass == 1 ? ROWS_GO(0, (k_rows_irn_a32<16, 2>), 16, 2 * 2048) : ROWS_GO(1, (k_rows_irn_b32<16, 4>), 16, 4 * 1024);
    } else if (pass == 1) {                                    // 112 KB table: 48 KB for the rings (12 waves, two 2 KB slots each)
        rc = ROWS_GO(2, (k_rows_irn_a64<12, 2, RowsPassA64H>), 12, 2 * 2048);
    } else {                                                   // 83 KB table: 8 KB per wave (ring slots of 2 KB; the epilogue scratch needs 5 KB)
        rc = ROWS_GO(3, (k_rows_irn_b64<12, 2>), 12, 5120);
    }
#undef ROWS_GO
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("irn_rows_pass");
    return 0;
}

// Plain k3 conv 32 -> 32 on a plain level (the encoder's conv1, autoencoder.py:90-96) through its own k3 map: 108 KB of fragments
// resident in LDS (ops.child_conv_table), one wave per 16-row tile.  Epilogue as pcgc_conv_gather: (+ bias) (+ residual) (relu).
extern "C" int pcgc_conv_rows(const int32_t* nbr, int64_t n, const float* in, int Cin, int in_ld, const float* table, int64_t table_bytes,
                              const float* bias, const float* residual, int res_ld, int relu, float* out, int Cout, int out_ld, void* stream) {
    PCGC_REQUIRE(nbr && in && table && out, "null argument");
    PCGC_REQUIRE((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0, "output rows must be 16-byte aligned");
    PCGC_REQUIRE(residual == nullptr || ((res_ld & 3) == 0 && (((uintptr_t)residual) & 15) == 0), "residual rows must be 16-byte aligned");
    PCGC_REQUIRE(n * (int64_t)in_ld * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets");
    PCGC_REQUIRE(Cin == 32 && Cout == 32, "conv_rows: 32 -> 32");
    PCGC_REQUIRE(in_ld >= Cin && out_ld >= Cout, "rows narrower than the layer");
    PCGC_REQUIRE(table_bytes == (int64_t)27 * Cin * Cout * 4, "table size");
    if (n == 0) return 0;
    hipStream_t s = S(stream);
    ChildEpi ep{bias, residual, res_ld, relu, out, out_ld, Cout / 16};
    static ChildLdsGrant granted[4];
    int rc;
#define ROWS_GO(SLOT, KERN, NW_, RINGBYTES) launch_rows(KERN, NW_, (size_t)table_bytes + (size_t)(NW_) * (RINGBYTES), nbr, n, in, in_ld, table, (int)table_bytes, ep, s, granted[SLOT])
    rc = ROWS_GO(3, (k_rows_conv<2, 2, 12, 2>), 12, 2 * 2048);
#undef ROWS_GO
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("conv_rows");
    return 0;
}

// k2 s2 down conv through the level pair's `down` map [8][n_coarse] (entries: fine rows, -1 = no such child): 16 -> 32, 32 -> 64, 64 -> 32
// (the encoder's down0 / down1 / down2).  table: ops.child_conv_table(kernel) ([k][n][cb] fragments, 16 / 64 / 64 KB).
extern "C" int pcgc_conv_down_rows(const int32_t* down, int64_t n_coarse, const float* in, int64_t n_in, int Cin, int in_ld, const float* table,
                                   int64_t table_bytes, const float* bias, int relu, float* out, int Cout, int out_ld, void* stream) {
    PCGC_REQUIRE(down && in && table && out, "null argument");
    PCGC_REQUIRE((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)table) & 15) == 0, "unaligned input");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0, "output rows must be 16-byte aligned");
    PCGC_REQUIRE(n_in * (int64_t)in_ld * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets");
    PCGC_REQUIRE((Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 32), "conv_down_rows: 16 -> 32, 32 -> 64 or 64 -> 32");
    PCGC_REQUIRE(in_ld >= Cin && out_ld >= Cout, "rows narrower than the layer");
    PCGC_REQUIRE(table_bytes == (int64_t)8 * Cin * Cout * 4, "table size");
    if (n_coarse == 0) return 0;
    hipStream_t s = S(stream);
    ChildEpi ep{bias, nullptr, 0, relu, out, out_ld, Cout / 16};
    static ChildLdsGrant granted[3];
    int rc;
#define DOWN_GO(SLOT, NB_, NT_, NW_, D_)                                                                                                       \
    do {                                                                                                                                       \
        constexpr int need = 16 * 16 * (NT_) * 4, ringb = ((D_) * (NB_) * 1024 > need) ? (D_) * (NB_) * 1024 : need;                           \
        const size_t lds = (size_t)table_bytes + (size_t)(NW_) * ringb;                                                                        \
        auto kern = k_rows_down<NB_, NT_, NW_, D_>;                                                                                            \
        rc = child_lds_limit(kern, lds, granted[SLOT]);                                                                                        \
        if (!rc) hipLaunchKernelGGL(kern, dim3(child_grid(n_coarse, NW_, lds)), dim3((NW_) * 64), lds, s, down, n_coarse, in, in_ld, table,   \
                                    (int)table_bytes, ep, n_in);                                                                               \
    } while (0)
    if (Cin == 16) DOWN_GO(0, 1, 2, 16, 2);
    else if (Cin == 32) DOWN_GO(1, 2, 4, 12, 2);
    else DOWN_GO(2, 4, 2, 8, 2);            // (eight waves with two ring slots each: 16.3 -> 12.2 us at 18.7 k coarse rows, 26 -> 23 at 55 k, against twelve waves with one)
#undef DOWN_GO
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("conv_down_rows");
    return 0;
}
CHILD_TIMING_READER(pcgc_child_timing_rows)
