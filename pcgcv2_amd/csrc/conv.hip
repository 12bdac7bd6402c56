// Sparse convolution family: MinkowskiConvolution (k3 s1, k2 s2, k1) and MinkowskiGenerativeConvolutionTranspose
// (k2 s2) of the reference's autoencoder.py, as output-stationary gather kernels.
//
// Canonical arithmetic (identical to oracle/pcgc_oracle.c):
//   acc = +0 ; for k ascending, for ci ascending: acc = fmaf(in[nbr[k][o]][ci], W[k][ci][co], acc)
//   out = acc + bias ; out += residual ; out = relu(out)
// Output-stationary => deterministic, no atomics, and the k-ascending order survives any tiling.
#include <cstdlib>
#include "pcgc_common.h"
#include "mfma_util.h"

// ----------------------------------------------------------------------------------------------------------------
// v0 generic kernel: one thread per output row, COUT accumulators in VGPRs, weights through wave-uniform (scalar)
// loads, input rows as 16-byte vector loads.  Handles every (Cin, Cout) of the model incl. Cin=1 and Cout=1.
// ----------------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ void __launch_bounds__(256)
k_conv_gather_valu(const int32_t* __restrict__ nbr, int K, int64_t n_out, const float* __restrict__ in, int Cin, int in_ld,
                   int in_coff, const float* __restrict__ W, const float* __restrict__ bias,
                   const float* __restrict__ res, int res_ld, int res_coff, int relu, float* __restrict__ out, int out_ld,
                   int out_coff) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    const bool vec4 = ((Cin & 3) == 0) && ((in_ld & 3) == 0) && ((in_coff & 3) == 0);
    for (int k = 0; k < K; ++k) {
        int64_t r = nbr ? (int64_t)nbr[(int64_t)k * n_out + o] : o;
        if (r < 0) continue;
        const float* x = in + r * in_ld + in_coff;
        const float* w = W + (int64_t)k * Cin * COUT;
        if (vec4) {
            for (int ci = 0; ci < Cin; ci += 4) {
                float4 xv = *(const float4*)(x + ci);
                const float* w0 = w + (int64_t)ci * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv.x, w0[co], acc[co]);
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv.y, w0[COUT + co], acc[co]);
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv.z, w0[2 * COUT + co], acc[co]);
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv.w, w0[3 * COUT + co], acc[co]);
            }
        } else {
            for (int ci = 0; ci < Cin; ++ci) {
                float a = x[ci];
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a, w[ci * COUT + co], acc[co]);
            }
        }
    }
    float* y = out + o * out_ld + out_coff;
    const float* rr = res ? res + o * res_ld + res_coff : nullptr;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = acc[co];
        if (bias) v = v + bias[co];
        if (rr) v = v + rr[co];
        if (relu) v = fmaxf(v, 0.0f);
        y[co] = v;
    }
}

template <int COUT>
static void launch_valu(const int32_t* nbr, int K, int64_t n_out, const float* in, int Cin, int in_ld, int in_coff,
                        const float* W, const float* bias, const float* res, int res_ld, int res_coff, int relu, float* out,
                        int out_ld, int out_coff, hipStream_t s) {
    hipLaunchKernelGGL((k_conv_gather_valu<COUT>), dim3(grid_for(n_out, 256)), dim3(256), 0, s, nbr, K, n_out, in, Cin, in_ld,
                       in_coff, W, bias, res, res_ld, res_coff, relu, out, out_ld, out_coff);
}


// The first layer of the codec sees the occupancy indicator: one input channel that is 1.0 on every occupied voxel
// (data_utils.py:104 `feats = torch.ones(...)`, :114 after a rescale).  fmaf(1.0f, w, acc) is exactly acc + w, so the layer is a sum
// of the kernel slices of the PRESENT offsets, in ascending offset order — the canonical chain — and the 27 four-byte feature
// gathers per row (8.1 M of them on a vox10 frame: the texture-addresser bound of k_conv_gather_valu<16>, 88-105 us) are not needed:
// the kernel map says which offsets are present.  One thread per output row; the map is read coalesced, the weights are wave-uniform.
template <int COUT>
__global__ void __launch_bounds__(256)
k_conv_unit(const int32_t* __restrict__ nbr, int K, int64_t n_out, const float* __restrict__ W, const float* __restrict__ bias,
            int relu, float* __restrict__ out, int out_ld) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    for (int k = 0; k < K; ++k) {
        if (nbr[(int64_t)k * n_out + o] < 0) continue;
        const float* w = W + (int64_t)k * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(1.0f, w[co], acc[co]);
    }
    float* y = out + o * out_ld;
#pragma unroll
    for (int co = 0; co < COUT; co += 4) {
        float4 v;
        v.x = acc[co] + (bias ? bias[co] : 0.0f); v.y = acc[co + 1] + (bias ? bias[co + 1] : 0.0f);
        v.z = acc[co + 2] + (bias ? bias[co + 2] : 0.0f); v.w = acc[co + 3] + (bias ? bias[co + 3] : 0.0f);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4*)(y + co) = v;
    }
}
// The same layer on a level whose kernel map has not been built: presence of offset k is derived on the fly from the PARENT level's map and
// the level pair's 8-slot down map, exactly as k_kmap_from_coarse would write it (coords.hip) — the [27][n] map of the finest encoder level
// (85 MB for a vox10 frame: written once, read once, by this layer only) never exists.
template <int COUT>
__global__ void __launch_bounds__(256)
k_conv_unit_coarse(const int4* __restrict__ fine, int64_t nf, int32_t stride_f, const int32_t* __restrict__ parent_of,
                   const int32_t* __restrict__ pnbr, const int32_t* __restrict__ down, int64_t nc, const float* __restrict__ W,
                   const float* __restrict__ bias, int relu, float* __restrict__ out, int out_ld) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nf) return;
    const int4 q = fine[c];
    const int j = ((q.y / stride_f) & 1) | (((q.z / stride_f) & 1) << 1) | (((q.w / stride_f) & 1) << 2);
    const int64_t p = parent_of[c];
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        bool present = true;
        if (k != 13) {
            int kp, jn; child_offset(j, k, kp, jn);
            const int32_t pn = pnbr[(int64_t)kp * nc + p];
            present = pn >= 0 && down[(int64_t)jn * nc + pn] >= 0;
        }
        if (!present) continue;
        const float* w = W + (int64_t)k * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(1.0f, w[co], acc[co]);
    }
    float* y = out + c * out_ld;
#pragma unroll
    for (int co = 0; co < COUT; co += 4) {
        float4 v;
        v.x = acc[co] + (bias ? bias[co] : 0.0f); v.y = acc[co + 1] + (bias ? bias[co + 1] : 0.0f);
        v.z = acc[co + 2] + (bias ? bias[co + 2] : 0.0f); v.w = acc[co + 3] + (bias ? bias[co + 3] : 0.0f);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4*)(y + co) = v;
    }
}
extern "C" int pcgc_conv_unit_from_coarse(const int32_t* fine, int64_t n_fine, int32_t stride_fine, const int32_t* parent_of,
                                          const int32_t* coarse_nbr, const int32_t* down, int64_t n_coarse, const float* W, const float* bias,
                                          int relu, float* out, int Cout, int out_ld, void* stream) {
    PCGC_REQUIRE(fine && parent_of && coarse_nbr && down && W && out, "null argument");
    PCGC_REQUIRE(stride_fine >= 1, "stride must be positive");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0, "output rows must be 16-byte aligned");
    if (n_fine == 0) return 0;
#define UNIT_COARSE(C_) hipLaunchKernelGGL((k_conv_unit_coarse<C_>), dim3(grid_for(n_fine, 256)), dim3(256), 0, S(stream), (const int4*)fine, n_fine, \
                                           stride_fine, parent_of, coarse_nbr, down, n_coarse, W, bias, relu, out, out_ld)
    switch (Cout) {
        case 16: UNIT_COARSE(16); break;                    // (autoencoder.py:88-94: the encoder's 1 -> 16 layer is the only unit-input conv)
        default: pcgc_set_error("conv_unit_from_coarse: unsupported Cout %d (16)", Cout); return -2;
    }
#undef UNIT_COARSE
    PCGC_CHECK_LAUNCH("conv_unit_from_coarse");
    return 0;
}
extern "C" int pcgc_conv_gather_unit(const int32_t* nbr, int K, int64_t n_out, const float* W, const float* bias, int relu,
                                     float* out, int Cout, int out_ld, void* stream) {
    PCGC_REQUIRE(nbr && W && out && K >= 1, "null argument");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0, "output rows must be 16-byte aligned");
    if (n_out == 0) return 0;
    switch (Cout) {
        case 16: hipLaunchKernelGGL((k_conv_unit<16>), dim3(grid_for(n_out, 256)), dim3(256), 0, S(stream), nbr, K, n_out, W, bias, relu, out, out_ld); break;
        default: pcgc_set_error("conv_gather_unit: unsupported Cout %d (16)", Cout); return -2;
    }
    PCGC_CHECK_LAUNCH("conv_gather_unit");
    return 0;
}

// ----------------------------------------------------------------------------------------------------------------
// v1 kernel: LDS-DMA gather.  One wave = one tile of 64 output rows x CT output channels (lane = row, accumulators in
// VGPRs, weights as scalar operands).  What changes vs v0 is how the gathered rows reach the lanes:
//   * per sub-step (offset k, <=32-channel block) the 64 neighbour rows are fetched by `buffer_load_dwordx4 ... lds`
//     with CH = CB/4 ADJACENT lanes per row, so one row is one contiguous 16*CH-byte request instead of CH separate
//     per-lane requests (v0 is bound by the texture-addresser request rate, not by FMAs).  Absent neighbours use an
//     out-of-range buffer offset: no fetch, EXEC stays full, the VMEM instruction count per sub-step is static, so the
//     DMA can be awaited with a counted `s_waitcnt vmcnt(1)` that leaves the NEXT offset's kernel-map load in flight;
//   * the kernel-map entry of each lane's own row lives in a VGPR (prefetched one offset ahead); the DMA lanes get
//     the entries of the rows they fetch by ds_bpermute — no LDS staging of the map;
//   * the LDS image is row-major [64][CH] 16-byte slots written lane-linear by the DMA; the SOURCE chunk index is
//     XOR-swizzled (slot p of row r holds chunk p ^ s(r), s(r) = (r*CH/16) % CH) and the per-lane row read-back
//     applies the same XOR => the ds_read_b128 of 64 different rows is bank-conflict free;
//   * only 1-8 KiB of LDS per wave and no intra-wave double buffering: latency is hidden by 4-8 waves per SIMD
//     (measured: the double-buffered variant with staged maps needed 15-23 KiB per wave and was DMA-latency bound).
// Numerics: identical fmaf chain (k ascending, ci ascending) — only the data path differs.
// ----------------------------------------------------------------------------------------------------------------
template <int CH, int ROWS = 64>
struct RowGather {                                   // one wave, ROWS (64/32/16) rows, CH 16-byte chunks per row
    static constexpr int RPI = 64 / CH;              // rows fetched by one DMA instruction
    static constexpr int NI = (ROWS + RPI - 1) / RPI;
    static constexpr int SH = (CH == 2) ? 3 : (CH == 4 ? 2 : 1);
    __device__ static inline void fetch(const __amdgpu_buffer_rsrc_t& rs, float4* rowbuf, int idx_cur, int in_ld, int col0, int lane) {
        const int dma_row_lo = lane / CH, dma_p = lane % CH;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = i * RPI + dma_row_lo;
            const int rid = r < ROWS ? __shfl(idx_cur, r, 64) : -1;
            const int chunk = dma_p ^ ((r >> SH) & (CH - 1));
            const unsigned voff = rid >= 0 ? (unsigned)(((int64_t)rid * in_ld + col0 + chunk * 4) * 4) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(rowbuf + i * 64), 16, (int)voff, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    __device__ static inline void read(const float4* rowbuf, int lane, float4 (&x)[CH]) {
        const int swz = (lane >> SH) & (CH - 1);
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = rowbuf[lane * CH + (c ^ swz)];
    }
    static constexpr int SLOTS = NI * 64;             // float4 slots one gather occupies in LDS
};

// acc[0..NO) += x4 (4 consecutive input channels) * w[4][ldw] rows, channel order preserved.
// Notes from measurements on MI355X (tools/ubench/fma_rate.hip): v_fma_f32 (vgpr) 100 TF, v_fmac_f32 with an SGPR
// operand 62 TF, v_pk_fma_f32 with an SGPR pair + op_sel broadcast 123 TF.  hipcc's SLP vectoriser turns these loops into
// v_pk_fma_f32 (+ two v_mov per instruction for the broadcast).  Two "obvious" improvements were tried and REJECTED:
// -fno-slp-vectorize (plain v_fmac with SGPR operands: 14.7 -> 16.8 ms per frame) and hand-emitted
// `v_pk_fma_f32 ... op_sel` via inline asm (the compiler then meters the weight s_loads in x8 chunks with a wait before
// every 4 FMAs: 14.7 -> 18.9 ms).  The compiler's own schedule (all s_load_dwordx16 of a sub-step up front) wins.
template <int NO>
__device__ static inline void fma4(float (&acc)[NO], const float4& x, const float* __restrict__ w, int ldw) {
#pragma unroll
    for (int co = 0; co < NO; ++co) acc[co] = fmaf(x.x, w[co], acc[co]);
#pragma unroll
    for (int co = 0; co < NO; ++co) acc[co] = fmaf(x.y, w[ldw + co], acc[co]);
#pragma unroll
    for (int co = 0; co < NO; ++co) acc[co] = fmaf(x.z, w[2 * ldw + co], acc[co]);
#pragma unroll
    for (int co = 0; co < NO; ++co) acc[co] = fmaf(x.w, w[3 * ldw + co], acc[co]);
}

// ----------------------------------------------------------------------------------------------------------------
// v2 kernel: LDS-DMA gather + fp32 MFMA channel GEMM, for Cin in {16,32,64} and Cout a multiple of 16.
// One wave = 64 output rows (4 M-tiles of 16) x CT = 16*NT output channels.  Per sub-step (offset k, 16 input channels):
//   * gather: 4 `buffer_load_dwordx4 ... lds` (4 adjacent lanes = one 64-byte row segment); a lane whose neighbour is
//     absent uses an out-of-range buffer offset, for which the LDS-DMA writes ZEROS into the lane's slot (the bounds-checked
//     load returns 0 and that is what lands in LDS — observed when such lanes overwrote live data, and relied on since:
//     every parity test runs through it).  MFMA cannot mask rows; fma(0, w, acc) == acc keeps the chain exact;
//   * A fragments: `v_mfma_f32_16x16x4_f32` wants A[row = lane&15][k = lane>>4].  Lane (i,q) reads the 16-byte chunk q of
//     row 16m+i with ONE ds_read_b128 — slot q ^ f(i>>2) of the source-swizzled image, f = (0,2,3,1), which puts the
//     four hardware lane groups of a b128 read on 16 distinct 16-byte slots (conflict-free) — and the 4x4 (lane-quarter x
//     component) transpose that turns "4 consecutive channels" into "channel 4j+q" is two v_permlane32_swap + two
//     v_permlane16_swap per M-tile;
//   * B fragments: W[k][16cb + 4j + q][16n + (lane&15)], one dword per lane per (j, n), straight from L1/L2;
//   * 16*NT MFMAs per sub-step, j outermost so consecutive MFMAs hit different accumulators (40-cycle dependent
//     latency vs 32-cycle issue); per accumulator the channel order stays ascending => bitwise the canonical fmaf chain.
// ----------------------------------------------------------------------------------------------------------------
template <int CIN, int NT>
__global__ void __launch_bounds__(256)
k_conv_gather_mfma(const int32_t* __restrict__ nbr, int K, int64_t n_out, const float* __restrict__ in, int64_t n_in,
                   int in_ld, const float* __restrict__ W, int Cout, const float* __restrict__ bias,
                   const float* __restrict__ res, int res_ld, int relu, float* __restrict__ out, int out_ld) {
    constexpr int NB = CIN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* rowbuf = (float4*)lds_raw + (size_t)wave * 256;            // [64 rows][4 slots]
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * 64;
    if (row0 >= n_out) return;
    const int co0 = blockIdx.y * (16 * NT);
    const int64_t my_row = row0 + lane;
    const bool valid = my_row < n_out;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(n_in * in_ld * 4), 0x00020000);

    const int mi = lane & 15, mq = lane >> 4;
    // f(0)=0, f(1)=2, f(2)=3, f(3)=1  ->  bits: f(3)f(2)f(1)f(0) = 01 11 10 00 = 0x78
    const int f_a = (0x78 >> (2 * (mi >> 2))) & 3;
    const int dma_row_lo = lane >> 2, dma_p = lane & 3;                // DMA: 16 rows per instruction, 4 lanes per row

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int idx_cur = valid ? nbr[my_row] : -1;
    for (int k = 0; k < K; ++k) {
        int idx_nxt = -1;
        if (k + 1 < K && valid) idx_nxt = nbr[(int64_t)(k + 1) * n_out + my_row];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            // ---- gather 64 rows x 16 channels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + dma_row_lo;                     // tile row; (r>>2)&3 == (dma_row_lo>>2)&3
                const int rid = __shfl(idx_cur, r, 64);
                const int chunk = dma_p ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3);
                const unsigned voff = rid >= 0 ? (unsigned)(((int64_t)rid * in_ld + cb * 16 + chunk * 4) * 4) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(rowbuf + i * 64), 16, (int)voff, 0, 0, 0);
            }
            // ---- B fragments for this (k, cb): [j][n]
            float b[4][NT];
            const float* wk = W + ((int64_t)k * CIN + cb * 16 + mq) * Cout + co0 + mi;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) b[j][n] = wk[(int64_t)(4 * j) * Cout + 16 * n];
            asm volatile("" ::: "memory");
            wait_vmcnt<0>();
            // ---- A fragments
            float4 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a[m] = rowbuf[(16 * m + mi) * 4 + (mq ^ f_a)];
                lane_transpose4(a[m]);
            }
            // ---- MFMA: j outermost, accumulators rotate
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[0][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[1][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b[2][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b[3][n], acc[m][n], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
        idx_cur = idx_nxt;
    }
    // ---- epilogue: D[row = 16m + 4*mq + r][col = 16n + mi]
    float bias_v[NT];                                          // (once per wave: a load inside the store loop is re-issued after every store)
#pragma unroll
    for (int n = 0; n < NT; ++n) bias_v[n] = bias ? bias[co0 + 16 * n + mi] : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 16 * m + 4 * mq + r;
            if (row >= n_out) continue;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = co0 + 16 * n + mi;
                float v = acc[m][n][r];
                if (bias) v = v + bias_v[n];
                if (res) v = v + res[row * res_ld + col];
                if (relu) v = fmaxf(v, 0.0f);
                out[row * out_ld + col] = v;
            }
        }
}

template <int CIN, int NT>
static void launch_mfma(const int32_t* nbr, int K, int64_t n_out, const float* in, int64_t n_in, int in_ld, const float* W,
                        int Cout, const float* bias, const float* res, int res_ld, int relu, float* out, int out_ld,
                        hipStream_t s) {
    hipLaunchKernelGGL((k_conv_gather_mfma<CIN, NT>), dim3(grid_for(n_out, 256), Cout / (16 * NT)), dim3(256), 4 * 4096, s, nbr,
                       K, n_out, in, n_in, in_ld, W, Cout, bias, res, res_ld, relu, out, out_ld);
}
template <int CIN>
static bool dispatch_mfma(int Cout, const int32_t* nbr, int K, int64_t n_out, const float* in, int64_t n_in, int in_ld,
                          const float* W, const float* bias, const float* res, int res_ld, int relu, float* out, int out_ld,
                          hipStream_t s) {
    // (one 16-column tile per wave at every size: the 32-column form only paid from 200 k rows on, where the rows / packed / children kernels run)
    if (Cout == 16 || Cout == 32 || Cout == 64) { launch_mfma<CIN, 1>(nbr, K, n_out, in, n_in, in_ld, W, Cout, bias, res, res_ld, relu, out, out_ld, s); return true; }
    return false;
}

// ----------------------------------------------------------------------------------------------------------------
// Fused InceptionResNet block (autoencoder.py:52-57) in two gather passes, built on the same LDS-DMA row gather:
//   A:  t[:, 0:Q]  = relu(conv0_0(x))   k3  C -> Q        t[:, Q:2Q] = relu(conv1_0(x))   k1  C -> Q   (Q = C/4)
//       the k1 conv reads the site's own row, which is exactly the gathered row of the centre offset (k = 13).
//   B:  out[:, 0:2Q]  = conv0_1(t[:, 0:Q]) + x[:, 0:2Q]                        k3  Q -> 2Q
//       out[:, 2Q:4Q] = conv1_2(relu(conv1_1(t[:, Q:2Q]))) + x[:, 2Q:4Q]       k3  Q -> Q, then k1  Q -> 2Q in registers
//       ONE gather of the 2Q-wide rows of t feeds both k3 convs (the unfused form gathers two Q-wide tensors).
// 5 launches / 3 gathers / 2 pointwise passes become 2 launches / 2 gathers; every fmaf chain is unchanged.
// ----------------------------------------------------------------------------------------------------------------
// Tile height: 16 rows per wave (round 5).  This pair serves what the rows / children kernels leave: levels below 1024 rows, children levels below
// 8192 rows — where the extra waves of short tiles win (18.7 k rows at C = 32: 64-row tiles 62.7 / 45.8 us, 16-row 43.7 / 32.8) — and, with
// the A/B switches off, any level as the comparison baseline.  The 32- / 64-row instantiations and the 16-channel sub-step form went with
// the levels they were tuned for.
// kernel offsets gathered per wait (27 = 9 x 3): more gathers in flight per wave.  Pays only while the extra row buffers do
// not cut occupancy: measured irn_b<16> 194 -> 167 us, but irn_b<32> 127 -> 160 us and irn_b<64> 196 -> 433 us with 3.
template <int C> struct IrnKG { static constexpr int value = (C == 16) ? 3 : 1; };       // pass B
template <int C> struct IrnKGA { static constexpr int value = 1; };                     // pass A: 227 vs 212 us with 3 at C=16

// CBMAX = channels gathered per sub-step (32, or 16): at C = 32 the 32-channel form holds 117 VGPRs (4 waves/SIMD); 16-channel
// sub-steps double the gather/wait steps but run 6 waves/SIMD — faster on the big level (570 k rows: 206 -> 175 us), slower on
// the small ones (256 k: 68 -> 74 us), so launch_irn picks by level size.
template <int C, int ROWS, int CBMAX = 32, int KG = IrnKGA<C>::value>
__global__ void __launch_bounds__(256)
k_irn_a(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ x, int x_ld,
        const float* __restrict__ W00, const float* __restrict__ b00, const float* __restrict__ W10,
        const float* __restrict__ b10, float* __restrict__ t /*[n, C/2]*/) {
    constexpr int Q = C / 4;
    constexpr int CB = C < CBMAX ? C : CBMAX, NB = C / CB, CH = CB / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    using RG = RowGather<CH, ROWS>;
    float4* rowbuf = (float4*)lds_raw + (size_t)wave * (KG * RG::SLOTS);
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * ROWS;
    if (row0 >= n) return;
    const int64_t my_row = row0 + lane;
    const bool valid = lane < ROWS && my_row < n;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(n * x_ld * 4), 0x00020000);
    __attribute__((aligned(8))) float acc0[Q];
    __attribute__((aligned(8))) float acc1[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    static_assert(KG == 1 || NB == 1, "offset groups need the whole row in one sub-step");
    if constexpr (KG > 1) {
        int idx[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) idx[g] = valid ? nbr[(int64_t)g * n + my_row] : -1;
        for (int k0 = 0; k0 < 27; k0 += KG) {
#pragma unroll
            for (int g = 0; g < KG; ++g) RG::fetch(rs, rowbuf + g * RG::SLOTS, idx[g], x_ld, 0, lane);
            int idx_n[KG];
#pragma unroll
            for (int g = 0; g < KG; ++g) idx_n[g] = (valid && k0 + KG + g < 27) ? nbr[(int64_t)(k0 + KG + g) * n + my_row] : -1;
            asm volatile("" ::: "memory");
            static_assert(27 % KG == 0, "offset groups must tile the 27 offsets");
            if (k0 + KG < 27) wait_vmcnt<KG>(); else wait_vmcnt<0>();       // last group: no prefetches, all DMAs must land
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                float4 xv[CH];
                RG::read(rowbuf + g * RG::SLOTS, lane, xv);
                if (idx[g] >= 0) {
                    const int k = k0 + g;
                    const float* w = W00 + (int64_t)k * C * Q;
#pragma unroll
                    for (int c = 0; c < CH; ++c) fma4<Q>(acc0, xv[c], w + (4 * c) * Q, Q);
                    if (k == 13) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) fma4<Q>(acc1, xv[c], W10 + (4 * c) * Q, Q);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < KG; ++g) idx[g] = idx_n[g];
        }
    } else {
    int idx_cur = valid ? nbr[my_row] : -1;
    for (int k = 0; k < 27; ++k) {
        int idx_nxt = -1;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            RG::fetch(rs, rowbuf, idx_cur, x_ld, cb * CB, lane);
            if (cb == NB - 1 && k + 1 < 27) {
                if (valid) idx_nxt = nbr[(int64_t)(k + 1) * n + my_row];
                asm volatile("" ::: "memory");
                wait_vmcnt<1>();
            } else wait_vmcnt<0>();
            float4 xv[CH];
            RG::read(rowbuf, lane, xv);
            if (idx_cur >= 0) {
                const float* w = W00 + ((int64_t)k * C + cb * CB) * Q;
#pragma unroll
                for (int c = 0; c < CH; ++c) fma4<Q>(acc0, xv[c], w + (4 * c) * Q, Q);
                if (k == 13) {                                       // own row: the k1 branch conv1_0
                    const float* w1 = W10 + (int64_t)(cb * CB) * Q;
#pragma unroll
                    for (int c = 0; c < CH; ++c) fma4<Q>(acc1, xv[c], w1 + (4 * c) * Q, Q);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        idx_cur = idx_nxt;
    }
    }
    if (!valid) return;
    float* y = t + my_row * (2 * Q);
#pragma unroll
    for (int i = 0; i < Q; ++i) y[i] = fmaxf(acc0[i] + b00[i], 0.0f);
#pragma unroll
    for (int i = 0; i < Q; ++i) y[Q + i] = fmaxf(acc1[i] + b10[i], 0.0f);
}

template <int C, int ROWS, int KG = IrnKG<C>::value>
__global__ void __launch_bounds__(256)
k_irn_b(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ t /*[n, C/2]*/, const float* __restrict__ x,
        int x_ld, const float* __restrict__ W01, const float* __restrict__ b01, const float* __restrict__ W11,
        const float* __restrict__ b11, const float* __restrict__ W12, const float* __restrict__ b12,
        float* __restrict__ out, int out_ld) {
    constexpr int Q = C / 4, H = C / 2;
    constexpr int CH = H / 4;                        // gathered row = H floats: 2, 4 or 8 chunks
    constexpr int CQ = Q / 4;                        // chunks per branch: 1, 2 or 4
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    using RG = RowGather<CH, ROWS>;
    float4* rowbuf = (float4*)lds_raw + (size_t)wave * (KG * RG::SLOTS);
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * ROWS;
    if (row0 >= n) return;
    const int64_t my_row = row0 + lane;
    const bool valid = lane < ROWS && my_row < n;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)t, 0, (int)(n * H * 4), 0x00020000);
    __attribute__((aligned(8))) float acc0[H];
    __attribute__((aligned(8))) float acc1[Q];
#pragma unroll
    for (int i = 0; i < H; ++i) acc0[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < Q; ++i) acc1[i] = 0.0f;
    // KG kernel offsets per wait: KG row buffers, KG gathers in flight per wave (27 = 9 x 3)
    int idx[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) idx[g] = valid ? nbr[(int64_t)g * n + my_row] : -1;
    for (int k0 = 0; k0 < 27; k0 += KG) {
#pragma unroll
        for (int g = 0; g < KG; ++g) RG::fetch(rs, rowbuf + g * RG::SLOTS, idx[g], H, 0, lane);
        int idx_n[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) idx_n[g] = (valid && k0 + KG + g < 27) ? nbr[(int64_t)(k0 + KG + g) * n + my_row] : -1;
        asm volatile("" ::: "memory");
        // counted wait: only the map prefetches issued AFTER the DMAs may stay in flight.  27 % KG == 0, so a group issues
        // either all KG prefetches or (the last group) none — in which case every outstanding VMEM op is a DMA: wait for all.
        static_assert(27 % KG == 0, "offset groups must tile the 27 offsets");
        if (k0 + KG < 27) wait_vmcnt<KG>(); else wait_vmcnt<0>();
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            float4 tv[CH];
            RG::read(rowbuf + g * RG::SLOTS, lane, tv);
            if (idx[g] >= 0) {
                const int k = k0 + g;
                const float* w0 = W01 + (int64_t)k * Q * H;          // [Q][H]
                const float* w1 = W11 + (int64_t)k * Q * Q;          // [Q][Q]
#pragma unroll
                for (int c = 0; c < CQ; ++c) fma4<H>(acc0, tv[c], w0 + (4 * c) * H, H);
#pragma unroll
                for (int c = 0; c < CQ; ++c) fma4<Q>(acc1, tv[CQ + c], w1 + (4 * c) * Q, Q);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < KG; ++g) idx[g] = idx_n[g];
    }
    if (!valid) return;
    // conv1_2 (k1, Q -> H) on u = relu(conv1_1 + bias), in registers
    float acc2[H];
#pragma unroll
    for (int i = 0; i < H; ++i) acc2[i] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < Q; ++ci) {
        const float u = fmaxf(acc1[ci] + b11[ci], 0.0f);
#pragma unroll
        for (int co = 0; co < H; ++co) acc2[co] = fmaf(u, W12[ci * H + co], acc2[co]);
    }
    const float* xr = x + my_row * x_ld;
    float* y = out + my_row * out_ld;
#pragma unroll
    for (int co = 0; co < H; co += 4) {
        float4 r0 = *(const float4*)(xr + co), r1 = *(const float4*)(xr + H + co), o0, o1;
        o0.x = (acc0[co] + b01[co]) + r0.x; o0.y = (acc0[co + 1] + b01[co + 1]) + r0.y;
        o0.z = (acc0[co + 2] + b01[co + 2]) + r0.z; o0.w = (acc0[co + 3] + b01[co + 3]) + r0.w;
        o1.x = (acc2[co] + b12[co]) + r1.x; o1.y = (acc2[co + 1] + b12[co + 1]) + r1.y;
        o1.z = (acc2[co + 2] + b12[co + 2]) + r1.z; o1.w = (acc2[co + 3] + b12[co + 3]) + r1.w;
        *(float4*)(y + co) = o0; *(float4*)(y + H + co) = o1;
    }
}

// phase: 1 = pass A only, 2 = pass B only, 3 = both.  ROWS = rows per wave (64 by default; 32 / 16 exist for A/B tests: the
// idea was to give small levels more waves per SIMD — a 71 k-row level is only 1.1 waves per SIMD with 64-row tiles).
// ----------------------------------------------------------------------------------------------------------------
// Row-split kernels for levels of a few ten thousand rows (16 output rows per wave, 1-2 workgroups per CU).  With lane = row
// such a wave keeps 16 lanes busy and streams every weight through the scalar cache (27.6 KB per pass against a 16 KB cache:
// each s_load batch pays an L2 round trip with nothing to hide it — 40 us for 18.7 k rows).  Here the four 16-lane groups of
// a wave split the OUTPUT channels of the same 16 rows (lane = (row, part)), and the weights are restaged once per workgroup
// into LDS as [k][part][ci][channels of the part] (+16 bytes per block: the four parts read four different bank groups).
// Every output element still sees ci ascending inside k ascending.
// ----------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
struct SplitW {
    static constexpr int P = CO / 4;                 // output channels per lane
    static constexpr int BLK = CI * P + 4;           // floats per (k, part) block
    static_assert(CO % 4 == 0 && (CI * P) % 4 == 0, "parts are whole float4 runs");
    static constexpr int floats(int K) { return K * 4 * BLK; }
    __device__ static inline void stage(float* lds, const float* __restrict__ W, int K, int tid, int nthreads) {
        for (int e = tid; e < K * CI * CO; e += nthreads) {
            const int k = e / (CI * CO), rem = e % (CI * CO), ci = rem / CO, co = rem % CO;
            lds[(k * 4 + co / P) * BLK + ci * P + co % P] = W[e];
        }
    }
    // acc[j] = fmaf(x[ci], W[k][ci][part * P + j], acc[j]), ci ascending; x = CI floats held as float4 chunks
    __device__ static inline void fma(float (&acc)[P], const float4* xv, const float* blk) {
#pragma unroll
        for (int m = 0; m < CI * P / 4; ++m) {
            const float4 w = *(const float4*)(blk + 4 * m);
            const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * m + i, ci = e / P, j = e % P;
                const float4 xc = xv[ci / 4];
                const float xs = (ci % 4 == 0) ? xc.x : (ci % 4 == 1) ? xc.y : (ci % 4 == 2) ? xc.z : xc.w;
                acc[j] = fmaf(xs, wv[i], acc[j]);
            }
        }
    }
};
// offsets in flight per wave: 9 while weights + the four waves' row buffers leave two workgroups per CU, else 3
template <int WFLOATS, int SLOTS> struct SplitKG { static constexpr int value = (WFLOATS * 4 + 4 * 9 * SLOTS * 16 <= 80 * 1024) ? 9 : 3; };

template <int CIN, int CT>
__global__ void __launch_bounds__(256)
k_conv_gather_split(const int32_t* __restrict__ nbr, int64_t n_out, const float* __restrict__ in, int64_t n_in, int in_ld,
                    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ res, int res_ld,
                    int relu, float* __restrict__ out, int out_ld) {
    constexpr int ROWS = 16, CH = CIN / 4, P = CT / 4;
    using RG = RowGather<CH, ROWS>;
    using SW = SplitW<CIN, CT>;
    constexpr int KG = SplitKG<SW::floats(27), RG::SLOTS>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* wl = (float*)lds_raw;
    const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* rowbuf = (float4*)(wl + SW::floats(27)) + (size_t)wave * (KG * RG::SLOTS);
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * ROWS;
    const int64_t my_row = row0 + r;
    const bool valid = my_row < n_out;
    int idx[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) idx[g] = valid ? nbr[(int64_t)g * n_out + my_row] : -1;
    SW::stage(wl, W, 27, threadIdx.x, 256);
    __syncthreads();
    if (row0 >= n_out) return;                        // (after the barrier: every wave takes part in the staging)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(n_in * in_ld * 4), 0x00020000);
    float acc[P];
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = 0.0f;
    for (int k0 = 0; k0 < 27; k0 += KG) {
#pragma unroll
        for (int g = 0; g < KG; ++g) RG::fetch(rs, rowbuf + g * RG::SLOTS, idx[g], in_ld, 0, lane);
        int idx_n[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) idx_n[g] = (valid && k0 + KG + g < 27) ? nbr[(int64_t)(k0 + KG + g) * n_out + my_row] : -1;
        asm volatile("" ::: "memory");
        static_assert(27 % KG == 0, "offset groups must tile the 27 offsets");
        if (k0 + KG < 27) wait_vmcnt<KG>(); else wait_vmcnt<0>();
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            float4 xv[CH];
            RG::read(rowbuf + g * RG::SLOTS, r, xv);
            if (idx[g] >= 0) SW::fma(acc, xv, wl + ((k0 + g) * 4 + part) * SW::BLK);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < KG; ++g) idx[g] = idx_n[g];
    }
    if (!valid) return;
    float* y = out + my_row * out_ld + part * P;
    const float* rr = res ? res + my_row * res_ld + part * P : nullptr;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        float v = acc[j];
        if (bias) v = v + bias[part * P + j];
        if (rr) v = v + rr[j];
        if (relu) v = fmaxf(v, 0.0f);
        y[j] = v;
    }
}

template <int C>
__global__ void __launch_bounds__(256)
k_irn_a_split(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ x, int x_ld, const float* __restrict__ W00,
              const float* __restrict__ b00, const float* __restrict__ W10, const float* __restrict__ b10, float* __restrict__ t) {
    constexpr int ROWS = 16, Q = C / 4, P = Q / 4, CH = C / 4;
    using RG = RowGather<CH, ROWS>;
    using SW = SplitW<C, Q>;
    constexpr int KG = SplitKG<SW::floats(28), RG::SLOTS>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* w00 = (float*)lds_raw;
    float* w10 = w00 + SW::floats(27);
    const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* rowbuf = (float4*)(w00 + SW::floats(28)) + (size_t)wave * (KG * RG::SLOTS);
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * ROWS;
    const int64_t my_row = row0 + r;
    const bool valid = my_row < n;
    int idx[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) idx[g] = valid ? nbr[(int64_t)g * n + my_row] : -1;
    SW::stage(w00, W00, 27, threadIdx.x, 256);
    SW::stage(w10, W10, 1, threadIdx.x, 256);
    __syncthreads();
    if (row0 >= n) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(n * x_ld * 4), 0x00020000);
    float acc0[P], acc1[P];
#pragma unroll
    for (int j = 0; j < P; ++j) { acc0[j] = 0.0f; acc1[j] = 0.0f; }
    for (int k0 = 0; k0 < 27; k0 += KG) {
#pragma unroll
        for (int g = 0; g < KG; ++g) RG::fetch(rs, rowbuf + g * RG::SLOTS, idx[g], x_ld, 0, lane);
        int idx_n[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) idx_n[g] = (valid && k0 + KG + g < 27) ? nbr[(int64_t)(k0 + KG + g) * n + my_row] : -1;
        asm volatile("" ::: "memory");
        static_assert(27 % KG == 0, "offset groups must tile the 27 offsets");
        if (k0 + KG < 27) wait_vmcnt<KG>(); else wait_vmcnt<0>();
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            float4 xv[CH];
            RG::read(rowbuf + g * RG::SLOTS, r, xv);
            if (idx[g] >= 0) {
                SW::fma(acc0, xv, w00 + ((k0 + g) * 4 + part) * SW::BLK);
                if (k0 + g == 13) SW::fma(acc1, xv, w10 + part * SW::BLK);        // own row: the k1 branch conv1_0
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < KG; ++g) idx[g] = idx_n[g];
    }
    if (!valid) return;
    float* y = t + my_row * (2 * Q) + part * P;
#pragma unroll
    for (int j = 0; j < P; ++j) y[j] = fmaxf(acc0[j] + b00[part * P + j], 0.0f);
#pragma unroll
    for (int j = 0; j < P; ++j) y[Q + j] = fmaxf(acc1[j] + b10[part * P + j], 0.0f);
}

template <int C>
__global__ void __launch_bounds__(256)
k_irn_b_split(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ t /*[n, C/2]*/, const float* __restrict__ x,
              int x_ld, const float* __restrict__ W01, const float* __restrict__ b01, const float* __restrict__ W11,
              const float* __restrict__ b11, const float* __restrict__ W12, const float* __restrict__ b12,
              float* __restrict__ out, int out_ld) {
    constexpr int ROWS = 16, Q = C / 4, H = C / 2, CH = H / 4, CQ = Q / 4, HP = H / 4, QP = Q / 4;
    using RG = RowGather<CH, ROWS>;
    using SW0 = SplitW<Q, H>;                        // conv0_1: t[:, 0:Q]  -> H
    using SW1 = SplitW<Q, Q>;                        // conv1_1: t[:, Q:2Q] -> Q
    constexpr int KG = SplitKG<SW0::floats(27) + SW1::floats(27), RG::SLOTS>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* w01 = (float*)lds_raw;
    float* w11 = w01 + SW0::floats(27);
    const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* rowbuf = (float4*)(w11 + SW1::floats(27)) + (size_t)wave * (KG * RG::SLOTS);
    const int64_t row0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * ROWS;
    const int64_t my_row = row0 + r;
    const bool valid = my_row < n;
    int idx[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) idx[g] = valid ? nbr[(int64_t)g * n + my_row] : -1;
    SW0::stage(w01, W01, 27, threadIdx.x, 256);
    SW1::stage(w11, W11, 27, threadIdx.x, 256);
    __syncthreads();
    if (row0 >= n) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)t, 0, (int)(n * H * 4), 0x00020000);
    float acc0[HP], acc1[QP];
#pragma unroll
    for (int j = 0; j < HP; ++j) acc0[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < QP; ++j) acc1[j] = 0.0f;
    for (int k0 = 0; k0 < 27; k0 += KG) {
#pragma unroll
        for (int g = 0; g < KG; ++g) RG::fetch(rs, rowbuf + g * RG::SLOTS, idx[g], H, 0, lane);
        int idx_n[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) idx_n[g] = (valid && k0 + KG + g < 27) ? nbr[(int64_t)(k0 + KG + g) * n + my_row] : -1;
        asm volatile("" ::: "memory");
        static_assert(27 % KG == 0, "offset groups must tile the 27 offsets");
        if (k0 + KG < 27) wait_vmcnt<KG>(); else wait_vmcnt<0>();
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            float4 tv[CH];
            RG::read(rowbuf + g * RG::SLOTS, r, tv);
            if (idx[g] >= 0) {
                SW0::fma(acc0, tv, w01 + ((k0 + g) * 4 + part) * SW0::BLK);
                SW1::fma(acc1, tv + CQ, w11 + ((k0 + g) * 4 + part) * SW1::BLK);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < KG; ++g) idx[g] = idx_n[g];
    }
    // conv1_2 (k1, Q -> H) on u = relu(conv1_1 + bias): every part needs the row's Q values of u, held QP per part
    float u[QP];
#pragma unroll
    for (int j = 0; j < QP; ++j) u[j] = fmaxf(acc1[j] + b11[part * QP + j], 0.0f);
    float acc2[HP];
#pragma unroll
    for (int j = 0; j < HP; ++j) acc2[j] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < Q; ++ci) {
        const float uc = __shfl(u[ci % QP], r + 16 * (ci / QP), 64);
#pragma unroll
        for (int j = 0; j < HP; ++j) acc2[j] = fmaf(uc, W12[ci * H + part * HP + j], acc2[j]);
    }
    if (!valid) return;
    const float* xr = x + my_row * x_ld + part * HP;
    float* y = out + my_row * out_ld + part * HP;
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        y[j] = (acc0[j] + b01[part * HP + j]) + xr[j];
        y[H + j] = (acc2[j] + b12[part * HP + j]) + xr[H + j];
    }
}

// dynamic LDS above the default limit: raise the attribute once per (kernel, device)
static int irn_lds_limit(const void* kern, size_t lds, size_t (&granted)[16]) {
    if (lds <= 48 * 1024) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > granted[dev & 15]) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        granted[dev & 15] = lds;
    }
    return 0;
}
template <int C> struct IrnSplit {
    static constexpr int Q = C / 4, H = C / 2;
    using RGA = RowGather<C / 4, 16>;
    using RGB = RowGather<H / 4, 16>;
    static constexpr int WA = SplitW<C, Q>::floats(28), WB = SplitW<Q, H>::floats(27) + SplitW<Q, Q>::floats(27);
    static constexpr size_t LDS_A = (size_t)WA * 4 + 4 * (size_t)(SplitKG<WA, RGA::SLOTS>::value * RGA::SLOTS * 16);
    static constexpr size_t LDS_B = (size_t)WB * 4 + 4 * (size_t)(SplitKG<WB, RGB::SLOTS>::value * RGB::SLOTS * 16);
};
template <int CIN, int CT>
static int launch_split(const int32_t* nbr, int64_t n_out, const float* in, int64_t n_in, int in_ld, const float* W, const float* bias,
                        const float* res, int res_ld, int relu, float* out, int out_ld, hipStream_t s) {
    using RG = RowGather<CIN / 4, 16>;
    constexpr int WF = SplitW<CIN, CT>::floats(27);
    constexpr size_t lds = (size_t)WF * 4 + 4 * (size_t)(SplitKG<WF, RG::SLOTS>::value * RG::SLOTS * 16);
    static size_t granted[16] = {0};
    auto kern = k_conv_gather_split<CIN, CT>;
    if (irn_lds_limit((const void*)kern, lds, granted)) { pcgc_set_error("conv_gather split: cannot raise the LDS limit to %zu", lds); return -1; }
    hipLaunchKernelGGL(kern, dim3(grid_for(n_out, 64)), dim3(256), lds, s, nbr, n_out, in, n_in, in_ld, W, bias, res, res_ld, relu, out, out_ld);
    return 0;
}
// -> 0 launched, 1 shape not covered, < 0 error
template <int CIN>
static int dispatch_split(int Cout, const int32_t* nbr, int64_t n_out, const float* in, int64_t n_in, int in_ld, const float* W,
                          const float* bias, const float* res, int res_ld, int relu, float* out, int out_ld, hipStream_t s) {
    switch (Cout) {
        case 4: return launch_split<CIN, 4>(nbr, n_out, in, n_in, in_ld, W, bias, res, res_ld, relu, out, out_ld, s);
        case 8: return launch_split<CIN, 8>(nbr, n_out, in, n_in, in_ld, W, bias, res, res_ld, relu, out, out_ld, s);
        case 16: return launch_split<CIN, 16>(nbr, n_out, in, n_in, in_ld, W, bias, res, res_ld, relu, out, out_ld, s);
    }
    return 1;
}
// 16-row tiles are what levels of a few ten thousand rows get (1-2 workgroups per CU): nothing hides the gather latency
// there, so those kernels keep 9 kernel offsets in flight per wave (three waits per pass instead of 27; the LDS is free).
template <int C, int ROWS> struct IrnBurst {
    static constexpr int A = (ROWS == 16 && C <= 32) ? 9 : IrnKGA<C>::value;      // pass A needs the whole row in one sub-step
    static constexpr int B = (ROWS == 16) ? 9 : IrnKG<C>::value;
};
template <int C>
static int launch_irn(const int32_t* nbr, int64_t n, const float* x, int x_ld, const float* const* P, float* t, float* out,
                       int out_ld, int phase, hipStream_t s) {
    constexpr int ROWS = 16;
    const dim3 grid(grid_for(n, 4 * ROWS));
    if constexpr (C <= 32) {                             // lane = (row, output-channel part), weights in LDS
        if (phase & 1) {
            static size_t granted[16] = {0};
            auto kern = k_irn_a_split<C>;
            if (irn_lds_limit((const void*)kern, IrnSplit<C>::LDS_A, granted)) { pcgc_set_error("irn pass A: cannot raise the LDS limit"); return -1; }
            hipLaunchKernelGGL(kern, grid, dim3(256), IrnSplit<C>::LDS_A, s, nbr, n, x, x_ld, P[0], P[1], P[4], P[5], t);
        }
        if (phase & 2) {
            static size_t granted[16] = {0};
            auto kern = k_irn_b_split<C>;
            if (irn_lds_limit((const void*)kern, IrnSplit<C>::LDS_B, granted)) { pcgc_set_error("irn pass B: cannot raise the LDS limit"); return -1; }
            hipLaunchKernelGGL(kern, grid, dim3(256), IrnSplit<C>::LDS_B, s, nbr, n, t, x, x_ld, P[2], P[3], P[6], P[7], P[8], P[9], out, out_ld);
        }
        return 0;
    } else {                                             // C = 64: lane = row, weights as scalar operands
        constexpr int KGA = IrnBurst<C, ROWS>::A, KGB = IrnBurst<C, ROWS>::B;
        const size_t lds_a = 4 * (size_t)(KGA * RowGather<32 / 4, ROWS>::SLOTS * 16);
        const size_t lds_b = 4 * (size_t)(KGB * RowGather<C / 8, ROWS>::SLOTS * 16);
        if (phase & 1) {
            static size_t granted[16] = {0};
            auto kern = k_irn_a<C, ROWS, 32, KGA>;
            if (irn_lds_limit((const void*)kern, lds_a, granted)) { pcgc_set_error("irn pass A: cannot raise the LDS limit to %zu", lds_a); return -1; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_a, s, nbr, n, x, x_ld, P[0], P[1], P[4], P[5], t);
        }
        if (phase & 2) {
            static size_t granted[16] = {0};
            auto kern = k_irn_b<C, ROWS, KGB>;
            if (irn_lds_limit((const void*)kern, lds_b, granted)) { pcgc_set_error("irn pass B: cannot raise the LDS limit to %zu", lds_b); return -1; }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds_b, s, nbr, n, t, x, x_ld, P[2], P[3], P[6], P[7], P[8], P[9], out, out_ld);
        }
        return 0;
    }
}

// params: {W00,b00, W01,b01, W10,b10, W11,b11, W12,b12} = conv0_0, conv0_1, conv1_0, conv1_1, conv1_2 (kernel, bias)
static int irn_launch(const int32_t* nbr, int64_t n, const float* x, int C, int x_ld, const float* const* params,
                      float* t_scratch, float* out, int out_ld, int phase, void* stream);
extern "C" int pcgc_irn_block(const int32_t* nbr, int64_t n, const float* x, int C, int x_ld, const float* const* params,
                              float* t_scratch, float* out, int out_ld, void* stream) {
    return irn_launch(nbr, n, x, C, x_ld, params, t_scratch, out, out_ld, 3, stream);
}
extern "C" int pcgc_irn_pass(const int32_t* nbr, int64_t n, const float* x, int C, int x_ld, const float* const* params,
                             float* t_scratch, float* out, int out_ld, int pass, void* stream) {
    PCGC_REQUIRE(pass == 1 || pass == 2, "pass must be 1 (A) or 2 (B)");
    return irn_launch(nbr, n, x, C, x_ld, params, t_scratch, out, out_ld, pass, stream);
}
static int irn_launch(const int32_t* nbr, int64_t n, const float* x, int C, int x_ld, const float* const* params,
                      float* t_scratch, float* out, int out_ld, int phase, void* stream) {
    PCGC_REQUIRE(nbr && x && params && t_scratch && out, "null argument");
    PCGC_REQUIRE(C == 16 || C == 32 || C == 64, "channels must be 16, 32 or 64");
    PCGC_REQUIRE((x_ld & 3) == 0 && (out_ld & 3) == 0, "leading dimensions must be multiples of 4");
    PCGC_REQUIRE(n * (int64_t)x_ld * 4 < (int64_t)0xFFFFFFF0, "tensor too large for 32-bit buffer offsets");
    for (int i = 0; i < 10; ++i) PCGC_REQUIRE(params[i] != nullptr, "null parameter tensor");
    PCGC_REQUIRE((((uintptr_t)x | (uintptr_t)t_scratch | (uintptr_t)out) & 15) == 0, "buffers must be 16-byte aligned");
    if (n == 0) return 0;
    int rc;
    if (C == 16) rc = launch_irn<16>(nbr, n, x, x_ld, params, t_scratch, out, out_ld, phase, S(stream));
    else if (C == 32) rc = launch_irn<32>(nbr, n, x, x_ld, params, t_scratch, out, out_ld, phase, S(stream));
    else rc = launch_irn<64>(nbr, n, x, x_ld, params, t_scratch, out, out_ld, phase, S(stream));
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("irn_block");
    return 0;
}

// kernel selection for pcgc_conv_gather (all families are bit-identical; tests run every one of them): -1 auto | 0 direct loads + VALU |
// 2 LDS-DMA + MFMA | 6 row-split
static int g_conv_impl = -1;
extern "C" int pcgc_set_conv_impl(int impl) { g_conv_impl = impl; return 0; }
// Which kernel family the calling thread's last pcgc_conv_gather launched: 0 = VALU, direct loads | 2 = MFMA, LDS-DMA row gather, weights
// from L2 | 6 = row-split (16-row tiles, lanes split the output channels).  (Round 5 removed the LDS-DMA + VALU forms 1 / 5 and the
// LDS-shared-weight MFMA forms 3 / 4 / 7: every level they were tuned for runs on the rows / packed / children kernels)
// The size / shape policy lives in ONE table, pcgcv2_amd/dispatch.py; the GPU tests read this back for every entry of that table, on
// both sides of every gate, and compare it with the table's prediction.
static thread_local int t_last_conv_impl = -1;
extern "C" int pcgc_last_conv_impl(void) { return t_last_conv_impl; }

extern "C" int pcgc_conv_gather(const int32_t* nbr, int K, int64_t n_out, const float* in, int64_t n_in, int Cin, int in_ld,
                                int in_coff, const float* W, const float* bias, const float* residual, int res_ld, int res_coff,
                                int relu, float* out, int Cout, int out_ld, int out_coff, void* stream) {
    PCGC_REQUIRE(K >= 1 && Cin >= 1, "bad K / Cin");
    PCGC_REQUIRE(nbr != nullptr || K == 1, "identity map only for K == 1");
    PCGC_REQUIRE(g_conv_impl == -1 || g_conv_impl == 0 || g_conv_impl == 2 || g_conv_impl == 6, "conv_gather: forced family must be -1 (auto), 0, 2 or 6");
    if (n_out == 0) return 0;
    hipStream_t s = S(stream);
    // the LDS-DMA row gather: gathered maps, 8..64 input channels, 16-byte aligned rows, 32-bit buffer offsets
    const float* in0 = in + in_coff;
    const bool aligned = (((uintptr_t)in0 | (uintptr_t)W) & 15) == 0 && (in_ld & 3) == 0;
    const bool small = n_in * in_ld * 4 < (int64_t)0xFFFFFFF0 && (int64_t)K * n_out * 4 < (int64_t)0xFFFFFFF0;
    const bool dma_eligible = nbr != nullptr && K <= 27 && aligned && small && (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64);
    const bool mfma_eligible = dma_eligible && (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 16 || Cout == 32 || Cout == 64);
    // narrow outputs: the row-split kernel (16 -> 16 at 1-18 k rows: 12-16 us against 27 on the MFMA form; 32 -> 8: 16-20 against 40;
    // 32 -> 8 at 64-71 k rows: 38-49 against 68-73).  Cout <= 8 has no MFMA form: row-split at every size.
    const bool split_shape = dma_eligible && K == 27 && Cin <= 32 && Cout <= 16 && (Cout & 3) == 0;
    const bool split_first = g_conv_impl < 0 && split_shape && (Cout <= 8 || n_out < 40000);
    // MFMA-sized channels from 512 rows on (64 -> 32 at 1-18 k rows: VALU 178-220 us, MFMA 81; the k2 s2 down convs 64 -> 32 / 32 -> 64
    // on an octant block's 3-10 k rows: VALU 93 us, down2 64 -> 32 at 18.7 k rows: 58 -> 38)
    if (mfma_eligible && !split_first && (g_conv_impl == 2 || (g_conv_impl < 0 && n_out >= 512))) {
        const float* res0 = residual ? residual + res_coff : nullptr;
        float* out0 = out + out_coff;
        bool ok = false;
        if (Cin == 16) ok = dispatch_mfma<16>(Cout, nbr, K, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        else if (Cin == 32) ok = dispatch_mfma<32>(Cout, nbr, K, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        else ok = dispatch_mfma<64>(Cout, nbr, K, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        if (ok) { t_last_conv_impl = 2; PCGC_CHECK_LAUNCH("conv_gather_mfma"); return 0; }
    }
    if (split_shape && (g_conv_impl == 6 || split_first)) {
        const float* res0 = residual ? residual + res_coff : nullptr;
        float* out0 = out + out_coff;
        int rc = 1;
        if (Cin == 8) rc = dispatch_split<8>(Cout, nbr, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        else if (Cin == 16) rc = dispatch_split<16>(Cout, nbr, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        else if (Cin == 32) rc = dispatch_split<32>(Cout, nbr, n_out, in0, n_in, in_ld, W, bias, res0, res_ld, relu, out0, out_ld, s);
        if (rc < 0) return rc;
        if (rc == 0) { t_last_conv_impl = 6; PCGC_CHECK_LAUNCH("conv_gather_split"); return 0; }
    }
    t_last_conv_impl = 0;
#define PCGC_CASE(C) case C: launch_valu<C>(nbr, K, n_out, in, Cin, in_ld, in_coff, W, bias, residual, res_ld, res_coff, relu, out, out_ld, out_coff, s); break;
    switch (Cout) {
        PCGC_CASE(1) PCGC_CASE(4) PCGC_CASE(8) PCGC_CASE(16) PCGC_CASE(32) PCGC_CASE(64)
        default: pcgc_set_error("conv_gather: unsupported Cout %d (1,4,8,16,32,64)", Cout); return -2;
    }
#undef PCGC_CASE
    PCGC_CHECK_LAUNCH("conv_gather");
    return 0;
}

// ----------------------------------------------------------------------------------------------------------------
// Generative transpose k2 s2: out[8i+k] = in[i] @ W[k] + bias.  blockIdx.y = k so the weight slice is wave-uniform.
// ----------------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ void __launch_bounds__(256)
k_conv_up2(int64_t n_in, const float* __restrict__ in, int Cin, int in_ld, const float* __restrict__ W,
           const float* __restrict__ bias, int relu, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (i >= n_in) return;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    const float* x = in + i * in_ld;
    const float* w = W + (int64_t)k * Cin * COUT;
    for (int ci = 0; ci < Cin; ++ci) {
        float a = x[ci];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a, w[ci * COUT + co], acc[co]);
    }
    float* y = out + (8 * i + k) * (int64_t)COUT;
#pragma unroll
    for (int co = 0; co < COUT; co += 4) {
        float4 v;
        v.x = acc[co] + (bias ? bias[co] : 0.0f); v.y = acc[co + 1] + (bias ? bias[co + 1] : 0.0f);
        v.z = acc[co + 2] + (bias ? bias[co + 2] : 0.0f); v.w = acc[co + 3] + (bias ? bias[co + 3] : 0.0f);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4*)(y + co) = v;
    }
}
// Coalesced form for the model's three shapes: one thread per (output row, 4-column chunk); consecutive lanes store
// consecutive 16-byte chunks (pure streaming write, the op is write-bandwidth bound: 8N*Cout*4 bytes), the 8 weight
// slices sit in LDS and the parent row is a broadcast read.
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
k_conv_up2_rows(int64_t n_in, const float* __restrict__ in, int in_ld, const float* __restrict__ W,
                const float* __restrict__ bias, int relu, float* __restrict__ out) {
    extern __shared__ float4 w_lds[];                       // [8][CIN][COUT/4]
    constexpr int CH = COUT / 4;
    for (int t = threadIdx.x; t < 8 * CIN * CH; t += 256) w_lds[t] = ((const float4*)W)[t];
    __syncthreads();
    const int64_t total = 8 * n_in * CH;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        int64_t r = t / CH; int c = (int)(t % CH);
        int64_t i = r >> 3; int k = (int)(r & 7);
        const float4* x = (const float4*)(in + i * in_ld);
        const float4* w = w_lds + (k * CIN) * CH + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int q = 0; q < CIN / 4; ++q) {
            float4 xv = x[q];
            float4 w0 = w[(4 * q + 0) * CH], w1 = w[(4 * q + 1) * CH], w2 = w[(4 * q + 2) * CH], w3 = w[(4 * q + 3) * CH];
            acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
            acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
            acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
            acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
        }
        if (bias) { float4 b = ((const float4*)bias)[c]; acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
        if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        ((float4*)out)[t] = acc;
    }
}
template <int CIN, int COUT>
static int launch_up2_rows(int64_t n_in, const float* in, int in_ld, const float* W, const float* bias, int relu, float* out,
                           hipStream_t s) {
    size_t lds = (size_t)8 * CIN * COUT * sizeof(float);
    int64_t total = 8 * n_in * (COUT / 4);
    unsigned g = grid_for(total, 256); if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL((k_conv_up2_rows<CIN, COUT>), dim3(g), dim3(256), lds, s, n_in, in, in_ld, W, bias, relu, out);
    return 0;
}

// fp32 MFMA form for the two large up-convs (64->32 on 8*N4', 32->16 on 8*N2'): out[8p + k] = in[p] @ W[k] is eight
// [N x Cin] @ [Cin x Cout] GEMMs that share the A operand.  One wave = 16*MT parent rows: the A fragments (lane (i,q) loads the
// 16-byte chunk q of row 16m+i of each 16-channel block and gets "channel 4j+q" by the 4x4 lane transpose, as in the gather
// kernels) are loaded ONCE and stay in registers for all eight k; B fragments W[k][16cb + 4j + q][16n + i] come from L1/L2.
// Channel order per accumulator: cb, j, q ascending = 0..Cin-1: bitwise the canonical chain.  The VALU form above issues
// Cin*4 dependent FMAs per 16-byte store and was compute-, not write-bound (123 us for 131 MB at 32->16).
template <int CIN, int COUT, int MT>
__global__ void __launch_bounds__(256)
k_conv_up2_mfma(int64_t n_in, const float* __restrict__ in, int in_ld, const int32_t* __restrict__ rows, const float* __restrict__ W,
                const float* __restrict__ bias, int relu, float* __restrict__ out) {
    constexpr int NB = CIN / 16, NT = COUT / 16;
    __shared__ __attribute__((aligned(16))) float stage_all[4][2 * 16 * (COUT + 4)];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* stage = stage_all[wave];
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * MT);
    if (p0 >= n_in) return;
    const int mi = lane & 15, mq = lane >> 4;
    float4 a[MT][NB];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int64_t p = p0 + 16 * m + mi;
        const int64_t src = (rows && p < n_in) ? (int64_t)rows[p] : p;        // (rows: input row p is row rows[p] of `in` — a pruned level read in place)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            a[m][cb] = p < n_in ? *(const float4*)(in + src * in_ld + 16 * cb + 4 * mq) : make_float4(0.f, 0.f, 0.f, 0.f);
            lane_transpose4(a[m][cb]);
        }
    }
    for (int k = 0; k < 8; ++k) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wk = W + ((int64_t)k * CIN + mq) * COUT + mi;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            float b[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) b[j][n] = wk[(16 * cb + 4 * j) * COUT + 16 * n];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float av = j == 0 ? a[m][cb].x : (j == 1 ? a[m][cb].y : (j == 2 ? a[m][cb].z : a[m][cb].w));
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j][n], acc[m][n], 0, 0, 0);
                    }
        }
        // Epilogue through a per-wave LDS scratch, two k at a time: from the MFMA layout (lane = column, 4 rows per lane) a row
        // would reach memory as 4-byte pieces; staged, the 2 * COUT floats of (parent, k, k + 1) are contiguous in memory and
        // leave as 16-byte-per-lane stores (rows padded by 4 floats in LDS: the four lane quarters write four different bank groups).
        static_assert(MT == 1, "the staged epilogue covers one M tile per wave");
        constexpr int LDW = COUT + 4, F4 = COUT / 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = 16 * n + mi;
                float v = acc[0][n][r];
                if (bias) v = v + bias[col];
                if (relu) v = fmaxf(v, 0.0f);
                stage[((k & 1) * 16 + 4 * mq + r) * LDW + col] = v;
            }
        if (k & 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < (16 * 2 * F4) / 64; ++it) {
                const int f = lane + 64 * it, pr = f / (2 * F4), rem = f % (2 * F4), kk = rem / F4, c4 = rem % F4;
                const float4 v = *(const float4*)(stage + (kk * 16 + pr) * LDW + 4 * c4);
                if (p0 + pr < n_in) *(float4*)(out + (8 * (p0 + pr) + (k - 1) + kk) * COUT + 4 * c4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
template <int CIN, int COUT>
static int launch_up2_mfma(int64_t n_in, const float* in, int in_ld, const int32_t* rows, const float* W, const float* bias, int relu, float* out,
                           hipStream_t s) {
    // 16 parent rows per wave: these levels are small (71 k / 256 k parents) and a wave runs its eight k serially, so more, shorter
    // waves win (measured us for 64->32 / 32->16: 16 rows 58 / 58, 32 rows 70 / 59, 64 rows 99 / 68; VALU form 97 / 123)
    hipLaunchKernelGGL((k_conv_up2_mfma<CIN, COUT, 1>), dim3(grid_for(n_in, 64)), dim3(256), 0, s, n_in, in, in_ld, rows, W, bias, relu, out);
    return 0;
}
// Third form (round 6): the same GEMMs with the eight weight slices staged ONCE per persistent workgroup in LDS as lane-linear B fragments —
// fragment (k, n, cb): lane (mi, mq) holds W[k][16 cb + 4 jj + mq][16 n + mi], jj = 0..3: one ds_read_b128 per four MFMAs — where the form above
// reads 4 NB NT dwords per k and wave from L2 (a load latency in front of every k's MFMAs: 52 us for 2.3 GFLOP and 73 MB of output at 64 -> 32,
// 0.27 of the matrix peak at issued / algorithmic 1.0), and FOUR k per store phase: the 4 COUT floats of (parent, k .. k + 3) are contiguous in
// memory (512 bytes at 64 -> 32).  Same products in the same order per output element (cb, jj, K index ascending): bit-identical.
template <int CIN, int COUT, int NW>
__global__ void __launch_bounds__(NW * 64)
k_conv_up2_tab(int64_t n_in, const float* __restrict__ in, int in_ld, const int32_t* __restrict__ rows, const float* __restrict__ W,
               const float* __restrict__ bias, int relu, float* __restrict__ out) {
    constexpr int NB = CIN / 16, NT = COUT / 16, NFRAG = 8 * NT * NB, LDW = COUT + 4, F4 = COUT / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char up2_lds[];
    float4* tab = (float4*)up2_lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, mq = lane >> 4;
    // W [8][CIN][COUT] read linearly (16 bytes per thread, every load independent of the others), scattered into the fragments: element
    // (k, ci, co) -> fragment (k, co / 16, ci / 16), lane (mi = co % 16, mq = ci % 4), component (ci % 16) / 4
    for (int e = threadIdx.x; e < 8 * CIN * F4; e += NW * 64) {
        const float4 w = ((const float4*)W)[e];
        const int c4 = e % F4, ci = (e / F4) % CIN, k = e / (F4 * CIN), co = 4 * c4;
        float* dst = (float*)tab + ((((k * NT + co / 16) * NB + ci / 16) * 64 + (ci & 3) * 16 + (co & 15)) * 4 + ((ci & 15) >> 2));
        dst[0] = w.x; dst[4] = w.y; dst[8] = w.z; dst[12] = w.w;
    }
    __syncthreads();
    float* stage = (float*)(up2_lds + NFRAG * 1024) + wave * (4 * 16 * LDW);
    float bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[n] = bias ? bias[16 * n + mi] : 0.0f;
    const int64_t ntiles = (n_in + 15) >> 4;
    const int64_t tstep = (int64_t)gridDim.x * NW;
    // the tile's rows (lane (mi, mq): chunk mq of row mi of each 16-channel block), requested one tile ahead: a wave runs ~2 tiles and two waves
    // share a SIMD, so a load latency in front of every tile would be a fifth of its time
    auto fetch = [&](int64_t tile, float4 (&raw)[NB]) {
        const int64_t p = tile * 16 + mi;
        const bool ok = tile < ntiles && p < n_in;
        const int64_t src = (rows && ok) ? (int64_t)rows[p] : p;                  // (rows: input row p is row rows[p] of `in` — a pruned level read in place)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) raw[cb] = ok ? *(const float4*)(in + src * in_ld + 16 * cb + 4 * mq) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 nxt[NB];
    fetch((int64_t)blockIdx.x * NW + wave, nxt);
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += tstep) {
        const int64_t p0 = tile * 16;
        float4 a[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) { a[cb] = nxt[cb]; lane_transpose4(a[cb]); }
        fetch(tile + tstep, nxt);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 4 * kh + kk;
                f32x4 acc[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    float4 b[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[n] = tab[((k * NT + n) * NB + cb) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const float av = j == 0 ? a[cb].x : (j == 1 ? a[cb].y : (j == 2 ? a[cb].z : a[cb].w));
                            const float bw = j == 0 ? b[n].x : (j == 1 ? b[n].y : (j == 2 ? b[n].z : b[n].w));
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc[n], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        float v = acc[n][r];
                        if (bias) v = v + bv[n];
                        if (relu) v = fmaxf(v, 0.0f);
                        stage[(kk * 16 + 4 * mq + r) * LDW + 16 * n + mi] = v;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < (16 * 4 * F4) / 64; ++it) {
                const int f = lane + 64 * it, pr = f / (4 * F4), rem = f % (4 * F4), kk = rem / F4, c4 = rem % F4;
                const float4 v = *(const float4*)(stage + (kk * 16 + pr) * LDW + 4 * c4);
                if (p0 + pr < n_in) *(float4*)(out + (8 * (p0 + pr) + 4 * kh) * COUT + 4 * rem) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
template <int CIN, int COUT>
static int launch_up2_tab(int64_t n_in, const float* in, int in_ld, const int32_t* rows, const float* W, const float* bias, int relu, float* out,
                          hipStream_t s) {
    constexpr int NW = 8, NB = CIN / 16, NT = COUT / 16;
    constexpr size_t lds = (size_t)8 * NT * NB * 1024 + (size_t)NW * 4 * 16 * (COUT + 4) * 4;
    static_assert(lds <= 160 * 1024, "table + staging fit one workgroup");
    auto kern = k_conv_up2_tab<CIN, COUT, NW>;
    static bool granted[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 48 * 1024 && !granted[dev & 15]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pcgc_set_error("conv_up2: cannot raise the LDS limit to %zu: %s", lds, hipGetErrorString(e)); return -1; }
        granted[dev & 15] = true;
    }
    static int cus = 0;
    if (!cus) { hipDeviceProp_t p; cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    const int64_t ntiles = (n_in + 15) >> 4;
    const int per_cu = (int)((160 * 1024) / lds) > 2 ? 2 : ((160 * 1024) / lds < 1 ? 1 : (int)((160 * 1024) / lds));      // (16 waves per CU at most)
    int64_t g = (ntiles + NW - 1) / NW;
    if (g > (int64_t)cus * per_cu) g = (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), lds, s, n_in, in, in_ld, rows, W, bias, relu, out);
    return 0;
}
static int g_up2_mfma = 2;          // 2 = LDS-resident fragment table (default), 1 = fragments from L2, 0 = VALU form (A/B tests)
extern "C" int pcgc_set_up2_impl(int mfma) { g_up2_mfma = mfma; return 0; }

// the same on a PRUNED level read in place: input row p = row rows[p] of `in` (rows = the survivors' candidate rows, pcgc_topk_select) —
// the compacted feature tensor of the pruned level is never written.  Only the shapes of the two large decoder stages; -3 = not one of them
// (the caller gathers the rows first and calls pcgc_conv_up2).
extern "C" int pcgc_conv_up2_gather(int64_t n_in, const float* in, int Cin, int in_ld, const int32_t* rows, const float* W, const float* bias,
                                    int relu, float* out, int Cout, void* stream) {
    PCGC_REQUIRE(rows != nullptr, "null row list");
    if (n_in == 0) return 0;
    if (!g_up2_mfma || (in_ld & 3) != 0 || (((uintptr_t)in | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias) & 15) != 0) return -3;
    int rc = 0;
    if (Cin == 64 && Cout == 32) rc = g_up2_mfma == 2 ? launch_up2_tab<64, 32>(n_in, in, in_ld, rows, W, bias, relu, out, S(stream)) : launch_up2_mfma<64, 32>(n_in, in, in_ld, rows, W, bias, relu, out, S(stream));
    else if (Cin == 32 && Cout == 16) rc = g_up2_mfma == 2 ? launch_up2_tab<32, 16>(n_in, in, in_ld, rows, W, bias, relu, out, S(stream)) : launch_up2_mfma<32, 16>(n_in, in, in_ld, rows, W, bias, relu, out, S(stream));
    else return -3;
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("conv_up2_gather");
    return 0;
}
extern "C" int pcgc_conv_up2(int64_t n_in, const float* in, int Cin, int in_ld, const float* W, const float* bias, int relu,
                             float* out, int Cout, void* stream) {
    if (n_in == 0) return 0;
    if ((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias) & 15) == 0) {
        bool done = true;
        if (Cin == 8 && Cout == 64) launch_up2_rows<8, 64>(n_in, in, in_ld, W, bias, relu, out, S(stream));
        else if (g_up2_mfma == 2 && Cin == 64 && Cout == 32) { if (int rc = launch_up2_tab<64, 32>(n_in, in, in_ld, nullptr, W, bias, relu, out, S(stream))) return rc; }
        else if (g_up2_mfma == 2 && Cin == 32 && Cout == 16) { if (int rc = launch_up2_tab<32, 16>(n_in, in, in_ld, nullptr, W, bias, relu, out, S(stream))) return rc; }
        else if (g_up2_mfma && Cin == 64 && Cout == 32) launch_up2_mfma<64, 32>(n_in, in, in_ld, nullptr, W, bias, relu, out, S(stream));
        else if (g_up2_mfma && Cin == 32 && Cout == 16) launch_up2_mfma<32, 16>(n_in, in, in_ld, nullptr, W, bias, relu, out, S(stream));
        else if (Cin == 64 && Cout == 32) launch_up2_rows<64, 32>(n_in, in, in_ld, W, bias, relu, out, S(stream));
        else if (Cin == 32 && Cout == 16) launch_up2_rows<32, 16>(n_in, in, in_ld, W, bias, relu, out, S(stream));
        else done = false;
        if (done) { PCGC_CHECK_LAUNCH("conv_up2_rows"); return 0; }
    }
    dim3 g(grid_for(n_in, 256), 8), b(256);
    switch (Cout) {
        case 16: hipLaunchKernelGGL((k_conv_up2<16>), g, b, 0, S(stream), n_in, in, Cin, in_ld, W, bias, relu, out); break;
        case 32: hipLaunchKernelGGL((k_conv_up2<32>), g, b, 0, S(stream), n_in, in, Cin, in_ld, W, bias, relu, out); break;
        case 64: hipLaunchKernelGGL((k_conv_up2<64>), g, b, 0, S(stream), n_in, in, Cin, in_ld, W, bias, relu, out); break;
        case 4: hipLaunchKernelGGL((k_conv_up2<4>), g, b, 0, S(stream), n_in, in, Cin, in_ld, W, bias, relu, out); break;
        case 8: hipLaunchKernelGGL((k_conv_up2<8>), g, b, 0, S(stream), n_in, in, Cin, in_ld, W, bias, relu, out); break;
        default: pcgc_set_error("conv_up2: unsupported Cout %d (4,8,16,32,64)", Cout); return -2;
    }
    PCGC_CHECK_LAUNCH("conv_up2");
    return 0;
}
