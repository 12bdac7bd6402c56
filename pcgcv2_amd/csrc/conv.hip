// Sparse convolution family: MinkowskiConvolution (k3 s1, k2 s2, k1) and MinkowskiGenerativeConvolutionTranspose
// (k2 s2) of the reference's autoencoder.py, as output-stationary gather kernels.
//
// Canonical arithmetic (identical to oracle/pcgc_oracle.c):
//   acc = +0 ; for k ascending, for ci ascending: acc = fmaf(in[nbr[k][o]][ci], W[k][ci][co], acc)
//   out = acc + bias ; out += residual ; out = relu(out)
// Output-stationary => deterministic, no atomics, and the k-ascending order survives any tiling.
#include "pcgc_common.h"

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

extern "C" int pcgc_conv_gather(const int32_t* nbr, int K, int64_t n_out, const float* in, int Cin, int in_ld, int in_coff,
                                const float* W, const float* bias, const float* residual, int res_ld, int res_coff, int relu,
                                float* out, int Cout, int out_ld, int out_coff, void* stream) {
    PCGC_REQUIRE(K >= 1 && Cin >= 1, "bad K / Cin");
    PCGC_REQUIRE(nbr != nullptr || K == 1, "identity map only for K == 1");
    if (n_out == 0) return 0;
    hipStream_t s = S(stream);
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

extern "C" int pcgc_conv_up2(int64_t n_in, const float* in, int Cin, int in_ld, const float* W, const float* bias, int relu,
                             float* out, int Cout, void* stream) {
    if (n_in == 0) return 0;
    if ((in_ld & 3) == 0 && (((uintptr_t)in | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias) & 15) == 0) {
        bool done = true;
        if (Cin == 8 && Cout == 64) launch_up2_rows<8, 64>(n_in, in, in_ld, W, bias, relu, out, S(stream));
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
