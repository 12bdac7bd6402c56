// Factorized entropy bottleneck on device (reference entropy_model.py:82-196): quantisation, symbol range, and the
// fused CDF table (likelihood MLP -> clamp -> cumsum -> torchac 16-bit normalisation).  The table is 8 x (L+1)
// values: the point of doing it on device is that neither the latents nor an [N8,8,L+1] expansion
// (entropy_model.py:173) ever exist; only int16 symbols and the 16-bit table cross PCIe.
#include "pcgc_common.h"

__device__ static inline int32_t f2ord(float f) { int32_t b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ static inline float ord2f(int32_t o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// single block (count is N8*8 ~ 1e5..1e6): grid-stride inside the block, LDS tree for the 16 waves
__global__ void __launch_bounds__(1024) k_round_minmax(const float* __restrict__ f, int64_t count, float* minmax) {
    __shared__ int32_t smin[16], smax[16];
    int32_t lo = 0x7fffffff, hi = (int32_t)0x80000000;
    auto take = [&](float x) { const int32_t o = f2ord(rintf(x) + 0.0f); lo = min(lo, o); hi = max(hi, o); };
    const bool vec = (((uintptr_t)f) & 15) == 0;          // 16-byte loads: a quarter of the dependent iterations of this one block
    const int64_t n4 = vec ? count / 4 : 0;
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = ((const float4*)f)[i];
        take(v.x); take(v.y); take(v.z); take(v.w);
    }
    for (int64_t i = 4 * n4 + threadIdx.x; i < count; i += 1024) take(f[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, 64)); hi = max(hi, __shfl_xor(hi, d, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { lo = min(lo, smin[i]); hi = max(hi, smax[i]); }
        minmax[0] = ord2f(lo); minmax[1] = ord2f(hi);
    }
}
__global__ void k_symbolize(const float* __restrict__ f, int64_t count, float min_v, int16_t* __restrict__ sym) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) sym[i] = (int16_t)(rintf(f[i]) - min_v);
}
__global__ void k_desymbolize(const int16_t* __restrict__ sym, int64_t count, float min_v, float* __restrict__ f) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) f[i] = (float)sym[i] + min_v;
}
extern "C" int pcgc_round_minmax(const float* feats, int64_t count, float* minmax, void* stream) {
    PCGC_REQUIRE(count > 0, "empty latent");
    hipLaunchKernelGGL(k_round_minmax, dim3(1), dim3(1024), 0, S(stream), feats, count, minmax);
    PCGC_CHECK_LAUNCH("round_minmax");
    return 0;
}

extern "C" int pcgc_symbolize(const float* feats, int64_t count, float min_v, int16_t* sym, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_symbolize, dim3(grid_for(count, 256)), dim3(256), 0, S(stream), feats, count, min_v, sym);
    PCGC_CHECK_LAUNCH("symbolize");
    return 0;
}
extern "C" int pcgc_desymbolize(const int16_t* sym, int64_t count, float min_v, float* feats, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_desymbolize, dim3(grid_for(count, 256)), dim3(256), 0, S(stream), sym, count, min_v, feats);
    PCGC_CHECK_LAUNCH("desymbolize");
    return 0;
}

// ---- pointwise operators of the unfused ME-style graph (MinkowskiReLU, SparseTensor.__add__: autoencoder.py:50,55).  The product's
//      own forward passes fuse both into the producing conv's epilogue; these exist for code written against the ME operator surface
//      (pcgcv2_amd/ME.py).  Same arithmetic as the fused epilogues: max(v, 0) and one fp32 add.
__global__ void k_relu(const float* __restrict__ in, int64_t count, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = fmaxf(in[i], 0.0f);
}
__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, int64_t count, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = a[i] + b[i];
}
extern "C" int pcgc_relu(const float* in, int64_t count, float* out, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_relu, dim3(grid_for(count, 256)), dim3(256), 0, S(stream), in, count, out);
    PCGC_CHECK_LAUNCH("relu");
    return 0;
}
extern "C" int pcgc_add(const float* a, const float* b, int64_t count, float* out, void* stream) {
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_add, dim3(grid_for(count, 256)), dim3(256), 0, S(stream), a, b, count, out);
    PCGC_CHECK_LAUNCH("add");
    return 0;
}

// ---- fused CDF table ----------------------------------------------------------------------------------------
// params packing (352 floats for C=8): matrices 0..3 [C,fo,fi] | biases 0..3 [C,fo,1] | factors 0..3 [C,fo,1],
// filters (1,3,3,3,1).  Evaluated in fp64 from the fp32 parameters; rounded to fp32 where the reference holds fp32
// tensors feeding a discontinuity (likelihood, clamp, running cumsum).
__device__ static inline double eb_softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ static inline double eb_sigmoid(double x) { return x >= 0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x)); }
// The parameter-only factors of the 4-layer chain — softplus(matrix) (24 per channel) and tanh(factor) (10 per channel) —
// are evaluated once per block into LDS; a table entry then costs 2 x 10 tanh + 2 sigmoid instead of 2 x 68 fp64
// transcendentals (the two table kernels sit on the critical path of encode and decode: 42 -> ~12 us each).  Same values,
// same operation order as evaluating them in place.
constexpr int EB_MAX_C = 16;
struct EbShared { double sp[EB_MAX_C * 24]; double tf[EB_MAX_C * 10]; };
__device__ static void eb_prepare(const float* __restrict__ P, int C, EbShared& sh) {
    const float* M = P; const float* Fa = P + 24 * C + 10 * C;
    const int F[5] = {1, 3, 3, 3, 1};
    for (int e = threadIdx.x; e < C * 34; e += blockDim.x) {
        const int c = e / 34, r = e % 34;
        if (r < 24) {                                   // matrix entry: local index -> (layer i, position inside the layer)
            int i = r < 3 ? 0 : (r < 12 ? 1 : (r < 21 ? 2 : 3));
            const int loff = i == 0 ? 0 : (i == 1 ? 3 : (i == 2 ? 12 : 21));
            int moff = 0;
            for (int j = 0; j < i; ++j) moff += C * F[j + 1] * F[j];
            sh.sp[c * 24 + r] = eb_softplus((double)M[moff + c * F[i + 1] * F[i] + (r - loff)]);
        } else {
            const int q = r - 24;                       // factor entry 0..9: layers of 3, 3, 3, 1
            const int i = q < 3 ? 0 : (q < 6 ? 1 : (q < 9 ? 2 : 3));
            const int boff = C * 3 * i;
            sh.tf[c * 10 + q] = tanh((double)Fa[boff + c * F[i + 1] + (q - 3 * i)]);
        }
    }
}
__device__ static double eb_logits(const float* __restrict__ P, int C, int c, double v, const EbShared& sh) {
    const int F[5] = {1, 3, 3, 3, 1};
    const float* B = P + 24 * C;
    const double* sp = sh.sp + c * 24; const double* tf = sh.tf + c * 10;
    double h[3] = {v, 0, 0}, t[3];
    int loff = 0, boff = 0, foff = 0;
    for (int i = 0; i < 4; ++i) {
        int fi = F[i], fo = F[i + 1];
        const float* b = B + boff + c * fo;
        for (int r = 0; r < fo; ++r) {
            double s = 0;
            for (int q = 0; q < fi; ++q) s += sp[loff + r * fi + q] * h[q];
            s += (double)b[r];
            s += tf[foff + r] * tanh(s);
            t[r] = s;
        }
        for (int r = 0; r < fo; ++r) h[r] = t[r];
        loff += fo * fi; boff += C * fo; foff += fo;
    }
    return h[0];
}
// phase 1: likelihood(c, s) -> cdf_f32[c][s+1] (temporarily holds the clamped pmf)
__global__ void k_cdf_likelihood(const float* __restrict__ P, int C, int L, float min_v, float* __restrict__ cdf_f32) {
    __shared__ EbShared sh;
    eb_prepare(P, C, sh);
    __syncthreads();
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * L) return;
    int c = t / L, s = t % L;
    double v = (double)min_v + s;
    double lo = eb_logits(P, C, c, v - 0.5, sh), up = eb_logits(P, C, c, v + 0.5, sh);
    double sum = lo + up, sign = sum > 0 ? -1.0 : (sum < 0 ? 1.0 : 0.0);
    float p = (float)fabs(eb_sigmoid(sign * up) - eb_sigmoid(sign * lo));
    cdf_f32[c * (L + 1) + s + 1] = p < 1e-9f ? 1e-9f : p;
}
// phase 2: one thread per channel: fp32 running sum (torch.cumsum on fp32), clamp(max=1), 16-bit normalisation
__global__ void k_cdf_finish(int C, int L, float* __restrict__ cdf_f32, uint16_t* __restrict__ cdf_u16) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float* row = cdf_f32 + c * (L + 1);
    uint16_t* q = cdf_u16 + c * (L + 1);
    const float scale = 65536.0f - (float)L;            // 2^16 - (Lp - 1), Lp = L + 1
    float run = 0.0f;
    row[0] = 0.0f; q[0] = 0;
    for (int s = 1; s <= L; ++s) {
        run = run + row[s];
        float v = run > 1.0f ? 1.0f : run;
        row[s] = v;
        q[s] = (uint16_t)((int32_t)rintf(v * scale) + s);
    }
}
extern "C" int pcgc_cdf_table(const float* params, int C, float min_v, float max_v, uint16_t* cdf_u16, float* cdf_f32,
                              void* stream) {
    PCGC_REQUIRE(max_v >= min_v, "max_v < min_v");
    PCGC_REQUIRE(cdf_f32 != nullptr, "cdf_f32 scratch/output buffer is required");
    int L = (int)(max_v - min_v) + 1;
    PCGC_REQUIRE(L >= 1 && L < 32768, "symbol alphabet out of int16 range");
    PCGC_REQUIRE(C >= 1 && C <= EB_MAX_C, "entropy bottleneck: at most 16 channels");
    hipLaunchKernelGGL(k_cdf_likelihood, dim3(grid_for((int64_t)C * L, 256)), dim3(256), 0, S(stream), params, C, L, min_v, cdf_f32);
    hipLaunchKernelGGL(k_cdf_finish, dim3(1), dim3(64), 0, S(stream), C, L, cdf_f32, cdf_u16);
    PCGC_CHECK_LAUNCH("cdf_table");
    return 0;
}

// ---- device-resident symbol range: min / max -> symbols is enqueued without a host round trip; the host then fetches {minmax, symbols}
// with one synchronising copy (the CDF table is evaluated on the host in the reference's arithmetic, csrc/reftable.cpp).
__global__ void k_symbolize_dev(const float* __restrict__ f, int64_t count, const float* __restrict__ minmax, int16_t* __restrict__ sym) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) sym[i] = (int16_t)(rintf(f[i]) - minmax[0]);
}
// symbol range + symbols in one enqueue (the host-table path of compress(): the CDF table is evaluated on the host)
extern "C" int pcgc_quantize_symbols(const float* feats, int64_t count, float* minmax, int16_t* sym, void* stream) {
    PCGC_REQUIRE(count > 0, "empty latent");
    hipLaunchKernelGGL(k_round_minmax, dim3(1), dim3(1024), 0, S(stream), feats, count, minmax);
    hipLaunchKernelGGL(k_symbolize_dev, dim3(grid_for(count, 256)), dim3(256), 0, S(stream), feats, count, minmax, sym);
    PCGC_CHECK_LAUNCH("quantize_symbols");
    return 0;
}
// Batched form: segment b = the rows of batch item b (contiguous), each with its OWN symbol range — the reference codes every cloud
// with its own (min_v, max_v) header.  minmax: [nseg][2] on the device; seg_rows: host array.
extern "C" int pcgc_quantize_symbols_segments(const float* feats, int C, int nseg, const int64_t* seg_rows, float* minmax, int16_t* sym,
                                              void* stream) {
    PCGC_REQUIRE(nseg >= 1 && seg_rows && C >= 1, "bad segments");
    int64_t off = 0;
    for (int b = 0; b < nseg; ++b) {
        PCGC_REQUIRE(seg_rows[b] > 0, "empty latent in a batch item");
        const int rc = pcgc_quantize_symbols(feats + off * C, seg_rows[b] * C, minmax + 2 * b, sym + off * C, stream);
        if (rc) return rc;
        off += seg_rows[b];
    }
    return 0;
}
