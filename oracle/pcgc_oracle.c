/*
 * pcgc_oracle.c — CPU restatement (ORACLE) of the PCGCv2 encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pcgcv2_amd/) never imports, links or executes anything under oracle/.
 *
 * Parity status
 *   - entropy tables (orc_cdf_float)      pinned to golden G1 (tests/golden/entropy_tables.npz, generated from
 *                                         /root/reference/entropy_model.py:82-149 by tests/golden/make_golden.py)
 *   - ordering / top-k / PLY / D1         pinned to golden G2..G4 (python side, oracle/pcgc_oracle.py)
 *   - sparse conv family, range coder     "parity unpinned": MinkowskiEngine >=0.5 and torchac 0.9.3 are
 *                                         un-vendored third-party dependencies that are not installable here;
 *                                         their published algorithms are restated below and anchored on the
 *                                         reference's call sites (cited per function).
 *
 * Canonical arithmetic (shared contract with the HIP path):
 *   out[o][co] = ( fmaf-chain over k ascending, ci ascending of in[nbr[k][o]][ci] * W[k][ci][co], start +0.0f )
 *                + bias[co]
 *   absent neighbours (nbr < 0) are skipped.  fp32 throughout, one rounding per product-accumulate (fmaf).
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* OpenMP thread count of the row-parallel loops (bench.py's cpu_baseline reports an all-cores and a 1-thread figure). */
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * coordinate keys.  coords are int32 [N,4] = (batch, x, y, z) as ME.SparseTensor.C
 * (reference call sites: data_utils.py:107-108, coder.py:102).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int64_t hi, lo; } ckey_t;                 /* full-width fields: (batch, z+1) | (y+1, x+1) */
static inline ckey_t pack_key(int32_t b, int32_t x, int32_t y, int32_t z) {
    ckey_t k;
    k.hi = ((int64_t)b << 32) | (int64_t)(uint32_t)(z + 1);
    k.lo = ((int64_t)(uint32_t)(y + 1) << 32) | (int64_t)(uint32_t)(x + 1);
    return k;
}
static inline int key_cmp(ckey_t a, ckey_t b) {
    if (a.hi != b.hi) return a.hi < b.hi ? -1 : 1;
    if (a.lo != b.lo) return a.lo < b.lo ? -1 : 1;
    return 0;
}
static inline int key_eq(ckey_t a, ckey_t b) { return a.hi == b.hi && a.lo == b.lo; }

typedef struct { ckey_t key; int32_t row; } kv_t;

static int kv_cmp(const void* a, const void* b) {
    const kv_t* p = (const kv_t*)a; const kv_t* q = (const kv_t*)b;
    int c = key_cmp(p->key, q->key);
    if (c) return c;
    return p->row < q->row ? -1 : (p->row > q->row);
}

static kv_t* build_sorted(const int32_t* coords, int64_t n) {
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* c = coords + 4 * i;
        kv[i].key = pack_key(c[0], c[1], c[2], c[3]); kv[i].row = (int32_t)i;
    }
    qsort(kv, (size_t)n, sizeof(kv_t), kv_cmp);
    return kv;
}

static int32_t find_row(const kv_t* kv, int64_t n, ckey_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (key_cmp(kv[m].key, key) < 0) lo = m + 1; else hi = m; }
    return (lo < n && key_eq(kv[lo].key, key)) ? kv[lo].row : -1;
}

/* ME.SparseTensor construction dedups coordinates (data_utils.py:108,116; coder.py:102).  ‡ ME keeps one row
 * per unique coordinate; canonical convention here: keep the FIRST occurrence, preserve input order.
 * keep[i] = 1 if row i survives.  returns the number of survivors. */
int64_t orc_unique_first(const int32_t* coords, int64_t n, uint8_t* keep) {
    kv_t* kv = build_sorted(coords, n);
    int64_t cnt = 0;
    memset(keep, 0, (size_t)n);
    for (int64_t i = 0; i < n; ++i)
        if (i == 0 || !key_eq(kv[i].key, kv[i - 1].key)) { keep[kv[i].row] = 1; ++cnt; }
    free(kv);
    return cnt;
}

/* Output coordinates of MinkowskiConvolution(kernel_size=2, stride=2) (autoencoder.py:78-84,97-103,116-122):
 * ‡ unique(floor(c / (2s)) * 2s).  Canonical order: first occurrence, in input-row order.
 * out_coords must hold up to n rows.  parent[i] = output row of fine row i.  returns N_coarse. */
int64_t orc_stride2_coords(const int32_t* coords, int64_t n, int32_t stride_out, int32_t* out_coords,
                           int32_t* parent) {
    int32_t* q = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        q[4 * i] = coords[4 * i];
        for (int d = 1; d < 4; ++d) {
            int32_t c = coords[4 * i + d];
            int32_t f = (c >= 0) ? c / stride_out : -((-c + stride_out - 1) / stride_out);   /* floor division */
            q[4 * i + d] = f * stride_out;
        }
    }
    kv_t* kv = build_sorted(q, n);
    /* rank each unique key by the smallest row holding it, then number them in that order */
    int32_t* first_row = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* rank_of_first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) rank_of_first[i] = -1;
    for (int64_t i = 0; i < n;) {
        int64_t j = i; while (j < n && key_eq(kv[j].key, kv[i].key)) ++j;
        for (int64_t t = i; t < j; ++t) first_row[kv[t].row] = kv[i].row;     /* kv sorted by (key,row): kv[i].row is min */
        i = j;
    }
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i)
        if (first_row[i] == (int32_t)i) { rank_of_first[i] = (int32_t)cnt; memcpy(out_coords + 4 * cnt, q + 4 * i, 16); ++cnt; }
    for (int64_t i = 0; i < n; ++i) parent[i] = rank_of_first[first_row[i]];
    free(first_row); free(rank_of_first); free(kv); free(q);
    return cnt;
}

/* Kernel map of MinkowskiConvolution(kernel_size=3, stride=1) on a tensor of stride s
 * (autoencoder.py:13-27,35-41,71-77,...).  ‡ offset index k -> (k%3-1, (k/3)%3-1, k/9-1) * s, x fastest.
 * nbr is [27][n]: row index of coords[o] + offset_k, or -1. */
void orc_kmap_k3(const int32_t* coords, int64_t n, int32_t stride, int32_t* nbr) {
    kv_t* kv = build_sorted(coords, n);
    #pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < n; ++o) {
        const int32_t* c = coords + 4 * o;
        for (int k = 0; k < 27; ++k) {
            int32_t dx = (k % 3 - 1) * stride, dy = ((k / 3) % 3 - 1) * stride, dz = (k / 9 - 1) * stride;
            nbr[(int64_t)k * n + o] = find_row(kv, n, pack_key(c[0], c[1] + dx, c[2] + dy, c[3] + dz));
        }
    }
    free(kv);
}

/* Kernel map of the k=2,s=2 down convolution: for coarse row o (stride 2s) and k in [0,8):
 * fine row at coarse + (k&1, (k>>1)&1, k>>2) * s  (‡ even kernel: offsets {0,1}^3 from the output origin, x fastest).
 * nbr is [8][n_coarse]. */
void orc_kmap_down(const int32_t* fine, int64_t n_fine, const int32_t* coarse, int64_t n_coarse, int32_t stride_fine,
                   int32_t* nbr) {
    kv_t* kv = build_sorted(fine, n_fine);
    #pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < n_coarse; ++o) {
        const int32_t* c = coarse + 4 * o;
        for (int k = 0; k < 8; ++k)
            nbr[(int64_t)k * n_coarse + o] = find_row(kv, n_fine, pack_key(c[0], c[1] + (k & 1) * stride_fine,
                                                c[2] + ((k >> 1) & 1) * stride_fine, c[3] + (k >> 2) * stride_fine));
    }
    free(kv);
}

/* Output coordinates of MinkowskiGenerativeConvolutionTranspose(k=2,s=2) (autoencoder.py:155-161,182-188,209-215):
 * each input site c (stride s) spawns 8 children c + (k&1,(k>>1)&1,k>>2) * (s/2).  Canonical row order: 8*i + k. */
void orc_children_coords(const int32_t* coords, int64_t n, int32_t stride_in, int32_t* out) {
    int32_t h = stride_in / 2;
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) {
            int32_t* o = out + 4 * (8 * i + k);
            o[0] = coords[4 * i]; o[1] = coords[4 * i + 1] + (k & 1) * h;
            o[2] = coords[4 * i + 2] + ((k >> 1) & 1) * h; o[3] = coords[4 * i + 3] + (k >> 2) * h;
        }
}

/* Generic gather convolution: the arithmetic of ME's conv family under the canonical order.
 *   nbr [K][n_out] (int32, -1 = absent), in [n_in][in_ld] (first Cin columns used), W [K][Cin][Cout], bias [Cout] or NULL,
 *   out [n_out][out_ld] (columns out_coff .. out_coff+Cout).
 * k3: K=27 (orc_kmap_k3) · down: K=8 (orc_kmap_down) · k1: K=1, nbr[o]=o · transpose: see orc_conv_up2. */
/* Accumulation structure (CONVENTIONS['accumulate'], round 5).  0 = 'chain': ONE fmaf chain through every offset and channel (the
 * canonical form above, what the HIP kernels compute).  1 = 'per_offset_gemm': MinkowskiEngine's documented structure
 * (per kernel offset a GEMM of the gathered rows with W[k], whose result is ADDED into the output rows —
 * out[o] += in[i] @ W[k]): d_k = own fmaf chain over ci from +0, then acc = acc + d_k for k ascending.  The inner order of the
 * library GEMM is not observable here; this form sizes what the per-offset rounding of the partial sums alone does to the latents
 * and the top-k decisions (tools/order_sensitivity.py -> profiles/r05_order_sensitivity.md). */
static int g_accumulate = 0;
int orc_set_accumulate(int mode) { if (mode == 0 || mode == 1) g_accumulate = mode; return g_accumulate; }

void orc_conv_gather(const int32_t* nbr, int K, int64_t n_out, const float* in, int Cin, int in_ld, const float* W,
                     const float* bias, float* out, int Cout, int out_ld, int out_coff) {
    const int per_offset = g_accumulate == 1;
    #pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < n_out; ++o) {
        float acc[64], d[64];
        for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
        for (int k = 0; k < K; ++k) {
            int32_t r = nbr[(int64_t)k * n_out + o];
            if (r < 0) continue;
            const float* x = in + (int64_t)r * in_ld;
            const float* w = W + (int64_t)k * Cin * Cout;
            if (per_offset) {
                for (int co = 0; co < Cout; ++co) d[co] = 0.0f;
                for (int ci = 0; ci < Cin; ++ci) {
                    float a = x[ci];
                    const float* wr = w + (int64_t)ci * Cout;
                    for (int co = 0; co < Cout; ++co) d[co] = fmaf(a, wr[co], d[co]);
                }
                for (int co = 0; co < Cout; ++co) acc[co] = acc[co] + d[co];
                continue;
            }
            for (int ci = 0; ci < Cin; ++ci) {
                float a = x[ci];
                const float* wr = w + (int64_t)ci * Cout;
                for (int co = 0; co < Cout; ++co) acc[co] = fmaf(a, wr[co], acc[co]);
            }
        }
        float* y = out + (int64_t)o * out_ld + out_coff;
        for (int co = 0; co < Cout; ++co) y[co] = bias ? acc[co] + bias[co] : acc[co];
    }
}

/* MinkowskiGenerativeConvolutionTranspose(k=2,s=2): out[8i+k] = in[i] @ W[k] + bias (no overlap, no reduction). */
void orc_conv_up2(int64_t n_in, const float* in, int Cin, const float* W, const float* bias, float* out, int Cout) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_in; ++i)
        for (int k = 0; k < 8; ++k) {
            float acc[64];
            for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
            const float* w = W + (int64_t)k * Cin * Cout;
            for (int ci = 0; ci < Cin; ++ci) {
                float a = in[i * Cin + ci];
                for (int co = 0; co < Cout; ++co) acc[co] = fmaf(a, w[ci * Cout + co], acc[co]);
            }
            float* y = out + (8 * i + k) * (int64_t)Cout;
            for (int co = 0; co < Cout; ++co) y[co] = bias ? acc[co] + bias[co] : acc[co];
        }
}

/* ------------------------------------------------------------------------------------------------
 * Factorized entropy bottleneck tables: entropy_model.py:82-101 (_logits_cumulative), :112-130 (_likelihood),
 * :142-149 (_pmf_to_cdf), :151-176 (compress: clamp at 1e-9, symbols = arange(min_v, max_v+1)).
 * params: 352 floats packed as matrices0..3 | biases0..3 | factors0..3, each [C=8, f_out, f_in] row-major.
 * Evaluated in fp64 from the fp32 parameters and rounded to fp32 at the points where the reference holds fp32
 * tensors that feed a discontinuity (the final likelihood, the clamp and the running cumsum).
 * ---------------------------------------------------------------------------------------------- */
static const int EB_F[5] = {1, 3, 3, 3, 1};

static double eb_softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
static double eb_sigmoid(double x) { return x >= 0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x)); }

static double eb_logits(const float* params, int C, int c, double v) {
    const float* M = params; const float* B = params + 192 * C / 8; const float* F = B + 80 * C / 8;
    double h[3] = {v, 0, 0}, t[3];
    int moff = 0, boff = 0;
    for (int i = 0; i < 4; ++i) {
        int fi = EB_F[i], fo = EB_F[i + 1];
        const float* m = M + moff + c * fo * fi; const float* b = B + boff + c * fo; const float* f = F + boff + c * fo;
        for (int r = 0; r < fo; ++r) {
            double s = 0;
            for (int q = 0; q < fi; ++q) s += eb_softplus((double)m[r * fi + q]) * h[q];
            s += (double)b[r];
            s += tanh((double)f[r]) * tanh(s);
            t[r] = s;
        }
        for (int r = 0; r < fo; ++r) h[r] = t[r];
        moff += C * fo * fi; boff += C * fo;
    }
    return h[0];
}

/* likelihood [L][C] fp32 at integer symbols min_v .. max_v  (entropy_model.py:112-130) */
void orc_likelihood(const float* params, int C, float min_v, float max_v, float* lik) {
    int L = (int)(max_v - min_v) + 1;
    for (int s = 0; s < L; ++s)
        for (int c = 0; c < C; ++c) {
            double v = (double)min_v + s;
            double lo = eb_logits(params, C, c, v - 0.5), up = eb_logits(params, C, c, v + 0.5);
            double sum = lo + up, sign = sum > 0 ? -1.0 : (sum < 0 ? 1.0 : 0.0);
            lik[s * C + c] = (float)fabs(eb_sigmoid(sign * up) - eb_sigmoid(sign * lo));
        }
}

/* cdf [C][L+1] fp32: clamp(pmf, 1e-9) -> cumsum -> prepend 0 -> clamp(max=1)  (entropy_model.py:142-149,165-170) */
void orc_cdf_float(const float* params, int C, float min_v, float max_v, float* cdf) {
    int L = (int)(max_v - min_v) + 1;
    float* lik = (float*)malloc(sizeof(float) * (size_t)L * C);
    orc_likelihood(params, C, min_v, max_v, lik);
    for (int c = 0; c < C; ++c) {
        float run = 0.0f;
        cdf[c * (L + 1)] = 0.0f;
        for (int s = 0; s < L; ++s) {
            float p = lik[s * C + c]; if (p < 1e-9f) p = 1e-9f;
            run = run + p;                                   /* fp32 running sum, as torch.cumsum on fp32 */
            cdf[c * (L + 1) + s + 1] = run > 1.0f ? 1.0f : run;
        }
    }
    free(lik);
}

/* torchac 0.9.3 ‡ _convert_to_int_and_normalize (call site entropy_model.py:174,192):
 * u16[j] = (uint16)( int16( round_half_even( cdf[j] * (65536 - (Lp-1)) ) ) + j ),  Lp = L+1. */
void orc_cdf_u16(const float* cdf, int C, int Lp, uint16_t* out) {
    float scale = 65536.0f - (float)(Lp - 1);
    for (int c = 0; c < C; ++c)
        for (int j = 0; j < Lp; ++j) {
            float v = rintf(cdf[c * Lp + j] * scale);
            out[c * Lp + j] = (uint16_t)((int32_t)v + j);
        }
}

/* ------------------------------------------------------------------------------------------------
 * torchac 0.9.3 ‡ range coder (32-bit low/high, E3 pending bits, MSB-first bit packing), restated from the
 * published algorithm; call sites entropy_model.py:174 (encode_float_cdf) and :192 (decode_float_cdf).
 * Symbols are coded in row-major [point, channel] order; the CDF row of symbol i is channel i % C
 * (the reference repeats one [C, Lp] table N times, entropy_model.py:173).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t* buf; int64_t cap, len; uint8_t cache; int count; } bitw_t;

static void bw_put(bitw_t* w, int bit) {
    w->cache = (uint8_t)((w->cache << 1) | (bit & 1));
    if (++w->count == 8) { if (w->len < w->cap) w->buf[w->len] = w->cache; w->len++; w->count = 0; w->cache = 0; }
}
static void bw_put_pending(bitw_t* w, int bit, uint64_t* pending) {
    bw_put(w, bit);
    while (*pending > 0) { bw_put(w, !bit); --*pending; }
}

int64_t orc_rc_encode(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap) {
    bitw_t w = {out, cap, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu; uint64_t pending = 0;
    const int max_symbol = Lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t* row = cdf + (i % C) * Lp;
        int s = sym[i];
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint32_t c_low = row[s];
        uint32_t c_high = (s == max_symbol) ? 0x10000u : row[s + 1];
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) { bw_put_pending(&w, 0, &pending); low <<= 1; high = (high << 1) | 1; }
            else if (low >= 0x80000000u) { bw_put_pending(&w, 1, &pending); low <<= 1; high = (high << 1) | 1; }
            else if (low >= 0x40000000u && high < 0xC0000000u) {
                ++pending; low = (low << 1) & 0x7FFFFFFFu; high = (high << 1) | 0x80000001u;
            } else break;
        }
    }
    pending += 1;
    bw_put_pending(&w, low < 0x40000000u ? 0 : 1, &pending);
    if (w.count > 0) { int pad = 8 - w.count; for (int i = 0; i < pad; ++i) bw_put(&w, 0); }
    return w.len;       /* bytes produced (may exceed cap: caller re-calls with a larger buffer) */
}

typedef struct { const uint8_t* buf; int64_t len, pos; uint8_t cache; int bits; } bitr_t;
static void br_get(bitr_t* r, uint32_t* value) {
    if (r->bits == 0) {
        if (r->pos == r->len) { *value <<= 1; return; }
        r->cache = r->buf[r->pos++]; r->bits = 8;
    }
    *value = (*value << 1) | ((r->cache >> (r->bits - 1)) & 1);
    r->bits--;
}

void orc_rc_decode(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n) {
    bitr_t r = {in, nbytes, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    const int max_symbol = Lp - 2;
    for (int i = 0; i < 32; ++i) br_get(&r, &value);
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t* row = cdf + (i % C) * Lp;
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint16_t count = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000u - 1) / span);
        uint16_t left = 0, right = (uint16_t)(max_symbol + 1);
        while (left + 1 < right) {
            uint16_t m = (uint16_t)((left + right) / 2); uint16_t v = row[m];
            if (v < count) left = m; else if (v > count) right = m; else { left = m; break; }
        }
        int s = left; sym[i] = (int16_t)s;
        if (i == n - 1) break;
        uint32_t c_low = row[s];
        uint32_t c_high = (s == max_symbol) ? 0x10000u : row[s + 1];
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (low >= 0x80000000u || high < 0x80000000u) { low <<= 1; high = (high << 1) | 1; br_get(&r, &value); }
            else if (low >= 0x40000000u && high < 0xC0000000u) {
                low = (low << 1) & 0x7FFFFFFFu; high = (high << 1) | 0x80000001u; value -= 0x40000000u; br_get(&r, &value);
            } else break;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * D1 (point-to-point) squared nearest-neighbour distance sums, both directions, brute force O(Na*Nb).
 * mpeg-pcc-dmetric 0.13.4 ‡ (pc_error.py:44-49): mse = mean NN squared distance; checked against golden G4.
 * Used only on small clouds; the python side has a KD-tree variant for large ones.
 * ---------------------------------------------------------------------------------------------- */
double orc_nn_sqdist_sum(const int32_t* a, int64_t na, const int32_t* b, int64_t nb, double* max_out) {
    double total = 0, mx = 0;
    #pragma omp parallel for reduction(+:total) reduction(max:mx) schedule(static)
    for (int64_t i = 0; i < na; ++i) {
        int64_t best = INT64_MAX;
        for (int64_t j = 0; j < nb; ++j) {
            int64_t dx = a[3 * i] - b[3 * j], dy = a[3 * i + 1] - b[3 * j + 1], dz = a[3 * i + 2] - b[3 * j + 2];
            int64_t d = dx * dx + dy * dy + dz * dz; if (d < best) best = d;
        }
        total += (double)best; if ((double)best > mx) mx = (double)best;
    }
    if (max_out) *max_out = mx;
    return total;
}
