// k3 conv 64 -> 64 with PRESENT-ROW PACKING (round 4): the two layers whose 442 KB of weights fit no LDS-resident table —
// the encoder's conv2 (autoencoder.py:109-115, 71 k rows) and the decoder's conv0 (:162-168, 150 k rows).
//
// Every other MFMA kernel of this library is output-stationary per 16-row tile: a tile multiplies all 27 offsets, and a row whose
// neighbour at an offset is absent contributes a zero A row — on a surface 27 / 15 = 1.8 x the MFMA work that carries information, and an
// M tile can skip an offset only when none of its 16 rows has that neighbour (almost never).  Here a workgroup owns R = 128 output rows;
// per offset the rows that HAVE the neighbour are compacted (ballot + popcount) into packed M tiles of 16 — 70 of 128 on average: 5 tiles
// instead of 8 — and the accumulators, which no longer sit at fixed (lane, register) positions, live in LDS: a packed tile reads its
// 16 x 16 accumulator block (C), runs the 16 K-steps of the offset, and writes it back (D).  At Cout = 64 that is 8 LDS operations per 16
// MFMAs (affordable; not at Cout = 16).  The four waves split the OUTPUT COLUMNS (wave w = columns 16 w .. 16 w + 15 of every row), so
//   * a wave's B fragments of an offset are 16 registers, loaded straight from the L2-resident fragment table (ops.child_conv_table:
//     4 x global_load_dwordx4 per offset, one offset ahead) — no weight staging through LDS;
//   * accumulator columns are wave-private: no synchronisation for C / D at all;
//   * only the gathered rows (A) are shared: packed tile i is fetched by wave i & 3 (LDS-DMA, 4 KB per tile), two barriers per offset.
// Per output element the products arrive in the canonical order — offsets ascending (absent ones skipped: fma(0, w, acc) = acc), then
// input channels ascending (block, K-step, K index) — and an fp32 accumulator survives its LDS round trip bit for bit: identical results.
#include "mfma_util.h"

namespace {

constexpr int PK_RMAX = 128;                    // output rows per workgroup: R <= 128 (two ballots), chosen per launch (pk_rows)
constexpr int PK_ACC_LD = 68;                   // floats per accumulator row in LDS (64 + 4: rows start 4 banks apart)
// LDS of a workgroup of R rows: accumulators (+ one dummy row for the padding slots of the last packed tile) | gathered rows of one offset
// (up to ceil(R / 16) packed tiles x 4 KB) | per wave: gather row + output row of every packed slot
__host__ __device__ constexpr int pk_tiles(int R) { return (R + 15) / 16; }
__host__ __device__ constexpr int pk_acc_bytes(int R) { return (R + 1) * PK_ACC_LD * 4; }
__host__ __device__ constexpr int pk_a_bytes(int R) { return pk_tiles(R) * 4096; }
__host__ __device__ constexpr int pk_slots(int R) { return pk_tiles(R) * 16; }
__host__ __device__ constexpr int pk_lds(int R, int NW) { return pk_acc_bytes(R) + pk_a_bytes(R) + NW * 2 * pk_slots(R) * 4; }

// NW = 4 or 8 waves: wave w multiplies against column tile w & 3; with eight waves the packed tiles of an offset alternate between the two
// waves of a column tile (group = w >> 2) — different rows inside an offset, and the barriers between offsets order the rest
// Operand roles (round 5): A = the weight fragment, B = the gathered rows, so that D = [column][row]: lane (row slot mi, quarter mq) holds
// the FOUR CONSECUTIVE columns 4 mq .. 4 mq + 3 of its row, and a packed tile's C / D traffic is one ds_read_b128 + one ds_write_b128 per
// lane (2 LDS instructions per 16 MFMAs, one accumulator row per lane; rows start 4 banks apart: contiguous row slots are conflict-free).
// Round 4 had the roles the other way round — four b32 reads and writes to four rows per lane: 289 -> 280 us on the 149 856-row level,
// 126 -> 117 on the 71 216-row one, same bits (fma(a, b, c) = fma(b, a, c); profiles/r05_conv_packed_roles.txt).
template <int NW>
__global__ void __launch_bounds__(NW * 64)
k_conv_packed64(const int32_t* __restrict__ nbr, int64_t n, const float* __restrict__ in, int in_ld, const float* __restrict__ table,
                const float* __restrict__ bias, int relu, float* __restrict__ out, int out_ld, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* accf = (float*)lds_raw;                                              // [R + 1][68]
    float* abuf = (float*)(lds_raw + pk_acc_bytes(R));                          // [tile][cb][16 rows][4 chunks][4] (swizzled, see below)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int32_t* slot_gid = (int32_t*)(lds_raw + pk_acc_bytes(R) + pk_a_bytes(R)) + wave * 2 * pk_slots(R);        // [gather row | output row][slots]
    const int mi = lane & 15, mq = lane >> 4;
    constexpr int NG = NW / 4;
    const int nt = wave & 3, group = wave >> 2;                                  // column tile, tile group
    const int64_t row0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * R;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(n * in_ld * 4), 0x00020000);
    // the A image of a packed tile: per 16-channel block a 1 KB image written lane-linearly by one LDS-DMA instruction — lane (r = lane >> 2,
    // p = lane & 3) fetches source chunk p ^ f(r >> 2) of row r, so chunk jj of row mi sits at position jj ^ f(mi >> 2): the four
    // ds_read_b32 of an MFMA A operand (channel 4 jj + mq of row mi) are bank-conflict-free (the layout of the children-level kernels)
    const int f_a = (0x78 >> (2 * (mi >> 2))) & 3;
    const int dma_r = lane >> 2;
    const int dma_chunk = (lane & 3) ^ ((0x78 >> (2 * ((dma_r >> 2) & 3))) & 3);

    // accumulators: this wave's 16 columns of all rows (and of the dummy row) start at +0
    for (int r = mq + 4 * group; r <= R; r += 4 * NG) accf[r * PK_ACC_LD + 16 * nt + mi] = 0.0f;

    const bool ok0 = lane < R && row0 + lane < n, ok1 = 64 + lane < R && row0 + 64 + lane < n;
    const float4* frag = (const float4*)table + ((size_t)nt * 4) * 64 + lane;              // fragment (k, n = nt, cb): + (k * 16 + cb) * 64
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));                           // lanes below this one
    const int a_floats = pk_a_bytes(R) / 4, nslots = pk_slots(R);

    // pack: the rows of the tile that have a neighbour at an offset (map entries e0 / e1 of rows lane / 64 + lane), in row order, into slot
    // list `sb` (every wave builds the same list for itself) -> packed tiles of the offset
    auto pack = [&](int e0, int e1, int sb) -> int {
        int32_t* gid = slot_gid + sb * 2 * nslots;
        int32_t* row = gid + nslots;
        const bool p0 = e0 >= 0, p1 = e1 >= 0;
        const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
        const int c0 = __popcll(m0), m = c0 + __popcll(m1);
        const int T = (m + 15) >> 4;
        if (p0) { const int pos = __popcll(m0 & lt); gid[pos] = e0; row[pos] = lane; }
        if (p1) { const int pos = c0 + __popcll(m1 & lt); gid[pos] = e1; row[pos] = 64 + lane; }
        if (lane < 16 && m + lane < 16 * T) { gid[m + lane] = -1; row[m + lane] = R; }     // padding slots: no row fetched, dummy accumulator row
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return T;
    };
    // gather: packed tile i by wave i mod NW, four 1 KB images (16-channel blocks) per tile, into A buffer `ab`
    auto gather = [&](int T, int sb, int ab) {
        const int32_t* gidl = slot_gid + sb * 2 * nslots;
        for (int i = wave; i < T; i += NW) {
            const int gid = gidl[16 * i + dma_r];
            const unsigned voff = gid >= 0 ? (unsigned)(((int64_t)gid * in_ld + dma_chunk * 4) * 4) : 0xFFFFFFF0u;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(abuf + ab * a_floats + (i * 4 + cb) * 256), 16, (int)(gid >= 0 ? voff + cb * 64 : voff), 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    };

    // Per offset: pack -> barrier (everyone is done reading the previous offset's rows) -> gather, with the next offset's map entries and B
    // fragments travelling behind it -> wait -> barrier (all packed rows have landed) -> multiply.  The gather latency is covered by the
    // OTHER workgroups of the CU (three to four are resident).  Measured and dropped: a second A buffer and slot list so that offset k + 1
    // is gathered while offset k is multiplied, one barrier per offset — the LDS it takes costs a resident workgroup and every size tried
    // got slower (profiles/r04_conv_packed.md): as with the rows kernels, waves in flight beat deeper buffering.
    float4 b_cur[4], b_nxt[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) b_cur[cb] = frag[cb * 64];
    int e0 = ok0 ? nbr[row0 + lane] : -1, e1 = ok1 ? nbr[row0 + 64 + lane] : -1;          // map entries of offset 0
    for (int k = 0; k < 27; ++k) {
        const int T_cur = pack(e0, e1, 0);
        __syncthreads();
        gather(T_cur, 0, 0);
        int e0n = -1, e1n = -1;
        if (k + 1 < 27) {
            if (ok0) e0n = nbr[(int64_t)(k + 1) * n + row0 + lane];
            if (ok1) e1n = nbr[(int64_t)(k + 1) * n + row0 + 64 + lane];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) b_nxt[cb] = frag[((k + 1) * 16 + cb) * 64];
        }
        wait_vmcnt<0>();
        __syncthreads();
        // ---- multiply: the packed tiles of offset k against this wave's 16 columns; C from / D to the LDS accumulators
        const int32_t* rowl = slot_gid + nslots;
        const float* abase = abuf + (mi * 4) * 4 + mq;
        // two packed tiles at a time: two independent accumulator chains (a dependent fp32 MFMA issues every ~40 cycles, an independent
        // one every 32) and both tiles' LDS reads in flight before the first MFMA
        // C / D of a packed tile: lane (row slot mi, quarter mq) owns columns 4 mq .. 4 mq + 3 of its row — one 16-byte LDS read and write
        auto cd = [&](int t) { return (f32x4*)(accf + rowl[16 * t + mi] * PK_ACC_LD + 16 * nt + 4 * mq); };
        int i = group;
        for (; i + NG < T_cur; i += 2 * NG) {
            f32x4* const c0 = cd(i);
            f32x4* const c1 = cd(i + NG);
            f32x4 acc0 = *c0, acc1 = *c1;
            const float* img0 = abase + i * 1024;
            const float* img1 = abase + (i + NG) * 1024;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                float a0[4], a1[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { a0[jj] = img0[cb * 256 + (jj ^ f_a) * 4]; a1[jj] = img1[cb * 256 + (jj ^ f_a) * 4]; }
                const float bw[4] = {b_cur[cb].x, b_cur[cb].y, b_cur[cb].z, b_cur[cb].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[jj], a0[jj], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[jj], a1[jj], acc1, 0, 0, 0);
                }
            }
            *c0 = acc0; *c1 = acc1;
        }
        if (i < T_cur) {
            f32x4* const c = cd(i);
            f32x4 acc = *c;
            const float* img = abase + i * 1024;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                float a[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a[jj] = img[cb * 256 + (jj ^ f_a) * 4];
                const float bw[4] = {b_cur[cb].x, b_cur[cb].y, b_cur[cb].z, b_cur[cb].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[jj], a[jj], acc, 0, 0, 0);
            }
            *c = acc;
        }
        e0 = e0n; e1 = e1n;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b_cur[cb] = b_nxt[cb];
    }
    // ---- epilogue: this wave's 16 columns (four waves: of every row, wave-private; eight: of every second 16-row group), 16 bytes per lane
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (NG > 1) __syncthreads();
    const int c4 = lane & 3;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *(const float4*)(bias + 16 * nt + 4 * c4);
    for (int it = group; it < pk_tiles(R); it += NG) {
        const int r = it * 16 + (lane >> 2);
        if (r < R && row0 + r < n) {
            float4 v = *(const float4*)(accf + r * PK_ACC_LD + 16 * nt + 4 * c4);
            v.x = v.x + bv.x; v.y = v.y + bv.y; v.z = v.z + bv.z; v.w = v.w + bv.w;
            if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            *(float4*)(out + (row0 + r) * out_ld + 16 * nt + 4 * c4) = v;
        }
    }
}

// Rows per workgroup, from a cost model fitted to sweeps of R on both layers of a vox10 frame (profiles/r04_conv_packed.md).  A launch is
// `tiles` workgroups on cus x occ(R) slots (occ: LDS-limited workgroups per CU; LDS is granted in 1280-byte granules); a CU runs c workgroups
// side by side, each offset costing about L + c W tau(c) us: L = what nobody covers (barriers, gather latency), W = R p / 16 + 1/2 packed tiles
// per offset (the + 1/2 is the half-empty last tile), tau = time per packed tile, larger when few waves share a SIMD.  Large R packs better
// but leaves two workgroups per CU; and e.g. 71 216 rows in tiles of 96 are 742 workgroups on 512 slots — two rounds — where tiles of 94
// are 758 on 768: one.  p = 0.6 (a surface level has 14-19 of 27 neighbours; the choice is flat in p).
int pk_occ(int R, int nw) {
    const int granules = (pk_lds(R, nw) + 1279) / 1280;
    int occ = (160 * 1024) / (granules * 1280);
    const int by_waves = 24 / nw;                               // 84 VGPRs: six waves per SIMD
    if (occ > by_waves) occ = by_waves;
    return occ < 1 ? 1 : occ;
}
int pk_rows(int64_t n, int cus) {
    auto tau = [](int c) { return c >= 4 ? 0.219 : (c == 3 ? 0.240 : (c == 2 ? 0.276 : 0.370)); };      // (least-squares fit to the two sweeps: 4 % rms)
    auto cost_of = [&](int R) {
        const int occ = pk_occ(R, 4);
        const int64_t tiles = (n + R - 1) / R, slots = (int64_t)cus * occ;
        const double W = R * 0.6 / 16.0 + 0.346, L = 1.27;
        const int64_t full = tiles / slots, rem = tiles - full * slots;
        double cost = (double)full * (L + occ * W * tau(occ));
        if (rem) { const int c = (int)((rem + cus - 1) / cus); cost += L + c * W * tau(c); }
        return cost;
    };
    double best_cost = 1e300;
    int best = PK_RMAX;
    for (int R = PK_RMAX; R >= 40; R -= 2)
        if (cost_of(R) < best_cost) { best_cost = cost_of(R); best = R; }
    // The model is flat to within its own error (4 % rms) over wide ranges of R — 149 856 rows: 260.5 at R = 50, 264.4 at 60, where the
    // measured sweep has 298 and 288 us (profiles/r04_conv_packed.md) — so among the sizes within 2 % of its minimum the LARGEST is taken
    // larger tiles pack better than the model's fixed + 1/2 tile per offset credits them with (round 5: conv0 picks 60 instead of 50).
    // (only among tiles of the minimum's own occupancy: across an occupancy step the model's error is not a few per cent)
    for (int R = PK_RMAX; R >= 40; R -= 2)
        if (cost_of(R) <= 1.02 * best_cost && pk_occ(R, 4) == pk_occ(best, 4)) return R;
    return best;
}
int g_pk_rows = 0;                                              // A/B: rows per workgroup forced (0 = default)

}  // namespace
extern "C" int pcgc_set_packed_tuning(int rows, int waves) {
    if (rows < 0 || rows > PK_RMAX || (waves != 0 && waves != 4)) return -1;      // (four waves per workgroup: the eight-wave form lost everywhere it was measured)
    g_pk_rows = rows; return 0;
}

// MinkowskiConvolution k3 64 -> 64 on a level with its own kernel map nbr [27][n] (-1 = absent); table = ops.child_conv_table(kernel)
// ([k][n][cb] lane-linear fragments, 442 368 bytes); out = (acc + bias) (relu).  No residual form (neither layer has one).
extern "C" int pcgc_conv_packed64(const int32_t* nbr, int64_t n, const float* in, int in_ld, const float* table, int64_t table_bytes,
                                  const float* bias, int relu, float* out, int out_ld, void* stream) {
    PCGC_REQUIRE(nbr && in && table && out, "null argument");
    PCGC_REQUIRE(table_bytes == (int64_t)27 * 64 * 64 * 4, "table size");
    PCGC_REQUIRE(in_ld >= 64 && out_ld >= 64 && (in_ld & 3) == 0 && (out_ld & 3) == 0, "rows of at least 64 floats, 16-byte multiples");
    PCGC_REQUIRE((((uintptr_t)in | (uintptr_t)table | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "unaligned tensor");
    PCGC_REQUIRE(n * (int64_t)in_ld * 4 < (int64_t)0xF0000000, "tensor too large for 32-bit buffer offsets");
    if (n == 0) return 0;
    static int granted[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int cus = 0;
    if (!cus) { hipDeviceProp_t p; cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    const int R = g_pk_rows > 0 ? g_pk_rows : pk_rows(n, cus);
    constexpr int nw = 4;
    const int lds = pk_lds(R, nw);
    if (!granted[dev & 15]) {
        for (const void* f : {(const void*)k_conv_packed64<4>}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, pk_lds(PK_RMAX, 8));
            if (e != hipSuccess) { pcgc_set_error("conv_packed64: cannot raise the LDS limit to %d: %s", pk_lds(PK_RMAX, 8), hipGetErrorString(e)); return -1; }
        }
        granted[dev & 15] = 1;
    }
    hipLaunchKernelGGL(k_conv_packed64<4>, dim3(grid_for(n, R)), dim3(256), lds, S(stream), nbr, n, in, in_ld, table, bias, relu, out, out_ld, R);
    PCGC_CHECK_LAUNCH("conv_packed64");
    return 0;
}
