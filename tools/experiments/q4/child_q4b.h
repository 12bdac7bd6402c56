// Quad-block pass B of the C = 16 InceptionResNet on a children level (round 5; pass A and the geometry: child_q4.h):
//   out[row][0:8]  = (conv0_1(t0) + b01) + x[row][0:8]                          conv0_1: k3 4 -> 8 on t0 = relu(conv0_0 x)
//   out[row][8:16] = (conv1_2(relu(conv1_1(t1) + b11)) + b12) + x[row][8:16]    conv1_1: k3 4 -> 4 on t1 = relu(conv1_0 x); conv1_2: k1 4 -> 8
// (autoencoder.py:52-57, second half).  t is read in the T2 layout pass A writes (per parent 256 bytes = [z half][conv][child & 3][4 ch]).
//
// One wave = 64 parents (lane = parent).  Per (cell, child) pair twelve `v_mfma_f32_4x4x1_16b_f32`: four input channels x {conv0_1 columns
// 0-3, conv0_1 columns 4-7, conv1_1 columns 0-3} into three accumulators per child (96 registers for the eight children — which is why a
// wave holds ONE M tile here; pass A holds two).  Operand A = the weights of the lane's column (lane & 3): 12 floats per pair = three
// broadcast ds_read_b128 from a [k][s][12] table; operand B = the lane's gathered row: t0 and t1, one ds_read_b128 each per cell.  The
// packed-N pass B issues 248 sixteen-column MFMAs per 16 parents (992 per 64: 31.7 k pipe cycles); this form 216 x 12 + 8 x 24 = 2 784
// four-column ones (23.4 k) — conv0_1's tiles were three-quarters full already, so the saving is 0.74x, not pass A's 0.54x.
//
// Gather: a row's two pieces sit 64 bytes apart in one 128-byte line of the T2 layout; gather instruction d fetches rows 32 d .. 32 d + 31,
// lane L piece (L & 1) ^ f(row) of row 32 d + (L >> 1) (f = (row >> 3) & 1: the image [row][2 pieces] is then read back conflict-free by
// lane = row).  The map entries of row (L >> 1) + 32 (L & 1) live in lane L; instruction d takes them from lane (L & ~1) | d of the pair
// (quad_perm {d, d, 2 + d, 2 + d}).  Eight cells of rows are in flight (2 KB each): a cell is only 1-8 pairs = 100-800 pipe cycles long.
//
// Epilogue per child, two groups after its last product (k = 26): u = relu(conv1_1 + b11) stays in the lane (lane = row IS the B-operand
// layout of the k1 conv: 8 more MFMAs); the four 16-byte pieces of the child's output row [conv0_1 0:4 | 4:8 | conv1_2 0:4 | 4:8] + biases
// are transposed across each quad by the matrix pipe (child_q4.h: exact but for the sign of a zero: (acc + b) = -0.0 becomes +0.0, which
// changes a result only if the residual is -0.0 as well) so that in store step t the quad writes the 64 contiguous bytes of parent
// 4 b + t's row; the residual x is loaded in that same transposed pattern a group earlier and added after the transposition.
#pragma once
#include "child_q4.h"

namespace {

struct Q4bGroup {
    int cell;
    int npairs, child[2], k[2];   // one or two (cell, child) pairs: children and their kernel offsets
    bool first, last;             // first / last group of its cell
    int fin[2];                   // pair p completes child `child[p]` (k = 26)
};
struct Q4bSched {
    int n;
    Q4bGroup g[128];
    int vm_wait[128];             // last group of cell c (c < 63): vmcnt that guarantees cell c + 1's rows have landed
    int ev_load[128], ev_store[128];   // child whose residual loads / whose transposes + stores are issued at the end of group n (-1: none)
    int x_wait[128];              // vmcnt that guarantees the residual pieces loaded for ev_store[n] have arrived
};
constexpr int Q4B_D = 8;          // ring slots (cells of rows in flight)
constexpr Q4bSched q4b_sched() {
    Q4bSched S{};
    for (int c = 0; c < 64; ++c) {
        const int n0 = S.n;
        int pend = -1;
        for (int j = 0; j < 8; ++j) {
            if (!((cell_reach(c) >> j) & 1)) continue;
            if (pend < 0) { pend = j; continue; }
            Q4bGroup G{}; G.cell = c; G.npairs = 2; G.child[0] = pend; G.child[1] = j; G.k[0] = cell_k(c, pend); G.k[1] = cell_k(c, j);
            G.fin[0] = G.k[0] == 26; G.fin[1] = G.k[1] == 26;
            S.g[S.n++] = G; pend = -1;
        }
        if (pend >= 0) {
            Q4bGroup G{}; G.cell = c; G.npairs = 1; G.child[0] = pend; G.child[1] = pend; G.k[0] = cell_k(c, pend); G.k[1] = 0; G.fin[0] = G.k[0] == 26;
            S.g[S.n++] = G;
        }
        S.g[n0].first = true;
        S.g[S.n - 1].last = true;
    }
    // epilogue events: a child finished by group F gets its transposes + stores at the end of group F + 2 (one slot each) and its residual
    // loads LEAD groups before that — vmcnt retires in order, so a wait for a residual piece also waits for every gather issued before it:
    // requested a group ahead it drained the whole gather ring once per child (115 us; 68 without residual / output traffic).  Two register
    // sets of residual pieces (child & 1): the load of child j waits for the store of child j - 2.
    for (int n = 0; n < 128; ++n) { S.ev_load[n] = -1; S.ev_store[n] = -1; }
    {
        constexpr int LEAD = 8;
        int store_slot[8] = {}, at = 0;
        for (int n = 0; n < S.n; ++n)
            for (int p = 0; p < S.g[n].npairs; ++p)
                if (S.g[n].fin[p]) {
                    const int j = S.g[n].child[p];
                    int slot = n + 2 > at ? n + 2 : at;
                    S.ev_store[slot] = j; store_slot[j] = slot; at = slot + 1;
                }
        int lat = 0;
        for (int j = 0; j < 8; ++j) {
            int slot = store_slot[j] - LEAD;
            if (j >= 2 && slot <= store_slot[j - 2]) slot = store_slot[j - 2] + 1;      // the register set is free after that store
            if (slot < lat) slot = lat;
            if (slot < 0) slot = 0;
            S.ev_load[slot] = j; lat = slot + 1;                                         // (one load event per slot; slot < store_slot[j] by construction)
        }
    }
    // VMEM instruction order: [gathers 0 .. D-1], then per group: (first: gather cell + D) (last: wait for cell + 1) ... MFMAs ... (wait x, stores of
    // ev_store) (loads of ev_load) — stores first: the one register set of residual pieces is consumed before it is refilled.  Slots past the
    // last group (children 6 / 7 finish at the very end) are emitted after the loop in the same order.
    int ops = 0;
    int gather_end[64 + Q4B_D + 2] = {};
    for (int c = 0; c < Q4B_D; ++c) { ops += 2; gather_end[c] = ops; }
    int load_end[8] = {};
    for (int n = 0; n < S.n + 3; ++n) {
        if (n < S.n) {
            const Q4bGroup& G = S.g[n];
            if (G.first && G.cell + Q4B_D < 64) { ops += 2; gather_end[G.cell + Q4B_D] = ops; }
            if (G.last && G.cell + 1 < 64) S.vm_wait[n] = ops - gather_end[G.cell + 1];
        }
        if (S.ev_store[n] >= 0) { S.x_wait[n] = ops - load_end[S.ev_store[n]]; ops += 4; }
        if (S.ev_load[n] >= 0) { ops += 4; load_end[S.ev_load[n]] = ops; }
    }
    return S;
}

// table (ops.child_q4b_tables): [k = 0..26][s = 0..3][12] = { W01[k][ci 0..3][s], W01[k][ci][4 + s], W11[k][ci][s] }, then
// [s][8] = { W12[ci 0..3][s], W12[ci][4 + s] }  (27 * 192 + 128 = 5312 bytes, padded to 5376)
template <int NW>
__global__ void __launch_bounds__(NW * 64)
k_child_q4b16(const int32_t* __restrict__ pnbr, int64_t n_p, const float* __restrict__ in /* t, T2 layout */, int in_ld,
              const float* __restrict__ table, int table_bytes, IrnEpi ep) {
    constexpr int D = Q4B_D, SLOT_F4 = 128;                     // float4 per ring slot: 64 rows x 2 pieces
    constexpr Q4bSched S = q4b_sched();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4* ring = (float4*)(lds_raw + table_bytes) + wave * (D * SLOT_F4);
    child_stage_table<NW>(table, table_bytes, lds_raw);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(8 * n_p * 8 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)ep.x, 0, (int)(8 * n_p * ep.x_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)ep.out, 0, (int)(8 * n_p * ep.out_ld * 4), 0x00020000);
    const int sidx = lane & 3;
    float b01[8], b11[4], b12[8];                              // (wave-uniform: scalar registers)
#pragma unroll
    for (int r = 0; r < 8; ++r) { b01[r] = ep.b0[r]; b12[r] = ep.b2[r]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) b11[r] = ep.b1[r];
    const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + sidx * 12);
    const unsigned w12_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_raw + 27 * 48 + sidx * 8);
    // the lane's row in a ring slot: piece c at (c ^ f) * 16, f = (row >> 3) & 1
    const unsigned a_row = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + lane * 8 + (((lane >> 3) & 1) << 2));
    // gather instruction d: lane L fetches for row 32 d + (L >> 1) the piece that belongs at slot position L & 1
    const unsigned piece_off = (unsigned)((((lane & 1) ^ ((lane >> 4) & 1))) * 64);       // f(32 d + (L >> 1)) = (L >> 4) & 1
    constexpr unsigned ABSENT = 0xF0000000u;
    const int64_t ntiles = (n_p + 63) / 64;
    const unsigned x_row_bytes = (unsigned)ep.x_ld * 4u, out_row_bytes = (unsigned)ep.out_ld * 4u;

    for (int it = 0;; ++it) {
        const int64_t tile = child_tile<NW>(it, wave, ntiles);
        if (tile < 0) break;
        const int64_t p_base = tile * 64;
        unsigned rowb[27];                                     // map entries of row (L >> 1) + 32 (L & 1)
        {
            const int64_t rho = p_base + (lane >> 1) + 32 * (lane & 1);
            const bool ok = rho < n_p;
#pragma unroll
            for (int kp = 0; kp < 27; ++kp) rowb[kp] = (unsigned)pnbr[(int64_t)kp * n_p + (ok ? rho : 0)];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // map entries here, the previous tile's stores retired: vmcnt counts from zero
#pragma unroll
#ifdef Q4_KO_GATHER
            for (int kp = 0; kp < 27; ++kp) rowb[kp] = ((int)rowb[kp] == 0x7FFFFFF1) ? rowb[kp] * 256u : ABSENT;     // (timing experiment: no row is fetched)
#else
            for (int kp = 0; kp < 27; ++kp) rowb[kp] = (ok && (int)rowb[kp] >= 0) ? rowb[kp] * 256u : ABSENT;
#endif
        }
        f32x4 acc0[8][2], acc1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc0[j][0] = acc0[j][1] = acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        f32x4 rows[2][2], w[2][2][3], xres[2][4];
#ifdef Q4_KO_STORE
        f32x4 sink = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif

        auto gather = [&](auto ic) {
            constexpr int c = decltype(ic)::value, kp = cell_kp(c), ch = cell_child(c), slot = c % D;
            const unsigned cell_off = (unsigned)((ch >> 2) * 128 + (ch & 3) * 16) + piece_off;
            const int rb = (int)rowb[kp];
            const unsigned v0 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xA0, 0xF, 0xF, true) + cell_off;      // quad_perm {0, 0, 2, 2}
            const unsigned v1 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xF5, 0xF, 0xF, true) + cell_off;      // quad_perm {1, 1, 3, 3}
            float4* dst = ring + slot * SLOT_F4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst), 16, (int)v0, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 64), 16, (int)v1, 0, 0, 0);
            asm volatile("" ::: "memory");
        };
        auto load_rows = [&](auto ic, auto ibuf) {              // t0 / t1 of the lane's row of cell c
            constexpr int c = decltype(ic)::value, buf = decltype(ibuf)::value, slot = c % D;
            rows[buf][0] = lds_ld128_off<slot * 2048>(a_row);
            rows[buf][1] = lds_ld128_off<slot * 2048>(a_row ^ 16u);
        };
        auto load_w = [&](auto in_, auto ibuf) {
            constexpr int n = decltype(in_)::value, buf = decltype(ibuf)::value;
            static_for<0, S.g[n].npairs>([&](auto ip) {
                constexpr int p = decltype(ip)::value;
                static_for<0, 3>([&](auto ie) {
                    constexpr int e = decltype(ie)::value;
                    w[buf][p][e] = lds_ld128_off<S.g[n].k[p] * 192 + e * 16>(tab_lane);
                });
            });
        };
        // residual pieces of child j in the transposed pattern: step t, lane i of a quad = piece i (16 bytes) of parent 4 b + t's row
        auto load_x = [&](auto ij) {
            constexpr int j = decltype(ij)::value;
            const unsigned v = (unsigned)(p_base + (lane & ~3)) * (8u * x_row_bytes) + (unsigned)j * x_row_bytes + (unsigned)sidx * 16u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#ifdef Q4_KO_STORE
                xres[j & 1][t] = (f32x4){(float)v, 1.f, 2.f, 3.f};     // (timing experiment: no residual / output traffic)
#else
                const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(v + (unsigned)t * 8u * x_row_bytes), 0, 0);
                xres[j & 1][t] = (f32x4){__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
#endif
            }
            asm volatile("" ::: "memory");
        };
        auto finish = [&](auto ij) {                            // conv1_2 on u, biases, quad transposes, residual, stores
            constexpr int j = decltype(ij)::value;
            f32x4 wk[2];
            wk[0] = lds_ld128_off<0>(w12_lane);
            wk[1] = lds_ld128_off<16>(w12_lane);
            f32x4 u;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = fmaxf(acc1[j][r] + b11[r], 0.0f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_tie(wk[0]); lds_tie(wk[1]);
            f32x4 P[4];
            P[2] = P[3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                P[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[0][ci], u[ci], P[2], 0, 0, 0);
                P[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[1][ci], u[ci], P[3], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                P[0][r] = acc0[j][0][r] + b01[r]; P[1][r] = acc0[j][1][r] + b01[4 + r];
                P[2][r] = P[2][r] + b12[r]; P[3][r] = P[3][r] + b12[4 + r];
            }
            float onehot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) onehot[k] = (sidx == k) ? 1.0f : 0.0f;
            f32x4 W[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) W[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) W[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[k][r], onehot[k], W[r], 0, 0, 0);
            const unsigned v = (unsigned)(p_base + (lane & ~3)) * (8u * out_row_bytes) + (unsigned)j * out_row_bytes + (unsigned)sidx * 16u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 o = {W[0][t], W[1][t], W[2][t], W[3][t]};
                o = o + xres[j & 1][t];
#ifdef Q4_KO_STORE
                sink = sink + o;
#else
                q4_store(o, rs_out, v + (unsigned)t * 8u * out_row_bytes);
#endif
            }
            asm volatile("" ::: "memory");
        };
        auto events = [&](auto in_) {                           // what the schedule puts behind group n
            constexpr int n = decltype(in_)::value;
            if constexpr (S.ev_store[n] >= 0) {
                wait_vmcnt<S.x_wait[n]>();
                static_for<0, 4>([&](auto it_) { lds_tie(xres[(S.ev_store[n] >= 0 ? S.ev_store[n] : 0) & 1][decltype(it_)::value]); });
                finish(std::integral_constant<int, (S.ev_store[n] >= 0 ? S.ev_store[n] : 0)>{});
            }
            if constexpr (S.ev_load[n] >= 0) load_x(std::integral_constant<int, (S.ev_load[n] >= 0 ? S.ev_load[n] : 0)>{});
        };

        using I0 = std::integral_constant<int, 0>;
        static_for<0, D>(gather);
        wait_vmcnt<2 * (D - 1)>();
        load_w(I0{}, I0{});
        load_rows(I0{}, I0{});

        static_for<0, S.n>([&](auto in_) {
            constexpr int n = decltype(in_)::value, cur = n & 1;
            constexpr Q4bGroup G = S.g[n];
            constexpr int rb = G.cell & 1;                      // rows buffer of this cell
            constexpr bool has_next = n + 1 < S.n, next_cell = G.cell + 1 < 64;
            wait_lgkmcnt<0>();                                  // W(n) (requested a group ago) and — first group of a cell — the cell's rows
            static_for<0, G.npairs>([&](auto ip) { static_for<0, 3>([&](auto ie) { lds_tie(w[cur][decltype(ip)::value][decltype(ie)::value]); }); });
            if constexpr (G.first) { lds_tie(rows[rb][0]); lds_tie(rows[rb][1]); }
            if constexpr (G.first && G.cell + D < 64) gather(std::integral_constant<int, (G.cell + D < 64) ? G.cell + D : 0>{});   // this cell's slot has been read
            if constexpr (has_next) load_w(std::integral_constant<int, has_next ? n + 1 : n>{}, std::integral_constant<int, cur ^ 1>{});
            if constexpr (G.last && next_cell) {
                wait_vmcnt<S.vm_wait[n]>();                     // the next cell's rows have landed
                load_rows(std::integral_constant<int, next_cell ? G.cell + 1 : G.cell>{}, std::integral_constant<int, rb ^ 1>{});
            }
            static_for<0, 4>([&](auto ici) {
                constexpr int ci = decltype(ici)::value;
                static_for<0, G.npairs>([&](auto ip) {
                    constexpr int p = decltype(ip)::value, j = G.child[p];
#ifdef Q4_KO_MFMA
                    if constexpr (ci == 0) { acc0[j][0][0] += w[cur][p][0][0] * rows[rb][0][0]; acc0[j][1][0] += w[cur][p][1][0] * rows[rb][0][1]; acc1[j][0] += w[cur][p][2][0] * rows[rb][1][0]; }   // (timing experiment)
#else
                    acc0[j][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cur][p][0][ci], rows[rb][0][ci], acc0[j][0], 0, 0, 0);
                    acc0[j][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cur][p][1][ci], rows[rb][0][ci], acc0[j][1], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cur][p][2][ci], rows[rb][1][ci], acc1[j], 0, 0, 0);
#endif
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            events(in_);
        });
        events(std::integral_constant<int, S.n>{});
        events(std::integral_constant<int, S.n + 1>{});
        events(std::integral_constant<int, S.n + 2>{});
#ifdef Q4_KO_STORE
        q4_store(sink, rs_out, (unsigned)(p_base * 8u * out_row_bytes));
#endif
    }
}

template <int NW>
int launch_child_q4b16(const int32_t* pnbr, int64_t n_p, const float* in, int in_ld, const float* table, int table_bytes,
                       const IrnEpi& ep, hipStream_t s) {
    const size_t lds = (size_t)table_bytes + (size_t)NW * (Q4B_D * 2048);
    auto kern = k_child_q4b16<NW>;
    static ChildLdsGrant granted;
    if (int rc = child_lds_limit(kern, lds, granted)) return rc;
    const int64_t units = (n_p + 63) / 64;
    hipLaunchKernelGGL(kern, dim3(child_grid_units(units, NW, lds)), dim3(NW * 64), lds, s, pnbr, n_p, in, in_ld, table, table_bytes, ep);
    return 0;
}

}  // namespace
