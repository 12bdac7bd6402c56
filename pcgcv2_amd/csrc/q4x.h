// Quad-block engine (round 6): the 4x4x1 form of child_q4.h as a policy-driven main loop, for layers whose gathered rows are wider than one
// 64-byte cell or whose groups mix accumulators — the C = 32 InceptionResNet passes (Cout groups of 4 channels: conv0_0 / conv1_0 have two,
// conv0_1 four, conv1_1 two) on PLAIN levels (rows_q4.hip) where the packed-N 16x16x4 kernels multiply 2.7-3.9x zero-padded tiles
// (profiles/r06_rows_irn32_pmc.txt: pass A 107.9 k of its 130 k cycles per SIMD are matrix-pipe cycles, a quarter of them useful).
//
// Vocabulary.  A CELL is 64 bytes (16 channels) of one gathered row per tile row: map row kp, a row offset (children levels: the child) and
// a byte offset inside the row (rows wider than 64 bytes are two half cells).  A GROUP is 16 MT `v_mfma_f32_4x4x1_16b_f32`: slot (e, u),
// e = weight quarter (one ds_read_b128 of the group's 256-byte fragment [co 4][16 floats]), u = 0..3:
//     acc[m][G.acc[e]] += W_frag[co = lane & 3][4 e + u] * row(lane)[4 G.rowq[e] + u]           (one single-rounding fma per product)
// so a group is any 16 (input value, weight column) steps over the cell's 16 channels into up to four accumulators; slots whose acc is -1
// are not issued.  Lane = tile row (M tile m: row0 + 64 m + lane) and holds its row's four output channels per accumulator.
//
// Schedule (straight-line code over the policy's group list, as child_q4.h): weights of group n + 1 are requested at the start of group n
// (two register sets); the rows of cell c + 1 are read from the ring into a SECOND row register set at the start of cell c's last group,
// behind a counted vmcnt (constexpr simulation of the VMEM issue order) that guarantees they have landed; the gather of cell c + D is
// issued as soon as cell c's rows are in registers.  One lgkmcnt(0) per group: everything it covers was requested a group (128 MT pipe
// cycles) earlier.  Per output element the products arrive in the order of the policy's list: ascending cell, then ascending slot.
#pragma once
#include "child_kernels.h"

namespace {

#include "q4x_sched.h"

// Per-tile state of the engine: map entries -> byte offsets, the ring, the table base.  P (policy): NMAP (map rows), ROW_MUL, NACC, sched<MT, D>().
template <class P, int MT, int D>
struct Q4XTile {
    static constexpr auto S = P::template sched<MT, D>();
    static constexpr unsigned ABSENT = 0xF0000000u;
    static_assert(4 * MT * (D + 1) < 64, "vmcnt is 6 bits");

    // One tile: rows row0 + 64 m + lane.  pnbr [NMAP][n]; `in` rows of in_ld floats behind rs_in; ring: D slots of MT x 4 KB; acc out.
    __device__ __forceinline__ static void run(const int32_t* __restrict__ pnbr, int64_t n, int64_t row0, const __amdgpu_buffer_rsrc_t& rs_in,
                                               unsigned row_bytes, const unsigned char* lds_table, float4* ring, f32x4 (&acc)[MT][P::NACC]) {
        const int lane = threadIdx.x & 63;
        // map entries: lane L holds the entry of tile row 16 (L & 3) + (L >> 2) of each M tile; gather instruction d takes it from quad lane d
        unsigned rowb[MT][P::NMAP];
        bool ok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int64_t rho = row0 + 64 * m + 16 * (lane & 3) + (lane >> 2);
            ok[m] = rho < n;
#pragma unroll
            for (int kp = 0; kp < P::NMAP; ++kp) rowb[m][kp] = (unsigned)pnbr[(int64_t)kp * n + (ok[m] ? rho : 0)];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // map entries here, the previous tile's stores retired: vmcnt counts from zero
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int kp = 0; kp < P::NMAP; ++kp)
                rowb[m][kp] = (ok[m] && (int)rowb[m][kp] >= 0) ? rowb[m][kp] * ((unsigned)P::ROW_MUL * row_bytes) : ABSENT;
        const int co = lane & 3;
        const unsigned tab_lane = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)lds_table + co * 16);
        // chunk q of row `lane` sits at slot position q ^ ((row >> 2) & 3): address = a_row ^ (q << 4)
        const unsigned a_row = (unsigned)(uintptr_t)(lds_void_ptr)((const float*)ring + lane * 16 + (((lane >> 2) & 3) << 2));
        // gather instruction d of M tile m: lane L fetches for row 16 d + (L >> 2) the chunk that belongs at slot position L & 3
        const unsigned lane_off = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);

#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < P::NACC; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 a[2][MT][4], b[2][4];

        auto gather = [&](auto ic) {
            constexpr int c = decltype(ic)::value, slot = c % D;
            constexpr Q4XCell C = S.c[c];
            const unsigned cell_off = (unsigned)C.row_off * row_bytes + (unsigned)C.byte_off + lane_off;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int rb = (int)rowb[m][C.kp];
                const unsigned v0 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0x00, 0xF, 0xF, true) + cell_off;
                const unsigned v1 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0x55, 0xF, 0xF, true) + cell_off;
                const unsigned v2 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xAA, 0xF, 0xF, true) + cell_off;
                const unsigned v3 = (unsigned)__builtin_amdgcn_update_dpp(0, rb, 0xFF, 0xF, 0xF, true) + cell_off;
                float4* dst = ring + (slot * MT + m) * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst), 16, (int)v0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 64), 16, (int)v1, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 128), 16, (int)v2, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void_ptr)(dst + 192), 16, (int)v3, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        };
        auto load_a = [&](auto ic) {                           // the lane's 16 channels of cell c, every M tile, into register set c & 1
            constexpr int c = decltype(ic)::value, slot = c % D, buf = c & 1;
            static_for<0, MT>([&](auto im) {
                constexpr int m = decltype(im)::value;
                static_for<0, 4>([&](auto iq) {
                    constexpr int q = decltype(iq)::value;
                    a[buf][m][q] = lds_ld128_off<(slot * MT + m) * 4096>(a_row ^ (unsigned)(q << 4));
                });
            });
        };
        auto load_b = [&](auto in_) {
            constexpr int i = decltype(in_)::value, buf = i & 1;
            static_for<0, 4>([&](auto ie) {
                constexpr int e = decltype(ie)::value;
                if constexpr (S.g[i].acc[e] >= 0) b[buf][e] = lds_ld128_off<S.g[i].frag * 256 + e * 16>(tab_lane);
            });
        };

        using I0 = std::integral_constant<int, 0>;
        static_for<0, (D < S.ncells ? D : S.ncells)>(gather);
        wait_vmcnt<4 * MT * ((D < S.ncells ? D : S.ncells) - 1)>();
        load_b(I0{});
        load_a(I0{});

        static_for<0, S.n>([&](auto in_) {
            constexpr int i = decltype(in_)::value, cur = i & 1;
            constexpr Q4XGroup G = S.g[i];
            constexpr int abuf = G.cell & 1;
            constexpr bool has_next = i + 1 < S.n, next_cell = G.cell + 1 < S.ncells;
            wait_lgkmcnt<0>();
            static_for<0, 4>([&](auto ie) { if constexpr (G.acc[decltype(ie)::value] >= 0) lds_tie(b[cur][decltype(ie)::value]); });
            if constexpr (G.first)
                static_for<0, MT>([&](auto im) { static_for<0, 4>([&](auto iq) { lds_tie(a[abuf][decltype(im)::value][decltype(iq)::value]); }); });
            if constexpr (has_next) load_b(std::integral_constant<int, has_next ? i + 1 : i>{});
            if constexpr (G.first) {                           // this cell's slot is in registers: the planned gathers
                if constexpr (S.issue[G.cell][0] >= 0) gather(std::integral_constant<int, S.issue[G.cell][0] >= 0 ? S.issue[G.cell][0] : 0>{});
                if constexpr (S.issue[G.cell][1] >= 0) gather(std::integral_constant<int, S.issue[G.cell][1] >= 0 ? S.issue[G.cell][1] : 0>{});
            }
            if constexpr (G.last && next_cell) {
                wait_vmcnt<S.vm_wait[i]>();                     // the next cell's rows have landed in their ring slot
                load_a(std::integral_constant<int, next_cell ? G.cell + 1 : G.cell>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 4>([&](auto ie) {
                constexpr int e = decltype(ie)::value;
                if constexpr (G.acc[e] >= 0) {
                    static_for<0, 4>([&](auto iu) {
                        constexpr int u = decltype(iu)::value;
                        static_for<0, MT>([&](auto im) {
                            constexpr int m = decltype(im)::value;
                            acc[m][G.acc[e]] = __builtin_amdgcn_mfma_f32_4x4x1f32(b[cur][e][u], a[abuf][m][G.rowq[e]][u], acc[m][G.acc[e]], 0, 0, 0);
                        });
                    });
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
};

// Row-major staging of a tile's outputs through the (idle) ring: lane = tile row writes its NCH sixteen-byte chunks at XOR-swizzled chunk
// positions (a ds_write_b128 is served eight consecutive lanes at a time: their rows must fall on eight different 16-byte bank groups of
// 128 bytes); the read-back is lane-linear — piece q = lane + 64 i is chunk q % NCH of row q / NCH — so that a store instruction writes
// 1 KB of contiguous memory.  NCH = 4 (64-byte rows) or 8 (128-byte rows).
template <int NCH>
__device__ __forceinline__ unsigned q4x_stage_addr(int row, int chunk) {
    if constexpr (NCH == 4) return (unsigned)(row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4));
    else return (unsigned)(row * 128 + ((chunk ^ (row & 7)) << 4));
}

}  // namespace
