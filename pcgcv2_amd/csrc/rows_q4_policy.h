// Policies of the plain-level C = 32 quad-block passes (rows_q4.hip) for the engine of q4x.h: which cells, which groups, which accumulators.
// Plain constexpr C++17 (included inside rows_q4.hip's anonymous namespace, and by the CPU schedule test).
#pragma once

struct RowsQ4A32Base {
    static constexpr int NMAP = 27, ROW_MUL = 1, NACC = 4;      // accumulators: conv0_0 channels 0-3, 4-7; conv1_0 channels 0-3, 4-7
    static constexpr int NFRAG = 28 * 2 * 2;                    // [k (27 = conv1_0)][h][g] fragments of [co 4][ci 16]
};
// PAIRED: the two halves of a row are gathered back to back (q4x.h: Q4XSched::finish)
template <bool PAIRED>
struct RowsQ4A32 : RowsQ4A32Base {
    template <int MT, int D>
    static constexpr auto sched() {
        static_assert(!PAIRED || (D >= 4 && D % 2 == 0), "paired gathers: an even ring of at least four half cells");
        Q4XSched<54, 112> S{};
        for (int k = 0; k < 27; ++k)
            for (int h = 0; h < 2; ++h) {
                S.add_cell(k, 0, 64 * h);
                for (int g = 0; g < 2; ++g) S.add_group((k * 2 + h) * 2 + g, g, g, g, g, 0, 1, 2, 3);
                if (k == 13)
                    for (int g = 0; g < 2; ++g) S.add_group((27 * 2 + h) * 2 + g, 2 + g, 2 + g, 2 + g, 2 + g, 0, 1, 2, 3);
            }
        S.finish(MT, D, PAIRED);
        return S;
    }
};
struct RowsQ4B32 {
    static constexpr int NMAP = 27, ROW_MUL = 1, NACC = 6;      // conv0_1 channels 0-3 .. 12-15; conv1_1 channels 0-3, 4-7
    static constexpr int FRAG_W12 = 81, NFRAG = 83;             // [k][X0, X1, Y], then conv1_2's two fragments
    template <int MT, int D>
    static constexpr auto sched() {
        Q4XSched<27, 81> S{};
        for (int k = 0; k < 27; ++k) {
            S.add_cell(k, 0, 0);
            S.add_group(3 * k + 0, 0, 0, 1, 1, 0, 1, 0, 1);
            S.add_group(3 * k + 1, 2, 2, 3, 3, 0, 1, 0, 1);
            S.add_group(3 * k + 2, 4, 4, 5, 5, 2, 3, 2, 3);
        }
        S.finish(MT, D);
        return S;
    }
};
