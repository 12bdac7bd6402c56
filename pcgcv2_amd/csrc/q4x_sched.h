// The schedule of the quad-block engine (q4x.h) as plain constexpr C++17 — no HIP in this header: tests/test_host_cpu.py compiles it with g++ and
// checks the counted waits against an independent re-simulation of the VMEM issue order (a count that is too large is a data race on the ring).
#pragma once

struct Q4XCell { int kp, row_off, byte_off; };                  // source row = map[kp][tile row] * ROW_MUL + row_off, bytes [byte_off, byte_off + 64)
struct Q4XGroup {
    int cell, frag;                                             // cell index; fragment index in the LDS table (256 bytes each)
    int acc[4], rowq[4];                                        // per weight quarter e: accumulator (-1: slot not issued), row quarter
    bool first, last;                                           // first / last group of its cell
};
template <int MAXC, int MAXG>
struct Q4XSched {
    int ncells, n;
    Q4XCell c[MAXC];
    Q4XGroup g[MAXG];
    int vm_wait[MAXG];                                          // last group of cell c: vmcnt that guarantees cell c + 1's rows have landed
    int issue[MAXC][2];                                         // cells whose gather is issued in the first group of cell c (-1: none)
    constexpr void add_cell(int kp, int row_off, int byte_off) { c[ncells++] = Q4XCell{kp, row_off, byte_off}; }
    constexpr void add_group(int frag, int a0, int a1, int a2, int a3, int q0, int q1, int q2, int q3) {
        g[n++] = Q4XGroup{ncells - 1, frag, {a0, a1, a2, a3}, {q0, q1, q2, q3}, false, false};
    }
    // marks first / last, plans the gathers and simulates the VMEM issue order: [gathers of cells 0 .. D - 1], then per group: (first: the
    // cell's planned gathers) (last: wait for cell + 1); 4 MT instructions per gather.
    // Plan: cell c + D is gathered once cell c's rows are in registers (its ring slot is free).  PAIRED (cells 2 k, 2 k + 1 are the two
    // 64-byte halves of ONE 128-byte row; D even, >= 4): both halves are requested back to back, in the first group of the ODD cell c — cells
    // c + D - 1 and c + D — so that the second request finds the line the first one brought into the L1 (requested a cell apart, 2.6x the
    // L1 misses of the packed-N kernel went to the L2: profiles/r06_rows_irn32_pmc.txt).
    constexpr void finish(int MT, int D, bool paired = false) {
        for (int i = 0; i < n; ++i) {
            g[i].first = i == 0 || g[i - 1].cell != g[i].cell;
            g[i].last = i == n - 1 || g[i + 1].cell != g[i].cell;
        }
        for (int cc = 0; cc < ncells; ++cc) {
            issue[cc][0] = issue[cc][1] = -1;
            if (!paired) { if (cc + D < ncells) issue[cc][0] = cc + D; }
            else if (cc & 1) {
                if (cc + D - 1 < ncells) issue[cc][0] = cc + D - 1;
                if (cc + D < ncells) issue[cc][1] = cc + D;
            }
        }
        int ops = 0;
        int gather_end[MAXC + 16] = {};
        for (int cc = 0; cc < D && cc < ncells; ++cc) { ops += 4 * MT; gather_end[cc] = ops; }
        for (int i = 0; i < n; ++i) {
            if (g[i].first)
                for (int j = 0; j < 2; ++j)
                    if (issue[g[i].cell][j] >= 0) { ops += 4 * MT; gather_end[issue[g[i].cell][j]] = ops; }
            if (g[i].last && g[i].cell + 1 < ncells) vm_wait[i] = ops - gather_end[g[i].cell + 1];
        }
    }
};
