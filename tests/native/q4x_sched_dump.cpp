// Dumps the compile-time schedules of the quad-block engine (pcgcv2_amd/csrc/q4x_sched.h + rows_q4_policy.h) as JSON lines: built with g++ by
// tests/test_host_cpu.py::test_q4x_schedules_counted_waits, which re-simulates the VMEM issue order independently.
#include <cstdio>
#include "../../pcgcv2_amd/csrc/q4x_sched.h"
#include "../../pcgcv2_amd/csrc/rows_q4_policy.h"

template <class S>
static void dump(const char* name, int MT, int D, bool paired, const S& s) {
    std::printf("{\"name\": \"%s\", \"MT\": %d, \"D\": %d, \"paired\": %s, \"ncells\": %d, \"cells\": [", name, MT, D, paired ? "true" : "false", s.ncells);
    for (int c = 0; c < s.ncells; ++c) std::printf("%s[%d, %d, %d, %d, %d]", c ? ", " : "", s.c[c].kp, s.c[c].row_off, s.c[c].byte_off, s.issue[c][0], s.issue[c][1]);
    std::printf("], \"groups\": [");
    for (int i = 0; i < s.n; ++i) {
        const auto& g = s.g[i];
        std::printf("%s{\"cell\": %d, \"frag\": %d, \"acc\": [%d, %d, %d, %d], \"rowq\": [%d, %d, %d, %d], \"first\": %d, \"last\": %d, \"vm_wait\": %d}", i ? ", " : "",
                    g.cell, g.frag, g.acc[0], g.acc[1], g.acc[2], g.acc[3], g.rowq[0], g.rowq[1], g.rowq[2], g.rowq[3], (int)g.first, (int)g.last, s.vm_wait[i]);
    }
    std::printf("]}\n");
}

int main() {
    // the instantiations pcgc_irn_rows_q4_pass launches (rows_q4.hip): (MT, D) = (2, 2), (1, 4) paired for pass A, (1, 2)
    { constexpr auto s = RowsQ4A32<false>::sched<2, 2>(); dump("A32", 2, 2, false, s); }
    { constexpr auto s = RowsQ4A32<true>::sched<1, 4>(); dump("A32", 1, 4, true, s); }
    { constexpr auto s = RowsQ4A32<false>::sched<1, 2>(); dump("A32", 1, 2, false, s); }
    { constexpr auto s = RowsQ4B32::sched<2, 2>(); dump("B32", 2, 2, false, s); }
    { constexpr auto s = RowsQ4B32::sched<1, 4>(); dump("B32", 1, 4, false, s); }
    { constexpr auto s = RowsQ4B32::sched<1, 2>(); dump("B32", 1, 2, false, s); }
    return 0;
}
