// Two M tiles per wave (32 parents): the C = 16 pass A with MT = 2 (kernels: child_kernels.h; entry point: child_irn.hip).
// Built in round 4 on the reviewer's request ("two M tiles per wave sharing every fragment read and wait"), bit-identical, and measured
// at the speed of the one-tile kernel in steady state (profiles/r04_child_mt2_experiment.md): NOT the product path.  One instantiation
// stays reachable through pcgc_set_child_tuning(316, 0) so that the MT = 2 form of the shared main loop keeps its parity test.
#include "child_kernels.h"

#define DEF_IRN_LAUNCH(NAME) int NAME(int nw, const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                      int table_bytes, const IrnEpi& ep, hipStream_t s)
DEF_IRN_LAUNCH(pcgc_irn_child_a16_mt2) {
    (void)nw;                                                  // <C, waves per workgroup, ring depth, M tiles>: 16 waves x 2 cells x 2 KB = the one-tile kernel's LDS
    return launch_child_irn_a<16, 16, 2, 2>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s);
}
CHILD_TIMING_READER(pcgc_child_timing_a16_mt2)
