// One instantiation unit of the children-level fused InceptionResNet passes (kernels: child_kernels.h; entry point: child_irn.hip).
#include "child_kernels.h"

#define DEF_IRN_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                      int table_bytes, const IrnEpi& ep, hipStream_t s)
// half units (see k_child_irn_a)
DEF_IRN_LAUNCH(pcgc_irn_child_b32) { return launch_child_irn_b_split<32, 12, 8>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
CHILD_TIMING_READER(pcgc_child_timing_b32)
