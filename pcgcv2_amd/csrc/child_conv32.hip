// One instantiation unit of the children-level plain convs / classification heads (kernels: child_kernels.h; entry point: child_conv.hip).
#include "child_kernels.h"

#define DEF_CONV_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                       int table_bytes, const ChildEpi& ep, hipStream_t s)
// 32 -> 32: 108 KB of weights, one 8-wave workgroup per CU
// half units (4451 tiles of the 570 k-row level on 2048 waves were three rounds of whole tiles: 314.9 -> 271.8 us)
DEF_CONV_LAUNCH(pcgc_child_conv32) { return launch_child_conv_split<2, 2, 8, 2>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
CHILD_TIMING_READER(pcgc_child_timing_conv32)
