// One instantiation unit of the children-level plain convs / classification heads (kernels: child_kernels.h; entry point: child_conv.hip).
#include "child_kernels.h"

#define DEF_CONV_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                       int table_bytes, const ChildEpi& ep, hipStream_t s)
// classification heads 32 -> 1 (64 KB table: one 16-wave group per CU) and 64 -> 1 (128 KB table: seven waves with one ring slot each —
// with four, the 1171 tiles of the 150 k-row level were two rounds of a tile that waits a gather round trip per cell: 84 us)
DEF_CONV_LAUNCH(pcgc_child_cls32) { return launch_child_cls<2, 16, 2>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
DEF_CONV_LAUNCH(pcgc_child_cls64) { return launch_child_cls<4, 7, 1>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
CHILD_TIMING_READER(pcgc_child_timing_cls_w)
