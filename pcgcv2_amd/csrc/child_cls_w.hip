// One instantiation unit of the children-level plain convs / classification heads (kernels: child_kernels.h; entry point: child_conv.hip).
#include "child_kernels.h"

#define DEF_CONV_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                       int table_bytes, const ChildEpi& ep, hipStream_t s)
// classification heads 32 -> 1 (18 KB table, sixteen waves with four 2 KB ring slots each) and 64 -> 1 (34 KB table, seven waves with four
// 4 KB slots each).  A sweep of (waves, ring depth, groups per CU) moved neither by more than 3 % (profiles/r05_cls_compact.txt): with the
// tables out of the way the heads are bound by their own MFMA chain — 16 x 16 x 4 tiles with 8 of 16 columns used and zero rows for the
// (cell, child) pairs that do not exist, 0.21 of the issued products are real.
DEF_CONV_LAUNCH(pcgc_child_cls32) { return launch_child_cls<2, 16, 4>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
DEF_CONV_LAUNCH(pcgc_child_cls64) { return launch_child_cls<4, 7, 4>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
CHILD_TIMING_READER(pcgc_child_timing_cls_w)
