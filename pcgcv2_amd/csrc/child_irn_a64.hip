// One instantiation unit of the children-level fused InceptionResNet passes (kernels: child_kernels.h; entry point: child_irn.hip).
#include "child_kernels.h"

#define DEF_IRN_LAUNCH(NAME) int NAME(int nw, const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                      int table_bytes, const IrnEpi& ep, hipStream_t s)
// The 112 KB table leaves 48 KB for the rings: six waves with two ring slots each (the block-wise main loop requests every operand one
// step ahead and needs the next cell's rows to have been requested a cell earlier).
DEF_IRN_LAUNCH(pcgc_irn_child_a64) { (void)nw; return launch_child_irn_a_split<64, 6, 2>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }      // half units
