// One instantiation unit of the children-level fused InceptionResNet passes (kernels: child_kernels.h; entry point: child_irn.hip).
#include "child_kernels.h"

#define DEF_IRN_LAUNCH(NAME) int NAME(int nw, const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                      int table_bytes, const IrnEpi& ep, hipStream_t s)
// 83 KB table, 8 waves, ring of 4 (also the epilogue scratch); half units (child_kernels.h: k_child_irn_a)
DEF_IRN_LAUNCH(pcgc_irn_child_b64) { (void)nw; return launch_child_irn_b64_split<8, 4>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
