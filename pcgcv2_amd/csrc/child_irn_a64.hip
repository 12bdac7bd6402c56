// One instantiation unit of the children-level fused InceptionResNet passes (kernels: child_kernels.h; entry point: child_irn.hip).
#include "child_kernels.h"

#define DEF_IRN_LAUNCH(NAME) int NAME(int nw, const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                      int table_bytes, const IrnEpi& ep, hipStream_t s)
// the 112 KB table leaves one 8-wave group per CU and a single ring slot per wave (the next cell's gather flies behind the ~54 MFMAs
// of the current one)
DEF_IRN_LAUNCH(pcgc_irn_child_a64) { (void)nw; return launch_child_irn_a<64, 8, 1>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s); }
