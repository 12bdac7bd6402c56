// C = 64 instantiations of the children-level fused InceptionResNet passes (kernels: child_kernels.h; entry point: child_irn.hip).
#include "child_kernels.h"

int pcgc_irn_child64_launch(int pass, int nw, const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table,
                            int table_bytes, const float* b0, const float* b1, const float* b2, const float* x, int x_ld, float* out,
                            int out_ld, hipStream_t s) {
    IrnEpi ep{b0, b1, b2, x, x_ld, out, out_ld};
    // pass A: the 112 KB table leaves one 8-wave group per CU and a single ring slot per wave (the next cell's gather flies behind the
    // ~54 MFMAs of the current one); pass B: 83 KB table, 8 waves, ring of 4 (also the epilogue scratch)
    if (pass == 1)
        return (nw == 4) ? launch_child_irn_a<64, 4, 1>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s)
                         : launch_child_irn_a<64, 8, 1>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s);
    return (nw == 4) ? launch_child_irn_b64<4, 4>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s)
                     : launch_child_irn_b64<8, 4>(parent_nbr, n_parent, in, in_ld, table, table_bytes, ep, s);
}
