// Entry points of the children-level plain convs and classification heads (kernels: child_kernels.h).
#include "child_kernels.h"


// wide instantiations: one translation unit each (child_conv_w.hip) so they compile in parallel
#define DECL_CONV_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                        int table_bytes, const ChildEpi& ep, hipStream_t s)
DECL_CONV_LAUNCH(pcgc_child_conv32); DECL_CONV_LAUNCH(pcgc_child_cls32); DECL_CONV_LAUNCH(pcgc_child_cls64);


// Plain k3 conv on a children level.  parent_nbr: [27][n_parent] k3 map of the PARENT level; in/out: children-level rows (8 n_parent).
// No residual form (round 5: no layer of the model has one outside the fused InceptionResNet passes; `residual` must be NULL).  table: the layer's `kernel` re-laid-out as B fragments (ops.child_conv_table).
extern "C" int pcgc_conv_child(const int32_t* parent_nbr, int64_t n_parent, const float* in, int Cin, int in_ld,
                               const float* table, int64_t table_bytes, const float* bias,
                               const float* residual, int res_ld, int relu, float* out, int Cout, int out_ld, void* stream) {
    CHILD_COMMON_CHECKS(in_ld)
    PCGC_REQUIRE(out != nullptr, "null output");
    PCGC_REQUIRE(Cout == 1 || ((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0), "output rows must be 16-byte aligned");
    PCGC_REQUIRE(residual == nullptr, "conv_child: no residual form");
    (void)res_ld;
    hipStream_t s = S(stream);
    int rc = -2;
    const int tb = (int)table_bytes;
    if (Cout == 1 && (Cin == 16 || Cin == 32 || Cin == 64)) {        // classification head: 8 columns = the 8 children
        PCGC_REQUIRE(table_bytes == (int64_t)(125 * ((Cin / 16) * 64 + 16) + 1023) / 1024 * 1024, "cls table size");
        PCGC_REQUIRE(residual == nullptr && !relu, "cls head has no fused epilogue");
        ChildEpi ep{bias, nullptr, 0, 0, out, out_ld, 0};
        // <16-channel blocks, waves per group, ring depth>: compact tables (10 / 18 / 34 KB), the rings take the rest of the LDS
        if (Cin == 64) rc = pcgc_child_cls64(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else if (Cin == 16) rc = launch_child_cls<1, 8, 4>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        else rc = pcgc_child_cls32(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
    } else {
        PCGC_REQUIRE(table_bytes == (int64_t)27 * Cin * Cout * 4, "table size");
        ChildEpi ep{bias, nullptr, 0, relu, out, out_ld, Cout / 16};
        if (Cin == 16 && Cout == 16) {
            // measured on 2.05 M rows: 4 waves per group (3 groups = 12 waves per CU) 251 us, 8 waves per group (16 per CU) 272 us
            rc = launch_child_conv<1, 1, 4, 4>(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        } else if (Cin == 32 && Cout == 32) {       // 108 KB of weights: one workgroup per CU
            rc = pcgc_child_conv32(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
        } else {
            pcgc_set_error("conv_child: unsupported shape %d -> %d (16->16, 32->32, 16->1, 32->1, 64->1)", Cin, Cout);
            return -2;
        }
    }
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("conv_child");
    return 0;
}


#ifdef PCGC_CHILD_TIMING
extern "C" int pcgc_child_timing(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_child_dbg), 8 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_child_dbg), z, sizeof(z)); }
    return 0;
}
#endif
