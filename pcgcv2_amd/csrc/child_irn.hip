// Entry point of the children-level fused InceptionResNet passes (kernels: child_kernels.h).
#include "child_kernels.h"

// Fused InceptionResNet passes on a children level (C = 16, 32).  pass 1 (A): in = x [8 n_parent, C] -> t [8 n_parent, C/2];
// pass 2 (B): in = t -> out [.., C] with the residual x.  tables: ops.child_irn_tables.
// The instantiations live in one translation unit each (child_irn_*.hip): every kernel is a fully unrolled 64-cell pipeline and takes
// 1-3 minutes to compile; separate units build in parallel.
#define DECL_IRN_LAUNCH(NAME) int NAME(const int32_t* parent_nbr, int64_t n_parent, const float* in, int in_ld, const float* table, \
                                       int table_bytes, const IrnEpi& ep, hipStream_t s)
DECL_IRN_LAUNCH(pcgc_irn_child_a16); DECL_IRN_LAUNCH(pcgc_irn_child_b16);
DECL_IRN_LAUNCH(pcgc_irn_child_a32); DECL_IRN_LAUNCH(pcgc_irn_child_b32);

extern "C" int pcgc_irn_child_pass(const int32_t* parent_nbr, int64_t n_parent, int C, int pass, const float* in, int in_ld,
                                   const float* table, int64_t table_bytes, const float* b0, const float* b1, const float* b2,
                                   const float* x, int x_ld, float* out, int out_ld, void* stream) {
    CHILD_COMMON_CHECKS(in_ld)
    PCGC_REQUIRE(C == 16 || C == 32, "channels must be 16 or 32 (C = 64 blocks run on pcgc_irn_rows_pass through the level's own map)");
    PCGC_REQUIRE(pass == 1 || pass == 2, "pass must be 1 (A) or 2 (B)");
    PCGC_REQUIRE(out && b0 && b1 && (pass == 1 || (b2 && x)), "null argument");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (pass == 1 || ((x_ld & 3) == 0 && (((uintptr_t)x) & 15) == 0)),
                 "rows must be 16-byte aligned");
    PCGC_REQUIRE(pass == 2 || out_ld == C / 2, "pass A writes a dense [rows, C/2] tensor");
    hipStream_t s = S(stream);
    IrnEpi ep{b0, b1, b2, x, x_ld, out, out_ld};
    const int tb = (int)table_bytes;
    int rc;
    if (pass == 1) {
        PCGC_REQUIRE(table_bytes == (int64_t)(C == 16 ? 52 : 38) * (C / 16) * 1024, "pass A table size");
        rc = C == 16 ? pcgc_irn_child_a16(parent_nbr, n_parent, in, in_ld, table, tb, ep, s) : pcgc_irn_child_a32(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
    } else {
        PCGC_REQUIRE(table_bytes == (int64_t)(C == 16 ? 85 * 64 * 4 : 64 * 128 * 4), "pass B table size");
        rc = C == 16 ? pcgc_irn_child_b16(parent_nbr, n_parent, in, in_ld, table, tb, ep, s) : pcgc_irn_child_b32(parent_nbr, n_parent, in, in_ld, table, tb, ep, s);
    }
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("irn_child_pass");
    return 0;
}

#ifdef PCGC_CHILD_TIMING
extern "C" int pcgc_child_timing_irn(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_child_dbg), 8 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_child_dbg), z, sizeof(z)); }
    return 0;
}
#endif
