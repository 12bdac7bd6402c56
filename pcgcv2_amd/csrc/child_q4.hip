// Entry point of the quad-block (v_mfma_f32_4x4x1_16b_f32) children-level kernels (kernels: child_q4.h).
#include "child_q4.h"

// Fused InceptionResNet (autoencoder.py:52-57) at C = 16 on a children level with the quad-block pass A:
//   pass 1 (A): in = x [8 n_parent, 16] -> out = t [8 n_parent, 8] = [relu(conv0_0 x + b0) | relu(conv1_0 x + b1)], stored in the T2
//               layout (per parent: [z half][conv][child & 3][4 channels]); table: ops.child_q4_tables (7168 bytes)
//   pass 2 (B): in = t (T2 layout) -> out [.., 16], the packed-N pass B of child_kernels.h with T2 gather addresses; table, biases and
//               residual as pcgc_irn_child_pass (ops.child_irn_tables()[1])
extern "C" int pcgc_irn_child_q4(const int32_t* parent_nbr, int64_t n_parent, int C, int pass, const float* in, int in_ld,
                                 const float* table, int64_t table_bytes, const float* b0, const float* b1, const float* b2,
                                 const float* x, int x_ld, float* out, int out_ld, void* stream) {
    CHILD_COMMON_CHECKS(in_ld)
    PCGC_REQUIRE(C == 16, "the quad-block kernels serve C = 16");
    PCGC_REQUIRE(pass == 1 || pass == 2, "pass must be 1 (A) or 2 (B)");
    PCGC_REQUIRE(out && b0 && b1 && (pass == 1 || (b2 && x)), "null argument");
    PCGC_REQUIRE((out_ld & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (pass == 1 || ((x_ld & 3) == 0 && (((uintptr_t)x) & 15) == 0)),
                 "rows must be 16-byte aligned");
    hipStream_t s = S(stream);
    IrnEpi ep{b0, b1, b2, x, x_ld, out, out_ld};
    int rc;
    if (pass == 1) {
        PCGC_REQUIRE(out_ld == C / 2, "pass A writes a dense [rows, C/2] tensor");
        PCGC_REQUIRE(table_bytes == 28 * 64 * 4, "pass A table size");
        rc = launch_child_q4<Q4_PASS_A, 8, 2>(parent_nbr, n_parent, in, in_ld, table, (int)table_bytes, ep, s);
    } else {
        PCGC_REQUIRE(in_ld == C / 2, "pass B reads the dense [rows, C/2] tensor of pass A");
        PCGC_REQUIRE(table_bytes == 85 * 64 * 4, "pass B table size");
        rc = launch_child_irn_b<16, 16, 8, 1, true>(parent_nbr, n_parent, in, in_ld, table, (int)table_bytes, ep, s);   // packed-N pass B with T2 gather addresses
    }
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("irn_child_q4");
    return 0;
}

// Classification head k3 16 -> 1 on a children level (autoencoder.py:228-234 conv2_cls; :174-180 / :201-207 at other widths stay on
// pcgc_conv_child) in quad-block form: out [8 n_parent, 1] dense.  table: ops.child_q4_cls_table (96 fragments of 256 bytes).
extern "C" int pcgc_cls_child_q4(const int32_t* parent_nbr, int64_t n_parent, const float* in, int Cin, int in_ld,
                                 const float* table, int64_t table_bytes, const float* bias, float* out, void* stream) {
    CHILD_COMMON_CHECKS(in_ld)
    PCGC_REQUIRE(Cin == 16, "the quad-block classification head serves C = 16");
    PCGC_REQUIRE(out && (((uintptr_t)out) & 15) == 0, "the output must be 16-byte aligned");
    PCGC_REQUIRE(table_bytes == 96 * 64 * 4, "cls table size");
    IrnEpi ep{bias, nullptr, nullptr, nullptr, 0, out, 1};
    const int rc = launch_child_q4<Q4_CLS, 8, 2>(parent_nbr, n_parent, in, in_ld, table, (int)table_bytes, ep, S(stream));
    if (rc) return rc;
    PCGC_CHECK_LAUNCH("cls_child_q4");
    return 0;
}
