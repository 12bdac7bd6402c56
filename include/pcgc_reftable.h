/*
 * pcgc_reftable.h — C-ABI of libpcgc_reftable.so (host only): the factorized entropy bottleneck's CDF table evaluated with the
 * reference's own arithmetic.
 *
 * Replaces, for both EntropyBottleneck.compress and .decompress, the reference lines that turn (min_v, max_v) into the table handed
 * to torchac: entropy_model.py:157-172 / 180-189 -> `_likelihood` :112-130 -> `_logits_cumulative` :82-101 -> `_pmf_to_cdf`
 * :142-149, followed by torchac 0.9.3's `_convert_to_int_and_normalize` (16-bit normalisation).  The function issues the same ATen
 * CPU operators, in the same order and on the same tensor shapes as those Python lines (torch's CPU kernels select code by
 * shape and host), so on a given host the table equals the reference's bit for bit; it is pinned to golden tables generated from
 * the reference (tests/golden/entropy_tables.npz).  Links libtorch_cpu — the library the reference computes this table with.
 */
#ifndef PCGC_REFTABLE_H
#define PCGC_REFTABLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* params: the 12 parameter tensors flattened in the order _matrices.0..3 | _biases.0..3 | _factors.0..3 (44*C floats, filters
 * (3,3,3));  table_u16: [C, L+1] with L = max_v - min_v + 1 (uint16 bit patterns of torchac's int16 table);  cdf_f32: optional
 * [C, L+1] float cdf before normalisation.  Returns 0, -1 on an ATen error, -2 on bad arguments. */
int pcgc_reference_table(const float* params /*[host 44*C]*/, int C, float min_v, float max_v, uint16_t* table_u16 /*[host]*/,
                         float* cdf_f32 /*[host] or NULL*/);
/* The softplus / tanh of the parameter tensors (12 of the table's operators, functions of the parameters alone) are kept per
 * parameter set; this drops them, so that the next table is evaluated operator for operator as entropy_model.py:82-101 does on every
 * call (entropy_model.table_cache(clear=True) calls it together with pcgc_table_cache(0)).  Returns the number of sets dropped. */
int pcgc_reference_table_clear(void);
#ifdef __cplusplus
}
#endif
#endif
