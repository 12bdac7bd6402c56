/*
 * pcgc_hip.h — C-ABI of libpcgc_hip.so: the MI355X (gfx950) native operators under PCGCv2's encode/decode path.
 *
 * The reference (NJUVISION/PCGCv2) has no FFI layer of its own: its native code is reached through
 * `import MinkowskiEngine` and `import torchac`.  Each entry point below replaces one of those Python->native call
 * sites (cited as reference file:line); INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary.
 *   - every pointer marked [dev] is a device buffer owned by the caller (PyTorch-ROCm's allocator in our host code);
 *     [host] is host memory.  The library never allocates a result the caller must free.
 *   - variable-size results are two-phase: a count comes back through a [dev] int32, the caller sizes the output.
 *   - `stream` is a hipStream_t passed as void*; every device call is asynchronous on that stream.
 *   - return value: 0 = OK, <0 = error (message via pcgc_last_error()).
 *   - coordinates: int32 [N,4] rows (batch, x, y, z), 0 <= x,y,z < 2^20, 0 <= batch < 16  (ME.SparseTensor.C layout).
 *   - features: fp32 row-major, leading dimension given explicitly (`*_ld`, in floats) so concat / slices are views.
 *   - kernel maps: int32 [K][N_out], entry = input row feeding output row o through kernel offset k, or -1.
 *     offset order: k3 -> (k%3-1, (k/3)%3-1, k/9-1)*stride ; k2 -> (k&1, (k>>1)&1, k>>2)*stride  (x fastest, ME ‡).
 *   - canonical arithmetic: out = ( fp32 fmaf chain over k ascending, ci ascending, from +0 ) + bias ; then
 *     + residual ; then ReLU.   Bit-identical to oracle/pcgc_oracle.c.
 */
#ifndef PCGC_HIP_H
#define PCGC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* pcgc_last_error(void);
int pcgc_version(void);

/* ---- coordinate hash map: replaces ME's CoordinateManager / coordinate map (ME.SparseTensor ctor at
 *      data_utils.py:96,108,116 and coder.py:102).  Open addressing, 64-slot spatial blocks (4x4x4 voxels). ---- */
int64_t pcgc_hash_capacity(int64_t n);                                   /* slots needed for n coordinates (pow2) */
int pcgc_hash_clear(uint64_t* keys /*[dev cap]*/, int32_t* vals /*[dev cap]*/, int64_t cap, void* stream);
/* `stride` = tensor stride of the level the table indexes.  Kept in the signatures for the callers' bookkeeping: the slot of a
 * key is a mix of the whole coordinate key and does not depend on it (the round-1 spatially blocked layout did). */
int pcgc_hash_insert(const int32_t* coords /*[dev n,4]*/, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals,
                     int64_t cap, void* stream);                         /* vals[slot] = smallest row with that key */
/* dedup policy as an argument: keep_last = 0 is pcgc_hash_insert; 1 keeps the LARGEST row per key (ME's dedup policy for equal
 * coordinates is a ‡ convention; hash_first_mask then marks the kept rows either way: keep[i] = (vals[slot] == i)). */
int pcgc_hash_insert_policy(const int32_t* coords, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals, int64_t cap,
                            int keep_last, void* stream);
int pcgc_hash_first_mask(const int32_t* coords, int64_t n, int32_t stride, const uint64_t* keys, const int32_t* vals,
                         int64_t cap, uint8_t* keep /*[dev n]*/, int32_t* first_row /*[dev n] or NULL*/, void* stream);
                         /* keep[i] = row i is the first occurrence of its coordinate; first_row[i] = that first row */

/* Validation of caller-supplied coordinates (ME accepts any int32 coordinate; this library keys levels by 4+20+20+20 bits, so the host layer
 * — SparseTensor ctor, scale_sparse_tensor — raises instead of silently dropping rows) + how ordered the rows are: out2[0] = rows out of range
 * (negative, >= 2^20, batch >= 16), out2[1] = descents of the (batch, z, y, x) key along the
 * rows (0: the rows are in sort_spare_tensor's order, data_utils.py:91-101).  The reference takes rows in any order (ME hashes them); here
 * the canonical row order of every level follows the input order, so an unordered PLY would drive every encoder gather through a random
 * row order — Coder.encode sorts such a cloud once at ingest (no output byte depends on it: the latent is sorted before coding). */
int pcgc_coords_check_order(const int32_t* coords /*[dev n,4]*/, int64_t n, int32_t* out2 /*[dev 2]*/, void* stream);

/* ---- coordinate transforms ---- */
/* output coords of MinkowskiConvolution(kernel_size=2, stride=2): floor(c / stride_out) * stride_out per row
 * (autoencoder.py:78-84,97-103,116-122); dedup with the hash calls above. */
int pcgc_coords_quantize(const int32_t* coords, int64_t n, int32_t stride_out, int32_t* out /*[dev n,4]*/, void* stream);
/* output coords of MinkowskiGenerativeConvolutionTranspose(k=2,s=2): row 8*i+k = c_i + (k&1,(k>>1)&1,k>>2)*stride_in/2
 * (autoencoder.py:155-161,182-188,209-215). */
int pcgc_coords_children(const int32_t* coords, int64_t n, int32_t stride_in, int32_t* out /*[dev 8n,4]*/, void* stream);
/* scale_sparse_tensor: round_half_even(float32(c) * float32(factor)) (data_utils.py:112-118); batch column kept. */
int pcgc_coords_scale(const int32_t* coords, int64_t n, float factor, int32_t* out /*[dev n,4]*/, void* stream);

/* ---- stream compaction (MinkowskiPruning, autoencoder.py:237,247; also dedup) ---- */
size_t pcgc_scan_workspace_bytes(int64_t n);
/* prefix[i] = number of set mask bytes before i; *total = number set. */
int pcgc_mask_scan(const uint8_t* mask /*[dev n]*/, int64_t n, int32_t* prefix /*[dev n]*/, int32_t* total /*[dev 1]*/,
                   void* workspace /*[dev]*/, size_t workspace_bytes, void* stream);
/* the same for a workspace (and *total, if n may be 0) the caller has already zeroed: several scans behind one memset */
int pcgc_mask_scan_zeroed(const uint8_t* mask, int64_t n, int32_t* prefix, int32_t* total, void* workspace, size_t workspace_bytes,
                          void* stream);
int pcgc_compact_coords(const int32_t* coords, const uint8_t* mask, const int32_t* prefix, int64_t n,
                        int32_t* out /*[dev total,4]*/, void* stream);
int pcgc_compact_feats(const float* in, int C, int in_ld, const uint8_t* mask, const int32_t* prefix, int64_t n,
                       float* out /*[dev total,C]*/, void* stream);

/* ---- kernel maps (ME kernel maps, built once per coordinate level and reused by every conv on it) ---- */
int pcgc_kmap_k3(const int32_t* coords, int64_t n, int32_t stride, const uint64_t* keys, const int32_t* vals,
                 int64_t cap, int32_t* nbr /*[dev 27,n]*/, void* stream);

/* Hierarchical kernel maps: a level's 27-neighbourhood is a gather through its PARENT level's map (octree relation), so
 * only the coarsest level of a pyramid probes the hash.  (ME rebuilds a hash-probed map per level ‡.) */
/* children of a generative transpose (all 8 children exist, rows 8*i+j):  [27, 8*n_parent] from [27, n_parent] */
int pcgc_kmap_k3_children(const int32_t* parent_nbr, int64_t n_parent, int32_t* nbr /*[dev 27,8n]*/, void* stream);
/* pruned level: surviving rows orig[r] of a candidate level, neighbours renumbered through mask/prefix */
int pcgc_kmap_k3_prune(const int32_t* cand_nbr /*[27,n_cand]*/, int64_t n_cand, const uint8_t* mask, const int32_t* prefix,
                       const int32_t* orig /*[n_out]*/, int64_t n_out, int32_t* nbr /*[dev 27,n_out]*/, void* stream);
/* the same pruned-level map derived straight from the PARENT level's map when the candidate level is a children level
 * (mask / prefix / orig index candidate rows 8 i + j): the candidates' own [27][8 n_parent] map is never materialised. */
int pcgc_kmap_k3_prune_parent(const int32_t* parent_nbr /*[dev 27,n_parent]*/, int64_t n_parent, const uint8_t* mask,
                              const int32_t* prefix, const int32_t* orig, int64_t n_out, int32_t* nbr /*[dev 27,n_out]*/, void* stream);
/* strided pyramid (encoder): fine level from the coarse level's map + the down map + each fine row's parent row */
int pcgc_kmap_k3_from_coarse(const int32_t* fine /*[n_fine,4]*/, int64_t n_fine, int32_t stride_fine,
                             const int32_t* parent_of /*[n_fine]*/, const int32_t* coarse_nbr /*[27,n_coarse]*/,
                             const int32_t* down /*[8,n_coarse]*/, int64_t n_coarse, int32_t* nbr /*[dev 27,n_fine]*/,
                             void* stream);
/* parent_of[c] = prefix[first_row[c]]; down[slot(c)][parent_of[c]] = c — the k2s2 kernel map without hash probes */
int pcgc_down_maps(const int32_t* fine, const int32_t* first_row, const int32_t* prefix, int64_t n_fine, int32_t stride_fine,
                   int64_t n_coarse, int32_t* parent_of /*[dev n_fine]*/, int32_t* down /*[dev 8,n_coarse]*/, void* stream);
/* One strided pyramid level = pcgc_coords_quantize + hash clear/insert/first_mask + pcgc_mask_scan (prepare), then — after
 * the host has read *total = n_coarse — pcgc_compact_coords + pcgc_down_maps (finish).  Same kernels, two calls. */
int pcgc_down_prepare(const int32_t* fine /*[dev n,4]*/, int64_t n, int32_t stride_fine, int32_t* q /*[dev n,4] quantised*/,
                      uint64_t* keys, int32_t* vals, int64_t cap, uint8_t* keep /*[dev n]*/, int32_t* first_row /*[dev n]*/,
                      int32_t* prefix /*[dev n]*/, int32_t* total /*[dev 1]*/, void* scan_ws, size_t scan_ws_bytes, void* stream);
int pcgc_down_finish(const int32_t* fine, const int32_t* q, const uint8_t* keep, const int32_t* first_row, const int32_t* prefix,
                     int64_t n, int32_t stride_fine, int64_t n_coarse, int32_t* coarse /*[dev n_coarse,4]*/,
                     int32_t* parent_of /*[dev n]*/, int32_t* down /*[dev 8,n_coarse]*/, void* stream);
/* prepare + read-back of n_coarse (synchronises `stream`) + finish in one call.  coarse / down are caller buffers of upper-bound
 * size ([n,4] and 8*n int32: a coarse level has at most n rows); down is written as [8][n_coarse] at the front of its buffer. */
int pcgc_down_level(const int32_t* fine /*[dev n,4]*/, int64_t n, int32_t stride_fine, int32_t* q, uint64_t* keys, int32_t* vals,
                    int64_t cap, uint8_t* keep, int32_t* first_row, int32_t* prefix, int32_t* total, void* scan_ws,
                    size_t scan_ws_bytes, int32_t* coarse /*[dev n,4] capacity*/, int32_t* parent_of /*[dev n]*/,
                    int32_t* down /*[dev 8*n] capacity*/, int64_t* n_coarse_out /*host*/, void* stream);
/* `levels` (1..4) successive levels at once, every one deduplicated straight from the input rows (the canonical order of a level —
 * first occurrence in the level below — equals first occurrence in the input), so that the sizes of all levels come back in ONE
 * synchronising copy.  Level l (0-based) has stride `stride << (l + 1)`; coarse[l] / down[l] are caller buffers of upper-bound size
 * ([n,4] and 8*n int32), parent_of[l] has one entry per row of the level below it ([n] for l = 0, counts[l-1] after); counts[l] (host)
 * = rows of level l.  scratch: pcgc_pyramid_scratch_bytes(n, levels) device bytes, 16-byte aligned. */
size_t pcgc_pyramid_scratch_bytes(int64_t n, int levels);
int pcgc_pyramid(const int32_t* fine /*[dev n,4]*/, int64_t n, int32_t stride, int levels, void* scratch, size_t scratch_bytes,
                 int32_t* const* coarse, int32_t* const* parent_of, int32_t* const* down, int64_t* counts /*host*/, void* stream);
/* orig[prefix[i]] = i for set mask bytes (row indices that survive a compaction) */
int pcgc_compact_index(const uint8_t* mask, const int32_t* prefix, int64_t n, int32_t* orig /*[dev total]*/, void* stream);

/* ---- sparse convolution family ---- */
/* Gather convolution: MinkowskiConvolution k=3,s=1 (K=27), k=2,s=2 (K=8), k=1 (K=1, nbr may be NULL = identity)
 * (autoencoder.py:13-48,71-134,162-234).  W = ME `kernel` [K,Cin,Cout]; bias [Cout] or NULL;
 * residual (same row, columns res_coff..) or NULL (SparseTensor.__add__, autoencoder.py:55); relu = MinkowskiReLU. */
int pcgc_conv_gather(const int32_t* nbr, int K, int64_t n_out, const float* in, int64_t n_in /*rows of `in`*/, int Cin,
                     int in_ld, int in_coff, const float* W, const float* bias, const float* residual, int res_ld,
                     int res_coff, int relu, float* out, int Cout, int out_ld, int out_coff, void* stream);
/* Which kernel family the calling thread's last pcgc_conv_gather launched (0 v0 VALU | 1 v1 LDS-DMA + VALU | 2 v2 MFMA | 3 v2b MFMA with
 * LDS-shared weights, 2 M tiles | 4 v2b, 4 M tiles | 5 v1 burst | 6 row-split | 7 v2c MFMA, both operands in LDS; -1 none yet).  The policy
 * is written down once, in pcgcv2_amd/dispatch.py; the parity tests read this back for every entry of that table.  HOST. */
int pcgc_last_conv_impl(void);
/* kernel selection for the gather conv: -1 auto (default), 0 = v0 direct-load kernel, 1 = v1 LDS-DMA kernel, 2 / 3 = the fp32-MFMA
 * kernels, 5 / 6 = the 16-row "burst" and row-split forms that small levels get (each where eligible).  All produce bit-identical results; the
 * switch exists for A/B measurements and tests. */
/* The codec's first layer (Encoder.conv0, autoencoder.py:71-77) on its actual input, the all-ones single-channel occupancy indicator
 * (data_utils.py:104,114): out = relu?(sum over PRESENT offsets k, ascending, of W[k][0][:] + bias).  fmaf(1, w, acc) = acc + w exactly,
 * so this IS pcgc_conv_gather's chain on that input, without the 27 feature gathers per row.  W: [K, 1, Cout]. */
int pcgc_conv_gather_unit(const int32_t* nbr /*[dev K,n_out]*/, int K, int64_t n_out, const float* W, const float* bias, int relu,
                          float* out, int Cout, int out_ld, void* stream);
/* the same layer on a level whose own kernel map has not been built (a level of the encoder's strided pyramid): presence of every offset
 * is derived on the fly from the PARENT level's map + the level pair's down map + parent_of — what pcgc_kmap_k3_from_coarse would write —
 * so the finest level's [27][n] map, which only this layer reads, is never materialised. */
int pcgc_conv_unit_from_coarse(const int32_t* fine /*[n_fine,4]*/, int64_t n_fine, int32_t stride_fine, const int32_t* parent_of,
                               const int32_t* coarse_nbr /*[27,n_coarse]*/, const int32_t* down /*[8,n_coarse]*/, int64_t n_coarse,
                               const float* W /*[27,1,Cout]*/, const float* bias, int relu, float* out, int Cout, int out_ld, void* stream);
/* force a family of pcgc_conv_gather: -1 auto (default) | 0 VALU | 2 MFMA | 6 row-split (A/B tests; bit-identical) */
int pcgc_set_conv_impl(int impl);
/* generative transpose 64->32 / 32->16: 2 fp32-MFMA kernel with the weight fragments resident in LDS (default), 1 fragments read from L2, 0 VALU kernel.  Bit-identical. */
int pcgc_set_up2_impl(int mfma);
/* Fused InceptionResNet block (autoencoder.py:7-57):  out = cat(conv0_1(relu(conv0_0 x)), conv1_2(relu(conv1_1(relu(conv1_0 x))))) + x
 * in two gather passes.  params[10] = {conv0_0.kernel, .bias, conv0_1.kernel, .bias, conv1_0.kernel, .bias, conv1_1.kernel,
 * .bias, conv1_2.kernel, .bias} (ME layouts).  t_scratch: [n, C/2] fp32 workspace.  Bit-identical to the five
 * pcgc_conv_gather calls it replaces. */
int pcgc_irn_block(const int32_t* nbr /*[27,n]*/, int64_t n, const float* x /*[n,C], ld x_ld*/, int C, int x_ld,
                   const float* const* params, float* t_scratch, float* out, int out_ld, void* stream);
/* the two gather passes of pcgc_irn_block separately (pass 1 = A: x -> t_scratch, pass 2 = B: t_scratch, x -> out);
 * same arguments; used to time the passes individually. */
int pcgc_irn_pass(const int32_t* nbr, int64_t n, const float* x, int C, int x_ld, const float* const* params,
                  float* t_scratch, float* out, int out_ld, int pass, void* stream);
/* MinkowskiGenerativeConvolutionTranspose(k=2,s=2): out[8i+k] = in[i] @ W[k] + bias (+ReLU). */
int pcgc_conv_up2(int64_t n_in, const float* in, int Cin, int in_ld, const float* W /*[8,Cin,Cout]*/, const float* bias,
                  int relu, float* out /*[dev 8n,Cout]*/, int Cout, void* stream);
/* the same on a pruned level read in place (autoencoder.py:247 followed by :155-161 / :182-188 of the next stage): input row p = row rows[p]
 * of `in`, rows = the survivors' candidate rows (pcgc_topk_select's orig), so MinkowskiPruning's compacted feature tensor is never written.
 * Returns -3 (nothing launched, no error text) for shapes other than 64 -> 32 / 32 -> 16: gather the rows, then pcgc_conv_up2. */
int pcgc_conv_up2_gather(int64_t n_in, const float* in, int Cin, int in_ld, const int32_t* rows, const float* W /*[8,Cin,Cout]*/,
                         const float* bias, int relu, float* out /*[dev 8n,Cout]*/, int Cout, void* stream);

/* ---- k3 convolution on a CHILDREN level (the output level of a generative transpose: row 8p+j = child j of parent p) through
 *      the PARENT level's kernel map: each neighbour row is gathered once per 16-parent tile and feeds every child that
 *      reaches it (csrc/child.hip).  Replaces MinkowskiConvolution(k=3) at the decoder call sites autoencoder.py:162-168,
 *      189-195,216-222 (conv0/1/2 after up0/1/2).  `table` = the layer's `kernel` re-laid-out as MFMA B fragments
 *      [k][Cout/16][Cin/16][lane 64][4] (pcgcv2_amd/ops.py:child_conv_table).  Same canonical arithmetic. ---- */
int pcgc_conv_child(const int32_t* parent_nbr /*[dev 27,n_parent]*/, int64_t n_parent, const float* in /*[dev 8 n_parent rows]*/,
                    int Cin, int in_ld, const float* table /*[dev]*/, int64_t table_bytes,
                    const float* bias, const float* residual, int res_ld, int relu, float* out, int Cout, int out_ld, void* stream);
/* Cout == 1 (the classification heads conv0/1/2_cls, autoencoder.py:169-175,196-202,223-229) is served by the same entry with a
 * 64-fragment table (ops.child_cls_table): the 8 children are the 8 used columns of one accumulator tile. */
/* Fused InceptionResNet (autoencoder.py:52-57) on a children level, C = 16 or 32, as two parent-map passes:
 *   pass 1 (A): in = x [8 n_parent, C]      -> out = t [8 n_parent, C/2] = [relu(conv0_0 x + b0) | relu(conv1_0 x + b1)]
 *   pass 2 (B): in = t [8 n_parent, C/2]    -> out [.., C] = [conv0_1(t[:, :Q]) + b0 | conv1_2(relu(conv1_1(t[:, Q:]) + b1)) + b2] + x
 * The narrow layers pack (child, output channel) pairs into the MFMA N dimension; tables: ops.child_irn_tables. */
int pcgc_irn_child_pass(const int32_t* parent_nbr, int64_t n_parent, int C, int pass, const float* in, int in_ld,
                        const float* table, int64_t table_bytes, const float* b0, const float* b1, const float* b2,
                        const float* x, int x_ld, float* out, int out_ld, void* stream);
/* The same block at C = 16 with pass 1 (A) in QUAD-BLOCK form (round 5, csrc/child_q4.h): `v_mfma_f32_4x4x1_16b_f32`, one 4 x 4 block per
 * (4 parents, cell, child) — only the (cell, child) pairs that exist are issued, where the packed-N tiles of pcgc_irn_child_pass multiply
 * 44 % structural zero columns.  Replaces autoencoder.py:52-57 (InceptionResNet) on the level of autoencoder.py:209-237, bit for bit the
 * same chains.  pass 1: table = ops.child_q4_tables: [27][4 co][16 ci] = conv0_0.kernel, then [4 co][16 ci] = conv1_0.kernel (7168
 * bytes); it writes t in the T2 layout (per parent 256 bytes = [z half][conv0_0 | conv1_0][child & 3][4 channels]: a quad of lanes stores
 * 64 contiguous bytes).  pass 2: the packed-N pass B of pcgc_irn_child_pass gathering through the T2 layout (table =
 * ops.child_irn_tables()[1]).  The two passes of one block must come from the same entry point.  Arguments as pcgc_irn_child_pass. */
int pcgc_irn_child_q4(const int32_t* parent_nbr, int64_t n_parent, int C, int pass, const float* in, int in_ld,
                      const float* table, int64_t table_bytes, const float* b0, const float* b1, const float* b2,
                      const float* x, int x_ld, float* out, int out_ld, void* stream);
/* Classification head k3 16 -> 1 on a children level (autoencoder.py:228-234 conv2_cls) in quad-block form (csrc/child_q4.h): a 4 x 4
 * block = 4 parents x the four children of one z half; out [8 n_parent, 1] dense, a lane stores its parent's eight logits (32 bytes).
 * table = ops.child_q4_cls_table (96 fragments [4 children][16 ci], zero where a child does not reach the cell).  Same chain as
 * pcgc_conv_child with Cout = 1. */
int pcgc_cls_child_q4(const int32_t* parent_nbr, int64_t n_parent, const float* in, int Cin, int in_ld, const float* table,
                      int64_t table_bytes, const float* bias, float* out, void* stream);
/* The same two passes at C = 64 on a PLAIN level (the encoder's stride-4 level, autoencoder.py:104-110) through the level's own
 * k3 map nbr [27][n]: LDS-resident fragment table, one wave per 16-row tile walking the 27 offsets (csrc/rows_irn.hip); tables as
 * for the children-level C = 64 passes (ops.child_irn_tables). */
int pcgc_irn_rows_pass(const int32_t* nbr, int64_t n, int C, int pass, const float* in, int in_ld, const float* table,
                       int64_t table_bytes, const float* b0, const float* b1, const float* b2, const float* x, int x_ld, float* out,
                       int out_ld, void* stream);
/* The two passes at C = 32 on a plain level (the encoder's block0 / block2, autoencoder.py:85-89,123-127) in QUAD-BLOCK form
 * (csrc/q4x.h, csrc/rows_q4.hip: v_mfma_f32_4x4x1_16b_f32, lane = row, no zero column).  pass 1 (A): in = x [n, >= 32] -> out = t [n, 16]
 * dense; pass 2 (B): in = t -> out [n, >= 32] with the residual x.  tables = ops.rows_q4_tables(params): pass A 112 fragments [co 4][ci 16]
 * ordered [k (27 = conv1_0)][channel half][output group], pass B 83 ([k][conv0_1 outputs 0-7, 8-15, conv1_1 0-7], then conv1_2's two).
 * Same chain as pcgc_irn_rows_pass / pcgc_irn_block: bit-identical. */
int pcgc_irn_rows_q4_pass(const int32_t* nbr, int64_t n, int C, int pass, const float* in, int in_ld, const float* table,
                          int64_t table_bytes, const float* b0, const float* b1, const float* b2, const float* x, int x_ld, float* out,
                          int out_ld, void* stream);
/* A/B knob of that entry (tools/rows32_ab.py): 0 = the default instantiation (3); 1 = 8 waves x 2 M tiles x ring 2; 2 = 8 x 1 x 4 with paired
 * half-row gathers; 3 = pass A 16 x 1 x 2, pass B 12 x 1 x 2.  Results do not depend on it.  Replaces nothing in the reference. */
int pcgc_set_rows_q4_variant(int v);
/* Plain k3 conv 32 -> 32 on a plain level (the encoder's conv1, autoencoder.py:90-96) by the same kernel family: table =
 * ops.child_conv_table(kernel) (108 KB, LDS-resident); epilogue as pcgc_conv_gather. */
int pcgc_conv_rows(const int32_t* nbr, int64_t n, const float* in, int Cin, int in_ld, const float* table, int64_t table_bytes,
                   const float* bias, const float* residual, int res_ld, int relu, float* out, int Cout, int out_ld, void* stream);
/* k2 s2 down conv (the encoder's down0 / down1 / down2, autoencoder.py:78-84,97-103,116-122: 16 -> 32, 32 -> 64, 64 -> 32) by the same
 * kernel family: tile = 16 coarse rows, the 8 child offsets through the level pair's `down` map [8][n_coarse] (fine rows, -1 = absent);
 * table = ops.child_conv_table(kernel). */
int pcgc_conv_down_rows(const int32_t* down, int64_t n_coarse, const float* in, int64_t n_in, int Cin, int in_ld, const float* table,
                        int64_t table_bytes, const float* bias, int relu, float* out, int Cout, int out_ld, void* stream);

/* k3 conv 64 -> 64 (the encoder's conv2, autoencoder.py:109-115, and the decoder's conv0, :162-168 — the two layers whose 442 KB of weights
 * fit no LDS-resident table) with PRESENT-ROW PACKING: per 128-row workgroup tile and kernel offset only the rows that have the neighbour
 * are gathered and multiplied (packed 16 at a time; accumulators in LDS), instead of zero rows for absent neighbours.  nbr [27][n] of the
 * level itself (-1 = absent); table = ops.child_conv_table(kernel) (442 368 bytes); out = (acc + bias) (relu).  Same fmaf chain per
 * output element as pcgc_conv_gather: bit-identical results. */
int pcgc_conv_packed64(const int32_t* nbr, int64_t n, const float* in, int in_ld, const float* table, int64_t table_bytes,
                       const float* bias, int relu, float* out, int out_ld, void* stream);
/* A/B switch of pcgc_conv_packed64: rows per workgroup tile (1 .. 128; 0 = chosen per launch from the level size); `waves` must be 0 or 4
 * (the eight-wave form was removed in round 5).  Results do not depend on it. */
int pcgc_set_packed_tuning(int rows, int waves);

/* ‡ conventions the reference's results depend on but its sources do not pin (un-vendored MinkowskiEngine / torch.topk on ME's row
 * order): what = 0: top-k tie rule, value 0 = the lower row wins (default), 1 = the higher row wins.  The dedup policy is an argument of
 * pcgc_hash_insert_policy; the kernel-offset order is a weight permutation done by the host (pcgcv2_amd/conventions.py). */
int pcgc_set_convention(int what, int value);

/* ---- top-k pruning mask: istopk (data_utils.py:77-89).  mask[i]=1 for the k largest logits;
 *      ties -> lower row index; -0.0 == +0.0. ---- */
size_t pcgc_topk_workspace_bytes(int64_t n);
int pcgc_topk_mask(const float* logits /*[dev n], stride ld*/, int ld, int64_t n, int64_t k, uint8_t* mask /*[dev n]*/,
                   void* workspace, size_t workspace_bytes, void* stream);
/* ---- batches (ME.utils.sparse_collate, data_utils.py:107: batch index in column 0; istopk loops over the items, :80-87).  Every
 *      level of a collated batch is the concatenation of its items' levels, so the per-item operators work on contiguous row
 *      segments: seg_rows[b] rows for item b (HOST arrays; workspace as for the largest segment). ---- */
int pcgc_topk_mask_segments(const float* logits, int ld, int nseg, const int64_t* seg_rows, const int64_t* seg_k, uint8_t* mask,
                            void* workspace, size_t workspace_bytes, void* stream);
/* ---- prune_voxel in one sweep (round 4): istopk (data_utils.py:77-89) + MinkowskiPruning (autoencoder.py:237,247) of a candidate level.
 *      After the radix passes one single-pass scan keeps, per segment b, the seg_k[b] largest logits of its seg_rows[b] rows (same tie rule
 *      and ‡ convention switch as pcgc_topk_mask) and writes the pruned level directly: out_coords [K,4] and orig [K] (the candidate row of
 *      every surviving row, ascending), K = sum of the clamped seg_k, plus a RANK BITMAP of the candidate level for the kernel-map
 *      derivation: bits = one bit per candidate row (row m = bit m & 7 of byte m >> 3; ((n + 63) / 64) * 8 bytes, 8-byte aligned),
 *      wprefix [(n + 63) / 64] = survivors before row 64 w.  The candidates' coordinates are either given (`coords` [n,4]) or — a children
 *      level, rows 8 i + j of a generative transpose that never materialised its coordinates — derived from `parent_coords` [n / 8, 4] at
 *      tensor stride `parent_stride` (exactly one of the two is non-NULL).  seg_rows / seg_k: HOST arrays. ---- */
size_t pcgc_topk_select_workspace_bytes(int64_t n);
int pcgc_topk_select(const float* logits /*[dev n], stride ld*/, int ld, int nseg, const int64_t* seg_rows, const int64_t* seg_k,
                     const int32_t* coords, const int32_t* parent_coords, int32_t parent_stride,
                     uint8_t* bits, int32_t* wprefix, int32_t* orig, int32_t* out_coords, void* workspace, size_t workspace_bytes, void* stream);
/* pcgc_kmap_k3_prune / _prune_parent through that rank bitmap instead of a byte mask + int32 prefix per candidate row */
int pcgc_kmap_k3_prune_sel(const int32_t* cand_nbr /*[27,n_cand]*/, int64_t n_cand, const uint8_t* bits, const int32_t* wprefix,
                           const int32_t* orig /*[n_out]*/, int64_t n_out, int32_t* nbr /*[dev 27,n_out]*/, void* stream);
int pcgc_kmap_k3_prune_parent_sel(const int32_t* parent_nbr /*[dev 27,n_parent]*/, int64_t n_parent, const uint8_t* bits, const int32_t* wprefix,
                                  const int32_t* orig, int64_t n_out, int32_t* nbr /*[dev 27,n_out]*/, void* stream);
/* out[r] = in[orig[r]]: rows of C floats (C % 4 == 0, leading dimension in_ld) — the surviving feature rows of a pruned level */
int pcgc_gather_rows_f32_ld(const float* in, int C, int in_ld, const int32_t* orig, int64_t n_out, float* out /*[n_out,C]*/, void* stream);
/* rows per batch item: counts[16] (device) <- histogram of coords[:, 0]. */
int pcgc_batch_counts(const int32_t* coords /*[dev n,4]*/, int64_t n, int32_t* counts /*[dev 16]*/, void* stream);

/* ---- canonical ordering: sort_spare_tensor / array2vector (data_utils.py:55-61,91-101; coder.py:97-99):
 *      perm = argsort of (z, y, x, batch) most-significant first. ---- */
size_t pcgc_sort_workspace_bytes(int64_t n);
int pcgc_sort_zyx(const int32_t* coords, int64_t n, int32_t* perm /*[dev n]*/, void* workspace, size_t workspace_bytes,
                  void* stream);
/* the same with the batch index MOST significant: the items of a batch stay contiguous, each in the (z, y, x) order it has when
 * coded alone */
int pcgc_sort_bzyx(const int32_t* coords, int64_t n, int32_t* perm /*[dev n]*/, void* workspace, size_t workspace_bytes, void* stream);
int pcgc_gather_rows_i32x4(const int32_t* in, const int32_t* perm, int64_t n, int32_t* out, void* stream);
int pcgc_gather_rows_f32(const float* in, int C, const int32_t* perm, int64_t n, float* out, void* stream);

/* ---- pointwise operators of the unfused ME-style graph: ME.MinkowskiReLU (autoencoder.py:50) and SparseTensor.__add__ (:55) on
 *      contiguous feature buffers (in place allowed: out == in / out == a).  The fused forward passes do not call them. ---- */
int pcgc_relu(const float* in, int64_t count, float* out, void* stream);
int pcgc_add(const float* a, const float* b, int64_t count, float* out, void* stream);

/* ---- factorized entropy bottleneck (entropy_model.py:82-196) ---- */
/* values = round_half_even(feats); minmax[0]=min, minmax[1]=max (fp32, -0 canonicalised to +0). */
int pcgc_round_minmax(const float* feats, int64_t count, float* minmax /*[dev 2]*/, void* stream);
/* sym = int16(round(feats) - min_v)  (entropy_model.py:161-163). */
int pcgc_symbolize(const float* feats, int64_t count, float min_v, int16_t* sym /*[dev count]*/, void* stream);
/* feats = float(sym) + min_v  (entropy_model.py:193-194). */
int pcgc_desymbolize(const int16_t* sym, int64_t count, float min_v, float* feats, void* stream);
/* the two calls above with the symbol range kept on the device: minmax[2] <- (min, max), sym <- int16(round(feats) - min).
 * The host fetches both with one copy and evaluates the CDF table itself (reference arithmetic, see
 * pcgcv2_amd/entropy_model.py:reference_table). */
int pcgc_quantize_symbols(const float* feats, int64_t count, float* minmax /*[dev 2]*/, int16_t* sym /*[dev count]*/, void* stream);
/* per batch item (its own header range, coder.py:51-55): minmax [dev nseg,2], seg_rows [host nseg] rows of C channels each. */
int pcgc_quantize_symbols_segments(const float* feats, int C, int nseg, const int64_t* seg_rows, float* minmax, int16_t* sym, void* stream);
/* fused CDF table: _likelihood -> clamp(1e-9) -> cumsum -> clamp(1) -> torchac 16-bit normalisation
 * (entropy_model.py:112-149,165-170 + torchac ‡).  params: 352 fp32 packed matrices|biases|factors.
 * cdf_u16 [C, L+1], L = max_v-min_v+1.  cdf_f32 (optional, may be NULL) receives the fp32 cdf [C, L+1]. */
int pcgc_cdf_table(const float* params /*[dev 352]*/, int C, float min_v, float max_v, uint16_t* cdf_u16 /*[dev]*/,
                   float* cdf_f32 /*[dev] or NULL*/, void* stream);


/* ---- range coder, bit-compatible with torchac 0.9.3 ‡ encode_float_cdf / decode_float_cdf
 *      (entropy_model.py:174,192).  HOST functions; symbols row-major [point, channel], one CDF row per channel. ---- */
int64_t pcgc_rc_encode(const uint16_t* cdf /*[host C,Lp]*/, int C, int Lp, const int16_t* sym /*[host n]*/, int64_t n,
                       uint8_t* out /*[host cap]*/, int64_t cap);          /* returns bytes, or -needed if cap too small */
int pcgc_rc_decode(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n);
/* The same stream plus a decoding index.  A range-coded stream is sequential, but only because the decoder state at a later symbol is
 * unknown: pcgc_rc_encode_indexed also returns that state — (first symbol, bit position, low, span - 1, value - low), PCGC_RC_CKPT_WORDS
 * uint32 each — at n_ckpt evenly spread row boundaries, and pcgc_rc_decode_indexed decodes the segments in between on several threads
 * (pcgc_set_rc_threads; 0 = min(8, hardware threads)).  `_F.bin` is bit-identical with or without the index and decodes without it
 * (pcgc_rc_decode, torchac); the host side stores the index in a sidecar file next to it (`_F.idx`, coder.py). */
#define PCGC_RC_CKPT_WORDS 6
int64_t pcgc_rc_encode_indexed(const uint16_t* cdf, int C, int Lp, const int16_t* sym, int64_t n, uint8_t* out, int64_t cap, int n_ckpt,
                               uint32_t* ckpt /*[host n_ckpt][PCGC_RC_CKPT_WORDS]; unused entries have first symbol 0xFFFFFFFF*/);
int pcgc_rc_decode_indexed(const uint16_t* cdf, int C, int Lp, const uint8_t* in, int64_t nbytes, int16_t* sym, int64_t n, int n_ckpt,
                           const uint32_t* ckpt);
int pcgc_set_rc_threads(int threads);
/* The lane-parallel form of the indexed decoder: eight segments per 512-bit register on the calling thread (AVX-512 F/BW/DQ/CD/VL) instead
 * of one or two segments per pool thread.  -1 (default): when the thread budget (pcgc_set_rc_threads) is one or two; 0 never; 1 always.
 * Same symbols either way.  HOST. */
int pcgc_set_rc_lanes(int mode);
/* decoder selection for A/B tests: 0 automatic (AVX-512 boundary count when the host CPU has it and Lp <= 64, else the
 * portable scalar search), 1 portable scalar.  Both are bit-identical. */
int pcgc_set_rc_impl(int impl);

/* ---- native lossless coordinate codec for _C.bin when the external tmc3 binary (gpcc.py:6-41) is absent.
 *      Occupancy-octree + adaptive range coder.  NOT interoperable with G-PCC; flagged by its magic "PCGO". HOST. ---- */
int64_t pcgc_oct_encode(const int32_t* xyz /*[host n,3]*/, int64_t n, uint8_t* out, int64_t cap);   /* bytes or -needed */
int64_t pcgc_oct_decode_count(const uint8_t* in, int64_t nbytes);                                   /* points or <0 */
int pcgc_oct_decode(const uint8_t* in, int64_t nbytes, int32_t* xyz /*[host n,3]*/, int64_t n);
/* clouds of >= 8192 points are coded as up to 8 independent groups of subtrees (stream version 3), encoded and decoded side by side on
 * the threads of pcgc_set_rc_threads; 0 = always the single-stream form (version 2); n > 1 = that many groups (A/B tests).  The
 * decoder reads both versions. */
int pcgc_set_oct_tiled(int on);
/* Context model the ENCODER uses (the decoder reads every version): 1 (default, round 5) = stream versions 4 / 5 — every stream starts from
 * contexts trained on a mix of integer-defined shapes (sphere, ellipsoid, tilted plane, ragged noisy shell: none of them a bench cloud) and
 * a context adapts fast on its first visits; 0 = the round-3 versions 2 / 3 (p = 1/2 / sphere-trained prior), kept so that files written
 * by earlier builds stay covered by the format tests.  Replaces nothing in the reference (gpcc.py:6-41 hands `_C.bin` to tmc3). */
int pcgc_set_oct_model(int model);
/* Builds the octree coder's trained priors now rather than inside the first encode / decode call of the process (a cold decoder's first
 * frame would pay tens of milliseconds).  Idempotent and thread-safe; pcgcv2_amd.Coder calls it at construction.  (pcgc_set_oct_model /
 * pcgc_set_oct_tiled are process-wide A/B and test knobs, not per-call options.)  Replaces nothing in the reference (gpcc.py:6-41). */
int pcgc_oct_warm(void);

/* ---- D1 point-to-point distortion (pc_error.py:27-74 -> mpeg-pcc-dmetric ‡): sum and max over A of the squared distance to
 *      the nearest point of B, B given by its coordinate hash (stride 1).  offsets: int32 [n,4] = (dx,dy,dz,d2) sorted by d2. ---- */
int pcgc_d1_nn(const int32_t* a /*[dev na,4]*/, int64_t na, const uint64_t* b_keys, const int32_t* b_vals, int64_t b_cap,
               const int32_t* offsets /*[dev n_offsets,4]*/, int n_offsets, double* sum /*[dev 1]*/, uint64_t* max_d2 /*[dev 1]*/,
               int32_t* unresolved /*[dev 1]*/, void* stream);
/* The same sums through 4 x 4 x 4 cells of cloud B: cell_keys / cell_vals = coordinate hash of B's stride-4 level (vals = row), masks [n_cells] =
 * 64-bit voxel occupancy per cell (pcgc_d1_cell_masks fills it: bit x & 3 | (y & 3) << 2 | (z & 3) << 4); offsets [n,4] = cell offsets (x, y, z) +
 * the lower bound of the squared distance to any voxel of that cell, ascending; reach2 = squared distance from which a voxel outside the offset
 * table could be nearer (such points count as unresolved).  A few dozen probes per point whatever the distance. */
int pcgc_d1_cell_masks(const int32_t* b /*[dev nb,4]*/, int64_t nb, const uint64_t* cell_keys, const int32_t* cell_vals, int64_t cell_cap,
                       uint64_t* masks /*[dev n_cells]*/, int64_t n_cells, void* stream);
int pcgc_d1_nn_cells(const int32_t* a /*[dev na,4]*/, int64_t na, const uint64_t* cell_keys, const int32_t* cell_vals, int64_t cell_cap,
                     const uint64_t* masks, const int32_t* offsets /*[dev n,4]*/, int n_offsets, int32_t reach2, double* sum, uint64_t* max_d2,
                     int32_t* unresolved, void* stream);

/* ---- ASCII PLY geometry I/O (data_utils.py:19-48: read_ply_ascii_geo / write_ply_ascii_geo), HOST.
 *      read: returns the number of data rows (call with xyz = NULL to size the buffer); same acceptance rule as the
 *      reference (a line is data iff all its ' '-separated tokens parse as floats); columns 0:3 truncated to int. ---- */
int64_t pcgc_ply_read_ascii_geo(const char* path, int32_t* xyz /*[host cap,3] or NULL*/, int64_t cap);
int pcgc_ply_write_ascii_geo(const char* path, const int32_t* xyz /*[host n,3]*/, int64_t n);

/* ---- the bitstream files of several items at once (HOST; native threads): the host half of Coder.encode / Coder.decode
 *      (coder.py:49-55,85-87,93-100) for the items of a collated batch or for one cloud.  stems[i] = "<prefix><postfix>" of item i;
 *      sym: int16 symbols [sum rows, C] (items contiguous), ranges [n,2] = (min_v, max_v) per item, counts [n,3] = (N4, N2, N1),
 *      xyz = stride-8 coordinates / 8, [sum rows, 3].  table_fn = pcgc_reference_table of libpcgc_reftable.so (include/pcgc_reftable.h)
 *      or any function of that signature; eb_params as it expects them.  index_segments = coder.INDEX_SEGMENTS (0: no sidecar).
 *      write_coords = 0: `_C.bin` is left to the caller (tmc3).  threads <= 0: the CPUs this process may use. ---- */
typedef int (*pcgc_table_fn)(const float* params, int C, float min_v, float max_v, uint16_t* table_u16, float* cdf_f32);
int pcgc_items_encode(int n_items, const char* const* stems, const int16_t* sym, const int32_t* xyz, const int64_t* rows, const float* ranges,
                      int C, const int32_t* counts, const float* eb_params, pcgc_table_fn table_fn, int index_segments, int write_coords,
                      int threads);
/* sizes first: rows[i], channels[0], ranges, counts, native_coords[i] (1: `_C.bin` is a native octree stream of rows[i] points) ... */
int pcgc_items_probe(int n_items, const char* const* stems, int64_t* rows, int32_t* channels, float* ranges, int32_t* counts,
                     int32_t* native_coords);
/* ... then the streams: sym [sum rows, C] and the coordinates of the items with native_coords — coord_layout 0: xyz [sum rows, 3], voxel
 * indices in stream order; coord_layout 1: xyz [sum rows, 4] = the coordinate level Coder.decode starts from (coder.py:97-102), rows
 * (item index, coord_scale x, coord_scale y, coord_scale z), every item in (z, y, x) order (sort_spare_tensor, data_utils.py:91-101).
 * -5: a sidecar names another CDF table than this host derives (the stream would decode to noise). */
int pcgc_items_decode(int n_items, const char* const* stems, const int64_t* rows, int C, const float* ranges, const int32_t* native_coords,
                      const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int16_t* sym, int32_t* xyz, int coord_layout,
                      int coord_scale, int threads);

/* The coordinate-only part of the first decoder stage on a freshly decoded level in one call: hash (cap = pcgc_hash_capacity(n)), k3 map
 * nbr [27][n], children level [8 n][4] (MinkowskiGenerativeConvolutionTranspose's coordinates, autoencoder.py:155-161) and its k3 map
 * [27][8 n] — pcgc_hash_clear + _insert + pcgc_kmap_k3 + pcgc_coords_children + pcgc_kmap_k3_children without the host in between. */
int pcgc_level_prepare_children(const int32_t* coords, int64_t n, int32_t stride, uint64_t* keys, int32_t* vals, int64_t cap, int32_t* nbr,
                                int32_t* children, int32_t* nbr_children, void* stream);

/* One cloud in one call — Coder.decode's host half (coder.py:93-104): probe + both streams, into buffers the caller keeps (pinned memory:
 * both uploads are then asynchronous copies).  sym [cap_rows, C], level [cap_rows, 4] (coord_layout 1 above, coord_scale = the level's
 * tensor stride).  info[6] = rows, channels, N4, N2, N1, native_coords; range[2] = min_v, max_v.  -> 0; 1: cap_rows too small (info[0]
 * rows needed, nothing decoded); < 0: error (-5 as above).  `_C.bin` not a native stream (tmc3): info[5] = 0, `level` untouched. */
int pcgc_frame_decode(const char* stem, int C, const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int coord_scale,
                      int64_t cap_rows, int16_t* sym, int32_t* level, int64_t* info, float* range, int threads);
/* The same call in two halves: `_begin` returns as soon as the coordinate level is in `level` (info / range filled; 1 = buffers too small,
 * nothing pending) while the feature stream keeps decoding on the library's threads; the caller uploads the level and enqueues the decoder's
 * coordinate-only kernels (pcgc_level_prepare_children), then `_end` waits for `sym` and returns pcgc_frame_decode's code.  Every
 * successful `_begin` must be followed by `_end` on the same thread.  With fewer than three CPUs, or while another thread's frame is
 * pending, `_begin` does everything synchronously and `_end` nothing. */
int pcgc_frame_decode_begin(const char* stem, int C, const float* eb_params, pcgc_table_fn table_fn, int use_sidecar, int coord_scale,
                            int64_t cap_rows, int16_t* sym, int32_t* level, int64_t* info, float* range, int threads);
int pcgc_frame_decode_end(void);
/* Test hook of the two-halves form: the library's frame worker takes every job `delay_us` late (0 = off; negative = leave unchanged)
 * -> how many feature-stream jobs `_end` has run on the calling thread because the worker had not taken them yet. */
int pcgc_frame_worker_test(int delay_us);

/* The CDF-table cache behind pcgc_items_encode / pcgc_items_decode / pcgc_frame_decode (a table is a pure function of the entropy
 * parameters and the symbol range; the reference evaluates it in every compress() and decompress(), entropy_model.py:165-171,185-190).
 * mode 0: drop every cached table; 1: cache (default); -1: never cache.  -> tables dropped.  HOST. */
int pcgc_table_cache(int mode);

/* zlib's crc32(crc, buf, len) (the CRC-32 of the `_F.idx` sidecar's stream and table guards; coder.py of this package uses
 * zlib.crc32 for the same fields), folded with carry-less multiplies on long buffers.  HOST. */
uint32_t pcgc_crc32(uint32_t crc, const uint8_t* buf, int64_t len);

#ifdef __cplusplus
}
#endif
#endif /* PCGC_HIP_H */
