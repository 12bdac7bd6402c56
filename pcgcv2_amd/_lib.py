"""ctypes binding of libpcgc_hip.so (include/pcgc_hip.h).  There is NO fallback: if the HIP library is missing or a
call fails, the product raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCGC_LIB') or os.path.join(_HERE, 'libpcgc_hip.so')       # (PCGC_LIB: an experiment build, pcgcv2_amd/_build.py)

vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
ci = C.c_int

# name -> (restype, argtypes); mirrors include/pcgc_hip.h one to one
SIGNATURES = {
    'pcgc_last_error': (C.c_char_p, []),
    'pcgc_version': (ci, []),
    'pcgc_hash_capacity': (i64, [i64]),
    'pcgc_hash_clear': (ci, [vp, vp, i64, vp]),
    'pcgc_hash_insert': (ci, [vp, i64, i32, vp, vp, i64, vp]),
    'pcgc_hash_insert_policy': (ci, [vp, i64, i32, vp, vp, i64, ci, vp]),
    'pcgc_set_convention': (ci, [ci, ci]),
    'pcgc_hash_first_mask': (ci, [vp, i64, i32, vp, vp, i64, vp, vp, vp]),
    'pcgc_coords_check_order': (ci, [vp, i64, vp, vp]),
    'pcgc_coords_quantize': (ci, [vp, i64, i32, vp, vp]),
    'pcgc_coords_children': (ci, [vp, i64, i32, vp, vp]),
    'pcgc_coords_scale': (ci, [vp, i64, f32, vp, vp]),
    'pcgc_scan_workspace_bytes': (sz, [i64]),
    'pcgc_mask_scan': (ci, [vp, i64, vp, vp, vp, sz, vp]),
    'pcgc_mask_scan_zeroed': (ci, [vp, i64, vp, vp, vp, sz, vp]),
    'pcgc_compact_coords': (ci, [vp, vp, vp, i64, vp, vp]),
    'pcgc_compact_feats': (ci, [vp, ci, ci, vp, vp, i64, vp, vp]),
    'pcgc_kmap_k3': (ci, [vp, i64, i32, vp, vp, i64, vp, vp]),
    'pcgc_kmap_k3_children': (ci, [vp, i64, vp, vp]),
    'pcgc_kmap_k3_prune': (ci, [vp, i64, vp, vp, vp, i64, vp, vp]),
    'pcgc_kmap_k3_prune_parent': (ci, [vp, i64, vp, vp, vp, i64, vp, vp]),
    'pcgc_kmap_k3_from_coarse': (ci, [vp, i64, i32, vp, vp, vp, i64, vp, vp]),
    'pcgc_down_maps': (ci, [vp, vp, vp, i64, i32, i64, vp, vp, vp]),
    'pcgc_down_prepare': (ci, [vp, i64, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, sz, vp]),
    'pcgc_down_finish': (ci, [vp, vp, vp, vp, vp, i64, i32, i64, vp, vp, vp, vp]),
    'pcgc_down_level': (ci, [vp, i64, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
    'pcgc_pyramid_scratch_bytes': (sz, [i64, ci]),
    'pcgc_pyramid': (ci, [vp, i64, i32, ci, vp, sz, vp, vp, vp, vp, vp]),
    'pcgc_compact_index': (ci, [vp, vp, i64, vp, vp]),
    'pcgc_conv_gather': (ci, [vp, ci, i64, vp, i64, ci, ci, ci, vp, vp, vp, ci, ci, ci, vp, ci, ci, ci, vp]),
    'pcgc_conv_gather_unit': (ci, [vp, ci, i64, vp, vp, ci, vp, ci, ci, vp]),
    'pcgc_conv_unit_from_coarse': (ci, [vp, i64, i32, vp, vp, vp, i64, vp, vp, ci, vp, ci, ci, vp]),
    'pcgc_set_conv_impl': (ci, [ci]),
    'pcgc_last_conv_impl': (ci, []),
    'pcgc_set_up2_impl': (ci, [ci]),
    'pcgc_irn_block': (ci, [vp, i64, vp, ci, ci, vp, vp, vp, ci, vp]),
    'pcgc_irn_pass': (ci, [vp, i64, vp, ci, ci, vp, vp, vp, ci, ci, vp]),
    'pcgc_conv_child': (ci, [vp, i64, vp, ci, ci, vp, i64, vp, vp, ci, ci, vp, ci, ci, vp]),
    'pcgc_irn_child_pass': (ci, [vp, i64, ci, ci, vp, ci, vp, i64, vp, vp, vp, vp, ci, vp, ci, vp]),
    'pcgc_irn_child_q4': (ci, [vp, i64, ci, ci, vp, ci, vp, i64, vp, vp, vp, vp, ci, vp, ci, vp]),
    'pcgc_cls_child_q4': (ci, [vp, i64, vp, ci, ci, vp, i64, vp, vp, vp]),
    'pcgc_conv_down_rows': (ci, [vp, i64, vp, i64, ci, ci, vp, i64, vp, ci, vp, ci, ci, vp]),
    'pcgc_conv_rows': (ci, [vp, i64, vp, ci, ci, vp, i64, vp, vp, ci, ci, vp, ci, ci, vp]),
    'pcgc_irn_rows_pass': (ci, [vp, i64, ci, ci, vp, ci, vp, i64, vp, vp, vp, vp, ci, vp, ci, vp]),
    'pcgc_irn_rows_q4_pass': (ci, [vp, i64, ci, ci, vp, ci, vp, i64, vp, vp, vp, vp, ci, vp, ci, vp]),
    'pcgc_set_rows_q4_variant': (ci, [ci]),
    'pcgc_conv_packed64': (ci, [vp, i64, vp, ci, vp, i64, vp, ci, vp, ci, vp]),
    'pcgc_set_packed_tuning': (ci, [ci, ci]),
    'pcgc_conv_up2': (ci, [i64, vp, ci, ci, vp, vp, ci, vp, ci, vp]),
    'pcgc_conv_up2_gather': (ci, [i64, vp, ci, ci, vp, vp, vp, ci, vp, ci, vp]),
    'pcgc_topk_workspace_bytes': (sz, [i64]),
    'pcgc_topk_mask': (ci, [vp, ci, i64, i64, vp, vp, sz, vp]),
    'pcgc_topk_mask_segments': (ci, [vp, ci, ci, vp, vp, vp, vp, sz, vp]),
    'pcgc_topk_select_workspace_bytes': (sz, [i64]),
    'pcgc_topk_select': (ci, [vp, ci, ci, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    'pcgc_kmap_k3_prune_sel': (ci, [vp, i64, vp, vp, vp, i64, vp, vp]),
    'pcgc_kmap_k3_prune_parent_sel': (ci, [vp, i64, vp, vp, vp, i64, vp, vp]),
    'pcgc_gather_rows_f32_ld': (ci, [vp, ci, ci, vp, i64, vp, vp]),
    'pcgc_batch_counts': (ci, [vp, i64, vp, vp]),
    'pcgc_sort_bzyx': (ci, [vp, i64, vp, vp, sz, vp]),
    'pcgc_quantize_symbols_segments': (ci, [vp, ci, ci, vp, vp, vp, vp]),
    'pcgc_sort_workspace_bytes': (sz, [i64]),
    'pcgc_sort_zyx': (ci, [vp, i64, vp, vp, sz, vp]),
    'pcgc_gather_rows_i32x4': (ci, [vp, vp, i64, vp, vp]),
    'pcgc_gather_rows_f32': (ci, [vp, ci, vp, i64, vp, vp]),
    'pcgc_relu': (ci, [vp, i64, vp, vp]),
    'pcgc_add': (ci, [vp, vp, i64, vp, vp]),
    'pcgc_round_minmax': (ci, [vp, i64, vp, vp]),
    'pcgc_symbolize': (ci, [vp, i64, f32, vp, vp]),
    'pcgc_desymbolize': (ci, [vp, i64, f32, vp, vp]),
    'pcgc_quantize_symbols': (ci, [vp, i64, vp, vp, vp]),
    'pcgc_cdf_table': (ci, [vp, ci, f32, f32, vp, vp, vp]),
    'pcgc_d1_cell_masks': (ci, [vp, i64, vp, vp, i64, vp, i64, vp]),
    'pcgc_d1_nn_cells': (ci, [vp, i64, vp, vp, i64, vp, vp, ci, i32, vp, vp, vp, vp]),
    'pcgc_d1_nn': (ci, [vp, i64, vp, vp, i64, vp, ci, vp, vp, vp, vp]),
    'pcgc_rc_encode': (i64, [vp, ci, ci, vp, i64, vp, i64]),
    'pcgc_rc_decode': (ci, [vp, ci, ci, vp, i64, vp, i64]),
    'pcgc_set_rc_impl': (ci, [ci]),
    'pcgc_rc_encode_indexed': (i64, [vp, ci, ci, vp, i64, vp, i64, ci, vp]),
    'pcgc_rc_decode_indexed': (ci, [vp, ci, ci, vp, i64, vp, i64, ci, vp]),
    'pcgc_set_rc_threads': (ci, [ci]),
    'pcgc_set_rc_lanes': (ci, [ci]),
    'pcgc_set_oct_tiled': (ci, [ci]),
    'pcgc_set_oct_model': (ci, [ci]),
    'pcgc_oct_warm': (ci, []),
    'pcgc_oct_encode': (i64, [vp, i64, vp, i64]),
    'pcgc_oct_decode_count': (i64, [vp, i64]),
    'pcgc_oct_decode': (ci, [vp, i64, vp, i64]),
    'pcgc_items_encode': (ci, [ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci]),
    'pcgc_items_probe': (ci, [ci, vp, vp, vp, vp, vp, vp]),
    'pcgc_items_decode': (ci, [ci, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, ci, ci, ci]),
    'pcgc_level_prepare_children': (ci, [vp, i64, i32, vp, vp, i64, vp, vp, vp, vp]),
    'pcgc_frame_decode': (ci, [C.c_char_p, ci, vp, vp, ci, ci, i64, vp, vp, vp, vp, ci]),
    'pcgc_frame_decode_begin': (ci, [C.c_char_p, ci, vp, vp, ci, ci, i64, vp, vp, vp, vp, ci]),
    'pcgc_frame_decode_end': (ci, []),
    'pcgc_frame_worker_test': (ci, [ci]),
    'pcgc_table_cache': (ci, [ci]),
    'pcgc_crc32': (C.c_uint32, [C.c_uint32, vp, i64]),
    'pcgc_ply_read_ascii_geo': (i64, [C.c_char_p, vp, i64]),
    'pcgc_ply_write_ascii_geo': (ci, [C.c_char_p, vp, i64]),
}


class PcgcError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the bound library.  Raises if it is not built: there is no CPU / eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PcgcError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                            '(hipcc --offload-arch=gfx950). The HIP library is mandatory; there is no fallback path.')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


REFTABLE_PATH = os.path.join(_HERE, 'libpcgc_reftable.so')
_reftable = None


def reftable_lib():
    """libpcgc_reftable.so (csrc/reftable.cpp): int pcgc_reference_table(params, C, min_v, max_v, table_u16, cdf_f32)."""
    global _reftable
    if _reftable is None:
        if not os.path.exists(REFTABLE_PATH):
            raise PcgcError(f'{REFTABLE_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"`')
        import torch  # noqa: F401  (libtorch_cpu must be loaded first)
        l = C.CDLL(REFTABLE_PATH)
        l.pcgc_reference_table.restype = ci
        l.pcgc_reference_table.argtypes = [vp, ci, f32, f32, vp, vp]
        l.pcgc_reference_table_clear.restype = ci
        l.pcgc_reference_table_clear.argtypes = []
        _reftable = l
    return _reftable


def check(rc, what=''):
    if rc != 0:
        raise PcgcError(f'{what or "pcgc call"} failed ({rc}): {lib().pcgc_last_error().decode()}')
