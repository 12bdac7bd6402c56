"""ONE table: which kernel family serves which (operator, channel shape, level kind, level size).

Rounds 1-3 grew six InceptionResNet implementations, eight gather-conv families and three "rows" kernels, picked by a ladder of
`if`s spread over autoencoder.py, nn.py, ops.py and the native entry point pcgc_conv_gather.  This module is the single place where
that policy is written down.  (Round 5 pruned the library to what this table reaches with its default switches plus ONE comparison
family per operator: three gather families — VALU, MFMA, row-split — and one VALU InceptionResNet pair remain of those.)  `select(op, ...)` returns the entry that applies; nn.MinkowskiConvolution / autoencoder.InceptionResNet
switch on its `family` and nothing else.  For the gather family the native library still walks its own ladder (csrc/conv.hip) — the
table PREDICTS it (`gather_impl`) and tests/test_gpu_parity.py::test_dispatch_table_* read back what was actually launched
(pcgc_last_conv_impl) for every entry, on both sides of every row-count gate, and compare the output with the oracle:
an instantiation that is reachable is an instantiation that has a parity test.

Every family of one operator computes the same canonical fmaf chain: the choice affects speed, never a bit of the result.
Gates are measured (tools/rows_gate_ab.py; numbers in the `why` column).

The A/B switches and row-count gates are the fields of ONE frozen record, `pathconfig.PathConfig`; `select(..., cfg=None)` is a pure function
of it (default: the process-wide `ops.PATH`).  Nothing in the product path replaces that record, and nothing in the product path calls a
process-wide `pcgc_set_*` knob: concurrent coders cannot race on the policy.
"""
from collections import namedtuple

from . import ops

LIMIT = 0xF0000000          # the LDS-DMA kernels address rows with 32-bit buffer offsets: rows * row bytes must stay below this

# op:       'conv3' (k3 s1) | 'irn' (a whole InceptionResNet block) | 'down' (k2 s2) | 'conv1' (k1) | 'up' (generative transpose k2 s2) | 'prune' (top-k + pruning)
# shape:    (cin, cout) pairs the entry serves, or None = any
# level:    'children' (rows 8 p + j of a generative transpose: convs can run through the PARENT level's map) | 'plain' | None = any
# rows:     [rows_min, rows_max) of the level the operator runs on (ints, or names of PathConfig fields)
# switch:   name of the PathConfig boolean that must be on (None = always)
# family:   what nn.py / autoencoder.py switch on
# kernel:   the kernel template(s) behind it
Rule = namedtuple('Rule', 'op shape level rows_min rows_max switch family kernel why')
INF = 1 << 62

TABLE = (
    # ---- k3 s1 convolutions ---------------------------------------------------------------------------------------------------------
    Rule('conv3', ((16, 16), (32, 32), (16, 1), (32, 1), (64, 1)), 'children', 8192, INF, 'CHILD_MFMA', 'child',
         'k_child_conv<1,1> / k_child_conv<2,2,split> / k_child_cls<NB>; 16->1 with CHILD_Q4: k_child_q4<cls> (quad-block 4x4x1 MFMA)',
         'halo gather through the parent map, packed-N fp32 MFMA: conv 16->16 436 -> 259 us on 2.05 M rows, cls 16->1 209 -> 85 (quad-block form: 83)'),
    Rule('conv3', ((1, 16),), None, 0, INF, 'UNIT_INPUT_CONV', 'unit', 'k_conv_unit<16> / k_conv_unit_coarse<16>',
         'all-ones occupancy input (x.has_unit_features()): sum of kernel slices over the present offsets, no feature gathers: 88 -> 55 us'),
    Rule('conv3', ((32, 32),), None, 'ROWS_CONV_MIN', INF, 'ROWS_CONV', 'rows', 'k_rows_conv<2,2>',
         'LDS-resident fragment table, one wave per 16-row tile: 34 vs 51 us at 1-18 k rows, 127 vs 159 at 256 k'),
    Rule('conv3', ((64, 64),), None, 'PACKED_CONV64_MIN', INF, 'PACKED_CONV64', 'packed', 'k_conv_packed64',
         'present rows packed per workgroup tile and offset, accumulators in LDS, B fragments in registers: conv2 (71 k rows) 215 -> 118 us, conv0 (150 k) 335 -> 306 (profiles/r04_conv_packed.md)'),
    Rule('conv3', None, None, 0, INF, None, 'gather', 'pcgc_conv_gather (see GATHER below)', 'every other shape / size'),
    # ---- InceptionResNet blocks ----------------------------------------------------------------------------------------------------
    Rule('irn', (16, 32), 'children', 8192, INF, 'CHILD_MFMA', 'child',
         'k_child_irn_a<C> + k_child_irn_b<C> (C = 32: half units); C = 16 with CHILD_Q4: k_child_q4<pass A> (quad-block 4x4x1 MFMA) + k_child_irn_b<16, T2 gather>',
         'packed-N MFMA passes through the parent map: C = 16 212/170 -> 118/102 us on 2.05 M rows (quad-block pass A: 97/110), C = 32 174/130 -> 100/81 on 570 k'),
    Rule('irn', (64,), None, 'ROWS_IRN64_MIN', INF, 'ROWS_IRN64', 'rows64', 'k_rows_irn_a64<.., RowsPassA64H> + k_rows_irn_b64',
         "through the level's own map, plain or children level (the decoder's 64 -> 64 conv has built it): 65 vs 135 us per block at 1-18 k rows, 103 vs 198 at 71 k; "
         'on the 150 k-row children level 171 vs 197 us for the parent-map form (removed in round 5)'),
    Rule('irn', (32,), None, 'ROWS_Q4_MIN', 'ROWS_IRN32_MAX+1', 'ROWS_Q4', 'rows32q4', 'k_rows_q4_a32 + k_rows_q4_b32',
         'quad-block form (4x4x1 fp32 MFMA, lane = row, no zero column: issued / algorithmic 3.9 -> 2.1 in pass A): 120 -> 88 us per block at 256 k rows; '
         'below ~150 k rows the 64-row tiles are one round of lone waves and the packed-N kernels stay ahead (40 vs 46 us at 71 k)'),
    Rule('irn', (32,), None, 'ROWS_IRN32_MIN', 'ROWS_IRN32_MAX+1', 'ROWS_IRN32', 'rows32', 'k_rows_irn_a32 + k_rows_irn_b32',
         'plain level: 47 vs 74 us per block at 49 k rows, 119 vs 127 at 256 k'),
    Rule('irn', (16, 32, 64), None, 0, INF, 'FUSE_IRN', 'valu', 'k_irn_a_split<C> + k_irn_b_split<C> (C = 16, 32), k_irn_a<64,16> + k_irn_b<64,16>',
         'two fused gather passes on the VALU, 16-row tiles: levels below 1024 rows, children levels below 8192'),
    Rule('irn', None, None, 0, INF, None, 'unfused', 'five MinkowskiConvolution calls + fused epilogues', 'any other channel count'),
    # ---- k2 s2 down convolutions -------------------------------------------------------------------------------------------------------
    Rule('down', ((16, 32), (32, 64), (64, 32)), None, 'ROWS_DOWN_MIN', INF, 'ROWS_DOWN', 'rows_down', 'k_rows_down<NB,NT>',
         'tile = 16 coarse rows walking the 8 child offsets: 121 -> ~65 us for the three down convs of a vox10 frame (rows = COARSE rows)'),
    Rule('down', None, None, 0, INF, None, 'gather', 'pcgc_conv_gather, K = 8', 'other shapes / tiny levels'),
    Rule('conv1', None, None, 0, INF, None, 'gather', 'pcgc_conv_gather, K = 1 (k_conv_gather_valu)', 'k1 convs outside fused blocks'),
    # ---- prune_voxel: top-k + MinkowskiPruning (channel counts of the pruned features: multiples of 4 take the one-sweep form) ----------
    Rule('prune', (8, 16, 32, 64), None, 0, INF, 'ONE_SWEEP_PRUNE', 'select', 'k_topk_init + 3 x k_topk_hist + k_topk_select',
         'thresholds by radix select, then ONE scan writes coordinates, survivor rows and a rank bitmap: 10 -> 5 launches per decoder stage'),
    Rule('prune', None, None, 0, INF, None, 'mask', 'pcgc_topk_mask(_segments) + pcgc_mask_scan + pcgc_compact_*', 'byte mask + int32 prefix per row'),
    Rule('up', None, None, 0, INF, None, 'up2', 'k_conv_up2_mfma<64,32> / <32,16>, k_conv_up2_rows otherwise', 'generative transpose, one kernel per shape'),
)

# pcgc_conv_gather's own ladder (csrc/conv.hip), restated: -> the impl code pcgc_last_conv_impl() reports.
GATHER_IMPL_NAMES = {0: 'k_conv_gather_valu', 2: 'k_conv_gather_mfma', 6: 'k_conv_gather_split (row-split)'}
GATHER_GATES = {'mfma_min_rows': 512, 'split_below_cout16': 40000}


def gather_impl(K, cin, cout, rows, aligned=True):
    """The kernel family pcgc_conv_gather launches in auto mode for a [K, cin, cout] kernel on `rows` output rows (aligned: 16-byte
    aligned rows and weights, tensors below the 32-bit offset limit, a real kernel map)."""
    g = GATHER_GATES
    dma = aligned and K <= 27 and cin in (8, 16, 32, 64)
    split_shape = dma and K == 27 and cin <= 32 and cout in (4, 8, 16)
    split_first = split_shape and (cout <= 8 or rows < g['split_below_cout16'])
    if dma and cin in (16, 32, 64) and cout in (16, 32, 64) and not split_first and rows >= g['mfma_min_rows']:
        return 2
    if split_first:
        return 6
    return 0


def _value(v, cfg):
    """a gate: an int, or the name of a PathConfig field (optionally '+1': an inclusive upper bound turned exclusive)"""
    if isinstance(v, str):
        name, plus, inc = v.partition('+')
        return getattr(cfg, name) + (int(inc) if plus else 0)
    return v


def _switch_on(rule, cfg):
    if rule.switch is None:
        return True
    return bool(getattr(cfg, rule.switch))


def select(op, shape, rows, level='plain', extent=None, contiguous=True, unit_input=False, plain_output=True, cfg=None):
    """-> the first Rule of TABLE that applies.  shape: (cin, cout), or (C,) for 'irn'; rows: rows of the level the operator runs on
    (the COARSE rows for 'down'); level: 'children' | 'plain'; extent: bytes of the largest tensor the kernel would address with 32-bit
    buffer offsets (rows x leading dimension x 4; default rows x width x 4) — beyond LIMIT the generic kernels take over;
    contiguous: the feature tensor is dense; unit_input: the input is the all-ones
    occupancy indicator; plain_output: no `out=` / `residual=` (the unit and down kernels have no fused epilogue for those); cfg: the
    PathConfig to decide by (default: ops.PATH)."""
    cfg = ops.PATH if cfg is None else cfg
    width = shape[0]
    extent = rows * 4 * width if extent is None else extent
    key = tuple(shape) if len(shape) > 1 else shape[0]
    for rule in TABLE:
        if rule.op != op or (rule.shape is not None and key not in rule.shape):
            continue
        if not (_value(rule.rows_min, cfg) <= rows < _value(rule.rows_max, cfg)) or not _switch_on(rule, cfg):
            continue
        if rule.level is not None and level != 'children':
            continue
        fam = rule.family
        if fam in ('packed', 'child') and op == 'conv3' and not plain_output:     # (neither has a residual / `out=` form)
            continue
        if fam in ('child', 'rows', 'rows64', 'rows32', 'rows32q4', 'rows_down', 'packed') and extent >= LIMIT:
            continue                                                     # beyond 32-bit buffer offsets: the generic kernels take it
        if op == 'irn' and fam != 'unfused' and (not cfg.FUSE_IRN or rows * 4 * width >= 0xFFFFFFF0):
            continue
        if fam == 'child' and op == 'irn' and not contiguous:
            continue
        if fam == 'unit' and not (unit_input and plain_output):
            continue
        if fam == 'rows_down' and not plain_output:
            continue
        return rule
    raise ops.PcgcError(f'dispatch: no kernel for {op} {shape} on a {level} level of {rows} rows')


def entries(cfg=None):
    """(rule, gate row counts) for the exhaustive test: every rule with the row counts just inside its window (both ends)."""
    cfg = ops.PATH if cfg is None else cfg
    out = []
    for rule in TABLE:
        lo, hi = _value(rule.rows_min, cfg), _value(rule.rows_max, cfg)
        out.append((rule, [r for r in (lo, hi - 1) if 0 < r < INF]))
    return out


def describe(cfg=None):
    """The table as text (DESIGN.md §5 prints this)."""
    cfg = ops.PATH if cfg is None else cfg
    lines = []
    for r in TABLE:
        lo, hi = _value(r.rows_min, cfg), _value(r.rows_max, cfg)
        rng = f'{lo}..' + ('' if hi >= INF else str(hi - 1))
        lines.append(f'{r.op:6s} {str(r.shape or "any"):48s} {str(r.level or "any"):18s} rows {rng:16s} -> {r.family:9s} {r.kernel}')
    return '\n'.join(lines)
