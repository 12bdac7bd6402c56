"""Encoder / Decoder of PCGCv2 (reference autoencoder.py:7-273) on the HIP operator set.

Same module tree and attribute names as the reference (so state_dict keys match: encoder.conv0, encoder.block0.1.conv1_2,
decoder.conv2_cls, ...), but the forward passes use fused epilogues: ReLU, bias, the InceptionResNet residual add and
the channel concat are written by the producing kernel (ME.cat / MinkowskiReLU / SparseTensor.__add__ each cost ME an
extra HBM round trip)."""
import torch

from . import dispatch, ops
from .nn import MinkowskiConvolution as Conv, MinkowskiGenerativeConvolutionTranspose as UpConv, MinkowskiPruning
from .sparse import CoordMap, SparseTensor


class InceptionResNet(torch.nn.Module):
    """autoencoder.py:7-57: out = cat(conv0_1(relu(conv0_0 x)), conv1_2(relu(conv1_1(relu(conv1_0 x))))) + x."""

    def __init__(self, channels):
        super().__init__()
        c = channels
        self.conv0_0 = Conv(c, c // 4, 3)
        self.conv0_1 = Conv(c // 4, c // 2, 3)
        self.conv1_0 = Conv(c, c // 4, 1)
        self.conv1_1 = Conv(c // 4, c // 4, 3)
        self.conv1_2 = Conv(c // 4, c // 2, 1)

    def forward(self, x):
        c = x.F.shape[1]
        children = x.cmap.origin is not None and x.cmap.origin[0] == 'children'
        # which of the implementations: ONE table (pcgcv2_amd/dispatch.py); every family computes the same fmaf chains
        fam = dispatch.select('irn', (c,), x.F.shape[0], 'children' if children else 'plain', extent=x.F.shape[0] * max(c, x.F.stride(0)) * 4,
                              contiguous=x.F.is_contiguous()).family
        if fam != 'unfused':
            params = [p for m in (self.conv0_0, self.conv0_1, self.conv1_0, self.conv1_1, self.conv1_2) for p in (m.kernel, m.bias)]
            if fam == 'rows64':          # C = 64, LDS-resident fragment table, one wave per 16-row tile through the level's own map (csrc/rows_irn.hip)
                y = ops.irn_block_rows64(x.cmap.k3, x.F, params, self._tables('child', ops.child_irn_tables, params))
            elif fam == 'child':
                # the stride-1 level of a vox10+ frame: pass A in quad-block form (csrc/child_q4.h; smaller levels do not fill its wave slots)
                big = x.F.shape[0] >= 8 * ops.PATH.CHILD_Q4_MIN_PARENTS
                q4 = self._tables('q4', ops.child_q4_tables, params) if (c == 16 and ops.PATH.CHILD_Q4 and big) else None
                y = ops.irn_block_child(x.cmap.origin[1].k3, x.F, params, self._tables('child', ops.child_irn_tables, params), q4_table=q4)
            elif fam == 'rows32q4':      # large plain level, C = 32: the quad-block rows kernels (csrc/rows_q4.hip)
                y = ops.irn_block_rows32_q4(x.cmap.k3, x.F, params, self._tables('rows_q4', ops.rows_q4_tables, params))
            elif fam == 'rows32':        # plain level, C = 32: the rows kernels instead of the VALU passes
                y = ops.irn_block_rows32(x.cmap.k3, x.F, params, self._tables('rows32', ops.rows_irn32_tables, params))
            else:                        # 'valu': two fused gather passes
                y = ops.irn_block(x.cmap.k3, x.F, params)
            return SparseTensor(y, coordinate_map=x.cmap)
        out = torch.empty_like(x.F)
        a = self.conv0_0(x, relu=True)
        self.conv0_1(a, out=out[:, :c // 2], residual=x.F[:, :c // 2])          # cat slot 0 + residual
        b = self.conv1_1(self.conv1_0(x, relu=True), relu=True)
        self.conv1_2(b, out=out[:, c // 2:], residual=x.F[:, c // 2:])          # cat slot 1 + residual
        return SparseTensor(out, coordinate_map=x.cmap)

    def _tables(self, kind, build, params):
        """derived weight tables of one kind, rebuilt whenever a parameter tensor was replaced or modified"""
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault('_derived', {})
        if cache.get(kind, (None, None))[0] != stamp:
            cache[kind] = (stamp, build(params))
        return cache[kind][1]


def make_layer(block, block_layers, channels):
    return torch.nn.Sequential(*[block(channels=channels) for _ in range(block_layers)])


class Encoder(torch.nn.Module):
    """autoencoder.py:68-147."""

    def __init__(self, channels=[1, 16, 32, 64, 32, 8]):
        super().__init__()
        ch = channels
        self.conv0 = Conv(ch[0], ch[1], 3)
        self.down0 = Conv(ch[1], ch[2], 2, 2)
        self.block0 = make_layer(InceptionResNet, 3, ch[2])
        self.conv1 = Conv(ch[2], ch[2], 3)
        self.down1 = Conv(ch[2], ch[3], 2, 2)
        self.block1 = make_layer(InceptionResNet, 3, ch[3])
        self.conv2 = Conv(ch[3], ch[3], 3)
        self.down2 = Conv(ch[3], ch[4], 2, 2)
        self.block2 = make_layer(InceptionResNet, 3, ch[4])
        self.conv3 = Conv(ch[4], ch[5], 3)

    def forward(self, x):
        out0 = self.block0(self.down0(self.conv0(x, relu=True), relu=True))
        out1 = self.block1(self.down1(self.conv1(out0, relu=True), relu=True))
        out2 = self.block2(self.down2(self.conv2(out1, relu=True), relu=True))
        out2 = self.conv3(out2)
        return [out2, out1, out0]


class Decoder(torch.nn.Module):
    """autoencoder.py:150-273 (inference: training=False, ground truth unused)."""

    def __init__(self, channels=[8, 64, 32, 16]):
        super().__init__()
        ch = channels
        for l in range(3):
            setattr(self, f'up{l}', UpConv(ch[l], ch[l + 1], 2, 2))
            setattr(self, f'conv{l}', Conv(ch[l + 1], ch[l + 1], 3))
            setattr(self, f'block{l}', make_layer(InceptionResNet, 3, ch[l + 1]))
            setattr(self, f'conv{l}_cls', Conv(ch[l + 1], 1, 3))
        self.pruning = MinkowskiPruning()

    def prune_voxel(self, data, data_cls, nums, ground_truth=None, training=False):
        """autoencoder.py:239-249 with istopk (data_utils.py:77-89) on device: per batch item b the nums[b] largest logits among
        the item's own rows (contiguous segments, sparse.CoordMap.batch_rows)."""
        if training:
            raise NotImplementedError('training-time pruning (top-k ∪ ground truth) is outside the encode/decode path')
        cand = data.cmap
        rows = [len(data_cls)] if len(nums) == 1 else cand.batch_rows
        if len(rows) != len(nums):
            raise ValueError(f'prune_voxel: {len(nums)} budgets for a batch of {len(rows)} items')
        keep = [int(min(r, max(int(n), 0))) for r, n in zip(rows, nums)]
        if dispatch.select('prune', (data.F.shape[1],), len(data_cls)).family != 'select':
            mask = ops.topk_mask(data_cls.F, keep[0]) if len(nums) == 1 else ops.topk_mask_segments(data_cls.F, rows, keep)
            return self.pruning(data, mask, n_keep=sum(keep), keep_per_item=None if len(nums) == 1 else keep)
        # one sweep (csrc/select.hip, pcgc_topk_select): thresholds by radix select, then ONE scan that decides every row, ranks genuine
        # ties and writes the pruned level — coordinates (derived from the parent level's when the candidates are a children level whose
        # own coordinates were never materialised), the survivors' candidate rows and the rank bitmap the kernel-map derivation reads
        if cand._C is None and cand.origin is not None and cand.origin[0] == 'children':
            parent = cand.origin[1]
            bits, wprefix, orig, coords = ops.topk_select(data_cls.F, rows, keep, parent_coords=parent.C, parent_stride=parent.stride)
        else:
            bits, wprefix, orig, coords = ops.topk_select(data_cls.F, rows, keep, coords=cand.C)
        cmap = CoordMap(coords, cand.stride, unique=True, origin=('selected', cand, bits, wprefix, orig))
        if len(nums) > 1:
            cmap._batch_rows = list(keep)
        src = data.F
        # (the surviving feature rows are gathered on first use: the last decoder stage only hands on coordinates)
        out = SparseTensor(lambda: ops.gather_rows(src, orig), coordinate_map=cmap)
        out._F_rows = (src, orig)
        return out

    def forward(self, x, nums_list, ground_truth_list=(None, None, None), training=False):
        out, cls_list = x, []
        for l in range(3):
            out = getattr(self, f'up{l}')(out, relu=True)
            out = getattr(self, f'conv{l}')(out, relu=True)
            out = getattr(self, f'block{l}')(out)
            cls = getattr(self, f'conv{l}_cls')(out)
            cls_list.append(cls)
            out = self.prune_voxel(out, cls, nums_list[l], ground_truth_list[l], training)
        return cls_list, out
