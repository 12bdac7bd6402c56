"""Layer modules with MinkowskiEngine-compatible parameter names/shapes (`kernel`, `bias`) so that the reference's
checkpoints (`torch.load(p)['model']`, coder.py:141-142) load with a strict load_state_dict:
   k=3: kernel [27,Cin,Cout] · k=2: [8,Cin,Cout] · k=1: [Cin,Cout] (2-D) · bias [1,Cout]   (ME ‡ conventions)."""
import math
import torch

from . import dispatch, ops
from .sparse import SparseTensor


class _ConvBase(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, dimension=3):
        super().__init__()
        assert dimension == 3
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        vol = kernel_size ** 3
        shape = (in_channels, out_channels) if vol == 1 else (vol, in_channels, out_channels)
        self.kernel = torch.nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = torch.nn.Parameter(torch.empty((1, out_channels), dtype=torch.float32)) if bias else None
        bound = 1.0 / math.sqrt(vol * in_channels)
        with torch.no_grad():
            self.kernel.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.zero_()

    def extra_repr(self):
        return f'in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}'


class MinkowskiConvolution(_ConvBase):
    """kernel_size 3 / stride 1, kernel_size 1, or kernel_size 2 / stride 2 (the three forms autoencoder.py uses)."""

    def forward(self, x, relu=False, out=None, residual=None):
        k, s = self.kernel_size, self.stride
        cin, cout = self.in_channels, self.out_channels
        plain_out = out is None and residual is None
        if k == 3 and s == 1:
            # which kernel family: ONE table (pcgcv2_amd/dispatch.py)
            level = 'children' if x.cmap.origin is not None and x.cmap.origin[0] == 'children' else 'plain'
            fam = dispatch.select('conv3', (cin, cout), x.F.shape[0], level, extent=x.F.shape[0] * x.F.stride(0) * 4,
                                  unit_input=cin == 1 and x.has_unit_features(), plain_output=plain_out).family
            if fam == 'child':
                # children level (output of a generative transpose): gather through the PARENT level's map, csrc/child_kernels.h
                if cout == 1 and cin == 16 and ops.PATH.CHILD_Q4 and plain_out and not relu and x.F.shape[0] >= 8 * ops.PATH.CHILD_Q4_MIN_PARENTS:
                    # the stride-1 level's classification head in quad-block form (csrc/child_q4.h)
                    y = ops.cls_child_q4(x.cmap.origin[1].k3, x.F, self._table(ops.child_q4_cls_table), self.bias)
                    return SparseTensor(y, coordinate_map=x.cmap)
                table = self._table(ops.child_cls_table if cout == 1 else ops.child_conv_table)
                y = ops.conv_child(x.cmap.origin[1].k3, x.F, table, self.bias, cout, out=out, residual=residual, relu=relu)
                return SparseTensor(y, coordinate_map=x.cmap)
            if fam == 'unit':
                # the codec's first layer on the occupancy indicator (all ones): a sum of kernel slices over the present offsets
                cm = x.cmap
                if cm.mapless_unit_conv() and cout in (4, 8, 16) and not ops.PROFILE.counting:
                    # ... which, on a level of the strided pyramid, are known from the PARENT level's map: this level's own [27][n] map
                    # (read by this layer only) is not built
                    coarse, down = cm.down()
                    y = ops.conv_unit_from_coarse(cm.C, cm.stride, cm._parent_of, coarse.k3, down, self.kernel, self.bias, relu=relu)
                    return SparseTensor(y, coordinate_map=cm)
                return SparseTensor(ops.conv_gather_unit(cm.k3, self.kernel, self.bias, relu=relu), coordinate_map=cm)
            if fam == 'packed':
                # 64 -> 64 (conv2 of the encoder, conv0 of the decoder): present-row packing, accumulators in LDS (csrc/conv_packed.hip)
                y = ops.conv_packed64(x.cmap.k3, x.F, self._table(ops.child_conv_table), self.bias, relu=relu)
                return SparseTensor(y, coordinate_map=x.cmap)
            if fam == 'rows':
                # 32 -> 32 (the encoder's conv1): LDS-resident fragment table, one wave per 16-row tile (csrc/rows_irn.hip)
                y = ops.conv_rows(x.cmap.k3, x.F, self._table(ops.child_conv_table), self.bias, cout, out=out, residual=residual, relu=relu)
                return SparseTensor(y, coordinate_map=x.cmap)
            cmap, nbr = x.cmap, x.cmap.k3
        elif k == 1 and s == 1:
            cmap, nbr = x.cmap, None
        elif k == 2 and s == 2:
            cmap, nbr = x.cmap.down()
            fam = dispatch.select('down', (cin, cout), nbr.shape[1], extent=max(x.F.shape[0] * x.F.stride(0), nbr.shape[1] * cout) * 4,
                                  plain_output=plain_out).family
            if fam == 'rows_down':
                # LDS-resident fragment table, one wave per 16 coarse rows walking the 8 child offsets (csrc/rows_irn.hip)
                y = ops.conv_down_rows(nbr, x.F, self._table(ops.child_conv_table), self.bias, cout, relu=relu)
                return SparseTensor(y, coordinate_map=cmap)
        else:
            raise NotImplementedError(f'MinkowskiConvolution(kernel_size={k}, stride={s}) is not on the PCGCv2 path')
        y = ops.conv_gather(nbr, x.F, self.kernel, self.bias, out=out, residual=residual, relu=relu)
        return SparseTensor(y, coordinate_map=cmap)

    def _table(self, build):
        """the layer's kernel re-laid-out as MFMA B fragments, rebuilt whenever the parameter tensor was replaced or modified; one slot per
        layout (a head whose level sizes straddle a dispatch gate alternates between two layouts: neither evicts the other)"""
        stamp = (self.kernel.data_ptr(), self.kernel._version)
        slots = self.__dict__.setdefault('_child_tables', {})
        hit = slots.get(build.__name__)
        if hit is None or hit[0] != stamp:
            hit = slots[build.__name__] = (stamp, build(self.kernel))
        return hit[1]


class MinkowskiGenerativeConvolutionTranspose(_ConvBase):
    def forward(self, x, relu=False):
        assert self.kernel_size == 2 and self.stride == 2
        src = x.pending_rows()
        if src is not None:                       # a pruned level whose features are still "rows `orig` of the candidates' tensor"
            y = ops.conv_up2(src[0], self.kernel, self.bias, relu=relu, rows=src[1])
        else:
            y = ops.conv_up2(x.F, self.kernel, self.bias, relu=relu)
        return SparseTensor(y, coordinate_map=x.cmap.up())


class MinkowskiPruning(torch.nn.Module):
    """Keep rows where mask is set, order preserved (autoencoder.py:237,247).  n_keep, when the caller knows it
    (top-k), avoids a device->host sync."""

    def forward(self, x, mask, n_keep=None, keep_per_item=None):
        from .sparse import CoordMap
        prefix, total = ops.mask_scan(mask)
        n = int(total.item()) if n_keep is None else int(n_keep)
        coords = ops.compact_coords(x.C, mask, prefix, n)
        # the surviving feature rows are compacted on first use: the last decoder stage only hands on coordinates
        src = x.F
        feats = lambda: ops.compact_feats(src, mask, prefix, n)
        cmap = CoordMap(coords, x.cmap.stride, unique=True, origin=('pruned', x.cmap, mask, prefix))
        if keep_per_item is not None:
            cmap._batch_rows = [int(k) for k in keep_per_item]
        return SparseTensor(feats, coordinate_map=cmap)
