"""The part of MinkowskiEngine's Python surface that PCGCv2 touches, on the HIP operator set — `import pcgcv2_amd.ME as ME`.

The product's own network (pcgcv2_amd/autoencoder.py) fuses ReLU / bias / residual / concat into the producing kernels and never
goes through this module.  It exists so that code written against ME — the reference's `autoencoder.py` as it stands
(`ME.MinkowskiConvolution(in_channels=..., dimension=3)`, `ME.MinkowskiReLU(inplace=True)`, `ME.cat(a, b) + x`,
`ME.MinkowskiPruning()(data, mask)`), `data_utils.py` (`ME.SparseTensor(...)`, `ME.utils.sparse_collate`) — binds to the operators
unmodified.  Same results as the fused graph, bit for bit (tests/test_gpu_parity.py::test_me_facade_unfused_graph_equals_fused).

Not provided (outside the encode/decode path): coordinate_manager / coordinate_map_key plumbing of the
training graph (pcc_model.py:18-23), MinkowskiEngine's other layers."""
import torch

from . import ops
from .nn import MinkowskiConvolution, MinkowskiGenerativeConvolutionTranspose          # noqa: F401  (ME-compatible parameters)
from .nn import MinkowskiPruning as _Pruning
from .sparse import SparseTensor as _SparseTensor, sparse_collate as _sparse_collate

__version__ = '0.5.4-pcgcv2_amd'


class SparseTensor(_SparseTensor):
    """ME.SparseTensor(features=, coordinates=, tensor_stride=, device=) plus the two operators autoencoder.py:55 applies to it."""

    def __add__(self, other):
        if len(self) != len(other) or self.cmap is not other.cmap and not torch.equal(self.C, other.C):
            raise ValueError('SparseTensor + SparseTensor: the operands live on different coordinate levels')
        return SparseTensor(ops.add(self.F, other.F), coordinate_map=self.cmap)

    @property
    def _batchwise_row_indices(self):
        """row indices per batch item (data_utils.py:84 uses it in istopk)"""
        out, off = [], 0
        for r in self.cmap.batch_rows:
            out.append(torch.arange(off, off + r, device=self.device))
            off += r
        return out

    @property
    def decomposed_coordinates(self):
        """per batch item, its coordinates without the batch column (pcc_model.py:30)"""
        out, off = [], 0
        for r in self.cmap.batch_rows:
            out.append(self.C[off:off + r, 1:])
            off += r
        return out


def _wrap(x):
    """operator results are plain sparse tensors; hand them on as ME-style ones"""
    if isinstance(x, SparseTensor):
        return x
    y = SparseTensor.__new__(SparseTensor)
    y.__dict__.update(x.__dict__)
    return y


def _lift(cls):
    class Lifted(cls):
        def forward(self, *a, **k):
            return _wrap(super().forward(*a, **k))
    Lifted.__name__ = Lifted.__qualname__ = cls.__name__
    return Lifted


MinkowskiConvolution = _lift(MinkowskiConvolution)
MinkowskiGenerativeConvolutionTranspose = _lift(MinkowskiGenerativeConvolutionTranspose)


class MinkowskiPruning(_Pruning):
    """ME.MinkowskiPruning()(x, mask): keeps the rows where the boolean mask is set, order preserved (autoencoder.py:237,247)."""

    def forward(self, x, mask):
        m = mask.to(device=x.device)
        m = (m if m.dtype == torch.uint8 else m.to(torch.uint8)).contiguous()
        y = _wrap(super().forward(x, m))
        rows = x.cmap._batch_rows
        if rows is not None and len(rows) > 1:                       # per-item survivors, for the next stage's per-item top-k
            off, keep = 0, []
            cs = torch.cumsum(m.to(torch.int64), 0).cpu()
            for r in rows:
                keep.append(int(cs[off + r - 1] - (cs[off - 1] if off else 0)) if r else 0)
                off += r
            y.cmap._batch_rows = keep
        return y


class MinkowskiReLU(torch.nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return SparseTensor(ops.relu(x.F, inplace=self.inplace and x.F.is_contiguous()), coordinate_map=x.cmap)


def cat(*tensors):
    """ME.cat: channel concatenation of sparse tensors on the same coordinate level (autoencoder.py:55)."""
    first = tensors[0]
    for t in tensors[1:]:
        if len(t) != len(first):
            raise ValueError('ME.cat: the operands live on different coordinate levels')
    return SparseTensor(torch.cat([t.F for t in tensors], dim=1), coordinate_map=first.cmap)


class utils:
    sparse_collate = staticmethod(_sparse_collate)
