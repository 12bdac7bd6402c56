"""Tensor / IO helpers of the encode/decode path (reference data_utils.py:19-48,55-118)."""
import os
import numpy as np
import torch

from . import ops
from .sparse import SparseTensor, sparse_collate, CoordMap


def read_ply_ascii_geo(filedir):
    """data_utils.py:19-34: every line whose tokens all parse as floats is a data row; keep columns 0:3 as int.
    Parsed natively (pcgc_ply_read_ascii_geo): the reference's per-line Python loop takes seconds at ~10^6 points."""
    from ._lib import lib, PcgcError
    path = os.fsencode(filedir)
    n = int(lib().pcgc_ply_read_ascii_geo(path, None, 0))
    if n == -1:
        raise FileNotFoundError(filedir)
    if n < 0:
        raise PcgcError(f'{filedir}: malformed PLY data rows')
    out = np.empty((n, 3), dtype=np.int32)
    if int(lib().pcgc_ply_read_ascii_geo(path, out.ctypes.data, n)) != n:
        raise PcgcError(f'{filedir}: file changed while reading')
    return out.astype('int')


def write_ply_ascii_geo(filedir, coords):
    """data_utils.py:36-48: ASCII PLY, `property float x/y/z`, integer text (native writer)."""
    from ._lib import lib, PcgcError
    coords = np.ascontiguousarray(np.asarray(coords).astype('int'), dtype=np.int32).reshape(-1, 3)
    if int(lib().pcgc_ply_write_ascii_geo(os.fsencode(filedir), coords.ctypes.data, len(coords))) != 0:
        raise PcgcError(f'cannot write {filedir}')


def array2vector(array, step):
    """data_utils.py:55-61 (host-side; the device path is ops.sort_zyx)."""
    array = torch.as_tensor(array).long().cpu()
    step = int(step)
    return sum(array[:, i] * (step ** i) for i in range(array.shape[-1]))


def isin(data, ground_truth):
    """data_utils.py:63-75: boolean vector, True where a row of `data` (int coordinates [N, D]) occurs in `ground_truth`.  Training-time
    helper of the reference's Decoder.prune_voxel (autoencoder.py:241-243); provided so that `from data_utils import isin, istopk`
    binds — host implementation, as in the reference."""
    dev = data.device
    a, b = torch.as_tensor(data).long().cpu(), torch.as_tensor(ground_truth).long().cpu()
    step = int(max(a.max(), b.max())) + 1
    return torch.isin(array2vector(a, step), array2vector(b, step)).to(dev)


def istopk(data, nums, rho=1.0):
    """data_utils.py:77-89 on device: per batch item b, mask of its int(min(rows_b, nums[b] * rho)) largest values (the reference
    loops over the items on the host; here the items are contiguous row segments of one tensor)."""
    if len(nums) == 1:
        k = int(min(len(data), nums[0] * rho))
        return ops.topk_mask(data.F, k).bool()
    rows = data.cmap.batch_rows
    keep = [int(min(r, n * rho)) for r, n in zip(rows, nums)]
    return ops.topk_mask_segments(data.F, rows, keep).bool()


def sort_spare_tensor(sparse_tensor):
    """data_utils.py:91-101: rows ordered by (z, y, x, batch)."""
    perm = ops.sort_zyx(sparse_tensor.C)
    coords = ops.gather_coords(sparse_tensor.C, perm)
    feats = ops.gather_feats(sparse_tensor.F, perm)
    return SparseTensor(feats, coordinate_map=CoordMap(coords, sparse_tensor.cmap.stride, unique=True))


def load_sparse_tensor(filedir, device):
    """data_utils.py:103-110."""
    coords = torch.tensor(read_ply_ascii_geo(filedir)).int()
    feats = torch.ones((len(coords), 1)).float()
    coords, feats = sparse_collate([coords], [feats])
    return SparseTensor(features=feats, coordinates=coords, tensor_stride=1, device=device)


def scale_sparse_tensor(x, factor):
    """data_utils.py:112-118: (C*factor).round().int() in fp32, then re-collate (dedups)."""
    coords = ops.coords_scale(x.C, factor)
    feats = torch.ones((coords.shape[0], 1), dtype=torch.float32, device=coords.device)
    return SparseTensor(features=feats, coordinates=coords, tensor_stride=1, device=x.device)
