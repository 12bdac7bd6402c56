"""Tensor / IO helpers of the encode/decode path (reference data_utils.py:19-48,55-118)."""
import os
import numpy as np
import torch

from . import ops
from .sparse import SparseTensor, sparse_collate, CoordMap


def read_ply_ascii_geo(filedir):
    """data_utils.py:19-34: every line whose tokens all parse as floats is a data row; keep columns 0:3 as int."""
    with open(filedir, 'rb') as f:
        raw = f.read()
    end = raw.find(b'end_header')
    if end >= 0:
        body = raw[raw.find(b'\n', end) + 1:]
        try:
            first = body[:body.find(b'\n')].split()
            ncol = len(first)
            flat = np.array(body.split(), dtype=np.float64)
            if ncol > 0 and flat.size % ncol == 0:
                return flat.reshape(-1, ncol)[:, 0:3].astype('int')
        except ValueError:
            pass
    data = []                                   # general (slow) path, same acceptance rule as the reference
    for line in raw.decode('utf-8', 'replace').splitlines(keepends=True):
        try:
            vals = [float(v) for v in line.split(' ') if v != '\n']
        except ValueError:
            continue
        data.append(vals)
    return np.array(data)[:, 0:3].astype('int')


def write_ply_ascii_geo(filedir, coords):
    """data_utils.py:36-48: ASCII PLY, `property float x/y/z`, integer text."""
    coords = np.asarray(coords).astype('int')
    head = ('ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n'
            % coords.shape[0])
    body = '\n'.join(' '.join(map(str, row)) for row in coords.tolist())
    with open(filedir, 'w') as f:
        f.write(head)
        if body:
            f.write(body + '\n')


def array2vector(array, step):
    """data_utils.py:55-61 (host-side; the device path is ops.sort_zyx)."""
    array = torch.as_tensor(array).long().cpu()
    step = int(step)
    return sum(array[:, i] * (step ** i) for i in range(array.shape[-1]))


def istopk(data, nums, rho=1.0):
    """data_utils.py:77-89 for batch size 1, on device."""
    k = int(min(len(data), nums[0] * rho))
    return ops.topk_mask(data.F, k).bool()


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
