#!/usr/bin/env python3
"""How much MFMA work of the children-level kernels multiplies all-zero A tiles?  For the decoder's real parent levels (shell10
through the synthetic model): fraction of (16-parent tile, neighbour offset kp) pairs with no present neighbour, weighted by the MFMA
groups that offset feeds (centre 64, face 16, edge 4, corner 1)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.pcc_model import PCCModel
from pcgcv2_amd.coder import Coder
from pcgcv2_amd.sparse import SparseTensor

dev = torch.device('cuda:0')
p = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(p), 1), dtype=torch.int32, device=dev), p], 1).contiguous()
x = SparseTensor(torch.ones((len(p), 1), device=dev), coordinates=c, tensor_stride=1, device=dev)
m = PCCModel().to(dev); m.load_state_dict(synthetic.synthetic_state_dict())
coder = Coder(m, os.path.join(tempfile.mkdtemp(dir='/dev/shm'), 'f'))
seen = {}
orig = ops.conv_child
def spy(parent_nbr, *a, **k):
    seen[parent_nbr.shape[1]] = parent_nbr
    return orig(parent_nbr, *a, **k)
ops.conv_child = spy
import pcgcv2_amd.nn as nn_mod
coder.encode(x); coder.decode(); torch.cuda.synchronize()
w = np.zeros(27)
for kp, jc, reach in ops._halo_cells():
    w[kp] += len(reach)
for n_p, nbr in seen.items():
    nb = (nbr >= 0).cpu().numpy()
    pad = (-n_p) % 16
    nb = np.concatenate([nb, np.zeros((27, pad), bool)], 1).reshape(27, -1, 16)
    any_t = nb.any(2)                       # [27, tiles]
    frac_pairs_absent = 1 - any_t.mean()
    frac_work_absent = 1 - (any_t * w[:, None]).sum() / (w.sum() * any_t.shape[1])
    rows_present = nb.mean()
    print(f'parent level {n_p}: (tile,kp) pairs with no neighbour {frac_pairs_absent:.3f}; MFMA work on all-zero A tiles {frac_work_absent:.3f}; '
          f'mean present neighbours per parent {27 * rows_present:.2f}')
