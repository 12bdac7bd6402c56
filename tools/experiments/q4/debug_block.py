import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
import test_gpu_parity as T
from pcgcv2_amd import ops
from pcgcv2_amd.autoencoder import InceptionResNet
for name, prune in (('shell7', None), ('shell6', 1), ('shell8', 7)):
    parent, kids, kc = T._children_level(name, prune)
    n = len(kc); n_p = n // 8
    rng = np.random.default_rng(977 + n)
    blk = InceptionResNet(16).to(T.DEV)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.2))
    x = rng.standard_normal((n, 16)).astype(np.float32)
    params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
    tables = ops.child_irn_tables(params); q4 = ops.child_q4_tables(params)
    xt = T._t(x)
    got = ops.irn_block_child(parent.k3, xt, params, tables, q4_table=q4)
    packed = ops.irn_block_child(parent.k3, xt, params, tables)
    got2 = ops.irn_block_child(parent.k3, xt, params, tables, q4_table=q4)
    d = got != packed
    print(name, prune, 'n_p', n_p, 'k3', tuple(parent.k3.shape), parent.k3.dtype, parent.k3.is_contiguous(), 'mismatch', int(d.sum()), 'of', d.numel(), 'repeat equal', torch.equal(got, got2),
          'rows', int(d.any(1).sum()), 'cols', d.sum(0).tolist())
