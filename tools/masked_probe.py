#!/usr/bin/env python3
"""Runs the C = 64 fused InceptionResNet (block-sparse MFMA gather convs) on the decoder's 150 k-row candidate level a few
times — a small target for rocprofv3 --pmc passes (tools/masked_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l8 = CoordMap(c, 1, unique=True).down()[0].down()[0].down()[0]
lvl = l8.up(); nbr = lvl.k3; n = len(lvl)
g = torch.Generator(device='cpu').manual_seed(0)
C = 64; Q = 16
mk = lambda *s: (torch.randn(s, generator=g) / 30).to(dev)
params = [mk(27, C, Q), mk(1, Q), mk(27, Q, 2 * Q), mk(1, 2 * Q), mk(C, Q), mk(1, Q), mk(27, Q, Q), mk(1, Q), mk(Q, 2 * Q), mk(1, 2 * Q)]
f = ops.fuse_irn64(params)
x = torch.randn((n, C), generator=g).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    y = ops.irn_block_mfma64(nbr, x, f)
torch.cuda.synchronize()
print('rows', n, 'checksum', float(y.double().sum()))
