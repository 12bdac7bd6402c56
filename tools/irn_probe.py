#!/usr/bin/env python3
"""Runs the fused InceptionResNet passes (C = 16) on the decoder's 2 M-row candidate level a few times — a small target for
rocprofv3 --pmc passes (tools/irn_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
l2 = CoordMap(c, 1, unique=True).down()[0]
lvl = l2.up() if C == 16 else l2.down()[0].up()       # the level the decoder runs this width on: 8*N2 (C = 16), 8*N4 (C = 32)
nbr = lvl.k3; n = len(lvl)
Q = C // 4
g = torch.Generator(device='cpu').manual_seed(0)
mk = lambda *s: (torch.randn(s, generator=g) / 30).to(dev)
params = [mk(27, C, Q), mk(1, Q), mk(27, Q, 2 * Q), mk(1, 2 * Q), mk(C, Q), mk(1, Q), mk(27, Q, Q), mk(1, Q), mk(Q, 2 * Q), mk(1, 2 * Q)]
x = torch.randn((n, C), generator=g).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y = ops.irn_block(nbr, x, params)
torch.cuda.synchronize()
print('rows', n, 'checksum', float(y.double().sum()))
