#!/usr/bin/env python3
"""Does the row ORDER of a level matter for the gather kernels?  Times the fused C=16 IRN passes on the 2 M-row candidate level
built from the stride-2 level in (a) its pipeline order, (b) z-major sorted order, (c) Morton order, (d) random order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l2 = CoordMap(c, 1, unique=True).down()[0]
C2 = l2.C
def morton(cc):
    x, y, z = [(cc[:, i].long() // 2) for i in (1, 2, 3)]
    key = torch.zeros_like(x)
    for b in range(10):
        key |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return torch.argsort(key)
orders = {'pipeline': torch.arange(len(C2), device=dev), 'zyx': ops.sort_zyx(C2).long(), 'morton': morton(C2),
          'random': torch.randperm(len(C2), device=dev)}
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Q = Cc // 4
g = torch.Generator(device='cpu').manual_seed(0)
mk = lambda *s: (torch.randn(s, generator=g) / 30).to(dev)
params = [mk(27, Cc, Q), mk(1, Q), mk(27, Q, 2 * Q), mk(1, 2 * Q), mk(Cc, Q), mk(1, Q), mk(27, Q, Q), mk(1, Q), mk(Q, 2 * Q), mk(1, 2 * Q)]
for name, perm in orders.items():
    lvl = CoordMap(C2[perm].contiguous(), 2, unique=True).up()
    nbr = lvl.k3; n = len(lvl)
    x = torch.randn((n, Cc), generator=g).to(dev)
    for _ in range(2): y = ops.irn_block(nbr, x, params)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = ops.irn_block(nbr, x, params)
    e1.record(); torch.cuda.synchronize()
    print(f'{name:9s} rows {n}: IRN block (pass A + B) {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us')
