#!/usr/bin/env python3
"""How much of the C = 64 children-level kernels' time is SIMD quantisation?  Times both InceptionResNet passes on the children of the
stride-8 level of shell10 with the parent set truncated to P parents (neighbours beyond P dropped): 16 384 parents = exactly one
16-parent tile per SIMD (1024 SIMDs), 18 732 = the real level (1171 tiles: 147 SIMDs get a second tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l8 = CoordMap(c4, 1, unique=True).build_pyramid(3)
blk = InceptionResNet(64).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
tabs = ops.child_irn_tables(params)
full = l8.k3
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for P in (8192, 12288, 16384, 16400, 17408, len(l8)):
    nbr = full[:, :P].clone()
    nbr[nbr >= P] = -1
    x = torch.randn((8 * P, 64), device=dev)
    us = timeit(lambda: ops.irn_block_child64(nbr, x, params, tabs))
    print(f'{P:6d} parents ({(P + 15) // 16:5d} tiles): InceptionResNet C=64, both passes {us:7.1f} us')
