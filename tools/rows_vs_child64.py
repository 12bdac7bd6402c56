#!/usr/bin/env python3
"""C = 64 InceptionResNet on the decoder's first children level (8 x N8 rows of shell10): children-level kernels (parent map) against the rows kernels (the level's own map)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
parent = CoordMap(c4, 1, unique=True).build_pyramid(3)
kids = parent.up(); n = len(kids)
x = torch.randn((n, 64), device=dev)
blk = InceptionResNet(64).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.child_irn_tables(params)
own = kids.k3
def med(run, reps=20):
    for _ in range(3): run()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
a = ops.irn_block_child64(parent.k3, x, params, tabs)
b = ops.irn_block_rows64(own, x, params, tabs)
print(n, 'rows; identical:', bool(torch.equal(a, b)))
print(f'children-level kernels (parent map): {med(lambda: ops.irn_block_child64(parent.k3, x, params, tabs)):.1f} us per block')
print(f'rows kernels (own map):              {med(lambda: ops.irn_block_rows64(own, x, params, tabs)):.1f} us per block')
