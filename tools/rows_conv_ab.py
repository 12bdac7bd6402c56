#!/usr/bin/env python3
"""k3 32 -> 32 on the encoder's stride-2 level of shell10: the LDS-resident-table rows kernel (waves / ring depth builds) against the gather kernels."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
lvl = CoordMap(c4, 1, unique=True).build_pyramid(1)
n = len(lvl); nbr = lvl.k3
W = torch.randn((27, 32, 32), device=dev) * 0.03; b = torch.randn((1, 32), device=dev)
table = ops.child_conv_table(W)
x = torch.randn((n, 32), device=dev)
def med(call, reps=30):
    for _ in range(3): call()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
print(n, 'rows')
ref = ops.conv_gather(nbr, x, W, b, relu=True)
print(f'gather kernels (dispatcher): {med(lambda: ops.conv_gather(nbr, x, W, b, relu=True)):.1f} us')
for nw, d in ((0, 0), (8, 2), (16, 1), (12, 1)):
    ops.set_child_tuning(nw, d)
    us = med(lambda: ops.conv_rows(nbr, x, table, b, 32, relu=True))
    print(f'rows kernel waves {nw or "default"} depth {d or "default"}: {us:.1f} us  identical: {bool(torch.equal(ops.conv_rows(nbr, x, table, b, 32, relu=True), ref))}')
ops.set_child_tuning(0, 0)
