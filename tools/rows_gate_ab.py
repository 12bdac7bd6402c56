#!/usr/bin/env python3
"""Where the rows kernels (csrc/rows_irn.hip) overtake the gather kernels: C = 64 InceptionResNet block and k3 32 -> 32, by level size."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap, SparseTensor
from pcgcv2_amd.autoencoder import InceptionResNet
from pcgcv2_amd.nn import MinkowskiConvolution
dev = torch.device('cuda:0')
def med(call, reps=20):
    for _ in range(3): call()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
levels = []
for name in ('shell8', 'shell9', 'shell10'):
    pts = synthetic.shell(name, device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    top = CoordMap(c4, 1, unique=True)
    lvl = top
    for d in range(4):
        levels.append(lvl)
        lvl = lvl.build_pyramid(1)
levels = sorted({len(l): l for l in levels}.values(), key=len)
blk = InceptionResNet(64).to(dev); blk32 = InceptionResNet(32).to(dev); conv = MinkowskiConvolution(32, 32, kernel_size=3, stride=1, bias=True, dimension=3).to(dev)
with torch.no_grad():
    print('rows        IRN64 gather  IRN64 rows   conv32 gather  conv32 rows   IRN32 VALU  IRN32 rows   (us)')
    for lvl in levels:
        n = len(lvl)
        if n < 200 or n > 400000: continue
        lvl.k3
        x64 = SparseTensor(torch.randn((n, 64), device=dev), coordinate_map=lvl)
        x32 = SparseTensor(torch.randn((n, 32), device=dev), coordinate_map=lvl)
        res = []
        for on in (False, True):
            ops.ROWS_IRN64, ops.ROWS_IRN64_MIN = on, 1
            res.append(med(lambda: blk(x64)))
        for on in (False, True):
            ops.ROWS_CONV, ops.ROWS_CONV_MIN = on, 1
            res.append(med(lambda: conv(x32, relu=True)))
        for on in (False, True):
            ops.ROWS_IRN32, ops.ROWS_IRN32_MIN, ops.ROWS_IRN32_MAX = on, 1, 1 << 40
            res.append(med(lambda: blk32(x32)))
        print(f'{n:8d}   {res[0]:10.1f}  {res[1]:10.1f}   {res[2]:10.1f}  {res[3]:10.1f}   {res[4]:10.1f}  {res[5]:10.1f}')
