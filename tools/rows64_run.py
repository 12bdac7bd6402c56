#!/usr/bin/env python3
"""The C = 64 InceptionResNet block on the two levels it serves in a vox10 frame (encoder stride-4 level: 71 216 rows; decoder first level: 149 856
children rows, through the level's own map), 5 launches each: the workload of tools/rows64_pmc.sh (rocprofv3 --pmc passes).  usage: rows64_run.py [cloud] [pmc|time] [enc|dec|both]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
cloud = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
pts = synthetic.shell(cloud, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l4 = CoordMap(c4, 1, unique=True).down()[0].down()[0]
l8 = l4.down()[0]
kids = l8.up()
blk = InceptionResNet(64).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.child_irn_tables(params)
which = sys.argv[3] if len(sys.argv) > 3 else 'both'
for name, lv in ([('encoder stride-4 level', l4)] if which != 'dec' else []) + ([('decoder first level (children of the stride-8 level)', kids)] if which != 'enc' else []):
    nbr = lv.k3
    n = nbr.shape[1]
    x = torch.randn((n, 64), device=dev)
    P = int((nbr >= 0).sum().item())
    for _ in range(5): ops.irn_block_rows64(nbr, x, params, tabs)
    torch.cuda.synchronize()
    if len(sys.argv) > 2 and sys.argv[2] == 'time':
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.irn_block_rows64(nbr, x, params, tabs)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 2 * P * (64 * 16 + 16 * 32 + 16 * 16) + 2 * n * (64 * 16 + 16 * 32)
        print(f'{name}: {n} rows, {P} pairs ({P / n:.2f} per row): block {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s ({fl / us / 1e6 / 157.3:.3f} of peak)')
    else:
        print('pmc run', name, n, P)
