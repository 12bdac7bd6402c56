#!/usr/bin/env python3
"""A/B of the two 64 -> 64 convs of a vox10 frame: gather family (k_conv_gather_mfma) vs present-row packing (k_conv_packed64)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
pts = synthetic.cloud(name, device=dev) if name in synthetic.CLOUDS else synthetic.shell(name, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l4 = CoordMap(c4, 1, unique=True).build_pyramid(2)
l8 = l4.build_pyramid(1)
kids = l8.up()
W = torch.randn((27, 64, 64), device=dev) / 40
b = torch.randn((1, 64), device=dev)
table = ops.child_conv_table(W)
def med(f, reps=30):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for label, lvl in (('encoder conv2', l4), ('decoder conv0', kids)):
    if len(sys.argv) > 3 and sys.argv[3] not in label:
        continue
    n = len(lvl); nbr = lvl.k3
    x = torch.randn((n, 64), device=dev)
    present = float((nbr >= 0).float().mean()) * 27
    if len(sys.argv) > 2 and sys.argv[2] == 'sweep':
        for R in (128, 120, 112, 104, 98, 96, 94, 88, 80, 72, 64, 60, 56, 48):
            row = []
            for nw in (4, 8):
                ops.lib().pcgc_set_packed_tuning(R, nw)
                row.append(med(lambda: ops.conv_packed64(nbr, x, table, b, relu=True), 9))
            print(f'  R {R}: 4 waves {row[0]:.1f} us, 8 waves {row[1]:.1f} us')
        ops.lib().pcgc_set_packed_tuning(0, 0)
    for rnd in range(2):
        g = med(lambda: ops.conv_gather(nbr, x, W, b, relu=True))
        p = med(lambda: ops.conv_packed64(nbr, x, table, b, relu=True))
        same = torch.equal(ops.conv_gather(nbr, x, W, b, relu=True), ops.conv_packed64(nbr, x, table, b, relu=True))
        print(f'{label}: {n} rows, {present:.1f} neighbours per row: gather {g:.1f} us, packed {p:.1f} us, identical {same}')
if len(sys.argv) > 2 and sys.argv[2] == 'gate':
    base = l4.C
    for rows in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        lvl = CoordMap(base[:rows].contiguous(), l4.stride, unique=True)
        nbr = lvl.k3
        x = torch.randn((rows, 64), device=dev)
        g = med(lambda: ops.conv_gather(nbr, x, W, b, relu=True)); p = med(lambda: ops.conv_packed64(nbr, x, table, b, relu=True))
        print(f'gate: {rows} rows: gather {g:.1f} us, packed {p:.1f} us')
