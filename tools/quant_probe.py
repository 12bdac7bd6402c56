#!/usr/bin/env python3
"""k3 64 -> 64 on prefixes of the decoder's first children level: is the gather kernel's time a step function of the workgroup rounds?
(Measured: 291 us at 1024 groups of 128 rows, 339 us at 1031-1171: the step is one extra group per CU on an XCD, and 1171 / 1024 x 291 = 333 —
the kernel is throughput-bound at 2.2-2.3 ns per row, not tail-bound.  Capping the resident workgroups per CU with extra dynamic LDS:
8 or 6 per CU 341 us, 5 / 4 / 3 per CU 416-420 us — occupancy helps, it does not hurt.)"""
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
kids = CoordMap(c4, 1, unique=True).build_pyramid(3).up()
nbr_full = kids.k3; n_full = len(kids)
W = torch.randn((27, 64, 64), device=dev) * 0.02; b = torch.randn((1, 64), device=dev)
x = torch.randn((n_full, 64), device=dev)
def med(call, reps=15):
    for _ in range(3): call()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for n in (65536, 98304, 114688, 131072, 132000, 140000, n_full):
    nbr = nbr_full[:, :n].contiguous()
    us = med(lambda: ops.conv_gather(nbr, x, W, b, relu=True))
    print(f'{n:7d} rows ({n / 128:7.1f} groups of 128): {us:7.1f} us   {us / n * 1e3:.3f} ns/row')
