#!/usr/bin/env python3
"""Waves-per-group / ring-depth A/B of the plain-level C = 64 InceptionResNet passes (csrc/rows_irn.hip) on the encoder's stride-4 level of shell10."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
lvl = CoordMap(c4, 1, unique=True).build_pyramid(2)
if len(sys.argv) > 1 and sys.argv[1] == 'decoder':                 # the decoder's first level: the 8 x N8 children rows, through their own map
    lvl = lvl.build_pyramid(1).up()
n = len(lvl)
nbr = lvl.k3
blk = InceptionResNet(64).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
ta, tb = ops.child_irn_tables(params)
x = torch.randn((n, 64), device=dev)
t = torch.empty((n, 32), device=dev); out = torch.empty((n, 64), device=dev)
P = [p.data_ptr() for p in params]
s = ops._stream(x)
def time_pass(ps, reps=30):
    call = (lambda: ops.lib().pcgc_irn_rows_pass(ops._p(nbr), n, 64, 1, ops._p(x), 64, ops._p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, ops._p(t), 32, s)) if ps == 1 else \
           (lambda: ops.lib().pcgc_irn_rows_pass(ops._p(nbr), n, 64, 2, ops._p(t), 32, ops._p(tb), tb.numel() * 4, P[3], P[7], P[9], ops._p(x), 64, ops._p(out), 64, s))
    for _ in range(3): assert call() == 0
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
print(n, 'rows')
for ps, cfgs in ((1, [(0, 0), (8, 2), (12, 1), (6, 2)]), (2, [(0, 0), (15, 1), (12, 1), (8, 4)])):
    ref = None
    for nw, d in cfgs:
        ops.set_child_tuning(nw, d)
        us = time_pass(ps)
        res = (t if ps == 1 else out).clone()
        if ref is None: ref = res
        print(f'pass {"AB"[ps - 1]} waves {nw or "default"} depth {d or "default"}: {us:.1f} us   identical: {bool(torch.equal(res, ref))}')
ops.set_child_tuning(0, 0)
