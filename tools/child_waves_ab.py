#!/usr/bin/env python3
"""Waves per workgroup of the children-level InceptionResNet passes at C = 16 (2.05 M rows of shell10): pcgc_set_child_tuning(waves, 0)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
parent = CoordMap(c4, 1, unique=True).build_pyramid({16: 1, 32: 2, 64: 3}[C])
kids = parent.up(); n = len(kids)
x = torch.randn((n, C), device=dev)
blk = InceptionResNet(C).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.child_irn_tables(params)
run = (lambda: ops.irn_block_child64(parent.k3, x, params, tabs)) if C == 64 else (lambda: ops.irn_block_child(parent.k3, x, params, tabs))
def med(reps=20):
    for _ in range(3): run()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
print(n, 'rows, C =', C)
ref = None
for nw in [int(a) for a in sys.argv[2:]] or (0, 10, 8, 4):
    ops.set_child_tuning(nw, 0)
    us = med(); out = run()
    if ref is None: ref = out
    print(f'waves {nw or "default"}: block (pass A + pass B) {us:.1f} us  identical: {bool(torch.equal(out, ref))}')
ops.set_child_tuning(0, 0)
