#!/usr/bin/env python3
"""The three classification heads (k3 conv C -> 1 on a children level) on the decoder levels of a cloud: time per launch and equality with
the per-row gather conv on the level's own map."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'shell10'


def med(f, reps=30):
    for _ in range(5): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


pts = synthetic.cloud(name, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c4, 1, unique=True)
l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
for parent, C in ((l8, 64), (l4, 32), (l2, 16)):
    kids = parent.up(); n = len(kids)
    x = torch.randn((n, C), device=dev)
    W = torch.randn((27, C, 1), device=dev) * 0.05
    b = torch.randn((1, 1), device=dev)
    tab = ops.child_cls_table(W)
    got = ops.conv_child(parent.k3, x, tab, b, 1)
    ok = torch.equal(got, ops.conv_gather(kids.k3, x, W, b))
    us = med(lambda: ops.conv_child(parent.k3, x, tab, b, 1))
    line = f'{name} cls {C} -> 1 on {n} rows ({len(parent)} parents): k_child_cls {us:.1f} us  equal to the per-row conv: {ok}'
    if C == 16:
        t4 = ops.child_q4_cls_table(W)
        ok4 = torch.equal(ops.cls_child_q4(parent.k3, x, t4, b), got)
        line += f' | quad-block form {med(lambda: ops.cls_child_q4(parent.k3, x, t4, b)):.1f} us  equal: {ok4}'
    print(line, flush=True)
