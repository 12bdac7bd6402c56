#!/usr/bin/env python3
"""Classification head 64 -> 1 on the children of the stride-8 level of shell10 (149 856 rows): time of the children-level kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l8 = CoordMap(c4, 1, unique=True).build_pyramid(3)
x = torch.randn((8 * len(l8), 64), device=dev)
Wc = torch.randn((27, 64, 1), device=dev) * 0.05; b = torch.randn((1, 1), device=dev)
tc = ops.child_cls_table(Wc)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = ops.conv_gather(l8.up().k3, x, Wc, b)
got = ops.conv_child(l8.k3, x, tc, b, 1)
print(f'cls 64->1 on {8 * len(l8)} rows: {timeit(lambda: ops.conv_child(l8.k3, x, tc, b, 1)):.1f} us   bit-exact vs per-row kernel: {torch.equal(ref, got)}')
