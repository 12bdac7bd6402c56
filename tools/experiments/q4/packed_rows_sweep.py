"""Sweep of the packed conv's rows-per-workgroup on the two levels of the bench frame (swapped operand roles): input for pk_rows()."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
def med(f, reps=15):
    for _ in range(3): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
W = torch.randn((27, 64, 64), device=dev) / 40; b = torch.randn((1, 64), device=dev); table = ops.child_conv_table(W)
for name in ('shell10', 'noisy10', 'multi10'):
    pts = synthetic.cloud(name, device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l4 = CoordMap(c4, 1, unique=True).build_pyramid(2); l8 = l4.build_pyramid(1); kids = l8.up()
    for label, lvl in (('enc conv2', l4), ('dec conv0', kids)):
        n = len(lvl); nbr = lvl.k3; x = torch.randn((n, 64), device=dev)
        ops.lib().pcgc_set_packed_tuning(0, 0); auto = med(lambda: ops.conv_packed64(nbr, x, table, b, relu=True))
        row = []
        for R in range(40, 129, 4):
            ops.lib().pcgc_set_packed_tuning(R, 0); row.append((R, med(lambda: ops.conv_packed64(nbr, x, table, b, relu=True), 7)))
        ops.lib().pcgc_set_packed_tuning(0, 0)
        best = min(row, key=lambda t: t[1])
        print(f'{name} {label} n={n} auto {auto:.1f} us best R={best[0]} {best[1]:.1f} | ' + ' '.join(f'{R}:{t:.0f}' for R, t in row), flush=True)
