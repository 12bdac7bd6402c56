#!/usr/bin/env python3
"""Do the rows kernels pay for a partly filled last round of tiles?  Pass A / pass B of the C = 64 InceptionResNet on prefixes of the decoder's
first children level: 12 waves x 256 CUs = 3072 waves, 16-row tiles."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
kids = CoordMap(c4, 1, unique=True).build_pyramid(3).up()
nbr_full = kids.k3; n_full = len(kids)
blk = InceptionResNet(64).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
ta, tb = ops.child_irn_tables(params)
x = torch.randn((n_full, 64), device=dev); t = torch.empty((n_full, 32), device=dev); out = torch.empty((n_full, 64), device=dev)
P = [p.data_ptr() for p in params]; s = ops._stream(x)
def med(call, reps=20):
    for _ in range(3): call()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for n in (49152, 50000, 98304, 100000, 147456, n_full):
    nbr = nbr_full[:, :n].contiguous()
    a = med(lambda: ops.lib().pcgc_irn_rows_pass(ops._p(nbr), n, 64, 1, ops._p(x), 64, ops._p(ta), ta.numel() * 4, P[1], P[5], None, None, 0, ops._p(t), 32, s))
    b = med(lambda: ops.lib().pcgc_irn_rows_pass(ops._p(nbr), n, 64, 2, ops._p(t), 32, ops._p(tb), tb.numel() * 4, P[3], P[7], P[9], ops._p(x), 64, ops._p(out), 64, s))
    print(f'{n:7d} rows = {n / 16 / 3072:5.2f} tiles per wave: pass A {a:6.1f} us ({a / n * 1e3:.3f} ns/row)   pass B {b:6.1f} us ({b / n * 1e3:.3f} ns/row)')
