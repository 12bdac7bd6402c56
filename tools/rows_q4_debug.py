#!/usr/bin/env python3
"""Where the quad-block rows passes differ from the packed-N rows passes (development aid): per-column / per-lane mismatch census."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
pts = synthetic.shell(sys.argv[1] if len(sys.argv) > 1 else 'shell8', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
lv = CoordMap(c4, 1, unique=True).down()[0]
n = len(lv); nbr = lv.k3
blk = InceptionResNet(32).to(dev)
params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
with torch.no_grad():
    for p_ in params: p_.normal_(0, 0.1)
tabs = ops.rows_irn32_tables(params); q4 = ops.rows_q4_tables(params)
x = torch.randn((n, 32), device=dev)
t_ref = ops.rows32_pass(nbr, x, params, tabs, 1)
o_ref = ops.rows32_pass(nbr, x, params, tabs, 2, t_ref)
for v in (3, 1):
    ops.set_rows_q4_variant(v)
    for name, got, ref in (('pass A', ops.rows_q4_pass(nbr, x, params, q4, 1), t_ref), ('pass B', ops.rows_q4_pass(nbr, x, params, q4, 2, t_ref), o_ref)):
        bad = got != ref
        print(f'variant {v} {name}: n={n} mismatched {int(bad.sum())} of {bad.numel()}, max abs diff {float((got - ref).abs().max()):.3e}, nan {int(torch.isnan(got).sum())}')
        if bad.any():
            print('   per column:', bad.sum(0).tolist())
            rows = bad.any(1).nonzero().flatten()
            print('   bad rows', len(rows), 'first', rows[:12].tolist(), 'row % 64 census', torch.bincount(rows % 64, minlength=64).tolist())
            r = int(rows[0])
            print('   row', r, 'got', [round(float(v_), 4) for v_ in got[r]], '\n        ref', [round(float(v_), 4) for v_ in ref[r]])
            rel = ((got - ref).abs() / (ref.abs() + 1e-6))[bad]
            print('   relative diff: median %.3e max %.3e' % (float(rel.median()), float(rel.max())))
