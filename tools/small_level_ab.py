#!/usr/bin/env python3
"""A/B of the gather-conv kernel variants on the two small-level convs of the encoder (N8 = 18.7 k rows on shell10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c, 1, unique=True); l2 = l1.down()[0]; l4 = l2.down()[0]; l8, down = l4.down()
g = torch.Generator(device='cpu').manual_seed(0)
cases = [('conv3 k3 32->8 @N8', l8.k3, len(l8), 32, 8, 27), ('down2 k2 64->32 N4->N8', down, len(l4), 64, 32, 8),
         ('down1 k2 32->64 N2->N4', l2.down()[1], len(l2), 32, 64, 8)]
for name, nbr, n_in, cin, cout, K in cases:
    x = torch.randn((n_in, cin), generator=g).to(dev); W = (torch.randn((K, cin, cout), generator=g) / (K * cin) ** .5).to(dev)
    b = torch.randn((1, cout), generator=g).to(dev)
    out = {}
    for impl in (-1, 0, 1, 2):
        ops.set_conv_impl(impl)
        try:
            for _ in range(2): y = ops.conv_gather(nbr, x, W, b, relu=True)
        except Exception as e:
            print(name, impl, 'n/a', e); continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.conv_gather(nbr, x, W, b, relu=True)
        e1.record(); torch.cuda.synchronize()
        out[impl] = (e0.elapsed_time(e1) / 10 * 1e3, y.clone())
    ref = out[0][1]
    print(f'{name:28s} n_out {nbr.shape[1]:7d}: ' + '  '.join(f'impl {k}: {v[0]:6.1f} us{"" if torch.equal(v[1], ref) else " MISMATCH"}' for k, v in out.items()))
ops.set_conv_impl(-1)
