#!/usr/bin/env python3
"""A/B of the MFMA gather-conv variants for 32->32 and 16->16 on the levels where the model runs them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c, 1, unique=True); l2 = l1.down()[0]; l4 = l2.down()[0]
levels = {'N2 (256k)': l2, '8N4 (570k)': l4.up()}
g = torch.Generator(device='cpu').manual_seed(0)
for name, lvl in levels.items():
    nbr = lvl.k3; n = len(lvl)
    for cin, cout in ((32, 32),):
        x = torch.randn((n, cin), generator=g).to(dev); W = (torch.randn((27, cin, cout), generator=g) / 30).to(dev); b = torch.randn((1, cout), generator=g).to(dev)
        res = {}
        for label, impl, pipe in (('v2', 2, 0), ('v2b', 3, 0), ('v2c', 3, 1)):
            ops.set_conv_impl(impl); ops.set_mfma_pipe(pipe)
            for _ in range(2): y = ops.conv_gather(nbr, x, W, b, relu=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): y = ops.conv_gather(nbr, x, W, b, relu=True)
            e1.record(); torch.cuda.synchronize()
            res[label] = (e0.elapsed_time(e1) / 5 * 1e3, y.clone())
        ok = all(torch.equal(v[1], res['v2'][1]) for v in res.values())
        print(f'{name:12s} {cin}->{cout}: ' + '  '.join(f'{k} {v[0]:6.1f} us' for k, v in res.items()) + ('' if ok else '  MISMATCH'))
ops.set_conv_impl(-1); ops.set_mfma_pipe(-1)
