#!/usr/bin/env python3
"""A/B the gather-conv kernels (v0 direct loads vs v1 LDS-DMA) per layer shape of the decoder/encoder on shell10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap

dev = torch.device('cuda:0')
pts = synthetic.shell(sys.argv[1] if len(sys.argv) > 1 else 'shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c, 1, unique=True)
l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
levels = {'N1': l1, 'N2': l2, 'N4': l4, 'N8': l8, '8N8': l8.up(), '8N4': l4.up(), '8N2': l2.up()}
shapes = [('8N2', 16, 16), ('8N2', 16, 4), ('8N2', 4, 8), ('8N2', 4, 4), ('8N2', 16, 1), ('8N4', 32, 32), ('8N4', 32, 8), ('8N4', 8, 16),
          ('8N4', 8, 8), ('8N4', 32, 1), ('8N8', 64, 64), ('8N8', 64, 16), ('8N8', 16, 32), ('8N8', 16, 16), ('8N8', 64, 1),
          ('N2', 32, 32), ('N2', 32, 8), ('N2', 8, 16), ('N2', 8, 8), ('N4', 64, 64), ('N4', 64, 16), ('N4', 16, 32), ('N4', 16, 16), ('N8', 32, 8)]
g = torch.Generator(device='cpu').manual_seed(0)
print(f'{"level":>5} {"n":>8} {"cin":>3} {"cout":>4} {"v0_us":>8} {"v1_us":>8} {"speedup":>7}  v1: GB/s(alg)  TF(dense27)')
for name, cin, cout in shapes:
    lvl = levels[name]; n = len(lvl); nbr = lvl.k3
    P = int((nbr >= 0).sum().item())
    x = torch.randn((n, cin), generator=g).to(dev); W = (torch.randn((27, cin, cout), generator=g) / (27 * cin) ** .5).to(dev)
    b = torch.randn((1, cout), generator=g).to(dev)
    res = {}
    for impl in (0, 1, 2, 3):
        ops.set_conv_impl(impl)
        for _ in range(2): y = ops.conv_gather(nbr, x, W, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): y = ops.conv_gather(nbr, x, W, b)
        e1.record(); torch.cuda.synchronize()
        res[impl] = (e0.elapsed_time(e1) / 5 * 1e3, y.clone())
    assert torch.equal(res[0][1], res[1][1]), (name, cin, cout)
    assert torch.equal(res[0][1], res[2][1]), ('mfma', name, cin, cout)
    assert torch.equal(res[0][1], res[3][1]), ('wlds', name, cin, cout)
    us = res[1][0]
    alg = P * cin * 4 + P * 8 + n * cout * 4
    print(f'{name:>5} {n:8d} {cin:3d} {cout:4d} {res[0][0]:8.1f} {us:8.1f} {res[0][0] / us:7.2f}  {alg / us / 1e3:8.0f}  {2 * 27 * n * cin * cout / us / 1e6:6.1f}   mfma {res[2][0]:8.1f} us {2 * 27 * n * cin * cout / res[2][0] / 1e6:6.1f} TF   wlds {res[3][0]:8.1f} us')
ops.set_conv_impl(-1)
