#!/usr/bin/env python3
"""What would the C = 32 InceptionResNet passes cost as plain gather convs on the MFMA kernels?  pass A ~ k3 32->16
([conv0_0 | conv1_0 embedded]), pass B ~ k3 16->32 (block-diagonal [conv0_1 | conv1_1 | pad]).  Compared with the fused VALU passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l2 = CoordMap(c, 1, unique=True).down()[0]; l4 = l2.down()[0]
g = torch.Generator(device='cpu').manual_seed(0)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, lvl in (('8N4 (570k)', l4.up()), ('N2 (256k)', l2)):
    nbr = lvl.k3; n = len(lvl)
    for cin, cout in ((32, 16), (16, 32)):
        x = torch.randn((n, cin), generator=g).to(dev); W = (torch.randn((27, cin, cout), generator=g) / 30).to(dev); b = torch.randn((1, cout), generator=g).to(dev)
        res = []
        for label, impl in (('v1 valu', 1), ('v2 mfma', 2)):
            ops.set_conv_impl(impl)
            res.append(f'{label} {timeit(lambda: ops.conv_gather(nbr, x, W, b, relu=True)):6.1f} us')
        ops.set_conv_impl(-1)
        print(f'{name:12s} k3 {cin}->{cout}: ' + '  '.join(res))
    C = 32; Q = 8
    mk = lambda *s: (torch.randn(s, generator=g) / 30).to(dev)
    params = [mk(27, C, Q), mk(1, Q), mk(27, Q, 2 * Q), mk(1, 2 * Q), mk(C, Q), mk(1, Q), mk(27, Q, Q), mk(1, Q), mk(Q, 2 * Q), mk(1, 2 * Q)]
    x = torch.randn((n, C), generator=g).to(dev)
    print(f'{name:12s} fused VALU IRN block (A + B): {timeit(lambda: ops.irn_block(nbr, x, params)):6.1f} us')
