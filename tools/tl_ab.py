#!/usr/bin/env python3
"""A/B: tile-local-map gather conv vs the per-offset kernels, on the decoder's finest candidate level of shell10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap

dev = torch.device('cuda:0')
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c, 1, unique=True); l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
def zsorted(m):
    return CoordMap(ops.gather_coords(m.C, ops.sort_zyx(m.C)), m.stride, unique=True)
levels = {'8N2': zsorted(l2).up(), '8N4': zsorted(l4).up(), '8N8': zsorted(l8).up()}
g = torch.Generator(device='cpu').manual_seed(0)
def timeit(fn, reps=5):
    for _ in range(2): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y
for name, lvl in levels.items():
    nbr = lvl.k3; n = len(lvl)
    t_build, tm = timeit(lambda: ops.TileMap(nbr), 3)
    uc = tm.ucount.float()
    print(f'{name}: n={n} tilemap build {t_build:.1f} us; distinct rows/tile mean {uc.mean().item():.1f} max {int(uc.max().item())} overflow tiles {(uc > 255).float().mean().item() * 100:.2f}%')
    for cout in (1, 4, 16):
        x = torch.randn((n, 16), generator=g).to(dev); W = (torch.randn((27, 16, cout), generator=g) / 20).to(dev); b = torch.randn((1, cout), generator=g).to(dev)
        t0, y0 = timeit(lambda: ops.conv_gather(nbr, x, W, b))
        t1, y1 = timeit(lambda: ops.conv_gather_tl(tm, x, W, b))
        assert torch.equal(y0, y1)
        print(f'   16->{cout:2d}: per-offset {t0:8.1f} us   tile-local {t1:8.1f} us   x{t0 / t1:.2f}')
