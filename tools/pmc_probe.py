#!/usr/bin/env python3
"""Tiny workload for PMC collection: a few launches of the per-offset and tile-local gather convs on one level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pts = synthetic.shell('shell10', device=dev)
c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l2 = CoordMap(c, 1, unique=True).down()[0]
lvl = CoordMap(ops.gather_coords(l2.C, ops.sort_zyx(l2.C)), 2, unique=True).up()
nbr = lvl.k3; n = len(lvl)
tm = ops.TileMap(nbr)
g = torch.Generator().manual_seed(0)
x = torch.randn((n, 16), generator=g).to(dev); W = (torch.randn((27, 16, cout), generator=g) / 20).to(dev); b = torch.randn((1, cout), generator=g).to(dev)
for _ in range(3):
    ops.conv_gather(nbr, x, W, b); ops.conv_gather_tl(tm, x, W, b)
torch.cuda.synchronize()
