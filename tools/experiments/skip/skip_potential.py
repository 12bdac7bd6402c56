#!/usr/bin/env python3
"""How much MFMA work of the children-level kernels is structural zeros that a WAVE-UNIFORM skip could drop?  For every tile of 16 consecutive
parents and every neighbour-parent offset kp: is the neighbour absent for ALL 16 parents?  Reported in halo cells (a kp stands for 8 / 4 / 2 / 1
cells) for the canonical row order, and for the parents sorted by their 27-bit presence mask (the best any regrouping could do)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pcgcv2_amd import synthetic
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
pts = synthetic.cloud(name, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c4, 1, unique=True); l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
# cells per neighbour-parent offset: centre 8, faces 4, edges 2, corners 1
w = torch.tensor([[1, 2, 1][kx] * [1, 2, 1][ky] * [1, 2, 1][kz] for kz in range(3) for ky in range(3) for kx in range(3)], dtype=torch.float64, device=dev)
for lvl, tag in ((l8, 'stride 8 -> 150 k children rows (C = 64)'), (l4, 'stride 4 -> 570 k (C = 32)'), (l2, 'stride 2 -> 2.05 M (C = 16)')):
    nbr = lvl.k3                                         # [27][n_p]
    n = nbr.shape[1]
    pres = (nbr >= 0)                                    # [27][n]
    occ = (pres.double() * w[:, None]).sum() / (64.0 * n)
    def tile_skip(p):
        npad = (n + 15) // 16 * 16
        q = torch.zeros((27, npad), dtype=torch.bool, device=dev); q[:, :n] = p
        any_t = q.view(27, npad // 16, 16).any(2)        # [27][tiles]
        return 1.0 - float((any_t.double() * w[:, None]).sum() / (64.0 * any_t.shape[1]))
    mask = (pres.long() << torch.arange(27, device=dev)[:, None]).sum(0)
    order = torch.argsort(mask)
    print(f'{name} {tag}: {n} parents, cells present per parent {occ * 64:.1f} / 64 = {occ:.3f}; cells skippable per 16-parent tile: '
          f'canonical order {tile_skip(pres):.3f}, parents sorted by presence mask {tile_skip(pres[:, order]):.3f} (bound {1 - occ:.3f}); '
          f'distinct masks {len(torch.unique(mask))}')
    # sorting by mask only INSIDE chunks of consecutive parents (keeps the level's locality; one workgroup could sort a chunk in LDS)
    parts = []
    for chunk in (512, 1024, 2048, 4096, 16384):
        o = torch.arange(n, device=dev)
        key = (o // chunk) * (1 << 27) + mask
        oc = torch.argsort(key)
        parts.append(f'{chunk}: {tile_skip(pres[:, oc]):.3f}')
    print('      chunked sort, skippable by chunk size -> ' + ', '.join(parts))
