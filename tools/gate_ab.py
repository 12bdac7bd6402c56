#!/usr/bin/env python3
"""Where should the size gates of the gather-conv dispatcher sit?  Times every kernel family on the model's layer shapes at level
sizes from 1.5 k to 70 k rows (the levels of small clouds / octant blocks): the encoder's levels of shell7..shell9 + shell10's N8/N4.
python tools/gate_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv2_amd import synthetic, ops
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)

def timeit(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y

levels = []
for name in ('shell7', 'shell8', 'shell9', 'shell10'):
    pts = synthetic.shell(name, device=dev)
    c = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l = CoordMap(c, 1, unique=True)
    for s in (2, 4, 8):
        l = l.down()[0]
        if 1000 <= len(l) <= 80000:
            levels.append((f'{name}/N{s}', l))
levels.sort(key=lambda t: len(t[1]))
IMPLS = [(-1, 'auto'), (0, 'valu'), (1, 'dma'), (2, 'mfma'), (3, 'wlds'), (4, 'pipe'), (6, 'split')]
print('conv k3: us per launch by kernel family (auto = what the dispatcher picks now)')
print(f'{"level":12s} {"rows":>6s} {"shape":>8s} ' + ' '.join(f'{n:>7s}' for _, n in IMPLS))
for lname, lvl in levels:
    n, nbr = len(lvl), lvl.k3
    for cin, cout in ((64, 64), (32, 32), (64, 32), (32, 8), (16, 16)):
        x = torch.randn((n, cin), generator=g).to(dev); W = (torch.randn((27, cin, cout), generator=g) / (27 * cin) ** .5).to(dev)
        b = torch.randn((1, cout), generator=g).to(dev)
        row, ref = [], None
        for impl, _ in IMPLS:
            ops.set_conv_impl(min(impl, 3) if impl < 5 else impl); ops.set_mfma_pipe(1 if impl == 4 else (-1 if impl < 0 else 0))
            us, y = timeit(lambda: ops.conv_gather(nbr, x, W, b, relu=True))
            if ref is None: ref = y.clone()
            assert torch.equal(y, ref)
            row.append(us)
        print(f'{lname:12s} {n:6d} {cin:3d}->{cout:<3d} ' + ' '.join(f'{u:7.1f}' for u in row))
ops.set_conv_impl(-1); ops.set_mfma_pipe(-1)
print('InceptionResNet block: us per block')
for lname, lvl in levels:
    n, nbr = len(lvl), lvl.k3
    for C in (32, 64):
        blk = InceptionResNet(C).to(dev)
        params = [q for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for q in (m.kernel, m.bias)]
        x = torch.randn((n, C), generator=g).to(dev)
        res = {}
        for rows in (0, 64, 32, 16):
            ops.set_irn_rows(rows)
            res[f'valu rows={rows}'], ref = timeit(lambda: ops.irn_block(nbr, x, params))
        ops.set_irn_rows(0)
        if C == 64:
            f = ops.fuse_irn64(params)
            for mode, tag in ((0, 'mfma wlds'), (1, 'mfma pipe')):
                ops.set_mfma_pipe(mode)
                res[tag], y = timeit(lambda: ops.irn_block_mfma64(nbr, x, f))
                assert torch.equal(y, ref)
            ops.set_mfma_pipe(-1)
        print(f'{lname:12s} {n:6d} C={C}: ' + '  '.join(f'{k} {v:6.1f}' for k, v in res.items()))
