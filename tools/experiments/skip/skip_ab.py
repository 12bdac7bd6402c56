#!/usr/bin/env python3
"""Children-level kernels on parent levels whose rows are regrouped by neighbour-presence mask (globally, or inside chunks of consecutive
parents), with the skip build (PCGC_LIB=.../libpcgc_hip_skip.so) or the product library.  The regrouping is done by building the parent
level from permuted coordinates: the kernels see ordinary levels."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'shell10'


def med(f, reps=20):
    for _ in range(4): f()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


pts = synthetic.cloud(name, device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
l1 = CoordMap(c4, 1, unique=True); l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
print('library:', os.environ.get('PCGC_LIB', 'product'))
for lvl, C in ((l2, 16), (l4, 32)):
    n = len(lvl)
    mask = ((lvl.k3 >= 0).long() << torch.arange(27, device=dev)[:, None]).sum(0)
    idx = torch.arange(n, device=dev)
    orders = {'canonical': idx, 'chunks of 4096': torch.argsort((idx // 4096) * (1 << 27) + mask), 'chunks of 16384': torch.argsort((idx // 16384) * (1 << 27) + mask),
              'global': torch.argsort(mask)}
    n16 = n // 16 * 16
    def strided(chunk):                                   # inside chunks of `chunk` parents: tile i of the chunk = parents i, i + chunk / 16, i + 2 chunk / 16, ...
        base = torch.arange(0, n16 - n16 % chunk, chunk, device=dev)[:, None]
        inner = torch.arange(chunk, device=dev).view(16, chunk // 16).t().reshape(-1)[None, :]
        o = (base + inner).reshape(-1)
        return torch.cat([o, torch.arange(len(o), n, device=dev)])
    orders = {'canonical': idx, 'strided in 256': strided(256), 'strided in 4096': strided(4096), 'strided in 65536': strided(65536), 'random': torch.randperm(n, device=dev),
              'chunks of 4096': orders['chunks of 4096']}
    blk = InceptionResNet(C).to(dev)
    params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
    with torch.no_grad():
        for p_ in params: p_.normal_(0, 0.1)
    tabs = ops.child_irn_tables(params)
    W = torch.randn((27, C, C), device=dev) * 0.05; b = torch.randn((1, C), device=dev); tab = ops.child_conv_table(W)
    Wc = torch.randn((27, C, 1), device=dev) * 0.05; bc = torch.randn((1, 1), device=dev); tc = ops.child_cls_table(Wc)
    for tag, o in orders.items():
        parent = CoordMap(lvl.C[o].contiguous(), lvl.stride, unique=True)
        kids = parent.up(); rows = len(kids)
        x = torch.randn((rows, C), device=dev)
        pk = parent.k3
        ok = torch.equal(ops.conv_child(pk, x, tab, b, C), ops.conv_gather(kids.k3, x, W, b)) if tag != 'canonical' or True else True
        t_conv = med(lambda: ops.conv_child(pk, x, tab, b, C))
        t_cls = med(lambda: ops.conv_child(pk, x, tc, bc, 1))
        ok2 = torch.equal(ops.irn_block_child(pk, x, params, tabs), ops.irn_block(kids.k3, x, params))
        t_irn = med(lambda: ops.irn_block_child(pk, x, params, tabs))
        print(f'{name} C = {C} {rows} rows, parents {tag:16s}: conv {t_conv:6.1f} us  cls {t_cls:6.1f}  InceptionResNet (packed-N pair) {t_irn:6.1f}   equal to the per-row kernels: {ok} {ok2}', flush=True)
