#!/usr/bin/env python3
"""A/B timing of the children-level (parent-map) kernels against the per-row gather kernels on the decoder levels of shell10."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap

dev = torch.device('cuda:0')


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    only = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else None       # C 0 0 [conv|cls|irn]: that layer only, 5 launches (PMC runs)
    pts = synthetic.shell('shell10', device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l1 = CoordMap(c4, 1, unique=True)
    l2 = l1.down()[0]; l4 = l2.down()[0]
    l8 = l4.down()[0]
    for parent, C in ((l2, 16), (l4, 32)):
        if only and only[0] != C: continue
        kids = parent.up()
        n = len(kids)
        x = torch.randn((n, C), device=dev)
        W = torch.randn((27, C, C), device=dev) * 0.05
        b = torch.randn((1, C), device=dev)
        tab = ops.child_conv_table(W)
        if only:
            what = sys.argv[4] if len(sys.argv) > 4 else 'conv'
            if what == 'conv':
                for _ in range(5): ops.conv_child(parent.k3, x, tab, b, C)
            elif what == 'cls':
                Wc = torch.randn((27, C, 1), device=dev) * 0.05
                tc = ops.child_cls_table(Wc)
                for _ in range(5): ops.conv_child(parent.k3, x, tc, b[:, :1].contiguous(), 1)
            else:
                from pcgcv2_amd.autoencoder import InceptionResNet
                blk = InceptionResNet(C).to(dev)
                params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
                tabs = ops.child_irn_tables(params)
                for _ in range(5): ops.irn_block_child(parent.k3, x, params, tabs)
            torch.cuda.synchronize()
            print('pmc run', only, what)
            continue
        nbr = kids.k3
        ref = ops.conv_gather(nbr, x, W, b)
        us_ref = timeit(lambda: ops.conv_gather(nbr, x, W, b))
        tab = ops.child_conv_table(W)
        print(f'children level of {len(parent)} parents: {n} rows, C={C}: per-row kernel {us_ref:.1f} us')
        got = ops.conv_child(parent.k3, x, tab, b, C)
        ok = torch.equal(got, ref)
        us = timeit(lambda: ops.conv_child(parent.k3, x, tab, b, C))
        print(f'   conv_child: {us:.1f} us  bit-exact={ok}')
        # classification head and fused InceptionResNet
        Wc = torch.randn((27, C, 1), device=dev) * 0.05
        bc = torch.randn((1, 1), device=dev)
        ref = ops.conv_gather(nbr, x, Wc, bc)
        us_ref = timeit(lambda: ops.conv_gather(nbr, x, Wc, bc))
        tc = ops.child_cls_table(Wc)
        ok = torch.equal(ops.conv_child(parent.k3, x, tc, bc, 1), ref)
        us = timeit(lambda: ops.conv_child(parent.k3, x, tc, bc, 1))
        print(f'   cls head: per-row {us_ref:.1f} us, parent map {us:.1f} us  bit-exact={ok}')
        from pcgcv2_amd.autoencoder import InceptionResNet
        blk = InceptionResNet(C).to(dev)
        params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
        with torch.no_grad():
            for p_ in params: p_.normal_(0, 0.1)
        ref = ops.irn_block(nbr, x, params)
        us_ref = timeit(lambda: ops.irn_block(nbr, x, params))
        tabs = ops.child_irn_tables(params)
        ok = torch.equal(ops.irn_block_child(parent.k3, x, params, tabs), ref)
        us = timeit(lambda: ops.irn_block_child(parent.k3, x, params, tabs))
        print(f'   InceptionResNet: per-row {us_ref:.1f} us, parent map {us:.1f} us  bit-exact={ok}')


if __name__ == '__main__':
    main()
