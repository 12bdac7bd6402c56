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
    only = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else None       # C waves ring: one configuration, 5 launches (PMC runs)
    pts = synthetic.shell('shell10', device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l1 = CoordMap(c4, 1, unique=True)
    l2 = l1.down()[0]; l4 = l2.down()[0]
    for parent, C in ((l2, 16), (l4, 32)):
        if only and only[0] != C: continue
        kids = parent.up()
        n = len(kids)
        x = torch.randn((n, C), device=dev)
        W = torch.randn((27, C, C), device=dev) * 0.05
        b = torch.randn((1, C), device=dev)
        tab = ops.child_conv_table(W)
        if only:
            ops.set_child_tuning(only[1], only[2])
            for _ in range(5): ops.conv_child(parent.k3, x, tab, b, C)
            torch.cuda.synchronize()
            print('pmc run', only)
            continue
        nbr = kids.k3
        ref = ops.conv_gather(nbr, x, W, b)
        us_ref = timeit(lambda: ops.conv_gather(nbr, x, W, b))
        tab = ops.child_conv_table(W)
        print(f'children level of {len(parent)} parents: {n} rows, C={C}: per-row kernel {us_ref:.1f} us')
        for nw, d in ((0, 0), (0, 2), (4, 0), (4, 2), (16, 0), (0, 1)):
            if C == 32 and (nw, d) in ((4, 2), (16, 0)): continue
            if C == 16 and (nw, d) == (0, 1): continue
            ops.set_child_tuning(nw, d)
            got = ops.conv_child(parent.k3, x, tab, b, C)
            ok = torch.equal(got, ref)
            us = timeit(lambda: ops.conv_child(parent.k3, x, tab, b, C))
            print(f'   conv_child waves={nw or "default"} ring={d or "default"}: {us:.1f} us  bit-exact={ok}')
        ops.set_child_tuning(0, 0)


if __name__ == '__main__':
    main()
