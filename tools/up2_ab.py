#!/usr/bin/env python3
"""Generative transpose conv k2 s2 (64 -> 32 on the 71 216-row level, 32 -> 16 on the 255 692-row level of shell10): the three kernel forms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops
dev = torch.device('cuda:0')


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n, cin, cout in ((71216, 64, 32), (255692, 32, 16), (18732, 8, 64)):
    x = torch.randn((n, cin), device=dev)
    W = torch.randn((8, cin, cout), device=dev) * 0.1
    b = torch.randn((1, cout), device=dev)
    rows = torch.arange(n, dtype=torch.int32, device=dev)
    res = {}
    for impl, name in ((2, 'MFMA, fragments in LDS'), (1, 'MFMA, fragments from L2'), (0, 'VALU')):
        ops.set_up2_impl(impl)
        res[impl] = ops.conv_up2(x, W, b, relu=True)
        us = timeit(lambda: ops.conv_up2(x, W, b, relu=True))
        usr = timeit(lambda: ops.conv_up2(x, W, b, relu=True, rows=rows))
        gb = (n * cin + 8 * n * cout) * 4 / 1e9
        print(f'{cin}->{cout} on {n} rows, {name}: {us:.1f} us ({gb / us * 1e6 / 1e3:.2f} TB/s, {2 * 8 * n * cin * cout / us / 1e6:.1f} TFLOP/s), through a row list {usr:.1f} us, '
              f'same bits {torch.equal(res[impl], res[2])}')
    ops.set_up2_impl(2)
