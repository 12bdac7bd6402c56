#!/usr/bin/env python3
"""pcgc_sort_zyx: the one-workgroup radix sort (k_sort_zyx_one) against rocPRIM's launch sequence on the stride-8 levels of the bench clouds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
dev = torch.device('cuda:0')


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name in ('shell10', 'noisy10', 'multi10', 'shell11'):
    pts = synthetic.shell(name, device=dev) if name in synthetic.SHELLS else synthetic.cloud(name).to(dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l8 = CoordMap(c4, 1, unique=True).build_pyramid(3)
    C = l8.C
    ops.set_sort_one_max(0)
    ref = ops.sort_zyx(C)
    us_lib = timeit(lambda: ops.sort_zyx(C))
    ops.set_sort_one_max(1 << 18)
    got = ops.sort_zyx(C)
    us_one = timeit(lambda: ops.sort_zyx(C))
    ops.set_sort_one_max(65536)
    print(f'{name}: {len(C)} stride-8 rows: rocPRIM {us_lib:.1f} us, one workgroup {us_one:.1f} us, same permutation {torch.equal(ref, got)}')
