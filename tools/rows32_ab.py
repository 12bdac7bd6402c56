#!/usr/bin/env python3
"""The encoder's C = 32 InceptionResNet blocks on PLAIN levels (stride-2: 255 692 rows, stride-8: 18 732 rows of shell10): the rows kernels
(k_rows_irn_a32 / _b32, 16x16x4 fp32 MFMA with packed columns) against the quad-block rows kernels (k_rows_q4_a32 / _b32, 4x4x1) when the
library has them, both against the VALU pair (bit-exact check), timed per pass.
usage: rows32_ab.py [cloud]            timing table
       rows32_ab.py cloud pmc IMPL     5 launches of the block on the stride-2 level only (rocprofv3 --pmc runs); IMPL = rows | q4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet

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
    cloud = sys.argv[1] if len(sys.argv) > 1 else 'shell10'
    pmc = len(sys.argv) > 2 and sys.argv[2] == 'pmc'
    impl = sys.argv[3] if len(sys.argv) > 3 else 'rows'
    pts = synthetic.shell(cloud, device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l1 = CoordMap(c4, 1, unique=True)
    l2 = l1.down()[0]; l4 = l2.down()[0]; l8 = l4.down()[0]
    blk = InceptionResNet(32).to(dev)
    params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
    with torch.no_grad():
        for p_ in params: p_.normal_(0, 0.1)
    tabs = ops.rows_irn32_tables(params)
    has_q4 = hasattr(ops, 'irn_block_rows32_q4')
    q4tabs = ops.rows_q4_tables(params) if has_q4 else None
    for lv in ((l2,) if pmc else (l2, l8, l4)):
        n = len(lv)
        nbr = lv.k3
        x = torch.randn((n, 32), device=dev)
        if pmc:
            for _ in range(5):
                if impl == 'q4': ops.irn_block_rows32_q4(nbr, x, params, q4tabs)
                else: ops.irn_block_rows32(nbr, x, params, tabs)
            torch.cuda.synchronize()
            print('pmc run', cloud, impl, n)
            continue
        P = int((nbr >= 0).sum().item())
        ref = ops.irn_block(nbr, x, params)
        got = ops.irn_block_rows32(nbr, x, params, tabs)
        us = timeit(lambda: ops.irn_block_rows32(nbr, x, params, tabs))
        flops = 2 * P * (32 * 8 + 8 * 16 + 8 * 8) + 2 * n * (32 * 8 + 8 * 16)
        print(f'{cloud} level of {n} rows, {P} pairs ({P / n:.2f} per row): rows kernels {us:.1f} us per block = {flops / us / 1e6:.1f} TFLOP/s '
              f'({flops / us / 1e6 / 157.3:.3f} of peak)  bit-exact={torch.equal(got, ref)}')
        if has_q4:
            t_ref = ops.rows32_pass(nbr, x, params, tabs, 1)
            ua = timeit(lambda: ops.rows32_pass(nbr, x, params, tabs, 1))
            ub = timeit(lambda: ops.rows32_pass(nbr, x, params, tabs, 2, t_ref))
            print(f'    packed-N rows kernels: pass A {ua:.1f} us, pass B {ub:.1f} us')
            for v, what in ((1, '8 waves x 2 M tiles x ring 2'), (2, '8 x 1 x 4 paired'), (3, 'A 16 x 1 x 2 / B 12 x 1 x 2'), (0, 'default')):
                ops.set_rows_q4_variant(v)
                got = ops.irn_block_rows32_q4(nbr, x, params, q4tabs)
                ta = ops.rows_q4_pass(nbr, x, params, q4tabs, 1)
                ok = torch.equal(got, ref) and torch.equal(ta, t_ref)
                us = timeit(lambda: ops.irn_block_rows32_q4(nbr, x, params, q4tabs))
                ua = timeit(lambda: ops.rows_q4_pass(nbr, x, params, q4tabs, 1))
                ub = timeit(lambda: ops.rows_q4_pass(nbr, x, params, q4tabs, 2, t_ref))
                print(f'    quad-block ({what}): block {us:.1f} us = {flops / us / 1e6:.1f} TFLOP/s ({flops / us / 1e6 / 157.3:.3f} of peak), pass A {ua:.1f} us, '
                      f'pass B {ub:.1f} us  bit-exact={ok}')
            ops.set_rows_q4_variant(0)

if __name__ == '__main__':
    main()
