#!/usr/bin/env python3
"""Per-phase shader-clock cycles of the plain-rows kernels (csrc/rows_irn.hip) per tile and wave, on the encoder's C = 64 level (71 k rows)
and the decoder's (150 k rows).  Needs the timing build: PCGC_BUILD_VARIANT=timing PCGC_EXTRA_HIPCC_FLAGS=-DPCGC_CHILD_TIMING (pcgcv2_amd/_build.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib, LIB_PATH
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet
dev = torch.device('cuda:0')
L = ctypes.CDLL(LIB_PATH)


def read():
    buf = (ctypes.c_ulonglong * 8)()
    L.pcgc_child_timing_rows(buf, 1)
    return list(buf)


pts = synthetic.shell('shell10', device=dev)
c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
top = CoordMap(c4, 1, unique=True)
l4 = top.down()[0].down()[0]
l8 = l4.down()[0]
kids = l8.up()
for name, lvl in (('encoder level, 71 k rows', l4), ('decoder level, 150 k rows', kids)):
    nbr = lvl.k3; n = len(lvl)
    for C in (64, 32):
        x = torch.randn((n, C), device=dev)
        blk = InceptionResNet(C).to(dev)
        params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
        P = [p.data_ptr() for p in params]
        s = torch.cuda.current_stream().cuda_stream
        if C == 64:
            ta, tb = ops.child_irn_tables(params)
        else:
            ta, tb = ops.rows_irn32_tables(params)
        t = torch.empty((n, C // 2), device=dev); out = torch.empty((n, C), device=dev)
        runs = (('pass A', lambda: lib().pcgc_irn_rows_pass(nbr.data_ptr(), n, C, 1, x.data_ptr(), C, ta.data_ptr(), ta.numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s)),
                ('pass B', lambda: lib().pcgc_irn_rows_pass(nbr.data_ptr(), n, C, 2, t.data_ptr(), C // 2, tb.data_ptr(), tb.numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s)))
        for pname, run in runs:
            for _ in range(3): run()
            read()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            v = read(); tiles = max(v[3], 1)
            pro, loop, drain, tot = v[0] / tiles, v[1] / tiles, v[5] / tiles, v[4] / tiles
            print(f'{name} C={C} {pname}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us | prologue {pro:7.0f}  cell loop {loop:7.0f}  epilogue issue {tot - pro - loop - drain:7.0f}  '
                  f'store drain {drain:7.0f}  total {tot:7.0f}  ({tiles // 5} tiles)', flush=True)
