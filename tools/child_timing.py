#!/usr/bin/env python3
"""Per-phase cycle breakdown of the children-level kernels.  Needs a library built with PCGC_EXTRA_HIPCC_FLAGS=-DPCGC_CHILD_TIMING."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pcgcv2_amd import ops, synthetic
from pcgcv2_amd._lib import lib
from pcgcv2_amd.sparse import CoordMap
from pcgcv2_amd.autoencoder import InceptionResNet

dev = torch.device('cuda:0')
from pcgcv2_amd._lib import LIB_PATH
L = ctypes.CDLL(LIB_PATH)                  # PCGC_LIB=pcgcv2_amd/libpcgc_hip_timing.so (PCGC_BUILD_VARIANT=timing PCGC_EXTRA_HIPCC_FLAGS=-DPCGC_CHILD_TIMING python -m pcgcv2_amd._build)


def read(fn):
    buf = (ctypes.c_ulonglong * 8)()
    getattr(L, fn)(buf, 1)
    return list(buf)


def report(name, fn, run, n=5):
    read(fn)
    for _ in range(n): run()
    v = read(fn)
    tiles = max(v[3], 1)
    pro, loop, it = v[0] / tiles, v[1] / tiles, v[4] / tiles
    print(f'{name:34s} per tile and wave: prologue {pro:8.0f}  cell loop {loop:8.0f}  epilogue+drain {it - pro - loop:8.0f}  total {it:8.0f} cycles  ({tiles // n} tiles)')


def main():
    pts = synthetic.shell('shell10', device=dev)
    c4 = torch.cat([torch.zeros((len(pts), 1), dtype=torch.int32, device=dev), pts], 1).contiguous()
    l2 = CoordMap(c4, 1, unique=True).down()[0]
    l4 = l2.down()[0]
    for parent, C in ((l2, 16), (l4, 32)):
        n = 8 * len(parent)
        x = torch.randn((n, C), device=dev)
        W = torch.randn((27, C, C), device=dev) * 0.05
        b = torch.randn((1, C), device=dev)
        tab = ops.child_conv_table(W)
        Wc = torch.randn((27, C, 1), device=dev) * 0.05
        tc = ops.child_cls_table(Wc)
        blk = InceptionResNet(C).to(dev)
        params = [p for m in (blk.conv0_0, blk.conv0_1, blk.conv1_0, blk.conv1_1, blk.conv1_2) for p in (m.kernel, m.bias)]
        tabs = ops.child_irn_tables(params)
        P = [p.data_ptr() for p in params]
        t = torch.empty((n, C // 2), device=dev); out = torch.empty((n, C), device=dev)
        pk = parent.k3
        s = torch.cuda.current_stream().cuda_stream
        codes = [int(a) for a in sys.argv[1:]] or [0]
        for d in codes:
            ops.set_child_tuning(d, 0)
            tag = f'C={C} code {d}'
            report(f'conv {tag}', 'pcgc_child_timing', lambda: ops.conv_child(pk, x, tab, b, C))
            report(f'cls {tag}', 'pcgc_child_timing', lambda: ops.conv_child(pk, x, tc, b[:, :1].contiguous(), 1))
            report(f'irn A {tag}', 'pcgc_child_timing_irn', lambda: lib().pcgc_irn_child_pass(pk.data_ptr(), len(parent), C, 1, x.data_ptr(), C, tabs[0].data_ptr(), tabs[0].numel() * 4, P[1], P[5], None, None, 0, t.data_ptr(), C // 2, s))
            report(f'irn B {tag}', 'pcgc_child_timing_irn', lambda: lib().pcgc_irn_child_pass(pk.data_ptr(), len(parent), C, 2, t.data_ptr(), C // 2, tabs[1].data_ptr(), tabs[1].numel() * 4, P[3], P[7], P[9], x.data_ptr(), C, out.data_ptr(), C, s))
        ops.set_child_tuning(0, 0)


if __name__ == '__main__':
    main()
